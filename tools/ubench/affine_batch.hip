// tools/ubench/affine_batch.hip -- kill-criterion experiment for batched-affine bucket accumulation (VERDICT r1 "next" item 7).
//
// k_accumulate adds points into XYZZ accumulators: 8 M + 2 S = 3,416 v_mad_u64_u32 per mixed add, ~6.5 G adds/s on MI355X (97 % of the VALU issue
// bound of its instruction mix).  An AFFINE add needs only 1 S + 2 M once 1/(x2 - x1) is known, and Montgomery's trick shares ONE inversion
// over a batch: per add  1 M (running product) + 2 M (un-batching) + 1 M (lambda) + 1 S + 1 M = 5 M + 1 S (~2,180 mads) + inversion / B.
// On a GPU the inversion is paid in LANE TIME (the 64 lanes of a wave run it in lock-step whether one or all of them need it), so it only
// amortises over the B additions of the SAME lane: a Fermat inversion is ~377 S + ~190 M, so B must be in the hundreds, and B running products
// (56 B each) plus the B point pairs no longer fit registers or LDS: they stream through HBM (this kernel: 2 x 224 B read + 56 B written in pass 1,
// 56 B + 224 B read + 112 B written in pass 2 = ~900 B per add, against 112 B per add for k_accumulate, whose accumulator never leaves registers).
//
// This benchmark is the UPPER BOUND of what such a kernel can reach: no bucket logic, no sorting, no gather (operands are coalesced arrays),
// perfectly regular batches.  It reports additions per second for B = 64 ... 2048 next to the XYZZ mixed-add loop of k_accumulate on the same
// operands.  Keep criterion (VERDICT): >= 20 % over the XYZZ loop.  Result on MI355X: profiles/r02_affine_batch.md.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aes_zero_knowledge_proof_circuit_amd/csrc tools/ubench/affine_batch.hip -o /tmp/affine_batch && /tmp/affine_batch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ec28.cuh"
using namespace zk;
using P = Fq377P;
using G = Fp28<P>;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// a^(p-2) by square-and-multiply over the bits of p - 2 (Fermat): ~377 squarings + ~190 products, values stay < 1.2 p
__device__ __noinline__ G inverse28(const G &a) {
    uint32_t e[12];
#pragma unroll
    for (int i = 0; i < 12; i++) e[i] = P::mod(i);
    e[0] -= 2;                                   // p is odd and p mod 2^32 = 1 for BLS12-377: borrow-free only if low word >= 2 ...
    if (P::mod(0) < 2) { e[0] = P::mod(0) + 0xfffffffeu; int k = 1; while (e[k] == 0) { e[k] = 0xffffffffu; k++; } e[k] -= 1; }
    G r = G::k_2_392();
    for (int i = 12 * 32 - 1; i >= 0; i--) {
        r = r.sqr();
        if ((e[i >> 5] >> (i & 31)) & 1) r = r * a;
    }
    return r;
}

// element (i, t) of a lane-interleaved array: consecutive lanes touch consecutive structs (coalesced 16-byte accesses)
#define AT(arr, i, t, T) arr[(size_t)(i) * (T) + (t)]

// pass 1 + inversion + pass 2 in one kernel; one lane owns B independent additions  R_i = P_i + Q_i
template <int DUMMY>
__global__ void __launch_bounds__(64, 2) k_affine_batch(const Affine28<P> *__restrict__ Pp, const Affine28<P> *__restrict__ Qp, G *__restrict__ prefix, Affine28<P> *__restrict__ R,
                                                        uint32_t T, int B) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    G acc = G::k_2_392();
    for (int i = 0; i < B; i++) {
        G d = AT(Qp, i, t, T).x.template sub<2>(AT(Pp, i, t, T).x);        // x2 - x1 + 2p
        AT(prefix, i, t, T) = acc;
        acc = acc * d;
    }
    G inv = inverse28(acc);
    for (int i = B - 1; i >= 0; i--) {
        Affine28<P> p = AT(Pp, i, t, T), q = AT(Qp, i, t, T);
        G d = q.x.template sub<2>(p.x);
        G di = inv * AT(prefix, i, t, T);                                  // 1 / (x2 - x1)
        inv = inv * d;
        G lam = q.y.template sub<2>(p.y) * di;
        G x3 = (lam.sqr().template sub<2>(p.x)).template sub<2>(q.x);
        G y3 = (lam * p.x.template sub<7>(x3)).template sub<2>(p.y);
        Affine28<P> r; r.x = x3; r.y = y3;
        AT(R, i, t, T) = r;
    }
}

// the loop body of k_accumulate on the same operands: one XYZZ accumulator per lane, B mixed adds
__global__ void __launch_bounds__(64, 2) k_xyzz_loop(const Affine28<P> *__restrict__ Pp, const Affine28<P> *__restrict__ Qp, Acc28<P> *__restrict__ out, uint32_t T, int B) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    Acc28<P> acc;
    Affine28<P> p0 = AT(Pp, 0, t, T);
    acc.x = p0.x; acc.y = p0.y; acc.zz = G::k_2_392(); acc.zzz = acc.zz;
    Affine28<P> nxt = AT(Qp, 0, t, T);
    for (int i = 0; i < B; i++) {
        Affine28<P> q = nxt;
        if (i + 1 < B) nxt = AT(Qp, i + 1, t, T);
        madd28(acc, q);
    }
    out[t] = acc;
}

int main(int argc, char **argv) {
    const size_t total = argc > 1 ? (size_t)atoll(argv[1]) : ((size_t)1 << 24);      // additions per launch
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Affine28<P> *dP, *dQ, *dR; G *dpre; Acc28<P> *dacc;
    CK(hipMalloc(&dP, total * sizeof(Affine28<P>))); CK(hipMalloc(&dQ, total * sizeof(Affine28<P>))); CK(hipMalloc(&dR, total * sizeof(Affine28<P>)));
    CK(hipMalloc(&dpre, total * sizeof(G))); CK(hipMalloc(&dacc, total / 16 * sizeof(Acc28<P>)));
    {   // arbitrary field elements (28-bit limbs, top limb small): the timing does not depend on the values; x differences are non-zero
        std::vector<Affine28<P>> h(total);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        auto nx = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
        for (int pass = 0; pass < 2; pass++) {
            for (size_t i = 0; i < total; i++) { for (int k = 0; k < 14; k++) { h[i].x.l[k] = nx() & (k == 13 ? 0xfff : G::MASK); h[i].y.l[k] = nx() & (k == 13 ? 0xfff : G::MASK); } }
            CK(hipMemcpy(pass ? dQ : dP, h.data(), total * sizeof(Affine28<P>), hipMemcpyHostToDevice));
        }
    }
    printf("additions per launch: %zu\n", total);
    {
        const int B = 64;
        uint32_t T = (uint32_t)(total / B);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_xyzz_loop, dim3((T + 63) / 64), dim3(64), 0, 0, dP, dQ, dacc, T, B);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("XYZZ mixed-add loop (k_accumulate body), %d adds per lane : %8.3f ms  %6.2f G adds/s\n", B, ms, total / ms / 1e6);
    }
    for (int B : {64, 256, 1024, 2048}) {
        uint32_t T = (uint32_t)(total / B);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_affine_batch<0>, dim3((T + 63) / 64), dim3(64), 0, 0, dP, dQ, dpre, dR, T, B);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("batched affine, B = %4d adds per lane per inversion (%u lanes) : %8.3f ms  %6.2f G adds/s\n", B, T, ms, total / ms / 1e6);
    }
    return 0;
}
