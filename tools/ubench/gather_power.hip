// tools/ubench/gather_power.hip -- does the MEMORY side of k_accumulate<EdwardsLaw> cost it clock?  (round 6)
//
// tools/ubench/issue_cycles.hip found the hot loop at the same ~13,380 shader cycles per bucket addition per SIMD whether its 192-byte records come from L2 or from 3.2 GB of
// HBM -- the gather latency is hidden -- but at 1.85 GHz instead of 2.04-2.10 GHz: the chip gives back clock when 1.9 TB/s of random gathers are live.  This probe runs the
// production launch shape (8192 single-wave workgroups, 83 additions per lane, te_madd_hot's arithmetic) over record layouts / load policies that change what the memory
// system does per gather, and reports kernel time, shader cycles per addition and the effective shader clock (s_memtime / s_memrealtime) for each:
//   l2        4096 records (L2-resident): the no-HBM reference
//   s192      64-byte aligned 192-byte records (production layout: three lines per gather)
//   s256      the same records on a 256-byte stride (a gather never straddles a 256-byte interleave block)
//   s192nt    production layout, non-temporal loads (records are touched once: no reason to allocate them in L2 / MALL)
//   s256nt    both
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aes_zero_knowledge_proof_circuit_amd/csrc tools/ubench/gather_power.hip -o tools/ubench/gather_power.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "te28.cuh"
using namespace zk;
using P = Fq377P;
using G = FpMsm<P>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stamp { uint64_t cyc, rt; };
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ G load_coord(const uint32_t *w) {      // 56 bytes, 8-byte aligned
    G r;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        u2 v = NT ? __builtin_nontemporal_load((const u2 *)w + i) : ((const u2 *)w)[i];
        r.l[2 * i] = v.x; r.l[2 * i + 1] = v.y;
    }
    return r;
}
template <bool NT> __device__ __forceinline__ Niels28<P> load_signed(const uint32_t *rec, bool neg) {
    Niels28<P> r;
    r.ymx = load_coord<NT>(rec + (neg ? 14 : 0)); r.ypx = load_coord<NT>(rec + (neg ? 0 : 14)); r.td = load_coord<NT>(rec + 28);
    return r;
}
// te28.cuh te_madd_hot with the record load as a template parameter (same arithmetic, same fences)
template <bool NT> __device__ __forceinline__ void madd_hot(AccTE<P> &a, Niels28<P> &n, bool neg, const uint32_t *next, bool next_neg, uint64_t bias) {
    G A = G::mul_biased(a.y.template sub_lazy<3>(a.x), n.ymx, bias);
    G B = G::mul_biased(a.y.add_lazy(a.x), n.ypx, bias);
    G C = G::mul_biased(a.t, n.td, bias);
    asm volatile("" : "+v"(A.l[G::N - 1]), "+v"(B.l[G::N - 1]), "+v"(C.l[G::N - 1]) : : "memory");
    n = load_signed<NT>(next, next_neg);
    asm volatile("" : "+v"(A.l[0]), "+v"(C.l[0]) : : "memory");
    G E = B.template sub_lazy<2>(A), H = B.add_lazy(A);
    G D = a.z.dbl_lazy();
    G U = D.template sub_lazy<2>(C), V = D.add_lazy(C), F, Gg;
#pragma unroll
    for (int i = 0; i < G::N; i++) { F.l[i] = neg ? V.l[i] : U.l[i]; Gg.l[i] = neg ? U.l[i] : V.l[i]; }
    a.x = G::mul_biased(E, F, bias); a.y = G::mul_biased(Gg, H, bias); a.t = G::mul_biased(E, H, bias); a.z = G::mul_biased(F, Gg, bias);
}
template <int STRIDE_WORDS, bool NT>
__global__ void __launch_bounds__(64, 2) k_acc(Stamp *st, AccTE<P> *sink, const uint32_t *__restrict__ tab, uint32_t mask, int iters) {
    const uint64_t bias = G::hot_loop_bias();
    uint32_t t = blockIdx.x * 64 + threadIdx.x, x = t * 2654435761u + 12345u;
    AccTE<P> acc = te_identity<P>();
    Niels28<P> pt = load_signed<NT>(tab + (size_t)(x & mask) * STRIDE_WORDS, false);
    uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        const uint32_t cur = x;
        x = x * 1664525u + 1013904223u;
        madd_hot<NT>(acc, pt, cur >> 31, tab + (size_t)((x >> 8) & mask) * STRIDE_WORDS, x >> 31, bias);
    }
    uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
    sink[t] = acc;
}
__global__ void k_fill(uint32_t *tab, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) { uint32_t x = (uint32_t)i * 2654435761u + 99u; x ^= x >> 15; x *= 2246822519u; tab[i] = (x >> 4) & ((i % 14 == 13) ? 0xffffu : 0x0fffffffu); }
}

template <class F> static void run(const char *name, F launch, Stamp *d_st, int grid, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h(grid);
    CK(hipMemcpy(h.data(), d_st, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> c(grid), m(grid);
    for (int i = 0; i < grid; i++) { c[i] = (double)h[i].cyc / iters; m[i] = (double)h[i].cyc / ((double)h[i].rt / 1e8) / 1e6; }
    std::sort(c.begin(), c.end()); std::sort(m.begin(), m.end());
    printf("%-8s %8.3f ms per launch   %9.0f shader cycles per addition per wave (median; / 3 = %6.0f per SIMD)   shader clock %7.1f MHz (p10 %7.1f, p90 %7.1f)\n",
           name, ms / reps, c[grid / 2], c[grid / 2] / 3, m[grid / 2], m[grid / 10], m[grid * 9 / 10]);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int grid = 8192, iters = 83;
    const bool sweep = argc > 1;        // any argument: the production layout at 3 ... 208 additions per lane (how much of a SMALL MSM's accumulation is per-wave overhead?)
    const uint32_t big = 1u << 24;
    Stamp *st; AccTE<P> *sink; uint32_t *tab;
    CK(hipMalloc(&st, grid * sizeof(Stamp)));
    CK(hipMalloc(&sink, (size_t)grid * 64 * sizeof(AccTE<P>)));
    const size_t words = (size_t)big * 64;                      // 2^24 records on a 256-byte stride = 4.3 GB
    CK(hipMalloc(&tab, words * 4));
    k_fill<<<(unsigned)(words / 256), 256>>>(tab, words);
    CK(hipDeviceSynchronize());
    if (sweep) {
        for (int it : {3, 6, 13, 26, 52, 83, 208}) {
            char name[32]; snprintf(name, sizeof name, "s192x%d", it);
            run(name, [&] { k_acc<48, false><<<grid, 64>>>(st, sink, tab, big - 1, it); }, st, grid, it);
        }
        printf("# additions per second = 8192 x 64 x iterations / time: compare the rows\n");
        return 0;
    }
    for (int round = 0; round < 2; round++) {
        run("l2", [&] { k_acc<48, false><<<grid, 64>>>(st, sink, tab, 4095, iters); }, st, grid, iters);
        run("s192", [&] { k_acc<48, false><<<grid, 64>>>(st, sink, tab, big - 1, iters); }, st, grid, iters);
        run("s256", [&] { k_acc<64, false><<<grid, 64>>>(st, sink, tab, big - 1, iters); }, st, grid, iters);
        run("s192nt", [&] { k_acc<48, true><<<grid, 64>>>(st, sink, tab, big - 1, iters); }, st, grid, iters);
        run("s256nt", [&] { k_acc<64, true><<<grid, 64>>>(st, sink, tab, big - 1, iters); }, st, grid, iters);
    }
    return 0;
}
