// tools/ubench/rates.hip -- instruction-rate probes on gfx950 that the MSM design decisions rest on:
// v_mad_u64_u32 vs v_lshl_add_u64 vs v_add_u32 throughput per CU, and the current Fq / Fr Montgomery product rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../aes_zero_knowledge_proof_circuit_amd/csrc/ff28.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mad(uint64_t *out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
    uint32_t x = a + threadIdx.x, y = b;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (uint64_t)x * y + acc[k];
        x += 3;
    }
    uint64_t s = 0; for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add64(uint64_t *out, uint64_t a, int iters) {
    uint64_t acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
    uint64_t x = a + threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = acc[k] + x + (acc[(k + 1) & 7] >> 63);
    }
    uint64_t s = 0; for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(uint32_t *out, uint32_t a, int iters) {
    uint32_t acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (acc[k] ^ a) + acc[(k + 3) & 7];
    }
    uint32_t s = 0; for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(uint32_t *out, uint32_t a, int iters) {
    uint32_t acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k + 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = acc[k] * a;
    }
    uint32_t s = 0; for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
__global__ void k_fmul(F *out, F a, int iters) {
    F x = a, y = a;
    x.l[0] += threadIdx.x;
    for (int i = 0; i < iters; i++) { x = x * y; y = y * x; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
template <class G>
__global__ void k_fmul28(G *out, G a, int iters) {
    G x = a, y = a;
    x.l[0] += threadIdx.x & 0xff;
    for (int i = 0; i < iters; i++) { x = x * y; y = y * x; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
template <class G>
__global__ void k_fsqr28(G *out, G a, int iters) {
    G x = a, y = a;
    x.l[0] += threadIdx.x & 0xff;
    for (int i = 0; i < iters; i++) { x = x.sqr() + y; y = y.sqr() + x; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
// ---- FP64-FMA probes ("measure an FP64-FMA field product", kill criterion < 1.25 x the ff28 product rate).
// RATE probes: they issue the instruction mix and dependency structure of the two known FP64 big-integer schemes on 377-bit operands, in the default
// rounding mode (the exact schemes need round-toward-zero, which changes no rate) -- they are not validated field implementations.
__global__ void k_fma64(double *out, double a, int iters) {
    double acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
    double x = a + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = fma(acc[k], x, 0.5);
    }
    double s = 0; for (int k = 0; k < 8; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// (a) 16 x 24-bit limbs held as doubles: a limb product is < 2^48, a double column absorbs the 32 products of a Montgomery product exactly (< 2^53), so
//     v_fma_f64 plays the role v_mad_u64_u32 plays in ff28.cuh.  Per row: 16 product FMAs, the row's Montgomery factor m = -t_i mod 2^24 (p = 1 mod 2^24:
//     floor via a magic-constant add, one FMA for the remainder), 15 reduction FMAs, one carry add.  Final: 16-limb carry normalisation.
struct Fq24d { double l[16]; };
__device__ __forceinline__ Fq24d mul24d(const Fq24d &a, const Fq24d &b, const double *pm) {
    const double M52 = 6755399441055744.0, I24 = 1.0 / 16777216.0, T24 = 16777216.0;      // 1.5 * 2^52 (round-to-integer magic), 2^-24, 2^24
    double t[32];
#pragma unroll
    for (int i = 0; i < 32; i++) t[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) t[i + j] = fma(a.l[j], b.l[i], t[i + j]);
        double q = (t[i] * I24 + M52) - M52;              // t_i / 2^24 rounded
        double lo = fma(-q, T24, t[i]);                   // remainder
        double m = T24 - lo;                              // -t_i mod 2^24 (up to the boundary case the exact scheme patches)
#pragma unroll
        for (int j = 1; j < 16; j++) t[i + j] = fma(m, pm[j], t[i + j]);
        t[i + 1] += q + 1.0;
    }
    Fq24d r;
    double c = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        double v = t[16 + i] + c;
        double q = (v * I24 + M52) - M52;
        r.l[i] = fma(-q, T24, v);
        c = q;
    }
    return r;
}
__global__ void k_fmul24d(double *out, double seed, int iters) {
    double pm[16];
    Fq24d x, y;
    for (int i = 0; i < 16; i++) { pm[i] = 1234567.0 + 4099.0 * i; x.l[i] = seed + i + (threadIdx.x & 0xff); y.l[i] = seed * 3 + i; }
    for (int i = 0; i < iters; i++) { x = mul24d(x, y, pm); y = mul24d(y, x, pm); }
    double s = 0; for (int i = 0; i < 16; i++) s += x.l[i] + y.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// (b) Emmart-Zheng-Weems: 8 x 52-bit limbs, every 104-bit limb product split into halves with two FMAs against 2^104-scaled constants, halves accumulated
//     as 64-bit INTEGERS: per limb product 2 x v_fma_f64 + 1 x v_add_f64 + 2 x 64-bit integer add; 64 products + 64 for the reduction, 8 row factors.
struct Fq52d { double l[8]; };
__device__ __forceinline__ Fq52d mul52d(const Fq52d &a, const Fq52d &b, const double *pm) {
    const double C1 = 20282409603651670423947251286016.0, C2 = 20282409603651674927546878656512.0;   // 2^104, 2^104 + 2^52
    long long lo[17], hi[17];
#pragma unroll
    for (int i = 0; i < 17; i++) { lo[i] = 0; hi[i] = 0; }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            double ph = fma(a.l[j], b.l[i], C1);
            double pl = fma(a.l[j], b.l[i], C2 - ph);
            hi[i + j + 1] += __double_as_longlong(ph);
            lo[i + j] += __double_as_longlong(pl);
        }
        double m = (double)((lo[i] + hi[i]) & 0xfffffffffffffll);         // row factor (times -p^-1 mod 2^52 = -1 for this prime)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            double ph = fma(m, pm[j], C1);
            double pl = fma(m, pm[j], C2 - ph);
            hi[i + j + 1] += __double_as_longlong(ph);
            lo[i + j] += __double_as_longlong(pl);
        }
    }
    Fq52d r;
    long long c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { long long v = lo[8 + i] + hi[8 + i] + c; r.l[i] = (double)(v & 0xfffffffffffffll); c = v >> 52; }
    return r;
}
__global__ void k_fmul52d(double *out, double seed, int iters) {
    double pm[8];
    Fq52d x, y;
    for (int i = 0; i < 8; i++) { pm[i] = 1234567890123.0 + 4099.0 * i; x.l[i] = seed + i + (threadIdx.x & 0xff); y.l[i] = seed * 3 + i; }
    for (int i = 0; i < iters; i++) { x = mul52d(x, y, pm); y = mul52d(y, x, pm); }
    double s = 0; for (int i = 0; i < 8; i++) s += x.l[i] + y.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// ---- MFMA probe: the idle matrix pipe for the CONSTANT half of a Montgomery product.  k_accumulate sits at the floor of its VALU formulation; the
// reduction's m x p is a constant-matrix product (Toeplitz(p) x [digits x 64 lanes]) -- an int8 MFMA shape that would run beside the VALU.  This is an OPTIMISTIC RATE
// PROBE, not a field implementation: per product it issues the VALU half (a x b: 196 v_mad_u64_u32 into 28 columns), packs the 14 low limbs into 49 byte digits,
// moves them through LDS into MFMA operand layout (element e's digits are needed by lanes e % 16 + 16 q), runs the 28 v_mfma_i32_16x16x64_i8 of a [64 lanes x 64 digits]
// x [64 x 112 columns] product (7 column tiles x 4 element groups), moves the 98 i32 columns per element back through LDS, recombines them into the 64-bit columns and
// normalises.  Left OUT, all in the probe's favour: computing m = -T_lo p^-1 mod 2^392 at all (a second, dependent 49 x 49 constant product with its own two transposes
// and a serial 49-digit carry chain between the two), signed-digit recoding for the signed i8 operands, any validation.  Kill criterion: < 1.25 x the ff28 product rate.
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int MF_COLS = 116;          // 112 columns padded to a 16-byte multiple that spreads the lanes' rows over the LDS banks
__global__ void __launch_bounds__(64) k_fmul_mfma_probe(uint32_t *out, uint32_t seed, int iters) {
    __shared__ uint32_t lds_in[64 * 16];                 // 64 bytes of digits per element
    __shared__ uint32_t lds_out[64 * MF_COLS];
    const uint32_t lane = threadIdx.x, MASK = (1u << 28) - 1;
    uint32_t a[14], b[14];
    for (int i = 0; i < 14; i++) { a[i] = (seed * (i + 3) + lane * 2654435761u) & MASK; b[i] = (seed * (i + 7) + 12345u) & MASK; }
    v4i toep[7];                                         // this lane's fragment of the constant Toeplitz matrix, one per column tile (7 x 4 VGPRs)
    for (int t = 0; t < 7; t++) for (int j = 0; j < 4; j++) toep[t][j] = (int)((seed + 0x01010101u * (t * 4 + j + lane)) & 0x7f7f7f7fu);
    for (int it = 0; it < iters; it++) {
        uint64_t t[28];
#pragma unroll
        for (int i = 0; i < 28; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 14; i++)
#pragma unroll
            for (int j = 0; j < 14; j++) t[i + j] += (uint64_t)a[j] * b[i];
        // low half: carry-normalise the 14 columns (a real reduction needs the limbs of T mod R), then 13 dwords of byte digits (28-bit limbs packed contiguously, as
        // Fp28::pack does), pretending they were m
        uint32_t lo[16];
        {
            uint64_t cl = 0;
#pragma unroll
            for (int i = 0; i < 14; i++) { uint64_t v = t[i] + cl; lo[i] = (uint32_t)v & MASK; cl = v >> 28; }
            t[14] += cl;
            lo[14] = lo[15] = 0;
        }
        uint32_t w[16];
#pragma unroll
        for (int k = 0; k < 13; k++) {
            const int bit = 32 * k, i = bit / 28, sh = bit % 28;
            uint64_t v = lo[i] >> sh;
            v |= (uint64_t)lo[i + 1] << (28 - sh);
            if (28 - sh + 28 < 32) v |= (uint64_t)lo[i + 2] << (56 - sh);
            w[k] = (uint32_t)v & 0x7f7f7f7fu;
        }
        w[13] = w[14] = w[15] = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) *reinterpret_cast<uint4 *>(&lds_in[lane * 16 + k]) = make_uint4(w[k], w[k + 1], w[k + 2], w[k + 3]);
        __syncthreads();
        v4i acc[4][7];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 d = *reinterpret_cast<const uint4 *>(&lds_in[(q * 16 + (lane & 15)) * 16 + (lane >> 4) * 4]);
            v4i bf; bf[0] = (int)d.x; bf[1] = (int)d.y; bf[2] = (int)d.z; bf[3] = (int)d.w;
#pragma unroll
            for (int tl = 0; tl < 7; tl++) {
                v4i z = {0, 0, 0, 0};
                acc[q][tl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(toep[tl], bf, z, 0, 0, 0);
            }
        }
        // D[row = column within the tile][col = element within the group]: lane holds rows 4 (lane / 16) .. + 3 of column lane % 16
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int tl = 0; tl < 7; tl++)
                *reinterpret_cast<uint4 *>(&lds_out[(q * 16 + (lane & 15)) * MF_COLS + tl * 16 + (lane >> 4) * 4]) =
                    make_uint4((uint32_t)acc[q][tl][0], (uint32_t)acc[q][tl][1], (uint32_t)acc[q][tl][2], (uint32_t)acc[q][tl][3]);
        __syncthreads();
        uint32_t c[100];
#pragma unroll
        for (int k = 0; k < 100; k += 4) { const uint4 v = *reinterpret_cast<const uint4 *>(&lds_out[lane * MF_COLS + k]); c[k] = v.x; c[k + 1] = v.y; c[k + 2] = v.z; c[k + 3] = v.w; }
        // byte-weighted columns back into the 28-bit-spaced 64-bit columns, then the carry normalisation of the high half
#pragma unroll
        for (int j = 0; j < 98; j++) if ((8 * j) / 28 >= 14) t[(8 * j) / 28] += (uint64_t)c[j] << ((8 * j) % 28);      // (the columns below R are not needed: see cy)
        uint64_t cy = lo[13] != 0;             // the low half of T + m p is zero mod R: only its carry reaches the result
#pragma unroll
        for (int i = 0; i < 14; i++) { uint64_t v = t[14 + i] + cy; a[i] = (uint32_t)v & MASK; cy = v >> 28; }
#pragma unroll
        for (int i = 0; i < 14; i++) b[i] = (b[i] + a[(i + 1) % 14]) & MASK;
        __syncthreads();
    }
    uint32_t s = 0; for (int i = 0; i < 14; i++) s ^= a[i] ^ b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// ---- global-atomic rate on 2^19 counters with uniformly random keys (would a counting scatter beat the radix sort of the MSM's (bucket, point) pairs?)
__global__ void k_atomic_slots(uint32_t *cnt, uint32_t *slots, uint32_t nkeys_per_thread, uint32_t mask, uint32_t stride) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t i = 0; i < nkeys_per_thread; i++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        uint32_t k = x & mask;
        uint32_t slot = atomicAdd(&cnt[k], 1u);
        if (slots && slot < stride) slots[(size_t)k * stride + slot] = x;          // the scattered 4-byte write of a counting scatter
    }
}
template <class Fn> float timeit(Fn fn) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    fn(); hipDeviceSynchronize();
    hipEventRecord(a); fn(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    void *buf; CK(hipMalloc(&buf, (size_t)blocks * threads * 64));
    double lanes = (double)blocks * threads;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)buf, 12345u, 67891u, iters); });
    printf("v_mad_u64_u32 : %.2f Tops/s (%.3f ms)\n", lanes * iters * 8 / ms / 1e9, ms);
    // is the multiplier's speed operand-dependent?  (the limbs of ff28 are 28 bits wide, the probe above multiplies 14- and 17-bit numbers)
    for (uint32_t mag : {0x00003039u, 0x00f12345u, 0x0f123457u, 0xf1234567u}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)buf, mag, mag ^ 0x5a5a5u, iters); });
        printf("v_mad_u64_u32 with operands ~ 0x%08x: %.2f Tops/s (%.3f ms)\n", mag, lanes * iters * 8 / ms / 1e9, ms);
    }
    ms = timeit([&] { hipLaunchKernelGGL(k_add64, dim3(blocks), dim3(threads), 0, 0, (uint64_t *)buf, 12345ull, iters); });
    printf("add64 (x2 per step + shift): %.2f Tsteps/s (%.3f ms)\n", lanes * iters * 8 / ms / 1e9, ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_add32, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)buf, 12345u, iters); });
    printf("xor+add32     : %.2f Tsteps/s (%.3f ms)\n", lanes * iters * 8 / ms / 1e9, ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(threads), 0, 0, (uint32_t *)buf, 12345u, iters); });
    printf("v_mul_lo_u32  : %.2f Tops/s (%.3f ms)\n", lanes * iters * 8 / ms / 1e9, ms);
    Fq377 one = Fq377::one(); Fr377 oner = Fr377::one();
    ms = timeit([&] { hipLaunchKernelGGL((k_fmul<Fq377>), dim3(blocks), dim3(threads), 0, 0, (Fq377 *)buf, one, 200); });
    printf("Fq377 mul     : %.2f Gmul/s (%.3f ms)  => %.2f T limb-mads/s\n", lanes * 400 / ms / 1e6, ms, lanes * 400 * 288 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((k_fmul<Fr377>), dim3(blocks), dim3(threads), 0, 0, (Fr377 *)buf, oner, 200); });
    printf("Fr377 mul     : %.2f Gmul/s (%.3f ms)  => %.2f T limb-mads/s\n", lanes * 400 / ms / 1e6, ms, lanes * 400 * 128 / ms / 1e9);
    Fq377x28 g = Fq377x28::from_std(one);
    ms = timeit([&] { hipLaunchKernelGGL((k_fmul28<Fq377x28>), dim3(blocks), dim3(threads), 0, 0, (Fq377x28 *)buf, g, 200); });
    printf("Fq377x28 mul  : %.2f Gmul/s (%.3f ms)\n", lanes * 400 / ms / 1e6, ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_fsqr28<Fq377x28>), dim3(blocks), dim3(threads), 0, 0, (Fq377x28 *)buf, g, 200); });
    printf("Fq377x28 sqr  : %.2f Gsqr/s (%.3f ms)\n", lanes * 400 / ms / 1e6, ms);
    {
        const uint32_t nb = 1u << 19, stride = 256, per = 64;
        uint32_t *cnt, *slots; CK(hipMalloc(&cnt, nb * 4)); CK(hipMalloc(&slots, (size_t)nb * stride * 4));
        for (int with_write = 0; with_write < 2; with_write++) {
            CK(hipMemset(cnt, 0, nb * 4));
            hipLaunchKernelGGL(k_atomic_slots, dim3(blocks), dim3(threads), 0, 0, cnt, with_write ? slots : nullptr, 4u, nb - 1, stride); CK(hipDeviceSynchronize());
            CK(hipMemset(cnt, 0, nb * 4));
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_atomic_slots, dim3(blocks), dim3(threads), 0, 0, cnt, with_write ? slots : nullptr, per, nb - 1, stride);
            hipEventRecord(b); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b);
            printf("global atomicAdd (returning) on 2^19 random counters%s: %.2f G/s (%.3f ms for %.0f M)\n", with_write ? " + scattered 4-byte slot write" : "", lanes * per / t / 1e6, t, lanes * per / 1e6);
        }
        hipFree(cnt); hipFree(slots);
    }
    ms = timeit([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(threads), 0, 0, (double *)buf, 1.0000001, iters); });
    printf("v_fma_f64     : %.2f Tops/s (%.3f ms)\n", lanes * iters * 8 / ms / 1e9, ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_fmul24d, dim3(blocks), dim3(threads), 0, 0, (double *)buf, 1000.0, 200); });
    printf("Fq377 FP64 16x24-bit limbs (rate probe): %.2f Gmul/s (%.3f ms)\n", lanes * 400 / ms / 1e6, ms);
    ms = timeit([&] { hipLaunchKernelGGL(k_fmul52d, dim3(blocks), dim3(threads), 0, 0, (double *)buf, 1000.0, 200); });
    printf("Fq377 FP64 8x52-bit limbs, split FMAs + int64 sums (rate probe): %.2f Gmul/s (%.3f ms)\n", lanes * 400 / ms / 1e6, ms);
    {
        const int mblocks = 256 * 24, mit = 200;          // one wave per workgroup (30 KB of LDS each: 5 per CU)
        ms = timeit([&] { hipLaunchKernelGGL(k_fmul_mfma_probe, dim3(mblocks), dim3(64), 0, 0, (uint32_t *)buf, 0x9e3779b9u, mit); });
        printf("Fq377 28-bit limbs, a x b on the VALU + m x p on v_mfma_i32_16x16x64_i8 incl. digit packing and both LDS transposes, m itself assumed free "
               "(optimistic rate probe): %.2f Gmul/s (%.3f ms)\n", (double)mblocks * 64 * mit / ms / 1e6, ms);
    }
    return 0;
}
