// tools/ubench/lds_buckets.hip -- north_star's kernel sketch, timed: "LDS-staged Pippenger bucket windows".
//
// The product keeps the buckets of a window in HBM behind a radix sort and gives every bucket ONE lane whose accumulator never leaves its registers
// (csrc/kernels_msm.hip k_accumulate).  The alternative keeps a window's buckets in LDS: a workgroup owns 2^9 extended-Edwards buckets (224 B each = 112 KB of the
// 160 KB), streams (bucket, point) pairs in arrival order -- no sort -- and adds every point into its LDS bucket.  DESIGN.md section 3 rejects it on paper: only
// c = 10 fits, so a 253-bit scalar needs 26 windows instead of 13, i.e. twice the bucket additions of a kernel that is bound by exactly those.  This probe measures the
// other half of the argument, the RATE of LDS-resident bucket additions, with the same 7-product addition (csrc/te28.cuh te_madd) on the same 192-byte records:
//   A. register accumulators: every lane adds its own stream of gathered points into a register-resident bucket (the product's inner loop without the bucket bookkeeping);
//   B. LDS buckets: 256 lanes, 512 buckets per workgroup, uniformly random bucket per pair; a lane claims its bucket with an LDS compare-and-swap, loads the 56 limbs,
//      adds, stores, releases (lanes of a wave that hit the same bucket take turns).
// Arithmetic is data-independent (no branches on values), so arbitrary limbs < 2^28 stand in for curve points.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aes_zero_knowledge_proof_circuit_amd/csrc tools/ubench/lds_buckets.hip -o /tmp/lds_buckets && /tmp/lds_buckets
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "te28.cuh"
using namespace zk;
using P = Fq377P;
constexpr int NB = 512, WG = 256, LIMBS = 56;          // buckets per workgroup, lanes, 32-bit words per bucket (4 coordinates x 14 limbs)

__device__ __forceinline__ uint32_t rnd(uint32_t &x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

__global__ void __launch_bounds__(64, 2) k_reg_buckets(const Niels28<P> *__restrict__ pts, uint32_t npts, int per_lane, AccTE<P> *__restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, x = t * 2654435761u + 1;
    AccTE<P> acc = te_identity<P>();
    Niels28<P> nxt = pts[rnd(x) % npts];
    for (int i = 0; i < per_lane; i++) {
        Niels28<P> p = nxt;
        nxt = pts[rnd(x) % npts];                       // prefetch the next gather under this addition
        te_madd<P>(acc, p);
    }
    out[t] = acc;
}

__global__ void __launch_bounds__(WG) k_lds_buckets(const Niels28<P> *__restrict__ pts, uint32_t npts, int per_lane, AccTE<P> *__restrict__ out) {
    __shared__ uint32_t acc[LIMBS][NB];                  // limb-major: lanes on different buckets hit different banks
    __shared__ uint32_t own[NB];
    for (int i = threadIdx.x; i < NB; i += WG) {
        own[i] = 0;
        AccTE<P> id = te_identity<P>();
        const uint32_t *w = (const uint32_t *)&id;
        for (int k = 0; k < LIMBS; k++) acc[k][i] = w[k];
    }
    __syncthreads();
    uint32_t x = (blockIdx.x * WG + threadIdx.x) * 2654435761u + 1;
    Niels28<P> nxt = pts[rnd(x) % npts];
    for (int i = 0; i < per_lane; i++) {
        Niels28<P> p = nxt;
        nxt = pts[rnd(x) % npts];
        const uint32_t b = rnd(x) % NB;
        bool done = false;
        // Lanes that drew the same bucket (in this wave or another of the workgroup) take turns.  The loop condition is wave-uniform (a ballot) on purpose: with a per-lane
        // `while (!done)` the compiler may sink the critical section below the loop, where the winner waits at the reconvergence point for lanes spinning on its lock.
        while (__ballot(!done)) {
            if (!done && atomicCAS(&own[b], 0u, 1u) == 0u) {
                AccTE<P> a;
                uint32_t *w = (uint32_t *)&a;
#pragma unroll
                for (int k = 0; k < LIMBS; k++) w[k] = acc[k][b];
                te_madd<P>(a, p);
#pragma unroll
                for (int k = 0; k < LIMBS; k++) acc[k][b] = w[k];
                __threadfence_block();
                atomicExch(&own[b], 0u);
                done = true;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += WG) {
        AccTE<P> a;
        uint32_t *w = (uint32_t *)&a;
        for (int k = 0; k < LIMBS; k++) w[k] = acc[k][i];
        out[(size_t)blockIdx.x * NB + i] = a;
    }
}

int main() {
    const uint32_t npts = 1u << 22;                      // 805 MB of records: gathers miss the caches as in the product
    std::vector<Niels28<P>> h(npts);
    uint32_t x = 12345;
    for (auto &r : h) { uint32_t *w = (uint32_t *)&r; for (int k = 0; k < 42; k++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w[k] = x & 0x0fffffff; } }
    Niels28<P> *d; AccTE<P> *o;
    if (hipMalloc(&d, (size_t)npts * sizeof(Niels28<P>)) != hipSuccess || hipMalloc(&o, (size_t)1 << 28) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemcpy(d, h.data(), (size_t)npts * sizeof(Niels28<P>), hipMemcpyHostToDevice);
    printf("setup done\n"); fflush(stdout);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    {   // A: 2^19 lanes (the product's bucket count at c = 20), 64 additions each
        const int lanes = 1 << 19, per = 64;
        hipLaunchKernelGGL(k_reg_buckets, dim3(lanes / 64), dim3(64), 0, 0, d, npts, 4, o); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_reg_buckets, dim3(lanes / 64), dim3(64), 0, 0, d, npts, per, o); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("A register buckets : %.2f G bucket additions/s (%.3f ms for %.1f M)\n", (double)lanes * per / ms / 1e6, ms, (double)lanes * per / 1e6); fflush(stdout);
    }
    for (int wgs : {256, 512, 1024}) {   // B: one / two / four workgroups per CU worth of work (112 KB of LDS: one resident per CU)
        const int per = 128;
        hipLaunchKernelGGL(k_lds_buckets, dim3(wgs), dim3(WG), 0, 0, d, npts, 4, o); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_lds_buckets, dim3(wgs), dim3(WG), 0, 0, d, npts, per, o); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("B LDS buckets, %4d workgroups x 256 lanes x %d pairs, 512 buckets each: %.2f G bucket additions/s (%.3f ms)\n", wgs, per, (double)wgs * WG * per / ms / 1e6, ms); fflush(stdout);
    }
    printf("(a 253-bit scalar needs 13 windows of 20 bits with HBM buckets, 26 of 10 bits with 512 LDS buckets per workgroup)\n");
    return 0;
}
