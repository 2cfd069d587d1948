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
    return 0;
}
