// tools/ubench/gather_probe.hip -- calibration kernel for rocprofv3's FETCH_SIZE on k_accumulate's access pattern (tools/pmc_accumulate.py):
// every lane reads ONE 192-byte, 64-byte aligned record (a Niels28 SRS point, csrc/te28.cuh) at a pseudo-random index of a 6 GB array
// (far beyond the 256 MB Infinity Cache, every record read once: no reuse), with the same 16-byte loads the compiler emits for the bucket kernel.
// The host prints how many bytes that touches at 64-byte and at 128-byte granularity; FETCH_SIZE of this kernel divided by those numbers tells how
// the counter tallies such gathers on gfx950 (MI355X_MICROARCH.md: wide coalesced streams report exactly half; "other access widths: calibrate").
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct alignas(64) Rec { uint32_t w[48]; };       // 192 B, 64-byte aligned: the Niels28 record of csrc/te28.cuh (168 B of coordinates + padding)
__global__ void __launch_bounds__(64) k_gather_probe(const Rec *__restrict__ recs, uint64_t nrec, uint32_t n, uint32_t *__restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint64_t idx = ((uint64_t)t * 0x9E3779B97F4A7C15ull >> 17) % nrec;        // distinct for t < n << nrec? not necessarily -- the host counts exactly
    Rec r = recs[idx];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 48; i++) s ^= r.w[i];
    out[t] = s;
}
int main() {
    const uint64_t nrec = 32000000ull;            // 6.1 GB
    const uint32_t n = 1u << 24;
    Rec *d; uint32_t *o;
    if (hipMalloc(&d, nrec * sizeof(Rec)) != hipSuccess || hipMalloc(&o, (size_t)n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, nrec * sizeof(Rec));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_gather_probe, dim3(n / 64), dim3(64), 0, 0, d, nrec, n, o);
    hipDeviceSynchronize();
    // exact footprint of one launch: distinct 64-byte and 128-byte blocks touched (bitmaps), and the requested bytes
    const uint64_t bytes = nrec * sizeof(Rec);
    uint8_t *m64 = (uint8_t *)calloc(bytes / 64 / 8 + 2, 1), *m128 = (uint8_t *)calloc(bytes / 128 / 8 + 2, 1);
    uint64_t c64 = 0, c128 = 0;
    for (uint32_t t = 0; t < n; t++) {
        uint64_t idx = ((uint64_t)t * 0x9E3779B97F4A7C15ull >> 17) % nrec, lo = idx * sizeof(Rec), hi = lo + sizeof(Rec) - 1;
        for (uint64_t b = lo / 64; b <= hi / 64; b++) if (!(m64[b >> 3] & (1 << (b & 7)))) { m64[b >> 3] |= 1 << (b & 7); c64++; }
        for (uint64_t b = lo / 128; b <= hi / 128; b++) if (!(m128[b >> 3] & (1 << (b & 7)))) { m128[b >> 3] |= 1 << (b & 7); c128++; }
    }
    printf("gather_probe records %u requested_bytes %llu touched_64B_bytes %llu touched_128B_bytes %llu\n", n, (unsigned long long)n * sizeof(Rec), (unsigned long long)c64 * 64, (unsigned long long)c128 * 128);
    return 0;
}
