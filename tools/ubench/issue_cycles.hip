// tools/ubench/issue_cycles.hip -- what ONE SIMD of gfx950 spends, in shader-clock cycles, on each instruction class of k_accumulate<EdwardsLaw>, where the dispatcher really
// puts the waves, and what clock the chip runs at meanwhile.  Round 6, VERDICT r05 item 1: "calibrate per box ... or a counter-backed kill" of the four-waves-per-SIMD variant.
//
// Every wave brackets its loop with s_memtime (shader-clock counter) and s_memrealtime (constant 100 MHz) and records HW_REG_HW_ID + HW_REG_XCC_ID, so the host knows which
// (XCD, SE, CU, SIMD) it ran on and which other waves of the launch shared that SIMD with it in time.  Reported per kernel and grid size:
//   placement      histogram of "peak co-resident waves per SIMD" over the SIMDs the launch touched
//   cyc/instr/wave shader cycles per instruction as ONE wave sees it (median over waves), grouped by how many waves shared its SIMD
//   cyc/instr/SIMD = the above / co-resident waves: what the SIMD's pipeline sustains
//   clock          s_memtime ticks per second of s_memrealtime (if s_memtime runs at the shader clock the column moves with DVFS; a constant says it does not)
// Instruction streams are single asm blocks (separate asm statements get an s_nop each from the hazard recogniser, which doubles every figure):
//   mad8   eight independent v_mad_u64_u32 chains          mad1   one dependent chain (latency)
//   add64  v_lshl_add_u64 (the hot loop's 64-bit add)       shr64  v_lshrrev_b64          logic  v_and_b32         bitop  v_bitop3_b32
//   fqmul  chained ff28 mul_biased products (378 multiply-adds + 81 others each)
//   temadd te_madd_hot over an L2-resident table of 4096 records: the hot loop itself, gather included, memory latency taken out
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aes_zero_knowledge_proof_circuit_amd/csrc tools/ubench/issue_cycles.hip -o tools/ubench/issue_cycles.bin
//   tools/ubench/issue_cycles.bin [max waves per SIMD = 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#include "te28.cuh"
using namespace zk;
using P = Fq377P;
using G = FpMsm<P>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stamp { uint64_t cyc, rt0, rt1; uint32_t hw_id, xcc_id; };
struct Probe { uint64_t c, r; };
__device__ __forceinline__ Probe stamp_begin() { Probe p; p.r = __builtin_amdgcn_s_memrealtime(); p.c = __builtin_readcyclecounter(); return p; }
__device__ __forceinline__ void stamp_end(Stamp *out, Probe p) {
    uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        Stamp s; s.cyc = c1 - p.c; s.rt0 = p.r; s.rt1 = r1;
        s.hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));          // HW_REG_HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
        s.xcc_id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));         // HW_REG_XCC_ID[3:0]
        out[blockIdx.x] = s;
    }
}
#define R8(x) x x x x x x x x
#define MAD(d) "v_mad_u64_u32 %" #d ", vcc, %8, %9, %" #d "\n\t"

__global__ void __launch_bounds__(64) k_mad8(Stamp *st, uint64_t *sink, uint32_t a, uint32_t b, int iters) {
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t x = a + threadIdx.x, y = b;
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++)
        asm volatile(R8(MAD(0) MAD(1) MAD(2) MAD(3) MAD(4) MAD(5) MAD(6) MAD(7))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
    stamp_end(st, p);
    sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(64) k_mad1(Stamp *st, uint64_t *sink, uint32_t a, uint32_t b, int iters) {
    uint64_t acc = threadIdx.x;
    uint32_t x = a + threadIdx.x, y = b;
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++)
        asm volatile(R8(R8("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t")) : "+v"(acc) : "v"(x), "v"(y) : "vcc");
    stamp_end(st, p);
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}
#define OP2(op, d) op " %" #d ", %" #d ", 0, %8\n\t"
__global__ void __launch_bounds__(64) k_add64(Stamp *st, uint64_t *sink, uint64_t a, int iters) {
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, x = a + threadIdx.x;
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++)
        asm volatile(R8(OP2("v_lshl_add_u64", 0) OP2("v_lshl_add_u64", 1) OP2("v_lshl_add_u64", 2) OP2("v_lshl_add_u64", 3) OP2("v_lshl_add_u64", 4) OP2("v_lshl_add_u64", 5)
                        OP2("v_lshl_add_u64", 6) OP2("v_lshl_add_u64", 7))
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    stamp_end(st, p);
    sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
#define SHR(d) "v_lshrrev_b64 %" #d ", 1, %" #d "\n\t"
__global__ void __launch_bounds__(64) k_shr64(Stamp *st, uint64_t *sink, uint64_t a, int iters) {
    uint64_t a0 = ~(a + threadIdx.x), a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++)
        asm volatile(R8(SHR(0) SHR(1) SHR(2) SHR(3) SHR(4) SHR(5) SHR(6) SHR(7)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    stamp_end(st, p);
    sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
#define AND(d) "v_and_b32 %" #d ", %" #d ", %8\n\t"
__global__ void __launch_bounds__(64) k_logic(Stamp *st, uint64_t *sink, uint32_t a, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++)
        asm volatile(R8(AND(0) AND(1) AND(2) AND(3) AND(4) AND(5) AND(6) AND(7)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a));
    stamp_end(st, p);
    sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
#define BOP(d) "v_bitop3_b32 %" #d ", %" #d ", %8, %" #d " bitop3:0xc\n\t"
__global__ void __launch_bounds__(64) k_bitop(Stamp *st, uint64_t *sink, uint32_t a, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++)
        asm volatile(R8(BOP(0) BOP(1) BOP(2) BOP(3) BOP(4) BOP(5) BOP(6) BOP(7)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(a));
    stamp_end(st, p);
    sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(64) k_fqmul(Stamp *st, G *sink, G a, int iters) {
    const uint64_t bias = G::hot_loop_bias();
    G x = a, y = a;
    x.l[0] += threadIdx.x & 0xff;
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++) { x = G::mul_biased(x, y, bias); y = G::mul_biased(y, x, bias); }
    stamp_end(st, p);
    sink[blockIdx.x * 64 + threadIdx.x] = x + y;
}
// the hot loop itself (kernels_msm.hip k_accumulate<EdwardsLaw>), gathering from a table small enough to live in L2
__global__ void __launch_bounds__(64, 2) k_temadd(Stamp *st, AccTE<P> *sink, const Niels28<P> *__restrict__ tab, uint32_t mask, int iters) {
    const uint64_t bias = G::hot_loop_bias();
    uint32_t t = blockIdx.x * 64 + threadIdx.x, x = t * 2654435761u + 12345u;
    AccTE<P> acc = te_identity<P>();
    Niels28<P> pt = niels_load_signed<P>(tab + (x & mask), false);
    Probe p = stamp_begin();
    for (int i = 0; i < iters; i++) {
        const uint32_t cur = x;
        x = x * 1664525u + 1013904223u;
        const Niels28<P> *next = tab + ((x >> 8) & mask);
        te_madd_hot<P>(acc, pt, cur >> 31, next, x >> 31, bias);
    }
    stamp_end(st, p);
    sink[t] = acc;
}

__global__ void k_fill(Niels28<P> *tab, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = i * 2654435761u + 99u;
    Niels28<P> r;
    for (int k = 0; k < 14; k++) { x = x * 1664525u + 1013904223u; r.ymx.l[k] = x >> 4; x = x * 1664525u + 1013904223u; r.ypx.l[k] = x >> 4; x = x * 1664525u + 1013904223u; r.td.l[k] = x >> 4; }
    r.ymx.l[13] &= 0xffff; r.ypx.l[13] &= 0xffff; r.td.l[13] &= 0xffff;
    for (int k = 0; k < 6; k++) r.pad[k] = 0;
    tab[i] = r;
}

// ---- host: placement + per-SIMD concurrency from the stamps
struct Row { int co; double cyc_per_instr_wave, mhz; };
static void analyse(const char *name, int k, double instr_per_wave, const std::vector<Stamp> &h, float ms, double rt_hz) {
    std::map<uint32_t, std::vector<int>> by_simd;               // (xcc, se, sh, cu, simd) -> waves
    for (int i = 0; i < (int)h.size(); i++) {
        uint32_t id = h[i].hw_id, key = ((h[i].xcc_id & 0xf) << 16) | (((id >> 13) & 7) << 12) | (((id >> 12) & 1) << 11) | (((id >> 8) & 0xf) << 4) | ((id >> 4) & 3);
        by_simd[key].push_back(i);
    }
    // co-residency of a wave = the largest number of waves of its SIMD (itself included) alive at its MIDPOINT (waves of one launch start together or in rounds)
    std::map<int, std::vector<double>> cyc_by_co, mhz_by_co;
    std::map<int, int> simd_peak_hist;
    for (auto &kv : by_simd) {
        int peak = 0;
        for (int i : kv.second) {
            uint64_t mid = (h[i].rt0 + h[i].rt1) / 2;
            int co = 0;
            for (int j : kv.second) if (h[j].rt0 <= mid && mid <= h[j].rt1) co++;
            peak = std::max(peak, co);
            cyc_by_co[co].push_back((double)h[i].cyc / instr_per_wave);
            mhz_by_co[co].push_back((double)h[i].cyc / ((double)(h[i].rt1 - h[i].rt0) / rt_hz) / 1e6);
        }
        simd_peak_hist[peak]++;
    }
    printf("%-7s grid %4d x64 (%d per SIMD if spread evenly)  %8.3f ms  SIMDs touched %4zu, peak co-resident waves per SIMD:", name, (int)h.size(), k, ms, by_simd.size());
    for (auto &kv : simd_peak_hist) printf(" %dx:%d", kv.first, kv.second);
    printf("\n");
    for (auto &kv : cyc_by_co) {
        auto c = kv.second, m = mhz_by_co[kv.first];
        std::sort(c.begin(), c.end()); std::sort(m.begin(), m.end());
        double cm = c[c.size() / 2], mm = m[m.size() / 2];
        printf("        waves sharing a SIMD %d: %6zu waves   cycles/instr/wave %10.3f   cycles/instr/SIMD %10.3f   s_memtime clock %7.1f MHz\n", kv.first, c.size(), cm, cm / kv.first, mm);
    }
    fflush(stdout);
}
template <class F> static void run(const char *name, int k, double instr_per_wave, F launch, Stamp *d_st, int grid, double rt_hz) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();                                            // warm-up (clocks, caches)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h(grid);
    CK(hipMemcpy(h.data(), d_st, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    analyse(name, k, instr_per_wave, h, ms, rt_hz);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    const int simds = 1024;
    int maxk = argc > 1 ? atoi(argv[1]) : 4;
    Stamp *st; uint64_t *sink; G *gsink; AccTE<P> *asink; Niels28<P> *tab;
    const int gmax = simds * 8;
    CK(hipMalloc(&st, gmax * sizeof(Stamp)));
    CK(hipMalloc(&sink, gmax * 64 * 8));
    CK(hipMalloc(&gsink, gmax * 64 * sizeof(G)));
    CK(hipMalloc(&asink, gmax * 64 * sizeof(AccTE<P>)));
    const uint32_t nrec = 4096;
    std::vector<Niels28<P>> h(nrec);
    uint32_t x = 2463534242u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    // arbitrary limbs < 2^28 stand in for curve points: the arithmetic is data-independent
    for (auto &n : h) { for (int k = 0; k < 14; k++) { n.ymx.l[k] = rnd() & 0x0fffffffu; n.ypx.l[k] = rnd() & 0x0fffffffu; n.td.l[k] = rnd() & 0x0fffffffu; } n.ymx.l[13] &= 0xffff; n.ypx.l[13] &= 0xffff; n.td.l[13] &= 0xffff; }
    CK(hipMalloc(&tab, nrec * sizeof(Niels28<P>)));
    CK(hipMemcpy(tab, h.data(), nrec * sizeof(Niels28<P>), hipMemcpyHostToDevice));
    G ga; for (int k = 0; k < 14; k++) ga.l[k] = 0x0123457u + 977u * k;
    int clk_khz = 0, wall_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    const double rt_hz = wall_khz > 0 ? wall_khz * 1e3 : 1e8;
    printf("# device clock rate attribute %d kHz, wall clock rate %d kHz\n", clk_khz, wall_khz);
    for (int k = 1; k <= maxk; k++) {
        const int grid = simds * k;
        const int it = 4000;
        run("mad8", k, 64.0 * it, [&] { k_mad8<<<grid, 64>>>(st, sink, 0x0f123457u, 0x0abcdef1u, it); }, st, grid, rt_hz);
        run("mad1", k, 64.0 * it, [&] { k_mad1<<<grid, 64>>>(st, sink, 0x0f123457u, 0x0abcdef1u, it); }, st, grid, rt_hz);
        run("add64", k, 64.0 * it, [&] { k_add64<<<grid, 64>>>(st, sink, 0x123456789abcull, it); }, st, grid, rt_hz);
        run("shr64", k, 64.0 * it, [&] { k_shr64<<<grid, 64>>>(st, sink, 0x123456789abcdefull, it); }, st, grid, rt_hz);
        run("logic", k, 64.0 * it, [&] { k_logic<<<grid, 64>>>(st, sink, 0xfffffff7u, it); }, st, grid, rt_hz);
        run("bitop", k, 64.0 * it, [&] { k_bitop<<<grid, 64>>>(st, sink, 0x0fffffffu, it); }, st, grid, rt_hz);
        const int itf = 600;
        run("fqmul", k, 2.0 * itf, [&] { k_fqmul<<<grid, 64>>>(st, gsink, ga, itf); }, st, grid, rt_hz);          // "instruction" = one Fq product
        if (k <= 3) {
            const int ita = 300;
            run("temadd", k, (double)ita, [&] { k_temadd<<<grid, 64>>>(st, asink, tab, nrec - 1, ita); }, st, grid, rt_hz);   // "instruction" = one bucket addition
        }
    }
    // the production launch shape: 2^19 buckets = 8192 single-wave workgroups, ~83 additions per bucket, gathers from a table far larger than L2 + MALL
    {
        const uint32_t big = 1u << 24;                                            // 2^24 records x 192 B = 3.2 GB
        Niels28<P> *btab;
        CK(hipMalloc(&btab, (size_t)big * sizeof(Niels28<P>)));
        k_fill<<<big / 256, 256>>>(btab, big);
        CK(hipDeviceSynchronize());
        const int grid = 8192, ita = 83;
        run("acc8192", 3, (double)ita, [&] { k_temadd<<<grid, 64>>>(st, asink, btab, big - 1, ita); }, st, grid, rt_hz);
        run("acc8192-L2", 3, (double)ita, [&] { k_temadd<<<grid, 64>>>(st, asink, tab, nrec - 1, ita); }, st, grid, rt_hz);
        CK(hipFree(btab));
    }
    printf("# fqmul / temadd / acc8192: the 'instruction' is one Fq product / one bucket addition: cycles per product / per addition.\n");
    return 0;
}
