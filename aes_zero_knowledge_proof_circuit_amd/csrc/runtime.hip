// csrc/runtime.hip -- thin HIP runtime plumbing (allocation, copies, streams, events)
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <time.h>
#include "hip_util.hpp"

namespace zk {
namespace gpu {

// ROCclr multiplexes all HIP streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4): with 10-16 prover contexts several
// streams share a queue and a 10 ms k_accumulate of one context blocks the sort / reduction / NTT kernels of the contexts behind it.
// One hardware queue per prover context keeps them independent (measured +1.5 % without window tables, +4.7 % with: profiles/r02_msm_tables.md).
// The variable is read when the HIP runtime initialises, i.e. at the first HIP call of the process: libzkaes sets it (without overriding a value
// the caller exported) when the library is loaded; a process that initialised HIP earlier must export it itself.  An embedder whose other threads may read the
// environment while the library loads opts out with ZKAES_KEEP_ENV=1 (and exports GPU_MAX_HW_QUEUES itself if it wants the 16 queues).
namespace {
struct HwQueueDefault {
    HwQueueDefault() { if (!getenv("ZKAES_KEEP_ENV")) setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} g_hw_queue_default;
}  // namespace

#ifdef ZKAES_MEASURE
// measurement builds only (ZK_EXTRA_DEFINES=-DZKAES_MEASURE python -m ...build --force): ZKAES_KNOCKIN is a bit mask of pipeline parts to run TWICE (all idempotent,
// so proofs stay valid) -- 1 MSM sort, 2 bucket reductions, 8 accumulate, 32 every NTT.  The drop in blocks/s of a saturated bench run is that part's real cost beside
// the other contexts' kernels, which a one-context profile cannot show.  Release builds carry none of it.
int knockin() { static const int v = [] { const char *e = getenv("ZKAES_KNOCKIN"); return e ? atoi(e) : 0; }(); return v; }
#endif

namespace {
std::atomic<OpRecord *> g_oplog{nullptr};
std::mutex g_oplog_mu;
}  // namespace
void oplog_begin() { std::lock_guard<std::mutex> g(g_oplog_mu); delete g_oplog.exchange(new OpRecord()); }
OpRecord oplog_end() {
    std::lock_guard<std::mutex> g(g_oplog_mu);
    OpRecord *r = g_oplog.exchange(nullptr);
    OpRecord out;
    if (r) { out = std::move(*r); delete r; }
    return out;
}
void oplog_ntt(uint64_t n, int count) {
    if (!g_oplog.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> g(g_oplog_mu);
    if (OpRecord *r = g_oplog.load()) r->ntt.emplace_back(n, count);
}
void oplog_msm(uint64_t points, int kind) {
    if (!g_oplog.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> g(g_oplog_mu);
    if (OpRecord *r = g_oplog.load()) r->msm.emplace_back(points, kind);
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
void require_device() {
    if (device_count() <= 0) throw GpuError("no HIP device available: libzkaes proves on an AMD GPU (gfx950) and has no CPU fallback");
}
int current_device() { int d = 0; HIP_CHECK(hipGetDevice(&d)); return d; }
// ---- how a host thread waits for its stream.  hipStreamSynchronize spins (ROCclr's default "active wait"): right for a lone encrypt() call, whose waits are
// short and on the critical path, but a throughput call keeps one host thread per prover context waiting most of the time -- 16 spinning threads per GPU
// were ~13 busy cores per rank in round 2, 104 of a node's 128 cores at 8 ranks.  (hipDeviceScheduleBlockingSync hangs the bench on ROCm 7.2,
// profiles/r02_msm_tables.md section 6, so the interrupt path is not used.)  While a ThroughputWaits scope is alive, sync() polls hipStreamQuery with
// nanosleep back-off instead: 50 -> 200 us quanta, i.e. at most a few hundred microseconds of added latency per wait, hidden behind the other contexts' kernels.
// ZKAES_WAIT=spin / sleep forces one policy.
namespace {
std::atomic<int> g_throughput_waits{0};
int wait_override() { static const int v = [] { const char *e = getenv("ZKAES_WAIT"); return !e ? 0 : !strcmp(e, "spin") ? 1 : !strcmp(e, "sleep") ? 2 : 0; }(); return v; }
}  // namespace
ThroughputWaits::ThroughputWaits(bool on) : on_(on) { if (on_) g_throughput_waits.fetch_add(1); }
ThroughputWaits::~ThroughputWaits() { if (on_) g_throughput_waits.fetch_sub(1); }
bool throughput_mode() { return g_throughput_waits.load(std::memory_order_relaxed) > 0; }
void sync(stream_t s_) {
    hipStream_t s = (hipStream_t)s_;
    const int ov = wait_override();
    if (ov == 1 || (ov == 0 && g_throughput_waits.load(std::memory_order_relaxed) == 0)) { HIP_CHECK(hipStreamSynchronize(s)); return; }
    long ns = 50000;
    for (int spins = 0;; spins++) {
        hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) HIP_CHECK(e);
        if (spins < 4) continue;                       // a few immediate re-queries catch the short waits
        struct timespec ts = {0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 200000) ns += 50000;
    }
}
void set_device(int ordinal) { HIP_CHECK(hipSetDevice(ordinal)); }
void *dmalloc(size_t bytes) { void *p = nullptr; HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16)); return p; }
size_t mem_free_bytes() { size_t f = 0, t = 0; HIP_CHECK(hipMemGetInfo(&f, &t)); return f; }
void dfree(void *p) { if (p) (void)hipFree(p); }
void h2d(void *dst, const void *src, size_t bytes, stream_t s) { if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s)); }
// a device-to-host copy into pageable memory waits INSIDE hipMemcpyAsync (spinning) for everything queued before it: drain the stream with sync() first,
// which sleeps in throughput mode, so that only the copy itself (microseconds) is waited for actively
void d2h(void *dst, const void *src, size_t bytes, stream_t s) { if (bytes) { sync(s); HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s)); HIP_CHECK(hipStreamSynchronize((hipStream_t)s)); } }
void d2d(void *dst, const void *src, size_t bytes, stream_t s) { if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s)); }
void dzero(void *dst, size_t bytes, stream_t s) { if (bytes) HIP_CHECK(hipMemsetAsync(dst, 0, bytes, (hipStream_t)s)); }
// (stream priorities -- prover streams high, k_accumulate on a low-priority side stream -- measured neutral-to-worse on MI355X in round 2,
// profiles/r02_bench_stream_priority.md, and were removed in round 4)
stream_t stream_create() {
    hipStream_t s;
    HIP_CHECK(hipStreamCreate(&s));
    return (stream_t)s;
}
// a stream whose kernels yield to every other stream's when workgroup slots free up (the lowest priority the device offers): background work of a lone proof --
// the mask polynomial's commitment runs under the witness generation and must not starve it
stream_t stream_create_background() {
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s;
    HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamDefault, least));
    return (stream_t)s;
}
void stream_destroy(stream_t s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
void *event_create() { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return (void *)e; }
void event_record(void *ev, stream_t s) { HIP_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); }
void stream_wait_event(stream_t s, void *ev) { HIP_CHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0)); }
float event_elapsed_ms(void *a, void *b) { float ms = 0; HIP_CHECK(hipEventSynchronize((hipEvent_t)b)); HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b)); return ms; }
void event_destroy(void *ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }

}  // namespace gpu
}  // namespace zk
