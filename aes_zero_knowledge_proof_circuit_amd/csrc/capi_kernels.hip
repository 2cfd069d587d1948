// csrc/capi_kernels.hip -- kernel-level C entry points (parity tests + roofline measurement) and device selection
#include "../../include/zkaes.h"
#include <cstring>
#include <string>
#include <vector>
#include "hip_util.hpp"
#include "marlin.hpp"

struct zkaes_pk { std::unique_ptr<zk::ProvingKey> pk; };
extern "C" const char *zkaes_last_error(void);
namespace zk { void capi_set_error(const std::string &); }

namespace {
template <class Fn> int guardk(Fn &&fn) {
    try { zk::capi_set_error(""); fn(); return 0; }
    catch (const std::exception &e) { zk::capi_set_error(e.what()); return 1; }
    catch (...) { zk::capi_set_error("unknown error"); return 1; }
}
template <class Fr> void run_ntt(uint8_t *data, size_t n, int inverse) {
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    if (((size_t)1 << lg) != n) throw std::invalid_argument("zkaes_ntt: n must be a power of two");
    zk::gpu::require_device();
    zk::gpu::stream_t s = zk::gpu::stream_create();
    Fr *a = (Fr *)zk::gpu::dmalloc(n * sizeof(Fr)), *b = (Fr *)zk::gpu::dmalloc(n * sizeof(Fr));
    zk::gpu::h2d(a, data, n * sizeof(Fr), s);
    zk::gpu::ntt<Fr>(b, a, n, lg, inverse != 0, s);
    zk::gpu::d2h(data, b, n * sizeof(Fr), s);
    zk::gpu::dfree(a); zk::gpu::dfree(b); zk::gpu::stream_destroy(s);
}
template <class Curve> void run_msm(const uint8_t *bases, const uint8_t *scalars, size_t n, uint8_t *out_xy, int *out_inf, int reps, double *ms_total, double *ms_acc) {
    using Fq = typename Curve::Fq; using Fr = typename Curve::Fr;
    zk::gpu::require_device();
    zk::gpu::stream_t s = zk::gpu::stream_create();
    zk::Affine<Fq> *db = (zk::Affine<Fq> *)zk::gpu::dmalloc(n * 96);
    Fr *ds = (Fr *)zk::gpu::dmalloc(n * 32);
    zk::gpu::h2d(db, bases, n * 96, s); zk::gpu::h2d(ds, scalars, n * 32, s);
    zk::gpu::sync(s);
    zk::gpu::MsmWorkspace *ws = zk::gpu::msm_workspace_create();
    zk::XYZZ<Fq> r = zk::gpu::msm<Curve>(ws, db, ds, n, s);   // warm-up / result
    if (reps > 0) {
        zk::gpu::MsmStats before = zk::gpu::msm_stats(false);
        void *e0 = zk::gpu::event_create(), *e1 = zk::gpu::event_create();
        zk::gpu::event_record(e0, s);
        for (int i = 0; i < reps; i++) r = zk::gpu::msm<Curve>(ws, db, ds, n, s);
        zk::gpu::event_record(e1, s);
        float ms = zk::gpu::event_elapsed_ms(e0, e1);
        zk::gpu::MsmStats after = zk::gpu::msm_stats(false);
        if (ms_total) *ms_total = ms / reps;
        if (ms_acc) *ms_acc = (after.accumulate_ms - before.accumulate_ms) / reps;
        zk::gpu::event_destroy(e0); zk::gpu::event_destroy(e1);
    }
    zk::Affine<Fq> a = r.to_affine();
    if (out_inf) *out_inf = a.is_inf() ? 1 : 0;
    if (out_xy) { memcpy(out_xy, a.x.l, 48); memcpy(out_xy + 48, a.y.l, 48); }
    zk::gpu::msm_workspace_destroy(ws);
    zk::gpu::dfree(db); zk::gpu::dfree(ds); zk::gpu::stream_destroy(s);
}
}  // namespace

extern "C" {
int zkaes_set_device(int ordinal) { return guardk([&] { HIP_CHECK(hipSetDevice(ordinal)); }); }
int zkaes_ntt(int field_id, uint8_t *data, size_t n, int inverse) {
    return guardk([&] { if (field_id == 381) run_ntt<zk::Fr381>(data, n, inverse); else if (field_id == 377) run_ntt<zk::Fr377>(data, n, inverse); else throw std::invalid_argument("field_id must be 377 or 381"); });
}
int zkaes_msm(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, uint8_t *out_xy, int *out_inf) {
    return guardk([&] { if (curve_id == 381) run_msm<zk::Bls381>(bases, scalars, n, out_xy, out_inf, 0, nullptr, nullptr); else if (curve_id == 377) run_msm<zk::Bls377>(bases, scalars, n, out_xy, out_inf, 0, nullptr, nullptr); else throw std::invalid_argument("curve_id must be 377 or 381"); });
}
int zkaes_msm_bench(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, int reps, double *ms_total, double *ms_accumulate) {
    return guardk([&] { if (curve_id == 381) run_msm<zk::Bls381>(bases, scalars, n, nullptr, nullptr, reps, ms_total, ms_accumulate); else run_msm<zk::Bls377>(bases, scalars, n, nullptr, nullptr, reps, ms_total, ms_accumulate); });
}
}
