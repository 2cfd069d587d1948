// csrc/capi_common.hpp -- what the two halves of the extern "C" boundary share: the handle structs, the thread-local error slot, the exception guard.
// capi_host.cpp holds the host-only entry points (verifier, (de)serialisation, circuit queries: no device, also built under sanitizers / libFuzzer);
// capi.cpp the ones that need a GPU; capi_kernels.hip the kernel-level ones.
#pragma once
#include "../../include/zkaes.h"
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "marlin.hpp"

struct zkaes_pk { std::unique_ptr<zk::ProvingKey> pk; };
struct zkaes_vk { zk::VerifyingKey vk; };

namespace zk {
void capi_set_error(const std::string &m);       // the calling thread's zkaes_last_error() message (defined in capi_host.cpp)
namespace capi {
template <class Fn> int guard(Fn &&fn) {
    try { capi_set_error(""); fn(); return 0; }
    catch (const std::exception &e) { capi_set_error(e.what()); return 1; }
    catch (...) { capi_set_error("unknown error"); return 1; }
}
inline uint8_t *give(const std::vector<uint8_t> &v) {
    uint8_t *p = (uint8_t *)malloc(v.size() ? v.size() : 1);
    if (!p) throw std::bad_alloc();
    if (!v.empty()) memcpy(p, v.data(), v.size());
    return p;
}
inline void fill_info(const Circuit &c, uint64_t out[12]) {
    out[0] = c.raw_constraints; out[1] = c.raw_instance; out[2] = c.raw_witness;
    out[3] = c.A.nnz(); out[4] = c.B.nnz(); out[5] = c.C.nnz();
    out[6] = c.num_constraints; out[7] = c.num_instance; out[8] = c.num_witness; out[9] = 0; out[10] = 0; out[11] = 0;
}
inline size_t next_pow2(size_t n) { if (n > ((size_t)1 << 62)) throw std::length_error("size out of range"); size_t p = 1; while (p < n) p <<= 1; return p; }
}  // namespace capi
}  // namespace zk
