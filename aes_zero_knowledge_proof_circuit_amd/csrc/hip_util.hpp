// csrc/hip_util.hpp -- HIP error checking shared by the .hip translation units
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "gpu.hpp"

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw zk::gpu::GpuError(std::string("HIP error ") + hipGetErrorString(_e) + " at " __FILE__ ":" + std::to_string(__LINE__) + " (" #expr ")"); \
    } while (0)
#define HIP_LAUNCH_CHECK() HIP_CHECK(hipGetLastError())
