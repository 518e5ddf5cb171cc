// nd_amd/csrc/common.hpp -- shared host-side plumbing of libnd_amd.so
// (error strings, HIP call checking, per-kernel event timing).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nd_amd.h"

namespace nd_amd {

void set_error(const char *fmt, ...);

#define ND_HIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            ::nd_amd::set_error("%s failed: %s (%s:%d)", #expr,                   \
                                hipGetErrorString(_e), __FILE__, __LINE__);       \
            return ND_AMD_EHIP;                                                   \
        }                                                                         \
    } while (0)

// Records a start/stop event pair around one kernel launch when timing is on.
struct KernelTimer {
    KernelTimer(int kernel_id, hipStream_t stream);
    ~KernelTimer();
    int slot;
    hipStream_t stream;
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace nd_amd
