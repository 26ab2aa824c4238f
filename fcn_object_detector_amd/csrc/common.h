// Internal helpers shared by the gfx950 kernels of libfcnhip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fcnhip.h"

namespace fcn {

// thread-local error text behind fcn_last_error_string()
char* err_buf();
int set_err(int code, const char* fmt, ...);

inline hipStream_t as_stream(fcn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define FCN_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return ::fcn::set_err(-(int)e__, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

#define FCN_LAUNCH_CHECK(name)                                                                \
    do {                                                                                      \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess)                                                                \
            return ::fcn::set_err(-(int)e__, "launch %s failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define FCN_REQUIRE(cond, code, ...)                         \
    do {                                                     \
        if (!(cond)) return ::fcn::set_err(code, __VA_ARGS__); \
    } while (0)

// conv_fwd.hip: the per-device 16-byte zero page that masked lanes load (allocated by fcn_init)
const float* zero_page_for_current_device(int* rc);

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// grid for HBM-bound grid-stride kernels: one work item per thread while that takes at most 64 Ki blocks (a grid only
// slightly larger than the chip's resident capacity would otherwise run 1 vs 2 loop iterations per thread: a 2x tail)
inline int stream_grid(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 65536) g = 8192;
    return (int)g;
}

}  // namespace fcn
