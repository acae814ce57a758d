// Shared host-side helpers of libtgtc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/tgtc_hip.h"

namespace tgtc {

// thread-local error text returned by tgtc_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define TGTC_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return ::tgtc::fail(TGTC_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                __FILE__, __LINE__);                                          \
    } while (0)

#define TGTC_LAUNCH_CHECK()                                                                        \
    do {                                                                                           \
        hipError_t e__ = hipGetLastError();                                                        \
        if (e__ != hipSuccess)                                                                     \
            return ::tgtc::fail(TGTC_ERR_HIP, "kernel launch: %s (%s:%d)", hipGetErrorString(e__), \
                                __FILE__, __LINE__);                                               \
    } while (0)

#define TGTC_REQUIRE(cond, ...)                                 \
    do {                                                        \
        if (!(cond)) return ::tgtc::fail(TGTC_ERR_ARG, __VA_ARGS__); \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

}  // namespace tgtc
