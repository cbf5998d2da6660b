// common.h -- shared host-side helpers for libst3d (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/st3d.h"

namespace st3d {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(st3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// roctx range for rocprofv3 --marker-trace (st3d_trace_push / st3d_trace_pop, comm.hip): no-ops unless ST3D_ROCTX=1
struct TraceRange {
    explicit TraceRange(const char *name) { st3d_trace_push(name); }
    ~TraceRange() { st3d_trace_pop(); }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

#define ST3D_CHECK_ARG(cond)                                                             \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            st3d::set_error("%s: invalid argument: %s", __func__, #cond);                \
            return ST3D_E_INVALID;                                                       \
        }                                                                                \
    } while (0)

#define ST3D_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            st3d::set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_)); \
            return ST3D_E_HIP;                                                           \
        }                                                                                \
    } while (0)

#define ST3D_LAUNCH_CHECK() ST3D_HIP(hipGetLastError())

#define ST3D_TRY(call)                                                                   \
    do {                                                                                 \
        int r_ = (call);                                                                 \
        if (r_ != ST3D_OK) return r_;                                                    \
    } while (0)

constexpr int kWave = 64;
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace st3d
