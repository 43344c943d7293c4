// Shared host/device helpers for libvc_hip.so (error reporting, wave/block reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "vc_hip.h"

namespace vc {

char* last_error_buf();
int set_error(int code, const char* fmt, ...);

#define VC_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return vc::set_error(VC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                 \
    } while (0)

#define VC_REQUIRE(cond, ...)                                                                \
    do {                                                                                     \
        if (!(cond)) return vc::set_error(VC_ERR_INVALID, __VA_ARGS__);                      \
    } while (0)

constexpr int WAVE = 64;

#if defined(__HIPCC__)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif

}  // namespace vc
