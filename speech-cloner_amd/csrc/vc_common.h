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
// highwaynet gate (modules.py:315-319): relu(h) * t + x * (1 - t), t = sigmoid(tpre), written as
// x + t * (relu(h) - x) with v_exp_f32 / v_rcp_f32 (1 ulp each) -- the gate arithmetic, not the
// matrix work, bounds the fused highway chain, and an IEEE division costs ~10 instructions.
// gemm_kernel's highway epilogue and highway_chain_kernel share it (bit-identical paths).
__device__ __forceinline__ float highway_gate(float hpre, float tpre, float x) {
    const float hv = fmaxf(hpre, 0.0f);
    const float e = __builtin_amdgcn_exp2f(tpre * -1.4426950408889634f);
    const float tv = __builtin_amdgcn_rcpf(1.0f + e);
    return fmaf(tv, hv - x, x);
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif

}  // namespace vc
