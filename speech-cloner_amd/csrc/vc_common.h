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

// Kernel-selection options (vc_set_option / vc_get_option, include/vc_hip.h).  -1 = the library's own choice.
// They pick between HIP kernels that compute the SAME function (A/B measurements, regression tests); nothing in
// the library reads the process environment.  The result-corrupting ablation switches exist only in builds made
// with -DVC_ABLATE (tools/build_ablate.sh), never in the shipped libvc_hip.so.
enum Option {
    OPT_BANK256 = 0,      // 0: filter banks on conv_kernel instead of bank256_kernel
    OPT_BANK256_XCD,      // 0: plain 2-D grid, 1: whole pairs per XCD, 2: pairs split over two XCDs (default)
    OPT_CONV256,          // 0: long-K single filters on conv_kernel / gemm_kernel instead of conv256_kernel
    OPT_CONV256_MINK,     // shortest K that takes conv256_kernel (default 384)
    OPT_CONV256_WM,       // 2: keep 128-row blocks for 128-column launches (default: 256 rows from 2,048 rows up)
    OPT_PROJ256,          // 0: 256-channel long-K projection on conv256_kernel instead of the bank tiles
    OPT_PROJ256_SPLIT,    // 0: never split that projection's K over two workgroups per row tile
    OPT_WGRAD_XCD,        // 0: weight-gradient tiles dealt round-robin instead of group-per-XCD
    OPT_GRU_MFMA,         // 0: VALU recurrence always, 1: MFMA recurrence always (default: from 32 sequences up)
    OPT_FE_FUSED,         // 0: the shipped front-end configuration as two launches (statistics pass + feature pass) instead of one
    OPT_FE_FUSED_SPIN,    // polls a block of the one-launch front-end waits for its utterance's tiles (default 4000 ~ 4 ms); 0: never wait
    OPT_GRU_SMALL_MFMA,   // 1: the encoder's H = 40 bf16 recurrence on the 16-sequence MFMA wave (measured slower; default: gru_wave_kernel)
    OPT_GRU_MFMA4,        // 1: the four-wave, all-weights-in-registers MFMA recurrence (measured slower; default: eight waves)
    OPT_PRENET_LDS,       // 0: every wave of prenet_chain streams the weights itself (default: shared through LDS)
    OPT_GRU_TRAIN_RESIDENT,   // 0: the float32 training recurrences stream all their weights from L2 every step
    OPT_CBHG_FRONT_MI,    // 4: 128-row blocks in cbhg_small_kernel (default 2)
    OPT_GRU_F32_WIDE,     // 0: float32 inference recurrences of more than 128 units stay on gru_generic_kernel (default: the training forward kernel)
    OPT_F32_F16X3,        // 0: float32 inference convolutions stay on the f32-input MFMA kernels (default: vc_gemm16)
    OPT_GEMM16_SPLIT,     // vc_gemm16 single-pair launches: K split ways (1..8) + 16 * block map (0: a row tile's splits on one XCD, 1: a K range per XCD); default: automatic
    OPT_ABLATE_BANK256,   // -DVC_ABLATE only: bit mask, see vc_bank256.h
    OPT_ABLATE_BANK256_ONLY,   // -DVC_ABLATE only: launch one pair alone
    OPT_ABLATE_CBHG_FRONT,     // -DVC_ABLATE only: bit mask, see vc_cbhg_small.hip
    OPT_COUNT
};
int opt(Option o);        // current value, -1 if unset

#if defined(__HIPCC__)
// Wave-wide reductions on DPP (data-parallel primitives: the cross-lane operand of a VALU instruction), result in every
// lane.  __shfl_xor lowers to ds_bpermute_b32 on gfx950 -- an LDS-path round trip of ~100 cycles per step, 6 dependent
// steps per reduction; the DPP form is 6 plain vector instructions (quad swaps, half-row / row mirrors, then the
// row_bcast:15 / row_bcast:31 carries of the GFX9 family) and one v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
#define VC_WAVE_REDUCE(OP, ID)                                                                  \
    v = OP(v, dpp_move<0xB1, 0xF>(ID, v));     /* quad_perm [1,0,3,2] */                        \
    v = OP(v, dpp_move<0x4E, 0xF>(ID, v));     /* quad_perm [2,3,0,1] */                        \
    v = OP(v, dpp_move<0x141, 0xF>(ID, v));    /* row_half_mirror */                            \
    v = OP(v, dpp_move<0x140, 0xF>(ID, v));    /* row_mirror: every lane of a row of 16 */      \
    v = OP(v, dpp_move<0x142, 0xA>(ID, v));    /* row_bcast:15 into rows 1 and 3 */             \
    v = OP(v, dpp_move<0x143, 0xC>(ID, v));    /* row_bcast:31 into rows 2 and 3 */             \
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
__device__ __forceinline__ float vc_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) { VC_WAVE_REDUCE(vc_addf, 0.0f) }
__device__ __forceinline__ float wave_max(float v) { VC_WAVE_REDUCE(fmaxf, -3.402823466e38f) }
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
__device__ __forceinline__ float wave_min(float v) { VC_WAVE_REDUCE(fminf, 3.402823466e38f) }
#undef VC_WAVE_REDUCE
#endif

}  // namespace vc
