// Internal interface between vc_gemm.hip (vc_conv_gemm dispatch) and vc_conv256.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct Conv256Args {
    const void* X;
    int32_t M, T, Cin, ldx, N;
    const void* Bt;       // [N][taps * Cin] bf16, K contiguous
    int32_t K, taps, pad_l, c_off;
    int32_t pool;         // operand = max(frame, next frame) of a non-negative tensor
    const float* epi_scale;
    const float* epi_shift;
    int32_t act;
    const void* R;        // residual [M, ldr] bf16 or NULL
    int32_t ldr;
    void* C;
    int32_t ldc;
};

int vc_launch_conv256(const Conv256Args& a, hipStream_t st);
