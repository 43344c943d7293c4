// Internal interface between vc_gemm.hip (vc_conv_gemm dispatch) and vc_bank256.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct Bank256Pair {
    const void* Bt0;      // narrower filter: [128][taps0 * Cin] bf16, K contiguous
    const void* Bt1;      // wider filter:    [128][(taps0 + 1) * Cin]
    int32_t taps0, pad_l, c_off0, c_off1;
    int32_t extra;        // taps of the second filter - taps0: 1 for a bank pair, 0 for the two halves of ONE 256-channel filter
};

struct Bank256Args {
    const void* X;
    int32_t M, T, Cin, ldx;
    const float* epi_scale;
    const float* epi_shift;
    int32_t act;
    void* C;
    int32_t ldc, n_pairs;
    int32_t xcd_tiles;    // > 0: 1-D grid with the XCD-aware block -> (pair, row tile) mapping (set by the launcher)
    // per XCD (workgroup id & 7) up to 4 segments of work, walked in order: row tiles [first, first + count) of a pair
    int16_t seg_pair[8][4], seg_first[8][4], seg_count[8][4];
    int32_t pool;         // store max(y[t], y[t+1]) per window: row tiles advance by 255 frames
    // Split K (a single pair only: the long-K projection, launch_proj256): ksplit > 1 workgroups share a row tile, each
    // running a contiguous range of the channel slabs; every one but the LAST to finish writes its float32 accumulators
    // to a slab of `ws`, the last one adds them to its own and runs the epilogue (vc_bank256.hip, "split K").
    int32_t ksplit;       // 1 = off
    float* ws;            // [row tiles][ksplit - 1][256 * 256] float32 partial accumulators (register order)
    unsigned* tick;       // [row tiles][2] {arrival ticket, slabs published}; zeroed by the launcher before every launch
    int32_t dbg;          // -DVC_ABLATE builds only (option ablate_bank256): 1 = skip the K loop, 2 = skip the stores, 4 = no loads inside the K loop, 8 = no barrier (timing only, wrong results); ignored by the shipped build
    Bank256Pair p[16];
};

int vc_launch_bank256(const Bank256Args& a, hipStream_t st);
// Split-K form of a single-pair launch: how many ways K should be split for M rows (1 = not worth it) and the bytes of
// workspace ([tickets | slabs]) that takes.
int vc_bank256_ksplit(int M, int nslab);
size_t vc_bank256_ws_bytes(int M, int ksplit);
