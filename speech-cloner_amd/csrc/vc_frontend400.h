// Internal interface between vc_frontend.hip (plan, validation, dispatch) and vc_frontend400.hip (the two-launch
// front-end of the shipped configuration).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct Fe400Args {
    const float* wav;
    const int32_t* lens;
    int32_t max_samples, wav_stride, max_frames;
    int32_t out_rows;             // rows per utterance of mfcc / mel_db / pow_db (<= max_frames): later frames only feed the statistics
    const float* win_tw;          // window[400] | W400^(n2 k1) cos [13][16] | sin [13][16]   (16-byte aligned)
    const float* mel_w;           // sparse Slaney rows, <= 14 weights each
    const int32_t* mel_start;     // [80]
    const int32_t* mel_off;       // [81]
    const float* dct_half;        // [40][40]: librosa.filters.dct(40, 80)[i][j], j < 40   (16-byte aligned)
    float pre_emph, amp_norm, mfcc_norm, m_norm, p_norm;
    int32_t first_mfcc, deriv, clip;
    float* stats;                 // [B][nt1][8]: max P, min P, max mel, min mel, sum|x| per 16-frame tile (linear)
    float* mel0;                  // [B][80]: frame 0's mel power
    int32_t nt1;                  // 16-frame tiles per utterance (over max_frames)
    float* mfcc;
    float* mel_db;
    float* pow_db;
};

// stage_mask: 2 = statistics pass, 4 = feature pass
int vc_fe400_launch(const Fe400Args& a, int batch, int stage_mask, hipStream_t st);
