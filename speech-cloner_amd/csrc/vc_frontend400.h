// Internal interface between vc_frontend.hip (plan, validation, dispatch) and vc_frontend400.hip (the two-launch
// front-end of the shipped configuration).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int FE400_FUSED_MAX_BATCH = 1024;   // utterances per launch the plan's counter sets are sized for

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
    // one-launch form (vc_frontend400.hip, fe400_fused_kernel): 14-frame tiles
    float* fstats;                // [B][fstride] floats: tile records of 8 floats, each utterance's array padded to whole 128-B lines
    int32_t fstride;              // floats per utterance in fstats (a multiple of 32)
    unsigned* fcount;             // [FE400_FUSED_MAX_BATCH] (one per 256 bytes): tiles of the utterance that have published (zero at launch)
    unsigned* fcount_other;       // the set the previous launch used: zeroed by this launch for the next one
    int32_t spin_limit;           // polls before a waiting block computes the utterance's records itself (set by the launcher)
    float* mfcc;
    float* mel_db;
    float* pow_db;
};

// stage_mask: 2 = statistics pass, 4 = feature pass; with `fused` != 0 (both stages asked for, vc_fe400_fused_ok) they run
// as ONE launch (fe400_fused_kernel)
int vc_fe400_launch(const Fe400Args& a, int batch, int stage_mask, int fused, hipStream_t st);
// floats of fstats per utterance for max_frames frames, and whether the one-launch form takes such utterances
int vc_fe400_fused_stride(int max_frames);
int vc_fe400_fused_count_bytes(int batch);
bool vc_fe400_fused_ok(int max_frames);
