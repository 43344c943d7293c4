// Signal front-end on gfx950: batched calc_MFCC_input (/root/reference/audio_lib.py:89-244).
//
// Three launches per batch (all on the caller's stream, no host sync):
//   1. fe_abssum_kernel     per-utterance sum|x| partials (amplitude normalisation, :125-126)
//   2. fe_power_*_kernel    per tile of G frames: gather + reflect pad + scale + pre-emphasis
//                           into LDS, windowed 400-point real DFT (25x16 split, fe_dft400.h) or a
//                           direct DFT for other n_fft, |.|^2, 10log10 -> raw power dB (global),
//                           sparse Slaney mel from the LDS power tile, 20log10 -> raw mel dB
//                           (workspace), per-tile max/min partials
//   3. fe_finalize_kernel   per-utterance max/min from the partials, top_db clip, min shift,
//                           DCT-II (LDS-staged basis), first-coefficient subtraction, delta,
//                           scale, clip; zero-fills padding rows
// HBM bound: algorithmic traffic is 1,764 B/frame at the shipped config;
// this 2-pass form moves the raw dB tiles twice (second read mostly from Infinity Cache).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vc_common.h"
#include "fe_dft400.h"
#include "vc_frontend400.h"

namespace {

constexpr int FE_THREADS = 256;
constexpr int FE_NPART = 16;      // |x| partial sums per utterance
constexpr int FE_G400 = 16;       // frames per block, 400-point path
constexpr int FE_GGEN = 8;        // frames per block, generic path
constexpr int FE_G3 = 32;         // frames per block, finalize
constexpr int A_STRIDE = 17;      // LDS stride of one (g,k1) row of 16 complex values
constexpr float NEG_INF = -3.402823466e38f;
constexpr float POS_INF = 3.402823466e38f;
constexpr int FE_STAT = 8;        // floats per tile record

struct FeDev {
    const float* window;      // [n_fft]
    const float* tw400;       // [2][208]   W400^(n2 k1), k1-major
    const float* twg;         // [2][n_fft] cos, -sin of 2 pi m / n_fft
    const int32_t* mel_start; // [n_mels]
    const int32_t* mel_off;   // [n_mels + 1]
    const float* mel_w;       // [nnz]
    const float* dct;         // [n_mfcc][n_mels]
};

struct FeArgs {
    FeDev t;
    const float* wav;
    const int32_t* lens;
    int32_t max_samples, wav_stride, max_frames;
    int32_t hop, n_fft, n_bins, n_mels, n_mfcc, nnz;
    float pre_emph, amp_norm;
    float mfcc_norm, m_norm, p_norm;
    int32_t first_mfcc, deriv, clip;
    float* partial;           // [B][FE_NPART]
    float* stats;             // [B][ntiles][8]  pmax pmin mmax mmin sum|x| (of the tile's own samples) - - -
    int32_t lds_alias;        // 400-point kernel: power tile and mel tables reuse the (g,k1) row buffers
    int32_t fused_abs;        // 1: the STFT kernel accumulates sum|x| itself (hop <= n_fft/2), 0: fe_abssum_kernel ran
    float* mel_raw;           // [B][max_frames][n_mels]
    float* mfcc;              // outputs
    float* mel_db;
    float* pow_db;
    int32_t ntiles;           // tiles of kernel 2 per utterance (over max_frames)
    int32_t span;             // hop * (G - 1) + n_fft
};

// 10*log10(x) through v_log_f32 (1 ulp on log2): |error| < 3e-5 dB over the 100 dB range used here
__device__ __forceinline__ float db10(float x) { return 3.0102999566398120f * __log2f(x); }

__device__ __forceinline__ int utt_len(const FeArgs& a, int b) {
    int L = a.lens ? a.lens[b] : a.max_samples;
    return min(max(L, 1), a.max_samples);
}

// ------------------------------------------------------------------------------------------ 1
__global__ void __launch_bounds__(FE_THREADS)
fe_abssum_kernel(FeArgs a) {
    __shared__ float red[FE_THREADS / vc::WAVE];
    const int b = blockIdx.y, c = blockIdx.x;
    const int L = utt_len(a, b);
    const int chunk = (L + FE_NPART - 1) / FE_NPART;
    const int s = c * chunk, e = min(L, s + chunk);
    const float* x = a.wav + (size_t)b * a.wav_stride;
    float acc = 0.0f;
    for (int i = s + threadIdx.x; i < e; i += FE_THREADS) acc += fabsf(x[i]);
    acc = vc::wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < FE_THREADS / vc::WAVE; ++w) t += red[w];
        a.partial[b * FE_NPART + c] = t;
    }
}

// ------------------------------------------------------------------------------------------ 2

// Memory-level parallelism: every global-memory loop below issues U independent loads per thread
// before it consumes the first one (a load -> use -> store loop keeps one load per wave in
// flight and runs at a fraction of the HBM rate; MI355X wants ~32 KiB in flight per CU).
constexpr int FE_U = 8;

// dst[i] = src[i] for i in [0, n): global -> LDS, batched
__device__ __forceinline__ void copy_g2s(float* dst, const float* src, int n) {
    for (int base = 0; base < n; base += FE_THREADS * FE_U) {
        float v[FE_U];
#pragma unroll
        for (int u = 0; u < FE_U; ++u) {
            const int i = base + u * FE_THREADS + threadIdx.x;
            v[u] = src[min(i, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < FE_U; ++u) {
            const int i = base + u * FE_THREADS + threadIdx.x;
            if (i < n) dst[i] = v[u];
        }
    }
}

// Gather the samples a tile of frames needs into LDS: reflect padding of the pre-emphasised signal
// (np.pad(y_preem, n_fft//2, 'reflect') inside librosa.stft).  The amplitude normalisation
// (audio_lib.py:125-126: y *= c, c = norm / mean|y|) is NOT applied here: the whole chain up to the
// power spectrum is linear, so c becomes a dB offset that fe_finalize_kernel adds (20 log10 c on the
// power dB, 40 log10 c on the mel dB, before the amin clamps).  That removes the dependency of this
// kernel on a full pass over the utterance; instead it returns sum|x| over the samples of its own
// frames [f0*hop, (f0+G)*hop) (requires hop <= n_fft/2, else fe_abssum_kernel runs).
template <int G>
__device__ __forceinline__ float load_tile(const FeArgs& a, int b, int L, int f0, float* xs) {
    const float* x = a.wav + (size_t)b * a.wav_stride;
    const int half = a.n_fft / 2;
    const int base = f0 * a.hop - half;
    const int own_lo = f0 * a.hop, own_hi = min((f0 + G) * a.hop, L);
    float asum = 0.0f;
    for (int i0 = 0; i0 < a.span; i0 += FE_THREADS * FE_U) {
        float cur[FE_U], prev[FE_U];
#pragma unroll
        for (int u = 0; u < FE_U; ++u) {
            const int idx = base + i0 + u * FE_THREADS + threadIdx.x;
            int j = idx < 0 ? -idx : (idx >= L ? 2 * (L - 1) - idx : idx);
            j = min(max(j, 0), L - 1);
            cur[u] = x[j];
            prev[u] = x[max(j - 1, 0)];
        }
#pragma unroll
        for (int u = 0; u < FE_U; ++u) {
            const int i = i0 + u * FE_THREADS + threadIdx.x;
            const int idx = base + i;
            if (i < a.span) {
                float v = 0.0f;
                if (idx < L + half) {
                    const int j = idx < 0 ? -idx : (idx >= L ? 2 * (L - 1) - idx : idx);
                    const float c = cur[u];
                    const float pv = (j > 0) ? prev[u] : 0.0f;
                    v = (a.pre_emph != 0.0f) ? (c - a.pre_emph * pv) : c;
                }
                xs[i] = v;
                if (idx >= own_lo && idx < own_hi) asum += fabsf(cur[u]);
            }
        }
    }
    return asum;
}

// Power tile in LDS -> raw power dB (global), sparse mel -> raw mel dB (workspace), tile stats.
// CNB / CNM: compile-time bin / mel counts (0 = take them from the arguments).  The per-element
// loops below index by division; with runtime divisors those divisions were most of the kernel's
// instructions (the front-end kernels are issue-bound, not HBM-bound: DESIGN.md section 6).
template <int G, int CNB, int CNM>
__device__ __forceinline__ void power_epilogue(const FeArgs& a, int b, int f0, int F, const float* P,
                                               int pstride, const float* melw, const int32_t* mstart,
                                               const int32_t* moff, float* red, float asum) {
    const int tid = threadIdx.x;
    const int nvalid = min(G, F - f0);
    const int n_bins = CNB ? CNB : a.n_bins, n_mels = CNM ? CNM : a.n_mels;
    float pmax = NEG_INF, pmin = POS_INF, mmax = NEG_INF, mmin = POS_INF;
    {
        float* out = a.pow_db + ((size_t)b * a.max_frames + f0) * n_bins;
        const int total = nvalid * n_bins;
        for (int idx = tid; idx < total; idx += FE_THREADS) {
            const int g = idx / n_bins, k = idx - g * n_bins;
            const float db = db10(fmaxf(1e-30f, P[g * pstride + k]));   // amin clamp comes with the amplitude offset
            out[idx] = db;
            pmax = fmaxf(pmax, db);
            pmin = fminf(pmin, db);
        }
    }
    {
        float* out = a.mel_raw + ((size_t)b * a.max_frames + f0) * n_mels;
        const int total = nvalid * n_mels;
        for (int idx = tid; idx < total; idx += FE_THREADS) {
            const int g = idx / n_mels, m = idx - g * n_mels;
            const float* p = P + g * pstride + mstart[m];
            const int o = moff[m], cnt = moff[m + 1] - o;
            float acc = 0.0f;
            for (int j = 0; j < cnt; ++j) acc = fmaf(melw[o + j], p[j], acc);
            // amplitude_to_db applied to the mel POWER (audio_lib.py:172):
            // 10 log10(max(1e-10, acc^2)) == 20 log10(max(1e-5, |acc|)); clamp applied in fe_finalize_kernel
            const float db = 2.0f * db10(fmaxf(1e-18f, fabsf(acc)));
            out[idx] = db;
            mmax = fmaxf(mmax, db);
            mmin = fminf(mmin, db);
        }
    }
    pmax = vc::wave_max(pmax); pmin = vc::wave_min(pmin);
    mmax = vc::wave_max(mmax); mmin = vc::wave_min(mmin);
    asum = vc::wave_sum(asum);
    const int w = tid >> 6;
    if ((tid & 63) == 0) {
        red[w * 5 + 0] = pmax; red[w * 5 + 1] = pmin; red[w * 5 + 2] = mmax; red[w * 5 + 3] = mmin; red[w * 5 + 4] = asum;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int i = 1; i < FE_THREADS / vc::WAVE; ++i) {
            pmax = fmaxf(pmax, red[i * 5 + 0]); pmin = fminf(pmin, red[i * 5 + 1]);
            mmax = fmaxf(mmax, red[i * 5 + 2]); mmin = fminf(mmin, red[i * 5 + 3]);
            asum += red[i * 5 + 4];
        }
        float* s = a.stats + ((size_t)b * a.ntiles + blockIdx.x) * FE_STAT;
        s[0] = pmax; s[1] = pmin; s[2] = mmax; s[3] = mmin; s[4] = asum;
    }
}

__device__ __forceinline__ void write_neutral_stats(const FeArgs& a, int b) {
    if (threadIdx.x == 0) {
        float* s = a.stats + ((size_t)b * a.ntiles + blockIdx.x) * FE_STAT;
        s[0] = NEG_INF; s[1] = POS_INF; s[2] = NEG_INF; s[3] = POS_INF; s[4] = 0.0f;
    }
}

// LDS carve shared by both power kernels: [mel weights | mel start | mel off | red(16) | ...]
__device__ __forceinline__ void stage_mel(const FeArgs& a, float* melw, int32_t* mstart, int32_t* moff) {
    copy_g2s(melw, a.t.mel_w, a.nnz);
    // mel_start | mel_off are adjacent in the plan blob and in LDS: one batched copy of 2*n_mels+1 words
    copy_g2s(reinterpret_cast<float*>(mstart), reinterpret_cast<const float*>(a.t.mel_start), 2 * a.n_mels + 1);
}

// 400-point path: 16 frames per block, 256 threads.
template <int CNM>
__global__ void __launch_bounds__(FE_THREADS)
fe_power400_kernel(FeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int G = FE_G400;
    constexpr int NROW = G * 13;                       // (g,k1) rows
    float* xs = reinterpret_cast<float*>(smem);        // [span]
    float* win = xs + ((a.span + 3) & ~3);             // [400]
    float* tw = win + 400;                             // [2][208]
    float* Are = tw + 416;                             // [NROW*17]
    float* Aim = Are + NROW * A_STRIDE;
    // lds_alias (mel tables small enough to ride in registers meanwhile): the power tile takes over
    // Are and the mel tables Aim once step 3 has consumed them -> 38 KB instead of 53 KB of LDS, four
    // resident blocks per CU instead of three (the kernel is occupancy-, not HBM-bound)
    const bool alias = a.lds_alias != 0;
    float* P = alias ? Are : Aim + NROW * A_STRIDE;    // [G][201]
    float* melw = alias ? Aim : P + G * 201 + 1;
    int32_t* mstart = reinterpret_cast<int32_t*>(melw + a.nnz);
    int32_t* moff = mstart + a.n_mels;
    float* red = reinterpret_cast<float*>(moff + a.n_mels + 1);

    const int b = blockIdx.y, tid = threadIdx.x;
    const int L = utt_len(a, b);
    const int F = 1 + L / a.hop;
    const int f0 = blockIdx.x * G;
    if (f0 >= F) { write_neutral_stats(a, b); return; }

    const float asum = load_tile<G>(a, b, L, f0, xs);
    copy_g2s(win, a.t.window, 816);                     // window[400] | tw400[416] are adjacent in both
    float mreg[4], treg[2];
    const int ntab = 2 * a.n_mels + 1;
    if (alias) {
#pragma unroll
        for (int u = 0; u < 4; ++u) mreg[u] = a.t.mel_w[min(tid + FE_THREADS * u, a.nnz - 1)];
#pragma unroll
        for (int u = 0; u < 2; ++u) treg[u] = reinterpret_cast<const float*>(a.t.mel_start)[min(tid + FE_THREADS * u, ntab - 1)];
    } else {
        stage_mel(a, melw, mstart, moff);
    }
    __syncthreads();

    // step 1+2: thread (g, n2): real 25-point DFT over n1 of the windowed samples, twiddle
    {
        const int g = tid >> 4, n2 = tid & 15;
        const float* xp = xs + g * a.hop + n2;
        float v[25], ar[13], ai[13];
#pragma unroll
        for (int n1 = 0; n1 < 25; ++n1) v[n1] = xp[16 * n1] * win[16 * n1 + n2];
        vcfe::rdft25_13(v, ar, ai);
        const int i0 = A_STRIDE * (g * 13) + n2;
#pragma unroll
        for (int k1 = 0; k1 < 13; ++k1) {
            vcfe::cmul(ar[k1], ai[k1], tw[k1 * 16 + n2], tw[208 + k1 * 16 + n2]);
            Are[i0 + A_STRIDE * k1] = ar[k1];
            Aim[i0 + A_STRIDE * k1] = ai[k1];
        }
    }
    __syncthreads();
    // step 3: thread (g, k1): complex 16-point DFT over n2 -> |Y|^2 into the power tile
    const int g3 = tid / 13, k13 = tid - g3 * 13;
    float yr[16], yi[16];
    if (tid < NROW) {
        float zr[16], zi[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) { zr[n2] = Are[A_STRIDE * tid + n2]; zi[n2] = Aim[A_STRIDE * tid + n2]; }
        vcfe::cdft16(zr, zi, yr, yi);
    }
    if (alias) __syncthreads();                         // every row is in registers before its buffer is reused
    if (tid < NROW) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int bin = vcfe::bin_of(k13, k2);
            if (bin >= 0) P[g3 * 201 + bin] = yr[k2] * yr[k2] + yi[k2] * yi[k2];
        }
    }
    if (alias) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (tid + FE_THREADS * u < a.nnz) melw[tid + FE_THREADS * u] = mreg[u];
#pragma unroll
        for (int u = 0; u < 2; ++u) if (tid + FE_THREADS * u < ntab) reinterpret_cast<float*>(mstart)[tid + FE_THREADS * u] = treg[u];
    }
    __syncthreads();
    power_epilogue<G, 201, CNM>(a, b, f0, F, P, 201, melw, mstart, moff, red, asum);
}

// Generic path (any n_fft): direct DFT from an LDS twiddle table.  O(n_fft^2) per frame; kept
// for the non-default configurations calc_MFCC_input accepts (audio_lib.py:89-104).
__global__ void __launch_bounds__(FE_THREADS)
fe_power_generic_kernel(FeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int G = FE_GGEN;
    const int N = a.n_fft, NB = a.n_bins;
    float* xs = reinterpret_cast<float*>(smem);        // [span]
    float* win = xs + ((a.span + 3) & ~3);             // [N]
    float* twc = win + N;                              // [N] cos
    float* tws = twc + N;                              // [N] -sin
    float* P = tws + N;                                // [G][NB]
    float* melw = P + G * NB;
    int32_t* mstart = reinterpret_cast<int32_t*>(melw + a.nnz);
    int32_t* moff = mstart + a.n_mels;
    float* red = reinterpret_cast<float*>(moff + a.n_mels + 1);

    const int b = blockIdx.y, tid = threadIdx.x;
    const int L = utt_len(a, b);
    const int F = 1 + L / a.hop;
    const int f0 = blockIdx.x * G;
    if (f0 >= F) { write_neutral_stats(a, b); return; }

    const float asum = load_tile<G>(a, b, L, f0, xs);
    copy_g2s(win, a.t.window, N);
    copy_g2s(twc, a.t.twg, 2 * N);                      // cos | -sin
    stage_mel(a, melw, mstart, moff);
    __syncthreads();
    // windowed frames in place is impossible (frames overlap) -> multiply inside the loop
    for (int idx = tid; idx < G * NB; idx += FE_THREADS) {
        const int g = idx / NB, k = idx - g * NB;
        const float* xp = xs + g * a.hop;
        float re = 0.0f, im = 0.0f;
        int m = 0;
        for (int n = 0; n < N; ++n) {
            const float v = xp[n] * win[n];
            re = fmaf(v, twc[m], re);
            im = fmaf(v, tws[m], im);
            m += k;
            if (m >= N) m -= N;
        }
        P[idx] = re * re + im * im;
    }
    __syncthreads();
    power_epilogue<G, 0, 0>(a, b, f0, F, P, NB, melw, mstart, moff, red, asum);
}

// ------------------------------------------------------------------------------------------ 3
template <int CNM, int CNC, int CNB>
__global__ void __launch_bounds__(FE_THREADS)
fe_finalize_kernel(FeArgs a, int g2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int G = FE_G3;
    const int NM = CNM ? CNM : a.n_mels, NC = CNC ? CNC : a.n_mfcc, NB = CNB ? CNB : a.n_bins;
    const int NM4 = (NM + 3) & ~3;                     // rows padded to float4 (pad = 0)
    const int DS = NM4 + 4;                            // basis pitch: 16-B aligned, b128 conflict-free
    float* D = reinterpret_cast<float*>(smem);         // [NC][DS]
    float* Mc = D + NC * DS;                           // [G+2][NM4]  top_db-clipped raw mel dB
    float* Mf = Mc + (G + 2) * NM4;                    // [G+2][NC]  scaled MFCC
    float* M0 = Mf + (((G + 2) * NC + 3) & ~3);        // [NM4] frame-0 clipped mel dB
    float* sc = M0 + NM4;                              // [8] scalars

    const int b = blockIdx.y, tid = threadIdx.x;
    const int L = utt_len(a, b);
    const int F = 1 + L / a.hop;
    const int f0 = blockIdx.x * G;
    const int mw = a.deriv ? 2 * NC : NC;
    const size_t row0 = (size_t)b * a.max_frames;

    if (f0 >= F) {                                     // pure padding tile: zero-fill
        const int nrows = min(G, a.max_frames - f0);
        float* o1 = a.mfcc + (row0 + f0) * mw;
        float* o2 = a.mel_db + (row0 + f0) * NM;
        float* o3 = a.pow_db + (row0 + f0) * NB;
        for (int i = tid; i < nrows * mw; i += FE_THREADS) o1[i] = 0.0f;
        for (int i = tid; i < nrows * NM; i += FE_THREADS) o2[i] = 0.0f;
        for (int i = tid; i < nrows * NB; i += FE_THREADS) o3[i] = 0.0f;
        return;
    }

    // per-utterance max/min from kernel 2's tile partials (wave 0)
    if (tid < vc::WAVE) {
        const int nt = (F + g2 - 1) / g2;
        float pmax = NEG_INF, pmin = POS_INF, mmax = NEG_INF, mmin = POS_INF, asum = 0.0f;
        for (int t = tid; t < nt; t += vc::WAVE) {
            const float* s = a.stats + ((size_t)b * a.ntiles + t) * FE_STAT;
            pmax = fmaxf(pmax, s[0]); pmin = fminf(pmin, s[1]);
            mmax = fmaxf(mmax, s[2]); mmin = fminf(mmin, s[3]);
            asum += s[4];
        }
        if (!a.fused_abs) asum = tid < FE_NPART ? a.partial[b * FE_NPART + tid] : 0.0f;
        pmax = vc::wave_max(pmax); pmin = vc::wave_min(pmin);
        mmax = vc::wave_max(mmax); mmin = vc::wave_min(mmin);
        asum = vc::wave_sum(asum);
        if (tid == 0) {
            // amplitude normalisation as dB offsets (see load_tile): c = norm / mean|x|
            float offp = 0.0f;
            if (a.amp_norm != 1.0f) { const float c = a.amp_norm / (asum / (float)L); offp = 2.0f * db10(c); }
            const float offm = 2.0f * offp;
            // amin clamps of power_to_db / amplitude_to_db: 10 log10(1e-10) = 20 log10(1e-5) = -100 dB
            pmax = fmaxf(pmax + offp, -100.0f); pmin = fmaxf(pmin + offp, -100.0f);
            mmax = fmaxf(mmax + offm, -100.0f); mmin = fmaxf(mmin + offm, -100.0f);
            const float pfloor = fmaxf(pmax - 80.0f, -100.0f), mfloor = fmaxf(mmax - 80.0f, -100.0f);   // top_db = 80
            sc[0] = pfloor; sc[1] = fmaxf(pmin, pfloor);                // floor, min after clip
            sc[2] = mfloor; sc[3] = fmaxf(mmin, mfloor);
            sc[4] = offp; sc[5] = offm;
        }
    }
    for (int base = 0; base < NC * DS; base += FE_THREADS * FE_U) {
        float v[FE_U];
#pragma unroll
        for (int u = 0; u < FE_U; ++u) {
            const int i = min(base + u * FE_THREADS + tid, NC * DS - 1);
            const int r = i / DS, c = i - r * DS;
            v[u] = a.t.dct[r * NM + min(c, NM - 1)];
        }
#pragma unroll
        for (int u = 0; u < FE_U; ++u) {
            const int i = base + u * FE_THREADS + tid;
            if (i < NC * DS) D[i] = (i % DS) < NM ? v[u] : 0.0f;
        }
    }
    __syncthreads();
    const float pfloor = sc[0], pmin_c = sc[1], mfloor = sc[2], mmin_c = sc[3], offp = sc[4], offm = sc[5];
    const int nvalid = min(G, F - f0);
    const int nrows = min(G, a.max_frames - f0);

    // power dB: top_db clip, min shift, scale, clip -- in place (audio_lib.py:157,230-231,239)
    {
        float* o = a.pow_db + (row0 + f0) * NB;
        const int tv = nvalid * NB, tr = nrows * NB;
        for (int base = 0; base < tr; base += FE_THREADS * FE_U) {
            float v[FE_U];
#pragma unroll
            for (int u = 0; u < FE_U; ++u) v[u] = o[min(base + u * FE_THREADS + tid, tr - 1)];
#pragma unroll
            for (int u = 0; u < FE_U; ++u) {
                const int i = base + u * FE_THREADS + tid;
                if (i < tr) {
                    float w = 0.0f;
                    if (i < tv) {
                        w = fmaxf(v[u] + offp, pfloor);
                        if (a.p_norm != 1.0f) w = a.p_norm * (w - pmin_c);
                        if (a.clip) w = fminf(fmaxf(w, -1.0f), 1.0f);
                    }
                    o[i] = w;
                }
            }
        }
    }
    // mel dB tile with a one-frame halo each side (delta needs MFCC[t-1], MFCC[t+1])
    {
        const float* src = a.mel_raw + row0 * NM;
        const int tot = (G + 2) * NM4;
        for (int base = 0; base < tot; base += FE_THREADS * FE_U) {
            float v[FE_U];
#pragma unroll
            for (int u = 0; u < FE_U; ++u) {
                const int i = min(base + u * FE_THREADS + tid, tot - 1);
                const int r = i / NM4, c = min(i - r * NM4, NM - 1);
                const int f = min(max(f0 - 1 + r, 0), F - 1);
                v[u] = src[(size_t)f * NM + c];
            }
#pragma unroll
            for (int u = 0; u < FE_U; ++u) {
                const int i = base + u * FE_THREADS + tid;
                if (i < tot) {
                    const int r = i / NM4, c = i - r * NM4;
                    const int f = f0 - 1 + r;
                    Mc[i] = (c < NM && f >= 0 && f < F) ? fmaxf(v[u] + offm, mfloor) : 0.0f;
                }
            }
        }
    }
    // frame 0's clipped mel row: its first cepstral coefficient is subtracted from every frame
    // (audio_lib.py:221)
    for (int j = tid; j < NM4; j += FE_THREADS) M0[j] = j < NM ? fmaxf(a.mel_raw[row0 * NM + j] + offm, mfloor) : 0.0f;
    __syncthreads();
    {
        float* o = a.mel_db + (row0 + f0) * NM;
        const int tv = nvalid * NM, tr = nrows * NM;
        for (int i = tid; i < tr; i += FE_THREADS) {
            float v = 0.0f;
            if (i < tv) {
                v = Mc[(i / NM + 1) * NM4 + (i % NM)];
                if (a.m_norm != 1.0f) v = a.m_norm * (v - mmin_c);
                if (a.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
            }
            o[i] = v;
        }
    }
    // DCT-II (audio_lib.py:176-179) + first-coefficient shift + scale (:220-224)
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int i = tid; i < (G + 2) * NC; i += FE_THREADS) {
        const int r = i / NC, c = i - r * NC;
        const f4* d = reinterpret_cast<const f4*>(D + c * DS);
        const f4* m = reinterpret_cast<const f4*>(Mc + r * NM4);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;      // float4 LDS reads, 4 independent chains
#pragma unroll 4
        for (int j = 0; j < NM4 / 4; ++j) {
            const f4 dv = d[j], mv = m[j];
            a0 = fmaf(dv[0], mv[0], a0); a1 = fmaf(dv[1], mv[1], a1);
            a2 = fmaf(dv[2], mv[2], a2); a3 = fmaf(dv[3], mv[3], a3);
        }
        float acc = (a0 + a1) + (a2 + a3);
        if (c == 0 && a.first_mfcc) {
            // identical summation order on frame 0 => frame 0's own coefficient cancels to exactly 0
            const f4* m0 = reinterpret_cast<const f4*>(M0);
            float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f;
#pragma unroll 4
            for (int j = 0; j < NM4 / 4; ++j) {
                const f4 dv = d[j], mv = m0[j];
                b0 = fmaf(dv[0], mv[0], b0); b1 = fmaf(dv[1], mv[1], b1);
                b2 = fmaf(dv[2], mv[2], b2); b3 = fmaf(dv[3], mv[3], b3);
            }
            acc -= (b0 + b1) + (b2 + b3);
        }
        if (a.mfcc_norm != 1.0f) acc *= a.mfcc_norm;
        Mf[i] = acc;
    }
    __syncthreads();
    // outputs [MFCC | delta] (audio_lib.py:226-228, 238)
    {
        float* o = a.mfcc + (row0 + f0) * mw;
        const int tr = nrows * mw;
        for (int i = tid; i < tr; i += FE_THREADS) {
            const int g = a.deriv ? i / (2 * NC) : i / NC, c = i - g * mw;      // constant divisors when CNC != 0
            const int f = f0 + g;
            float v = 0.0f;
            if (f < F) {
                if (c < NC) {
                    v = Mf[(g + 1) * NC + c];
                } else if (f >= 1 && f <= F - 2) {
                    v = 2.0f * (Mf[(g + 2) * NC + (c - NC)] - Mf[g * NC + (c - NC)]);
                }
                if (a.clip) v = fminf(fmaxf(v, -1.0f), 1.0f);
            }
            o[i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ host tables
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm=1)  (audio_lib.py:160-166)
void build_mel(int sr, int n_fft, int n_mels, std::vector<double>& w) {
    const int nb = 1 + n_fft / 2;
    w.assign((size_t)n_mels * nb, 0.0);
    std::vector<double> mel_f(n_mels + 2);
    const double lo = hz_to_mel(0.0), hi = hz_to_mel(sr / 2.0);
    const double step = (hi - lo) / (double)(n_mels + 1);          // np.linspace(lo, hi, n_mels + 2)
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(i == n_mels + 1 ? hi : lo + (double)i * step);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < nb; ++k) {
            const double fk = (k == nb - 1) ? sr / 2.0 : (sr / 2.0) * (double)k / (double)(nb - 1);
            const double lower = (fk - mel_f[i]) / fd0, upper = (mel_f[i + 2] - fk) / fd1;
            const double v = std::fmax(0.0, std::fmin(lower, upper));
            w[(size_t)i * nb + k] = v * enorm;
        }
    }
}

// librosa.filters.dct(n_mfcc, n_mels)  (audio_lib.py:176)
void build_dct(int n_mfcc, int n_mels, std::vector<double>& d) {
    d.assign((size_t)n_mfcc * n_mels, 0.0);
    const double PI = 3.14159265358979323846;
    for (int j = 0; j < n_mels; ++j) d[j] = 1.0 / std::sqrt((double)n_mels);
    for (int i = 1; i < n_mfcc; ++i)
        for (int j = 0; j < n_mels; ++j)
            d[(size_t)i * n_mels + j] = std::cos(i * (2 * j + 1) * PI / (2.0 * n_mels)) * std::sqrt(2.0 / n_mels);
}

int validate_cfg(const vc_frontend_cfg* c) {
    VC_REQUIRE(c != nullptr, "cfg is NULL");
    VC_REQUIRE(c->sample_rate > 0 && c->hop_length > 0 && c->win_length > 0, "sample_rate/hop_length/win_length must be > 0");
    VC_REQUIRE(c->n_fft >= c->win_length, "n_fft (%d) must be >= win_length (%d)", c->n_fft, c->win_length);
    VC_REQUIRE(c->n_fft >= 2 && c->n_fft <= 2048 && (c->n_fft % 2) == 0, "n_fft must be even and in [2, 2048], got %d", c->n_fft);
    VC_REQUIRE(c->n_mels >= 1 && c->n_mels <= 512, "n_mels out of range: %d", c->n_mels);
    VC_REQUIRE(c->n_mfcc >= 1 && c->n_mfcc <= c->n_mels, "n_mfcc must be in [1, n_mels], got %d", c->n_mfcc);
    VC_REQUIRE(c->hop_length <= 4096, "hop_length too large: %d", c->hop_length);
    return VC_OK;
}

}  // namespace

struct vc_frontend_plan {
    vc_frontend_cfg cfg;
    int n_bins, mfcc_width, nnz;
    bool fft400;
    bool fast400;                 // shipped configuration: the two-launch path of vc_frontend400.hip
    std::vector<double> mel, dct;
    void* d_blob;
    FeDev dev;
    const float* d_dct_half;      // [n_mfcc][n_mels / 2] (fast400 only)
    // one-launch form (fast400): two sets of per-utterance arrival counters owned by the plan, zero from creation on.
    // Launch n counts in set n & 1 and zeroes the other one, which launch n - 1 used and -- the launches of a plan being
    // ordered on their stream -- has finished with: no memset between launches, and a late arrival of a launch whose
    // waiters had given up still lands before anybody reads that set again.
    unsigned* d_fcount;
    mutable unsigned fused_launches;
};

extern "C" {

int vc_frontend_host_tables(const vc_frontend_cfg* cfg, double* h_mel, double* h_dct) {
    if (int rc = validate_cfg(cfg)) return rc;
    std::vector<double> t;
    if (h_mel) { build_mel(cfg->sample_rate, cfg->n_fft, cfg->n_mels, t); std::memcpy(h_mel, t.data(), t.size() * sizeof(double)); }
    if (h_dct) { build_dct(cfg->n_mfcc, cfg->n_mels, t); std::memcpy(h_dct, t.data(), t.size() * sizeof(double)); }
    return VC_OK;
}

int vc_frontend_plan_create(const vc_frontend_cfg* cfg, const double* h_window, vc_frontend_plan** out_plan) {
    if (int rc = validate_cfg(cfg)) return rc;
    VC_REQUIRE(out_plan != nullptr, "out_plan is NULL");
    vc_frontend_plan* p = new vc_frontend_plan();
    p->cfg = *cfg;
    p->n_bins = 1 + cfg->n_fft / 2;
    p->mfcc_width = cfg->n_mfcc * (cfg->calc_mfcc_derivate ? 2 : 1);
    p->fft400 = (cfg->n_fft == 400);
    p->d_blob = nullptr;
    p->d_fcount = nullptr;
    p->fused_launches = 0;
    build_mel(cfg->sample_rate, cfg->n_fft, cfg->n_mels, p->mel);
    build_dct(cfg->n_mfcc, cfg->n_mels, p->dct);

    const int N = cfg->n_fft, NB = p->n_bins, NM = cfg->n_mels, NC = cfg->n_mfcc;
    const double PI = 3.14159265358979323846;
    // window, centre-padded to n_fft (librosa util.pad_center)
    std::vector<float> win(N, 0.0f);
    const int lpad = (N - cfg->win_length) / 2;
    for (int i = 0; i < cfg->win_length; ++i)
        win[lpad + i] = (float)(h_window ? h_window[i] : 0.5 - 0.5 * std::cos(2.0 * PI * i / cfg->win_length));
    std::vector<float> tw400(416, 0.0f), twg(2 * (size_t)N);
    for (int k1 = 0; k1 < 13; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            const double ang = -2.0 * PI * (double)(n2 * k1) / 400.0;
            tw400[k1 * 16 + n2] = (float)std::cos(ang);
            tw400[208 + k1 * 16 + n2] = (float)std::sin(ang);
        }
    for (int m = 0; m < N; ++m) {
        const double ang = -2.0 * PI * (double)m / (double)N;
        twg[m] = (float)std::cos(ang);
        twg[N + m] = (float)std::sin(ang);
    }
    // sparse mel rows: contiguous [first non-zero, last non-zero]
    std::vector<int32_t> mstart(NM, 0), moff(NM + 1, 0);
    std::vector<float> mw;
    for (int m = 0; m < NM; ++m) {
        int first = -1, last = -1;
        for (int k = 0; k < NB; ++k)
            if (p->mel[(size_t)m * NB + k] != 0.0) { if (first < 0) first = k; last = k; }
        moff[m] = (int32_t)mw.size();
        if (first >= 0) {
            mstart[m] = first;
            for (int k = first; k <= last; ++k) mw.push_back((float)p->mel[(size_t)m * NB + k]);
        }
    }
    moff[NM] = (int32_t)mw.size();
    p->nnz = (int)mw.size();
    std::vector<float> dctf((size_t)NC * NM);
    for (size_t i = 0; i < dctf.size(); ++i) dctf[i] = (float)p->dct[i];

    int max_cnt = 0;
    for (int m = 0; m < NM; ++m) max_cnt = std::max(max_cnt, (int)(moff[m + 1] - moff[m]));
    p->fast400 = p->fft400 && cfg->hop_length == 80 && NM == 80 && NC == 40 && max_cnt <= 14;
    std::vector<float> dcth((size_t)NC * (NM / 2));
    for (int i = 0; i < NC; ++i)
        for (int j = 0; j < NM / 2; ++j) dcth[(size_t)i * (NM / 2) + j] = (float)p->dct[(size_t)i * NM + j];
    // one device blob: window | tw400 | twg | mel_w | dct | (pad to 16 B) dct_half | mel_start | mel_off
    const size_t pre_f = (size_t)N + 416 + 2 * (size_t)N + mw.size() + dctf.size();
    const size_t pad_f = (4 - pre_f % 4) % 4;
    const size_t n_f = pre_f + pad_f + dcth.size();
    const size_t n_i = (size_t)NM + NM + 1;
    std::vector<char> host(n_f * 4 + n_i * 4);
    float* hf = reinterpret_cast<float*>(host.data());
    size_t o = 0;
    const size_t o_win = o; std::memcpy(hf + o, win.data(), N * 4); o += N;
    const size_t o_tw4 = o; std::memcpy(hf + o, tw400.data(), 416 * 4); o += 416;
    const size_t o_twg = o; std::memcpy(hf + o, twg.data(), 2 * (size_t)N * 4); o += 2 * (size_t)N;
    const size_t o_mw = o; if (!mw.empty()) std::memcpy(hf + o, mw.data(), mw.size() * 4); o += mw.size();
    const size_t o_dct = o; std::memcpy(hf + o, dctf.data(), dctf.size() * 4); o += dctf.size();
    o += pad_f;
    const size_t o_dcth = o; std::memcpy(hf + o, dcth.data(), dcth.size() * 4); o += dcth.size();
    int32_t* hi = reinterpret_cast<int32_t*>(hf + o);
    std::memcpy(hi, mstart.data(), NM * 4);
    std::memcpy(hi + NM, moff.data(), (NM + 1) * 4);

    hipError_t e = hipMalloc(&p->d_blob, host.size());
    if (e != hipSuccess) { delete p; return vc::set_error(VC_ERR_HIP, "hipMalloc(%zu) failed: %s", host.size(), hipGetErrorString(e)); }
    e = hipMemcpy(p->d_blob, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p->d_blob); delete p; return vc::set_error(VC_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e)); }
    float* df = reinterpret_cast<float*>(p->d_blob);
    p->dev.window = df + o_win;
    p->dev.tw400 = df + o_tw4;
    p->dev.twg = df + o_twg;
    p->dev.mel_w = df + o_mw;
    p->dev.dct = df + o_dct;
    p->d_dct_half = df + o_dcth;
    p->dev.mel_start = reinterpret_cast<int32_t*>(df + o);
    p->dev.mel_off = p->dev.mel_start + NM;
    if (p->fast400) {
        const size_t nb = 2 * (size_t)vc_fe400_fused_count_bytes(FE400_FUSED_MAX_BATCH);
        e = hipMalloc(reinterpret_cast<void**>(&p->d_fcount), nb);
        if (e == hipSuccess) e = hipMemset(p->d_fcount, 0, nb);
        if (e != hipSuccess) { (void)hipFree(p->d_blob); delete p; return vc::set_error(VC_ERR_HIP, "front-end counters: %s", hipGetErrorString(e)); }
    }
    *out_plan = p;
    return VC_OK;
}

void vc_frontend_plan_destroy(vc_frontend_plan* plan) {
    if (!plan) return;
    if (plan->d_blob) (void)hipFree(plan->d_blob);
    if (plan->d_fcount) (void)hipFree(plan->d_fcount);
    delete plan;
}

int32_t vc_frontend_num_frames(const vc_frontend_plan* plan, int32_t n_samples) {
    return plan ? 1 + n_samples / plan->cfg.hop_length : -1;
}
int32_t vc_frontend_mfcc_width(const vc_frontend_plan* plan) { return plan ? plan->mfcc_width : -1; }
int32_t vc_frontend_power_width(const vc_frontend_plan* plan) { return plan ? plan->n_bins : -1; }

int vc_frontend_get_mel(const vc_frontend_plan* plan, double* h_out) {
    VC_REQUIRE(plan && h_out, "NULL argument");
    std::memcpy(h_out, plan->mel.data(), plan->mel.size() * sizeof(double));
    return VC_OK;
}
int vc_frontend_get_dct(const vc_frontend_plan* plan, double* h_out) {
    VC_REQUIRE(plan && h_out, "NULL argument");
    std::memcpy(h_out, plan->dct.data(), plan->dct.size() * sizeof(double));
    return VC_OK;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static void ws_layout(const vc_frontend_plan* p, int32_t batch, int32_t max_samples, size_t* o_partial,
                      size_t* o_stats, size_t* o_mel, size_t* total, int* ntiles, int* max_frames) {
    const int mf = 1 + max_samples / p->cfg.hop_length;
    const int g = p->fft400 ? FE_G400 : FE_GGEN;
    const int nt = (mf + g - 1) / g;
    size_t o = 0;
    *o_partial = o; o = align256(o + (size_t)batch * FE_NPART * 4);
    *o_stats = o;   o = align256(o + (size_t)batch * nt * 8 * 4);
    *o_mel = o;     o = align256(o + (size_t)batch * mf * p->cfg.n_mels * 4);
    *total = o; *ntiles = nt; *max_frames = mf;
}

size_t vc_frontend_workspace_bytes(const vc_frontend_plan* plan, int32_t batch, int32_t max_samples) {
    if (!plan || batch <= 0 || max_samples <= 0) return 0;
    size_t a, b, c, t; int nt, mf;
    ws_layout(plan, batch, max_samples, &a, &b, &c, &t, &nt, &mf);
    return t;
}

int vc_frontend_stages_f32(const vc_frontend_plan* plan, const float* d_wav, const int32_t* d_lens, int32_t batch,
                           int32_t max_samples, int32_t wav_stride, int32_t out_rows, float* d_mfcc, float* d_mel_db, float* d_pow_db,
                           void* d_workspace, size_t workspace_bytes, void* stream, int32_t stage_mask) {
    VC_REQUIRE(plan && d_wav && d_mfcc && d_mel_db && d_pow_db && d_workspace, "NULL argument");
    VC_REQUIRE(batch > 0 && batch <= 65535, "batch out of range: %d", batch);
    VC_REQUIRE(max_samples > plan->cfg.n_fft / 2, "max_samples (%d) must exceed n_fft/2 (%d) for reflect padding", max_samples, plan->cfg.n_fft / 2);
    VC_REQUIRE(wav_stride >= max_samples, "wav_stride (%d) < max_samples (%d)", wav_stride, max_samples);
    size_t o_partial, o_stats, o_mel, total; int ntiles, max_frames;
    ws_layout(plan, batch, max_samples, &o_partial, &o_stats, &o_mel, &total, &ntiles, &max_frames);
    if (workspace_bytes < total) return vc::set_error(VC_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, total);
    if (out_rows == 0) out_rows = max_frames;
    VC_REQUIRE(out_rows > 0 && out_rows <= max_frames, "out_rows (%d) must be in [1, %d]", out_rows, max_frames);
    VC_REQUIRE(plan->fast400 || out_rows == max_frames,
               "out_rows < max_frames needs the two-pass 400-point front-end (n_fft 400, hop 80, 80 mels, 40 cepstra)");

    const vc_frontend_cfg& c = plan->cfg;
    if (plan->fast400) {
        // shipped configuration: statistics pass + feature pass, every output byte written once (vc_frontend400.hip)
        Fe400Args f;
        f.wav = d_wav; f.lens = d_lens; f.max_samples = max_samples; f.wav_stride = wav_stride; f.max_frames = max_frames;
        f.out_rows = out_rows;
        f.win_tw = plan->dev.window;                    // window[400] | tw400[416] are adjacent in the blob
        f.mel_w = plan->dev.mel_w; f.mel_start = plan->dev.mel_start; f.mel_off = plan->dev.mel_off;
        f.dct_half = plan->d_dct_half;
        f.pre_emph = c.pre_emphasis; f.amp_norm = c.mean_abs_amp_norm;
        f.mfcc_norm = c.mfcc_norm_factor; f.m_norm = c.M_dB_norm_factor; f.p_norm = c.P_dB_norm_factor;
        f.first_mfcc = c.mfcc_normaleze_first_mfcc; f.deriv = c.calc_mfcc_derivate; f.clip = c.clip_output;
        char* wsb = static_cast<char*>(d_workspace);
        f.stats = reinterpret_cast<float*>(wsb + o_stats);
        f.mel0 = reinterpret_cast<float*>(wsb + o_mel);
        f.nt1 = ntiles;
        // one-launch form: tile records in the mel slot (batch x max_frames x 80 floats, of which this path uses only frame
        // 0's rows): [mel0 | records]; arrival counters in the plan's own double-buffered block
        const size_t o_rec = align256((size_t)batch * c.n_mels * 4);
        f.fstride = vc_fe400_fused_stride(max_frames);
        f.fstats = reinterpret_cast<float*>(wsb + o_mel + o_rec);
        const bool room = o_rec + (size_t)batch * f.fstride * 4 <= total - o_mel && batch <= FE400_FUSED_MAX_BATCH;
        // One launch unless switched off (vc_set_option("fe_fused", 0): two launches, statistics pass + feature pass) or an
        // utterance is too long for it (vc_fe400_fused_ok).  Measured against the two launches with all rows stored: 23.6 vs
        // 29.8 us at 16 utterances of 4 s, 40.2 vs 46.7 at 32, 70.9 vs 76.6 at 64 (tools/fe_fused_probe.py).
        const bool fused = room && (stage_mask & 6) == 6 && vc_fe400_fused_ok(max_frames) && vc::opt(vc::OPT_FE_FUSED) != 0;
        const unsigned par = plan->fused_launches & 1;
        const size_t set_words = (size_t)vc_fe400_fused_count_bytes(FE400_FUSED_MAX_BATCH) / 4;
        f.fcount = plan->d_fcount + par * set_words;
        f.fcount_other = plan->d_fcount + (par ^ 1) * set_words;
        if (fused) ++plan->fused_launches;
        f.mfcc = d_mfcc; f.mel_db = d_mel_db; f.pow_db = d_pow_db;
        return vc_fe400_launch(f, batch, stage_mask, fused ? 1 : 0, static_cast<hipStream_t>(stream));
    }
    const int G = plan->fft400 ? FE_G400 : FE_GGEN;
    FeArgs a;
    a.t = plan->dev;
    a.wav = d_wav; a.lens = d_lens;
    a.max_samples = max_samples; a.wav_stride = wav_stride; a.max_frames = max_frames;
    a.hop = c.hop_length; a.n_fft = c.n_fft; a.n_bins = plan->n_bins; a.n_mels = c.n_mels; a.n_mfcc = c.n_mfcc; a.nnz = plan->nnz;
    a.pre_emph = c.pre_emphasis; a.amp_norm = c.mean_abs_amp_norm;
    a.mfcc_norm = c.mfcc_norm_factor; a.m_norm = c.M_dB_norm_factor; a.p_norm = c.P_dB_norm_factor;
    a.first_mfcc = c.mfcc_normaleze_first_mfcc; a.deriv = c.calc_mfcc_derivate; a.clip = c.clip_output;
    char* ws = static_cast<char*>(d_workspace);
    a.partial = reinterpret_cast<float*>(ws + o_partial);
    a.stats = reinterpret_cast<float*>(ws + o_stats);
    a.mel_raw = reinterpret_cast<float*>(ws + o_mel);
    a.mfcc = d_mfcc; a.mel_db = d_mel_db; a.pow_db = d_pow_db;
    a.ntiles = ntiles;
    a.span = c.hop_length * (G - 1) + c.n_fft;
    hipStream_t st = static_cast<hipStream_t>(stream);

    a.fused_abs = c.hop_length <= c.n_fft / 2;
    if ((stage_mask & 1) && a.amp_norm != 1.0f && !a.fused_abs)
        hipLaunchKernelGGL(fe_abssum_kernel, dim3(FE_NPART, batch), dim3(FE_THREADS), 0, st, a);

    const size_t span_pad = ((size_t)a.span + 3) & ~(size_t)3;
    const size_t mel_lds = (size_t)plan->nnz * 4 + ((size_t)c.n_mels * 2 + 1) * 4 + 32 * 4;
    if (!(stage_mask & 2)) {
    } else if (plan->fft400) {
        // mel tables ride in registers (4 + 2 words per thread) while the row buffers are live
        a.lds_alias = plan->nnz <= 4 * FE_THREADS && 2 * c.n_mels + 1 <= 2 * FE_THREADS &&
                      (size_t)plan->nnz + 2 * c.n_mels + 1 + 32 <= (size_t)FE_G400 * 13 * A_STRIDE;
        const size_t lds = a.lds_alias ? (span_pad + 400 + 416 + 2 * (size_t)FE_G400 * 13 * A_STRIDE) * 4
                                       : (span_pad + 400 + 416 + 2 * (size_t)FE_G400 * 13 * A_STRIDE + FE_G400 * 201 + 1) * 4 + mel_lds;
        VC_REQUIRE(lds <= 160 * 1024, "hop_length too large for the 400-point kernel's LDS tile (%zu B)", lds);
        if (a.n_mels == 80) hipLaunchKernelGGL(fe_power400_kernel<80>, dim3(ntiles, batch), dim3(FE_THREADS), lds, st, a);
        else hipLaunchKernelGGL(fe_power400_kernel<0>, dim3(ntiles, batch), dim3(FE_THREADS), lds, st, a);
    } else {
        const size_t lds = (span_pad + 3 * (size_t)c.n_fft + (size_t)FE_GGEN * plan->n_bins) * 4 + mel_lds;
        VC_REQUIRE(lds <= 160 * 1024, "n_fft/hop_length too large for the generic kernel's LDS tile (%zu B)", lds);
        if (lds > 64 * 1024)
            VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fe_power_generic_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fe_power_generic_kernel, dim3(ntiles, batch), dim3(FE_THREADS), lds, st, a);
    }
    if (stage_mask & 4) {
        const int nt3 = (max_frames + FE_G3 - 1) / FE_G3;
        const size_t nm4 = ((size_t)c.n_mels + 3) & ~(size_t)3;
        const size_t lds = ((size_t)c.n_mfcc * (nm4 + 4) + (size_t)(FE_G3 + 2) * (nm4 + c.n_mfcc) + 4 + nm4 + 8) * 4;
        VC_REQUIRE(lds <= 160 * 1024, "n_mels/n_mfcc too large for the finalize kernel's LDS tile (%zu B)", lds);
        if (a.n_mels == 80 && a.n_mfcc == 40 && a.n_bins == 201)
            hipLaunchKernelGGL((fe_finalize_kernel<80, 40, 201>), dim3(nt3, batch), dim3(FE_THREADS), lds, st, a, G);
        else hipLaunchKernelGGL((fe_finalize_kernel<0, 0, 0>), dim3(nt3, batch), dim3(FE_THREADS), lds, st, a, G);
    }
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_frontend_f32(const vc_frontend_plan* plan, const float* d_wav, const int32_t* d_lens, int32_t batch,
                    int32_t max_samples, int32_t wav_stride, int32_t out_rows, float* d_mfcc, float* d_mel_db, float* d_pow_db,
                    void* d_workspace, size_t workspace_bytes, void* stream) {
    return vc_frontend_stages_f32(plan, d_wav, d_lens, batch, max_samples, wav_stride, out_rows, d_mfcc, d_mel_db, d_pow_db,
                                  d_workspace, workspace_bytes, stream, 7);
}

}  // extern "C"
