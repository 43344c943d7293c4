// Signal front-end, shipped configuration (n_fft = 400, hop = 80, 80 mels, 40 cepstra): two launches,
// every output byte written exactly once (/root/reference/audio_lib.py:89-244).
//
//   pass 1  fe400_kernel<true>    16 frames per block: gather + reflect + pre-emphasis -> windowed 400-point real
//                                 DFT (25 x 16 split, fe_dft400.h) -> |.|^2 -> sparse Slaney mel; keeps ONLY the
//                                 tile's max / min of the power and of the mel power (linear: the dB maps are
//                                 monotone), sum|x| of its own samples, and frame 0's mel row.  Reads 320 B/frame,
//                                 writes 32 B per tile.
//   pass 2  fe400_kernel<false>   16 frames per block (14 output frames + one halo frame each side for the deltas):
//                                 the same transform again, then -- with the utterance's extremes known --
//                                 power_to_db + top_db clip + min shift + scale + clip -> P_dB, amplitude_to_db of
//                                 the mel POWER (the reference's quirk, audio_lib.py:172) -> M_dB, DCT-II (using
//                                 D[i][79-j] = (-1)^i D[i][j]: 40 instead of 80 products per coefficient), frame-0
//                                 shift, scale, delta, clip -> MFCC.  Reads 320 B/frame again (L2 / Infinity Cache
//                                 hits), writes 1,444 B/frame once.
// HBM traffic: 2 x 320 + 1,444 = 2,084 B/frame against 1,764 algorithmic (1.18x); the three-launch form in
// vc_frontend.hip (kept for every other configuration) moves the raw dB tiles through HBM twice (2.4x).
// Recomputing the transform is the cheaper side of that trade: the kernels are issue- and latency-bound, not
// byte-bound, and the transform is ~40 % of a pass's instructions.
#include <hip/hip_runtime.h>
#include "vc_common.h"
#include "fe_dft400.h"
#include "vc_frontend400.h"

namespace {

constexpr int NT = 256;
constexpr int G = 16;                 // frames transformed per block
constexpr int GO = 14;                // output frames per block in pass 2
constexpr int HOP = 80, NFFT = 400, HALF = 200, NB = 201, NM = 80, NC = 40, NH = 40;
constexpr int WSPAN = HOP * 3 + NFFT;               // 640 samples under a wave's 4 frames
constexpr int RP = 20;                              // row pitch (floats): 16 + 4 pad -> ds_read_b128 of 16 consecutive rows
                                                    // conflict-free (5 r mod 16 is a permutation), every address an immediate
constexpr int WROWS = 52;                           // (frame, k1) rows of one wave: 4 frames x 13
constexpr int RE_W = WROWS * RP;                    // 1040 floats of real parts per wave
constexpr int IM_W = 48 * RP;                       // 960 floats of imaginary parts per wave (k1 = 0 is real: not stored)
constexpr int PP = RE_W / 4;                        // 260: pitch of a power-tile row (4 rows per wave region, 59 floats of slack each)
constexpr float NEG_INF = -3.402823466e38f, POS_INF = 3.402823466e38f;
constexpr float DB10 = 3.0102999566398120f;         // 10 log10(x) = DB10 * log2(x)

typedef float f4 __attribute__((ext_vector_type(4)));

// In-kernel phase stamps (s_memtime), -DVC_ABLATE builds only (tools/fe_phase_stamps.py reads them back): wave 0 of
// every block writes the shader clock at each phase boundary into an unused part of the workspace.  The shipped
// library contains none of this.
#ifdef VC_ABLATE
#define FE_STAMP(i)                                                                                         \
    do {                                                                                                    \
        if (threadIdx.x == 0)                                                                               \
            reinterpret_cast<unsigned long long*>(a.mel0 + 65536)[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = \
                __builtin_amdgcn_s_memtime();                                                               \
    } while (0)
#else
#define FE_STAMP(i) do { } while (0)
#endif

// LDS carve (floats): A_re [4 waves][52 rows] | A_im [4 waves][48 rows] | scalars.
// Everything up to the power tile is WAVE-LOCAL: a wave loads the 640 samples under its own 4 frames (into its own
// imaginary-row region, which it overwrites only after reading them: LDS operations of one wave execute in order), runs
// the 25-point stage on them, reads its own 52 rows back for the 16-point stage and writes its 4 power rows over its own
// real-row region.  No workgroup barrier before "power tile complete", so the four waves of a block -- and the blocks of
// a CU -- drift apart and one wave's load latency runs under another's arithmetic.  (Measured alternatives, all slower
// or equal: samples straight from global memory in the 25-point stage, 6-12 % slower (50 loads per thread against 14);
// a fifth block per CU by XOR-swizzled 16-float rows, no change: resident blocks are not the limit.)
constexpr int O_ARE = 0, O_AIM = 4 * RE_W, O_SC = O_AIM + 4 * IM_W, LDS_FLOATS = O_SC + 48;      // sc: [0, 9) constants, [16, 36) wave partials, [40] flag
static_assert(WSPAN <= IM_W, "a wave's samples alias its imaginary rows");
static_assert(PP >= NB + 14, "a power row plus the mel loop's over-read fit the row pitch");
static_assert(G * NC + 2 * G * NM <= 4 * IM_W, "cepstra + sum/difference + mel dB tiles alias the imaginary rows");
static_assert(4 * LDS_FLOATS * 4 <= 160 * 1024, "four blocks per CU");

__device__ __forceinline__ int utt_len(const Fe400Args& a, int b) {
    const int L = a.lens ? a.lens[b] : a.max_samples;
    return min(max(L, 1), a.max_samples);
}

// amplitude_to_db applied to the mel POWER (audio_lib.py:172: 10 log10(max(1e-10, v^2)) = 20 log10(max(1e-5, v))), the
// amplitude-normalisation offset, and the top_db floor.  One function: frame 0's reference value and the tile's own
// values must round identically.  (The amin clamp -100 dB is below every floor: floor >= -100.)
__device__ __forceinline__ float mel_db_clipped(float v, float offm, float mfloor) {
    return fmaxf(fmaf(2.0f * DB10, __log2f(fmaxf(v, 1e-18f)), offm), mfloor);
}

// power_to_db (audio_lib.py:157: 10 log10(max(1e-10, P)), then the top_db floor) with the amplitude-normalisation offset;
// the amin clamp sits 200 dB under the 1e-30 used here once the offset is added back, i.e. below every floor >= -100.
__device__ __forceinline__ float pow_db_clipped(float v, float offp, float pfloor) {
    return fmaxf(fmaf(DB10, __log2f(fmaxf(v, 1e-30f)), offp), pfloor);
}

// One filter of the sparse mel matrix on one frame: 14 reads issued together (reads past the filter's own count meet
// zero weights; past the tile's last row they meet the zero pad), then one FMA chain.  A per-term `if (j < count)`
// here serialised every LDS round trip behind a branch: 10,280 of a block's 20,810 cycles (s_memtime stamps).
__device__ __forceinline__ float mel_dot(const float* p, const float (&w)[14]) {
    float v[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) v[j] = p[j];
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 14; ++j) acc = fmaf(w[j], v[j], acc);
    return acc;
}

// float index of element n2 of row r in a row buffer
__device__ __forceinline__ int row_at(int r, int n2) { return r * RP + n2; }

template <bool STATS>
__global__ void __launch_bounds__(NT, 4)
fe400_kernel(Fe400Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Are = smem + O_ARE;
    float* const Aim = smem + O_AIM;
    float* const sc = smem + O_SC;
    float* const Pt = Are;                              // power tile: row g at Pt + g * PP (4 rows per wave region)

    const int tid = threadIdx.x, b = blockIdx.y;
    const int L = utt_len(a, b);
    const int F = 1 + L / HOP;
    // first transformed frame of the tile (pass 2: one halo frame ahead of the first output frame)
    const int fo = (STATS ? G : GO) * (int)blockIdx.x;          // first frame this block is responsible for
    const int f0 = STATS ? fo : fo - 1;
    const size_t row0 = (size_t)b * a.out_rows;          // the output arrays hold out_rows <= max_frames rows per utterance

    if (fo >= F) {
        if constexpr (STATS) {
            if (tid == 0) {
                float* s = a.stats + ((size_t)b * a.nt1 + blockIdx.x) * 8;
                s[0] = NEG_INF; s[1] = POS_INF; s[2] = NEG_INF; s[3] = POS_INF; s[4] = 0.0f;
            }
        } else {                                        // padding rows of a ragged batch: zeros
            const int nrows = min(GO, a.out_rows - fo);
            const int mw = a.deriv ? 2 * NC : NC;
            float* o1 = a.mfcc + (row0 + fo) * mw;
            float* o2 = a.mel_db + (row0 + fo) * NM;
            float* o3 = a.pow_db + (row0 + fo) * NB;
            for (int i = tid; i < nrows * mw; i += NT) o1[i] = 0.0f;
            for (int i = tid; i < nrows * NM; i += NT) o2[i] = 0.0f;
            for (int i = tid; i < nrows * NB; i += NT) o3[i] = 0.0f;
        }
        return;
    }

    FE_STAMP(0);
    // ---------------- tables this thread needs in registers (L2 hits; issued before anything waits)
    const int n2 = tid & 15;
    float wreg[25];
#pragma unroll
    for (int n1 = 0; n1 < 25; ++n1) wreg[n1] = a.win_tw[16 * n1 + n2];
    // sparse mel row of this thread (m, frame group): its descriptor now, its <= 14 weights behind the 25-point stage's
    // operand reads -- two dependent L2 round trips that used to sit, exposed, in front of the mel loop
    const int mm = tid % NM, mg = tid / NM;             // mg = 3: idle lanes of the last wave
    int ms = 0, mo = 0, mcnt = 0;
    if (mg < 3) {
        ms = a.mel_start[mm];
        mo = a.mel_off[mm];
        mcnt = a.mel_off[mm + 1] - mo;
    }

    // ---------------- pass 2, wave 0: the utterance's tile records and frame 0's mel row are requested FIRST, so that
    // their round trips run beside the block's sample loads; the constants are computed while the samples are in flight
    float r_pmx = NEG_INF, r_pmn = POS_INF, r_mmx = NEG_INF, r_mmn = POS_INF, r_as = 0.0f, r_v0 = 1.0f, r_v1 = 1.0f;
    if constexpr (!STATS) {
        if (tid < 64) {
            const int nt = (F + G - 1) / G;
            for (int t = tid; t < nt; t += 64) {
                const float* s = a.stats + ((size_t)b * a.nt1 + t) * 8;
                r_pmx = fmaxf(r_pmx, s[0]); r_pmn = fminf(r_pmn, s[1]);
                r_mmx = fmaxf(r_mmx, s[2]); r_mmn = fminf(r_mmn, s[3]);
                r_as += s[4];
            }
            if (a.first_mfcc && tid < NH) {
                r_v0 = a.mel0[(size_t)b * NM + tid];
                r_v1 = a.mel0[(size_t)b * NM + NM - 1 - tid];
            }
        }
    }
    // ---------------- samples: reflect padding of the pre-emphasised signal (np.pad(y_preem, 200, 'reflect'))
    // The amplitude normalisation (audio_lib.py:125-126: y *= norm / mean|y|) is linear all the way to the power
    // spectrum, so it is applied as a dB offset once mean|y| is known (pass 2).
    const float* x = a.wav + (size_t)b * a.wav_stride;
    const int wv = tid >> 6, lane = tid & 63;           // wave, lane: the wave owns frames f0 + 4 wv .. f0 + 4 wv + 3
    const int gl = lane >> 4;                           // frame within the wave (== g & 3)
    float* const are_w = Are + wv * RE_W;               // this wave's 52 real rows (later: its 4 power rows)
    float* const aim_w = Aim + wv * IM_W;               // this wave's 48 imaginary rows (first: its 640 samples)
    float* const xs = aim_w;
    const int fw = f0 + 4 * wv;                         // first frame of the wave
    const int base = fw * HOP - HALF;
    float asum = 0.0f;
    {
        const bool interior = base >= 1 && base + WSPAN <= L;   // wave-uniform: no reflection, no clamping
        const float pe = a.pre_emph;
        float cur[10], prv[10];
        if (interior) {
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                cur[u] = x[base + lane + 64 * u];
                prv[u] = x[base + lane + 64 * u - 1];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int idx = base + lane + 64 * u;
                int j = idx < 0 ? -idx : (idx >= L ? 2 * (L - 1) - idx : idx);
                j = min(max(j, 0), L - 1);
                cur[u] = x[j];
                prv[u] = j > 0 ? x[j - 1] : 0.0f;               // lfilter's zero initial state: y[0] = x[0]
            }
        }
        if constexpr (!STATS) {
            if (tid < 64) {
                const float pmx = vc::wave_max(r_pmx), pmn = vc::wave_min(r_pmn);
                const float mmx = vc::wave_max(r_mmx), mmn = vc::wave_min(r_mmn);
                const float as = vc::wave_sum(r_as);
                // amplitude normalisation as dB offsets: c = norm / mean|x| -> + 20 log10 c on the power dB, + 40 log10 c on
                // the mel dB; then the amin clamps (10 log10 1e-10 = 20 log10 1e-5 = -100 dB) and top_db = 80
                float offp = 0.0f;
                if (a.amp_norm != 1.0f) offp = 2.0f * DB10 * __log2f(a.amp_norm / (as / (float)L));
                const float offm = 2.0f * offp;
                // (the same two functions the elements go through below: the utterance's minimum then maps to exactly 0)
                const float pfloor = fmaxf(pow_db_clipped(pmx, offp, -100.0f) - 80.0f, -100.0f);
                const float mfloor = fmaxf(mel_db_clipped(mmx, offm, -100.0f) - 80.0f, -100.0f);
                const float pmin_c = pow_db_clipped(pmn, offp, pfloor), mmin_c = mel_db_clipped(mmn, offm, mfloor);
                // the min shift and the scale are skipped at factor 1.0 (audio_lib.py:230-235)
                const bool pn = a.p_norm != 1.0f, mn = a.m_norm != 1.0f;
                if (tid == 0) {
                    sc[0] = offp; sc[1] = pfloor;
                    sc[2] = pn ? a.p_norm : 1.0f;
                    sc[8] = pn ? pmin_c : 0.0f;
                    sc[3] = offm; sc[4] = mfloor;
                    sc[5] = mn ? a.m_norm : 1.0f;
                    sc[6] = mn ? mmin_c : 0.0f;
                }
                // frame 0's first cepstral coefficient (audio_lib.py:221), summed exactly like the tile's own
                // coefficient 0 below so that frame 0's own value cancels to 0
                float c00 = 0.0f;
                if (a.first_mfcc) {
                    const float d0 = mel_db_clipped(r_v0, offm, mfloor), d1 = mel_db_clipped(r_v1, offm, mfloor);
                    const float s0 = d0 + d1;
                    const float dc = a.dct_half[0];                     // row 0 is constant: 1 / sqrt(80)
                    float acc4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < NH; ++j) {
                        const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s0), j));
                        acc4[j & 3] = fmaf(dc, sj, acc4[j & 3]);
                    }
                    c00 = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                }
                if (tid == 0) sc[7] = c00;
            }
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int i = lane + 64 * u;
            const int idx = base + i;
            // beyond the reflected tail (idx >= L + 200) no frame of this utterance reads
            xs[i] = idx < L + HALF ? cur[u] - pe * prv[u] : 0.0f;
            if constexpr (STATS) {
                // sum|x| of the wave's OWN hops [fw*80, (fw+4)*80): every sample of the utterance counted once
                if (idx >= fw * HOP && idx < min((fw + 4) * HOP, L)) asum += fabsf(cur[u]);
            }
        }
    }
    FE_STAMP(1);

    float mw_[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) mw_[j] = 0.0f;
    // ---------------- steps 1 + 2: lane (gl, n2): real 25-point DFT over n1, twiddle W400^(n2 k1).  Wave-local: the
    // wave's own LDS writes above are ordered before these reads, and every read of the samples is issued before the
    // row stores that overwrite them (same wave, in order).
    {
        const float* xp = xs + gl * HOP + n2;
        float v[25], ar[13], ai[13];
#pragma unroll
        for (int n1 = 0; n1 < 25; ++n1) v[n1] = xp[16 * n1] * wreg[n1];
        // twiddles and this thread's mel weights: requested here, consumed behind the 25-point transform
        if (mg < 3) {
#pragma unroll
            for (int j = 0; j < 14; ++j) mw_[j] = a.mel_w[mo + min(j, max(mcnt - 1, 0))];
        }
        float twr[13], twi[13];
#pragma unroll
        for (int k1 = 1; k1 < 13; ++k1) { twr[k1] = a.win_tw[400 + k1 * 16 + n2]; twi[k1] = a.win_tw[608 + k1 * 16 + n2]; }
        vcfe::rdft25_13(v, ar, ai);
        // the reads of v[] above must have RETURNED before any row store may land on the sample buffer: the compiler
        // orders them (ar / ai depend on v), the LDS unit executes a wave's operations in order
        const int r0 = gl * 13, i0 = gl * 12 - 1;
        are_w[(r0)*RP + n2] = ar[0];                    // ai[0] == 0: not stored
#pragma unroll
        for (int k1 = 1; k1 < 13; ++k1) {
            vcfe::cmul(ar[k1], ai[k1], twr[k1], twi[k1]);
            are_w[(r0 + k1) * RP + n2] = ar[k1];
            aim_w[(i0 + k1) * RP + n2] = ai[k1];
        }
    }
    FE_STAMP(2);

    // ---------------- step 3: lane = row (g3l, k13) of the wave's 52: complex 16-point DFT over n2 -> |Y|^2
    const int g3l = lane / 13, k13 = lane - g3l * 13;
    const bool row_ok = lane < WROWS;
    const int g3 = 4 * wv + g3l;                        // frame within the tile
    float pw[16];
    {
        float zr[16], zi[16], yr[16], yi[16];
        const int r = row_ok ? lane : 0;
        const bool has_im = row_ok && k13 > 0;
        const int ri = has_im ? g3l * 12 + k13 - 1 : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 vr = *reinterpret_cast<const f4*>(are_w + r * RP + 4 * q);
            const f4 vi = *reinterpret_cast<const f4*>(aim_w + ri * RP + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) { zr[4 * q + e] = vr[e]; zi[4 * q + e] = has_im ? vi[e] : 0.0f; }
        }
        vcfe::cdft16(zr, zi, yr, yi);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) pw[k2] = yr[k2] * yr[k2] + yi[k2] * yi[k2];
    }
    // DCT basis row of this thread's coefficient (rows of librosa.filters.dct(40, 80), first half): requested here,
    // used four phases later
    const int ci = tid / 6, cf = tid - ci * 6;          // coefficient, frame phase (tid < 240)
    float drow[STATS ? 1 : NH];
    if constexpr (!STATS) {
        if (tid < 240) {
#pragma unroll
            for (int j = 0; j < NH; j += 4) {
                const f4 d = *reinterpret_cast<const f4*>(a.dct_half + ci * NH + j);
                drow[j] = d[0]; drow[j + 1] = d[1]; drow[j + 2] = d[2]; drow[j + 3] = d[3];
            }
        }
    }
    FE_STAMP(3);
    // (no barrier: the wave's power rows go over its OWN real rows, all of which are in its registers by now)
    // power tile: bin k1 + 25 k2 directly for k2 <= 7 (and 200 = 0 + 25 * 8); the bins with residue 13..24 are the
    // mirror images 400 - k of the outputs with k2 >= 8 (hermitian symmetry, fe_dft400.h bin_of)
    float pmax = NEG_INF, pmin = POS_INF;
    const bool frame_ok = row_ok && (f0 + g3 >= 0) && (f0 + g3 < F);
    if (row_ok) {
        float* pg = Pt + g3 * PP;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) pg[k13 + 25 * k2] = pw[k2];
        if (k13 == 0) pg[200] = pw[8];
        else {
#pragma unroll
            for (int k2 = 8; k2 < 16; ++k2) pg[400 - 25 * k2 - k13] = pw[k2];
        }
        if constexpr (STATS) {
            if (frame_ok) {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) { pmax = fmaxf(pmax, pw[k2]); pmin = fminf(pmin, pw[k2]); }
                if (k13 == 0) { pmax = fmaxf(pmax, pw[8]); pmin = fminf(pmin, pw[8]); }
                else {
#pragma unroll
                    for (int k2 = 8; k2 < 16; ++k2) { pmax = fmaxf(pmax, pw[k2]); pmin = fminf(pmin, pw[k2]); }
                }
            }
        }
    }

    // ---------------- sparse mel: weights beyond the filter's own count are zeroed (the loads above clamp their index)
#pragma unroll
    for (int j = 0; j < 14; ++j) mw_[j] = j < mcnt ? mw_[j] : 0.0f;
    if constexpr (STATS) {
        float mmax = NEG_INF, mmin = POS_INF;
        FE_STAMP(4);
        __syncthreads();                                // power tile complete
        FE_STAMP(5);
        if (mg < 3) {
            for (int gg = mg; gg < G; gg += 3) {
                const float acc = mel_dot(Pt + gg * PP + ms, mw_);
                if (f0 + gg < F) { mmax = fmaxf(mmax, acc); mmin = fminf(mmin, acc); }
                if (f0 + gg == 0) a.mel0[(size_t)b * NM + mm] = acc;
            }
        }
        pmax = vc::wave_max(pmax); pmin = vc::wave_min(pmin);
        mmax = vc::wave_max(mmax); mmin = vc::wave_min(mmin);
        asum = vc::wave_sum(asum);
        const int w = tid >> 6;
        if ((tid & 63) == 0) { sc[w * 5 + 0] = pmax; sc[w * 5 + 1] = pmin; sc[w * 5 + 2] = mmax; sc[w * 5 + 3] = mmin; sc[w * 5 + 4] = asum; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int i = 1; i < NT / 64; ++i) {
                pmax = fmaxf(pmax, sc[i * 5 + 0]); pmin = fminf(pmin, sc[i * 5 + 1]);
                mmax = fmaxf(mmax, sc[i * 5 + 2]); mmin = fminf(mmin, sc[i * 5 + 3]);
                asum += sc[i * 5 + 4];
            }
            float* s = a.stats + ((size_t)b * a.nt1 + blockIdx.x) * 8;
            s[0] = pmax; s[1] = pmin; s[2] = mmax; s[3] = mmin; s[4] = asum;
        }
        FE_STAMP(6);
        return;
    } else {
        float* const Mc = Aim + G * NC + G * NM;        // [G][80]  clipped mel dB
        float* const Mf = Aim;                          // [G][40]  scaled cepstra
        float* const SD = Aim + G * NC;                 // [G][80]  j < 40: m[j] + m[79-j], j >= 40: m[j-40] - m[119-j]
        FE_STAMP(4);
        __syncthreads();                                // power tile + constants
        FE_STAMP(5);
        const float offp = sc[0], pfloor = sc[1], pS = sc[2], pM = sc[8], offm = sc[3], mfloor = sc[4], mS = sc[5], mM = sc[6],
                    c00 = sc[7];
        const int nvalid = min(GO, F - fo);             // output frames that exist
        const int nrows = min(GO, a.out_rows - fo);     // output rows of the buffers (the rest of them: zeros); <= 0 past out_rows

        // ---------------- P_dB: thread k < 201 walks column k of tile rows 1..14 (every LDS address an immediate, the
        // global stores of a wave are consecutive floats)
        if (tid < NB) {
            float* o = a.pow_db + (row0 + fo) * NB + tid;
            const float* p = Pt + PP + tid;
            const bool clip = a.clip != 0;
#pragma unroll
            for (int gg = 0; gg < GO; ++gg) {
                float w = pS * (pow_db_clipped(p[gg * PP], offp, pfloor) - pM);
                if (clip) w = fminf(fmaxf(w, -1.0f), 1.0f);
                if (gg < nrows) o[gg * NB] = gg < nvalid ? w : 0.0f;
            }
        }
        FE_STAMP(6);
        // ---------------- mel power -> dB of the "amplitude" (quirk) -> top_db clip: all 16 frames (DCT halo)
        if (mg < 3) {
            for (int gg = mg; gg < G; gg += 3) {
                Mc[gg * NM + mm] = mel_db_clipped(mel_dot(Pt + gg * PP + ms, mw_), offm, mfloor);
            }
        }
        __syncthreads();
        FE_STAMP(7);
        // ---------------- M_dB out (float4 rows) and the sum / difference halves for the DCT
        {
            const bool clip = a.clip != 0;
            f4* o = reinterpret_cast<f4*>(a.mel_db + (row0 + fo) * NM);
            for (int i = tid; i < nrows * (NM / 4); i += NT) {
                f4 v = *reinterpret_cast<const f4*>(Mc + NM + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w = mS * (v[e] - mM);
                    if (clip) w = fminf(fmaxf(w, -1.0f), 1.0f);
                    v[e] = 4 * i < nvalid * NM ? w : 0.0f;
                }
                o[i] = v;
            }
            for (int i = tid; i < G * NH; i += NT) {
                const int gg = i / NH, j = i - gg * NH;
                const float lo = Mc[gg * NM + j], hi = Mc[gg * NM + NM - 1 - j];
                SD[gg * NM + j] = lo + hi;
                SD[gg * NM + NH + j] = lo - hi;
            }
        }
        __syncthreads();
        FE_STAMP(8);
        // ---------------- DCT-II: coefficient ci (even: sums, odd: differences), frames cf, cf + 6, cf + 12
        if (tid < 240) {
            const float norm = a.mfcc_norm;
            for (int gg = cf; gg < G; gg += 6) {
                const f4* sd = reinterpret_cast<const f4*>(SD + gg * NM + (ci & 1) * NH);
                float acc4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < NH / 4; ++j) {
                    const f4 s = sd[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc4[e] = fmaf(drow[4 * j + e], s[e], acc4[e]);
                }
                float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                if (ci == 0) acc -= c00;
                if (norm != 1.0f) acc *= norm;
                Mf[gg * NC + ci] = acc;
            }
        }
        __syncthreads();
        FE_STAMP(9);
        // ---------------- [MFCC | delta] out (audio_lib.py:226-228, 238): delta = 2 (M[t+1] - M[t-1]), 0 at both ends
        {
            const bool clip = a.clip != 0;
            const int mw = a.deriv ? 2 * NC : NC;
            f4* o = reinterpret_cast<f4*>(a.mfcc + (row0 + fo) * mw);
            const int per_row = mw / 4;
            for (int i = tid; i < nrows * per_row; i += NT) {
                const int gg = i / per_row, c = 4 * (i - gg * per_row);
                const int f = fo + gg;
                f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (f < F) {
                    if (c < NC) {
                        v = *reinterpret_cast<const f4*>(Mf + (gg + 1) * NC + c);
                    } else if (f >= 1 && f <= F - 2) {
                        const f4 nx = *reinterpret_cast<const f4*>(Mf + (gg + 2) * NC + (c - NC));
                        const f4 pv = *reinterpret_cast<const f4*>(Mf + gg * NC + (c - NC));
                        v = 2.0f * (nx - pv);
                    }
                    if (clip) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], -1.0f), 1.0f);
                    }
                }
                o[i] = v;
            }
        }
        FE_STAMP(10);
    }
}


// ============================================================================================================
// ONE-LAUNCH FORM: every frame transformed once.
//
// The two-pass form transforms every frame twice because the normalisations need the utterance's extremes before a
// single output value can be written.  Here a block transforms its 16 frames (14 outputs + one halo frame each side),
// keeps the power tile and the mel-power tile in LDS, PUBLISHES its tile record (max / min of the power and of the mel
// power, sum|x| of its own hops; the tile holding frame 0 also that frame's mel row), and then waits until all tiles of
// ITS OWN UTTERANCE have published -- a per-utterance counter, not a grid barrier -- before it finishes dB / DCT /
// delta from LDS and stores.  Blocks whose frames lie past `out_rows` only publish and leave.
//
// Hand-off (cdna_hip_programming.md Guideline 16, form R1): record words are write-through (sc1) stores, every storing
// wave drains them (vmcnt(0)), a workgroup barrier, then ONE lane adds to the utterance's counter (agent scope).  The tile
// whose add completes the count reduces the utterance's records to one line (write-through, drained) and raises the
// utterance's READY word; the consumers poll that word relaxed from one lane and read the line with sc1 loads only (they
// bypass this CU's L1: nothing here can be stale; every utterance's records, its line, its counter and its READY word
// sit on cache lines of their own).
//
// Forward progress without any assumption about dispatch order or residency: a publisher never waits, and a waiter
// whose poll runs out (4 ms; an utterance's tiles normally arrive within microseconds of each other because workgroups
// of one utterance are neighbours in the grid) stops waiting and computes the utterance's records ITSELF -- it walks
// all tiles of the utterance through the same transform code, writes their records (the same values their owners
// write), re-transforms its own tile to restore the LDS tiles, and goes on.  Slow, never wrong, never stuck.
constexpr unsigned FUSED_SPIN_LIMIT = 4000;          // x (s_sleep 24 + one L2 round trip) ~ 4 ms
// One counter per 256 bytes: with the 32 counters of a batch in ONE cache line every arrival and every poll of ~1,000
// resident blocks went through one L2 channel (a word serves ~88 requests per microsecond): blocks waited a median of
// 69,000 cycles for their utterance and the launch took 103 us (tools/fe_phase_stamps.py fused; profiles/r03).
constexpr int FCOUNT_PITCH = 64;                     // unsigned words

__device__ __forceinline__ float ld_sc1(const float* p) {
    typedef __attribute__((address_space(1))) unsigned gu32;
    return __uint_as_float(__hip_atomic_load((gu32*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    typedef __attribute__((address_space(1))) unsigned gu32;
    __hip_atomic_store((gu32*)(uintptr_t)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(NT, 4)
fe400_fused_kernel(Fe400Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Are = smem + O_ARE;
    float* const Aim = smem + O_AIM;
    float* const sc = smem + O_SC;
    float* const Pt = Are;                              // power tile: row g at Pt + g * PP (4 rows per wave region)
    float* const Mc = Aim + G * NC + G * NM;            // [G][80]  mel POWER until the utterance's constants are known, then clipped mel dB
    float* const Mf = Aim;                              // [G][40]  scaled cepstra
    float* const SD = Aim + G * NC;                     // [G][80]  j < 40: m[j] + m[79-j], j >= 40: m[j-40] - m[119-j]

    const int tid = threadIdx.x, b = blockIdx.y;
    const int L = utt_len(a, b);
    const int F = 1 + L / HOP;
    const int own = blockIdx.x;
    const int fo_own = GO * own;
    const size_t row0 = (size_t)b * a.out_rows;
    const int mw = a.deriv ? 2 * NC : NC;

    if (fo_own >= F) {                                  // padding rows of a ragged batch: zeros (such tiles publish nothing)
        const int nrows = min(GO, a.out_rows - fo_own);
        float* o1 = a.mfcc + (row0 + fo_own) * mw;
        float* o2 = a.mel_db + (row0 + fo_own) * NM;
        float* o3 = a.pow_db + (row0 + fo_own) * NB;
        for (int i = tid; i < nrows * mw; i += NT) o1[i] = 0.0f;
        for (int i = tid; i < nrows * NM; i += NT) o2[i] = 0.0f;
        for (int i = tid; i < nrows * NB; i += NT) o3[i] = 0.0f;
        return;
    }
    if (own == 0) {                                     // the previous launch's counters, for the launch after this one
        for (int u = b + (int)gridDim.y * tid; u < FE400_FUSED_MAX_BATCH; u += (int)gridDim.y * NT)
            { a.fcount_other[(size_t)u * FCOUNT_PITCH] = 0u; a.fcount_other[(size_t)u * FCOUNT_PITCH + 32] = 0u; }
    }
    const int nt_b = (F + GO - 1) / GO;                 // tiles of this utterance that publish
    const bool has_out = fo_own < a.out_rows;
    float* const recs = a.fstats + (size_t)b * a.fstride;
    typedef __attribute__((address_space(1))) unsigned gu32;
    gu32* const cnt = (gu32*)(uintptr_t)(a.fcount + (size_t)b * FCOUNT_PITCH);

    const float* x = a.wav + (size_t)b * a.wav_stride;

    // mode 0: own tile (publish, wait); 1: walking the utterance's tiles after a poll ran out; 2: own tile again
    int mode = 0, cur = own;
    FE_STAMP(0);
    for (;;) {
        const int fo = GO * cur, f0 = fo - 1;
        // Every per-thread index of the transform is derived INSIDE the loop from an opaque copy of the thread index: the
        // loop runs once unless a poll ran out, but hoisted out of it the tables, LDS addresses and row maps (loop
        // invariants) stay live through the wait and the gather and the kernel spills 60 registers.
        int tid_l = threadIdx.x;
        asm volatile("" : "+v"(tid_l));
        const int n2 = tid_l & 15, n2o = n2;
        const int mm = tid_l % NM, mg = tid_l / NM;         // mg = 3: idle lanes of the last wave
        int ms = 0, moo = 0, mcnt = 0;
        if (mg < 3) {
            ms = a.mel_start[mm];
            moo = a.mel_off[mm];
            mcnt = a.mel_off[mm + 1] - moo;
        }
        const int wv = tid_l >> 6, lane = tid_l & 63;
        const int gl = lane >> 4;
        float* const are_w = Are + wv * RE_W;
        float* const aim_w = Aim + wv * IM_W;
        float* const xs = aim_w;
        const int g3l = lane / 13, k13 = lane - g3l * 13;
        const bool row_ok = lane < WROWS;
        const int g3 = 4 * wv + g3l;
        float wreg[25];
#pragma unroll
        for (int n1 = 0; n1 < 25; ++n1) wreg[n1] = a.win_tw[16 * n1 + n2o];
        // ---------------- samples (see fe400_kernel): wave-local, reflect padding of the pre-emphasised signal
        const int fw = f0 + 4 * wv;
        const int base = fw * HOP - HALF;
        float asum = 0.0f;
        {
            const bool interior = base >= 1 && base + WSPAN <= L;
            const float pe = a.pre_emph;
            float curv[10], prv[10];
            if (interior) {
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    curv[u] = x[base + lane + 64 * u];
                    prv[u] = x[base + lane + 64 * u - 1];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const int idx = base + lane + 64 * u;
                    int j = idx < 0 ? -idx : (idx >= L ? 2 * (L - 1) - idx : idx);
                    j = min(max(j, 0), L - 1);
                    curv[u] = x[j];
                    prv[u] = j > 0 ? x[j - 1] : 0.0f;
                }
            }
            // sum|x| of the tile's OWN hops [fo*80, (fo+14)*80), every sample of the utterance counted once: this wave's
            // share is the part of its four frames' hops that lies inside
            const int alo = max(fw, fo) * HOP, ahi = min(min((fw + 4), fo + GO) * HOP, L);
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int i = lane + 64 * u;
                const int idx = base + i;
                xs[i] = idx < L + HALF ? curv[u] - pe * prv[u] : 0.0f;
                if (idx >= alo && idx < ahi) asum += fabsf(curv[u]);
            }
        }
        if (mode == 0) FE_STAMP(1);                     // samples in LDS
        // ---------------- 25-point stage + twiddle
        float mw_[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) mw_[j] = 0.0f;
        {
            const float* xp = xs + gl * HOP + n2;
            float v[25], ar[13], ai[13];
#pragma unroll
            for (int n1 = 0; n1 < 25; ++n1) v[n1] = xp[16 * n1] * wreg[n1];
            // twiddles and this thread's mel weights: requested here (L2 hits), consumed behind the 25-point transform.  They
            // are NOT kept across iterations of the tile loop (it runs once unless a poll ran out): registers
            if (mg < 3) {
#pragma unroll
                for (int j = 0; j < 14; ++j) mw_[j] = a.mel_w[moo + min(j, max(mcnt - 1, 0))];
            }
            float twr[13], twi[13];
#pragma unroll
            for (int k1 = 1; k1 < 13; ++k1) { twr[k1] = a.win_tw[400 + k1 * 16 + n2o]; twi[k1] = a.win_tw[608 + k1 * 16 + n2o]; }
            vcfe::rdft25_13(v, ar, ai);
            const int r0 = gl * 13, i0 = gl * 12 - 1;
            are_w[(r0)*RP + n2] = ar[0];
#pragma unroll
            for (int k1 = 1; k1 < 13; ++k1) {
                vcfe::cmul(ar[k1], ai[k1], twr[k1], twi[k1]);
                are_w[(r0 + k1) * RP + n2] = ar[k1];
                aim_w[(i0 + k1) * RP + n2] = ai[k1];
            }
        }
        if (mode == 0) FE_STAMP(2);                     // 25-point stage
        // ---------------- 16-point stage -> |Y|^2 -> the wave's four power rows (over its own real rows)
        float pmax = NEG_INF, pmin = POS_INF;
        {
            float pw[16];
            {
                float zr[16], zi[16], yr[16], yi[16];
                const int r = row_ok ? lane : 0;
                const bool has_im = row_ok && k13 > 0;
                const int ri = has_im ? g3l * 12 + k13 - 1 : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 vr = *reinterpret_cast<const f4*>(are_w + r * RP + 4 * q);
                    const f4 vi = *reinterpret_cast<const f4*>(aim_w + ri * RP + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { zr[4 * q + e] = vr[e]; zi[4 * q + e] = has_im ? vi[e] : 0.0f; }
                }
                vcfe::cdft16(zr, zi, yr, yi);
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) pw[k2] = yr[k2] * yr[k2] + yi[k2] * yi[k2];
            }
            const bool frame_ok = row_ok && (f0 + g3 >= 0) && (f0 + g3 < F);
            if (row_ok) {
                float* pg = Pt + g3 * PP;
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) pg[k13 + 25 * k2] = pw[k2];
                if (k13 == 0) pg[200] = pw[8];
                else {
#pragma unroll
                    for (int k2 = 8; k2 < 16; ++k2) pg[400 - 25 * k2 - k13] = pw[k2];
                }
                if (frame_ok) {
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) { pmax = fmaxf(pmax, pw[k2]); pmin = fminf(pmin, pw[k2]); }
                    if (k13 == 0) { pmax = fmaxf(pmax, pw[8]); pmin = fminf(pmin, pw[8]); }
                    else {
#pragma unroll
                        for (int k2 = 8; k2 < 16; ++k2) { pmax = fmaxf(pmax, pw[k2]); pmin = fminf(pmin, pw[k2]); }
                    }
                }
            }
        }
        if (mode == 0) FE_STAMP(3);                     // 16-point stage, power rows
        __syncthreads();                                // power tile complete; every wave is past its imaginary rows
        if (mode == 0) FE_STAMP(4);
        // ---------------- mel power of all 16 frames -> LDS; extremes; frame 0's row
#pragma unroll
        for (int j = 0; j < 14; ++j) mw_[j] = j < mcnt ? mw_[j] : 0.0f;
        float mmax = NEG_INF, mmin = POS_INF;
        if (mg < 3) {
            for (int gg = mg; gg < G; gg += 3) {
                const float acc = mel_dot(Pt + gg * PP + ms, mw_);
                Mc[gg * NM + mm] = acc;
                if (f0 + gg >= 0 && f0 + gg < F) { mmax = fmaxf(mmax, acc); mmin = fminf(mmin, acc); }
                if (f0 + gg == 0 && mode != 2) st_sc1(a.mel0 + (size_t)b * NM + mm, acc);
            }
        }
        pmax = vc::wave_max(pmax); pmin = vc::wave_min(pmin);
        mmax = vc::wave_max(mmax); mmin = vc::wave_min(mmin);
        asum = vc::wave_sum(asum);
        if (lane == 0) { sc[16 + wv * 5 + 0] = pmax; sc[16 + wv * 5 + 1] = pmin; sc[16 + wv * 5 + 2] = mmax; sc[16 + wv * 5 + 3] = mmin; sc[16 + wv * 5 + 4] = asum; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (frame 0's mel row: every storing wave drains)
        __syncthreads();
        if (mode == 0) FE_STAMP(5);                     // mel power, reductions, barrier
        if (mode == 2) break;                           // own tiles restored; the records are all in memory
        if (tid == 0) {
#pragma unroll
            for (int i = 1; i < NT / 64; ++i) {
                pmax = fmaxf(pmax, sc[16 + i * 5 + 0]); pmin = fminf(pmin, sc[16 + i * 5 + 1]);
                mmax = fmaxf(mmax, sc[16 + i * 5 + 2]); mmin = fminf(mmin, sc[16 + i * 5 + 3]);
                asum += sc[16 + i * 5 + 4];
            }
            float* r8 = recs + (size_t)cur * 8;
            st_sc1(r8 + 0, pmax); st_sc1(r8 + 1, pmin); st_sc1(r8 + 2, mmax); st_sc1(r8 + 3, mmin); st_sc1(r8 + 4, asum);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (mode == 0 && wv == 0) {
            // Arrival (wave 0).  The LAST tile of the utterance to arrive reduces the records to one line and raises the
            // utterance's READY word; everybody else polls that word and reads the one line (58 records gathered by
            // every block cost 7 k cycles of a 52 k-cycle block life and most of the L2 traffic the polls compete with).
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
            if (old == (unsigned)(nt_b - 1)) {
                float r_pmx = NEG_INF, r_pmn = POS_INF, r_mmx = NEG_INF, r_mmn = POS_INF, r_as = 0.0f;
                for (int t = lane; t < nt_b; t += 64) {
                    const float* q8 = recs + (size_t)t * 8;
                    r_pmx = fmaxf(r_pmx, ld_sc1(q8 + 0)); r_pmn = fminf(r_pmn, ld_sc1(q8 + 1));
                    r_mmx = fmaxf(r_mmx, ld_sc1(q8 + 2)); r_mmn = fminf(r_mmn, ld_sc1(q8 + 3));
                    r_as += ld_sc1(q8 + 4);
                }
                r_pmx = vc::wave_max(r_pmx); r_pmn = vc::wave_min(r_pmn);
                r_mmx = vc::wave_max(r_mmx); r_mmn = vc::wave_min(r_mmn);
                r_as = vc::wave_sum(r_as);
                if (lane == 0) {
                    float* s8 = recs + a.fstride - 32;
                    st_sc1(s8 + 0, r_pmx); st_sc1(s8 + 1, r_pmn); st_sc1(s8 + 2, r_mmx); st_sc1(s8 + 3, r_mmn); st_sc1(s8 + 4, r_as);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(cnt + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (lane == 0) {
                int ok = 1;
                if (has_out) {
                    ok = 0;
                    for (int spins = 0; spins < a.spin_limit; ++spins) {
                        if (__hip_atomic_load(cnt + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 1; break; }
                        __builtin_amdgcn_s_sleep(16);
                    }
                }
                reinterpret_cast<int*>(sc)[40] = ok;
            }
        }
        if (mode == 0) {
            if (!has_out) return;                       // statistics-only tile (frames past out_rows): published, done
            __syncthreads();
            if (reinterpret_cast<const int*>(sc)[40]) break;
            mode = 1; cur = 0;                          // the poll ran out: compute the utterance's records here
        } else {
            __syncthreads();                            // (sc is rewritten by the next tile)
            if (++cur == nt_b) { mode = 2; cur = own; }
        }
    }

    FE_STAMP(6);                                        // record published, the utterance's tiles waited for
    // ---------------- the utterance's constants (wave 0), from the published records: sc1 loads only
    const int fo = fo_own;
    // (everything below is addressed from an opaque copy of the thread index: left to itself the compiler computes the
    // output addresses of all phases BEFORE the tile loop and spills 64 registers to carry them across it)
    int tid_c = threadIdx.x;
    asm volatile("" : "+v"(tid_c));
    const int mm_c = tid_c % NM, mg_c = tid_c / NM;
    if (tid_c < 64) {
        float r_v0 = 1.0f, r_v1 = 1.0f;
        if (a.first_mfcc && tid_c < NH) {
            r_v0 = ld_sc1(a.mel0 + (size_t)b * NM + tid_c);
            r_v1 = ld_sc1(a.mel0 + (size_t)b * NM + NM - 1 - tid_c);
        }
        float pmx, pmn, mmx, mmn, as;
        if (mode == 2) {
            // this block computed the records itself (its poll ran out): the same reduction, in the same order, as the
            // last arriver's
            float r_pmx = NEG_INF, r_pmn = POS_INF, r_mmx = NEG_INF, r_mmn = POS_INF, r_as = 0.0f;
            for (int t = tid_c; t < nt_b; t += 64) {
                const float* r8 = recs + (size_t)t * 8;
                r_pmx = fmaxf(r_pmx, ld_sc1(r8 + 0)); r_pmn = fminf(r_pmn, ld_sc1(r8 + 1));
                r_mmx = fmaxf(r_mmx, ld_sc1(r8 + 2)); r_mmn = fminf(r_mmn, ld_sc1(r8 + 3));
                r_as += ld_sc1(r8 + 4);
            }
            pmx = vc::wave_max(r_pmx); pmn = vc::wave_min(r_pmn);
            mmx = vc::wave_max(r_mmx); mmn = vc::wave_min(r_mmn);
            as = vc::wave_sum(r_as);
        } else {
            const float* s8 = recs + a.fstride - 32;     // the utterance's line, written by its last arriver
            pmx = ld_sc1(s8 + 0); pmn = ld_sc1(s8 + 1); mmx = ld_sc1(s8 + 2); mmn = ld_sc1(s8 + 3); as = ld_sc1(s8 + 4);
        }
        float offp = 0.0f;
        if (a.amp_norm != 1.0f) offp = 2.0f * DB10 * __log2f(a.amp_norm / (as / (float)L));
        const float offm = 2.0f * offp;
        const float pfloor = fmaxf(pow_db_clipped(pmx, offp, -100.0f) - 80.0f, -100.0f);
        const float mfloor = fmaxf(mel_db_clipped(mmx, offm, -100.0f) - 80.0f, -100.0f);
        const float pmin_c = pow_db_clipped(pmn, offp, pfloor), mmin_c = mel_db_clipped(mmn, offm, mfloor);
        const bool pn = a.p_norm != 1.0f, mn = a.m_norm != 1.0f;
        if (tid_c == 0) {
            sc[0] = offp; sc[1] = pfloor;
            sc[2] = pn ? a.p_norm : 1.0f;
            sc[8] = pn ? pmin_c : 0.0f;
            sc[3] = offm; sc[4] = mfloor;
            sc[5] = mn ? a.m_norm : 1.0f;
            sc[6] = mn ? mmin_c : 0.0f;
        }
        float c00 = 0.0f;
        if (a.first_mfcc) {
            const float d0 = mel_db_clipped(r_v0, offm, mfloor), d1 = mel_db_clipped(r_v1, offm, mfloor);
            const float s0 = d0 + d1;
            const float dc = a.dct_half[0];
            float acc4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < NH; ++j) {
                const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s0), j));
                acc4[j & 3] = fmaf(dc, sj, acc4[j & 3]);
            }
            c00 = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
        }
        if (tid_c == 0) sc[7] = c00;
    }
    // DCT basis row of this thread's coefficient
    const int ci = tid_c / 6, cf = tid_c - ci * 6;
    float drow[NH];
    if (tid_c < 240) {
#pragma unroll
        for (int j = 0; j < NH; j += 4) {
            const f4 d = *reinterpret_cast<const f4*>(a.dct_half + ci * NH + j);
            drow[j] = d[0]; drow[j + 1] = d[1]; drow[j + 2] = d[2]; drow[j + 3] = d[3];
        }
    }
    __syncthreads();                                    // constants
    FE_STAMP(7);
    const float offp = sc[0], pfloor = sc[1], pS = sc[2], pM = sc[8], offm = sc[3], mfloor = sc[4], mS = sc[5], mM = sc[6],
                c00 = sc[7];
    const int nvalid = min(GO, F - fo);
    const int nrows = min(GO, a.out_rows - fo);

    // ---------------- P_dB out (column walk over tile rows 1..14)
    if (tid_c < NB) {
        float* o = a.pow_db + (row0 + fo) * NB + tid_c;
        const float* p = Pt + PP + tid_c;
        const bool clip = a.clip != 0;
#pragma unroll
        for (int gg = 0; gg < GO; ++gg) {
            float w = pS * (pow_db_clipped(p[gg * PP], offp, pfloor) - pM);
            if (clip) w = fminf(fmaxf(w, -1.0f), 1.0f);
            if (gg < nrows) o[gg * NB] = gg < nvalid ? w : 0.0f;
        }
    }
    FE_STAMP(8);                                        // P_dB out
    // ---------------- mel power -> clipped dB, in place (each thread the elements it wrote)
    if (mg_c < 3) {
        for (int gg = mg_c; gg < G; gg += 3) Mc[gg * NM + mm_c] = mel_db_clipped(Mc[gg * NM + mm_c], offm, mfloor);
    }
    __syncthreads();
    FE_STAMP(9);                                        // mel dB + barrier
    // ---------------- M_dB out (float4 rows) and the sum / difference halves for the DCT
    {
        const bool clip = a.clip != 0;
        f4* o = reinterpret_cast<f4*>(a.mel_db + (row0 + fo) * NM);
        for (int i = tid_c; i < nrows * (NM / 4); i += NT) {
            f4 v = *reinterpret_cast<const f4*>(Mc + NM + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w = mS * (v[e] - mM);
                if (clip) w = fminf(fmaxf(w, -1.0f), 1.0f);
                v[e] = 4 * i < nvalid * NM ? w : 0.0f;
            }
            o[i] = v;
        }
        for (int i = tid_c; i < G * NH; i += NT) {
            const int gg = i / NH, j = i - gg * NH;
            const float lo = Mc[gg * NM + j], hi = Mc[gg * NM + NM - 1 - j];
            SD[gg * NM + j] = lo + hi;
            SD[gg * NM + NH + j] = lo - hi;
        }
    }
    __syncthreads();
    FE_STAMP(10);                                       // M_dB out, sum / difference, barrier
    // ---------------- DCT-II
    if (tid_c < 240) {
        const float norm = a.mfcc_norm;
        for (int gg = cf; gg < G; gg += 6) {
            const f4* sd = reinterpret_cast<const f4*>(SD + gg * NM + (ci & 1) * NH);
            float acc4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < NH / 4; ++j) {
                const f4 sv = sd[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc4[e] = fmaf(drow[4 * j + e], sv[e], acc4[e]);
            }
            float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
            if (ci == 0) acc -= c00;
            if (norm != 1.0f) acc *= norm;
            Mf[gg * NC + ci] = acc;
        }
    }
    __syncthreads();
    FE_STAMP(11);                                       // DCT + barrier
    // ---------------- [MFCC | delta] out
    {
        const bool clip = a.clip != 0;
        f4* o = reinterpret_cast<f4*>(a.mfcc + (row0 + fo) * mw);
        const int per_row = mw / 4;
        for (int i = tid_c; i < nrows * per_row; i += NT) {
            const int gg = i / per_row, c = 4 * (i - gg * per_row);
            const int f = fo + gg;
            f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (f < F) {
                if (c < NC) {
                    v = *reinterpret_cast<const f4*>(Mf + (gg + 1) * NC + c);
                } else if (f >= 1 && f <= F - 2) {
                    const f4 nx = *reinterpret_cast<const f4*>(Mf + (gg + 2) * NC + (c - NC));
                    const f4 pv = *reinterpret_cast<const f4*>(Mf + gg * NC + (c - NC));
                    v = 2.0f * (nx - pv);
                }
                if (clip) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], -1.0f), 1.0f);
                }
            }
            o[i] = v;
        }
    }
    FE_STAMP(12);
}

}  // namespace

int vc_fe400_fused_count_bytes(int batch) { return batch * FCOUNT_PITCH * 4; }
int vc_fe400_fused_stride(int max_frames) { return ((((max_frames + GO - 1) / GO) * 8 + 31) & ~31) + 32; }       // + the utterance's summary line
// (utterances up to ~36 s; longer ones would keep early tiles waiting for tiles many rounds of workgroups away)
bool vc_fe400_fused_ok(int max_frames) { return (max_frames + GO - 1) / GO <= 512; }

int vc_fe400_launch(const Fe400Args& a, int batch, int stage_mask, int fused, hipStream_t st) {
    const size_t lds = (size_t)LDS_FLOATS * 4;
    if (fused) {
        const int ntf = (a.max_frames + GO - 1) / GO;
        Fe400Args f = a;
        const int sl = vc::opt(vc::OPT_FE_FUSED_SPIN);          // tests: 0 = every waiting block takes the no-wait path
        f.spin_limit = sl >= 0 ? sl : (int)FUSED_SPIN_LIMIT;
        hipLaunchKernelGGL(fe400_fused_kernel, dim3(ntf, batch), dim3(NT), lds, st, f);
        VC_HIP_CHECK(hipGetLastError());
        return VC_OK;
    }
    if (stage_mask & 2)
        hipLaunchKernelGGL(fe400_kernel<true>, dim3(a.nt1, batch), dim3(NT), lds, st, a);
    if (stage_mask & 4) {
        const int nt2 = (a.out_rows + GO - 1) / GO;    // tiles past the stored rows have nothing to write
        hipLaunchKernelGGL(fe400_kernel<false>, dim3(nt2, batch), dim3(NT), lds, st, a);
    }
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}
