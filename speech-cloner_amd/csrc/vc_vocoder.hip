// Griffin-Lim vocoder for gfx950: magnitude spectrogram -> waveform.
//
// Replaces /root/reference/audio_lib.py:249-308 (griffin_lim_alg, from_power_to_wav) and :31-47
// (calc_inv_preemphasis), which loop librosa.istft / librosa.stft on the CPU.
//
// One iteration of the reference is  wav = istft(S);  S' = amp * phase(stft(wav)).  Here the state
// between iterations is the set of windowed synthesis frames  y_f[n] = w[n] * irfft(S_f)[n]
// ([frames, n_fft] float32, ping-pong), and ONE kernel per iteration does, per tile of 16 frames:
//   gather   the overlap-add of the previous frames at the samples this tile's analysis windows
//            cover, divided by the window sum-square (librosa.istft), with librosa.stft's reflect
//            padding folded into the index,
//   forward  400-point real DFT (25 x 16 split of fe_dft400.h),
//   project  S = amp * Z / |Z|  (phase of a zero bin is 0),
//   inverse  16-point inverse DFTs + hermitian 25-point inverse, synthesis window, store.
// The thread that finishes bin group k1 of the forward transform holds exactly the 16 bins
// S[k1 + 25 k2] the inverse needs, so the spectrum never leaves registers.
// The kernel is bound by LDS traffic and launch latency (a 1000-frame utterance is 63 blocks);
// HBM sees 2 x frames x n_fft x 4 bytes per iteration, all L2-resident.
#include <cmath>
#include <cstring>
#include <vector>
#include "vc_common.h"
#include "fe_dft400.h"

namespace {

constexpr int VT = 256;            // threads per block
constexpr int VG = 16;             // frames per block (400-point path)
constexpr int VGG = 4;             // frames per block (generic path)
constexpr int VA_STRIDE = 17;
constexpr float F32_TINY = 1.17549435e-38f;

struct GlArgs {
    const float* amp;        // [B][maxF][nb]
    const float* phase0;     // [B][maxF][nb]  (init kernel only)
    const float* prev;       // [B][maxF][N]
    float* next;             // [B][maxF][N]
    const int32_t* n_frames; // [B] or null
    const float* window;     // [N] | tw400[416] | twg[2N]
    int maxF, nb, N, hop, span, nov;
};

__device__ __forceinline__ int utt_frames(const GlArgs& a, int b) {
    return a.n_frames ? min(max(a.n_frames[b], 0), a.maxF) : a.maxF;
}

__device__ __forceinline__ void copy_lds(float* dst, const float* src, int n) {
    for (int i = threadIdx.x; i < n; i += VT) dst[i] = src[i];
}

// Overlap-add of the stored frames at untrimmed position q of utterance frames `fr` ([F][N]),
// normalised like librosa.istft.  win = padded window (LDS).
__device__ __forceinline__ float ola_at(const float* fr, const float* win, int q, int F, int N, int hop, int nov) {
    const int fhi = min(q / hop, F - 1);
    float acc = 0.0f, wss = 0.0f;
#pragma unroll 5
    for (int j = 0; j < nov; ++j) {
        const int f = fhi - j;
        const int o = q - f * hop;
        const bool ok = (f >= 0) && (o < N);
        const int fc = max(f, 0), oc = min(o, N - 1);
        const float v = fr[(size_t)fc * N + oc];
        const float w = win[oc];
        acc += ok ? v : 0.0f;
        wss += ok ? w * w : 0.0f;
    }
    return wss > F32_TINY ? acc / wss : acc;
}

// Samples of the reflect-padded, trimmed signal this tile needs -> xs[0..span)
__device__ __forceinline__ void gather_tile(const GlArgs& a, const float* fr, const float* win, int F, int f0, float* xs) {
    const int half = a.N / 2;
    const int L = a.hop * (F - 1);
    for (int i = threadIdx.x; i < a.span; i += VT) {
        const int p = f0 * a.hop + i;                 // position in the padded signal
        float v = 0.0f;
        if (p < L + a.N) {
            int n = p - half;
            n = n < 0 ? -n : (n >= L ? 2 * (L - 1) - n : n);
            n = min(max(n, 0), L - 1);
            v = ola_at(fr, win, n + half, F, a.N, a.hop, a.nov);
        }
        xs[i] = v;
    }
}

// 400-point iteration (INIT: spectrum = amp * exp(i phase0) instead of the analysis of prev).
template <bool INIT>
__global__ void __launch_bounds__(VT)
gl_iter400_kernel(GlArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NROW = VG * 13;
    float* win = reinterpret_cast<float*>(smem);       // [400]
    float* tw = win + 400;                             // [2][208]
    float* Are = tw + 416;                             // [NROW*17]
    float* Aim = Are + NROW * VA_STRIDE;
    // INIT keeps the magnitude tile and the phase tile in LDS; the iteration kernel holds each
    // thread's 16 target magnitudes in registers instead (38 KB of LDS: four blocks per CU)
    float* amps = Aim + NROW * VA_STRIDE;              // INIT: [VG][201]
    float* xs = INIT ? amps + VG * 201 : amps;         // [span]  (INIT: phase tile [VG][201])

    const int b = blockIdx.y, tid = threadIdx.x;
    const int F = utt_frames(a, b);
    const int f0 = blockIdx.x * VG;
    if (f0 >= F) return;
    const int nvalid = min(VG, F - f0);

    copy_lds(win, a.window, 816);
    float amr[16];
    if constexpr (INIT) {
        const float* src = a.amp + ((size_t)b * a.maxF + f0) * 201;
        for (int i = tid; i < VG * 201; i += VT) amps[i] = (i < nvalid * 201) ? src[i] : 0.0f;
        const float* ps = a.phase0 + ((size_t)b * a.maxF + f0) * 201;
        for (int i = tid; i < VG * 201; i += VT) xs[i] = (i < nvalid * 201) ? ps[i] : 0.0f;
    } else {
        const int g = min(tid / 13, VG - 1), k1 = tid - (tid / 13) * 13;
        const bool live = tid < NROW && g < nvalid;
        const float* src = a.amp + ((size_t)b * a.maxF + f0 + (live ? g : 0)) * 201;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            bool cj;
            const int bin = vcfe::src_bin(live ? k1 : 0, k2, cj);
            const float v = src[bin];
            amr[k2] = live ? v : 0.0f;
        }
    }
    __syncthreads();
    if (!INIT) {
        gather_tile(a, a.prev + (size_t)b * a.maxF * 400, win, F, f0, xs);
        __syncthreads();
        // forward step 1+2: thread (g, n2)
        const int g = tid >> 4, n2 = tid & 15;
        const float* xp = xs + g * a.hop + n2;
        float v[25], ar[13], ai[13];
#pragma unroll
        for (int n1 = 0; n1 < 25; ++n1) v[n1] = xp[16 * n1] * win[16 * n1 + n2];
        vcfe::rdft25_13(v, ar, ai);
        const int i0 = VA_STRIDE * (g * 13) + n2;
#pragma unroll
        for (int k1 = 0; k1 < 13; ++k1) {
            vcfe::cmul(ar[k1], ai[k1], tw[k1 * 16 + n2], tw[208 + k1 * 16 + n2]);
            Are[i0 + VA_STRIDE * k1] = ar[k1];
            Aim[i0 + VA_STRIDE * k1] = ai[k1];
        }
        __syncthreads();
    }
    // thread (g, k1): forward 16-point stage, projection onto the target magnitude, inverse
    // 16-point stage, conjugate twiddle -> B[k1][n2]
    if (tid < NROW) {
        const int g = tid / 13, k1 = tid - g * 13;
        float zr[16], zi[16], yr[16], yi[16];
        if (!INIT) {
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) { zr[n2] = Are[VA_STRIDE * tid + n2]; zi[n2] = Aim[VA_STRIDE * tid + n2]; }
            vcfe::cdft16(zr, zi, yr, yi);
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            bool cj;
            const int bin = vcfe::src_bin(k1, k2, cj);
            const float am = INIT ? amps[g * 201 + bin] : amr[k2];
            float ur, ui;
            if (INIT) {
                float s, c;
                sincosf(xs[g * 201 + bin], &s, &c);
                ur = c; ui = cj ? -s : s;
            } else {
                const float m2 = yr[k2] * yr[k2] + yi[k2] * yi[k2];
                const float inv = m2 > 0.0f ? rsqrtf(m2) : 0.0f;
                ur = m2 > 0.0f ? yr[k2] * inv : 1.0f;
                ui = yi[k2] * inv;
            }
            zr[k2] = am * ur;
            zi[k2] = am * ui;
        }
        vcfe::cidft16(zr, zi, yr, yi);
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            vcfe::cmul(yr[n2], yi[n2], tw[k1 * 16 + n2], -tw[208 + k1 * 16 + n2]);
            Are[VA_STRIDE * tid + n2] = yr[n2];
            Aim[VA_STRIDE * tid + n2] = yi[n2];
        }
    }
    __syncthreads();
    // thread (g, n2): hermitian 25-point inverse, synthesis window, store the frame
    {
        const int g = tid >> 4, n2 = tid & 15;
        if (g < nvalid) {
            float br[13], bi[13], x[25];
            const int i0 = VA_STRIDE * (g * 13) + n2;
#pragma unroll
            for (int k1 = 0; k1 < 13; ++k1) { br[k1] = Are[i0 + VA_STRIDE * k1]; bi[k1] = Aim[i0 + VA_STRIDE * k1]; }
            vcfe::hdft25_real(br, bi, x);
            float* out = a.next + ((size_t)b * a.maxF + f0 + g) * 400 + n2;
#pragma unroll
            for (int n1 = 0; n1 < 25; ++n1) out[16 * n1] = x[n1] * win[16 * n1 + n2] * (1.0f / 400.0f);
        }
    }
}

// Generic path (any even n_fft): direct O(N^2) DFTs from the LDS twiddle table twg[m] =
// exp(-2 pi i m / N).  VGG frames per block.
template <bool INIT>
__global__ void __launch_bounds__(VT)
gl_iter_generic_kernel(GlArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, nb = a.nb;
    float* win = reinterpret_cast<float*>(smem);       // [N]
    float* twr = win + N;                              // [N]
    float* twi = twr + N;                              // [N]
    float* Sr = twi + N;                               // [VGG][nb]
    float* Si = Sr + VGG * nb;
    float* xs = Si + VGG * nb;                         // [span]

    const int b = blockIdx.y, tid = threadIdx.x;
    const int F = utt_frames(a, b);
    const int f0 = blockIdx.x * VGG;
    if (f0 >= F) return;
    const int nvalid = min(VGG, F - f0);
    copy_lds(win, a.window, N);
    copy_lds(twr, a.window + N + 416, 2 * N);
    __syncthreads();
    if (!INIT) {
        gather_tile(a, a.prev + (size_t)b * a.maxF * N, win, F, f0, xs);
        __syncthreads();
    }
    for (int idx = tid; idx < nvalid * nb; idx += VT) {
        const int g = idx / nb, k = idx - g * nb;
        const float am = a.amp[((size_t)b * a.maxF + f0 + g) * nb + k];
        float ur, ui;
        if (INIT) {
            float s, c;
            sincosf(a.phase0[((size_t)b * a.maxF + f0 + g) * nb + k], &s, &c);
            ur = c; ui = s;
        } else {
            const float* xp = xs + g * a.hop;
            float re = 0.0f, im = 0.0f;
            int m = 0;
            for (int n = 0; n < N; ++n) {
                const float xv = xp[n] * win[n];
                re = fmaf(xv, twr[m], re);
                im = fmaf(xv, twi[m], im);
                m += k; if (m >= N) m -= N;
            }
            const float m2 = re * re + im * im;
            const float inv = m2 > 0.0f ? rsqrtf(m2) : 0.0f;
            ur = m2 > 0.0f ? re * inv : 1.0f;
            ui = im * inv;
        }
        Sr[idx] = am * ur;
        Si[idx] = am * ui;
    }
    __syncthreads();
    const float invN = 1.0f / (float)N;
    for (int idx = tid; idx < nvalid * N; idx += VT) {
        const int g = idx / N, n = idx - g * N;
        const float* sr = Sr + g * nb;
        const float* si = Si + g * nb;
        // x[n] = (1/N) [S0 + (-1)^n S_{N/2} + 2 sum_{k=1}^{N/2-1} (Sr cos(2 pi k n/N) - Si sin(2 pi k n/N))]
        float acc = 0.0f;
        int m = n;                                    // (k n) mod N for k = 1
        for (int k = 1; k < nb - 1; ++k) {
            acc = fmaf(sr[k], twr[m], acc);           // twr = cos
            acc = fmaf(si[k], twi[m], acc);           // twi = -sin
            m += n; if (m >= N) m -= N;
        }
        const float edge = sr[0] + ((n & 1) ? -sr[nb - 1] : sr[nb - 1]);
        a.next[((size_t)b * a.maxF + f0 + g) * N + n] = (edge + 2.0f * acc) * invN * win[n];
    }
}

// wav[b][n] = trimmed, normalised overlap-add of the frames; optionally accumulates
// sum (wav - prev_wav)^2 into *delta (what the reference's verbose mode prints).
__global__ void __launch_bounds__(VT)
gl_ola_kernel(const float* frames, const int32_t* n_frames, const float* window, int maxF, int N, int hop, int nov,
              float* wav, int wav_stride, const float* prev_wav, int prev_stride, float* delta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* win = reinterpret_cast<float*>(smem);
    const int b = blockIdx.y;
    const int F = n_frames ? min(max(n_frames[b], 0), maxF) : maxF;
    const int L = hop * (F - 1);
    copy_lds(win, window, N);
    __syncthreads();
    const int n = blockIdx.x * VT + threadIdx.x;
    float d2 = 0.0f;
    if (n < wav_stride) {
        float v = 0.0f;
        if (n < L) v = ola_at(frames + (size_t)b * maxF * N, win, n + N / 2, F, N, hop, nov);
        wav[(size_t)b * wav_stride + n] = v;
        if (delta && n < L) { const float d = v - prev_wav[(size_t)b * prev_stride + n]; d2 = d * d; }
    }
    if (delta) {
        d2 = vc::wave_sum(d2);
        if ((threadIdx.x & 63) == 0 && d2 != 0.0f) atomicAdd(delta + b, d2);
    }
}

// One block per utterance: y[n] = x[n] + c*y[n-1] (scipy.signal.lfilter([1],[1,-c])), then
// y *= target / mean|y|.  Each thread owns a contiguous chunk; chunk carries are combined with a
// scan over the affine maps  carry -> e_t + d_t * carry.
constexpr int PT = 1024;
__global__ void __launch_bounds__(PT)
inv_preemph_norm_kernel(float* wav, const int32_t* n_frames, int maxF, int hop, int stride, float coeff, float target) {
    __shared__ float sa[PT], sb[PT], red[PT / 64];
    const int b = blockIdx.x, t = threadIdx.x;
    const int F = n_frames ? min(max(n_frames[b], 0), maxF) : maxF;
    const int L = min(hop * (F - 1), stride);
    float* y = wav + (size_t)b * stride;
    const int C = (L + PT - 1) / PT;
    const int s0 = min(t * C, L), s1 = min(s0 + C, L);
    float asum = 0.0f;
    if (coeff != 0.0f) {
        float carry = 0.0f, d = 1.0f;
        for (int n = s0; n < s1; ++n) { carry = fmaf(coeff, carry, y[n]); y[n] = carry; d *= coeff; }
        sa[t] = d; sb[t] = carry;
        __syncthreads();
        for (int off = 1; off < PT; off <<= 1) {          // inclusive scan of (a, b): later o earlier
            float a1 = 1.0f, b1 = 0.0f;
            if (t >= off) { a1 = sa[t - off]; b1 = sb[t - off]; }
            __syncthreads();
            if (t >= off) { sb[t] = fmaf(sa[t], b1, sb[t]); sa[t] *= a1; }
            __syncthreads();
        }
        const float cin = t > 0 ? sb[t - 1] : 0.0f;
        float pw = coeff;
        for (int n = s0; n < s1; ++n) { const float v = fmaf(pw, cin, y[n]); y[n] = v; asum += fabsf(v); pw *= coeff; }
    } else {
        for (int n = s0; n < s1; ++n) asum += fabsf(y[n]);
    }
    asum = vc::wave_sum(asum);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = asum;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < PT / 64; ++i) tot += red[i];
    const float sc = (L > 0 && tot > 0.0f) ? target / (tot / (float)L) : 1.0f;
    if (target > 0.0f)
        for (int n = s0; n < s1; ++n) y[n] *= sc;
}

// One block per utterance: audio_lib.py:289-298.  P [F][nb] -> amplitude [F][nb].
__global__ void __launch_bounds__(PT)
power_to_amp_kernel(const float* P, const int32_t* n_frames, int maxF, int nb, float inv_norm, float realse, float* amp) {
    __shared__ float r0[PT / 64], r1[PT / 64];
    const int b = blockIdx.x, t = threadIdx.x;
    const int F = n_frames ? min(max(n_frames[b], 0), maxF) : maxF;
    const size_t total = (size_t)F * nb, all = (size_t)maxF * nb;
    const float* p = P + (size_t)b * all;
    float* o = amp + (size_t)b * all;
    float gain = 1.0f;
    if (realse != 1.0f) {
        float s0 = 0.0f, s1 = 0.0f;
        for (size_t i = t; i < total; i += PT) {
            const float v = fmaxf(0.0f, p[i]);
            s0 += v; s1 += powf(v, realse);
        }
        s0 = vc::wave_sum(s0); s1 = vc::wave_sum(s1);
        if ((t & 63) == 0) { r0[t >> 6] = s0; r1[t >> 6] = s1; }
        __syncthreads();
        float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
        for (int i = 0; i < PT / 64; ++i) { t0 += r0[i]; t1 += r1[i]; }
        gain = t0 / t1;                                  // p_mean / mean(P ** realse)
    }
    for (size_t i = t; i < all; i += PT) {
        float v = 0.0f;
        if (i < total) {
            v = fmaxf(0.0f, p[i]);
            if (realse != 1.0f) v = gain * powf(v, realse);
            v = exp10f(0.05f * (v * inv_norm - 80.0f));  // sqrt(db_to_power(P/norm - 80))
        }
        o[i] = v;
    }
}

}  // namespace

struct vc_vocoder_plan {
    int32_t win_length, hop, N, nb, nov, span, span_g;
    float* d_tables;       // window[N] | tw400[416] | twg[2N]
    size_t smem400, smem400_init, smem_gen;
};

extern "C" {

int vc_vocoder_plan_create(int32_t win_length, int32_t hop_length, int32_t n_fft, const double* h_window,
                           vc_vocoder_plan** out_plan) {
    VC_REQUIRE(out_plan, "out_plan is NULL");
    if (n_fft <= 0) n_fft = win_length;
    VC_REQUIRE(win_length > 0 && hop_length > 0 && n_fft >= win_length && (n_fft % 2) == 0 && n_fft <= 4096,
               "vocoder: need 0 < win_length <= n_fft <= 4096, n_fft even, hop_length > 0 (got %d, %d, %d)",
               win_length, n_fft, hop_length);
    VC_REQUIRE(hop_length <= n_fft, "vocoder: hop_length %d > n_fft %d leaves gaps", hop_length, n_fft);
    const double PI = 3.14159265358979323846;
    const int N = n_fft;
    std::vector<float> h((size_t)3 * N + 416, 0.0f);
    const int lpad = (N - win_length) / 2;
    for (int i = 0; i < win_length; ++i)
        h[lpad + i] = (float)(h_window ? h_window[i] : 0.5 - 0.5 * std::cos(2.0 * PI * i / win_length));
    for (int k1 = 0; k1 < 13; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            const double ang = -2.0 * PI * (double)(n2 * k1) / 400.0;
            h[N + k1 * 16 + n2] = (float)std::cos(ang);
            h[N + 208 + k1 * 16 + n2] = (float)std::sin(ang);
        }
    for (int m = 0; m < N; ++m) {
        const double ang = -2.0 * PI * (double)m / (double)N;
        h[N + 416 + m] = (float)std::cos(ang);
        h[N + 416 + N + m] = (float)std::sin(ang);
    }
    vc_vocoder_plan* p = new vc_vocoder_plan();
    p->win_length = win_length; p->hop = hop_length; p->N = N; p->nb = 1 + N / 2;
    p->nov = (N + hop_length - 1) / hop_length;
    p->span = (VG - 1) * hop_length + N;
    p->span_g = (VGG - 1) * hop_length + N;
    p->d_tables = nullptr;
    if (hipMalloc(&p->d_tables, h.size() * 4) != hipSuccess) {
        delete p;
        return vc::set_error(VC_ERR_HIP, "vocoder plan: hipMalloc failed");
    }
    if (hipMemcpy(p->d_tables, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->d_tables); delete p;
        return vc::set_error(VC_ERR_HIP, "vocoder plan: hipMemcpy failed");
    }
    p->smem400 = sizeof(float) * (400 + 416 + 2 * (VG * 13) * VA_STRIDE + (size_t)p->span);
    p->smem400_init = sizeof(float) * (400 + 416 + 2 * (VG * 13) * VA_STRIDE + 2 * VG * 201);
    p->smem_gen = sizeof(float) * ((size_t)3 * N + 2 * VGG * p->nb + p->span_g);
    if (N != 400 && p->smem_gen > 160 * 1024) {
        (void)hipFree(p->d_tables); delete p;
        return vc::set_error(VC_ERR_INVALID, "vocoder: n_fft %d needs %zu bytes of LDS", N, p->smem_gen);
    }
    *out_plan = p;
    return VC_OK;
}

void vc_vocoder_plan_destroy(vc_vocoder_plan* plan) {
    if (!plan) return;
    if (plan->d_tables) (void)hipFree(plan->d_tables);
    delete plan;
}

int32_t vc_vocoder_num_samples(const vc_vocoder_plan* plan, int32_t n_frames) {
    return plan && n_frames > 0 ? plan->hop * (n_frames - 1) : 0;
}

size_t vc_vocoder_workspace_bytes(const vc_vocoder_plan* plan, int32_t batch, int32_t max_frames, int32_t trace) {
    if (!plan || batch <= 0 || max_frames <= 0) return 0;
    size_t frames = (size_t)batch * max_frames * plan->N * sizeof(float);
    frames = (frames + 255) & ~(size_t)255;
    size_t wav = trace ? (((size_t)batch * plan->hop * (max_frames - 1) * sizeof(float) + 255) & ~(size_t)255) : 0;
    return 2 * frames + 2 * wav + 256;
}

int vc_power_to_amp(const float* d_P, const int32_t* d_n_frames, int32_t batch, int32_t max_frames, int32_t n_bins,
                    float P_dB_norm_factor, float realse, float* d_amp, void* stream) {
    VC_REQUIRE(d_P && d_amp && batch > 0 && max_frames > 0 && n_bins > 0, "vc_power_to_amp: bad arguments");
    VC_REQUIRE(P_dB_norm_factor != 0.0f, "vc_power_to_amp: P_dB_norm_factor is 0");
    hipLaunchKernelGGL(power_to_amp_kernel, dim3(batch), dim3(PT), 0, (hipStream_t)stream, d_P, d_n_frames, max_frames,
                       n_bins, 1.0f / P_dB_norm_factor, realse, d_amp);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_griffin_lim_f32(const vc_vocoder_plan* p, const float* d_amp, const float* d_phase0, const int32_t* d_n_frames,
                       int32_t batch, int32_t max_frames, int32_t num_iters, float* d_wav, int32_t wav_stride,
                       float* d_trace, void* d_workspace, size_t workspace_bytes, void* stream) {
    VC_REQUIRE(p && d_amp && d_phase0 && d_wav && d_workspace, "vc_griffin_lim_f32: NULL argument");
    VC_REQUIRE(batch > 0 && max_frames >= 2 && num_iters >= 1, "vc_griffin_lim_f32: need batch > 0, frames >= 2, num_iters >= 1");
    VC_REQUIRE(p->hop * (max_frames - 1) > p->N / 2, "vc_griffin_lim_f32: %d frames are shorter than the reflect padding", max_frames);
    VC_REQUIRE(wav_stride >= p->hop * (max_frames - 1), "vc_griffin_lim_f32: wav_stride %d < %d samples", wav_stride,
               p->hop * (max_frames - 1));
    VC_REQUIRE(workspace_bytes >= vc_vocoder_workspace_bytes(p, batch, max_frames, d_trace != nullptr),
               "vc_griffin_lim_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    size_t fbytes = ((size_t)batch * max_frames * p->N * sizeof(float) + 255) & ~(size_t)255;
    float* fr[2] = {reinterpret_cast<float*>(d_workspace), reinterpret_cast<float*>((char*)d_workspace + fbytes)};
    const int L = p->hop * (max_frames - 1);
    const size_t wbytes = ((size_t)batch * L * sizeof(float) + 255) & ~(size_t)255;
    float* wavs[2] = {reinterpret_cast<float*>((char*)d_workspace + 2 * fbytes),
                      reinterpret_cast<float*>((char*)d_workspace + 2 * fbytes + wbytes)};
    const bool fast = (p->N == 400);
    GlArgs a;
    a.amp = d_amp; a.phase0 = d_phase0; a.n_frames = d_n_frames; a.window = p->d_tables;
    a.maxF = max_frames; a.nb = p->nb; a.N = p->N; a.hop = p->hop; a.nov = p->nov;
    a.span = fast ? p->span : p->span_g;
    const dim3 grid(((unsigned)max_frames + (fast ? VG : VGG) - 1) / (fast ? VG : VGG), (unsigned)batch);
    const size_t smem = fast ? p->smem400 : p->smem_gen;
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter_generic_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter_generic_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter400_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gl_iter400_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const dim3 ogrid(((unsigned)wav_stride + VT - 1) / VT, (unsigned)batch);
    const dim3 sgrid(((unsigned)L + VT - 1) / VT, (unsigned)batch);
    const size_t osmem = (size_t)p->N * sizeof(float);
    const bool trace = d_trace != nullptr;
    if (trace) VC_HIP_CHECK(hipMemsetAsync(d_trace, 0, sizeof(float) * (size_t)num_iters * batch, st));
    // iteration i leaves the frames of waveform i in fr[cur]; trace mode also materialises every
    // intermediate waveform (scratch, stride L) to accumulate sum (wav_i - wav_{i-1})^2.
    int cur = 0;
    for (int i = 0; i < num_iters; ++i) {
        a.prev = fr[cur]; a.next = fr[cur ^ 1];
        if (i == 0) {
            if (fast) hipLaunchKernelGGL(gl_iter400_kernel<true>, grid, dim3(VT), p->smem400_init, st, a);
            else hipLaunchKernelGGL(gl_iter_generic_kernel<true>, grid, dim3(VT), smem, st, a);
        } else {
            if (fast) hipLaunchKernelGGL(gl_iter400_kernel<false>, grid, dim3(VT), smem, st, a);
            else hipLaunchKernelGGL(gl_iter_generic_kernel<false>, grid, dim3(VT), smem, st, a);
        }
        cur ^= 1;
        if (trace && i < num_iters - 1)
            hipLaunchKernelGGL(gl_ola_kernel, sgrid, dim3(VT), osmem, st, fr[cur], d_n_frames, p->d_tables, max_frames,
                               p->N, p->hop, p->nov, wavs[i & 1], L, (const float*)(i > 0 ? wavs[(i - 1) & 1] : nullptr), L,
                               i > 0 ? d_trace + (size_t)i * batch : (float*)nullptr);
    }
    const bool last_delta = trace && num_iters > 1;
    hipLaunchKernelGGL(gl_ola_kernel, ogrid, dim3(VT), osmem, st, fr[cur], d_n_frames, p->d_tables, max_frames, p->N,
                       p->hop, p->nov, d_wav, wav_stride, (const float*)(last_delta ? wavs[(num_iters - 2) & 1] : nullptr), L,
                       last_delta ? d_trace + (size_t)(num_iters - 1) * batch : (float*)nullptr);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_inv_preemphasis_normalize(const vc_vocoder_plan* p, float* d_wav, const int32_t* d_n_frames, int32_t batch,
                                 int32_t max_frames, int32_t wav_stride, float coeff, float mean_abs_amp_norm, void* stream) {
    VC_REQUIRE(p && d_wav && batch > 0 && max_frames >= 1 && wav_stride >= 0, "vc_inv_preemphasis_normalize: bad arguments");
    hipLaunchKernelGGL(inv_preemph_norm_kernel, dim3(batch), dim3(PT), 0, (hipStream_t)stream, d_wav, d_n_frames, max_frames,
                       p->hop, wav_stride, coeff, mean_abs_amp_norm);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

}  // extern "C"
