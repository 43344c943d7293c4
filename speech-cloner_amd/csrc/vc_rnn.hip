// Recurrent + row-wise kernels of the network path (contract: include/vc_hip.h):
//   vc_gru_bidir       bidirectional GRU recurrence (/root/reference/modules.py:168-204)
//   vc_softmax_argmax  tf.nn.softmax + tf.argmax (/root/reference/encoder.py:110-111)
//   vc_convert         f32 <-> bf16
//
// GRU: the time loop is strictly serial (400 steps per direction), so one workgroup owns one
// (window, direction) pair for the whole sequence: h never leaves the CU (LDS), the recurrent
// weights stay on chip whenever they fit (LDS here; registers in gru_resident below) and only the
// hoisted input projections stream in (prefetched one step ahead).  No cross-CU exchange per
// step -- that costs >= 1 us on this chip (MI355X_MICROARCH.md, hand-off price list), more than a
// whole step.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "vc_common.h"

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ld_w(const float* p) { return *p; }
__device__ __forceinline__ float ld_w(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }
// bf16 recurrences: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of libm tanhf / IEEE division --
// the gate arithmetic of 16 sequences lands on one CU in the MFMA kernel, so it must be cheap.
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f); }

struct GruArgs {
    const float* xproj;     // [n_seq*T, 6H]
    const void* Wh[2];      // [H, 3H] per direction
    void* out;              // [n_seq*T, 2H]
    int32_t n_seq, T, H, out_bf16, w_in_lds;
};

template <typename T> __device__ __forceinline__ void st_out(void* o, size_t i, float v, int bf) {
    if (bf) reinterpret_cast<__bf16*>(o)[i] = (__bf16)v;
    else reinterpret_cast<float*>(o)[i] = v;
}

// Generic recurrence: any H.  Thread (col, ks): column `col` of the phase's weight block, K-slice
// ks of KS (KS a power of two <= 64, lanes of one column adjacent => shuffle reduction).
template <typename WT>
__global__ void __launch_bounds__(512)
gru_generic_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, H3 = 3 * H, NT = blockDim.x, tid = threadIdx.x;
    float* h = reinterpret_cast<float*>(smem);      // [H]
    float* rh = h + H;                              // [H]
    float* u = rh + H;                              // [H]
    WT* wl = reinterpret_cast<WT*>(u + H);          // [H][3H] when cached
    const int seq = blockIdx.x, dir = blockIdx.y;
    const WT* Wg = reinterpret_cast<const WT*>(a.Wh[dir]);
    if (a.w_in_lds) {
        for (int i = tid; i < H * H3; i += NT) wl[i] = Wg[i];
        Wg = wl;
    }
    for (int i = tid; i < H; i += NT) h[i] = 0.0f;

    int KS1 = 1, KS2 = 1;
    while (KS1 * 2 <= 64 && KS1 * 2 * 2 * H <= NT) KS1 *= 2;
    while (KS2 * 2 <= 64 && KS2 * 2 * H <= NT) KS2 *= 2;
    const int col1 = tid / KS1, ks1 = tid % KS1;      // gate column in [0, 2H)
    const int col2 = tid / KS2, ks2 = tid % KS2;      // candidate column in [0, H)
    const bool act1 = col1 < 2 * H, act2 = col2 < H;
    const size_t xrow = 6 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3;
    __syncthreads();

    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    float xg = (act1 && ks1 == 0) ? xbase[(size_t)t * xrow + col1] : 0.0f;
    float xc = (act2 && ks2 == 0) ? xbase[(size_t)t * xrow + 2 * H + col2] : 0.0f;
    for (int step = 0; step < a.T; ++step, t += dt) {
        // prefetch next step's input projections
        float xg_n = 0.0f, xc_n = 0.0f;
        if (step + 1 < a.T) {
            if (act1 && ks1 == 0) xg_n = xbase[(size_t)(t + dt) * xrow + col1];
            if (act2 && ks2 == 0) xc_n = xbase[(size_t)(t + dt) * xrow + 2 * H + col2];
        }
        // phase 1: gates
        float acc = 0.0f;
        if (act1) {
            const WT* w = Wg + col1;
            for (int k = ks1; k < H; k += KS1) acc = fmaf(h[k], ld_w(w + (size_t)k * H3), acc);
        }
        for (int o = KS1 >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (act1 && ks1 == 0) {
            const float g = sigmoidf_(acc + xg);
            if (col1 < H) rh[col1] = g * h[col1];       // r first (GRUCell split order)
            else u[col1 - H] = g;
        }
        __syncthreads();
        // phase 2: candidate
        float acc2 = 0.0f;
        if (act2) {
            const WT* w = Wg + 2 * H + col2;
            for (int k = ks2; k < H; k += KS2) acc2 = fmaf(rh[k], ld_w(w + (size_t)k * H3), acc2);
        }
        for (int o = KS2 >> 1; o > 0; o >>= 1) acc2 += __shfl_xor(acc2, o, 64);
        float hn = 0.0f;
        if (act2 && ks2 == 0) {
            const float c = tanhf(acc2 + xc);
            const float uu = u[col2];
            hn = uu * h[col2] + (1.0f - uu) * c;
        }
        __syncthreads();                                // everyone done reading h / rh
        if (act2 && ks2 == 0) {
            h[col2] = hn;
            st_out<WT>(a.out, ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + col2, hn, a.out_bf16);
        }
        __syncthreads();
        xg = xg_n;
        xc = xc_n;
    }
}


// ------------------------------------------------------------------------------------------
// LSTM recurrence (modules.py:207-243 -> tf.contrib.rnn.LSTMCell, no peepholes, no projection, forget_bias 1.0):
//   z = [x, h] W + b;  i, j, f, o = split(z, 4);  c' = sigmoid(f + 1) c + sigmoid(i) tanh(j);  h' = sigmoid(o) tanh(c')
// xproj [n_seq*T, 8H] float32 holds x W_x + b of both directions (fw | bw, 4H each); Wh[dir] is the recurrent half
// [H, 4H].  One workgroup per (sequence, direction); a thread owns gate columns tid, tid + NT, ...  No shipped
// configuration enables use_lstm: a plain, any-H kernel (weights from LDS when they fit, else L2), not a tuned one.
struct LstmArgs {
    const float* xproj;
    const void* Wh[2];
    int32_t n_seq, T, H;
    void* out;
    int32_t out_bf16, w_in_lds;
};

template <typename WT>
__global__ void __launch_bounds__(512)
lstm_generic_kernel(LstmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, H4 = 4 * H, NT = blockDim.x, tid = threadIdx.x;
    float* h = reinterpret_cast<float*>(smem);      // [H]
    float* z = h + H;                               // [4H]
    WT* wl = reinterpret_cast<WT*>(z + H4);         // [H][4H] when cached
    const int seq = blockIdx.x, dir = blockIdx.y;
    const WT* W = reinterpret_cast<const WT*>(a.Wh[dir]);
    if (a.w_in_lds) {
        for (int i = tid; i < H * H4; i += NT) wl[i] = W[i];
        W = wl;
    }
    for (int i = tid; i < H; i += NT) h[i] = 0.0f;
    float c = 0.0f;                                 // cell state of unit tid (threads tid < H; H <= NT asserted by the host)
    const size_t xrow = 8 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H4;
    __syncthreads();
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    for (int step = 0; step < a.T; ++step, t += dt) {
        const float* xr = xbase + (size_t)t * xrow;
        for (int col = tid; col < H4; col += NT) {
            float acc = xr[col];
            const WT* w = W + col;
#pragma unroll 4
            for (int k = 0; k < H; ++k) acc = fmaf(h[k], ld_w(w + (size_t)k * H4), acc);
            z[col] = acc;
        }
        __syncthreads();
        if (tid < H) {
            const float gi = sigmoidf_(z[tid]), gj = tanhf(z[H + tid]);
            const float gf = sigmoidf_(z[2 * H + tid] + 1.0f), go = sigmoidf_(z[3 * H + tid]);
            c = gf * c + gi * gj;
            const float hn = go * tanhf(c);
            h[tid] = hn;
            st_out<WT>(a.out, ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + tid, hn, a.out_bf16);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Register-resident recurrence (H = 128 or 256): NT threads hold the WHOLE recurrent matrix
// [H, 3H] in VGPRs for all T steps (bf16 at H = 256: 3H^2*2 B = 393 KB of the CU's 512 KB register
// file = 192 VGPRs per lane at NT = 512, two waves per SIMD; H = 128: NT = 1024, 24 (bf16) or 48
// (f32) VGPRs per lane).  Per step only h (LDS, bf16
// copy for v_dot2c_f32_bf16 + f32 copy for the gate arithmetic) and one prefetched row of the
// hoisted input projections move.
//   phase 1: thread (col1 = tid / KS1, ks1 = tid % KS1) owns K-slice ks1 of gate column col1 (2H
//            columns), KS1 = NT / 2H adjacent lanes are summed with DPP-style shuffles;
//   phase 2: thread (col2 = tid / KS2, ks2) the same for the H candidate columns, KS2 = NT / H.
// Two barriers per step; h_old of a column lives in the owning thread's register.
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Weights arrive PRE-PACKED by gru_pack_kernel in the exact register order:
//   packed[dir][chunk c][thread tid][EPC elements]  (one 16-byte chunk per lane => coalesced)
// chunk c of thread tid covers elements j = c*EPC .. c*EPC+EPC-1 of its concatenated slices
// [phase-1 slice (KL1) | phase-2 slice (KL2)], so the preload is (KL1+KL2)/EPC direct 16-B loads.
template <typename WT> struct Res;
template <> struct Res<__bf16> {
    static constexpr int EPC = 8;                            // elements per 16-byte chunk
    typedef __bf16 hstore_t;
    typedef bf16x8v chunk_t;
    static __device__ __forceinline__ void dot(const chunk_t& w, const __bf16* hs, float& a0, float& a1) {
        const bf16x8v hv = *reinterpret_cast<const bf16x8v*>(hs);
        const bf16x2 w0 = {w[0], w[1]}, w1 = {w[2], w[3]}, w2 = {w[4], w[5]}, w3 = {w[6], w[7]};
        const bf16x2 h0 = {hv[0], hv[1]}, h1 = {hv[2], hv[3]}, h2 = {hv[4], hv[5]}, h3 = {hv[6], hv[7]};
        a0 = __builtin_amdgcn_fdot2_f32_bf16(w0, h0, a0, false);
        a1 = __builtin_amdgcn_fdot2_f32_bf16(w1, h1, a1, false);
        a0 = __builtin_amdgcn_fdot2_f32_bf16(w2, h2, a0, false);
        a1 = __builtin_amdgcn_fdot2_f32_bf16(w3, h3, a1, false);
    }
};
template <> struct Res<float> {
    static constexpr int EPC = 4;
    typedef float hstore_t;
    typedef f32x4v chunk_t;
    static __device__ __forceinline__ void dot(const chunk_t& w, const float* hs, float& a0, float& a1) {
        const f32x4v hv = *reinterpret_cast<const f32x4v*>(hs);
        a0 = fmaf(w[0], hv[0], a0);
        a1 = fmaf(w[1], hv[1], a1);
        a0 = fmaf(w[2], hv[2], a0);
        a1 = fmaf(w[3], hv[3], a1);
    }
};

template <int H, int NT> struct ResGeom {
    static constexpr int KS1 = NT / (2 * H), KL1 = H / KS1, KS2 = NT / H, KL2 = H / KS2;
};

// [H, 3H] row-major -> packed register order (see above); one launch handles both directions.
template <int H, typename WT, int NT>
__global__ void __launch_bounds__(256)
gru_pack_kernel(const WT* W0, const WT* W1, WT* packed) {
    typedef ResGeom<H, NT> G;
    constexpr int EPC = Res<WT>::EPC, PER = G::KL1 + G::KL2, H3 = 3 * H;
    const int total = 2 * NT * PER;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int dir = idx / (NT * PER);
        int r = idx - dir * NT * PER;
        const int c = r / (NT * EPC);
        r -= c * NT * EPC;
        const int tid = r / EPC, e = r - tid * EPC;
        const int j = c * EPC + e;
        const WT* W = dir ? W1 : W0;
        WT v;
        if (j < G::KL1) {
            const int col1 = tid / G::KS1, ks1 = tid % G::KS1;
            v = W[(size_t)(ks1 * G::KL1 + j) * H3 + col1];
        } else {
            const int col2 = tid / G::KS2, ks2 = tid % G::KS2;
            v = W[(size_t)(ks2 * G::KL2 + (j - G::KL1)) * H3 + 2 * H + col2];
        }
        packed[idx] = v;
    }
}

template <int H, typename WT, int NT>
__global__ void __launch_bounds__(NT, NT / 256)
gru_resident_kernel(GruArgs a, const WT* packed) {
    typedef Res<WT> R;
    typedef ResGeom<H, NT> G;
    typedef typename R::hstore_t hs_t;
    typedef typename R::chunk_t chunk_t;
    constexpr int H3 = 3 * H, EPC = R::EPC;
    constexpr int KS1 = G::KS1, KL1 = G::KL1, KS2 = G::KS2, KL2 = G::KL2;
    constexpr int NC1 = KL1 / EPC, NC2 = KL2 / EPC;
    constexpr int P = EPC;                                   // 16-byte pad between K-slices
    static_assert(KS1 >= 1 && KS2 <= 64 && KL1 % EPC == 0 && KL2 % EPC == 0, "unsupported H");
    static_assert(2 * H <= NT, "one thread per gate column at least");
    __shared__ __attribute__((aligned(16))) hs_t hb[KS1 * (KL1 + P)];     // h, sliced for phase 1
    __shared__ __attribute__((aligned(16))) hs_t rhb[KS2 * (KL2 + P)];    // r*h, sliced for phase 2
    __shared__ float hf[H];                                               // h in f32 (for r*h)
    __shared__ float ul[H];                                               // update gate

    const int tid = threadIdx.x;
    const int seq = blockIdx.x, dir = blockIdx.y;
    const int col1 = tid / KS1, ks1 = tid % KS1;
    const int col2 = tid / KS2, ks2 = tid % KS2;
    chunk_t w1[NC1], w2[NC2];
    {
        const chunk_t* pk = reinterpret_cast<const chunk_t*>(packed) + (size_t)dir * NT * (NC1 + NC2) + tid;
#pragma unroll
        for (int c = 0; c < NC1; ++c) w1[c] = pk[(size_t)c * NT];
#pragma unroll
        for (int c = 0; c < NC2; ++c) w2[c] = pk[(size_t)(NC1 + c) * NT];
    }
    for (int i = tid; i < KS1 * (KL1 + P); i += NT) hb[i] = (hs_t)0.0f;
    for (int i = tid; i < KS2 * (KL2 + P); i += NT) rhb[i] = (hs_t)0.0f;
    if (tid < H) hf[tid] = 0.0f;
    const size_t xrow = 6 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    float xg = (ks1 == 0) ? xbase[(size_t)t * xrow + col1] : 0.0f;
    float xc = (ks2 == 0) ? xbase[(size_t)t * xrow + 2 * H + col2] : 0.0f;
    float hreg = 0.0f;
    const hs_t* hs1 = hb + ks1 * (KL1 + P);
    const hs_t* hs2 = rhb + ks2 * (KL2 + P);
    __syncthreads();

    for (int step = 0; step < a.T; ++step, t += dt) {
        float xg_n = 0.0f, xc_n = 0.0f;
        if (step + 1 < a.T) {
            if (ks1 == 0) xg_n = xbase[(size_t)(t + dt) * xrow + col1];
            if (ks2 == 0) xc_n = xbase[(size_t)(t + dt) * xrow + 2 * H + col2];
        }
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int c = 0; c < NC1; ++c) R::dot(w1[c], hs1 + c * EPC, a0, a1);
        float acc = a0 + a1;
#pragma unroll
        for (int o = KS1 >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (ks1 == 0) {
            const float g = (sizeof(WT) == 2) ? fast_sigmoid(acc + xg) : sigmoidf_(acc + xg);
            if (col1 < H) {
                const int s = col1 / KL2, off = col1 - s * KL2;
                rhb[s * (KL2 + P) + off] = (hs_t)(g * hf[col1]);
            } else {
                ul[col1 - H] = g;
            }
        }
        __syncthreads();
        float b0 = 0.0f, b1 = 0.0f;
#pragma unroll
        for (int c = 0; c < NC2; ++c) R::dot(w2[c], hs2 + c * EPC, b0, b1);
        float acc2 = b0 + b1;
#pragma unroll
        for (int o = KS2 >> 1; o > 0; o >>= 1) acc2 += __shfl_xor(acc2, o, 64);
        if (ks2 == 0) {
            const float c = (sizeof(WT) == 2) ? fast_tanh(acc2 + xc) : tanhf(acc2 + xc);
            const float uu = ul[col2];
            const float hn = uu * hreg + (1.0f - uu) * c;
            hreg = hn;
            const int s = col2 / KL1, off = col2 - s * KL1;
            hb[s * (KL1 + P) + off] = (hs_t)hn;
            hf[col2] = hn;
            st_out<WT>(a.out, ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + col2, hn, a.out_bf16);
        }
        __syncthreads();
        xg = xg_n;
        xc = xc_n;
    }
}

template <int H, typename WT, int NT>
int launch_resident(const GruArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = 2 * (size_t)3 * H * H * sizeof(WT);
    if (ws == nullptr || ws_bytes < need)
        return vc::set_error(VC_ERR_WORKSPACE, "vc_gru_bidir: workspace too small (%zu < %zu)", ws_bytes, need);
    WT* packed = static_cast<WT*>(ws);
    hipLaunchKernelGGL((gru_pack_kernel<H, WT, NT>), dim3(256), dim3(256), 0, st,
                       static_cast<const WT*>(a.Wh[0]), static_cast<const WT*>(a.Wh[1]), packed);
    hipLaunchKernelGGL((gru_resident_kernel<H, WT, NT>), dim3(a.n_seq, 2), dim3(NT), 0, st, a,
                       static_cast<const WT*>(packed));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}


// ------------------------------------------------------------------------------------------
// MFMA recurrence (bf16, H = 128 / 256): ONE workgroup advances 16 sequences of one direction.
//   v_mfma_f32_16x16x32_bf16:  D[unit i][seq n] += A[unit i][k] * B[k][seq n]
//   A = transposed recurrent weights (tile of 16 hidden units x 32 k), resident for all T steps:
//       gate (r,u) tiles in VGPRs, candidate tiles in VGPRs (H = 128) or LDS (H = 256: 128 KB);
//   B = the 16 hidden-state vectors (bf16, LDS, one ds_read_b128 per lane and k-step).
// Wave w owns hidden units [w*H/8, (w+1)*H/8): its r, u and c accumulators for a (unit, sequence)
// pair live in the same lane and register, so the whole gate arithmetic is lane-local; only r*h
// and h cross waves (LDS, two barriers per step).  The matrix work of a step costs 3H^2*16 MAC
// at 2048 MAC/clk/CU = 0.64 us (H = 256), independent of how many of the 16 slots are used.
typedef float f32x4m __attribute__((ext_vector_type(4)));

template <int H> struct MfGeom {
    static constexpr int NW = 8, UW = H / NW, TPW = UW / 16, KSN = H / 32;
    static constexpr int NF_G = 2 * TPW * KSN, NF_C = TPW * KSN, NF = NF_G + NF_C;   // fragments per wave
    static constexpr int PITCH = H + 8;                       // bf16 elements per LDS row of h
    static constexpr bool CAND_LDS = (H >= 256);
};

// packed[dir][wave][frag][lane][8]: frag f < NF_G: gate g = f / (TPW*KSN) (0 = r, 1 = u), else candidate
template <int H>
__global__ void __launch_bounds__(256)
gru_mfma_pack_kernel(const __bf16* W0, const __bf16* W1, __bf16* packed) {
    typedef MfGeom<H> G;
    const int total = 2 * G::NW * G::NF * 64 * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int r = idx;
        const int j = r & 7; r >>= 3;
        const int lane = r & 63; r >>= 6;
        const int f = r % G::NF; r /= G::NF;
        const int wave = r % G::NW;
        const int dir = r / G::NW;
        int g, rem;
        if (f < G::NF_G) { g = f / (G::TPW * G::KSN); rem = f % (G::TPW * G::KSN); }
        else { g = 2; rem = f - G::NF_G; }
        const int tl = rem / G::KSN, ks = rem % G::KSN;
        const int k = ks * 32 + 8 * (lane >> 4) + j;
        const int col = g * H + wave * G::UW + tl * 16 + (lane & 15);
        const __bf16* W = dir ? W1 : W0;
        packed[idx] = W[(size_t)k * 3 * H + col];
    }
}

// -DVC_ABLATE builds only: per-phase cycle sums of the MFMA recurrence (wave 0 of workgroup (0, 0); s_memtime around
// the phases of every step), read back with vc_ablate_read_stamps.  The shipped library contains none of this.
#ifdef VC_ABLATE
__device__ unsigned long long g_gru_stamps[16];
#define GRU_T(i) do { if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_t[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define GRU_T(i) do { } while (0)
#endif

// The two barriers of a step order LDS traffic only (h and r*h change hands through LDS): every wave's LDS operations have
// completed (lgkmcnt), nothing else is waited for -- __syncthreads() also drains vmcnt (the prefetch of the next step's
// input projections, the output stores issued just in front of barrier B).  Measured: no difference (barrier A 663, barrier
// B 576 cycles per step either way, gpurun_out r03y): the "barrier" time of this kernel is the partner wave's matrix
// products on the shared SIMD, not a memory drain.
#define GRU_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int H>
__global__ void __launch_bounds__(512, 2)
gru_mfma_kernel(GruArgs a, const __bf16* packed) {
    typedef MfGeom<H> G;
    constexpr int TPW = G::TPW, KSN = G::KSN, PITCH = G::PITCH, UW = G::UW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* hb = reinterpret_cast<__bf16*>(smem);                 // [16][PITCH]
    __bf16* rhb = hb + 16 * PITCH;                                // [16][PITCH]
    bf16x8v* candL = reinterpret_cast<bf16x8v*>(rhb + 16 * PITCH);   // [NW][NF_C][64] when CAND_LDS

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y, seq0 = blockIdx.x * 16;
    const int n = lane & 15, q = lane >> 4;
    const int seq = min(seq0 + n, a.n_seq - 1);
    const bool seq_ok = (seq0 + n) < a.n_seq;
    const bf16x8v* pk = reinterpret_cast<const bf16x8v*>(packed) + ((size_t)(dir * G::NW + wave) * G::NF) * 64 + lane;

    bf16x8v wg[G::NF_G];
#pragma unroll
    for (int f = 0; f < G::NF_G; ++f) wg[f] = pk[(size_t)f * 64];
    bf16x8v wc[G::CAND_LDS ? 1 : G::NF_C];
    if (G::CAND_LDS) {
#pragma unroll
        for (int f = 0; f < G::NF_C; ++f) candL[(wave * G::NF_C + f) * 64 + lane] = pk[(size_t)(G::NF_G + f) * 64];
    } else {
#pragma unroll
        for (int f = 0; f < G::NF_C; ++f) wc[f] = pk[(size_t)(G::NF_G + f) * 64];
    }
    for (int i = tid; i < 2 * 16 * PITCH; i += 512) hb[i] = (__bf16)0.0f;       // hb and rhb are adjacent
    float hreg[TPW][4];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
        for (int e = 0; e < 4; ++e) hreg[tl][e] = 0.0f;

    const int H3 = 3 * H;
    const size_t xrow = 6 * (size_t)H;
    const int ucol = wave * UW + q * 4;                          // first of this lane's 4 units in tile 0
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3 + ucol;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    f32x4m xr[TPW], xu[TPW], xc[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        const float* xp = xbase + (size_t)t * xrow + tl * 16;
        xr[tl] = *reinterpret_cast<const f32x4m*>(xp);
        xu[tl] = *reinterpret_cast<const f32x4m*>(xp + H);
        xc[tl] = *reinterpret_cast<const f32x4m*>(xp + 2 * H);
    }
    __syncthreads();

    const __bf16* hrow = hb + n * PITCH + 8 * q;                 // B-fragment source of this lane
    const __bf16* rrow = rhb + n * PITCH + 8 * q;
#ifdef VC_ABLATE
    const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif
    for (int step = 0; step < a.T; ++step, t += dt) {
        const bool more = step + 1 < a.T;
        const float* xn = xbase + (size_t)(more ? t + dt : t) * xrow;
        // ---- phase 1: r and u pre-activations.  All h fragments are fetched up front so the MFMA
        // chain never waits on an LDS read it has just issued.
        bf16x8v bfr[KSN];
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8v*>(hrow + ks * 32);
        f32x4m ar[TPW], au[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            ar[tl] = xr[tl];
            au[tl] = xu[tl];
            xr[tl] = *reinterpret_cast<const f32x4m*>(xn + tl * 16);          // next step's, in place
            xu[tl] = *reinterpret_cast<const f32x4m*>(xn + tl * 16 + H);
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) {
                ar[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[(0 * TPW + tl) * KSN + ks], bfr[ks], ar[tl], 0, 0, 0);
                au[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[(1 * TPW + tl) * KSN + ks], bfr[ks], au[tl], 0, 0, 0);
            }
        }
        GRU_T(0);                                                    // h fragments read, 32 gate MFMAs issued
        float uu[TPW][4];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
            bf16x4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = fast_sigmoid(ar[tl][e]);
                uu[tl][e] = fast_sigmoid(au[tl][e]);
                o[e] = (__bf16)(r * hreg[tl][e]);
            }
            *reinterpret_cast<bf16x4v*>(rhb + n * PITCH + ucol + tl * 16) = o;
        }
        GRU_T(1);                                                    // MFMA results waited for, sigmoids, r*h stored
        GRU_LDS_BARRIER();
        GRU_T(2);                                                    // barrier A
        // ---- phase 2: candidate, state update
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8v*>(rrow + ks * 32);
        f32x4m ac[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            ac[tl] = xc[tl];
            xc[tl] = *reinterpret_cast<const f32x4m*>(xn + tl * 16 + 2 * H);
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) {
                const bf16x8v w = G::CAND_LDS ? candL[(wave * G::NF_C + tl * KSN + ks) * 64 + lane] : wc[G::CAND_LDS ? 0 : tl * KSN + ks];
                ac[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, bfr[ks], ac[tl], 0, 0, 0);
            }
        }
        GRU_T(3);                                                    // r*h fragments read, 16 candidate MFMAs issued
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
            bf16x4v o;
            f32x4m hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = fast_tanh(ac[tl][e]);
                const float hn = uu[tl][e] * hreg[tl][e] + (1.0f - uu[tl][e]) * c;
                hreg[tl][e] = hn;
                hv[e] = hn;
                o[e] = (__bf16)hn;
            }
            *reinterpret_cast<bf16x4v*>(hb + n * PITCH + ucol + tl * 16) = o;
            if (seq_ok) {
                const size_t oi = ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + ucol + tl * 16;
                if (a.out_bf16) *reinterpret_cast<bf16x4v*>(reinterpret_cast<__bf16*>(a.out) + oi) = o;
                else *reinterpret_cast<f32x4m*>(reinterpret_cast<float*>(a.out) + oi) = hv;
            }
        }
        GRU_T(4);                                                    // MFMA results waited for, tanh, update, stores issued
        GRU_LDS_BARRIER();
        GRU_T(5);                                                    // barrier B
    }
#ifdef VC_ABLATE
    if (stamp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_gru_stamps[i] = acc_t[i];
    }
#endif
}

template <int H>
int launch_mfma(const GruArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    typedef MfGeom<H> G;
    const size_t need = 2 * (size_t)3 * H * H * sizeof(__bf16);
    if (ws == nullptr || ws_bytes < need)
        return vc::set_error(VC_ERR_WORKSPACE, "vc_gru_bidir: workspace too small (%zu < %zu)", ws_bytes, need);
    __bf16* packed = static_cast<__bf16*>(ws);
    hipLaunchKernelGGL((gru_mfma_pack_kernel<H>), dim3(256), dim3(256), 0, st, static_cast<const __bf16*>(a.Wh[0]),
                       static_cast<const __bf16*>(a.Wh[1]), packed);
    const size_t lds = 2 * 16 * (size_t)G::PITCH * 2 + (G::CAND_LDS ? (size_t)G::NW * G::NF_C * 64 * 16 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_mfma_kernel<H>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((gru_mfma_kernel<H>), dim3((a.n_seq + 15) / 16, 2), dim3(512), lds, st, a,
                       static_cast<const __bf16*>(packed));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------
// MFMA recurrence, four-wave form (bf16, H = 128 / 256): the same 16 sequences per workgroup, but ONE wave per SIMD with the
// SIMD's whole register file (512 registers), so that
//   * ALL recurrent weights stay on chip in registers for the T steps except the second half of the candidate tiles'
//     k-steps at H = 256 (LDS): the r / u fragments are named as ACCUMULATOR-file operands of hand-placed matrix
//     instructions (hipcc keeps builtin MFMA operands in the 256 architectural registers and copied 214 registers per
//     step out of the accumulator half when left to itself),
//   * a wave owns 64 (H = 256) / 32 (H = 128) hidden units = 4 / 2 independent accumulator chains per gate, enough to
//     keep the matrix pipe fed from ONE gate at a time: r's products are issued first, u's behind them with r's sigmoids
//     dealt into the gaps between them (a 16x16x32 product holds the vector issue for 8 of its 16 cycles); u's sigmoids
//     sit between the candidate's products,
//   * the two barriers per step are 4-wave, LDS-only barriers (~100 cycles; in the eight-wave form each SIMD's two waves
//     serialise on the matrix pipe, which shows as ~600 cycles of "barrier" per phase: tools/gru_phase_stamps.py).
// MEASURED, NOT SHIPPED AS THE DEFAULT (vc_set_option("gru_mfma4", 1) selects it; bit-identical results): 2.79 us per
// step against the eight-wave kernel's 2.01 at H = 256, 1.07 against 1.03 at H = 128 (profiles/r03/ab_gru_mfma4.log,
// phase stamps beside it).  The step is bound by VECTOR issue, not by where the weights live: 96 transcendentals and
// ~250 other vector instructions per lane and step are ~1,900 issue cycles, of which one wave per SIMD can hide only what
// fits into the 8 free cycles beside each 16-cycle product, while two waves per SIMD overlap one wave's gate arithmetic
// with the other's products for free.  Also measured on the way: prefetching the input projections by LDS-direct loads
// into a ring costs ~200 cycles of issue PER 1-KB piece beside the products (12 pieces = 2,373 of 6,959 cycles per step).
template <int H> struct Mf4Geom {
    static constexpr int NW = 4, UW = H / NW, TPW = UW / 16, KSN = H / 32;
    static constexpr int NF_G = 2 * TPW * KSN, NF_C = TPW * KSN, NF = NF_G + NF_C;
    static constexpr int PITCH = H + 8;
    static constexpr int KL = (H >= 256) ? 4 : 0;            // k-steps of every candidate tile whose fragments sit in LDS
    static constexpr int KR = KSN - KL;
    static constexpr size_t LDS = 2 * 16 * (size_t)PITCH * 2 + (size_t)NW * TPW * KL * 64 * 16;
};

template <int H>
__global__ void __launch_bounds__(256)
gru_mfma4_pack_kernel(const __bf16* W0, const __bf16* W1, __bf16* packed) {
    typedef Mf4Geom<H> G;
    const int total = 2 * G::NW * G::NF * 64 * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int r = idx;
        const int j = r & 7; r >>= 3;
        const int lane = r & 63; r >>= 6;
        const int f = r % G::NF; r /= G::NF;
        const int wave = r % G::NW;
        const int dir = r / G::NW;
        int g, rem;
        if (f < G::NF_G) { g = f / (G::TPW * G::KSN); rem = f % (G::TPW * G::KSN); }
        else { g = 2; rem = f - G::NF_G; }
        const int tl = rem / G::KSN, ks = rem % G::KSN;
        const int k = ks * 32 + 8 * (lane >> 4) + j;
        const int col = g * H + wave * G::UW + tl * 16 + (lane & 15);
        const __bf16* W = dir ? W1 : W0;
        packed[idx] = W[(size_t)k * 3 * H + col];
    }
}

template <int H>
__global__ void __launch_bounds__(256, 1)
gru_mfma4_kernel(GruArgs a, const __bf16* packed) {
    typedef Mf4Geom<H> G;
    constexpr int TPW = G::TPW, KSN = G::KSN, PITCH = G::PITCH, UW = G::UW, KL = G::KL, KR = G::KR;
    typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* hb = reinterpret_cast<__bf16*>(smem);                 // [16][PITCH]
    __bf16* rhb = hb + 16 * PITCH;                                // [16][PITCH]
    bf16x8v* candL = reinterpret_cast<bf16x8v*>(rhb + 16 * PITCH);   // [NW][TPW][KL][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, seq0 = blockIdx.x * 16;
    const int n = lane & 15, q = lane >> 4;
    const int seq = min(seq0 + n, a.n_seq - 1);
    const bool seq_ok = (seq0 + n) < a.n_seq;
    const bf16x8v* pk = reinterpret_cast<const bf16x8v*>(packed) + ((size_t)(dir * G::NW + wave) * G::NF) * 64 + lane;

    bf16x8v wr[TPW][KSN], wu[TPW][KSN], wc[TPW][KR];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
            wr[tl][ks] = pk[(size_t)((0 * TPW + tl) * KSN + ks) * 64];
            wu[tl][ks] = pk[(size_t)((1 * TPW + tl) * KSN + ks) * 64];
            const bf16x8v c = pk[(size_t)(G::NF_G + tl * KSN + ks) * 64];
            if (ks < KR) wc[tl][ks < KR ? ks : 0] = c;
            else candL[((wave * TPW + tl) * KL + (ks - KR)) * 64 + lane] = c;
        }
    for (int i = tid; i < 2 * 16 * PITCH; i += 256) hb[i] = (__bf16)0.0f;       // hb and rhb are adjacent
    float hreg[TPW][4];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
        for (int e = 0; e < 4; ++e) hreg[tl][e] = 0.0f;

    const int H3 = 3 * H;
    const size_t xrow = 6 * (size_t)H;
    const int ucol = wave * UW + q * 4;                          // first of this lane's 4 units in tile 0
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3 + ucol;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    f32x4m xr[TPW], xu[TPW], xc[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        const float* xp = xbase + (size_t)t * xrow + tl * 16;
        xr[tl] = *reinterpret_cast<const f32x4m*>(xp);
        xu[tl] = *reinterpret_cast<const f32x4m*>(xp + H);
        xc[tl] = *reinterpret_cast<const f32x4m*>(xp + 2 * H);
    }
    __syncthreads();

    const __bf16* hrow = hb + n * PITCH + 8 * q;                 // B-fragment source of this lane
    const __bf16* rrow = rhb + n * PITCH + 8 * q;
    // LDS-only barrier: every wave's LDS writes have completed (lgkmcnt); the loads of the next step's input projections
    // and this step's output stores stay in flight across it (__syncthreads() would drain them: vmcnt(0))
#define GRU4_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // One product with the weight fragment as an accumulator-file ("a") or architectural ("v") operand.  hipcc pads
    // nothing around asm: the leading s_nop covers a compiler-made copy of an operand right in front of the statement;
    // a chain's result is first read by VALU code at least four matrix instructions later (an XDL result of this shape
    // needs about a dozen wait states before a VALU read), which the empty "+v" statements below pin.
#define GRU4_MFMA(acc, w, b, in_acc)                                                                                   \
    do {                                                                                                               \
        if (in_acc) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));   \
        else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));         \
    } while (0)
#ifdef VC_ABLATE
    const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif
    for (int step = 0; step < a.T; ++step, t += dt) {
        const float* xn = xbase + (size_t)(step + 1 < a.T ? t + dt : t) * xrow;
        // ---- phase 1: r, then u with r's sigmoids in its gaps
        bf16x8v bfr[KSN];
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8v*>(hrow + ks * 32);
        f32x4m ar[TPW], au[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            ar[tl] = xr[tl];
            au[tl] = xu[tl];
            xr[tl] = *reinterpret_cast<const f32x4m*>(xn + tl * 16);          // next step's, in place
            xu[tl] = *reinterpret_cast<const f32x4m*>(xn + tl * 16 + H);
        }
        GRU_T(0);                                                    // requests of the next step's projections issued
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) GRU4_MFMA(ar[tl], wr[tl][ks], bfr[ks], true);
        float rhf[TPW][4];
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) {
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) GRU4_MFMA(au[tl], wu[tl][ks], bfr[ks], !(H >= 256 && tl == TPW - 1 && ks >= KSN - 4));
            // r's sigmoids, dealt over u's k-steps 1 .. KSN - 1 (r's last product lies >= TPW matrix instructions back).
            // The empty statements pin each share between two groups of products: its inputs are opaque until the group
            // before it has been issued, its results are demanded before the group after it.
            constexpr int PER = (TPW * 4 + KSN - 2) / (KSN - 1);         // values per gap
            if (ks >= 1) {
#pragma unroll
                for (int v = (ks - 1) * PER; v < ks * PER && v < TPW * 4; ++v) {
                    const int tl = v >> 2, e = v & 3;
                    if (e == 0 || v == (ks - 1) * PER) asm volatile("" : "+v"(ar[tl]));
                    rhf[tl][e] = fast_sigmoid(ar[tl][e]) * hreg[tl][e];
                    asm volatile("" : "+v"(rhf[tl][e]));
                }
            }
        }
        GRU_T(1);                                                    // fragments read, r and u products issued, r's sigmoids
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            bf16x4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)rhf[tl][e];
            *reinterpret_cast<bf16x4v*>(rhb + n * PITCH + ucol + tl * 16) = o;
        }
        GRU_T(2);                                                    // r*h stored
        GRU4_BARRIER();                                              // barrier A: r*h complete
        GRU_T(3);
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) asm volatile("" : "+v"(au[tl]));      // u's readers: behind the barrier
        // ---- phase 2: candidate (u's sigmoids between its products), state update
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8v*>(rrow + ks * 32);
        f32x4m ac[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            ac[tl] = xc[tl];
            xc[tl] = *reinterpret_cast<const f32x4m*>(xn + tl * 16 + 2 * H);
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) {
                const bf16x8v w = ks < KR ? wc[tl][ks < KR ? ks : 0] : candL[((wave * TPW + tl) * KL + (ks < KR ? 0 : ks - KR)) * 64 + lane];
                ac[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, bfr[ks], ac[tl], 0, 0, 0);
            }
        float uu[TPW][4];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
            for (int e = 0; e < 4; ++e) uu[tl][e] = fast_sigmoid(au[tl][e]);
        bf16x4v ob[TPW];
        f32x4m hv[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = fast_tanh(ac[tl][e]);
                const float hn = uu[tl][e] * hreg[tl][e] + (1.0f - uu[tl][e]) * c;
                hreg[tl][e] = hn;
                hv[tl][e] = hn;
                ob[tl][e] = (__bf16)hn;
            }
            *reinterpret_cast<bf16x4v*>(hb + n * PITCH + ucol + tl * 16) = ob[tl];
        }
        GRU_T(4);                                                    // candidate products, u's sigmoids, tanh, update, h stored
        if (seq_ok) {
            if (a.out_bf16) {
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl)
                    *reinterpret_cast<bf16x4v*>(reinterpret_cast<__bf16*>(a.out) + ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + ucol + tl * 16) = ob[tl];
            } else {
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl)
                    *reinterpret_cast<f32x4m*>(reinterpret_cast<float*>(a.out) + ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + ucol + tl * 16) = hv[tl];
            }
        }
        GRU_T(5);                                                    // output stores issued
        GRU4_BARRIER();                                              // barrier B: h complete
        GRU_T(6);
    }
#undef GRU4_MFMA
#undef GRU4_BARRIER
#ifdef VC_ABLATE
    if (stamp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_gru_stamps[i] = acc_t[i];
    }
#endif
}

template <int H>
int launch_mfma4(const GruArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    typedef Mf4Geom<H> G;
    const size_t need = 2 * (size_t)3 * H * H * sizeof(__bf16);
    if (ws == nullptr || ws_bytes < need)
        return vc::set_error(VC_ERR_WORKSPACE, "vc_gru_bidir: workspace too small (%zu < %zu)", ws_bytes, need);
    __bf16* packed = static_cast<__bf16*>(ws);
    hipLaunchKernelGGL((gru_mfma4_pack_kernel<H>), dim3(256), dim3(256), 0, st, static_cast<const __bf16*>(a.Wh[0]),
                       static_cast<const __bf16*>(a.Wh[1]), packed);
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_mfma4_kernel<H>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
        attr_done = true;
    }
    hipLaunchKernelGGL((gru_mfma4_kernel<H>), dim3((a.n_seq + 15) / 16, 2), dim3(256), G::LDS, st, a,
                       static_cast<const __bf16*>(packed));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------
// MFMA recurrence for a SMALL hidden size (the encoder: H = 40; bf16): ONE WAVE advances 16 sequences of one direction.
// All 3H^2 recurrent weights (padded to 48 units x 64 k: 18 fragments = 72 registers) stay in the wave's registers, the
// 16 hidden-state vectors go through a wave-private LDS tile to change from the accumulator layout (lane = (sequence,
// 4 units)) into the B-operand layout (lane = (sequence, 8 k)) -- no workgroup barrier anywhere: a wave's LDS operations
// execute in order.  Against gru_wave_kernel (one wave per SEQUENCE, h broadcast by v_readlane, 0.94 us per step): a
// step here costs 18 products + the gate arithmetic of 16 x 40 values spread over 64 lanes, and a 64-window batch is 8
// waves on 2 CUs instead of 128 waves on 32 -- CUs that the register-filling MFMA launches of the other streams cannot
// use while a single such wave sits on them ("who blocks whom").
// MEASURED, NOT THE DEFAULT (vc_set_option("gru_small_mfma", 1) selects it): 1.40 us per step against gru_wave_kernel's 0.84
// (one wave alone on a SIMD pays every latency of the step's dependent chain in full: LDS hand-off, 18 products, 36
// transcendental pairs per lane), and the pipelined step did not gain from the freed CUs either (1.7655 vs 1.7537 ms;
// profiles/r03/ab_gru_small_mfma.log).
template <int H> struct MfsGeom {
    static constexpr int HP = (H + 15) / 16 * 16, KP = (H + 31) / 32 * 32;
    static constexpr int TPW = HP / 16, KSN = KP / 32, NF = 3 * TPW * KSN;
    static constexpr int PITCH = KP + 8;                                 // bf16 elements per LDS row of h
    static constexpr int WPB = 4;                                        // waves (groups of 16 sequences) per workgroup
    static constexpr size_t LDS = (size_t)WPB * 2 * 16 * PITCH * 2;
};

// packed[dir][frag = (g, tl, ks)][lane][8], zero outside the H x H block
template <int H>
__global__ void __launch_bounds__(256)
gru_mfma_small_pack_kernel(const __bf16* W0, const __bf16* W1, __bf16* packed) {
    typedef MfsGeom<H> G;
    const int total = 2 * G::NF * 64 * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int r = idx;
        const int j = r & 7; r >>= 3;
        const int lane = r & 63; r >>= 6;
        const int f = r % G::NF;
        const int dir = r / G::NF;
        const int g = f / (G::TPW * G::KSN), rem = f % (G::TPW * G::KSN);
        const int tl = rem / G::KSN, ks = rem % G::KSN;
        const int k = ks * 32 + 8 * (lane >> 4) + j;
        const int unit = tl * 16 + (lane & 15);
        const __bf16* W = dir ? W1 : W0;
        packed[idx] = (k < H && unit < H) ? W[(size_t)k * 3 * H + g * H + unit] : (__bf16)0.0f;
    }
}

template <int H>
__global__ void __launch_bounds__(64 * MfsGeom<H>::WPB)
gru_mfma_small_kernel(GruArgs a, const __bf16* packed) {
    typedef MfsGeom<H> G;
    constexpr int TPW = G::TPW, KSN = G::KSN, PITCH = G::PITCH;
    typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int dir = blockIdx.y, seq0 = (blockIdx.x * G::WPB + wave) * 16;
    if (seq0 >= a.n_seq) return;                                  // (no workgroup barrier below: a wave may leave)
    __bf16* hb = reinterpret_cast<__bf16*>(smem) + (size_t)wave * 2 * 16 * PITCH;     // [16][PITCH], this wave's own
    __bf16* rhb = hb + 16 * PITCH;
    const int n = lane & 15, q = lane >> 4;
    const int seq = min(seq0 + n, a.n_seq - 1);
    const bool seq_ok = (seq0 + n) < a.n_seq;
    const bf16x8v* pk = reinterpret_cast<const bf16x8v*>(packed) + (size_t)dir * G::NF * 64 + lane;
    bf16x8v w[3][TPW][KSN];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) w[g][tl][ks] = pk[(size_t)((g * TPW + tl) * KSN + ks) * 64];
    for (int i = lane; i < 2 * 16 * PITCH; i += 64) hb[i] = (__bf16)0.0f;        // hb and rhb, padding columns included
    float hreg[TPW][4];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
        for (int e = 0; e < 4; ++e) hreg[tl][e] = 0.0f;

    const int H3 = 3 * H;
    const size_t xrow = 6 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3 + q * 4;
    bool uok[TPW];                                                // this lane's 4 units of tile tl exist (H % 4 == 0)
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) uok[tl] = tl * 16 + q * 4 < H;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    const f32x4m zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4m xr[TPW], xu[TPW], xc[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        const float* xp = xbase + (size_t)t * xrow + tl * 16;
        xr[tl] = uok[tl] ? *reinterpret_cast<const f32x4m*>(xp) : zero4;
        xu[tl] = uok[tl] ? *reinterpret_cast<const f32x4m*>(xp + H) : zero4;
        xc[tl] = uok[tl] ? *reinterpret_cast<const f32x4m*>(xp + 2 * H) : zero4;
    }
    const __bf16* hrow = hb + n * PITCH + 8 * q;
    const __bf16* rrow = rhb + n * PITCH + 8 * q;
    for (int step = 0; step < a.T; ++step, t += dt) {
        const float* xn = xbase + (size_t)(step + 1 < a.T ? t + dt : t) * xrow;
        bf16x8v bfr[KSN];
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8v*>(hrow + ks * 32);
        f32x4m ar[TPW], au[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            ar[tl] = xr[tl];
            au[tl] = xu[tl];
            xr[tl] = uok[tl] ? *reinterpret_cast<const f32x4m*>(xn + tl * 16) : zero4;
            xu[tl] = uok[tl] ? *reinterpret_cast<const f32x4m*>(xn + tl * 16 + H) : zero4;
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) {
                ar[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0][tl][ks], bfr[ks], ar[tl], 0, 0, 0);
                au[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1][tl][ks], bfr[ks], au[tl], 0, 0, 0);
            }
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            bf16x4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(fast_sigmoid(ar[tl][e]) * hreg[tl][e]);
            if (uok[tl]) *reinterpret_cast<bf16x4v*>(rhb + n * PITCH + tl * 16 + q * 4) = o;
        }
        // (same wave: the LDS writes above are ordered before the reads below)
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8v*>(rrow + ks * 32);
        f32x4m ac[TPW];
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            ac[tl] = xc[tl];
            xc[tl] = uok[tl] ? *reinterpret_cast<const f32x4m*>(xn + tl * 16 + 2 * H) : zero4;
        }
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl)
                ac[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2][tl][ks], bfr[ks], ac[tl], 0, 0, 0);
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl) {
            bf16x4v o;
            f32x4m hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = fast_sigmoid(au[tl][e]);
                const float c = fast_tanh(ac[tl][e]);
                const float hn = u * hreg[tl][e] + (1.0f - u) * c;
                hreg[tl][e] = hn;
                hv[e] = hn;
                o[e] = (__bf16)hn;
            }
            if (uok[tl]) {
                *reinterpret_cast<bf16x4v*>(hb + n * PITCH + tl * 16 + q * 4) = o;
                if (seq_ok) {
                    const size_t oi = ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + tl * 16 + q * 4;
                    if (a.out_bf16) *reinterpret_cast<bf16x4v*>(reinterpret_cast<__bf16*>(a.out) + oi) = o;
                    else *reinterpret_cast<f32x4m*>(reinterpret_cast<float*>(a.out) + oi) = hv;
                }
            }
        }
    }
}

template <int H>
int launch_mfma_small(const GruArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    typedef MfsGeom<H> G;
    const size_t need = 2 * (size_t)G::NF * 64 * 16;
    if (ws == nullptr || ws_bytes < need)
        return vc::set_error(VC_ERR_WORKSPACE, "vc_gru_bidir: workspace too small (%zu < %zu)", ws_bytes, need);
    __bf16* packed = static_cast<__bf16*>(ws);
    hipLaunchKernelGGL((gru_mfma_small_pack_kernel<H>), dim3(16), dim3(256), 0, st, static_cast<const __bf16*>(a.Wh[0]),
                       static_cast<const __bf16*>(a.Wh[1]), packed);
    const int groups = (a.n_seq + 15) / 16;
    hipLaunchKernelGGL((gru_mfma_small_kernel<H>), dim3((groups + G::WPB - 1) / G::WPB, 2), dim3(64 * G::WPB), G::LDS, st, a,
                       static_cast<const __bf16*>(packed));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------
// Single-wave recurrence for small H (the encoder: H = 40): one 64-lane wave per (window,
// direction), no LDS and no barriers.  Lane j owns hidden unit j: its three weight columns
// (r_j, u_j, c_j: 3H f32 registers) and h_j.  h is broadcast to the wave one element at a time
// with v_readlane (the value becomes a scalar operand of the FMAs).
// WPB waves (= sequences) per workgroup, one per SIMD of a CU: as single-wave workgroups the 2 * n_seq
// waves were dealt to as many CUs, and a CU that holds one of them for 400 steps cannot take a
// workgroup of the register-filling MFMA kernels of the other streams (half the chip at 64 windows).
constexpr int GRU_WAVE_WPB = 4;
template <int H, typename WT>
__global__ void __launch_bounds__(64 * GRU_WAVE_WPB)
gru_wave_kernel(GruArgs a) {
    static_assert(H <= 64 && H % 4 == 0, "one lane per hidden unit, unrolled by 4");
    constexpr int H3 = 3 * H;
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * GRU_WAVE_WPB + (threadIdx.x >> 6), dir = blockIdx.y;
    if (seq >= a.n_seq) return;
    const bool act = lane < H;
    const int j = act ? lane : 0;
    const WT* W = reinterpret_cast<const WT*>(a.Wh[dir]);
    float wr[H], wu[H], wc[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        wr[k] = act ? ld_w(W + (size_t)k * H3 + j) : 0.0f;
        wu[k] = act ? ld_w(W + (size_t)k * H3 + H + j) : 0.0f;
        wc[k] = act ? ld_w(W + (size_t)k * H3 + 2 * H + j) : 0.0f;
    }
    const size_t xrow = 6 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3 + j;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    float xr = xbase[(size_t)t * xrow], xu = xbase[(size_t)t * xrow + H], xc = xbase[(size_t)t * xrow + 2 * H];
    float h = 0.0f;
    for (int step = 0; step < a.T; ++step, t += dt) {
        float xr_n = 0.0f, xu_n = 0.0f, xc_n = 0.0f;
        if (step + 1 < a.T) {
            const float* xn = xbase + (size_t)(t + dt) * xrow;
            xr_n = xn[0]; xu_n = xn[H]; xc_n = xn[2 * H];
        }
        // two partial sums per gate (four for the candidate): the FMA chains, not the issue rate,
        // bound a single wave
        float ar0 = xr, ar1 = 0.0f, au0 = xu, au1 = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k += 2) {
            const float h0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k));
            const float h1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k + 1));
            ar0 = fmaf(h0, wr[k], ar0);
            au0 = fmaf(h0, wu[k], au0);
            ar1 = fmaf(h1, wr[k + 1], ar1);
            au1 = fmaf(h1, wu[k + 1], au1);
        }
        constexpr bool FAST = sizeof(WT) == 2;            // bf16 model: v_exp/v_rcp gate functions
        const float r = FAST ? fast_sigmoid(ar0 + ar1) : sigmoidf_(ar0 + ar1);
        const float u = FAST ? fast_sigmoid(au0 + au1) : sigmoidf_(au0 + au1);
        const float rh = r * h;
        float ac0 = xc, ac1 = 0.0f, ac2 = 0.0f, ac3 = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), k));
            const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), k + 1));
            const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), k + 2));
            const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rh), k + 3));
            ac0 = fmaf(r0, wc[k], ac0);
            ac1 = fmaf(r1, wc[k + 1], ac1);
            ac2 = fmaf(r2, wc[k + 2], ac2);
            ac3 = fmaf(r3, wc[k + 3], ac3);
        }
        const float acs = (ac0 + ac1) + (ac2 + ac3);
        const float c = FAST ? fast_tanh(acs) : tanhf(acs);
        h = act ? (u * h + (1.0f - u) * c) : 0.0f;
        if (act) st_out<WT>(a.out, ((size_t)seq * a.T + t) * 2 * H + (size_t)dir * H + j, h, a.out_bf16);
        xr = xr_n; xu = xu_n; xc = xc_n;
    }
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
softmax_argmax_kernel(const float* logits, int M, int N, int ldl, void* prob, int ldp, int out_bf16, int32_t* cls,
                      __bf16* prob2, int ldp2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* x = logits + (size_t)row * ldl;
    float mx = -3.402823466e38f;
    int mi = 0x7fffffff;
    for (int c = lane; c < N; c += 64) {
        const float v = x[c];
        if (v > mx) { mx = v; mi = c; }                 // first maximum within the lane's stride
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    float s = 0.0f;
    for (int c = lane; c < N; c += 64) s += expf(x[c] - mx);
    s = vc::wave_sum(s);
    const float inv = 1.0f / s;
    const int wmax = ldp > ldp2 ? ldp : ldp2;
    for (int c = lane; c < wmax; c += 64) {
        const float p = c < N ? expf(x[c] - mx) * inv : 0.0f;
        if (c < ldp) {
            if (out_bf16) reinterpret_cast<__bf16*>(prob)[(size_t)row * ldp + c] = (__bf16)p;
            else reinterpret_cast<float*>(prob)[(size_t)row * ldp + c] = p;
        }
        if (prob2 && c < ldp2) prob2[(size_t)row * ldp2 + c] = (__bf16)p;      // second, zero-padded bf16 copy
    }
    if (cls && lane == 0) cls[row] = mi;
}

__global__ void __launch_bounds__(256)
convert_kernel(const void* src, int sdt, void* dst, int ddt, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float v = sdt == VC_F32 ? reinterpret_cast<const float*>(src)[i]
                                      : (float)reinterpret_cast<const __bf16*>(src)[i];
        if (ddt == VC_F32) reinterpret_cast<float*>(dst)[i] = v;
        else reinterpret_cast<__bf16*>(dst)[i] = (__bf16)v;
    }
}

}  // namespace

extern "C" {

size_t vc_gru_workspace_bytes(int32_t H, int32_t w_dtype) {
    if (H <= 0) return 0;
    if (H == 40 && w_dtype == VC_BF16) return 2 * (size_t)MfsGeom<40>::NF * 64 * 16;       // padded fragment order
    return 2 * (size_t)3 * H * H * (w_dtype == VC_F32 ? 4 : 2);
}

int vc_gru_bidir(const float* d_xproj, const void* d_Wh_fw, const void* d_Wh_bw, int32_t w_dtype, int32_t n_seq,
                 int32_t T, int32_t H, void* d_out, int32_t out_dtype, void* d_workspace, size_t workspace_bytes,
                 void* stream) {
    VC_REQUIRE(d_xproj && d_Wh_fw && d_Wh_bw && d_out, "NULL argument");
    VC_REQUIRE(n_seq > 0 && T > 0 && H > 0 && H <= 1024, "bad shape n_seq=%d T=%d H=%d", n_seq, T, H);
    VC_REQUIRE(w_dtype == VC_F32 || w_dtype == VC_BF16, "bad w_dtype %d", w_dtype);
    VC_REQUIRE(out_dtype == VC_F32 || out_dtype == VC_BF16, "bad out_dtype %d", out_dtype);
    GruArgs a;
    a.xproj = d_xproj; a.Wh[0] = d_Wh_fw; a.Wh[1] = d_Wh_bw; a.out = d_out;
    a.n_seq = n_seq; a.T = T; a.H = H; a.out_bf16 = out_dtype == VC_BF16;
    const size_t wbytes = (size_t)H * 3 * H * (w_dtype == VC_F32 ? 4 : 2);
    const size_t base = 3 * (size_t)H * 4;
    a.w_in_lds = (base + wbytes <= 150 * 1024);
    const size_t lds = base + (a.w_in_lds ? wbytes : 0);
    int nt = 256;
    while (nt < 512 && nt < 2 * H) nt *= 2;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(n_seq, 2);
    // register-resident kernels for the decoder's sizes
    // bf16, one sequence per workgroup, weights in VGPRs: 1.5 us/step at H = 256 on 2 * n_seq CUs.
    // bf16, 16 sequences per workgroup on MFMA: 3.3 us/step but 16x fewer CUs.  A batch that would
    // occupy a quarter of the chip or more with the first form takes the second, which leaves the
    // CUs to the MFMA-bound launches of the other streams (full path: +7.7 % frames/s at 64
    // windows); small batches keep the low-latency form.  vc_set_option("gru_mfma", 0 / 1) forces either.
    const int gm = vc::opt(vc::OPT_GRU_MFMA);
    const bool use_valu = gm >= 0 ? (gm != 1) : (n_seq < 32);
    const bool four = vc::opt(vc::OPT_GRU_MFMA4) == 1;        // 1: the four-wave form (measured slower: see its comment); default: eight waves
    if (w_dtype == VC_BF16 && H == 256) return use_valu ? launch_resident<256, __bf16, 512>(a, d_workspace, workspace_bytes, st)
                                               : four ? launch_mfma4<256>(a, d_workspace, workspace_bytes, st)
                                                      : launch_mfma<256>(a, d_workspace, workspace_bytes, st);
    if (w_dtype == VC_BF16 && H == 128) return use_valu ? launch_resident<128, __bf16, 256>(a, d_workspace, workspace_bytes, st)
                                               : four ? launch_mfma4<128>(a, d_workspace, workspace_bytes, st)
                                                      : launch_mfma<128>(a, d_workspace, workspace_bytes, st);
    // (H = 128 runs 256 threads: four fat waves beat sixteen thin ones, the step is barrier-bound)
    if (w_dtype == VC_F32 && H == 128) return launch_resident<128, float, 1024>(a, d_workspace, workspace_bytes, st);
    if (H == 40 && w_dtype == VC_BF16) {
        // the shipped encoder in bf16: 16 sequences per wave on MFMA -- only with option gru_small_mfma = 1 (see the kernel's comment)
        const int sm = vc::opt(vc::OPT_GRU_SMALL_MFMA);
        if (sm == 1) return launch_mfma_small<40>(a, d_workspace, workspace_bytes, st);      // measured slower: off unless asked for
    }
    if (H == 40) {                                      // the shipped encoder (hp/encoder_cfg_d.json)
        const dim3 gw((n_seq + GRU_WAVE_WPB - 1) / GRU_WAVE_WPB, 2);
        if (w_dtype == VC_F32) hipLaunchKernelGGL((gru_wave_kernel<40, float>), gw, dim3(64 * GRU_WAVE_WPB), 0, st, a);
        else hipLaunchKernelGGL((gru_wave_kernel<40, __bf16>), gw, dim3(64 * GRU_WAVE_WPB), 0, st, a);
        VC_HIP_CHECK(hipGetLastError());
        return VC_OK;
    }
    if (w_dtype == VC_F32) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_generic_kernel<float>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(gru_generic_kernel<float>, grid, dim3(nt), lds, st, a);
    } else {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_generic_kernel<__bf16>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(gru_generic_kernel<__bf16>, grid, dim3(nt), lds, st, a);
    }
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

#ifdef VC_ABLATE
int vc_ablate_read_gru_stamps(unsigned long long* h_out) {
    VC_HIP_CHECK(hipDeviceSynchronize());
    VC_HIP_CHECK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_gru_stamps), sizeof(unsigned long long) * 16));
    return VC_OK;
}
#endif

int vc_lstm_bidir(const float* d_xproj, const void* d_Wh_fw, const void* d_Wh_bw, int32_t w_dtype, int32_t n_seq, int32_t T,
                  int32_t H, void* d_out, int32_t out_dtype, void* stream) {
    VC_REQUIRE(d_xproj && d_Wh_fw && d_Wh_bw && d_out, "NULL argument");
    VC_REQUIRE(n_seq > 0 && T > 0 && H > 0 && H <= 512 && n_seq <= 65535, "vc_lstm_bidir: bad shape n_seq=%d T=%d H=%d (H <= 512)", n_seq, T, H);
    VC_REQUIRE((w_dtype == VC_F32 || w_dtype == VC_BF16) && (out_dtype == VC_F32 || out_dtype == VC_BF16), "bad dtype");
    LstmArgs a;
    a.xproj = d_xproj; a.Wh[0] = d_Wh_fw; a.Wh[1] = d_Wh_bw; a.n_seq = n_seq; a.T = T; a.H = H;
    a.out = d_out; a.out_bf16 = out_dtype == VC_BF16;
    const size_t base = (size_t)5 * H * sizeof(float);
    const size_t wbytes = (size_t)4 * H * H * (w_dtype == VC_F32 ? 4 : 2);
    a.w_in_lds = (base + wbytes <= 150 * 1024);
    const size_t lds = base + (a.w_in_lds ? wbytes : 0);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (w_dtype == VC_F32) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_generic_kernel<float>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lstm_generic_kernel<float>, dim3(n_seq, 2), dim3(512), lds, st, a);
    } else {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_generic_kernel<__bf16>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lstm_generic_kernel<__bf16>, dim3(n_seq, 2), dim3(512), lds, st, a);
    }
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_softmax_argmax(const float* d_logits, int32_t M, int32_t N, int32_t ldl, void* d_prob, int32_t ldp,
                      int32_t out_dtype, int32_t* d_class, void* stream) {
    VC_REQUIRE(d_logits && d_prob, "NULL argument");
    VC_REQUIRE(M > 0 && N > 0 && ldl >= N && ldp >= N, "bad shape M=%d N=%d ldl=%d ldp=%d", M, N, ldl, ldp);
    VC_REQUIRE(out_dtype == VC_F32 || out_dtype == VC_BF16, "bad out_dtype %d", out_dtype);
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_logits, M, N, ldl, d_prob, ldp, out_dtype == VC_BF16, d_class, static_cast<__bf16*>(nullptr), 0);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_softmax_argmax_dual(const float* d_logits, int32_t M, int32_t N, int32_t ldl, float* d_prob, int32_t ldp,
                           void* d_prob_bf16, int32_t ldp_bf16, int32_t* d_class, void* stream) {
    VC_REQUIRE(d_logits && d_prob && d_prob_bf16, "NULL argument");
    VC_REQUIRE(M > 0 && N > 0 && ldl >= N && ldp >= N && ldp_bf16 >= N, "bad shape M=%d N=%d ldl=%d ldp=%d/%d", M, N, ldl, ldp, ldp_bf16);
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_logits, M, N, ldl, static_cast<void*>(d_prob), ldp, 0, d_class, static_cast<__bf16*>(d_prob_bf16), ldp_bf16);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_convert(const void* d_src, int32_t src_dtype, void* d_dst, int32_t dst_dtype, size_t n, void* stream) {
    VC_REQUIRE(d_src && d_dst, "NULL argument");
    VC_REQUIRE((src_dtype == VC_F32 || src_dtype == VC_BF16) && (dst_dtype == VC_F32 || dst_dtype == VC_BF16), "bad dtype");
    if (n == 0) return VC_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(convert_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_src, src_dtype,
                       d_dst, dst_dtype, n);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

}  // extern "C"
