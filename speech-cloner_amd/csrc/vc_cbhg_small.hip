// The encoder's whole pre-recurrence chain in ONE launch (bf16, the shipped encoder shape):
//   features -> prenet (dense+relu, dense+relu)                /root/reference/modules.py:274-295
//            -> conv1d_banks k = 1..K, batch norm, relu         modules.py:144-166
//            -> max_pooling1d(2, 1, 'same')                     modules.py:331
//            -> conv1d k=3 + bn + relu, conv1d k=3 + bn, + prenet output   modules.py:334-340
//            -> highwaynet x L                                  modules.py:297-319, 342-345
//            -> x-halves of the bidirectional GRU's cell matmuls (float32)  modules.py:346, 168-204
// as called from encoder.py:101-107.  At 40 channels these layers are 0.7 % of the path's FLOPs but
// were eight launches of 8-60 us each (0.17 ms of a 2.2 ms step): tiles of 128 x 128 waste most of a
// 40-wide product and every layer made an HBM round trip.  Here a block carries ~100 frames of one
// window through all of them; HBM sees the features once and the GRU's projected input once.
//
// Decomposition (256 threads = 4 waves, one block per CU at 64 windows x 4 tiles):
//  * row r of the block's tile <-> frame t0 - HL + r of the window (HL = 2 + (K-1)/2 rows of left halo),
//    128 rows = 4 MFMA frame tiles; every stage computes all rows, halo rows feed the next stage's taps.
//  * Weights are the FIRST operand of v_mfma_f32_32x32x16_bf16 everywhere (result: lane = frame, a
//    register quad = 4 consecutive channels), pre-packed in fragment order (vc_mfma_pack) and loaded
//    straight from L2 -- 1 KB per fragment, coalesced.
//  * Row-local layers (prenet, highway, GRU projection) are chained IN REGISTERS: a result tile is
//    re-used as the next layer's second operand without leaving the lane, by giving the next layer's
//    weights the matching K order ("chained" packing: K slot (step s, half h, element e) holds channel
//    32 (s>>1) + 8 (2 (s&1) + (e>>2)) + 4 h + (e&3)).  Wave w owns frame tile w for these.
//  * The convolutions read their operand from a dense LDS tile (row pitch = 40 channels exactly), so
//    the im2col row of a frame -- taps x channels -- is CONTIGUOUS in LDS and a width-k filter is one
//    GEMM over K = 40 k (zero-padded weights to a multiple of 16; the over-read hits the next row x 0).
//  * Bank + first projection: wave w owns bank channels [32w, 32w+32) of every filter width.  Its
//    slice of the bank output goes through a wave-private LDS tile (norm + relu applied, frames outside
//    the window zeroed), comes back max-pooled (v_pk_max_u16 of rows r, r+1: post-ReLU bf16 orders like
//    u16) at row shifts -1, 0, +1 and feeds the k = 3 projection's partial sum over those 32 channels --
//    so the 768-channel bank output never exists, the 4 partial sums are added once through LDS in a
//    fixed order (deterministic), and the heavy weights are streamed once per block.
// TF's SAME padding: prenet / projection outputs of frames outside the window are stored as zeros;
// pooled frame -1 is forced to zero on the fragment (max(0, bank[0]) would leak into it).
#include "vc_common.h"
#include <cstdlib>

namespace {

// Timing-only ablations (wrong results) exist in -DVC_ABLATE builds alone (tools/build_ablate.sh).
#ifdef VC_ABLATE
#define ABL(mask) ((a.dbg & (mask)) != 0)
#else
#define ABL(mask) false
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int CS_C = 40;                 // CBHG width (embed_size / 2)
constexpr int CS_PITCH = 2 * CS_C;       // bytes per LDS row: dense, see above
constexpr int CS_GF = 4, CS_GB = 8;      // guard rows in front of / behind the 128 tile rows (zero)
constexpr int cs_tile_bytes(int mi) { return (CS_GF + 32 * mi + CS_GB) * CS_PITCH; }
constexpr int CS_RED_PITCH = 176;        // 40 float32 + 16: conflict-free b128 rows
constexpr int cs_red_bytes(int mi) { return 32 * mi * CS_RED_PITCH; }
constexpr int CS_MAX_HW = 4;

struct CbhgSmallArgs {
    const void* X; int32_t x_f32, ldx;
    int32_t n_windows, T, TF, tiles_per_win, n_hw;
    const bf16x8 *pk_d1, *pk_d2, *pk_bank, *pk_p1, *pk_p2, *pk_x;
    const bf16x8* pk_hw[CS_MAX_HW];
    const float* coef;             // CO_TOTAL floats, layout CO_* below
    float* P; int32_t ldp;
    int32_t dbg;                   // -DVC_ABLATE builds only (option ablate_cbhg_front) (timing only, wrong results): 1 no bank/proj1 phases, 2 no proj1 part, 4 no bank MFMAs, 8 no tail
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// second operand of the next (chained-K) layer, k-step s, from result tiles v[tile][16] held as bf16
template <int NT>
__device__ __forceinline__ bf16x8 chain(const bf16x4 (&v)[NT][4], int s) {
    const bf16x4 lo = v[s >> 1][2 * (s & 1)], hi = v[s >> 1][2 * (s & 1) + 1];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
    return o;
}
__device__ __forceinline__ bf16x8 max_nonneg(bf16x8 a, bf16x8 b) {
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(u16x8, a), __builtin_bit_cast(u16x8, b)));
}
// same-wave LDS hand-off (a wave's LDS instructions execute in order; this only stops the compiler)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int cbhg_small_lds_tiles(int mi) { return 2 * cs_tile_bytes(mi) + (cs_red_bytes(mi) > cs_tile_bytes(mi) ? 4 * cs_red_bytes(mi) : 4 * cs_tile_bytes(mi)); }
constexpr int bank_ksteps(int k) { return (k * CS_C + 15) / 16; }
constexpr int bank_frag_offset(int k) {          // fragments (of 64 lanes) before filter width k
    int o = 0;
    for (int j = 1; j < k; ++j) o += 4 * bank_ksteps(j);
    return o;
}

// float32 coefficient vectors staged in LDS (offsets in floats)
constexpr int CO_B1 = 0, CO_B2 = 96, CO_BS = 160, CO_BB = CO_BS + 1024, CO_P1S = CO_BB + 1024, CO_P1B = CO_P1S + 64,
              CO_P2S = CO_P1B + 64, CO_P2B = CO_P2S + 64, CO_BX = CO_P2B + 64, CO_HW = CO_BX + 256,
              CO_TOTAL = CO_HW + 128 * CS_MAX_HW;

template <int N>
__device__ __forceinline__ void load_frags(bf16x8 (&dst)[N], const bf16x8* src, int n) {   // src already + lane
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < n) dst[i] = src[i * 64];
}

template <int KB, int CF, int U1, int MI>
__global__ void __launch_bounds__(256, MI <= 2 ? 2 : 1)
cbhg_small_kernel(CbhgSmallArgs a) {
    static_assert(CF % 16 == 0 && U1 % 16 == 0 && KB >= 1 && KB <= 8, "shape");
    constexpr int C = CS_C, HL = 2 + (KB - 1) / 2;
    constexpr int KS1 = CF / 16, NT1 = (U1 + 31) / 32, KS2 = U1 / 16, NTC = (C + 31) / 32;   // NTC = 2
    constexpr int KSC = (C + 15) / 16;                                                     // 3 chained k-steps over C
    constexpr int KSP2 = (3 * C + 15) / 16;                                                // 8
    constexpr int NKS_P1 = KB * 4 * 3 * 2;
    constexpr int NP = 6 * C, NTP = (NP + 31) / 32;
    constexpr int MAXB = bank_ksteps(KB);
    constexpr int CS_TILE_BYTES = cs_tile_bytes(MI), CS_RED_BYTES = cs_red_bytes(MI);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Pt = smem + CS_GF * CS_PITCH;                            // row 0 of the prenet-output tile
    char* const Qt = smem + CS_TILE_BYTES + CS_GF * CS_PITCH;            // first projection's output tile
    char* const Bw0 = smem + 2 * CS_TILE_BYTES;                          // 4 wave-private bank tiles | partial sums
    float* const co = reinterpret_cast<float*>(smem + cbhg_small_lds_tiles(MI));
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int win = blockIdx.x / a.tiles_per_win, tile = blockIdx.x - win * a.tiles_per_win;
    const int t0 = tile * a.TF, T = a.T;
    const size_t g0 = (size_t)win * T;
    const bool roww = w < MI;                         // waves that own a frame tile in the row-local stages
    const int rw = 32 * (roww ? w : 0) + li;          // this lane's row there
    const int tw = t0 - HL + rw;
    const bool inw = tw >= 0 && tw < T;

    // ---- everything the prenet needs from memory, issued up front: features, both weight sets
    bf16x8 xb[KS1];
    if (roww) {
        const size_t xrow = (g0 + (size_t)min(max(tw, 0), T - 1)) * a.ldx;
        if (a.x_f32) {
            const float* xr = static_cast<const float*>(a.X) + xrow + 8 * lh;
            f32x4 lo[KS1], hi[KS1];
#pragma unroll
            for (int s = 0; s < KS1; ++s) { lo[s] = *reinterpret_cast<const f32x4*>(xr + 16 * s); hi[s] = *reinterpret_cast<const f32x4*>(xr + 16 * s + 4); }
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) { xb[s][e] = (__bf16)lo[s][e]; xb[s][4 + e] = (__bf16)hi[s][e]; }
        } else {
            const __bf16* xr = static_cast<const __bf16*>(a.X) + xrow + 8 * lh;
#pragma unroll
            for (int s = 0; s < KS1; ++s) xb[s] = *reinterpret_cast<const bf16x8*>(xr + 16 * s);
        }
    }
    bf16x8 w1f[NT1 * KS1], w2f[NTC * KS2];
    if (roww) {
        load_frags(w1f, a.pk_d1 + lane, NT1 * KS1);
        load_frags(w2f, a.pk_d2 + lane, NTC * KS2);
    }
    // ---- coefficient vectors -> LDS; zero every activation tile (guards and over-read rows must be finite zeros)
    {
        constexpr int NV = (CO_TOTAL / 4 + 255) / 256;
        f32x4 cv[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) cv[u] = reinterpret_cast<const f32x4*>(a.coef)[min(tid + 256 * u, CO_TOTAL / 4 - 1)];
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (tid + 256 * u < CO_TOTAL / 4) reinterpret_cast<f32x4*>(co)[tid + 256 * u] = cv[u];
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = tid * 16; i < 6 * CS_TILE_BYTES; i += 256 * 16) *reinterpret_cast<f32x4*>(smem + i) = z;
    }
    // first filter's weights: in flight across the barriers below
    bf16x8 wnext[MAXB];
    load_frags(wnext, a.pk_bank + (size_t)(w * bank_ksteps(1)) * 64 + lane, bank_ksteps(1));
    __syncthreads();                                   // coefficients staged, zero fill done

    // =================================================================== prenet (wave = frame tile w)
    bf16x4 pres[NTC][4];                              // prenet output of this lane's frame (residual, modules.py:340)
    if (roww) {
        bf16x4 y1[NT1][4];
#pragma unroll
        for (int tl = 0; tl < NT1; ++tl) {
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < KS1; ++s) acc = mfma(w1f[tl * KS1 + s], xb[s], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(co + CO_B1 + 32 * tl + 8 * q + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) y1[tl][q][e] = (__bf16)fmaxf(acc[4 * q + e] + bb[e], 0.0f);
            }
        }
#pragma unroll
        for (int tl = 0; tl < NTC; ++tl) {
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < KS2; ++s) acc = mfma(w2f[tl * KS2 + s], chain<NT1>(y1, s), acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(co + CO_B2 + 32 * tl + 8 * q + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) pres[tl][q][e] = (__bf16)(inw ? fmaxf(acc[4 * q + e] + bb[e], 0.0f) : 0.0f);
            }
        }
    }
#pragma unroll
    for (int tl = 0; tl < NTC; ++tl)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 32 * tl + 8 * q + 4 * lh;
            if (roww && c0 < C) *reinterpret_cast<bf16x4*>(Pt + rw * CS_PITCH + c0 * 2) = pres[tl][q];
        }
    __syncthreads();

    // =================================================================== banks + pool + first projection
    // wave w: bank channels [32w, 32w+32) of every width; partial sums of conv1d_1 over those channels.
    // hipcc does not move loads across the fences / barriers below, so weight fragments are requested
    // one phase ahead in source order: L2 latency (~1-2 us) is longer than a phase's matrix work.
    f32x16 acc1[NTC][MI];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc1[nt][mi] = zero16();
    char* const Bw = Bw0 + w * CS_TILE_BYTES + CS_GF * CS_PITCH;
    bool inr[MI], m1[MI][3];                            // frame of row 32 mi + li inside the window; pooled frame == -1 at tap
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int t = t0 - HL + 32 * mi + li;
        inr[mi] = t >= 0 && t < T;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) m1[mi][tap] = (t + tap == 0);
    }
    bf16x8 wt2[NTC * KSP2];                           // second projection's weights (requested in the last phase)
#pragma unroll
    for (int k = 1; k <= KB; ++k) {
        if (ABL(1)) break;
        const int nks = bank_ksteps(k), pad_l = (k - 1) / 2;
        bf16x8 wcur[MAXB];
#pragma unroll
        for (int s = 0; s < MAXB; ++s) wcur[s] = wnext[s];
        // first projection's weights of this (width, channel slice)
        const bf16x8* wp = a.pk_p1 + (size_t)(((k - 1) * 4 + w) * 6) * 64 + lane;
        bf16x8 wq[3][2][NTC];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) wq[tap][s][nt] = wp[(size_t)(nt * NKS_P1 + tap * 2 + s) * 64];
        f32x16 bacc[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) bacc[mi] = zero16();
        const char* prow = Pt + (li - pad_l) * CS_PITCH + lh * 16;
        if (!ABL(4))
#pragma unroll
        for (int s = 0; s < nks; ++s) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                bacc[mi] = mfma(wcur[s], *reinterpret_cast<const bf16x8*>(prow + mi * 32 * CS_PITCH + s * 32), bacc[mi]);
        }
        if (k < KB) load_frags(wnext, a.pk_bank + (size_t)(bank_frag_offset(k + 1) + w * bank_ksteps(k + 1)) * 64 + lane, bank_ksteps(k + 1));
        else if (roww) load_frags(wt2, a.pk_p2 + lane, NTC * KSP2);
        // norm + relu, frames outside the window -> 0, into the wave's tile
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = (k - 1) * 128 + 32 * w + 8 * q + 4 * lh;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(co + CO_BS + ch), sh = *reinterpret_cast<const f32x4*>(co + CO_BB + ch);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)(inr[mi] ? fmaxf(fmaf(bacc[mi][4 * q + e], sc[e], sh[e]), 0.0f) : 0.0f);
                *reinterpret_cast<bf16x4*>(Bw + (32 * mi + li) * CS_PITCH + (8 * q + 4 * lh) * 2) = o;
            }
        }
        wave_lds_fence();
        const bf16x8 zero8 = {};
        if (!ABL(2))
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const char* br = Bw + (32 * mi + li - 1) * CS_PITCH + s * 32 + lh * 16;
                bf16x8 f[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) f[d] = *reinterpret_cast<const bf16x8*>(br + d * CS_PITCH);
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    bf16x8 pl = max_nonneg(f[tap], f[tap + 1]);
                    pl = m1[mi][tap] ? zero8 : pl;
#pragma unroll
                    for (int nt = 0; nt < NTC; ++nt) acc1[nt][mi] = mfma(wq[tap][s][nt], pl, acc1[nt][mi]);
                }
            }
        wave_lds_fence();
    }

    // ---- add the four channel-slice partial sums (fixed order), norm + relu -> Qt
    __syncthreads();                                   // every wave is done with its bank tile (aliased below)
    {
        char* red = Bw0 + w * CS_RED_BYTES;
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * nt + 8 * q + 4 * lh;
                if (c0 < C) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc1[nt][mi][4 * q + e];
                        *reinterpret_cast<f32x4*>(red + (32 * mi + li) * CS_RED_PITCH + c0 * 4) = v;
                    }
                }
            }
    }
    // weights of the tail (first highway layer or, without one, nothing; GRU projection): long in flight
    bf16x8 wh[2 * NTC * KSC], wx[NTP * KSC];
    if (roww) {
        load_frags(wh, (a.n_hw > 0 ? a.pk_hw[0] : a.pk_x) + lane, 2 * NTC * KSC);
        load_frags(wx, a.pk_x + lane, NTP * KSC);
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 32 * nt + 8 * q + 4 * lh;
            if (roww && c0 < C) {
                f32x4 v = *reinterpret_cast<const f32x4*>(Bw0 + rw * CS_RED_PITCH + c0 * 4);
#pragma unroll
                for (int p = 1; p < 4; ++p) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(Bw0 + p * CS_RED_BYTES + rw * CS_RED_PITCH + c0 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += u[e];
                }
                const f32x4 sc = *reinterpret_cast<const f32x4*>(co + CO_P1S + c0), sh = *reinterpret_cast<const f32x4*>(co + CO_P1B + c0);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)(inw ? fmaxf(fmaf(v[e], sc[e], sh[e]), 0.0f) : 0.0f);
                *reinterpret_cast<bf16x4*>(Qt + rw * CS_PITCH + c0 * 2) = o;
            }
        }
    __syncthreads();

    if (!roww || ABL(8)) return;                  // no barrier below this line
    // =================================================================== second projection + residual
    bf16x4 ev[NTC][4];
    {
        f32x16 acc[NTC];
#pragma unroll
        for (int tl = 0; tl < NTC; ++tl) acc[tl] = zero16();
        const char* qrow = Qt + (rw - 1) * CS_PITCH + lh * 16;
#pragma unroll
        for (int s = 0; s < KSP2; ++s) {
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qrow + s * 32);
#pragma unroll
            for (int tl = 0; tl < NTC; ++tl) acc[tl] = mfma(wt2[tl * KSP2 + s], qf, acc[tl]);
        }
#pragma unroll
        for (int tl = 0; tl < NTC; ++tl)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * tl + 8 * q + 4 * lh;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(co + CO_P2S + c0), sh = *reinterpret_cast<const f32x4*>(co + CO_P2B + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // the unfused path stores conv1d_2's normalised output + residual as ONE bf16 rounding
                    const float v = fmaf(acc[tl][4 * q + e], sc[e], sh[e]) + (float)pres[tl][q][e];
                    ev[tl][q][e] = (__bf16)(c0 < C ? v : 0.0f);
                }
            }
    }

    // =================================================================== highway layers (in registers)
    for (int l = 0; l < a.n_hw; ++l) {
        bf16x8 wc[2 * NTC * KSC];
#pragma unroll
        for (int i = 0; i < 2 * NTC * KSC; ++i) wc[i] = wh[i];
        if (l + 1 < a.n_hw) load_frags(wh, a.pk_hw[l + 1] + lane, 2 * NTC * KSC);
        const float* hb = co + CO_HW + 128 * l;
        bf16x8 xk[KSC];
#pragma unroll
        for (int s = 0; s < KSC; ++s) xk[s] = chain<NTC>(ev, s);
#pragma unroll
        for (int tl = 0; tl < NTC; ++tl) {
            f32x16 aH = zero16(), aT = zero16();
#pragma unroll
            for (int s = 0; s < KSC; ++s) {
                aH = mfma(wc[(2 * tl) * KSC + s], xk[s], aH);
                aT = mfma(wc[(2 * tl + 1) * KSC + s], xk[s], aT);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bH = *reinterpret_cast<const f32x4*>(hb + 64 * tl + 8 * q + 4 * lh);
                const f32x4 bT = *reinterpret_cast<const f32x4*>(hb + 64 * tl + 32 + 8 * q + 4 * lh);
                const int c0 = 32 * tl + 8 * q + 4 * lh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = vc::highway_gate(aH[4 * q + e] + bH[e], aT[4 * q + e] + bT[e], (float)ev[tl][q][e]);
                    ev[tl][q][e] = (__bf16)(c0 < C ? g : 0.0f);
                }
            }
        }
    }

    // =================================================================== GRU input projection -> global
    {
        bf16x8 xk[KSC];
#pragma unroll
        for (int s = 0; s < KSC; ++s) xk[s] = chain<NTC>(ev, s);
        const bool out_row = rw >= HL && rw < HL + a.TF && tw < T;
        float* prow = a.P + (g0 + (size_t)min(max(tw, 0), T - 1)) * a.ldp;
#pragma unroll
        for (int tl = 0; tl < NTP; ++tl) {
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < KSC; ++s) acc = mfma(wx[tl * KSC + s], xk[s], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 32 * tl + 8 * q + 4 * lh;
                if (n < NP) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(co + CO_BX + n);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[4 * q + e] + bb[e];
                    if (out_row) *reinterpret_cast<f32x4*>(prow + n) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// prenet (modules.py:274-295) of the decoder stages as one launch: relu(relu(x W1 + b1) W2 + b2), bf16.
// As two dense launches the 512- (256-) wide intermediate made an HBM round trip and each short-K GEMM paid
// its per-block prologue / epilogue (27 + 68 us per step for 13 GFLOP).  Here a wave carries 32 frames
// through both layers in registers (layer 2's weights in the chained K order, see above) and only the
// output tile goes through a wave-private LDS tile for whole-row stores.  Every weight fragment is used for
// ONE MFMA per wave, so the launch is bound by the L2 -> CU weight stream (0.34 MB per wave), not by MFMA.
struct PrenetArgs {
    const void* X; int32_t x_f32, M, ldx;
    const bf16x8 *pk1, *pk2;
    const float *b1, *b2;
    __bf16* Y; int32_t ldy;
};

template <int CINP, int U1, int U2>
__global__ void __launch_bounds__(256, U1 <= 256 ? 2 : 1)
prenet_chain_kernel(PrenetArgs a) {
    static_assert(CINP % 16 == 0 && U1 % 32 == 0 && U2 % 32 == 0, "shape");
    constexpr int KS1 = CINP / 16, NT1 = U1 / 32, KS2 = U1 / 16, NT2 = U2 / 32;
    constexpr int RING = 8, PITCH = U2 * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128 + 32 * w;
    const int row = min(m0 + li, a.M - 1);
    char* const tile = smem + w * 32 * PITCH;

    bf16x8 xb[KS1];
    if (a.x_f32) {                                     // float32 features (y_mel of the previous stage): converted on load
        const float* xr = static_cast<const float*>(a.X) + (size_t)row * a.ldx + 8 * lh;
        f32x4 lo[KS1], hi[KS1];
#pragma unroll
        for (int s = 0; s < KS1; ++s) { lo[s] = *reinterpret_cast<const f32x4*>(xr + 16 * s); hi[s] = *reinterpret_cast<const f32x4*>(xr + 16 * s + 4); }
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) { xb[s][e] = (__bf16)lo[s][e]; xb[s][4 + e] = (__bf16)hi[s][e]; }
    } else {
        const __bf16* xr = static_cast<const __bf16*>(a.X) + (size_t)row * a.ldx + 8 * lh;
#pragma unroll
        for (int s = 0; s < KS1; ++s) xb[s] = *reinterpret_cast<const bf16x8*>(xr + 16 * s);
    }
    // ---- layer 1: fragments stream through a RING-deep register ring, flat index f = tile * KS1 + k-step
    bf16x4 y1[NT1][4];
    {
        const bf16x8* p1 = a.pk1 + lane;
        bf16x8 ring[RING];
#pragma unroll
        for (int f = 0; f < RING; ++f) ring[f] = p1[(f < NT1 * KS1 ? f : NT1 * KS1 - 1) * 64];
#pragma unroll
        for (int tl = 0; tl < NT1; ++tl) {
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const int f = tl * KS1 + s;
                const bf16x8 wf = ring[f % RING];
                if (f + RING < NT1 * KS1) ring[f % RING] = p1[(f + RING) * 64];
                acc = mfma(wf, xb[s], acc);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b1 + 32 * tl + 8 * q + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) y1[tl][q][e] = (__bf16)fmaxf(acc[4 * q + e] + bb[e], 0.0f);
            }
        }
    }
    // ---- layer 2: one output tile per (rolled) iteration, K fully unrolled (the operand is a register array)
    {
        const bf16x8* p2 = a.pk2 + lane;
        bf16x8 ring[RING];
#pragma unroll
        for (int f = 0; f < RING; ++f) ring[f] = p2[f * 64];
#pragma unroll 1
        for (int tl = 0; tl < NT2; ++tl) {
            f32x16 acc = zero16();
            const bf16x8* pt = p2 + (size_t)tl * KS2 * 64;
            const bool more = tl + 1 < NT2;
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const bf16x8 wf = ring[s % RING];
                // refill from this tile's stream, then from the next tile's first steps (clamped at the very end)
                const int nf = s + RING;
                ring[s % RING] = pt[(nf < KS2 || more ? nf : KS2 - 1) * 64];
                acc = mfma(wf, chain<NT1>(y1, s), acc);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b2 + 32 * tl + 8 * q + 4 * lh);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[4 * q + e] + bb[e], 0.0f);
                *reinterpret_cast<bf16x4*>(tile + li * PITCH + (32 * tl + 8 * q + 4 * lh) * 2) = o;
            }
        }
    }
    wave_lds_fence();
    // ---- the wave's 32 x U2 tile -> global, 16 bytes per lane, whole rows
    constexpr int CPR = U2 / 8;                        // 16-byte chunks per row
    for (int idx = lane; idx < 32 * CPR; idx += 64) {
        const int r = idx / CPR, c = idx - r * CPR;
        if (m0 + r < a.M)
            *reinterpret_cast<bf16x8*>(a.Y + (size_t)(m0 + r) * a.ldy + c * 8) = *reinterpret_cast<const bf16x8*>(tile + r * PITCH + c * 16);
    }
}

// The same launch with the weight stream SHARED by the block's four waves: 16-fragment chunks (16 KB) of the two packed
// matrices, one after the other, arrive by LDS-direct loads (each wave requests 4 fragments of a chunk, two chunks ahead
// of the one being multiplied, three buffers) and every wave reads all 16 from LDS.  L2 -> CU traffic per 128 frames
// drops from 4 x 0.34 MB to 0.34 MB; the kernel above is bound by exactly that stream (42 us for 4 us of MFMA work).
// Biases sit in LDS so that the only vector-memory operations in flight inside the loops are the chunk loads: the
// counted wait "all but my newest 4" then means "chunk c has landed, c + 1 may still fly".
__device__ __forceinline__ void pn_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 16, 0, 0);
}

template <int CINP, int U1, int U2>
__global__ void __launch_bounds__(256, U1 <= 256 ? 2 : 1)
prenet_chain_lds_kernel(PrenetArgs a) {
    static_assert(CINP % 16 == 0 && U1 % 32 == 0 && U2 % 32 == 0, "shape");
    constexpr int KS1 = CINP / 16, NT1 = U1 / 32, KS2 = U1 / 16, NT2 = U2 / 32;
    constexpr int PITCH = U2 * 2 + 16, CH = 16, NF1 = NT1 * KS1, NF2 = NT2 * KS2, C1 = NF1 / CH, NC = C1 + NF2 / CH;
    static_assert(NF1 % CH == 0 && NF2 % CH == 0, "whole chunks");
    constexpr int O_W = 4 * 32 * PITCH, O_B = O_W + 3 * CH * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128 + 32 * w;
    const int row = min(m0 + li, a.M - 1);
    char* const tile = smem + w * 32 * PITCH;
    char* const wbuf = smem + O_W;
    float* const bias = reinterpret_cast<float*>(smem + O_B);          // b1[U1] | b2[U2]

    // chunk c of the fragment stream -> buffer c % 3: this wave's 4 fragments
    auto request = [&](int c) {
        const bf16x8* src = (c < C1 ? a.pk1 + (size_t)c * CH * 64 : a.pk2 + (size_t)(c - C1) * CH * 64) + lane;
        char* dst = wbuf + (c % 3) * CH * 1024 + w * 4096;
#pragma unroll
        for (int u = 0; u < 4; ++u) pn_glds16(src + (w * 4 + u) * 64, dst + u * 1024);
    };

    bf16x8 xb[KS1];
    if (a.x_f32) {                                     // float32 features (y_mel of the previous stage): converted on load
        const float* xr = static_cast<const float*>(a.X) + (size_t)row * a.ldx + 8 * lh;
        f32x4 lo[KS1], hi[KS1];
#pragma unroll
        for (int s = 0; s < KS1; ++s) { lo[s] = *reinterpret_cast<const f32x4*>(xr + 16 * s); hi[s] = *reinterpret_cast<const f32x4*>(xr + 16 * s + 4); }
#pragma unroll
        for (int s = 0; s < KS1; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) { xb[s][e] = (__bf16)lo[s][e]; xb[s][4 + e] = (__bf16)hi[s][e]; }
    } else {
        const __bf16* xr = static_cast<const __bf16*>(a.X) + (size_t)row * a.ldx + 8 * lh;
#pragma unroll
        for (int s = 0; s < KS1; ++s) xb[s] = *reinterpret_cast<const bf16x8*>(xr + 16 * s);
    }
    for (int i = tid; i < U1 + U2; i += 256) bias[i] = i < U1 ? a.b1[i] : a.b2[i - U1];
    // (xb and the biases are consumed -- waited for -- before the first chunk request: nothing else stays in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    request(0);
    if (NC > 1) request(1);

    // publish chunk c: own share landed (chunk c + 1 may still be in flight), everyone's share visible, and the
    // buffer of chunk c - 1 -- which every wave has finished reading -- is requested again for chunk c + 2
    auto publish = [&](int c) {
        if (c + 1 < NC) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 2 < NC) request(c + 2);
    };
    auto frag = [&](int f) -> bf16x8 {                 // fragment f of the whole stream (compile-time f)
        return *reinterpret_cast<const bf16x8*>(wbuf + ((f / CH) % 3) * CH * 1024 + (f % CH) * 1024 + lane * 16);
    };

    bf16x4 y1[NT1][4];
#pragma unroll
    for (int tl = 0; tl < NT1; ++tl) {
        f32x16 acc = zero16();
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const int f = tl * KS1 + s;
            if (f % CH == 0) publish(f / CH);
            acc = mfma(frag(f), xb[s], acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + 32 * tl + 8 * q + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) y1[tl][q][e] = (__bf16)fmaxf(acc[4 * q + e] + bb[e], 0.0f);
        }
    }
#pragma unroll
    for (int tl = 0; tl < NT2; ++tl) {
        f32x16 acc = zero16();
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            const int f = NF1 + tl * KS2 + s;
            if (f % CH == 0) publish(f / CH);
            acc = mfma(frag(f), chain<NT1>(y1, s), acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + U1 + 32 * tl + 8 * q + 4 * lh);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[4 * q + e] + bb[e], 0.0f);
            *reinterpret_cast<bf16x4*>(tile + li * PITCH + (32 * tl + 8 * q + 4 * lh) * 2) = o;
        }
    }
    wave_lds_fence();
    // ---- the wave's 32 x U2 tile -> global, 16 bytes per lane, whole rows
    constexpr int CPR = U2 / 8;                        // 16-byte chunks per row
    for (int idx = lane; idx < 32 * CPR; idx += 64) {
        const int r = idx / CPR, c = idx - r * CPR;
        if (m0 + r < a.M)
            *reinterpret_cast<bf16x8*>(a.Y + (size_t)(m0 + r) * a.ldy + c * 8) = *reinterpret_cast<const bf16x8*>(tile + r * PITCH + c * 16);
    }
}

template <int CINP, int U1, int U2> int launch_prenet_chain(const PrenetArgs& a, hipStream_t st) {
    constexpr int LDS = 4 * 32 * (U2 * 2 + 16);
    constexpr int LDS_SHARED = LDS + 3 * 16 * 1024 + (U1 + U2) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(prenet_chain_kernel<CINP, U1, U2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(prenet_chain_lds_kernel<CINP, U1, U2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_SHARED));
        attr_done = true;
    }
    const dim3 grid((unsigned)((a.M + 127) / 128));
    if (vc::opt(vc::OPT_PRENET_LDS) != 0)
        hipLaunchKernelGGL((prenet_chain_lds_kernel<CINP, U1, U2>), grid, dim3(256), LDS_SHARED, st, a);
    else
        hipLaunchKernelGGL((prenet_chain_kernel<CINP, U1, U2>), grid, dim3(256), LDS, st, a);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

// out[(tile * nks + s) * 64 + lane][e] = W[32 tile + (lane & 31)][kmap(16 s + 8 (lane >> 5) + e)], zero outside W
__global__ void __launch_bounds__(256)
mfma_pack_kernel(const __bf16* W, int rows, int K, int ldw, int chained, int ntiles, int nks, __bf16* out) {
    const int total = ntiles * nks * 64 * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int r = idx;
        const int e = r & 7; r >>= 3;
        const int lane = r & 63; r >>= 6;
        const int s = r % nks, tl = r / nks;
        const int row = 32 * tl + (lane & 31), h = lane >> 5;
        const int col = chained ? 32 * (s >> 1) + 8 * (2 * (s & 1) + (e >> 2)) + 4 * h + (e & 3) : 16 * s + 8 * h + e;
        out[idx] = (row < rows && col < K) ? W[(size_t)row * ldw + col] : (__bf16)0.0f;
    }
}

constexpr int cbhg_small_lds(int mi) { return cbhg_small_lds_tiles(mi) + CO_TOTAL * 4; }

template <int MI> int launch_cbhg_small(const CbhgSmallArgs& a, int n_windows, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cbhg_small_kernel<6, 80, 80, MI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, cbhg_small_lds(MI)));
        attr_done = true;
    }
    hipLaunchKernelGGL((cbhg_small_kernel<6, 80, 80, MI>), dim3((unsigned)(n_windows * a.tiles_per_win)), dim3(256),
                       cbhg_small_lds(MI), st, a);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

}  // namespace

extern "C" {

int vc_mfma_pack(const void* d_W, int32_t rows, int32_t K, int32_t ldw, int32_t chained, void* d_packed, void* stream) {
    VC_REQUIRE(d_W && d_packed && rows > 0 && K > 0 && ldw >= K, "vc_mfma_pack: bad argument");
    const int ntiles = (rows + 31) / 32, nks = (K + 15) / 16;
    hipLaunchKernelGGL(mfma_pack_kernel, dim3(64), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(d_W), rows, K, ldw, chained, ntiles, nks, static_cast<__bf16*>(d_packed));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_prenet_chain_supported(int32_t cin_padded, int32_t units1, int32_t units2) {
    return (cin_padded == 64 && units1 == 256 && units2 == 128) || (cin_padded == 80 && units1 == 512 && units2 == 256);
}

int vc_prenet_chain(const void* d_X, int32_t x_f32, int32_t M, int32_t ldx, int32_t cin_padded, int32_t units1, int32_t units2,
                    const void* d_pk1, const float* d_b1, const void* d_pk2, const float* d_b2, void* d_Y, int32_t ldy,
                    void* stream) {
    VC_REQUIRE(d_X && d_pk1 && d_b1 && d_pk2 && d_b2 && d_Y && M > 0, "vc_prenet_chain: NULL argument or M <= 0");
    VC_REQUIRE(vc_prenet_chain_supported(cin_padded, units1, units2), "vc_prenet_chain: unsupported shape %d -> %d -> %d", cin_padded, units1, units2);
    VC_REQUIRE(ldx >= cin_padded && ldx % (x_f32 ? 4 : 8) == 0 && ldy >= units2 && ldy % 8 == 0, "vc_prenet_chain: leading dimensions must keep rows 16-byte aligned and cover them");
    VC_REQUIRE(((reinterpret_cast<uintptr_t>(d_X) | reinterpret_cast<uintptr_t>(d_Y) | reinterpret_cast<uintptr_t>(d_pk1) |
                 reinterpret_cast<uintptr_t>(d_pk2) | reinterpret_cast<uintptr_t>(d_b1) | reinterpret_cast<uintptr_t>(d_b2)) & 15) == 0,
               "vc_prenet_chain: operands must be 16-byte aligned");
    PrenetArgs a;
    a.X = d_X; a.x_f32 = x_f32 ? 1 : 0; a.M = M; a.ldx = ldx;
    a.pk1 = static_cast<const bf16x8*>(d_pk1); a.pk2 = static_cast<const bf16x8*>(d_pk2);
    a.b1 = d_b1; a.b2 = d_b2; a.Y = static_cast<__bf16*>(d_Y); a.ldy = ldy;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return cin_padded == 64 ? launch_prenet_chain<64, 256, 128>(a, st) : launch_prenet_chain<80, 512, 256>(a, st);
}

int32_t vc_cbhg_front_coef_floats(void) { return CO_TOTAL; }

int vc_cbhg_front_supported(int32_t n_features, int32_t prenet_units, int32_t width, int32_t n_banks, int32_t bank_filters,
                            int32_t n_highway, int32_t gru_units, int32_t T) {
    return n_features == 80 && prenet_units == 80 && width == CS_C && n_banks == 6 && bank_filters == 128 &&
           n_highway >= 0 && n_highway <= CS_MAX_HW && gru_units == CS_C && T >= 8;
}

int vc_cbhg_front(const vc_cbhg_front_desc* d, void* stream) {
    VC_REQUIRE(d, "vc_cbhg_front: NULL descriptor");
    VC_REQUIRE(vc_cbhg_front_supported(d->n_features, d->prenet_units, d->width, d->n_banks, d->bank_filters, d->n_highway,
                                       d->gru_units, d->T),
               "vc_cbhg_front: unsupported shape (features %d, prenet %d, width %d, banks %d x %d, highway %d, gru %d)",
               d->n_features, d->prenet_units, d->width, d->n_banks, d->bank_filters, d->n_highway, d->gru_units);
    VC_REQUIRE(d->d_x && d->d_xproj && d->n_windows > 0 && d->ldx >= d->n_features && d->ldp >= 6 * d->gru_units,
               "vc_cbhg_front: NULL tensor or bad leading dimension");
    VC_REQUIRE((d->ldx * (d->x_f32 ? 4 : 2)) % 16 == 0 && d->ldp % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(d->d_x) | reinterpret_cast<uintptr_t>(d->d_xproj)) & 15) == 0,
               "vc_cbhg_front: rows of x / xproj must be 16-byte aligned");
    const void* ptrs[] = {d->d_pk_dense1, d->d_pk_dense2, d->d_pk_bank, d->d_pk_proj1, d->d_pk_proj2, d->d_pk_gru, d->d_coef};
    for (const void* p : ptrs) VC_REQUIRE(p && (reinterpret_cast<uintptr_t>(p) & 15) == 0, "vc_cbhg_front: NULL or misaligned weight pointer");
    CbhgSmallArgs a = {};
    a.X = d->d_x; a.x_f32 = d->x_f32; a.ldx = d->ldx;
    a.n_windows = d->n_windows; a.T = d->T; a.n_hw = d->n_highway;
    constexpr int HL = 2 + (6 - 1) / 2;                                // rows needed: TF + HL + K/2 + 3
    a.pk_d1 = static_cast<const bf16x8*>(d->d_pk_dense1); a.pk_d2 = static_cast<const bf16x8*>(d->d_pk_dense2);
    a.pk_bank = static_cast<const bf16x8*>(d->d_pk_bank); a.pk_p1 = static_cast<const bf16x8*>(d->d_pk_proj1);
    a.pk_p2 = static_cast<const bf16x8*>(d->d_pk_proj2); a.pk_x = static_cast<const bf16x8*>(d->d_pk_gru);
    a.coef = d->d_coef;
    for (int l = 0; l < d->n_highway; ++l) {
        VC_REQUIRE(d->d_pk_highway[l] && (reinterpret_cast<uintptr_t>(d->d_pk_highway[l]) & 15) == 0, "vc_cbhg_front: highway layer %d weights NULL or misaligned", l);
        a.pk_hw[l] = static_cast<const bf16x8*>(d->d_pk_highway[l]);
    }
    a.P = d->d_xproj; a.ldp = d->ldp;
    // frame tiles per block: 2 (64 rows, two resident blocks per CU hide each other's epilogues and LDS
    // round trips) or 4 (128 rows, half the weight traffic per frame); vc_set_option("cbhg_front_mi", 4) overrides (A/B)
    a.dbg = 0;
#ifdef VC_ABLATE
    a.dbg = vc::opt(vc::OPT_ABLATE_CBHG_FRONT) > 0 ? vc::opt(vc::OPT_ABLATE_CBHG_FRONT) : 0;
#endif
    const int mi = vc::opt(vc::OPT_CBHG_FRONT_MI) == 4 ? 4 : 2;
    const int maxtf = 32 * mi - HL - 6;
    a.tiles_per_win = (d->T + maxtf - 1) / maxtf;
    a.TF = (d->T + a.tiles_per_win - 1) / a.tiles_per_win;
    return mi == 4 ? launch_cbhg_small<4>(a, d->n_windows, static_cast<hipStream_t>(stream))
                   : launch_cbhg_small<2>(a, d->n_windows, static_cast<hipStream_t>(stream));
}

}  // extern "C"
