// Implicit-GEMM kernel for the network blocks of /root/reference/modules.py on gfx950 MFMA:
// tf.layers.dense (:291-293,315-317), tf.layers.conv1d SAME/no-bias (:104-140), conv1d_banks
// (:144-166, grouped launch), max_pooling1d(2,1,same) (:331, fused into the consumer's operand
// load), inference FusedBatchNorm (:39-102, folded scale/shift epilogue), residual add (:340)
// and highwaynet (:297-319, paired-column epilogue).  Contract: include/vc_hip.h (vc_gemm_desc).
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 2 x 2 MFMA
// tiles of 32 x 32 (a 64 x 128 block variant, wave tile 32 x 64, is used when the grid would
// otherwise leave CUs idle).  K is consumed in slabs of 128 bytes per row (32 f32 / 64 bf16) staged
// through a double-buffered LDS image with 144-byte rows (16-byte pad => the four 16-lane groups
// of ds_read_b128 hit 16 distinct slots).  Both operands are K-contiguous (A is a Toeplitz view
// of the activations, B is the pre-transposed kernel), so global loads are 16 B/lane and a lane's
// MFMA fragment is one ds_read_b128:
//   f32 : v_mfma_f32_32x32x2_f32, exact f32 fma chain; the 4 floats of a 16-B chunk feed 4 MFMAs
//   bf16: v_mfma_f32_32x32x16_bf16, f32 accumulate; a 16-B chunk is one MFMA's 8-element fragment
// Register-staged pipeline: the (branch-free, unconditional) global loads of slab t+1 are issued
// before the MFMAs of slab t and written to the other LDS buffer after them; one barrier per slab.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include "vc_common.h"
#include "vc_bank256.h"
#include "vc_conv256.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BN = 128;                        // block tile: (64*MI) x 128, MI = 1 or 2
constexpr int ROWB = 144;                      // LDS row pitch in bytes (128 data + 16 pad)
constexpr int GEMM_THREADS = 256;
constexpr int lds_bytes(int MI) { return 2 * (64 * MI + BN) * ROWB; }   // A0 B0 A1 B1

template <typename T> struct Tr;
template <> struct Tr<float> {
    static constexpr int VEC = 4, BK = 32;
    typedef f32x4 vec_t;
};
template <> struct Tr<__bf16> {
    static constexpr int VEC = 8, BK = 64;
    typedef bf16x8 vec_t;
};

struct KGroup {
    const void* Bt;
    int32_t K, taps, pad_l, c_off;
};

struct KArgs {
    const void* X;
    int32_t M, T, Cin, ldx, N, n_groups;
    const float* pro_scale;
    const float* pro_shift;
    int32_t pro_relu, pro_pool;
    const float* epi_scale;
    const float* epi_shift;
    int32_t act;
    const void* R;
    int32_t ldr;
    void* C;
    int32_t ldc, out_f32;
    float drop_keep;               // 0 = no dropout, else keep probability (tf.layers.dropout, training)
    unsigned long long drop_seed;
    int32_t sum_groups;            // != 0: the groups are partial sums of ONE output; group g reads input channels [c_off, c_off + Cin)
    KGroup g[VC_GEMM_MAX_GROUPS];
};

// Counter-based dropout mask: splitmix64 of (element index, seed); keep iff u24 < keep * 2^24.
// Stateless, so tests can reproduce the mask on the host (tests/test_training_gpu.py).
__device__ __forceinline__ bool drop_keep_elem(unsigned long long idx, unsigned long long seed, float keep) {
    unsigned long long x = idx + seed * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (float)(unsigned)(x >> 40) < keep * 16777216.0f;
}

__device__ __forceinline__ float act_fn(float v, int act) {
    switch (act) {
        case VC_ACT_RELU: return fmaxf(v, 0.0f);
        case VC_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case VC_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ void store_out(void* C, size_t idx, float v, int out_f32) {
    if (out_f32) reinterpret_cast<float*>(C)[idx] = v;
    else reinterpret_cast<T*>(C)[idx] = (T)v;
}

// per-channel affine + relu on a 16-byte operand vector
__device__ __forceinline__ f32x4 pro_apply(f32x4 v, const float* sc, const float* sh, int c, int relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (sc) x = x * sc[c + e] + sh[c + e];
        if (relu) x = fmaxf(x, 0.0f);
        v[e] = x;
    }
    return v;
}
__device__ __forceinline__ bf16x8 pro_apply(bf16x8 v, const float* sc, const float* sh, int c, int relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = (float)v[e];
        if (sc) x = x * sc[c + e] + sh[c + e];
        if (relu) x = fmaxf(x, 0.0f);
        v[e] = (__bf16)x;
    }
    return v;
}
__device__ __forceinline__ f32x4 vmax(f32x4 a, f32x4 b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], b[e]);
    return a;
}
__device__ __forceinline__ bf16x8 vmax(bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = ((float)a[e] >= (float)b[e]) ? a[e] : b[e];
    return a;
}
// max of NON-NEGATIVE values (post-ReLU): IEEE ordering == unsigned integer ordering of the bit
// patterns, so bf16 pairs go through v_pk_max_u16 and f32 through v_max_u32.
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 vmax_nonneg(bf16x8 a, bf16x8 b) {
    const u16x8 r = __builtin_elementwise_max(__builtin_bit_cast(u16x8, a), __builtin_bit_cast(u16x8, b));
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ f32x4 vmax_nonneg(f32x4 a, f32x4 b) {
    const u32x4 r = __builtin_elementwise_max(__builtin_bit_cast(u32x4, a), __builtin_bit_cast(u32x4, b));
    return __builtin_bit_cast(f32x4, r);
}

template <typename T, int MI> struct Mma;

template <int MI> struct Mma<float, MI> {
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int wm, int wn, int lane,
                                                f32x16 (&acc)[MI][2]) {
        const int i = lane & 31, h = lane >> 5;
        const char* ap = As + (wm * 32 * MI + i) * ROWB + h * 16;
        const char* bp = Bs + (wn * 64 + i) * ROWB + h * 16;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x4 av[MI], bv[2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const f32x4*>(ap + mi * 32 * ROWB + p * 32);
            bv[0] = *reinterpret_cast<const f32x4*>(bp + p * 32);
            bv[1] = *reinterpret_cast<const f32x4*>(bp + 32 * ROWB + p * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][q], bv[ni][q], acc[mi][ni], 0, 0, 0);
        }
    }
};

template <int MI> struct Mma<__bf16, MI> {
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int wm, int wn, int lane,
                                                f32x16 (&acc)[MI][2]) {
        const int i = lane & 31, h = lane >> 5;
        const char* ap = As + (wm * 32 * MI + i) * ROWB + h * 16;
        const char* bp = Bs + (wn * 64 + i) * ROWB + h * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 av[MI], bv[2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) av[mi] = *reinterpret_cast<const bf16x8*>(ap + mi * 32 * ROWB + s * 32);
            bv[0] = *reinterpret_cast<const bf16x8*>(bp + s * 32);
            bv[1] = *reinterpret_cast<const bf16x8*>(bp + 32 * ROWB + s * 32);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
        }
    }
};


// Epilogue of the 128-column tile kernels: scale/shift, activation, dropout and residual per
// accumulator element, then the tile goes through LDS so that global memory sees whole 16-byte
// chunks of a row (a lane of the 32 x 32 MFMA result owns one column: storing straight from the
// accumulators writes 2-byte elements 64 B apart per row, which costs more than the K loop of the
// short-K dense layers).  smem: the kernel's operand buffers, free after the last barrier.
// scale/shift of this lane's two output columns, fetched before the K loop (two dependent global
// loads at epilogue time cost more than the whole K loop of a short-K dense layer)
struct EpiCoef { float s[2], b[2]; };
__device__ __forceinline__ EpiCoef epilogue_coef(const KArgs& a, const KGroup& grp, int n0, int wn, int lane) {
    EpiCoef c;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int oc = grp.c_off + min(n0 + wn * 64 + ni * 32 + (lane & 31), a.N - 1);
        c.s[ni] = a.epi_scale ? a.epi_scale[oc] : 1.0f;
        c.b[ni] = a.epi_shift ? a.epi_shift[oc] : 0.0f;
    }
    return c;
}

template <typename T, int MI>
__device__ __forceinline__ void epilogue_tile(const KArgs& a, const KGroup& grp, char* smem, const f32x16 (&acc)[MI][2],
                                              const EpiCoef& coef, int m0, int n0, int row_w, int wn, int lane, int tid) {
    constexpr int BM = 64 * MI;
    const bool f32o = a.out_f32 || sizeof(T) == 4;
    const int es = f32o ? 4 : 2;
    const int pitch = 128 * es + 16;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int lcol = wn * 64 + ni * 32 + i;
        const int gn = min(n0 + lcol, a.N - 1);
        const int oc = grp.c_off + gn;
        const float s = coef.s[ni], b = coef.b[ni];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = row_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int gm = min(m0 + lrow, a.M - 1);
                float v = act_fn(acc[mi][ni][r] * s + b, a.act);
                if (a.drop_keep > 0.0f)
                    v = drop_keep_elem((unsigned long long)gm * a.ldc + oc, a.drop_seed, a.drop_keep) ? v / a.drop_keep : 0.0f;
                if (a.R) v += to_f32(reinterpret_cast<const T*>(a.R)[(size_t)gm * a.ldr + gn]);
                char* dst = smem + lrow * pitch + lcol * es;
                if (f32o) *reinterpret_cast<float*>(dst) = v;
                else *reinterpret_cast<__bf16*>(dst) = (__bf16)v;
            }
        }
    }
    __syncthreads();
    const int chunk = 16 / es;                           // elements per 16-byte chunk
    const int cpr = 128 / chunk;                          // chunks per tile row
    char* C = reinterpret_cast<char*>(a.C);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && ((a.ldc * es) % 16 == 0) &&
                        (((grp.c_off + n0) * es) % 16 == 0);
    for (int idx = tid; idx < BM * cpr; idx += GEMM_THREADS) {
        const int row = idx / cpr, ch = idx - row * cpr;
        const int gm = m0 + row, gn0 = n0 + ch * chunk;
        if (gm >= a.M || gn0 >= a.N) continue;
        const char* src = smem + row * pitch + ch * 16;
        char* dst = C + ((size_t)gm * a.ldc + grp.c_off + gn0) * es;
        if (vec_ok && gn0 + chunk <= a.N) {
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
        } else {
            const int n = min(chunk, a.N - gn0);
            for (int e = 0; e < n; ++e) {
                if (f32o) reinterpret_cast<float*>(dst)[e] = reinterpret_cast<const float*>(src)[e];
                else reinterpret_cast<unsigned short*>(dst)[e] = reinterpret_cast<const unsigned short*>(src)[e];
            }
        }
    }
}

// PRO: 0 = plain operand, 1 = time max-pool of the operand, 2 = per-channel affine (+relu)
// (+ pool) -- the training path's "normalise the previous layer on the fly".
template <typename T, int MODE, int MI, int PRO>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
gemm_kernel(KArgs a) {
    typedef typename Tr<T>::vec_t vec_t;
    constexpr int VEC = Tr<T>::VEC, BK = Tr<T>::BK;
    constexpr int BM = 64 * MI, AP = 2 * MI;                 // A rows per block, 32-row staging passes
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, BUF_BYTES = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const KGroup grp = a.g[a.n_groups - 1 - (int)blockIdx.y];   // heaviest group first
    const int ntn = (a.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int K = grp.K, Cin = a.Cin, Tn = a.T;
    const int nk = (K + BK - 1) / BK;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* Bt = reinterpret_cast<const T*>(grp.Bt);

    // ---- operand staging: lane (sr, sc) owns the 16-byte chunk sc of rows sr + 32 p.
    // Loads are UNCONDITIONAL (addresses clamped into the tensors, zero selected afterwards): a
    // branch around a load makes hipcc wait vmcnt(0) per load and serialises the whole slab.
    const int sc = tid & 7, sr = tid >> 3;
    const T* a_row[AP];                                  // X row of this lane's tile row (clamped)
    int a_t[AP];
    bool a_ok[AP], b_ok[4];
    const T* b_row[4];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int m = m0 + sr + 32 * p;
        a_ok[p] = m < a.M;
        const int mm = min(m, a.M - 1);
        a_t[p] = mm % Tn;
        a_row[p] = X + (size_t)mm * a.ldx;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int n = n0 + sr + 32 * p;
        b_ok[p] = n < a.N;
        b_row[p] = Bt + (size_t)min(n, a.N - 1) * K;
    }
    const int pad_l = grp.pad_l, ldx = a.ldx;
    const bool pool = PRO >= 1 && a.pro_pool != 0;
    const bool nonneg = a.pro_pool == 2 || a.pro_relu;

    // gload() only ISSUES the 16-byte loads (and remembers which chunks are real); everything that
    // consumes them -- prologue, pool max, zero select, ds_write -- happens in lstore() after the
    // slab's MFMAs, so all loads of a slab are in flight together behind the matrix work.
    vec_t ra[AP], ra2[PRO >= 1 ? AP : 1], rb[4];
    unsigned okbits = 0;                                 // bit p: A chunk p valid, bit 8+p: B chunk p valid
    int st_c = 0;                                        // channel of the staged A chunk (prologue tables)
    int g_kk = sc * VEC, g_j = 0, g_c = sc * VEC;        // this lane's chunk in the next slab to load
    if (grp.taps > 1) { g_j = g_c / Cin; g_c -= g_j * Cin; }
    auto gload = [&]() {
        const bool kok = g_kk < K;
        const int j = kok ? g_j : 0, c = kok ? g_c : 0, kb = kok ? g_kk : 0;
        const int dj = j - pad_l;
        unsigned bits = 0;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int tt = a_t[p] + dj;
            const bool ok = kok && a_ok[p] && tt >= 0 && tt < Tn;
            const T* ptr = a_row[p] + (ok ? dj * ldx : 0) + c;
            ra[p] = *reinterpret_cast<const vec_t*>(ptr);
            if (PRO >= 1) {
                const T* ptr2 = ptr + ((pool && ok && tt + 1 < Tn) ? ldx : 0);   // last frame pools with itself
                ra2[p] = *reinterpret_cast<const vec_t*>(ptr2);
            }
            bits |= (ok ? 1u : 0u) << p;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            rb[p] = *reinterpret_cast<const vec_t*>(b_row[p] + kb);
            bits |= ((kok && b_ok[p]) ? 1u : 0u) << (8 + p);
        }
        okbits = bits;
        st_c = c;
        g_kk += BK;
        g_c += BK;
        if (grp.taps > 1)
            while (g_c >= Cin) { g_c -= Cin; ++g_j; }
    };
    auto lstore = [&](int buf) {
        char* As = smem + buf * BUF_BYTES;
        char* Bs = As + A_BYTES;
        const vec_t zero = {};
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            vec_t v = ra[p];
            if (PRO == 2) v = pro_apply(v, a.pro_scale, a.pro_shift, st_c, a.pro_relu);
            if (PRO >= 1) {
                vec_t v2 = ra2[p];
                if (PRO == 2) v2 = pro_apply(v2, a.pro_scale, a.pro_shift, st_c, a.pro_relu);
                if (pool) v = nonneg ? vmax_nonneg(v, v2) : vmax(v, v2);
            }
            *reinterpret_cast<vec_t*>(As + (sr + 32 * p) * ROWB + sc * 16) = ((okbits >> p) & 1u) ? v : zero;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<vec_t*>(Bs + (sr + 32 * p) * ROWB + sc * 16) = ((okbits >> (8 + p)) & 1u) ? rb[p] : zero;
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const EpiCoef coef = (MODE == VC_GEMM_PLAIN) ? epilogue_coef(a, grp, n0, wn, lane) : EpiCoef{};
    gload();
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload();                               // slab kt+1 in flight during the MFMAs
        const char* As = smem + (kt & 1) * BUF_BYTES;
        Mma<T, MI>::slab(As, As + A_BYTES, wm, wn, lane, acc);
        if (more) lstore((kt + 1) & 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------------------- epilogue
    const int i = lane & 31, h = lane >> 5;
    if (MODE == VC_GEMM_PLAIN) {
        epilogue_tile<T, MI>(a, grp, smem, acc, coef, m0, n0, wm * 32 * MI, wn, lane, tid);
    } else {
        // highway: columns come in (32 x dense1 | 32 x dense2) pairs; output unit index:
        const int hc = (n0 + wn * 64) / 2 + i;
        const int Hn = Cin;                               // highway keeps the width (modules.py:311-312)
        if (hc < Hn) {
            const int gnH = n0 + wn * 64 + i, gnT = gnH + 32;
            const float bH = a.epi_shift ? a.epi_shift[gnH] : 0.0f;
            const float bT = a.epi_shift ? a.epi_shift[gnT] : 0.0f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = m0 + wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (gm >= a.M) continue;
                    const float xv = to_f32(X[(size_t)gm * a.ldx + hc]);
                    store_out<T>(a.C, (size_t)gm * a.ldc + hc, vc::highway_gate(acc[mi][0][r] + bH, acc[mi][1][r] + bT, xv), a.out_f32);
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Convolution-specialised variant (taps > 1, Cin a multiple of the slab width): the K loop runs
// channel-slab OUTER, tap INNER, and ONE activation tile of BM + taps - 1 rows stays in LDS for
// all taps of a channel slab (tap j's A fragment is the same image read j rows further down), so
// per slab only the B tile streams.  That cuts the global->LDS operand traffic of the banks by
// ~2x and of the k = 3 projections by ~3x (their pooled operand is built once, not per tap) --
// the 128 x 128 bf16 tile is otherwise L2->LDS bandwidth bound.
// The zeros of TF's SAME padding depend on (output row, tap), not on the LDS row, so they are
// applied as a select on the fragment after the ds_read (lane i owns output rows with fixed t).
// LDS image: the padded 144-byte rows of gemm_kernel (a third resident block per CU from an
// unpadded XOR-swizzled image measured no faster, and the padded form keeps every fragment
// address = one per-slab base + compile-time immediates).
constexpr int CONV_MAX_TAPS = 32;
constexpr int CROWB = ROWB;
constexpr int conv_lds_bytes() {                       // A | B0 B1, and room for the f32 output tile of the epilogue
    return (128 + CONV_MAX_TAPS - 1) * CROWB + 2 * BN * CROWB > 128 * 528 ? (128 + CONV_MAX_TAPS - 1) * CROWB + 2 * BN * CROWB : 128 * 528;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * CROWB + (chunk << 4); }

template <typename T, int PRO>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
conv_kernel(KArgs a) {
    typedef typename Tr<T>::vec_t vec_t;
    constexpr int VEC = Tr<T>::VEC, BK = Tr<T>::BK;      // BK = channels per slab
    constexpr int BM = 128, MI = 2;
    constexpr int A_BYTES = (BM + CONV_MAX_TAPS - 1) * CROWB, B_BYTES = BN * CROWB;
    constexpr int NAP = 5;                                // ceil((128 + 31) * 8 / 256) staging passes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;
    char* const Bs0 = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int ntn = (a.N + BN - 1) / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int Cin = a.Cin, Tn = a.T, ldx = a.ldx;
    const int ncs = Cin / BK;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    // One group per block (blockIdx.y, heaviest first) -- or, sum_groups: every group in turn into the SAME accumulators
    // (the data gradient of a filter bank: one launch with the banks' whole K instead of one short-K launch per bank)
    // With sum_groups = S > 1 the groups are dealt to S blocks per tile (blockIdx.y; group g and its mirror n-1-g go
    // together, so equal work when the taps grow linearly) and the S partial tiles are added to C with float atomics:
    // a tile's worth of blocks alone (M/128 x N/128) would leave most of the chip idle.
    const int g_lo = a.sum_groups ? 0 : a.n_groups - 1 - (int)blockIdx.y, g_hi = a.sum_groups ? a.n_groups : g_lo + 1;
    for (int gi = g_lo; gi < g_hi; ++gi) {
    if (a.sum_groups > 1 && min(gi, a.n_groups - 1 - gi) % a.sum_groups != (int)blockIdx.y) continue;
    const KGroup grp = a.g[gi];
    const int K = grp.K, taps = grp.taps, pad_l = grp.pad_l;
    const int NR = BM + taps - 1;                         // rows of the resident A tile
    const T* X = reinterpret_cast<const T*>(a.X) + (a.sum_groups ? grp.c_off : 0);
    const T* Bt = reinterpret_cast<const T*>(grp.Bt);
    const bool pool = PRO >= 1 && a.pro_pool != 0;
    const bool nonneg = a.pro_pool == 2 || a.pro_relu;

    // ---- staging roles: chunk sc of tile rows sr + 32 p
    const int sc = tid & 7, sr = tid >> 3;
    const T* a_row[NAP];
    const T* a_row2[NAP];
    bool a_in[NAP];
#pragma unroll
    for (int p = 0; p < NAP; ++p) {
        const int r = sr + 32 * p;                        // tile row
        a_in[p] = r < NR;
        const int g = min(max(m0 - pad_l + r, 0), a.M - 1);   // global row (clamped; masked at read)
        a_row[p] = X + (size_t)g * ldx + sc * VEC;
        const bool nxt = pool && (g % Tn) + 1 < Tn;       // TF same-pool: last frame pools with itself
        a_row2[p] = a_row[p] + (nxt ? ldx : 0);
    }
    const T* b_row[4];
    bool b_ok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int n = n0 + sr + 32 * p;
        b_ok[p] = n < a.N;
        b_row[p] = Bt + (size_t)min(n, a.N - 1) * K + sc * VEC;
    }
    // ---- MFMA roles: lane i of wave (wm, wn) owns output rows wm*64 + mi*32 + i
    int jlo[MI], jhi[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = min(m0 + wm * 64 + mi * 32 + li, a.M - 1);
        const int t = m % Tn;
        jlo[mi] = max(0, pad_l - t);                      // taps [jlo, jhi) see a real frame
        jhi[mi] = min(taps, Tn - t + pad_l);
    }

    vec_t ra[NAP], ra2[PRO >= 1 ? NAP : 1];
    vec_t rbE[4], rbO[4];                                // B register sets of even / odd slabs
    auto gloadA = [&](int cs) {
        const int c = cs * BK;
#pragma unroll
        for (int p = 0; p < NAP; ++p) {
            ra[p] = *reinterpret_cast<const vec_t*>(a_row[p] + c);
            if (PRO >= 1) ra2[p] = *reinterpret_cast<const vec_t*>(a_row2[p] + c);
        }
    };
    auto lstoreA = [&](int cs) {
        const int c = cs * BK + sc * VEC;
#pragma unroll
        for (int p = 0; p < NAP; ++p) {
            vec_t v = ra[p];
            if (PRO == 2) v = pro_apply(v, a.pro_scale, a.pro_shift, c, a.pro_relu);
            if (PRO >= 1) {
                vec_t v2 = ra2[p];
                if (PRO == 2) v2 = pro_apply(v2, a.pro_scale, a.pro_shift, c, a.pro_relu);
                if (pool) v = nonneg ? vmax_nonneg(v, v2) : vmax(v, v2);
            }
            if (a_in[p]) *reinterpret_cast<vec_t*>(As + swz(sr + 32 * p, sc)) = v;
        }
    };
    // B loads run TWO slabs ahead of the MFMAs: a tile that arrives from L2 / Infinity Cache takes
    // longer than one slab of matrix work (16 MFMAs = 512 cycles per wave).
    int l_cs = 0, l_j = 0;                               // (channel slab, tap) of the next B slab to load
    auto gloadB = [&](vec_t* rb) {
        const int kb = l_j * Cin + l_cs * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const vec_t*>(b_row[p] + kb);
        if (++l_j == taps) { l_j = 0; ++l_cs; }
    };
    const bool full_n = n0 + BN <= a.N;                   // block-uniform: no B row needs zeroing
    auto lstoreB = [&](int buf, const vec_t* rb) {
        char* Bs = Bs0 + buf * B_BYTES + sr * CROWB + sc * 16;
        const vec_t zero = {};
        if (full_n) {
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<vec_t*>(Bs + 32 * p * CROWB) = rb[p];
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<vec_t*>(Bs + 32 * p * CROWB) = b_ok[p] ? rb[p] : zero;
        }
    };

    const int nit = ncs * taps;
    gloadA(0);
    gloadB(rbE);                                         // slab 0
    if (nit > 1) gloadB(rbO);                            // slab 1
    lstoreA(0);
    lstoreB(0, rbE);
    __syncthreads();
    int cs = 0, j = 0;
    auto body = [&](int it, vec_t* rb_load, const vec_t* rb_store) {
        // rb_load: set of slab `it` (already in LDS) -> refilled with slab it+2;  rb_store: slab it+1
        if (it + 2 < nit) gloadB(rb_load);
        const bool stageA = (j == 0) && (cs + 1 < ncs);   // next channel slab's tile: long flight
        if (stageA) gloadA(cs + 1);
        {   // MFMAs of slab (cs, j): A fragment = resident tile shifted down by j rows
            const char* ap = As + (wm * 64 + li + j) * CROWB + lh * 16;     // tile shifted down by j rows
            const char* bp = Bs0 + (it & 1) * B_BYTES + (wn * 64 + li) * CROWB + lh * 16;
            const bool v0 = j >= jlo[0] && j < jhi[0], v1 = j >= jlo[1] && j < jhi[1];
            const vec_t zero = {};
            // Straight-line k-step loop (no branch: the SAME-padding select is unconditional), with
            // the fragments of k-step s+1 fetched before the MFMAs of k-step s are issued.
            vec_t fa0[2], fa1[2], fb0[2], fb1[2];
            fa0[0] = *reinterpret_cast<const vec_t*>(ap);
            fa1[0] = *reinterpret_cast<const vec_t*>(ap + 32 * CROWB);
            fb0[0] = *reinterpret_cast<const vec_t*>(bp);
            fb1[0] = *reinterpret_cast<const vec_t*>(bp + 32 * CROWB);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int cur = s4 & 1, nxt = cur ^ 1;
                if (s4 < 3) {
                    fa0[nxt] = *reinterpret_cast<const vec_t*>(ap + (s4 + 1) * 32);
                    fa1[nxt] = *reinterpret_cast<const vec_t*>(ap + 32 * CROWB + (s4 + 1) * 32);
                    fb0[nxt] = *reinterpret_cast<const vec_t*>(bp + (s4 + 1) * 32);
                    fb1[nxt] = *reinterpret_cast<const vec_t*>(bp + 32 * CROWB + (s4 + 1) * 32);
                }
                const vec_t av0 = v0 ? fa0[cur] : zero;
                const vec_t av1 = v1 ? fa1[cur] : zero;
                const vec_t bv0 = fb0[cur], bv1 = fb1[cur];
                if constexpr (sizeof(T) == 2) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av0, bv0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av0, bv1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av1, bv0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av1, bv1, acc[1][1], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[q], bv0[q], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[q], bv1[q], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[q], bv0[q], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[q], bv1[q], acc[1][1], 0, 0, 0);
                    }
                }
            }
        }
        if (j == taps - 1 && cs + 1 < ncs) {              // retire the resident tile, bring in the next
            __syncthreads();
            lstoreA(cs + 1);
        }
        if (it + 1 < nit) lstoreB((it + 1) & 1, rb_store);
        __syncthreads();
        if (++j == taps) { j = 0; ++cs; }
    };
    for (int it = 0; it < nit; it += 2) {
        body(it, rbE, rbO);
        if (it + 1 < nit) body(it + 1, rbO, rbE);
    }
    }   // groups (every K loop ends behind a barrier: the next group's first tiles may overwrite the buffers)

    // ---------------------------------------------------------------------------- epilogue
    KGroup og = a.g[g_lo];
    if (a.sum_groups) og.c_off = 0;
    if (a.sum_groups > 1) {                               // partial tile: C[m, n] += acc (float32 output, no epilogue terms)
        float* C = reinterpret_cast<float*>(a.C);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int gn = n0 + wn * 64 + ni * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (gm < a.M && gn < a.N) unsafeAtomicAdd(C + (size_t)gm * a.ldc + gn, acc[mi][ni][r]);
                }
            }
        return;
    }
    const EpiCoef coef = epilogue_coef(a, og, n0, wn, lane);
    epilogue_tile<T, MI>(a, og, smem, acc, coef, m0, n0, wm * 64, wn, lane, tid);
}

template <typename T, int PRO> int launch_conv(const vc_gemm_desc* d, const KArgs& ka, hipStream_t st) {
    const int ntm = (d->M + 127) / 128, ntn = (d->N + BN - 1) / BN;
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<T, PRO>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_bytes()));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_kernel<T, PRO>), dim3(ntm * ntn, d->sum_groups ? d->sum_groups : d->n_groups), dim3(GEMM_THREADS), conv_lds_bytes(), st, ka);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}


// ------------------------------------------------------------------------------------------
// Weight gradient: dW[j*Cin + c, o] = sum_m X[m + j + shift0, c] * dY[m, o]   (frames of other
// windows excluded), i.e. the filter gradient of tf.layers.conv1d / dense, written straight in
// TF layout [taps, Cin, Cout].  The reduction runs over frames, so both operands come
// TRANSPOSED ([channels, frames], frames contiguous; vc_transpose_pad builds them with a zero
// margin so shifted reads stay inside the allocation): A row r = (j, c) is XT row c read
// (j + shift0) frames further along, B row o is dYT row o.  Same tiling / MFMA core as
// gemm_kernel; the SAME-padding rule is a per-element select while staging.
struct WGroup {
    const void* dYT;    // [N, ldyt] rows of this group's output channels
    void* dW;           // [taps*Cin, N] float32, row stride ldw
    int32_t N, taps, shift0, ldw;
};
struct WArgs {
    const void* XT;     // [Cin, ldxt], pointer at frame 0 (margin before it)
    int32_t ldxt, ldyt, Cin, M, T, n_groups;
    int32_t splits;     // frame range split over gridDim.z; > 1 => float atomics into a zeroed dW
    int32_t xcd_tiles;  // > 0: 1-D grid, XCD-aware block -> (group, tile, split) mapping; tiles per XCD and split
    WGroup g[VC_GEMM_MAX_GROUPS];
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ void __launch_bounds__(GEMM_THREADS, 2)
wgrad_kernel(WArgs a) {
    constexpr int MI = 2, BK = 32, BM = 128;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, BUF_BYTES = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // Block -> (group, tile, frame split).  Plain form: grid (tiles, groups, splits).  XCD-aware form
    // (grouped filter-bank launches): consecutive workgroup ids go round-robin to the 8 XCDs, each
    // with its own 4 MB L2, and every tile of a group streams the same dY^T rows -- so a group's
    // tiles are all given to ONE XCD (groups dealt to XCDs in snake order of their size, which
    // balances the tile counts), ordered split-major so that the blocks resident together read
    // the same frames.  With the plain mapping each XCD saw a slice of every group and dY^T
    // (210 MB for the decoder's step-2 bank) was re-fetched from HBM for each tile.
    int gsel, tile, zsplit;
    if (a.xcd_tiles > 0) {
        const int xcd = blockIdx.x & 7;
        int slot = blockIdx.x >> 3;
        zsplit = slot / a.xcd_tiles;
        slot -= zsplit * a.xcd_tiles;
        gsel = -1; tile = 0;
        for (int gi = 0; gi < a.n_groups; ++gi) {            // gi = rank by size (a.g is sorted ascending)
            const int r16 = gi & 15;
            if ((r16 < 8 ? r16 : 15 - r16) != xcd) continue;
            const WGroup& gg = a.g[a.n_groups - 1 - gi];
            const int tg = ((gg.taps * a.Cin + BM - 1) / BM) * ((gg.N + BN - 1) / BN);
            if (slot < tg) { gsel = a.n_groups - 1 - gi; tile = slot; break; }
            slot -= tg;
        }
        if (gsel < 0) return;
    } else {
        gsel = a.n_groups - 1 - (int)blockIdx.y;
        tile = blockIdx.x;
        zsplit = blockIdx.z;
    }
    const WGroup grp = a.g[gsel];
    const int R = grp.taps * a.Cin;
    const int ntn = (grp.N + BN - 1) / BN, ntr = (R + BM - 1) / BM;
    if (tile >= ntr * ntn) return;                       // groups have different tile counts
    const int rt = tile / ntn, nt = tile - rt * ntn;
    const int r0 = rt * BM, n0 = nt * BN;
    const int Tn = a.T, M = a.M;
    const int nk_all = (M + BK - 1) / BK;
    const int per = (nk_all + a.splits - 1) / a.splits;
    const int k_lo = zsplit * per, k_hi = min(nk_all, k_lo + per);
    const int nk = k_hi - k_lo;
    if (nk <= 0) return;
    const float* XT = reinterpret_cast<const float*>(a.XT);
    const float* YT = reinterpret_cast<const float*>(grp.dYT);

    const int sc = tid & 7, sr = tid >> 3;
    const float* a_base[4];
    const float* b_base[4];
    int a_dj[4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + sr + 32 * p;
        a_ok[p] = r < R;
        const int rr = min(r, R - 1);
        const int j = rr / a.Cin, c = rr - j * a.Cin;
        a_dj[p] = j + grp.shift0;
        a_base[p] = XT + (size_t)c * a.ldxt + a_dj[p] + sc * 4;
        const int n = n0 + sr + 32 * p;
        b_ok[p] = n < grp.N;
        b_base[p] = YT + (size_t)min(n, grp.N - 1) * a.ldyt + sc * 4;
    }
    f32x4 ra[4], rb[4];
    int g_m = k_lo * BK;                                  // first frame of the next slab to load
    int st_m = 0;
    auto gload = [&]() {
        // a tail slab (M % 32 != 0) reads up to 31 frames past M: inside the operands' zero margin
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = *reinterpret_cast<const f32x4u*>(a_base[p] + g_m);
            rb[p] = *reinterpret_cast<const f32x4u*>(b_base[p] + g_m);
        }
        st_m = g_m;
        g_m += BK;
    };
    auto lstore = [&](int buf) {
        char* As = smem + buf * BUF_BYTES;
        char* Bs = As + A_BYTES;
        const int m = st_m + sc * 4;
        const bool exact = true;
        const int t0 = m % Tn;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f32x4 va = ra[p], vb = rb[p];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ts = t0 + e + a_dj[p];
                const bool ok = exact && a_ok[p] && (m + e) < M && ts >= 0 && ts < Tn;
                va[e] = ok ? va[e] : 0.0f;
                vb[e] = (exact && b_ok[p] && (m + e) < M) ? vb[e] : 0.0f;
            }
            *reinterpret_cast<f32x4*>(As + (sr + 32 * p) * ROWB + sc * 16) = va;
            *reinterpret_cast<f32x4*>(Bs + (sr + 32 * p) * ROWB + sc * 16) = vb;
        }
    };
    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    gload();
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload();
        const char* As = smem + (kt & 1) * BUF_BYTES;
        Mma<float, MI>::slab(As, As + A_BYTES, wm, wn, lane, acc);
        if (more) lstore((kt + 1) & 1);
        __syncthreads();
    }
    const int i = lane & 31, h = lane >> 5;
    float* dW = reinterpret_cast<float*>(grp.dW);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int gn = n0 + wn * 64 + ni * 32 + i;
        if (gn >= grp.N) continue;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = r0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gr < R) {
                    if (a.splits > 1) atomicAdd(dW + (size_t)gr * grp.ldw + gn, acc[mi][ni][r]);
                    else dW[(size_t)gr * grp.ldw + gn] = acc[mi][ni][r];
                }
            }
    }
}

template <typename T, int MODE, int MI, int PRO> int launch_one(const vc_gemm_desc* d, const KArgs& ka, hipStream_t st) {
    constexpr int BM = 64 * MI;
    const int ntm = (d->M + BM - 1) / BM, ntn = (d->N + BN - 1) / BN;
    dim3 grid(ntm * ntn, d->n_groups), block(GEMM_THREADS);
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, MODE, MI, PRO>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(MI)));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_kernel<T, MODE, MI, PRO>), grid, block, lds_bytes(MI), st, ka);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

template <typename T, int MI> int launch_mi(const vc_gemm_desc* d, const KArgs& ka, hipStream_t st) {
    if (d->mode == VC_GEMM_HIGHWAY) return launch_one<T, VC_GEMM_HIGHWAY, MI, 0>(d, ka, st);
    if (d->d_pro_scale || d->pro_relu) return launch_one<T, VC_GEMM_PLAIN, MI, 2>(d, ka, st);
    if (d->pro_pool) return launch_one<T, VC_GEMM_PLAIN, MI, 1>(d, ka, st);
    return launch_one<T, VC_GEMM_PLAIN, MI, 0>(d, ka, st);
}

// The filter-bank launch in its paired 256 x 256 form (vc_bank256.hip): bf16, plain operand, groups
// (2p+1, 2p+2) taps wide with a common left padding, 128 filters each, BatchNorm/ReLU epilogue.
bool bank256_ok(const vc_gemm_desc* d) {
    if (vc::opt(vc::OPT_BANK256) == 0 || d->dtype != VC_BF16 || d->mode != VC_GEMM_PLAIN) return false;
    if (d->n_groups < 2 || (d->n_groups & 1) || d->n_groups > 32 || d->N != 128 || d->Cin % 64 || d->M < 256) return false;
    if (d->d_pro_scale || d->pro_relu || d->pro_pool || d->d_R || d->out_f32 || d->drop_keep > 0.0f) return false;
    if (d->act != VC_ACT_NONE && d->act != VC_ACT_RELU) return false;
    if ((reinterpret_cast<uintptr_t>(d->d_C) & 15) || d->ldx % 8 || d->ldc % 8) return false;
    for (int g = 0; g < d->n_groups; g += 2) {
        const vc_gemm_group& a = d->groups[g];
        const vc_gemm_group& b = d->groups[g + 1];
        if (b.taps != a.taps + 1 || a.pad_l != b.pad_l || b.taps > 32 || a.taps < 1) return false;
        if (a.c_off % 8 || b.c_off % 8) return false;
    }
    return true;
}

int launch_bank256(const vc_gemm_desc* d, hipStream_t st) {
    Bank256Args b;
    b.X = d->d_X; b.M = d->M; b.T = d->T; b.Cin = d->Cin; b.ldx = d->ldx;
    b.epi_scale = d->d_epi_scale; b.epi_shift = d->d_epi_shift; b.act = d->act;
    b.C = d->d_C; b.ldc = d->ldc; b.n_pairs = d->n_groups / 2;
    b.pool = d->epi_pool != 0;
    b.ksplit = 1; b.ws = nullptr; b.tick = nullptr;
    for (int g = 0; g < d->n_groups; g += 2) {
        Bank256Pair& p = b.p[g / 2];
        p.Bt0 = d->groups[g].d_Bt; p.Bt1 = d->groups[g + 1].d_Bt;
        p.taps0 = d->groups[g].taps; p.pad_l = d->groups[g].pad_l;
        p.c_off0 = d->groups[g].c_off; p.c_off1 = d->groups[g + 1].c_off;
        p.extra = 1;
    }
    b.dbg = 0;
#ifdef VC_ABLATE
    b.dbg = vc::opt(vc::OPT_ABLATE_BANK256) > 0 ? vc::opt(vc::OPT_ABLATE_BANK256) : 0;
    if (const int p = vc::opt(vc::OPT_ABLATE_BANK256_ONLY); p >= 0 && p < b.n_pairs) { b.p[0] = b.p[p]; b.n_pairs = 1; }
#endif
    return vc_launch_bank256(b, st);
}

// Single-filter bf16 convolutions / dense layers with a long K on the deep-pipelined 128-row tiles
// (vc_conv256.hip): plain operand or non-negative max-pool, N a multiple of 128, taps <= 7.
bool conv256_ok(const vc_gemm_desc* d) {
    if (vc::opt(vc::OPT_CONV256) == 0 || d->dtype != VC_BF16 || d->mode != VC_GEMM_PLAIN || d->n_groups != 1) return false;
    const vc_gemm_group& g = d->groups[0];
    const int min_k = vc::opt(vc::OPT_CONV256_MINK) > 0 ? vc::opt(vc::OPT_CONV256_MINK) : 384;           // measured in the pipelined step: 1024 -> 384 = -0.9 % ms/step (the second k = 3 projections)
    if (d->N % 128 || d->Cin % 64 || d->M < 128 || g.taps > 7 || g.taps * d->Cin < min_k) return false;
    if (d->d_pro_scale || d->pro_relu || d->pro_pool == 1 || d->out_f32 || d->drop_keep > 0.0f) return false;
    if ((reinterpret_cast<uintptr_t>(d->d_C) & 15) || d->ldx % 8 || d->ldc % 8 || g.c_off % 8) return false;
    if (d->d_R && ((reinterpret_cast<uintptr_t>(d->d_R) & 7) || d->ldr % 4)) return false;
    return true;
}

// A single 256-channel filter with a long K (the first k = 3 projection of decoder stage 2: K = 3 x 4096) on the
// bank kernel's 256 x 256 tiles, its two 128-channel halves standing in for a pair of equal width: 0.75 fragment
// reads per MFMA instead of conv256_kernel's 1.0 and half the weight traffic per frame.  Only M / 256 workgroups
// (100 at 64 windows), so ALONE on the chip the launch is slower than conv256_kernel's 200 (0.29 vs 0.23 ms) -- but
// CU time, not the makespan, is what a launch costs once several batches are in flight:
// 100 x 0.29 ms against 200 x 0.23 ms.  vc_set_option("proj256", 0) switches it off.
bool proj256_ok(const vc_gemm_desc* d) {
    if (vc::opt(vc::OPT_PROJ256) == 0 || d->dtype != VC_BF16 || d->mode != VC_GEMM_PLAIN || d->n_groups != 1) return false;
    const vc_gemm_group& g = d->groups[0];
    if (d->N != 256 || d->Cin % 64 || d->M < 1024 || g.taps < 2 || g.taps > 32 || g.taps * d->Cin < 4096) return false;
    if (d->d_pro_scale || d->pro_relu || d->pro_pool || d->d_R || d->out_f32 || d->drop_keep > 0.0f || d->epi_pool) return false;
    if (d->act != VC_ACT_NONE && d->act != VC_ACT_RELU) return false;
    if ((reinterpret_cast<uintptr_t>(d->d_C) & 15) || d->ldx % 8 || d->ldc % 8 || g.c_off % 8) return false;
    return true;
}

// Split of that projection's K over two workgroups per row tile (vc_bank256.hip, "split K"): 100 row tiles at 64 windows
// leave 156 CUs idle when the launch has the chip to itself; two halves of K per tile fill 200.
int proj256_ksplit(const vc_gemm_desc* d) {
    if (vc::opt(vc::OPT_PROJ256_SPLIT) == 0) return 1;
    return vc_bank256_ksplit(d->M, d->Cin / 64);
}

int launch_proj256(const vc_gemm_desc* d, hipStream_t st) {
    Bank256Args b;
    const vc_gemm_group& g = d->groups[0];
    b.ksplit = 1; b.ws = nullptr; b.tick = nullptr;
    if (const int ks = proj256_ksplit(d); ks > 1 && d->d_workspace && (reinterpret_cast<uintptr_t>(d->d_workspace) & 255) == 0 &&
        d->workspace_bytes >= vc_bank256_ws_bytes(d->M, ks)) {
        b.ksplit = ks;
        b.tick = static_cast<unsigned*>(d->d_workspace);
    }
    b.X = d->d_X; b.M = d->M; b.T = d->T; b.Cin = d->Cin; b.ldx = d->ldx;
    b.epi_scale = d->d_epi_scale; b.epi_shift = d->d_epi_shift; b.act = d->act;
    b.C = d->d_C; b.ldc = d->ldc; b.n_pairs = 1; b.pool = 0; b.dbg = 0;
    Bank256Pair& p = b.p[0];
    p.Bt0 = g.d_Bt;
    p.Bt1 = static_cast<const char*>(g.d_Bt) + (size_t)128 * g.K * 2;      // rows 128.. of the [256, K] bf16 matrix
    p.taps0 = g.taps; p.pad_l = g.pad_l; p.c_off0 = g.c_off; p.c_off1 = g.c_off + 128; p.extra = 0;
    return vc_launch_bank256(b, st);
}

int launch_conv256(const vc_gemm_desc* d, hipStream_t st) {
    Conv256Args c;
    const vc_gemm_group& g = d->groups[0];
    c.X = d->d_X; c.M = d->M; c.T = d->T; c.Cin = d->Cin; c.ldx = d->ldx; c.N = d->N;
    c.Bt = g.d_Bt; c.K = g.K; c.taps = g.taps; c.pad_l = g.pad_l; c.c_off = g.c_off;
    c.pool = d->pro_pool != 0;
    c.epi_scale = d->d_epi_scale; c.epi_shift = d->d_epi_shift; c.act = d->act;
    c.R = d->d_R; c.ldr = d->ldr; c.C = d->d_C; c.ldc = d->ldc;
    return vc_launch_conv256(c, st);
}

template <typename T> int launch(const vc_gemm_desc* d, const KArgs& ka, hipStream_t st) {
    if (d->sum_groups) {             // partial sums over the groups: conv_kernel's group loop is the only form
        bool ok = d->mode == VC_GEMM_PLAIN && d->Cin % Tr<T>::BK == 0 && d->M >= 128 && !d->epi_pool;
        for (int g = 0; g < d->n_groups; ++g) ok = ok && d->groups[g].taps <= CONV_MAX_TAPS;
        if (!ok) return vc::set_error(VC_ERR_INVALID, "vc_conv_gemm: sum_groups needs plain mode, Cin %% %d == 0, M >= 128, taps <= %d",
                                      Tr<T>::BK, CONV_MAX_TAPS);
        if (d->sum_groups > 1)
            VC_REQUIRE((d->out_f32 || sizeof(T) == 4) && !d->d_R && !d->d_epi_scale && !d->d_epi_shift && d->act == VC_ACT_NONE &&
                           d->drop_keep == 0.0f && d->sum_groups <= 16 && 2 * d->sum_groups <= d->n_groups + 1,
                       "vc_conv_gemm: sum_groups > 1 adds bare float32 partial tiles to C (no residual / scale / shift / activation), at most (n_groups + 1) / 2 splits");
        if (d->d_pro_scale || d->pro_relu) return launch_conv<T, 2>(d, ka, st);
        if (d->pro_pool) return launch_conv<T, 1>(d, ka, st);
        return launch_conv<T, 0>(d, ka, st);
    }
    if (sizeof(T) == 2 && proj256_ok(d)) return launch_proj256(d, st);
    if (sizeof(T) == 2 && conv256_ok(d)) return launch_conv256(d, st);
    // convolution-specialised kernel: every group has taps in [2, 32] (a grouped bank launch may
    // include its k = 1 member) and Cin is a whole number of channel slabs
    bool conv_ok = d->mode == VC_GEMM_PLAIN && d->Cin % Tr<T>::BK == 0 && d->M >= 128;
    int max_taps = 1;
    for (int g = 0; g < d->n_groups; ++g) {
        if (d->groups[g].taps > CONV_MAX_TAPS) conv_ok = false;
        if (d->groups[g].taps > max_taps) max_taps = d->groups[g].taps;
    }
    if (conv_ok && max_taps > 1) {
        if (sizeof(T) == 2 && bank256_ok(d)) return launch_bank256(d, st);
        if (d->d_pro_scale || d->pro_relu) return launch_conv<T, 2>(d, ka, st);
        if (d->pro_pool) return launch_conv<T, 1>(d, ka, st);
        return launch_conv<T, 0>(d, ka, st);
    }
    // 128-row tiles unless that leaves the 256 CUs with fewer than ~2 blocks each
    const long blocks128 = (long)((d->M + 127) / 128) * ((d->N + BN - 1) / BN) * d->n_groups;
    return blocks128 < 512 ? launch_mi<T, 1>(d, ka, st) : launch_mi<T, 2>(d, ka, st);
}

}  // namespace

extern "C" int vc_conv_gemm(const vc_gemm_desc* d, void* stream) {
    VC_REQUIRE(d != nullptr, "desc is NULL");
    VC_REQUIRE(d->dtype == VC_F32 || d->dtype == VC_BF16, "bad dtype %d", d->dtype);
    VC_REQUIRE(d->mode == VC_GEMM_PLAIN || d->mode == VC_GEMM_HIGHWAY, "bad mode %d", d->mode);
    VC_REQUIRE(d->d_X && d->d_C, "NULL X or C");
    VC_REQUIRE(d->M > 0 && d->T > 0 && d->Cin > 0 && d->N > 0, "bad shape M=%d T=%d Cin=%d N=%d", d->M, d->T, d->Cin, d->N);
    VC_REQUIRE(d->M % d->T == 0, "M (%d) must be a multiple of T (%d)", d->M, d->T);
    VC_REQUIRE(d->n_groups >= 1 && d->n_groups <= VC_GEMM_MAX_GROUPS, "n_groups out of range: %d", d->n_groups);
    const int vec = d->dtype == VC_F32 ? 4 : 8;
    VC_REQUIRE(d->Cin % vec == 0 && d->ldx % vec == 0 && d->ldx >= d->Cin,
               "Cin (%d) and ldx (%d) must be multiples of %d with ldx >= Cin", d->Cin, d->ldx, vec);
    VC_REQUIRE((reinterpret_cast<uintptr_t>(d->d_X) & 15) == 0, "X must be 16-byte aligned");
    VC_REQUIRE(!(d->d_pro_scale == nullptr) == !(d->d_pro_shift == nullptr), "pro_scale and pro_shift go together");
    if (d->mode == VC_GEMM_HIGHWAY)
        VC_REQUIRE(d->n_groups == 1 && d->groups[0].taps == 1 && d->N % 64 == 0 && d->N >= 2 * d->Cin && !d->d_R &&
                       !d->d_pro_scale && !d->pro_relu && !d->pro_pool,
                   "highway mode: one group, taps 1, N = 64*ceil(H/32) paired columns, no residual, no prologue");
    KArgs ka;
    ka.X = d->d_X; ka.M = d->M; ka.T = d->T; ka.Cin = d->Cin; ka.ldx = d->ldx; ka.N = d->N; ka.n_groups = d->n_groups;
    ka.pro_scale = d->d_pro_scale; ka.pro_shift = d->d_pro_shift; ka.pro_relu = d->pro_relu; ka.pro_pool = d->pro_pool;
    ka.epi_scale = d->d_epi_scale; ka.epi_shift = d->d_epi_shift; ka.act = d->act;
    ka.R = d->d_R; ka.ldr = d->ldr; ka.C = d->d_C; ka.ldc = d->ldc; ka.out_f32 = d->out_f32;
    ka.drop_keep = d->drop_keep; ka.drop_seed = d->drop_seed; ka.sum_groups = d->sum_groups;
    VC_REQUIRE(d->drop_keep >= 0.0f && d->drop_keep <= 1.0f, "drop_keep must be in [0, 1]");
    for (int g = 0; g < d->n_groups; ++g) {
        const vc_gemm_group& gg = d->groups[g];
        VC_REQUIRE(gg.d_Bt != nullptr && (reinterpret_cast<uintptr_t>(gg.d_Bt) & 15) == 0, "group %d: Bt NULL or misaligned", g);
        VC_REQUIRE(gg.taps >= 1 && gg.K == gg.taps * d->Cin, "group %d: K (%d) != taps (%d) * Cin (%d)", g, gg.K, gg.taps, d->Cin);
        VC_REQUIRE(gg.pad_l >= 0 && gg.pad_l < gg.taps, "group %d: bad pad_l %d", g, gg.pad_l);
        if (d->sum_groups)
            VC_REQUIRE(gg.c_off >= 0 && gg.c_off % vec == 0 && gg.c_off + d->Cin <= d->ldx && d->N <= d->ldc,
                       "group %d (sum_groups): input channels [c_off, c_off + Cin) must lie inside a row of X", g);
        else
            VC_REQUIRE(gg.c_off >= 0 && gg.c_off + d->N <= d->ldc || d->mode == VC_GEMM_HIGHWAY, "group %d: columns exceed ldc", g);
        ka.g[g].Bt = gg.d_Bt; ka.g[g].K = gg.K; ka.g[g].taps = gg.taps; ka.g[g].pad_l = gg.pad_l; ka.g[g].c_off = gg.c_off;
    }
    if (d->epi_pool)
        VC_REQUIRE(d->act == VC_ACT_RELU && bank256_ok(d),
                   "epi_pool needs act = ReLU and a launch vc_conv_gemm_epi_pool_supported() accepts");
    hipStream_t st = static_cast<hipStream_t>(stream);
    return d->dtype == VC_F32 ? launch<float>(d, ka, st) : launch<__bf16>(d, ka, st);
}

extern "C" size_t vc_conv_gemm_workspace_bytes(const vc_gemm_desc* d) {
    if (d == nullptr || d->d_X == nullptr || d->n_groups < 1 || d->n_groups > VC_GEMM_MAX_GROUPS) return 0;
    if (d->sum_groups || d->dtype != VC_BF16 || !proj256_ok(d)) return 0;
    return vc_bank256_ws_bytes(d->M, proj256_ksplit(d));
}

extern "C" int vc_conv_gemm_epi_pool_supported(const vc_gemm_desc* d) {
    return d != nullptr && d->act == VC_ACT_RELU && d->M > 0 && d->T > 0 && d->M % d->T == 0 && bank256_ok(d) ? 1 : 0;
}

extern "C" int vc_conv_wgrad(const vc_wgrad_desc* d, void* stream) {
    VC_REQUIRE(d != nullptr && d->d_XT != nullptr, "NULL desc / XT");
    VC_REQUIRE(d->Cin > 0 && d->M > 0 && d->T > 0 && d->M % d->T == 0 && d->T % 4 == 0, "bad shape Cin=%d M=%d T=%d", d->Cin, d->M, d->T);
    VC_REQUIRE(d->n_groups >= 1 && d->n_groups <= VC_GEMM_MAX_GROUPS, "n_groups out of range");
    VC_REQUIRE(d->margin >= 32, "operand margin must be >= 32 frames (and one slack row after the last)");
    WArgs wa;
    wa.XT = d->d_XT; wa.ldxt = d->ldxt; wa.ldyt = d->ldyt; wa.Cin = d->Cin; wa.M = d->M; wa.T = d->T; wa.n_groups = d->n_groups;
    int max_tiles = 0;
    for (int g = 0; g < d->n_groups; ++g) {
        const vc_wgrad_group& gg = d->groups[g];
        VC_REQUIRE(gg.d_dYT && gg.d_dW && gg.N > 0 && gg.taps >= 1 && gg.taps <= 32, "group %d: bad arguments", g);
        VC_REQUIRE(gg.shift0 > -d->margin && gg.shift0 + gg.taps - 1 < d->margin, "group %d: shift exceeds the margin", g);
        wa.g[g].dYT = gg.d_dYT; wa.g[g].dW = gg.d_dW; wa.g[g].N = gg.N; wa.g[g].taps = gg.taps; wa.g[g].shift0 = gg.shift0;
        wa.g[g].ldw = gg.ldw > 0 ? gg.ldw : gg.N;
        const int tiles = ((gg.taps * d->Cin + 127) / 128) * ((gg.N + BN - 1) / BN);
        if (tiles > max_tiles) max_tiles = tiles;
    }
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2)));
        attr_done = true;
    }
    // Small grids (dense / highway / GRU filters) would run hundreds of K slabs serially on a few
    // CUs: split the frame range over gridDim.z until ~2 blocks per CU exist.  The caller zeroes dW
    // when it passes splits_allowed (atomic accumulation); the summation order is then not fixed.
    int splits = 1;
    const long blocks = (long)max_tiles * d->n_groups;
    const int nk_all = (d->M + 31) / 32;
    if (d->splits_allowed)
        while (blocks * splits < 512 && splits * 2 <= 32 && nk_all / (splits * 2) >= 8) splits *= 2;
    wa.splits = splits;
    wa.xcd_tiles = 0;
    // Grouped launches with enough tiles to fill the chip: XCD-aware mapping (see wgrad_kernel).
    // The groups arrive sorted by size (the kernel reads a.g[] back to front as "heaviest first").
    bool sorted = d->n_groups >= 8;
    for (int g = 1; g < d->n_groups && sorted; ++g) sorted = d->groups[g].taps >= d->groups[g - 1].taps;
    if (sorted && vc::opt(vc::OPT_WGRAD_XCD) != 0) {
        long per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, total = 0;
        for (int gi = 0; gi < d->n_groups; ++gi) {
            const vc_wgrad_group& gg = d->groups[d->n_groups - 1 - gi];
            const int r16 = gi & 15;
            const long tg = (long)((gg.taps * d->Cin + 127) / 128) * ((gg.N + BN - 1) / BN);
            per_xcd[r16 < 8 ? r16 : 15 - r16] += tg;
            total += tg;
        }
        long mx = 0;
        for (int x = 0; x < 8; ++x) mx = per_xcd[x] > mx ? per_xcd[x] : mx;
        if (total >= 512 && mx > 0) {
            // 64 blocks are resident per XCD; a frame split that lands the rounds on a whole number
            // beats a ragged last round (atomics: only where the caller allows them)
            int best = 1;
            if (d->splits_allowed) {
                double best_cost = 1e30;
                for (int sp = 1; sp <= 8 && nk_all / sp >= 16; sp *= 2) {
                    const double cost = (double)((mx * sp + 63) / 64) / sp * (1.0 + 0.02 * (sp - 1));
                    if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
                }
            }
            wa.splits = best;
            wa.xcd_tiles = (int)mx;
            hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(8 * mx * best)), dim3(GEMM_THREADS), lds_bytes(2),
                               static_cast<hipStream_t>(stream), wa);
            VC_HIP_CHECK(hipGetLastError());
            return VC_OK;
        }
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3(max_tiles, d->n_groups, splits), dim3(GEMM_THREADS), lds_bytes(2),
                       static_cast<hipStream_t>(stream), wa);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}
