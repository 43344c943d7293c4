// Training GEMMs on split-float16 operands ("f16x3"): float32-accurate convolutions at the 16-bit MFMA rate.
//
// The decoder's training step (/root/reference/decoder.py:185-263, 327-345) is float32 in the reference, and gfx950's
// f32-input MFMA runs at 1/16 of the 16-bit rate.  A float32 value x scaled by a power of two s splits EXACTLY into
// x*s = hi + lo + r with hi, lo float16 (11 significand bits each) and |r| <= 2^-22 |x*s|; the three products
// hi*hi + hi*lo + lo*hi, exact in the MFMA and summed in float32, reproduce the float32 product to 2^-22 -- below the
// rounding of a float32 accumulation itself (tests/test_gemm16_gpu.py: relative L2 error vs float64 equal to the
// f32-MFMA kernel's, 4e-7 at K = 8,192).  Scales are powers of two (per frame for activations: vc_split16; per
// tensor for weights: vc_weights16), so scaling and un-scaling are exact.
//
// gemm16_kernel is the tile structure of bank256_kernel (vc_bank256.hip: 256 frames x (128 + 128) output channels per
// workgroup, 8 waves x 128 x 64 accumulators, LDS-direct operand loads into XOR-swizzled 128-byte rows, one barrier
// per K tile) with
//   * float16 operands stored as two planes [hi | lo]; the K loop walks three plane PRODUCTS (lo*hi, hi*lo, hi*hi:
//     small terms first) by re-addressing the same planes -- nothing is stored twice,
//   * float32 output: acc * row_scale[frame] * col_scale[channel] + col_shift[channel] (+ the previous contents),
//     staged through LDS 128 rows at a time into full 512-byte row stores,
//   * a "ragged" K walk for the filter bank's data gradient (the sum over the banks k of conv(dZ_k, W_k^T flipped),
//     tf.gradients through /root/reference/modules.py:144-166): channel slab -> bank -> its own tap count and padding,
//   * K split over up to 8 workgroups per row tile for single-pair launches (every workgroup publishes its accumulators
//     write-through, then takes a ticket; the last one adds the slabs in split order -- bit-identical run to run).
#include <cstdlib>
#include "vc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512;
constexpr int BM = 256;
constexpr int A_ROWS = 288;                        // 256 + 32 halo rows
constexpr int A_BYTES = A_ROWS * 128;
constexpr int B_BYTES = 256 * 128;
constexpr int COEF_OFF = 2 * A_BYTES + 2 * B_BYTES;    // 139,264: col_scale[256] | col_shift[256] of the pair (f32)
constexpr int LDS_BYTES = COEF_OFF + 2 * 256 * 4;
constexpr int EP = 1040;                           // LDS row pitch of the [128][256] float32 output half tile
static_assert(128 * EP <= COEF_OFF, "output half tile must not reach the coefficients");
constexpr int RAG_PAD = 16;                        // ragged walk: the activation tile starts 16 frames before the row tile
constexpr int MAX_SPLIT = 8;

struct G16Pair {
    const void* Bt0;      // [128][K0] float16, K contiguous: per tap [hi plane | lo plane] of the channels
    const void* Bt1;      // [128][K1]
    int32_t taps0, extra, pad_l, c_off0, c_off1, K0, K1;
    int32_t row0;         // first row of X this pair's row tile 0 reads (0 unless the pairs walk different row ranges)
    int32_t nrows0, nrows1;   // output rows (counted from row0) each filter stores
    int32_t s_off0, s_off1;   // first entry of col_scale / col_shift of each filter (= c_off* unless the output is scattered)
};

struct G16Args {
    const void* X;        // [M][ldx] float16: [hi plane (C) | lo plane (C)]
    const float* row_scale;
    int32_t M, T, C, ldx;
    const float* col_scale;
    const float* col_shift;
    float* Cout;
    int32_t ldc, accumulate, n_pairs, ragged, act;
    int32_t zsplit;       // >= 1: blockIdx.z walks its own K range and ADDS its tile to Cout with float atomics (any pair count)
    int32_t xcd_tiles;
    int16_t seg_pair[8][4], seg_first[8][4], seg_count[8][4];
    int32_t ksplit, split_map;           // split_map 1: K range ks on the XCDs == ks (mod ksplit); ksplit divides 8
    int16_t split_cs[MAX_SPLIT + 1];     // K slabs [split_cs[ks], split_cs[ks + 1]) of the 3 * C / 64
    float* ws;            // [row tiles][ksplit][256 * 256] float32 partial accumulators (register order)
    unsigned* tick;       // [row tiles] arrival counters, zeroed by the launcher before every launch
    G16Pair p[16];
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 16, 0, 0);
}

// Position of the K walk (all wave-uniform): K slab cs = plane * nsl + cr, tap j of the slab's `taps`, left padding `pad`.
struct Walk { int cs, cr, plane, j, taps, pad; };

__global__ void __launch_bounds__(NT, 1)
gemm16_kernel(G16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;                             // [2][288][128]
    char* const Bs = smem + 2 * A_BYTES;               // [2][256][128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int psel, rt, ks = 0;
    if (a.ksplit > 1) {
        if (a.split_map) {
            // one K range per XCD (its weight stream stays in that L2); ksplit divides 8
            const int q = blockIdx.x >> 3, xcd = blockIdx.x & 7;
            ks = xcd % a.ksplit;
            rt = q * (8 / a.ksplit) + xcd / a.ksplit;
        } else {
            // compact grid: exactly row tiles x splits workgroups, so that up to 256 of them are ONE round of the CUs
            // (the slabs travel through memory at agent scope: the splits of a row tile need not share an XCD)
            ks = blockIdx.x % a.ksplit;
            rt = blockIdx.x / a.ksplit;
        }
        psel = 0;
    } else if (a.xcd_tiles > 0) {
        const int xcd = blockIdx.x & 7;
        int slot = blockIdx.x >> 3;
        psel = -1; rt = 0;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const int cnt = a.seg_count[xcd][sg];
            if (psel < 0 && slot < cnt) { psel = a.seg_pair[xcd][sg]; rt = a.seg_first[xcd][sg] + slot; }
            slot -= cnt;
        }
        if (psel < 0) return;
    } else {
        psel = a.n_pairs - 1 - (int)blockIdx.y;               // widest pair first
        rt = blockIdx.x;
        if (a.zsplit >= 1) ks = blockIdx.z;
    }
    const G16Pair pr = a.p[psel];
    if (rt * BM >= max(pr.nrows0, pr.nrows1)) return;
    // wave tile: rows wr*128.., cols wc*64.. (wc < 2: the first filter).  A pair whose second filter stores nothing is a
    // single 128-column filter: waves 0-3 (one per SIMD) take its 2 x 2 wave tiles, waves 4-7 only stage operands
    const bool half_only = pr.nrows1 == 0;
    const int wr = half_only ? (wid >> 1) & 1 : wid >> 2, wc = half_only ? wid & 1 : wid & 3;
    const bool dead_wave = half_only && wid >= 4;
    const int wpass = __builtin_amdgcn_readfirstlane(dead_wave ? 2 : wr);     // epilogue pass this wave writes (2: none)
    const int m0 = pr.row0 + rt * BM;
    const bool ragged = a.ragged != 0;
    const int ntap_u = pr.taps0 + pr.extra;            // uniform walk: taps of the wider filter
    const int narrow = ragged ? 0x7fffffff : pr.taps0; // taps of the first filter (it has no tap `narrow`)
    const int nsl = a.C >> 6;                          // channel slabs per plane
    const int PL = (nsl >> 1) * ((nsl >> 1) + 1) / 2 * 128;    // ragged weights: elements per plane of a row
    const int cs0 = a.split_cs[ks], cs1 = a.split_cs[ks + 1];
    const int nslab = cs1 - cs0;
    const int padA = ragged ? RAG_PAD : pr.pad_l;      // the activation tile starts at frame m0 - padA
    const _Float16* X = reinterpret_cast<const _Float16*>(a.X);

    auto seg_set = [&](Walk& w) {
        if (ragged) { const int k = (w.cr >> 1) + 1; w.taps = k; w.pad = k >> 1; }
        else { w.taps = ntap_u; w.pad = pr.pad_l; }
    };
    auto advance = [&](Walk& w) {
        if (++w.j == w.taps) {
            w.j = 0; ++w.cs;
            if (++w.cr == nsl) { w.cr = 0; ++w.plane; }
            seg_set(w);
        }
    };
    Walk w0;
    w0.cs = cs0; w0.plane = cs0 / nsl; w0.cr = cs0 - w0.plane * nsl; w0.j = 0;
    seg_set(w0);
    int ntiles = 0;
    if (ragged) {
        int cr = w0.cr;
        for (int c = cs0; c < cs1; ++c) { ntiles += (cr >> 1) + 1; if (++cr == nsl) cr = 0; }
    } else {
        ntiles = nslab * ntap_u;
    }

    // ---------------- staging roles (LDS-direct loads; one wave instruction = 8 rows = 1 KB)
    const int srow = lane >> 3;
    const int pslot = lane & 7;
    const _Float16* a_src[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int rho = (q * 8 + wid) * 8 + srow;
        const int g = min(max(m0 - padA + rho, 0), a.M - 1);
        const int slot = pslot ^ ((rho >> 1) & 7);
        a_src[q] = X + (size_t)g * a.ldx + slot * 8;
    }
    const int a_rows_needed = ragged ? A_ROWS : BM + ntap_u - 1;
    const _Float16* b_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = (q * 8 + wid) * 8 + srow;
        const int slot = pslot ^ ((n >> 1) & 7);
        const bool left = n < 128;
        const _Float16* Bt = reinterpret_cast<const _Float16*>(left ? pr.Bt0 : pr.Bt1);
        b_src[q] = Bt + (size_t)(n & 127) * (left ? pr.K0 : pr.K1) + slot * 8;
    }
    // plane products, small terms first: plane 0 = x_lo * w_hi, 1 = x_hi * w_lo, 2 = x_hi * w_hi
    auto stageA = [&](int plane, int cr, int buf) {
        const int xo = (plane == 0 ? a.C : 0) + cr * 64;
        char* dst = As + buf * A_BYTES + wid * 1024;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int rb = q * 8 + wid;
            if (rb * 8 < a_rows_needed && rb < A_ROWS / 8) glds16(a_src[q] + xo, dst + q * 8192);
        }
    };
    auto stageB = [&](const Walk& w, int buf) {
        const int wpl = w.plane == 1 ? 1 : 0;
        int koffL, koffR;
        if (ragged) {
            koffL = koffR = wpl * PL + (w.taps * (w.taps - 1) / 2 + w.j) * 128 + (w.cr & 1) * 64;
        } else {
            const int inner = wpl * a.C + w.cr * 64;
            koffR = w.j * 2 * a.C + inner;
            koffL = min(w.j, narrow - 1) * 2 * a.C + inner;
        }
        char* dst = Bs + buf * B_BYTES + wid * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q >= 2 ? !half_only : w.j < narrow) glds16(b_src[q] + (q < 2 ? koffL : koffR), dst + q * 8192);
    };

    // ---------------- MFMA roles
    const int li = lane & 31, lh = lane >> 5;
    const int a_row0 = wr * 128 + li;
    const int xb = (li >> 1) & 7;
    int b_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b_off[s] = (wc * 64 + li) * 128 + (((2 * s + lh) ^ xb) << 4);
    // SAME padding per window of T frames: frame t of a window takes shift sh = j - pad iff -t <= sh < T - t
    int slo[4], shi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + wr * 128 + i * 32 + li, a.M - 1);
        const int t = m % a.T;
        slo[i] = -t;
        shi[i] = a.T - t;
    }
    int S_lo = max(max(slo[0], slo[1]), max(slo[2], slo[3]));
    int S_hi = min(min(shi[0], shi[1]), min(shi[2], shi[3]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        S_lo = max(S_lo, __shfl_xor(S_lo, o, 64));
        S_hi = min(S_hi, __shfl_xor(S_hi, o, 64));
    }
    S_lo = __builtin_amdgcn_readfirstlane(S_lo);
    S_hi = __builtin_amdgcn_readfirstlane(S_hi);
    const bool left_wave = wc < 2;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;

    f16x8 fa[2][4], fb[2][2];
    int a_base = 0, a_o[4];
    auto tap_setup = [&](const Walk& w) {
        const int rho = a_row0 + padA - w.pad + w.j;
        const int x = (rho >> 1) & 7;
        a_base = ((w.cs - cs0) & 1) * A_BYTES + rho * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) a_o[s] = ((2 * s + lh) ^ x) << 4;
    };
    auto load_frags = [&](int set, int s, int bbuf) {
        const char* ap = As + a_base + a_o[s];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const f16x8*>(ap + i * 4096);
        const char* bp = Bs + bbuf * B_BYTES + b_off[s];
        fb[set][0] = *reinterpret_cast<const f16x8*>(bp);
        fb[set][1] = *reinterpret_cast<const f16x8*>(bp + 4096);
    };
    // the slab after w's: (plane, cr) of K slab w.cs + 1
    auto next_slab = [&](const Walk& w, int& plane, int& cr) {
        cr = w.cr + 1; plane = w.plane;
        if (cr == nsl) { cr = 0; ++plane; }
    };

    // ---------------- prologue
    {
        float* coef = reinterpret_cast<float*>(smem + COEF_OFF);
        const int ch = tid & 255, oc = (ch < 128 ? pr.s_off0 : pr.s_off1) + (ch & 127);
        const float* src = tid < 256 ? a.col_scale : a.col_shift;
        coef[tid] = src ? src[oc] : (tid < 256 ? 1.0f : 0.0f);
    }
    Walk w1 = w0;                                      // tile n + 1
    advance(w1);
    Walk w2 = w1;                                      // tile n + 2
    advance(w2);
    stageA(w0.plane, w0.cr, 0);
    stageB(w0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ntiles > 1) stageB(w1, 1);
    if (nslab > 1) { int pl, cr; next_slab(w0, pl, cr); stageA(pl, cr, 1); }
    tap_setup(w0);
    load_frags(0, 0, 0);

    auto tile = [&](int n) {
        const int sh = w0.j - w0.pad;
        const bool need_mask = !(sh >= S_lo && sh < S_hi);
        const bool active = !dead_wave && !(left_wave && w0.j >= narrow);
        if (!active) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (n + 2 < ntiles) stageB(w2, n & 1);
            if (n + 1 < ntiles) {
                if (w1.j == 0 && w1.cs + 1 < cs1) { int pl, cr; next_slab(w1, pl, cr); stageA(pl, cr, (w1.cs + 1 - cs0) & 1); }
                tap_setup(w1);
                load_frags(0, 0, (n + 1) & 1);
            }
            return;
        }
        bool v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = sh >= slo[i] && sh < shi[i];
        const f16x8 zero = {};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            bool have_next = true;
            int nb = n & 1;
            if (s == 3) {
                // every read of tile n has been issued; retire them, publish tile n+1, recycle tile n's buffer
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                if (n + 2 < ntiles) stageB(w2, n & 1);
                have_next = n + 1 < ntiles;
                if (have_next) {
                    // first tile of a slab: bring in the slab after it (its buffer was last read a slab ago)
                    if (w1.j == 0 && w1.cs + 1 < cs1) { int pl, cr; next_slab(w1, pl, cr); stageA(pl, cr, (w1.cs + 1 - cs0) & 1); }
                    tap_setup(w1);
                }
                nb = (n + 1) & 1;
            }
            const int sn = (s + 1) & 3;
            const char* ap = As + a_base + a_o[sn];
            const char* bp = Bs + nb * B_BYTES + b_off[sn];
            if (need_mask) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[cur][i] = v[i] ? fa[cur][i] : zero;
            }
            f16x8* av = fa[cur];
            __builtin_amdgcn_s_setprio(1);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][0], av[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][1], av[0], acc[0][1], 0, 0, 0);
            if (have_next) {
                fa[nxt][0] = *reinterpret_cast<const f16x8*>(ap);
                fa[nxt][1] = *reinterpret_cast<const f16x8*>(ap + 4096);
            }
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][0], av[1], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][1], av[1], acc[1][1], 0, 0, 0);
            if (have_next) {
                fa[nxt][2] = *reinterpret_cast<const f16x8*>(ap + 2 * 4096);
                fa[nxt][3] = *reinterpret_cast<const f16x8*>(ap + 3 * 4096);
            }
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][0], av[2], acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][1], av[2], acc[2][1], 0, 0, 0);
            if (have_next) {
                fb[nxt][0] = *reinterpret_cast<const f16x8*>(bp);
                fb[nxt][1] = *reinterpret_cast<const f16x8*>(bp + 4096);
            }
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][0], av[3], acc[3][0], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][1], av[3], acc[3][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };
    for (int n = 0; n < ntiles; ++n) {
        tile(n);
        w0 = w1; w1 = w2;
        advance(w2);
    }

    __syncthreads();                                   // all fragment reads retired; no load in flight
    if (a.ksplit > 1) {
        // ---------------- split K: publish (write-through, every wave drains), THEN take a ticket.  Whoever draws the last
        // one finds every slab of the row tile published, acquires, and sums them in split order -- its own included, so
        // the result does not depend on who was last.  Nobody waits for anybody (cdna_hip_programming.md Guideline 16).
        typedef __attribute__((address_space(1))) unsigned gu32;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        gu32* const tk = (gu32*)(uintptr_t)(a.tick + rt);
        int* const tsh = reinterpret_cast<int*>(smem);
        const int voff = (wid * 32 * 64 + lane) * 16;              // + ((i*2 + c)*4 + q) * 1024
        float* const slab0 = a.ws + (size_t)rt * a.ksplit * 65536;
        {
            const uintptr_t base = (uintptr_t)(slab0 + (size_t)ks * 65536);
            const unsigned b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
            const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)b_hi << 32) | (uintptr_t)b_lo), 0, 262144, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4v v = {acc[i][c][4 * q], acc[i][c][4 * q + 1], acc[i][c][4 * q + 2], acc[i][c][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                               voff + ((i * 2 + c) * 4 + q) * 1024, 0, 16);     // aux 16 = sc1
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // EVERY storing wave drains
        __syncthreads();
        if (tid == 0) tsh[0] = (int)__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(tsh[0]);
        if (ticket != a.ksplit - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int sl = 0; sl < a.ksplit; ++sl) {
            const __attribute__((address_space(1))) char* sp =
                (const __attribute__((address_space(1))) char*)(uintptr_t)(slab0 + (size_t)sl * 65536) + voff;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4v v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4v*>(sp + ((i * 2 + c) * 4 + q) * 1024);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][c][4 * q + e] = sl == 0 ? v[e] : acc[i][c][4 * q + e] + v[e];
                    }
        }
        __syncthreads();                               // tsh[0] was read by every wave before the tile is written over it
    }

    // ---------------- epilogue: un-scale (+ shift) -> float32 half tile in LDS -> full-row stores, 128 rows at a time.
    // The weights are the MFMA's first operand, so a lane holds ONE frame (row li of the 32 x 32 tile) and, per
    // register quad q, 4 consecutive channels 8q + 4lh + {0..3}: 16-byte LDS writes.
    float rsc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rsc[i] = a.row_scale ? a.row_scale[min(m0 + wr * 128 + i * 32 + li, a.M - 1)] : 1.0f;
    const int l32 = tid & 31, hr = tid >> 5;               // 32 lanes x 16 B = one 128-channel half row
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (wpass == pass) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int chp = wc * 64 + c * 32 + 8 * q + 4 * lh;             // channel within the pair
                    const float4 sv = *reinterpret_cast<const float4*>(smem + COEF_OFF + chp * 4);
                    const float4 bv = *reinterpret_cast<const float4*>(smem + COEF_OFF + 1024 + chp * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4v o;
                        o[0] = fmaf(acc[i][c][4 * q + 0] * rsc[i], sv.x, bv.x);
                        o[1] = fmaf(acc[i][c][4 * q + 1] * rsc[i], sv.y, bv.y);
                        o[2] = fmaf(acc[i][c][4 * q + 2] * rsc[i], sv.z, bv.z);
                        o[3] = fmaf(acc[i][c][4 * q + 3] * rsc[i], sv.w, bv.w);
                        if (a.act == VC_ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.0f);
                        }
                        *reinterpret_cast<f32x4v*>(smem + (i * 32 + li) * EP + chp * 4) = o;
                    }
                }
            }
        }
        __syncthreads();
        if (a.zsplit >= 1) {
            // float atomics, one 256-byte row segment per wave instruction (64 consecutive floats: the full atomic rate;
            // a lane-strided form of the same adds ran at a quarter of it)
#pragma unroll 4
            for (int it = 0; it < 64; ++it) {
                const int flat = it * NT + tid;            // element of the [128][256] half tile
                const int row = flat >> 8, col = flat & 255, half = col >> 7;
                const int lm = rt * BM + pass * 128 + row;
                const float v = *reinterpret_cast<const float*>(smem + row * EP + col * 4);
                if (lm < (half ? pr.nrows1 : pr.nrows0))
                    unsafeAtomicAdd(a.Cout + (size_t)lm * a.ldc + (half ? pr.c_off1 : pr.c_off0) + (col & 127), v);
            }
        } else {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int h = it * 16 + hr;                    // half-row index: row = h >> 1, half = h & 1
                const int row = h >> 1, half = h & 1;
                const int lm = rt * BM + pass * 128 + row;     // output row, counted from the pair's row0
                f32x4v vv = *reinterpret_cast<const f32x4v*>(smem + row * EP + half * 512 + l32 * 16);
                if (lm < (half ? pr.nrows1 : pr.nrows0)) {
                    f32x4v* dst = reinterpret_cast<f32x4v*>(a.Cout + (size_t)lm * a.ldc + (half ? pr.c_off1 : pr.c_off0) + l32 * 4);
                    if (a.accumulate) {
                        const f32x4v old = *dst;
                        vv += old;
                        *dst = vv;
                    } else {
                        __builtin_nontemporal_store(vv, dst);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// vc_split16: float32 rows -> [hi | lo] float16 planes of x * s, s the power of two that puts the largest magnitude of
// the row's WINDOW (T rows) in [2^14, 2^15); row_scale[m] = 1 / s.  (One scale per window, not per row: a convolution
// sums taps over neighbouring rows, which must share their scale; taps never cross a window.)  Optional prologue: per-channel affine, relu, max-pool(2, 1, same)
// along time (the operand of the post-bank projection, /root/reference/modules.py:331-334).
__device__ __forceinline__ void w16_scale(unsigned bits, float& s, float& rs) {
    const float amax = __uint_as_float(bits);
    int e = (int)((bits >> 23) & 255u) - 127;
    s = 1.0f; rs = 1.0f;
    if (amax > 0.0f && e < 128) {
        e = max(e, -100);
        s = __uint_as_float((unsigned)(14 - e + 127) << 23);
        rs = __uint_as_float((unsigned)(e - 14 + 127) << 23);
    }
}

__device__ __forceinline__ float pro_val(float x, float sc, float sh, int relu) {
    float v = fmaf(x, sc, sh);
    return relu ? fmaxf(v, 0.0f) : v;
}

// Pass 1: the largest magnitude of every window after the prologue.  grid (windows, segments): a block walks its rows and
// issues ONE atomicMax on the window's word (bit patterns of non-negative floats order like the values; NaN patterns
// sort above Inf, so a NaN sticks).
__global__ void __launch_bounds__(256)
window_absmax_kernel(const float* __restrict__ X, int C, int ldx, int T, const float* __restrict__ scale,
                     const float* __restrict__ shift, int relu, int pool, unsigned* __restrict__ wmaxg) {
    __shared__ unsigned red[4];
    const int w = blockIdx.x, S = gridDim.y;
    const int r0 = (int)((long)T * blockIdx.y / S), r1 = (int)((long)T * (blockIdx.y + 1) / S);
    const bool pro = scale || shift || relu;
    float amax = 0.0f;
    bool bad = false;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) sc = *reinterpret_cast<const float4*>(scale + c);
        if (shift) sh = *reinterpret_cast<const float4*>(shift + c);
        // max-pool(2, 1, same) never creates a magnitude its inputs do not have after relu, and without relu
        // max(|a|, |b|) bounds |max(a, b)|: the pooled tensor's maximum is bounded by the un-pooled one's over the
        // window (equal whenever relu precedes the pool, the only use) -- no neighbour row needed here
        for (int r = r0; r < r1; ++r) {
            float4 x = *reinterpret_cast<const float4*>(X + ((size_t)w * T + r) * ldx + c);
            if (pro) {
                x.x = pro_val(x.x, sc.x, sh.x, relu); x.y = pro_val(x.y, sc.y, sh.y, relu);
                x.z = pro_val(x.z, sc.z, sh.z, relu); x.w = pro_val(x.w, sc.w, sh.w, relu);
            }
            bad = bad || !(x.x == x.x && x.y == x.y && x.z == x.z && x.w == x.w);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
        }
    }
    (void)pool;
    unsigned bits = bad ? 0x7fc00000u : __float_as_uint(amax);
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(wmaxg + w, max(max(red[0], red[1]), max(red[2], red[3])));
}

// Pass 2: scale by the window's power of two and split.
__global__ void __launch_bounds__(256)
split16_kernel(const float* __restrict__ X, int M, int C, int ldx, int T, const float* __restrict__ scale,
               const float* __restrict__ shift, int relu, int pool, _Float16* __restrict__ out, float* __restrict__ row_scale,
               const unsigned* __restrict__ wmaxg, int tpr_log2) {
    const int tpr = 1 << tpr_log2;                     // threads per row: 16 .. 256
    const int rpb = 256 >> tpr_log2;                   // rows per block
    const int r = threadIdx.x >> tpr_log2, t = threadIdx.x & (tpr - 1);
    const int m = blockIdx.x * rpb + r;
    const int mm = min(m, M - 1);
    const bool pooled = pool && (mm % T) != T - 1;
    const int ng = C / (4 * tpr);                      // float4 groups per thread (<= 4)
    float4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < ng) {
            const int c = (g * tpr + t) * 4;
            float4 x = *reinterpret_cast<const float4*>(X + (size_t)mm * ldx + c);
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (scale) sc = *reinterpret_cast<const float4*>(scale + c);
            if (shift) sh = *reinterpret_cast<const float4*>(shift + c);
            if (scale || shift || relu) {
                x.x = pro_val(x.x, sc.x, sh.x, relu); x.y = pro_val(x.y, sc.y, sh.y, relu);
                x.z = pro_val(x.z, sc.z, sh.z, relu); x.w = pro_val(x.w, sc.w, sh.w, relu);
            }
            if (pooled) {
                float4 y = *reinterpret_cast<const float4*>(X + (size_t)(mm + 1) * ldx + c);
                if (scale || shift || relu) {
                    y.x = pro_val(y.x, sc.x, sh.x, relu); y.y = pro_val(y.y, sc.y, sh.y, relu);
                    y.z = pro_val(y.z, sc.z, sh.z, relu); y.w = pro_val(y.w, sc.w, sh.w, relu);
                }
                x.x = fmaxf(x.x, y.x); x.y = fmaxf(x.y, y.y); x.z = fmaxf(x.z, y.z); x.w = fmaxf(x.w, y.w);
            }
            v[g] = x;
        }
    }
    // s = 2^(14 - floor(log2 amax)); all-zero windows and windows holding Inf / NaN keep s = 1 (Inf / NaN propagate)
    float s, rs;
    w16_scale(wmaxg[mm / T], s, rs);
    if (m < M) {
        if (t == 0) row_scale[m] = rs;
        _Float16* o = out + (size_t)m * 2 * C;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < ng) {
                const int c = (g * tpr + t) * 4;
                const float xs[4] = {v[g].x * s, v[g].y * s, v[g].z * s, v[g].w * s};
                f16x4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hi[i] = (_Float16)xs[i];
                    lo[i] = (_Float16)(xs[i] - (float)hi[i]);
                }
                *reinterpret_cast<f16x4*>(o + c) = hi;
                *reinterpret_cast<f16x4*>(o + C + c) = lo;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// vc_weights16: TF-layout float32 convolution kernels [k][cin][cout] -> the float16 [hi | lo] operand layouts of
// gemm16_kernel, one power-of-two scale per item GROUP (largest magnitude of the group -> [2^14, 2^15)).
__global__ void __launch_bounds__(256)
w16_absmax_kernel(const vc_w16_item* __restrict__ items, unsigned* __restrict__ gmax) {
    const vc_w16_item it = items[blockIdx.x];
    const size_t n = (size_t)it.k * it.cin * it.cout;
    float mx = 0.0f;
    for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < n; i += (size_t)gridDim.y * 256) {
        const float x = fabsf(it.src[i]);
        mx = x > mx || !(x == x) ? x : mx;             // a NaN sticks
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float y = __shfl_xor(mx, o, 64);
        mx = y > mx || !(y == y) ? y : mx;
    }
    // bit pattern order == value order for non-negative floats; NaN patterns sort above Inf
    if ((threadIdx.x & 63) == 0) atomicMax(gmax + it.group, __float_as_uint(mx));
}

__global__ void __launch_bounds__(256)
w16_split_kernel(const vc_w16_item* __restrict__ items, const unsigned* __restrict__ gmax) {
    const vc_w16_item it = items[blockIdx.x];
    float s, rs;
    w16_scale(gmax[it.group], s, rs);
    if (blockIdx.y == 0 && it.scale_dst)
        for (int i = threadIdx.x; i < it.scale_n; i += 256) it.scale_dst[i] = rs;
    const int k = it.k, cin = it.cin, cout = it.cout;
    _Float16* dst = reinterpret_cast<_Float16*>(it.dst);
    if (it.mode == 0) {
        // forward operand: row = output channel o, column = base + j * tap_stride + plane * plane_stride + c
        // (src rows r = j * cin + c, 32 x 32 tiles through LDS; cin % 32 == 0 keeps a tile inside one tap)
        __shared__ float tile[32][33];
        const int rows = k * cin, cols = cout;
        const int tr = rows / 32, tc = (cols + 31) / 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int t = blockIdx.y; t < tr * tc; t += gridDim.y) {
            const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + ty + 8 * i, c = c0 + tx;
                tile[ty + 8 * i][tx] = c < cols ? it.src[(size_t)r * cols + c] : 0.0f;
            }
            __syncthreads();
            const int j = r0 / cin, cb = r0 - j * cin;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = c0 + ty + 8 * i;
                if (o < cols) {
                    const float x = tile[tx][ty + 8 * i] * s;
                    const _Float16 hi = (_Float16)x;
                    const _Float16 lo = (_Float16)(x - (float)hi);
                    _Float16* p = dst + (size_t)o * it.row_len + it.base + (size_t)j * it.tap_stride + cb + tx;
                    p[0] = hi;
                    p[it.plane_stride] = lo;
                }
            }
            __syncthreads();
        }
    } else {
        // data-gradient operand: row = input channel c, column = base + (k - 1 - j) * tap_stride + plane * plane_stride + o
        const size_t n = (size_t)k * cin * cout;
        for (size_t idx = (size_t)blockIdx.y * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.y * 256) {
            const int o = (int)(idx % cout);
            const size_t rest = idx / cout;
            const int c = (int)(rest % cin), j = (int)(rest / cin);
            const float x = it.src[idx] * s;
            const _Float16 hi = (_Float16)x;
            const _Float16 lo = (_Float16)(x - (float)hi);
            _Float16* p = dst + (size_t)c * it.row_len + it.base + (size_t)(k - 1 - j) * it.tap_stride + o;
            p[0] = hi;
            p[it.plane_stride] = lo;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// vc_transpose_split16: the weight-gradient operands.  tf.gradients w.r.t. a conv kernel contracts over the FRAMES:
// dW[j, c, o] = sum_m X[m + j - pad, c] dY[m, o].  Both operands are therefore laid out frame-contiguous (transposed),
// the activations once per tap shift s (row (s, c) = channel c read s frames later, zero where that leaves the window),
// each row split into [hi | lo] float16 planes under ONE power-of-two scale per channel.
__global__ void __launch_bounds__(256)
col_absmax_kernel(const float* __restrict__ X, int M, int C, int ldx, const float* __restrict__ scale,
                  const float* __restrict__ shift, int relu, unsigned* __restrict__ cmax) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int rb = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rb, r1 = min(M, r0 + rb);
    const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
    const bool pro = scale || shift || relu;
    float mx = 0.0f;
    bool bad = false;
    for (int r = r0; r < r1; ++r) {
        float x = X[(size_t)r * ldx + c];
        if (pro) x = pro_val(x, sc, sh, relu);
        bad = bad || !(x == x);
        mx = fmaxf(mx, fabsf(x));
    }
    if (r1 > r0) atomicMax(cmax + c, bad ? 0x7fc00000u : __float_as_uint(mx));
}

__global__ void __launch_bounds__(256)
transpose_split16_kernel(const float* __restrict__ X, int M, int C, int ldx, int T, const float* __restrict__ scale,
                         const float* __restrict__ shift, int relu, int pool, int shift0, _Float16* __restrict__ out,
                         float* __restrict__ row_scale, const unsigned* __restrict__ cmax) {
    __shared__ _Float16 th[64][72], tl[64][72];        // [channel][frame], 144-byte rows
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64, si = blockIdx.z, s = shift0 + si;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x float4 channels, 16 rows x 4
    const int c = c0 + tx * 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) sc = *reinterpret_cast<const float4*>(scale + c);
    if (shift) sh = *reinterpret_cast<const float4*>(shift + c);
    const bool pro = scale || shift || relu;
    float cs[4], rs4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w16_scale(cmax[c + e], cs[e], rs4[e]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 16 * i, m = m0 + r;
        const int t = m % T, ts = t + s;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ts >= 0 && ts < T) {
            x = *reinterpret_cast<const float4*>(X + (size_t)(m + s) * ldx + c);
            if (pro) {
                x.x = pro_val(x.x, sc.x, sh.x, relu); x.y = pro_val(x.y, sc.y, sh.y, relu);
                x.z = pro_val(x.z, sc.z, sh.z, relu); x.w = pro_val(x.w, sc.w, sh.w, relu);
            }
            if (pool && ts != T - 1) {
                float4 y = *reinterpret_cast<const float4*>(X + (size_t)(m + s + 1) * ldx + c);
                if (pro) {
                    y.x = pro_val(y.x, sc.x, sh.x, relu); y.y = pro_val(y.y, sc.y, sh.y, relu);
                    y.z = pro_val(y.z, sc.z, sh.z, relu); y.w = pro_val(y.w, sc.w, sh.w, relu);
                }
                x.x = fmaxf(x.x, y.x); x.y = fmaxf(x.y, y.y); x.z = fmaxf(x.z, y.z); x.w = fmaxf(x.w, y.w);
            }
        }
        const float xs[4] = {x.x * cs[0], x.y * cs[1], x.z * cs[2], x.w * cs[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const _Float16 hi = (_Float16)xs[e];
            th[tx * 4 + e][r] = hi;
            tl[tx * 4 + e][r] = (_Float16)(xs[e] - (float)hi);
        }
    }
    __syncthreads();
    const int cc = threadIdx.x >> 2, q = threadIdx.x & 3;            // channel row, 16-frame segment
    _Float16* o = out + ((size_t)si * C + c0 + cc) * (2 * (size_t)M) + m0 + q * 16;
    const f16x8* ph = reinterpret_cast<const f16x8*>(&th[cc][q * 16]);
    const f16x8* pl = reinterpret_cast<const f16x8*>(&tl[cc][q * 16]);
    reinterpret_cast<f16x8*>(o)[0] = ph[0];
    reinterpret_cast<f16x8*>(o)[1] = ph[1];
    reinterpret_cast<f16x8*>(o + M)[0] = pl[0];
    reinterpret_cast<f16x8*>(o + M)[1] = pl[1];
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        float s1, r1;
        w16_scale(cmax[c0 + threadIdx.x], s1, r1);
        row_scale[(size_t)si * C + c0 + threadIdx.x] = r1;
    }
}

static size_t tick_bytes(int ntm) { return ((size_t)ntm * 4 + 255) & ~(size_t)255; }

// K slabs of one plane product walk; how many ways a single-pair launch splits them
static int choose_ksplit(int M, int n_pairs, int kslabs) {
    const int ntm = (M + BM - 1) / BM;
    if (n_pairs != 1) return 1;
    int s = 256 / ntm;                                 // ONE round of the 256 CUs
    if (s > MAX_SPLIT) s = MAX_SPLIT;
    if (s > kslabs / 4) s = kslabs / 4;                // at least 4 slabs each
    return s < 2 ? 1 : s;
}

}  // namespace

extern "C" {

size_t vc_gemm16_workspace_bytes(int32_t M, int32_t C, int32_t n_pairs) {
    if (M <= 0 || C <= 0 || (C & 63)) return 0;
    if (n_pairs != 1 || 3 * (C >> 6) < 8) return 0;
    const int ntm = (M + BM - 1) / BM;
    return tick_bytes(ntm) + (size_t)ntm * MAX_SPLIT * 262144;     // room for any split the launch may choose
}

int vc_split16(const float* d_X, int32_t M, int32_t C, int32_t ldx, int32_t T, const float* d_scale, const float* d_shift,
               int32_t relu, int32_t pool, void* d_out16, float* d_row_scale, void* stream) {
    VC_REQUIRE(d_X && d_out16 && d_row_scale, "vc_split16: NULL argument");
    VC_REQUIRE(M > 0 && T > 0 && M % T == 0 && C >= 64 && C <= 4096 && (C & 63) == 0 && ldx >= C && (ldx & 3) == 0,
               "vc_split16: bad shape M=%d T=%d C=%d ldx=%d (C: multiple of 64 up to 4096)", M, T, C, ldx);
    int tl = 4;                                        // threads per row: C / 4 up to 256, a power of two
    while ((1 << tl) < C / 4 && tl < 8) ++tl;
    while (C % (4 << tl)) --tl;                        // C = 64 * odd: fewer threads, more groups
    VC_REQUIRE(C / (4 << tl) <= 4, "vc_split16: unsupported channel count %d", C);
    const int rpb = 256 >> tl;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned* wmax = reinterpret_cast<unsigned*>(d_row_scale + M);       // scratch behind the M scales: one word per window
    VC_HIP_CHECK(hipMemsetAsync(wmax, 0, (size_t)(M / T) * 4, st));
    const int nwin = M / T;
    int seg = 1024 / nwin;
    seg = seg < 1 ? 1 : (seg > T ? T : seg);
    hipLaunchKernelGGL(window_absmax_kernel, dim3((unsigned)nwin, (unsigned)seg), dim3(256), 0, st, d_X, C, ldx, T, d_scale,
                       d_shift, relu, pool, wmax);
    hipLaunchKernelGGL(split16_kernel, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, st, d_X, M, C, ldx, T, d_scale,
                       d_shift, relu, pool, reinterpret_cast<_Float16*>(d_out16), d_row_scale, wmax, tl);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_transpose_split16(const float* d_X, int32_t M, int32_t C, int32_t ldx, int32_t T, const float* d_scale,
                         const float* d_shift, int32_t relu, int32_t pool, int32_t shift0, int32_t n_shifts, void* d_out16,
                         float* d_row_scale, void* stream) {
    VC_REQUIRE(d_X && d_out16 && d_row_scale, "vc_transpose_split16: NULL argument");
    VC_REQUIRE(M > 0 && T > 0 && M % T == 0 && (M & 63) == 0 && C >= 64 && (C & 63) == 0 && ldx >= C && (ldx & 3) == 0 &&
               n_shifts >= 1 && n_shifts <= 64 && shift0 > -T && shift0 + n_shifts - 1 < T,
               "vc_transpose_split16: bad shape M=%d T=%d C=%d ldx=%d shifts %d+%d (M, C: multiples of 64)", M, T, C, ldx, shift0, n_shifts);
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned* cmax = reinterpret_cast<unsigned*>(d_row_scale + (size_t)n_shifts * C);    // scratch behind the scales
    VC_HIP_CHECK(hipMemsetAsync(cmax, 0, (size_t)C * 4, st));
    int yb = 2048 / ((C + 255) / 256);
    yb = yb < 1 ? 1 : (yb > (M + 63) / 64 ? (M + 63) / 64 : yb);
    hipLaunchKernelGGL(col_absmax_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)yb), dim3(256), 0, st, d_X, M, C, ldx,
                       d_scale, d_shift, relu, cmax);
    hipLaunchKernelGGL(transpose_split16_kernel, dim3((unsigned)(M / 64), (unsigned)(C / 64), (unsigned)n_shifts), dim3(256), 0,
                       st, d_X, M, C, ldx, T, d_scale, d_shift, relu, pool, shift0, reinterpret_cast<_Float16*>(d_out16),
                       d_row_scale, cmax);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_weights16(const vc_w16_item* d_items, int32_t n_items, uint32_t* d_gmax, int32_t n_groups, void* stream) {
    VC_REQUIRE(d_items && d_gmax && n_items > 0 && n_items <= 65535 && n_groups > 0, "vc_weights16: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    VC_HIP_CHECK(hipMemsetAsync(d_gmax, 0, (size_t)n_groups * 4, st));
    hipLaunchKernelGGL(w16_absmax_kernel, dim3((unsigned)n_items, 16), dim3(256), 0, st, d_items, d_gmax);
    hipLaunchKernelGGL(w16_split_kernel, dim3((unsigned)n_items, 16), dim3(256), 0, st, d_items, d_gmax);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_gemm16(const vc_gemm16_desc* d, void* stream) {
    VC_REQUIRE(d && d->d_X16 && d->d_C, "vc_gemm16: NULL argument");
    VC_REQUIRE(d->M > 0 && d->T > 0 && d->M % d->T == 0 && d->C >= 64 && (d->C & 63) == 0 && d->C <= 16384 && d->ldx >= 2 * d->C &&
               (d->ldx & 7) == 0, "vc_gemm16: bad shape M=%d T=%d C=%d ldx=%d", d->M, d->T, d->C, d->ldx);
    VC_REQUIRE(d->n_pairs >= 1 && d->n_pairs <= 16 && (d->ldc & 3) == 0, "vc_gemm16: bad pair count / ldc");
    const bool atomic = d->atomic_splits >= 1;
    VC_REQUIRE(d->act == VC_ACT_NONE || (d->act == VC_ACT_RELU && !atomic), "vc_gemm16: act must be none or relu (none with atomic_splits)");
    VC_REQUIRE(d->atomic_splits >= 0 && d->atomic_splits <= MAX_SPLIT && !(atomic && (d->d_col_shift || d->ragged)),
               "vc_gemm16: atomic_splits 0..8, without col_shift / ragged");
    VC_REQUIRE((reinterpret_cast<uintptr_t>(d->d_X16) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->d_C) & 15) == 0, "vc_gemm16: unaligned");
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    G16Args a{};
    a.X = d->d_X16; a.row_scale = d->d_row_scale; a.M = d->M; a.T = d->T; a.C = d->C; a.ldx = d->ldx;
    a.col_scale = d->d_col_scale; a.col_shift = d->d_col_shift; a.Cout = d->d_C; a.ldc = d->ldc; a.accumulate = d->accumulate;
    a.n_pairs = d->n_pairs; a.ragged = d->ragged; a.act = d->act;
    const int nsl = d->C >> 6;
    if (d->ragged) {
        // the bank data gradient: 128 input channels per bank, bank k = 1 .. C/128 with k taps, left padding k / 2
        VC_REQUIRE(d->n_pairs == 1 && (d->C & 127) == 0 && d->C / 128 <= 32, "vc_gemm16: ragged walk needs one pair and C = 128 * banks <= 4096");
    }
    for (int i = 0; i < d->n_pairs; ++i) {
        const vc_gemm16_pair& s = d->pairs[i];
        G16Pair& p = a.p[i];
        VC_REQUIRE(s.d_Bt0 && s.d_Bt1 && (reinterpret_cast<uintptr_t>(s.d_Bt0) & 15) == 0 && (reinterpret_cast<uintptr_t>(s.d_Bt1) & 15) == 0,
                   "vc_gemm16: pair %d: NULL / unaligned weights", i);
        // (atomic accumulation: c_off* are element offsets of each filter's [rows, ldc] block from d_C, any alignment)
        VC_REQUIRE(s.c_off0 >= 0 && s.c_off1 >= 0 && (atomic || ((s.c_off0 & 3) == 0 && (s.c_off1 & 3) == 0 && s.c_off0 + 128 <= d->ldc &&
                   s.c_off1 + 128 <= d->ldc)), "vc_gemm16: pair %d: bad output columns", i);
        VC_REQUIRE(s.row0 >= 0 && s.nrows0 >= 0 && s.nrows1 >= -1 && s.row0 + (s.nrows0 > s.nrows1 ? s.nrows0 : s.nrows1) <= d->M,
                   "vc_gemm16: pair %d: bad row range", i);
        p.Bt0 = s.d_Bt0; p.Bt1 = s.d_Bt1; p.c_off0 = s.c_off0; p.c_off1 = s.c_off1;
        p.row0 = s.row0;
        p.s_off0 = atomic ? s.s_off0 : s.c_off0;
        p.s_off1 = atomic ? s.s_off1 : s.c_off1;
        p.nrows0 = s.nrows0 ? s.nrows0 : d->M - s.row0;
        p.nrows1 = s.nrows1 < 0 ? 0 : (s.nrows1 ? s.nrows1 : d->M - s.row0);      // -1: a single 128-column filter
        if (d->ragged) {
            p.taps0 = 0; p.extra = 0; p.pad_l = 0;
            p.K0 = p.K1 = 2 * ((nsl >> 1) * ((nsl >> 1) + 1) / 2) * 128;
        } else {
            VC_REQUIRE(s.taps0 >= 1 && (s.extra == 0 || s.extra == 1) && s.taps0 + s.extra <= 33 && s.pad_l >= 0 &&
                       s.pad_l < s.taps0 + s.extra, "vc_gemm16: pair %d: bad taps %d+%d / pad %d", i, s.taps0, s.extra, s.pad_l);
            p.taps0 = s.taps0; p.extra = s.extra; p.pad_l = s.pad_l;
            p.K0 = s.taps0 * 2 * d->C;
            p.K1 = (s.taps0 + s.extra) * 2 * d->C;
        }
    }
    int max_rows = 0;
    for (int i = 0; i < d->n_pairs; ++i) {
        const int r = a.p[i].nrows0 > a.p[i].nrows1 ? a.p[i].nrows0 : a.p[i].nrows1;
        max_rows = r > max_rows ? r : max_rows;
    }
    const int ntm = (max_rows + BM - 1) / BM;
    const int kslabs = 3 * nsl;
    if (atomic) {
        // every (row tile, pair, K range) its own workgroup, partial tiles added with float atomics to a pre-initialised C
        const int zs = d->atomic_splits;
        VC_REQUIRE(zs <= kslabs, "vc_gemm16: more K ranges than K slabs");
        a.zsplit = zs;
        a.ksplit = 1;
        for (int i = 0; i <= zs; ++i) a.split_cs[i] = (int16_t)((long)kslabs * i / zs);
        hipLaunchKernelGGL(gemm16_kernel, dim3(ntm, d->n_pairs, zs), dim3(NT), LDS_BYTES, st, a);
        VC_HIP_CHECK(hipGetLastError());
        return VC_OK;
    }
    int ks = choose_ksplit(max_rows, d->n_pairs, kslabs);
    int split_map = 0;
    const int forced = vc::opt(vc::OPT_GEMM16_SPLIT);
    if (forced >= 1 && d->n_pairs == 1) {
        const int want = forced & 15;
        if (want >= 1 && want <= MAX_SPLIT && want * 2 <= kslabs) {
            ks = want;
            split_map = (forced >> 4) == 1 && (8 % want) == 0 && want > 1;
        }
    }
    if (ks > 1 && (!d->d_workspace || d->workspace_bytes < tick_bytes(ntm) + (size_t)ntm * ks * 262144)) ks = 1;
    a.ksplit = ks;
    a.split_map = split_map;
    if (ks > 1) {
        VC_REQUIRE((reinterpret_cast<uintptr_t>(d->d_workspace) & 255) == 0, "vc_gemm16: workspace must be 256-byte aligned");
        // split points balanced by K tiles (ragged: slab cr of every plane has (cr >> 1) + 1 taps)
        long total = 0;
        for (int c = 0; c < kslabs; ++c) total += d->ragged ? ((c % nsl) >> 1) + 1 : 1;
        long run = 0;
        int nxt = 1;
        a.split_cs[0] = 0;
        for (int c = 0; c < kslabs && nxt < ks; ++c) {
            run += d->ragged ? ((c % nsl) >> 1) + 1 : 1;
            if (run * ks >= total * nxt) a.split_cs[nxt++] = (int16_t)(c + 1);
        }
        for (; nxt <= ks; ++nxt) a.split_cs[nxt] = (int16_t)kslabs;
        a.split_cs[ks] = (int16_t)kslabs;
        for (int i = 0; i < ks; ++i)
            VC_REQUIRE(a.split_cs[i + 1] > a.split_cs[i], "vc_gemm16: empty K split (internal)");
        a.tick = reinterpret_cast<unsigned*>(d->d_workspace);
        a.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->d_workspace) + tick_bytes(ntm));
        VC_HIP_CHECK(hipMemsetAsync(a.tick, 0, tick_bytes(ntm), st));
        const unsigned nblk = split_map ? 8u * (unsigned)((ntm + 8 / ks - 1) / (8 / ks)) : (unsigned)(ntm * ks);
        hipLaunchKernelGGL(gemm16_kernel, dim3(nblk), dim3(NT), LDS_BYTES, st, a);
        VC_HIP_CHECK(hipGetLastError());
        return VC_OK;
    }
    a.split_cs[0] = 0;
    a.split_cs[1] = (int16_t)kslabs;
    if (d->n_pairs == 16 && ntm < 32000) {
        // pairs split over two XCDs, every XCD halves of four pairs: one heavy, two medium, one light (vc_launch_bank256)
        a.xcd_tiles = ntm;
        int max_slots = 0;
        const int h0 = (ntm + 1) / 2;
        for (int x = 0; x < 8; ++x) {
            const int s4 = x >> 1, half = x & 1;
            const int ranks[4] = {s4, 7 - s4, 8 + s4, 15 - s4};
            int slots = 0;
            for (int sg = 0; sg < 4; ++sg) {
                a.seg_pair[x][sg] = (int16_t)(15 - ranks[sg]);
                a.seg_first[x][sg] = (int16_t)(half ? h0 : 0);
                a.seg_count[x][sg] = (int16_t)(half ? ntm - h0 : h0);
                slots += a.seg_count[x][sg];
            }
            max_slots = slots > max_slots ? slots : max_slots;
        }
        hipLaunchKernelGGL(gemm16_kernel, dim3((unsigned)(8 * max_slots)), dim3(NT), LDS_BYTES, st, a);
    } else {
        hipLaunchKernelGGL(gemm16_kernel, dim3(ntm, d->n_pairs), dim3(NT), LDS_BYTES, st, a);
    }
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

}  // extern "C"
