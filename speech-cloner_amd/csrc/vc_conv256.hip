// conv1d / dense on 128 x (128|256) tiles with the deep pipeline of vc_bank256.hip, for the
// single-filter bf16 launches whose K is long enough to pay for it: the CBHG projections
// (/root/reference/modules.py:331-340: max_pooling1d(2, 1, same) -> conv1d k=3 -> bn -> relu,
// conv1d k=3 -> bn -> + residual), whose K is 3 x 4096 / 3 x 512.
//
// Same structure as bank256_kernel: operands by global_load_lds into XOR-swizzled 128-byte-row LDS
// tiles, activation tile (128 + taps - 1 [+1] rows of one 64-channel slab) resident across the taps
// and double-buffered across slabs, weight tile per (slab, tap) double-buffered one tile ahead, one
// barrier per tile before its last k-step, weights as the first MFMA operand (lane = frame), the
// output tile through LDS into whole-row stores.  Differences:
//   * one filter, N a multiple of 128: 2 x WN waves (WN = 2: 128 columns, 4 waves, two blocks per
//     CU; WN = 4: 256 columns, 8 waves), 64 x 64 accumulators per wave;
//   * POOL: tf.layers.max_pooling1d(pool_size=2, strides=1, padding='same') of the operand is taken
//     on the fragments: max(row, row + 1) with the window's last frame pooling with itself.  The
//     operand is post-ReLU (>= 0), so the bf16 maximum is the unsigned 16-bit maximum
//     (v_pk_max_u16), the same trick conv_kernel uses while staging;
//   * residual add and any activation in the epilogue.
#include "vc_common.h"
#include "vc_conv256.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// rows per block = 64 * WM (WM row groups of waves); the activation slab holds 8 more rows: taps - 1 <= 6 and 1 pool row
constexpr int a_rows(int wm) { return 64 * wm + 8; }

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 max_nonneg(bf16x8 x, bf16x8 y) {
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(u16x8, x), __builtin_bit_cast(u16x8, y)));
}

__device__ __forceinline__ float act_fn(float v, int act) {
    switch (act) {
        case VC_ACT_RELU: return fmaxf(v, 0.0f);
        case VC_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case VC_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// counted wait: everything but the newest `n` LDS-direct loads of this wave has landed
__device__ __forceinline__ void wait_loads_but(int n) {
    if (n >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <int WN, bool POOL, int NBUF, int WM>
__global__ void __launch_bounds__(64 * WM * WN, 1)
conv256_kernel(Conv256Args a) {
    constexpr int BM = 64 * WM, A_ROWS = a_rows(WM), A_BYTES = A_ROWS * 128;
    constexpr int BN = 64 * WN, NWAVE = WM * WN, NTHR = 64 * NWAVE;
    constexpr int BQ = (BN / 8) / NWAVE;               // weight-tile loads per wave: 4 (two row groups) or 2 (four)
    static_assert(BQ * NWAVE * 8 == BN && (BQ == 2 || BQ == 4), "weight tile split");
    constexpr int DIST = NBUF - 1;                     // weight tiles in flight ahead of the one being read
    constexpr int B_BYTES = BN * 128;
    constexpr int AQ = (A_ROWS / 8 + NWAVE - 1) / NWAVE;        // staging passes over the slab's row blocks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;                             // [2][136][128]
    char* const Bs = smem + 2 * A_BYTES;               // [NBUF][BN][128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;            // wave tile: rows wr*64.., cols wc*64..
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int ntap = a.taps, nslab = a.Cin >> 6, ntiles = nslab * ntap, pad_l = a.pad_l;
    const __bf16* X = reinterpret_cast<const __bf16*>(a.X);
    const __bf16* Bt = reinterpret_cast<const __bf16*>(a.Bt);

    // ---------------- staging roles
    const int srow = lane >> 3, pslot = lane & 7;
    const __bf16* a_src[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        const int rho = (q * NWAVE + wid) * 8 + srow;
        const int g = min(max(m0 - pad_l + rho, 0), a.M - 1);
        a_src[q] = X + (size_t)g * a.ldx + (pslot ^ ((rho >> 1) & 7)) * 8;
    }
    const int a_rows_needed = BM + ntap - 1 + (POOL ? 1 : 0);
    const __bf16* b_src[BQ];
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
        const int n = (q * NWAVE + wid) * 8 + srow;
        b_src[q] = Bt + (size_t)min(n0 + n, a.N - 1) * a.K + (pslot ^ ((n >> 1) & 7)) * 8;
    }
    auto stageA = [&](int cs, int buf) {
        char* dst = As + buf * A_BYTES + wid * 1024;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int rb = q * NWAVE + wid;
            if (rb * 8 < a_rows_needed && rb < A_ROWS / 8) glds16(a_src[q] + cs * 64, dst + q * NWAVE * 1024);
        }
    };
    auto stageB = [&](int n, int buf) {
        const int cs = n / ntap, j = n - cs * ntap;
        const int koff = j * a.Cin + cs * 64;
        char* dst = Bs + buf * B_BYTES + wid * 1024;
#pragma unroll
        for (int q = 0; q < BQ; ++q) glds16(b_src[q] + koff, dst + q * NWAVE * 1024);
    };

    // ---------------- MFMA roles
    const int li = lane & 31, lh = lane >> 5;
    const int a_row0 = wr * 64 + li;
    const int xb = (li >> 1) & 7;
    int b_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b_off[s] = (wc * 64 + li) * 128 + (((2 * s + lh) ^ xb) << 4);
    int jlo[2], jhi[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min(m0 + wr * 64 + i * 32 + li, a.M - 1);
        const int t = m % a.T;
        jlo[i] = max(0, pad_l - t);                    // taps [jlo, jhi) read a real frame
        jhi[i] = a.T - t + pad_l;                      // ... and taps < jhi - 1 a real NEXT frame (pool partner)
    }
    int J_lo = max(jlo[0], jlo[1]);
    int J_hi = min(jhi[0], jhi[1]) - (POOL ? 1 : 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        J_lo = max(J_lo, __shfl_xor(J_lo, o, 64));
        J_hi = min(J_hi, __shfl_xor(J_hi, o, 64));
    }
    J_lo = __builtin_amdgcn_readfirstlane(J_lo);
    J_hi = __builtin_amdgcn_readfirstlane(J_hi);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;

    bf16x8 fa[2][2], fp[2][2], fb[2][2];
    int a_base = 0, a_o[4], p_base = 0, p_o[4];
    int j = 0;
    auto tap_setup = [&](int n) {
        const int cs = n / ntap, jj = n - cs * ntap;
        const int rho = a_row0 + jj;
        a_base = (cs & 1) * A_BYTES + rho * 128;
        p_base = a_base + 128;
        const int x = (rho >> 1) & 7, xp = ((rho + 1) >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a_o[s] = ((2 * s + lh) ^ x) << 4;
            p_o[s] = ((2 * s + lh) ^ xp) << 4;
        }
        return jj;
    };

    // ---------------- prologue: tiles 0 .. DIST-1 in flight, tile 0 (and slab 0) awaited.
    // Every wave issues exactly BQ (4 or 2) loads per weight tile and any activation-slab loads BEFORE the
    // weight tile of the same section, so "all but the newest BQ*k loads" always means "all but the
    // newest k weight tiles": the counted s_waitcnt below needs no other bookkeeping.  The barrier
    // is the raw s_barrier: __syncthreads() would drain the loads that are meant to stay in flight.
    // With ONE tap per slab the activation slab of tile n+2 is requested in tile n's section, right before weight
    // tile n+1+DIST, and is needed one section later -- it would still be among the "newest 2 weight tiles" worth
    // of loads.  Only the newest weight tile may then stay in flight across a barrier.  (taps >= 2: a slab is
    // requested when the previous one is entered and has aged past the count by the time it is read.)
    // INVARIANT of every counted wait below: the loads allowed to stay in flight were all ISSUED AFTER the newest
    // operand the next tile reads.  In the steady state that is `keep` weight tiles; at the tail the weight tiles
    // run out first, so the count is clamped by the tiles that were really requested after that operand:
    //   taps >= 2: after weight tile n+1 come tiles n+2 .. min(n+DIST, ntiles-1)        -> ntiles - 2 - n
    //              (a slab is followed by >= min(taps, tiles left) weight tiles before it is read; taps >= DIST - 1)
    //   taps == 1: after slab n+1 (requested in section n-1) comes only weight tile n+DIST -> ntiles - DIST - n,
    //              i.e. vmcnt(0) from n = ntiles - DIST on (the slab's own loads are the newest ones there).
    const int keep = ntap == 1 ? DIST - 2 : DIST - 1;
    const int tail0 = ntap == 1 ? ntiles - DIST : ntiles - 2;          // in-flight tiles allowed at section n: tail0 - n
    static_assert(DIST == 2 || DIST == 3, "the taps >= DIST - 1 argument above assumes 3 or 4 weight buffers");
    stageA(0, 0);
    stageB(0, 0);
#pragma unroll
    for (int d = 1; d < DIST; ++d)
        if (d < ntiles) stageB(d, d);
    wait_loads_but(BQ * min(DIST - 1, ntiles - 1));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nslab > 1) stageA(1, 1);
    if (DIST < ntiles) stageB(DIST, DIST);
    int bcur = 0;                                       // buffer of the tile being read
    j = tap_setup(0);
    {
        const char* ap = As + a_base + a_o[0];
        fa[0][0] = *reinterpret_cast<const bf16x8*>(ap);
        fa[0][1] = *reinterpret_cast<const bf16x8*>(ap + 4096);
        if constexpr (POOL) {
            const char* pp = As + p_base + p_o[0];
            fp[0][0] = *reinterpret_cast<const bf16x8*>(pp);
            fp[0][1] = *reinterpret_cast<const bf16x8*>(pp + 4096);
        }
        const char* bp = Bs + b_off[0];
        fb[0][0] = *reinterpret_cast<const bf16x8*>(bp);
        fb[0][1] = *reinterpret_cast<const bf16x8*>(bp + 4096);
    }

    for (int n = 0; n < ntiles; ++n) {
        const bool need_mask = !(j >= J_lo && j < J_hi);        // wave-uniform
        bool v[2], hn[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[i] = j >= jlo[i] && j < jhi[i];
            hn[i] = j < jhi[i] - 1;
        }
        const bf16x8 zero = {};
        int jn = j;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            bool have_next = true;
            int nb = bcur;
            if (s == 3) {
                // every read of tile n is issued: retire them, publish tile n+1 (tiles n+2 .. n+DIST
                // stay in flight), recycle tile n's buffer for tile n+1+DIST
                wait_loads_but(BQ * max(0, min(keep, tail0 - n)));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                have_next = n + 1 < ntiles;
                if (have_next) {
                    const int cs1 = (n + 1) / ntap;
                    if ((n + 1) - cs1 * ntap == 0 && cs1 + 1 < nslab) stageA(cs1 + 1, (cs1 + 1) & 1);
                    jn = tap_setup(n + 1);
                }
                if (n + 1 + DIST < ntiles) stageB(n + 1 + DIST, bcur);
                nb = bcur + 1 == NBUF ? 0 : bcur + 1;
            }
            const int sn = (s + 1) & 3;
            const char* ap = As + a_base + a_o[sn];
            const char* pp = As + p_base + p_o[sn];
            const char* bp = Bs + nb * B_BYTES + b_off[sn];
            bf16x8 av[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (POOL) av[i] = max_nonneg(fa[cur][i], fp[cur][i]);
                else av[i] = fa[cur][i];
            }
            if (need_mask) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (POOL) av[i] = hn[i] ? av[i] : fa[cur][i];   // last frame pools with itself
                    av[i] = v[i] ? av[i] : zero;                               // SAME padding
                }
            }
            __builtin_amdgcn_s_setprio(1);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][0], av[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][1], av[0], acc[0][1], 0, 0, 0);
            if (have_next) {
                fa[nxt][0] = *reinterpret_cast<const bf16x8*>(ap);
                fa[nxt][1] = *reinterpret_cast<const bf16x8*>(ap + 4096);
                if constexpr (POOL) {
                    fp[nxt][0] = *reinterpret_cast<const bf16x8*>(pp);
                    fp[nxt][1] = *reinterpret_cast<const bf16x8*>(pp + 4096);
                }
                fb[nxt][0] = *reinterpret_cast<const bf16x8*>(bp);
                fb[nxt][1] = *reinterpret_cast<const bf16x8*>(bp + 4096);
            }
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][0], av[1], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][1], av[1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        j = jn;
        bcur = bcur + 1 == NBUF ? 0 : bcur + 1;
    }

    // ---------------- epilogue: scale/shift, activation, residual -> bf16 tile in LDS -> row stores
    constexpr int EP = BN * 2 + 16;
    __syncthreads();                                   // (no load is in flight any more: the last wait was vmcnt(0))
    {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wc * 64 + c * 32 + 8 * q + 4 * lh;       // column of the tile
                const int gn = min(n0 + col, a.N - 4);
                float4 sv = make_float4(1.0f, 1.0f, 1.0f, 1.0f), bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (a.epi_scale) sv = *reinterpret_cast<const float4*>(a.epi_scale + a.c_off + gn);
                if (a.epi_shift) bv = *reinterpret_cast<const float4*>(a.epi_shift + a.c_off + gn);
                const float svv[4] = {sv.x, sv.y, sv.z, sv.w}, bvv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wr * 64 + i * 32 + li;
                    bf16x4 rr = {};
                    if (a.R) {
                        const int gm = min(m0 + row, a.M - 1);
                        rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(a.R) + (size_t)gm * a.ldr + gn);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float val = act_fn(acc[i][c][4 * q + e] * svv[e] + bvv[e], a.act);
                        if (a.R) val += (float)rr[e];
                        o[e] = (__bf16)val;
                    }
                    *reinterpret_cast<bf16x4*>(smem + row * EP + col * 2) = o;
                }
            }
        }
    }
    __syncthreads();
    {
        constexpr int CPR = BN / 8;                        // 16-byte chunks per tile row
        __bf16* C = reinterpret_cast<__bf16*>(a.C);
        for (int idx = tid; idx < BM * CPR; idx += NTHR) {
            const int row = idx / CPR, ch = idx - row * CPR;
            const int gm = m0 + row, gn = n0 + ch * 8;
            if (gm < a.M && gn < a.N)
                *reinterpret_cast<bf16x8*>(C + (size_t)gm * a.ldc + a.c_off + gn) =
                    *reinterpret_cast<const bf16x8*>(smem + row * EP + ch * 16);
        }
    }
}

template <int WN, bool POOL, int WM> int launch(const Conv256Args& a, hipStream_t st) {
    constexpr int NBUF = WN == 4 ? 3 : 4;              // 32 KB / 16 KB weight tiles
    constexpr int BM = 64 * WM;
    constexpr int LDS = 2 * a_rows(WM) * 128 + NBUF * 64 * WN * 128;
    static_assert(LDS <= 160 * 1024 && BM * (64 * WN * 2 + 16) <= LDS, "LDS budget (K loop, epilogue tile)");
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv256_kernel<WN, POOL, NBUF, WM>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done = true;
    }
    hipLaunchKernelGGL((conv256_kernel<WN, POOL, NBUF, WM>), dim3((a.M + BM - 1) / BM, a.N / (64 * WN)), dim3(64 * WM * WN), LDS, st, a);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

}  // namespace

int vc_launch_conv256(const Conv256Args& a, hipStream_t st) {
    const bool wide = (a.N % 256) == 0;
    // 128-column tiles: 256 rows per block (eight waves, half the weight traffic per frame) once there are enough
    // rows; a launch's CU time, not its block count, is what it costs with several batches in flight.
    // vc_set_option("conv256_wm", 2) keeps the 128-row blocks (A/B).
    const bool tall = !wide && a.M >= 2048 && vc::opt(vc::OPT_CONV256_WM) != 2;
    if (a.pool) return wide ? launch<4, true, 2>(a, st) : (tall ? launch<2, true, 4>(a, st) : launch<2, true, 2>(a, st));
    return wide ? launch<4, false, 2>(a, st) : (tall ? launch<2, false, 4>(a, st) : launch<2, false, 2>(a, st));
}
