// conv1d_banks on 256 x 256 tiles, 8 waves, bf16 MFMA -- the deep-pipelined form of conv_kernel
// (vc_gemm.hip) for the dominant launch of the path: the decoder's K-wide filter bank
// (/root/reference/modules.py:144-166: K convolutions of width 1..K over the same input,
// concatenated, batch-normalised, ReLU).
//
// Why a second kernel: the 128 x 128 / 4-wave structure reads one ds_read_b128 per MFMA, which
// saturates the CU's LDS port at ~41 % of the matrix peak.  Here every wave
// owns a 128 x 64 accumulator (128 VGPRs), so 24 fragment reads feed 32 MFMAs (LDS port 75 % busy
// at full MFMA rate), and operands reach LDS by global_load_lds (no staging registers).
//
// Work decomposition: filter widths come in pairs (2p+1, 2p+2) that share TF's SAME left padding
// (p), so tap j of both reads the same shifted activation rows; a block computes 256 frames x
// (128 + 128) output channels of one pair.  K loop = channel slab (64) outer, tap inner:
//   A  activation tile of 256 + 32 rows of one slab, resident in LDS for all taps (double
//      buffered across slabs); tap j's fragments are the same image read j rows further down,
//   B  per (slab, tap) a 256 x 64 weight tile (left half = narrower filter; absent at its
//      missing last tap), double buffered, fetched one full tile ahead,
//   one barrier per tile, placed before the LAST k-step of a tile: by then every wave has retired
//      its reads of the tile (so its buffer may be refilled) and the next tile's loads, issued one
//      tile earlier, are waited for with vmcnt(0) -- nobody waits at the tile boundary itself.
// LDS rows are 128 B; the 16-byte slot of row r is XORed with (r >> 1) & 7, applied on the SOURCE
// address of the LDS-direct loads (the LDS image of one wave instruction is linear), so the 16
// lanes of a ds_read_b128 group hit 16 different bank groups.  SAME-padding zeros depend on
// (output frame, tap) and are a per-lane select on the A fragment.
#include <cstdlib>
#include "vc_common.h"
#include "vc_bank256.h"

namespace {

// Timing-only ablations (wrong results) are compiled in with -DVC_ABLATE alone (tools/build_ablate.sh); the shipped
// library has no such path: ABL() is the constant false.
#ifdef VC_ABLATE
#define ABL(mask) ((a.dbg & (mask)) != 0)
#else
#define ABL(mask) false
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NT = 512;
constexpr int BM = 256;
constexpr int A_ROWS = 288;                        // 256 + 32 halo rows
constexpr int A_BYTES = A_ROWS * 128;
constexpr int B_BYTES = 256 * 128;
constexpr int COEF_OFF = 2 * A_BYTES + 2 * B_BYTES;    // 139,264: scale[256] | shift[256] of the pair (f32)
constexpr int LDS_BYTES = COEF_OFF + 2 * 256 * 4;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(uintptr_t)l, 16, 0, 0);
}

__global__ void __launch_bounds__(NT, 1)
bank256_kernel(Bank256Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const As = smem;                             // [2][288][128]
    char* const Bs = smem + 2 * A_BYTES;               // [2][256][128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform
    const int wr = wid >> 2, wc = wid & 3;             // wave tile: rows wr*128.., cols wc*64.. (wc < 2: the narrower filter)
    // Block -> (pair, row tile).  XCD-aware form: workgroup ids go round-robin to the 8 XCDs, so id & 7
    // names the XCD; the launcher gives every XCD a short list of (pair, row-tile range) segments so that a
    // pair's weight stream (up to 4 MB) is fetched into one or two L2s instead of eight (see vc_launch_bank256).
    int psel, rt, ks = 0;
    if (a.ksplit > 1) {
        // split K: ids b and b + 8 share an XCD (round-robin placement: speed only), so the ksplit workgroups of a row
        // tile are consecutive multiples of 8 apart -- dispatched together, finishing together, exchanging through one L2
        const int q = blockIdx.x >> 3;
        ks = q % a.ksplit;
        rt = (q / a.ksplit) * 8 + (blockIdx.x & 7);
        psel = 0;
        if (rt * BM >= a.M) return;
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
    }
    const Bank256Pair pr = a.p[psel];
    const int m0 = rt * (a.pool ? BM - 1 : BM);            // pooled output: tiles overlap by one frame
    const int ntap = pr.taps0 + pr.extra;              // taps of the wider filter
    const int nslab_all = a.Cin >> 6;
    const int cs0 = ks * nslab_all / a.ksplit;         // this workgroup's channel slabs [cs0, cs0 + nslab)
    const int nslab = (ks + 1) * nslab_all / a.ksplit - cs0;
    const int ntiles = nslab * ntap;
    const int pad_l = pr.pad_l;
    const __bf16* X = reinterpret_cast<const __bf16*>(a.X);

    // ---------------- staging roles (LDS-direct loads; one wave instruction = 8 rows = 1 KB)
    const int srow = lane >> 3;                        // row within the 8-row block
    const int pslot = lane & 7;                        // physical 16-byte slot
    // A: row block rb = q*8 + wid (q = 0..4; q = 4 only for waves 0..3), LDS row rho = rb*8 + srow
    const __bf16* a_src[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int rho = (q * 8 + wid) * 8 + srow;
        const int g = min(max(m0 - pad_l + rho, 0), a.M - 1);
        const int slot = pslot ^ ((rho >> 1) & 7);
        a_src[q] = X + (size_t)g * a.ldx + slot * 8 + cs0 * 64;
    }
    const int a_rows_needed = BM + ntap - 1;           // rows >= this are never read
    // B: row block rb = q*8 + wid (q = 0..3), row n = rb*8 + srow; n < 128 -> narrower filter
    const __bf16* b_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = (q * 8 + wid) * 8 + srow;
        const int slot = pslot ^ ((n >> 1) & 7);
        const bool left = n < 128;
        const __bf16* Bt = reinterpret_cast<const __bf16*>(left ? pr.Bt0 : pr.Bt1);
        const int K = (left ? pr.taps0 : ntap) * a.Cin;
        b_src[q] = Bt + (size_t)(n & 127) * K + slot * 8 + cs0 * 64;
    }
    auto stageA = [&](int cs, int buf) {
        char* dst = As + buf * A_BYTES + wid * 1024;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int rb = q * 8 + wid;
            if (rb * 8 < a_rows_needed && rb < A_ROWS / 8) glds16(a_src[q] + cs * 64, dst + q * 8192);
        }
    };
    auto stageB = [&](int n, int buf) {
        const int cs = n / ntap, j = n - cs * ntap;
        // At the wider filter's extra tap the narrower one has no weights: its half of the tile is not
        // fetched and its waves skip the tile's products (tile(), `active`).  (Multiplying zeroed activations
        // with whatever the buffer held there turned stale Inf/NaN bytes into NaN once; and the skipped
        // products are worth 1.5-1.9 % of the launch: profiles/r02/ab_bank_skip_extra_tap.log.)
        const int koffR = j * a.Cin + cs * 64;
        const int koffL = min(j, pr.taps0 - 1) * a.Cin + cs * 64;
        char* dst = Bs + buf * B_BYTES + wid * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q >= 2 || j < pr.taps0) glds16(b_src[q] + (q < 2 ? koffL : koffR), dst + q * 8192);
    };

    // ---------------- MFMA roles
    const int li = lane & 31, lh = lane >> 5;
    // A fragment (row tile i, k-step s, tap j): LDS row rho = wr*128 + i*32 + li + j, slot (2s+lh) ^ ((rho>>1)&7)
    const int a_row0 = wr * 128 + li;
    // B fragment (col tile c, k-step s): row n = wc*64 + c*32 + li; (n>>1)&7 = (li>>1)&7
    const int xb = (li >> 1) & 7;
    int b_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b_off[s] = (wc * 64 + li) * 128 + (((2 * s + lh) ^ xb) << 4);
    int jlo[4], jhi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + wr * 128 + i * 32 + li, a.M - 1);
        const int t = m % a.T;
        jlo[i] = max(0, pad_l - t);
        jhi[i] = a.T - t + pad_l;
    }
    const bool left_wave = wc < 2;
    // taps [J_lo, J_hi) need no select anywhere in this wave (wave-uniform: scalar branch per tile)
    int J_lo = max(max(jlo[0], jlo[1]), max(jlo[2], jlo[3]));
    int J_hi = min(min(jhi[0], jhi[1]), min(jhi[2], jhi[3]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        J_lo = max(J_lo, __shfl_xor(J_lo, o, 64));
        J_hi = min(J_hi, __shfl_xor(J_hi, o, 64));
    }
    J_lo = __builtin_amdgcn_readfirstlane(J_lo);
    J_hi = __builtin_amdgcn_readfirstlane(J_hi);
    if (left_wave) J_hi = min(J_hi, pr.taps0);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;

    bf16x8 fa[2][4], fb[2][2];
    int a_base = 0, a_o[4];                             // per-tap A addressing
    auto tap_setup = [&](int n) {
        const int cs = n / ntap, j = n - cs * ntap;
        const int rho = a_row0 + j;
        const int x = (rho >> 1) & 7;
        a_base = (cs & 1) * A_BYTES + rho * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) a_o[s] = ((2 * s + lh) ^ x) << 4;
        return j;
    };
    auto load_frags = [&](int set, int s, int bbuf) {
        const char* ap = As + a_base + a_o[s];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(ap + i * 4096);
        const char* bp = Bs + bbuf * B_BYTES + b_off[s];
        fb[set][0] = *reinterpret_cast<const bf16x8*>(bp);
        fb[set][1] = *reinterpret_cast<const bf16x8*>(bp + 4096);
    };

    // ---------------- prologue
    {   // BatchNorm scale / shift of the pair's 256 channels -> LDS (read again only in the epilogue)
        float* coef = reinterpret_cast<float*>(smem + COEF_OFF);
        const int ch = tid & 255, oc = (ch < 128 ? pr.c_off0 : pr.c_off1) + (ch & 127);
        const float* src = tid < 256 ? a.epi_scale : a.epi_shift;
        coef[tid] = src ? src[oc] : (tid < 256 ? 1.0f : 0.0f);
    }
    stageA(0, 0);
    stageB(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ntiles > 1) stageB(1, 1);
    if (nslab > 1) stageA(1, 1);
    int j = tap_setup(0);
    load_frags(0, 0, 0);

#ifndef B256_RP
#define B256_RP 2        // fragment-read placement: 0 = ahead of the step's MFMAs, 1 = between them, 2 = pinned after 2 MFMAs
#endif
    // one K tile (4 k-steps of 8 MFMAs); MASK = some row of this wave sees SAME padding at this tap
    auto tile = [&](int n, const bool need_mask) {
        const bool active = !(left_wave && j >= pr.taps0);     // the narrower filter has no tap taps0
        if (!active) {
            // nothing to multiply: keep the tile's one barrier and this wave's share of the staging, then fetch the
            // next tile's first fragments (wave-uniform branch)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (n + 2 < ntiles) stageB(n + 2, n & 1);
            if (n + 1 < ntiles) {
                const int cs1 = (n + 1) / ntap;
                if ((n + 1) - cs1 * ntap == 0 && cs1 + 1 < nslab) stageA(cs1 + 1, (cs1 + 1) & 1);
                j = tap_setup(n + 1);
                load_frags(0, 0, (n + 1) & 1);
            }
            return;
        }
        bool v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = j >= jlo[i] && j < jhi[i];
        const bf16x8 zero = {};
        int jn = j;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            bool have_next = true;
            int nb = n & 1;
            if (s == 3) {
                // every read of tile n has been issued; retire them, publish tile n+1, recycle tile n's buffer
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (!ABL(8)) __syncthreads();
                if (n + 2 < ntiles && !ABL(4)) stageB(n + 2, n & 1);
                have_next = n + 1 < ntiles;
                if (have_next) {
                    const int cs1 = (n + 1) / ntap;
                    // first tile of a slab: bring in the slab after it (its buffer was last read a slab ago)
                    if ((n + 1) - cs1 * ntap == 0 && cs1 + 1 < nslab && !ABL(4)) stageA(cs1 + 1, (cs1 + 1) & 1);
                    jn = tap_setup(n + 1);
                }
                nb = (n + 1) & 1;
            }
            const int sn = (s + 1) & 3;
            const char* ap = As + a_base + a_o[sn];
            const char* bp = Bs + nb * B_BYTES + b_off[sn];
            if (need_mask) {                                   // wave-uniform: scalar branch around 16 selects
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[cur][i] = v[i] ? fa[cur][i] : zero;
            }
            bf16x8* av = fa[cur];
#if B256_RP == 0
            if (have_next) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[nxt][i] = *reinterpret_cast<const bf16x8*>(ap + i * 4096);
                fb[nxt][0] = *reinterpret_cast<const bf16x8*>(bp);
                fb[nxt][1] = *reinterpret_cast<const bf16x8*>(bp + 4096);
            }
#endif
            __builtin_amdgcn_s_setprio(1);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][0], av[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][1], av[0], acc[0][1], 0, 0, 0);
#if B256_RP != 0
            if (have_next) {
                fa[nxt][0] = *reinterpret_cast<const bf16x8*>(ap);
                fa[nxt][1] = *reinterpret_cast<const bf16x8*>(ap + 4096);
            }
#endif
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][0], av[1], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][1], av[1], acc[1][1], 0, 0, 0);
#if B256_RP != 0
            if (have_next) {
                fa[nxt][2] = *reinterpret_cast<const bf16x8*>(ap + 2 * 4096);
                fa[nxt][3] = *reinterpret_cast<const bf16x8*>(ap + 3 * 4096);
            }
#endif
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][0], av[2], acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][1], av[2], acc[2][1], 0, 0, 0);
#if B256_RP != 0
            if (have_next) {
                fb[nxt][0] = *reinterpret_cast<const bf16x8*>(bp);
                fb[nxt][1] = *reinterpret_cast<const bf16x8*>(bp + 4096);
            }
#endif
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][0], av[3], acc[3][0], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][1], av[3], acc[3][1], 0, 0, 0);
#if B256_RP == 2
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#elif B256_RP == 3      // two reads per MFMA gap
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
#elif B256_RP == 4      // reads in two groups of three
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#elif B256_RP == 5      // one read per gap
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#endif
            __builtin_amdgcn_s_setprio(0);
        }
        j = jn;
    };
    const int nrun = ABL(1) ? 0 : ntiles;
    for (int n = 0; n < nrun; ++n) {
        tile(n, !(j >= J_lo && j < J_hi));
    }

    // ---------------- epilogue: BatchNorm scale/shift + activation -> bf16 tile in LDS -> full-row stores
    // The weights are the MFMA's first operand, so a lane holds ONE frame (row li of the 32 x 32
    // tile) and, per register quad q, 4 consecutive channels 8q + 4lh + {0..3}: 8-byte LDS writes.
    constexpr int EP = 528;                            // LDS row pitch of the [256][256] bf16 tile
    __syncthreads();                                   // all fragment reads retired; no load in flight
    if (a.ksplit > 1) {
        // ---------------- split K: the workgroups of a row tile take a ticket as they finish.  Every one but the last
        // publishes its accumulators (register order: one wave instruction = 1 KB) with write-through stores, drains
        // them, and signals; the last polls that word, acquires, adds the slab(s) and runs the epilogue.  Writers wait
        // for nobody, so the exchange cannot deadlock whatever the residency; with two splits the sum a + b does not
        // depend on who was last (cdna_hip_programming.md Guideline 16, form R1).
        typedef __attribute__((address_space(1))) unsigned gu32;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        gu32* const tk = (gu32*)(uintptr_t)(a.tick + 2 * rt);
        int* const tsh = reinterpret_cast<int*>(smem);
        if (tid == 0) tsh[0] = (int)__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(tsh[0]);
        const int voff = (wid * 32 * 64 + lane) * 16;              // + ((i*2 + c)*4 + q) * 1024
        float* const slab0 = a.ws + (size_t)rt * (a.ksplit - 1) * 65536;
        if (ticket < a.ksplit - 1) {
            const uintptr_t base = (uintptr_t)(slab0 + (size_t)ticket * 65536);
            // (wave-uniform by construction; readfirstlane makes that provable.  It returns int: the halves go through
            // unsigned, or the low one sign-extends into the high one)
            const unsigned b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
            const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)b_hi << 32) | (uintptr_t)b_lo), 0, 262144, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][c][4 * q], acc[i][c][4 * q + 1], acc[i][c][4 * q + 2], acc[i][c][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                               voff + ((i * 2 + c) * 4 + q) * 1024, 0, 16);     // aux 16 = sc1
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(tk + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            // bounded: the writers of this row tile hold their tickets already and wait for nothing
            for (unsigned spins = 0; spins < (1u << 21); ++spins) {
                if (__hip_atomic_load(tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(a.ksplit - 1)) break;
                __builtin_amdgcn_s_sleep(16);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        for (int sl = 0; sl < a.ksplit - 1; ++sl) {
            const __attribute__((address_space(1))) char* sp =
                (const __attribute__((address_space(1))) char*)(uintptr_t)(slab0 + (size_t)sl * 65536) + voff;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(sp + ((i * 2 + c) * 4 + q) * 1024);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][c][4 * q + e] += v[e];
                    }
        }
        __syncthreads();                               // tsh[0] was read by every wave before the tile is written over it
    }
    {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int chp = wc * 64 + c * 32 + 8 * q + 4 * lh;             // channel within the pair
                const float4 sv = *reinterpret_cast<const float4*>(smem + COEF_OFF + chp * 4);
                const float4 bv = *reinterpret_cast<const float4*>(smem + COEF_OFF + 1024 + chp * 4);
                const float svv[4] = {sv.x, sv.y, sv.z, sv.w}, bvv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float val = acc[i][c][4 * q + e] * svv[e] + bvv[e];
                        if (a.act == VC_ACT_RELU) val = fmaxf(val, 0.0f);
                        o[e] = (__bf16)val;
                    }
                    const int row = wr * 128 + i * 32 + li;
                    const int col = wc * 64 + c * 32 + 8 * q + 4 * lh;             // column of the 256-wide pair tile
                    *reinterpret_cast<bf16x4*>(smem + row * EP + col * 2) = o;
                }
            }
        }
    }
    __syncthreads();
    {
        const int l16 = tid & 15, hr = tid >> 4;           // 16 lanes x 16 B = one 128-channel half row
        __bf16* C = reinterpret_cast<__bf16*>(a.C);
        const int nrows = a.pool ? BM - 1 : BM;
#pragma unroll 4
        for (int pss = 0; pss < 16; ++pss) {
            const int h = pss * 32 + hr;                   // half-row index: row = h >> 1, half = h & 1
            const int row = h >> 1, half = h & 1;
            const int gm = m0 + row;
            const char* src = smem + row * EP + half * 256 + l16 * 16;
            bf16x8 vv = *reinterpret_cast<const bf16x8*>(src);
            if (a.pool && row < BM - 1) {
                // max_pooling1d(2, 1, 'same') of the post-ReLU result: u16 order == bf16 order for x >= 0;
                // the window's last frame pools with itself
                typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                const bf16x8 nx = *reinterpret_cast<const bf16x8*>(src + EP);
                const bf16x8 mx = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(u16x8, vv),
                                                                                      __builtin_bit_cast(u16x8, nx)));
                vv = (min(gm, a.M - 1) % a.T == a.T - 1) ? vv : mx;
            }
            // streaming store: the 210 MB output must not displace the weight tiles from L2
            if (gm < a.M && row < nrows && !ABL(2))
                __builtin_nontemporal_store(vv, reinterpret_cast<bf16x8*>(C + (size_t)gm * a.ldc + (half ? pr.c_off1 : pr.c_off0) + l16 * 8));
        }
    }
}

}  // namespace

int vc_bank256_ksplit(int M, int nslab) {
    // Two workgroups per row tile while that still fits ONE round of the 256 CUs (a third would open a second round, and
    // with more than two the sum would depend on the arrival order).
    const int ntm = (M + BM - 1) / BM;
    return (2 * ntm <= 256 && nslab >= 8) ? 2 : 1;
}

static size_t tick_bytes(int ntm) { return ((size_t)ntm * 8 + 255) & ~(size_t)255; }

size_t vc_bank256_ws_bytes(int M, int ksplit) {
    if (ksplit <= 1) return 0;
    const int ntm = (M + BM - 1) / BM;
    return tick_bytes(ntm) + (size_t)ntm * (ksplit - 1) * 262144;
}

int vc_launch_bank256(const Bank256Args& a, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bank256_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    const int stride = a.pool ? BM - 1 : BM;
    const int ntm = (a.M + stride - 1) / stride;
    Bank256Args b = a;
    if (a.ksplit > 1) {
        // [tickets | slabs] in the caller's workspace; the polled words are zeroed before EVERY launch
        VC_REQUIRE(a.n_pairs == 1 && !a.pool && a.tick && (reinterpret_cast<uintptr_t>(a.tick) & 255) == 0,
                   "split K: one pair, no pooled output, 256-byte aligned workspace");
        b.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(a.tick) + tick_bytes(ntm));
        b.xcd_tiles = 0;
        VC_HIP_CHECK(hipMemsetAsync(a.tick, 0, tick_bytes(ntm), st));
        hipLaunchKernelGGL(bank256_kernel, dim3((unsigned)(8 * a.ksplit * ((ntm + 7) / 8))), dim3(NT), LDS_BYTES, st, b);
        VC_HIP_CHECK(hipGetLastError());
        return VC_OK;
    }
    const int xcd_mode = vc::opt(vc::OPT_BANK256_XCD);     // 0 plain grid, 1 whole pairs per XCD, else (default) split pairs
    b.xcd_tiles = (a.n_pairs >= 8 && a.n_pairs <= 16 && ntm < 32000 && xcd_mode != 0) ? ntm : 0;
    if (b.xcd_tiles > 0) {
        // Work lists per XCD.  Cost of a row tile of pair p ~ (wider taps) * slabs * 1.3 us + 10 us, 32 CUs per
        // XCD take tiles in list order.  One pair per XCD pairing (rank x with rank 15 - x, whole pairs) balances
        // the totals but quantises badly: 101 heavy tiles on 32 CUs are 3.2 rounds, and the XCD holding ranks 7
        // and 8 runs 6.3 rounds of equal medium tiles -- the model (and the kernel) lose 14 % to those tails.
        // With 16 pairs every pair is therefore split over TWO XCDs and every XCD gets halves of four pairs: one
        // heavy, two medium, one light (ranks s, 7 - s, 8 + s, 15 - s), heaviest first, so short tiles fill the
        // tail: 5 % over the ideal in the model, at two L2 fetches per weight tile instead of one (eight without
        // this mapping).  vc_set_option("bank256_xcd", 1) selects the whole-pair form (A/B).
        for (int x = 0; x < 8; ++x)
            for (int sg = 0; sg < 4; ++sg) b.seg_pair[x][sg] = b.seg_first[x][sg] = b.seg_count[x][sg] = 0;
        const bool split = a.n_pairs == 16 && xcd_mode != 1;
        int max_slots = 0;
        for (int x = 0; x < 8; ++x) {
            int slots = 0;
            if (split) {
                const int s4 = x >> 1, half = x & 1;
                const int ranks[4] = {s4, 7 - s4, 8 + s4, 15 - s4};              // by width, widest first
                const int h0 = (ntm + 1) / 2;
                for (int sg = 0; sg < 4; ++sg) {
                    b.seg_pair[x][sg] = (int16_t)(a.n_pairs - 1 - ranks[sg]);
                    b.seg_first[x][sg] = (int16_t)(half ? h0 : 0);
                    b.seg_count[x][sg] = (int16_t)(half ? ntm - h0 : h0);
                    slots += b.seg_count[x][sg];
                }
            } else {
                int sg = 0;
                for (int gi = 0; gi < a.n_pairs; ++gi) {                        // snake: rank x and 15 - x together
                    const int r16 = gi & 15;
                    if ((r16 < 8 ? r16 : 15 - r16) != x) continue;
                    b.seg_pair[x][sg] = (int16_t)(a.n_pairs - 1 - gi);
                    b.seg_first[x][sg] = 0;
                    b.seg_count[x][sg] = (int16_t)ntm;
                    slots += ntm;
                    ++sg;
                }
            }
            max_slots = slots > max_slots ? slots : max_slots;
        }
        hipLaunchKernelGGL(bank256_kernel, dim3((unsigned)(8 * max_slots)), dim3(NT), LDS_BYTES, st, b);
    } else {
        hipLaunchKernelGGL(bank256_kernel, dim3(ntm, a.n_pairs), dim3(NT), LDS_BYTES, st, b);
    }
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}
