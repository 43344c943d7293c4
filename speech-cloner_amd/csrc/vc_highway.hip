// Highway chain: the L consecutive highwaynet layers of a CBHG block in ONE launch.
//
// /root/reference/modules.py:297-319 (highwaynet), called L times at modules.py:342-345:
//   H = relu(x W1 + b1), T = sigmoid(x W2 + b2), y = H*T + x*(1-T), x <- y.
// Every layer is row-local (a dense layer over channels), so a block keeps its 128 frames on chip
// for all L layers: the activation tile [128][H] (bf16) lives in LDS, double-buffered (layer l reads
// buffer l&1 as the MFMA operand and writes its output into the other one), and only the first
// load and the last store touch HBM.  The per-layer launches moved 2 x M x H x 2 B each at
// 0.8 TB/s and ran the short K loop (K = H) at ~210 TFLOP/s; here HBM traffic drops L-fold.
//
// Decomposition: 2H/64 waves; wave w owns output units [32w, 32w+32): its 64 weight columns are the
// paired (32 x dense1 | 32 x dense2) block w of the layout gemm_kernel's highway mode uses, so the
// H and T pre-activations of a (frame, unit) pair sit in the same lane and register of two
// accumulators and the gate is lane-local.  Weights are the FIRST MFMA operand (lane = frame,
// registers = units) and come straight from global memory, pre-packed in fragment order
// ([layer][wave][k-step][H|T][lane][8]: one coalesced 1 KB load per fragment, L2-resident, a
// 4-deep register ring ahead of the MFMAs) -- LDS holds nothing but the activation tile.
// LDS image: rows of H bf16; the 16-byte slot s of row r sits at slot s ^ (r & 15), so the 16 rows
// of a ds_read_b128 lane group hit 16 different bank groups.
#include "vc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4h __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int HW_BM = 128;
constexpr int HW_MAX_LAYERS = 8;
constexpr int HW_RING = 4;

struct HwChainArgs {
    const __bf16* X;
    __bf16* Y;
    int32_t M, ldx, ldy, n_layers;
    const __bf16* W[HW_MAX_LAYERS];     // packed [2H/64][H/16][2][64][8]
    const float* bias[HW_MAX_LAYERS];   // [2H] in the paired order
    // optional tail: a dense layer over the final activations (the GRU's input projection,
    // modules.py:197-201 -> GRUCell's x-halves), float32 output, no activation
    const __bf16* PW;                   // packed [NP/64][H/16][2][64][8], or NULL
    const float* Pbias;                 // [NP]
    float* P;                           // [M, ldp]
    int32_t NP, ldp;
};

// One 128 x 64 accumulator tile over K = H: 4 frame tiles x (2 weight fragments) per k-step.
// Weight fragments come from a 4-deep register ring that is refilled 4 k-steps ahead -- from this
// tile's own stream, or from `next` (the first steps of whatever tile follows) near the end.  The
// outer loop is kept rolled so the loads address off ONE running pointer (fully unrolled, hipcc
// materialises a 64-bit address per load and spills).
template <int H>
__device__ __forceinline__ void hw_tile(const bf16x8* pw, const bf16x8* next, const char* xrow, int lh, int x15,
                                        bf16x8 (&ring)[HW_RING][2], f32x16 (&acc)[4][2]) {
    constexpr int KS = H / 16, RB = 2 * H;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;
    bf16x8 xf[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[0][i] = *reinterpret_cast<const bf16x8*>(xrow + i * 32 * RB + ((lh ^ x15) << 4));
#pragma unroll 1
    for (int s0 = 0; s0 < KS; s0 += HW_RING) {
        const bf16x8* src = (s0 + HW_RING < KS) ? pw + (size_t)(s0 + HW_RING) * 128 : next;
#pragma unroll
        for (int u = 0; u < HW_RING; ++u) {
            const int s = s0 + u, cb = u & 1;
            if (s + 1 < KS) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xf[cb ^ 1][i] = *reinterpret_cast<const bf16x8*>(xrow + i * 32 * RB + (((2 * (s + 1) + lh) ^ x15) << 4));
            }
            const bf16x8 w0 = ring[u][0], w1 = ring[u][1];
            if (src) {
                ring[u][0] = src[u * 128];
                ring[u][1] = src[u * 128 + 64];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[cb][i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf[cb][i], acc[i][1], 0, 0, 0);
            }
        }
    }
}

// -DVC_ABLATE builds only: cycle sums per phase (s_memtime, thread 0 of workgroup 7), read back with
// vc_ablate_read_highway_stamps (tools/highway_phase_stamps.py).  The shipped library contains none of this.
#ifdef VC_ABLATE
__device__ unsigned long long g_hw_stamps[8];
#define HW_T(i) do { if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_t[i] += t_ - t_prev; t_prev = t_; } } while (0)
#else
#define HW_T(i) do { } while (0)
#endif

template <int H>
__global__ void __launch_bounds__(2 * H, H == 256 ? 1 : 2)
highway_chain_kernel(HwChainArgs a) {
    constexpr int NT = 2 * H, KS = H / 16, RB = 2 * H, NS = H / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5, x15 = li & 15;
    const int m0 = blockIdx.x * HW_BM;
#ifdef VC_ABLATE
    const bool stamp = blockIdx.x == 7 && tid == 0;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif

    // ---- activation tile -> LDS buffer 0
    for (int idx = tid; idx < HW_BM * NS; idx += NT) {
        const int row = idx / NS, slot = idx - row * NS;
        const int gm = min(m0 + row, a.M - 1);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.X + (size_t)gm * a.ldx + slot * 8);
        *reinterpret_cast<bf16x8*>(smem + row * RB + ((slot ^ (row & 15)) << 4)) = v;
    }

    // ---- biases of all layers -> LDS (behind the two activation buffers)
    float* bias_s = reinterpret_cast<float*>(smem + 2 * HW_BM * RB);
    for (int idx = tid; idx < a.n_layers * 2 * H; idx += NT) {
        const int l = idx / (2 * H);
        bias_s[idx] = a.bias[l][idx - l * 2 * H];
    }

    // ---- weight fragment ring: steps 0 .. RING-1 of layer 0
    bf16x8 wr[HW_RING][2];
    auto wptr = [&](int layer, int s) {
        return reinterpret_cast<const bf16x8*>(a.W[layer]) + ((size_t)(w * KS + s) * 2) * 64 + lane;
    };
    if (a.n_layers > 0) {
#pragma unroll
        for (int s = 0; s < HW_RING; ++s) {
            const bf16x8* p = wptr(0, s);
            wr[s][0] = p[0];
            wr[s][1] = p[64];
        }
    }
    __syncthreads();
    HW_T(0);

    for (int layer = 0; layer < a.n_layers; ++layer) {
        const char* cur = smem + (layer & 1) * (HW_BM * RB);
        char* nxt = smem + ((layer & 1) ^ 1) * (HW_BM * RB);
        const bool more = layer + 1 < a.n_layers;
        f32x16 acc[4][2];
        const bf16x8* nextw = more ? wptr(layer + 1, 0) : (a.PW ? reinterpret_cast<const bf16x8*>(a.PW) + ((size_t)w * KS * 2) * 64 + lane : nullptr);
        hw_tile<H>(wptr(layer, 0), nextw, cur + li * RB, lh, x15, wr, acc);
#ifdef VC_ABLATE
        if (stamp) asm volatile("" :: "v"(acc[3][1][15]));     // the stamp waits for the last accumulator
#endif
        HW_T(1);
        // ---- gate (lane-local) -> next activation tile
        f32x4h bH[4], bT[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bH[q] = *reinterpret_cast<const f32x4h*>(bias_s + layer * 2 * H + w * 64 + 8 * q + 4 * lh);
            bT[q] = *reinterpret_cast<const f32x4h*>(bias_s + layer * 2 * H + w * 64 + 32 + 8 * q + 4 * lh);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = (i * 32 + li) * RB + (((4 * w + q) ^ x15) << 4) + lh * 8;
                const bf16x4 xin = *reinterpret_cast<const bf16x4*>(cur + off);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (__bf16)vc::highway_gate(acc[i][0][4 * q + e] + bH[q][e], acc[i][1][4 * q + e] + bT[q][e], (float)xin[e]);
                }
                *reinterpret_cast<bf16x4*>(nxt + off) = o;
            }
        }
        HW_T(2);
        __syncthreads();
        HW_T(3);
    }

    const char* fin = smem + (a.n_layers & 1) * (HW_BM * RB);
    // ---- tail projection: P[frame, n] = sum_k X[frame, k] * PW[n, k] + Pbias[n], 64 columns per wave
    // and pass (lane = frame; a register quad = 4 consecutive columns -> 16-byte float32 stores)
    if (a.PW) {
        constexpr int NW = 2 * H / 64;
        const int ngroups = a.NP / 64;
        const char* xrow = fin + li * RB;
        char* const stg = smem + ((a.n_layers & 1) ^ 1) * (HW_BM * RB) + w * 8192;    // the other activation buffer is idle
        static_assert((2 * H / 64) * 8192 <= HW_BM * 2 * H, "a 32 x 64 float32 sub-tile per wave fits the idle buffer");
        if (a.n_layers == 0) {
            const bf16x8* p0 = reinterpret_cast<const bf16x8*>(a.PW) + ((size_t)w * KS * 2) * 64 + lane;
#pragma unroll
            for (int s = 0; s < HW_RING; ++s) { wr[s][0] = p0[s * 128]; wr[s][1] = p0[s * 128 + 64]; }
        }
        for (int grp = w; grp < ngroups; grp += NW) {
            const bf16x8* pw = reinterpret_cast<const bf16x8*>(a.PW) + ((size_t)grp * KS * 2) * 64 + lane;
            const bf16x8* nextw = (grp + NW < ngroups) ? pw + (size_t)NW * KS * 128 : nullptr;
            f32x16 acc[4][2];
            hw_tile<H>(pw, nextw, xrow, lh, x15, wr, acc);
            // A lane holds one frame and 4 consecutive columns per register quad: stored directly, one instruction
            // writes 32 rows x 32 B.  The launch's second half is this 786 KB-per-block store stream, so each 32 x 64
            // sub-tile goes through a wave-private 8 KB of the idle activation buffer (16-byte chunk c of row r at
            // chunk c ^ (r & 15); LDS operations of one wave execute in order: no barrier) and leaves as 256-byte rows.
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4h bb = *reinterpret_cast<const f32x4h*>(a.Pbias + grp * 64 + c * 32 + 8 * q + 4 * lh);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][c][4 * q + e] += bb[e];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4h o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[i][c][4 * q + e];
                        *reinterpret_cast<f32x4h*>(stg + li * 256 + (((c * 8 + 2 * q + lh) ^ x15) << 4)) = o;
                    }
#pragma unroll 2
                for (int j = 0; j < 8; ++j) {
                    const int row = j * 4 + (lane >> 4), ch = lane & 15;
                    const f32x4h v = *reinterpret_cast<const f32x4h*>(stg + row * 256 + ((ch ^ (row & 15)) << 4));
                    const int gm = m0 + i * 32 + row;
                    if (gm < a.M) *reinterpret_cast<f32x4h*>(a.P + (size_t)gm * a.ldp + grp * 64 + ch * 4) = v;
                }
            }
        }
    }
    HW_T(4);
#ifdef VC_ABLATE
    if (stamp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_hw_stamps[i] = acc_t[i];
    }
#endif
    if (a.Y == nullptr) return;
    // ---- final tile -> global
    for (int idx = tid; idx < HW_BM * NS; idx += NT) {
        const int row = idx / NS, slot = idx - row * NS;
        const int gm = m0 + row;
        if (gm < a.M)
            *reinterpret_cast<bf16x8*>(a.Y + (size_t)gm * a.ldy + slot * 8) =
                *reinterpret_cast<const bf16x8*>(fin + row * RB + ((slot ^ (row & 15)) << 4));
    }
}

template <int H> int launch_chain(const HwChainArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * HW_BM * 2 * H + HW_MAX_LAYERS * 2 * H * 4;
    static bool attr_done = false;
    if (!attr_done) {
        VC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(highway_chain_kernel<H>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done = true;
    }
    hipLaunchKernelGGL(highway_chain_kernel<H>, dim3((a.M + HW_BM - 1) / HW_BM), dim3(2 * H), LDS, st, a);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

// packed[w][s][c][lane][j] = Bt[w*64 + c*32 + (lane & 31)][s*16 + (lane >> 5)*8 + j]   (Bt: [ncols][H])
__global__ void __launch_bounds__(256)
highway_pack_kernel(const __bf16* Bt, int ncols, int H, __bf16* packed) {
    const int KS = H / 16, total = (ncols / 64) * KS * 2 * 64 * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        int r = idx;
        const int j = r & 7; r >>= 3;
        const int lane = r & 63; r >>= 6;
        const int c = r & 1; r >>= 1;
        const int s = r % KS;
        const int w = r / KS;
        packed[idx] = Bt[(size_t)(w * 64 + c * 32 + (lane & 31)) * H + s * 16 + (lane >> 5) * 8 + j];
    }
}

}  // namespace

extern "C" {

int vc_highway_pack(const void* d_Bt, int32_t n_cols, int32_t H, void* d_packed, void* stream) {
    VC_REQUIRE(d_Bt && d_packed && (H == 128 || H == 256) && n_cols > 0 && n_cols % 64 == 0,
               "vc_highway_pack: H must be 128 or 256 and n_cols a multiple of 64");
    hipLaunchKernelGGL(highway_pack_kernel, dim3(128), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(d_Bt), n_cols, H, static_cast<__bf16*>(d_packed));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

#ifdef VC_ABLATE
int vc_ablate_read_highway_stamps(unsigned long long* h_out) {
    VC_HIP_CHECK(hipDeviceSynchronize());
    VC_HIP_CHECK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_hw_stamps), sizeof(unsigned long long) * 8));
    return VC_OK;
}
#endif

int vc_highway_chain(const void* d_X, int32_t M, int32_t H, int32_t ldx, int32_t n_layers, const void* const* d_packed,
                     const float* const* d_bias, void* d_Y, int32_t ldy, const void* d_proj_packed,
                     const float* d_proj_bias, int32_t n_proj, float* d_P, int32_t ldp, void* stream) {
    VC_REQUIRE(d_X && d_packed && d_bias && (d_Y || d_proj_packed), "vc_highway_chain: NULL argument");
    if (d_proj_packed)
        VC_REQUIRE(d_proj_bias && d_P && n_proj > 0 && n_proj % 64 == 0 && ldp >= n_proj && ldp % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(d_proj_packed) | reinterpret_cast<uintptr_t>(d_proj_bias) |
                         reinterpret_cast<uintptr_t>(d_P)) & 15) == 0,
                   "vc_highway_chain: projection tail needs bias, output, n_proj %% 64 == 0, 16-byte aligned operands");
    VC_REQUIRE(H == 128 || H == 256, "vc_highway_chain: H must be 128 or 256 (got %d)", H);
    VC_REQUIRE(M > 0 && n_layers >= 0 && n_layers <= HW_MAX_LAYERS, "vc_highway_chain: bad M / n_layers");
    VC_REQUIRE(ldx >= H && ldx % 8 == 0 && (!d_Y || (ldy >= H && ldy % 8 == 0)), "vc_highway_chain: ldx / ldy must be multiples of 8 and >= H");
    VC_REQUIRE(((reinterpret_cast<uintptr_t>(d_X) | reinterpret_cast<uintptr_t>(d_Y)) & 15) == 0, "vc_highway_chain: X / Y must be 16-byte aligned");
    HwChainArgs a;
    a.X = static_cast<const __bf16*>(d_X); a.Y = static_cast<__bf16*>(d_Y);
    a.M = M; a.ldx = ldx; a.ldy = ldy; a.n_layers = n_layers;
    a.PW = static_cast<const __bf16*>(d_proj_packed); a.Pbias = d_proj_bias; a.P = d_P; a.NP = n_proj; a.ldp = ldp;
    for (int l = 0; l < n_layers; ++l) {
        VC_REQUIRE(d_packed[l] && d_bias[l] && (reinterpret_cast<uintptr_t>(d_packed[l]) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(d_bias[l]) & 15) == 0, "vc_highway_chain: layer %d operands NULL or misaligned", l);
        a.W[l] = static_cast<const __bf16*>(d_packed[l]);
        a.bias[l] = d_bias[l];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    return H == 256 ? launch_chain<256>(a, st) : launch_chain<128>(a, st);
}

}  // extern "C"
