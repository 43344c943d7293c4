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
};

template <int H>
__global__ void __launch_bounds__(2 * H, H == 256 ? 1 : 2)
highway_chain_kernel(HwChainArgs a) {
    constexpr int NT = 2 * H, KS = H / 16, RB = 2 * H, NS = H / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5, x15 = li & 15;
    const int m0 = blockIdx.x * HW_BM;

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
#pragma unroll
    for (int s = 0; s < HW_RING; ++s) {
        const bf16x8* p = wptr(0, s);
        wr[s][0] = p[0];
        wr[s][1] = p[64];
    }
    __syncthreads();

    for (int layer = 0; layer < a.n_layers; ++layer) {
        const char* cur = smem + (layer & 1) * (HW_BM * RB);
        char* nxt = smem + ((layer & 1) ^ 1) * (HW_BM * RB);
        const bool more = layer + 1 < a.n_layers;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;

        bf16x8 xf[2][4];
        const char* xrow = cur + li * RB;
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[0][i] = *reinterpret_cast<const bf16x8*>(xrow + i * 32 * RB + ((lh ^ x15) << 4));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int cb = s & 1;
            if (s + 1 < KS) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xf[cb ^ 1][i] = *reinterpret_cast<const bf16x8*>(xrow + i * 32 * RB + (((2 * (s + 1) + lh) ^ x15) << 4));
            }
            const bf16x8 w0 = wr[s % HW_RING][0], w1 = wr[s % HW_RING][1];
            // refill the ring slot just consumed: step s + RING of this layer, or of the next one
            {
                const int sn = s + HW_RING;
                if (sn < KS) {
                    const bf16x8* p = wptr(layer, sn);
                    wr[s % HW_RING][0] = p[0];
                    wr[s % HW_RING][1] = p[64];
                } else if (more) {
                    const bf16x8* p = wptr(layer + 1, sn - KS);
                    wr[s % HW_RING][0] = p[0];
                    wr[s % HW_RING][1] = p[64];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[cb][i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf[cb][i], acc[i][1], 0, 0, 0);
            }
        }
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
        __syncthreads();
    }

    // ---- final tile -> global
    const char* fin = smem + (a.n_layers & 1) * (HW_BM * RB);
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

// packed[w][s][c][lane][j] = Bt[w*64 + c*32 + (lane & 31)][s*16 + (lane >> 5)*8 + j]
__global__ void __launch_bounds__(256)
highway_pack_kernel(const __bf16* Bt, int H, __bf16* packed) {
    const int KS = H / 16, total = (2 * H / 64) * KS * 2 * 64 * 8;
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

int vc_highway_pack(const void* d_Bt, int32_t H, void* d_packed, void* stream) {
    VC_REQUIRE(d_Bt && d_packed && (H == 128 || H == 256), "vc_highway_pack: H must be 128 or 256");
    hipLaunchKernelGGL(highway_pack_kernel, dim3(128), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(d_Bt), H, static_cast<__bf16*>(d_packed));
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_highway_chain(const void* d_X, int32_t M, int32_t H, int32_t ldx, int32_t n_layers, const void* const* d_packed,
                     const float* const* d_bias, void* d_Y, int32_t ldy, void* stream) {
    VC_REQUIRE(d_X && d_Y && d_packed && d_bias, "vc_highway_chain: NULL argument");
    VC_REQUIRE(H == 128 || H == 256, "vc_highway_chain: H must be 128 or 256 (got %d)", H);
    VC_REQUIRE(M > 0 && n_layers >= 1 && n_layers <= HW_MAX_LAYERS, "vc_highway_chain: bad M / n_layers");
    VC_REQUIRE(ldx >= H && ldy >= H && ldx % 8 == 0 && ldy % 8 == 0, "vc_highway_chain: ldx / ldy must be multiples of 8 and >= H");
    VC_REQUIRE(((reinterpret_cast<uintptr_t>(d_X) | reinterpret_cast<uintptr_t>(d_Y)) & 15) == 0, "vc_highway_chain: X / Y must be 16-byte aligned");
    HwChainArgs a;
    a.X = static_cast<const __bf16*>(d_X); a.Y = static_cast<__bf16*>(d_Y);
    a.M = M; a.ldx = ldx; a.ldy = ldy; a.n_layers = n_layers;
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
