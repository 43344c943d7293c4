// Training-step kernels for the decoder (/root/reference/decoder.py:185-263, 327-345) that are
// not GEMMs: batch statistics + train-mode FusedBatchNorm (modules.py:39-102), its backward fused
// with ReLU / max-pool backward, highway gate backward (modules.py:297-319), transposes that feed
// the weight-gradient GEMM, MSE loss + gradient (decoder.py:187-195), bias gradients, the
// TF-style Adam update (decoder.py:236-246) on one flat parameter buffer, and the GRU forward
// (with saved gates) / BPTT recurrences (modules.py:168-204).  Contract: include/vc_hip.h.
// Activations here are float32 (the reference trains in float32).
#include <hip/hip_runtime.h>

#include "vc_common.h"

namespace {

constexpr int TB = 256;
// rows summed by one block of the column-statistics kernels: 64 (was 256) gives the 12,800 x 4,096 bank tensors 3,200
// blocks instead of 800 -- these loops are load-latency bound, more waves hide more of it
constexpr int BN_ROWS = 64;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// ---------------------------------------------------------------------------- column statistics
// X [M, C] (row stride ld) -> partial sums over row blocks: part[blk][2][C].  Deterministic:
// fixed row partition, fixed summation order; the finalize kernel adds the blocks in order.
__global__ void __launch_bounds__(TB)
col_stats_partial_kernel(const float* X, int M, int C, int ld, int rows_per_blk, float* part) {
    const int c = blockIdx.x * TB + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
    float s = 0.0f, q = 0.0f;
    for (int r = r0; r < r1; ++r) {
        const float v = X[(size_t)r * ld + c];
        s += v;
        q = fmaf(v, v, q);
    }
    part[((size_t)blockIdx.y * 2 + 0) * C + c] = s;
    part[((size_t)blockIdx.y * 2 + 1) * C + c] = q;
}

// Column sums over `nblk` rows of per-block partials, two interleaved quantities per block (rows 2b, 2b + 1 of `part`;
// NQ = 1: one quantity, rows b): 256 threads = 32 channels x 8 row segments, double accumulation in a fixed order.  One
// thread per channel walking all the partials was a 200-deep chain of dependent loads: 55-72 us per call, six calls on
// the critical path of a training step.  Results land in lanes with seg == 0 (returns false elsewhere).
template <int NQ>
__device__ __forceinline__ bool sum_partials(const float* __restrict__ part, int nblk, int C, int& c, double& s, double& q) {
    __shared__ double sh[2][8][33];
    const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    c = blockIdx.x * 32 + cl;
    s = 0.0; q = 0.0;
    if (c < C)
        for (int b = seg; b < nblk; b += 8) {
            s += (double)part[((size_t)b * NQ + 0) * C + c];
            if (NQ == 2) q += (double)part[((size_t)b * NQ + 1) * C + c];
        }
    sh[0][seg][cl] = s;
    sh[1][seg][cl] = q;
    __syncthreads();
    if (seg != 0 || c >= C) return false;
#pragma unroll
    for (int k = 1; k < 8; ++k) { s += sh[0][k][cl]; q += sh[1][k][cl]; }
    return true;
}

// Train-mode batch norm bookkeeping: batch mean / biased variance -> scale, shift for the
// consumer's prologue, saved mean / rstd for backward, moving statistics updated in place with
// the Bessel-corrected variance (tf.nn.fused_batch_norm semantics), decay 0.999, eps 1e-3.
__global__ void __launch_bounds__(TB)
bn_train_finalize_kernel(const float* part, int nblk, int M, int C, const float* gamma, const float* beta,
                         float* moving_mean, float* moving_var, float decay, float eps,
                         float* scale, float* shift, float* mean_out, float* rstd_out) {
    int c;
    double s, q;
    if (!sum_partials<2>(part, nblk, C, c, s, q)) return;
    const double mean = s / M;
    double var = q / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    mean_out[c] = (float)mean;
    rstd_out[c] = rstd;
    if (moving_mean) {
        const double unb = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
        moving_mean[c] = moving_mean[c] * decay + (float)mean * (1.0f - decay);
        moving_var[c] = moving_var[c] * decay + (float)unb * (1.0f - decay);
    }
}

// out = act(X * scale[c] + shift[c]) + R      (BatchNorm apply + residual, modules.py:338-340)
__global__ void __launch_bounds__(TB)
affine_act_kernel(const float* X, const float* scale, const float* shift, int relu, const float* R, float* out,
                  size_t n, int C) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * TB;
    for (; i < n; i += stride) {
        const int c = (int)(i % C);
        float v = X[i] * (scale ? scale[c] : 1.0f) + (shift ? shift[c] : 0.0f);
        if (relu) v = fmaxf(v, 0.0f);
        if (R) v += R[i];
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------- BN backward
// Upstream gradient G is w.r.t.  Y = post(bn(X)) where post = identity | relu | pool(relu):
//   mode 0: dBN = G
//   mode 1: dBN = G * [bn(X) > 0]                                            (relu)
//   mode 2: G is w.r.t. the max-pooled relu output P[t] = max(A[t], A[t+1]) (last frame: A[t]):
//           dA[t] = G[t]*[t==T-1 or A[t] >= A[t+1]] + G[t-1]*[t>0 and A[t] > A[t-1]],  dBN = dA*[A>0]
// The routing decisions of modes 1 / 2 for element (row, c): bit 0: bn(X) > 0 (relu passes), bit 1: the element
// takes G[t] (it is the pool winner of its own frame), bit 2: it takes G[t-1] (winner of the previous frame's pool).
// vc_bn_post_routing (the export the parity tests hand to the oracle) goes through this; bn_bwd_pass_kernel takes the same
// decisions from the same float32 expressions, with the neighbouring frames' values kept in registers.
__device__ __forceinline__ int bn_route(const float* X, int ld, size_t i, int t, int T, float sc, float sh) {
    const float a = fmaxf(X[i] * sc + sh, 0.0f);
    if (!(a > 0.0f)) return 0;
    int bits = 1;
    if (t == T - 1) bits |= 2;
    else {
        const float an = fmaxf(X[i + ld] * sc + sh, 0.0f);
        if (a >= an) bits |= 2;
    }
    if (t > 0) {
        const float ap = fmaxf(X[i - ld] * sc + sh, 0.0f);
        if (a > ap) bits |= 4;
    }
    return bits;
}

__global__ void __launch_bounds__(TB)
bn_routing_kernel(const float* X, int M, int C, int ld, int T, const float* scale, const float* shift, unsigned char* out) {
    const size_t n = (size_t)M * C;
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * TB;
    for (; i < n; i += stride) {
        const size_t r = i / C;
        const int c = (int)(i - r * C);
        out[i] = (unsigned char)bn_route(X, ld, r * ld + c, (int)(r % T), T, scale[c], shift[c]);
    }
}

// One pass over a block of rows, one thread per channel (coalesced 4-byte columns), every element of G and X read ONCE:
// the relu / pool routing of mode 2 needs the activations of the neighbouring frames and the previous frame's gradient,
// which a thread walking down its column keeps in registers (bn_route's decisions, computed from the same float32
// expressions).  APPLY = false: partial column sums of dBN and dBN * xhat; APPLY = true: dX = gamma * rstd * (dBN -
// dbeta / M - xhat * dgamma / M).
template <bool APPLY>
__global__ void __launch_bounds__(TB)
bn_bwd_pass_kernel(const float* __restrict__ G, const float* __restrict__ X, int M, int C, int ld, int T,
                   const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                   const float* __restrict__ rstd, int mode, int rows_per_blk, float* __restrict__ part,
                   const float* __restrict__ gamma, const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                   float* __restrict__ dX) {
    const int c = blockIdx.x * TB + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
    if (r0 >= r1) return;
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    float k0 = 0.0f, k1 = 0.0f, k2 = 0.0f;
    if (APPLY) {
        const float invM = 1.0f / (float)M;
        k0 = gamma[c] * rs; k1 = dbeta[c] * invM; k2 = dgamma[c] * invM;
    }
    float s = 0.0f, q = 0.0f;
    int t = r0 % T;
    const float* xp = X + (size_t)r0 * ld + c;
    const float* gp = G + (size_t)r0 * ld + c;
    float x_cur = xp[0];
    float a_prev = 0.0f, g_prev = 0.0f;
    if (mode == 2 && t > 0) {
        a_prev = fmaxf(xp[-ld] * sc + sh, 0.0f);
        g_prev = gp[-ld];
    }
    for (int r = r0; r < r1; ++r) {
        const float g_cur = gp[0];
        const bool has_next = r + 1 < M;                 // (a window's last frame never looks at it)
        const float x_next = has_next ? xp[ld] : 0.0f;
        float d;
        if (mode == 0) {
            d = g_cur;
        } else {
            const float a = fmaxf(x_cur * sc + sh, 0.0f);
            if (mode == 1) {
                d = a > 0.0f ? g_cur : 0.0f;
            } else {
                d = 0.0f;
                if (a > 0.0f) {
                    const float an = fmaxf(x_next * sc + sh, 0.0f);
                    if (t == T - 1 || a >= an) d += g_cur;
                    if (t > 0 && a > a_prev) d += g_prev;
                }
                a_prev = a;
                g_prev = g_cur;
            }
        }
        const float xh = (x_cur - mu) * rs;
        if (APPLY) {
            dX[(size_t)r * ld + c] = k0 * (d - k1 - xh * k2);
        } else {
            s += d;
            q = fmaf(d, xh, q);
        }
        x_cur = x_next;
        xp += ld; gp += ld;
        if (++t == T) t = 0;
    }
    if (!APPLY) {
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = s;
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = q;
    }
}

// reduce partials -> dbeta, dgamma ; dX = gamma*rstd*(dBN - dbeta/M - xhat*dgamma/M)
__global__ void __launch_bounds__(TB)
bn_bwd_reduce_kernel(const float* part, int nblk, int C, float* dbeta, float* dgamma) {
    int c;
    double s, q;
    if (!sum_partials<2>(part, nblk, C, c, s, q)) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

// ---------------------------------------------------------------------------- relu/dropout bwd
// Y = dropout(relu(z)) is stored; dz = (Y > 0) ? dY * inv_keep : 0   (inv_keep = 1 without dropout)
__global__ void __launch_bounds__(TB)
relu_drop_bwd_kernel(const float* dY, const float* Y, float inv_keep, float* dZ, size_t n) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * TB;
    for (; i < n; i += stride) dZ[i] = Y[i] > 0.0f ? dY[i] * inv_keep : 0.0f;
}

// ---------------------------------------------------------------------------- highway backward
// pre [M, NP] holds the re-computed pre-activations in the paired layout of the forward kernel
// (per 64 columns: 32 x dense1 | 32 x dense2, bias included).  With h = relu(ph), t = sig(pt):
//   out = h t + x (1 - t)
//   dpre_h = dO * t * [ph > 0] ; dpre_t = dO * (h - x) * t (1 - t) ; dx_direct = dO * (1 - t)
__global__ void __launch_bounds__(TB)
highway_bwd_kernel(const float* pre, int NP, const float* X, const float* dO, int M, int H, float* dpre, float* dXd) {
    const size_t n = (size_t)M * H;
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * TB;
    for (; i < n; i += stride) {
        const size_t r = i / H;
        const int j = (int)(i - r * H);
        const int ch = 64 * (j >> 5) + (j & 31), ct = ch + 32;
        const float ph = pre[r * NP + ch], pt = pre[r * NP + ct];
        const float h = fmaxf(ph, 0.0f), t = sigmoidf_(pt);
        const float g = dO[i], x = X[i];
        dpre[r * NP + ch] = ph > 0.0f ? g * t : 0.0f;
        dpre[r * NP + ct] = g * (h - x) * t * (1.0f - t);
        dXd[i] = g * (1.0f - t);
    }
}

// zero the padding columns of a paired-layout gradient buffer (units >= H inside the last 64-block)
__global__ void __launch_bounds__(TB)
fill_kernel(float* p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * TB;
    for (; i < n; i += stride) p[i] = v;
}

// out[m][c] = a * X[m][c] + b * Y[m][c] on [M, C] views with row strides (teacher-forced blend of decoder stage 2's
// input, /root/reference/decoder.py:152, and its gradient)
__global__ void __launch_bounds__(TB)
axpby_kernel(const float* X, int ldx, float a, const float* Y, int ldy, float b, float* out, int ldo, int M, int C) {
    const size_t n = (size_t)M * C;
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * TB;
    for (; i < n; i += stride) {
        const size_t r = i / C;
        const int c = (int)(i - r * C);
        out[r * ldo + c] = a * X[r * ldx + c] + b * Y[r * ldy + c];
    }
}

// ---------------------------------------------------------------------------- transpose (+prologue)
// X [M, C] (ld) -> XT [C, ldt] at column offset `pad`:  XT[c][pad + m] = pro(X)[m][c]
//   pro: optional per-channel affine, relu, time max-pool (same rule as the forward operand load),
//   optional row shift by `shift` frames inside each window (zero where the source leaves it)
//   -- the latter builds h_{t-1} / h_{t+1} for the recurrent weight gradients.
__global__ void __launch_bounds__(TB)
transpose_pad_kernel(const float* X, int M, int C, int ld, int T, const float* scale, const float* shiftv, int relu,
                     int pool, int row_shift, float* XT, int ldt, int pad, int slack_row) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int m = m0 + k, c = c0 + tx;
        float v = 0.0f;
        if (m < M && c < C) {
            const int t = m % T;
            const int ts = t + row_shift;
            if (ts >= 0 && ts < T) {
                const size_t i = (size_t)(m + row_shift) * ld + c;
                const float sc = scale ? scale[c] : 1.0f, sh = shiftv ? shiftv[c] : 0.0f;
                v = X[i] * sc + sh;
                if (relu) v = fmaxf(v, 0.0f);
                if (pool && ts + 1 < T) {
                    float v2 = X[i + ld] * sc + sh;
                    if (relu) v2 = fmaxf(v2, 0.0f);
                    v = fmaxf(v, v2);
                }
            }
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, m = m0 + tx;
        if (c < C && m < M) XT[(size_t)c * ldt + pad + m] = tile[tx][k];
    }
    // the zero margins on both sides of every row and the slack row C (the weight-gradient tiles read them): written
    // here, so the caller hands over an uninitialised buffer instead of zero-filling all (C + 1) x ldt floats first
    const int tid = threadIdx.x;
    if (blockIdx.x == 0)
        for (int i = tid; i < 32 * pad; i += TB) {
            const int c = c0 + i / pad;
            if (c < C) XT[(size_t)c * ldt + i % pad] = 0.0f;
        }
    if (blockIdx.x == gridDim.x - 1) {
        const int wr = ldt - (pad + M);
        for (int i = tid; i < 32 * wr; i += TB) {
            const int c = c0 + i / wr;
            if (c < C) XT[(size_t)c * ldt + pad + M + i % wr] = 0.0f;
        }
    }
    if (blockIdx.y == gridDim.y - 1 && slack_row) {
        const int lo = blockIdx.x == 0 ? 0 : pad + m0, hi = blockIdx.x == gridDim.x - 1 ? ldt : pad + m0 + 32;
        for (int col = lo + tid; col < hi; col += TB) XT[(size_t)C * ldt + col] = 0.0f;
    }
}

// column sums (bias gradients): out[c] (+)= sum_m X[m][c].  One block = 64 columns x 4 row
// groups; rows are strided over blockIdx.y * 4 + group, partials combined in a fixed order
// (deterministic) by the last kernel.
__global__ void __launch_bounds__(TB)
col_sum_partial_kernel(const float* X, int M, int C, int ld, int nrb, float* part) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.0f;
    if (c < C)
        for (int r = blockIdx.y * 4 + rg; r < M; r += nrb * 4) s += X[(size_t)r * ld + c];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C) part[(size_t)blockIdx.y * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
__global__ void __launch_bounds__(TB)
col_sum_final_kernel(const float* part, int nrb, int C, float* out, int accumulate) {
    int c;
    double s, q;
    if (!sum_partials<1>(part, nrb, C, c, s, q)) return;
    out[c] = (accumulate ? out[c] : 0.0f) + (float)s;
}

// ---------------------------------------------------------------------------- loss
// loss = w * mean((y - t)^2) ; dY = 2 w / n * (y - t).  One block -> one partial (deterministic
// second pass on the host side of the ABI: partials are summed in order by loss_final_kernel).
__global__ void __launch_bounds__(TB)
mse_partial_kernel(const float* y, const float* t, size_t n, float w, float* dY, int C, int ld, float* part) {
    __shared__ float red[TB / 64];
    const float g = 2.0f * w / (float)n;
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += (size_t)gridDim.x * TB) {
        const float d = y[i] - t[i];
        s = fmaf(d, d, s);
        if (dY) dY[(i / C) * ld + (i % C)] = g * d;
    }
    s = vc::wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void loss_final_kernel(const float* part, int nblk, size_t n, float w, float* out) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[b];
    out[0] = (float)((double)w * s / (double)n);
}

// ---------------------------------------------------------------------------- Adam
// tf.train.AdamOptimizer (decoder.py:236-246): lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (host),
// m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t * m / (sqrt(v) + eps)
__global__ void __launch_bounds__(TB)
adam_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps,
            float gscale) {
    size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * TB;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// ---------------------------------------------------------------------------- GRU (training)
// Generic one-workgroup-per-(window, direction) recurrences in float32 with the weights read
// through L2 (or LDS when they fit).  Forward saves r, u, c, r*h for the backward pass.
struct GruTrainArgs {
    const float* xproj;      // [n_seq*T, 6H]
    const float* Wh[2];      // [H, 3H]
    float* out;              // [n_seq*T, 2H]
    float* gates;            // [2][n_seq*T, 3H]  r | u | c (post-activation)
    float* rh;               // [2][n_seq*T, H]
    int32_t n_seq, T, H;
};

__global__ void __launch_bounds__(512)
gru_train_fwd_kernel(GruTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, H3 = 3 * H, NT = blockDim.x, tid = threadIdx.x;
    float* h = reinterpret_cast<float*>(smem);
    float* rhs = h + H;
    float* us = rhs + H;
    const int seq = blockIdx.x, dir = blockIdx.y;
    const float* W = a.Wh[dir];
    const size_t MT = (size_t)a.n_seq * a.T;
    float* gates = a.gates + (size_t)dir * MT * H3;
    float* rhg = a.rh + (size_t)dir * MT * H;
    for (int i = tid; i < H; i += NT) h[i] = 0.0f;
    __syncthreads();
    const size_t xrow = 6 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    for (int step = 0; step < a.T; ++step, t += dt) {
        const size_t row = (size_t)seq * a.T + t;
        for (int col = tid; col < 2 * H; col += NT) {
            float acc = xbase[(size_t)t * xrow + col];
            for (int k = 0; k < H; ++k) acc = fmaf(h[k], W[(size_t)k * H3 + col], acc);
            const float g = sigmoidf_(acc);
            gates[row * H3 + col] = g;
            if (col < H) { const float v = g * h[col]; rhs[col] = v; rhg[row * H + col] = v; }
            else us[col - H] = g;
        }
        __syncthreads();
        float hn = 0.0f;
        for (int col = tid; col < H; col += NT) {
            float acc = xbase[(size_t)t * xrow + 2 * H + col];
            for (int k = 0; k < H; ++k) acc = fmaf(rhs[k], W[(size_t)k * H3 + 2 * H + col], acc);
            const float c = tanhf(acc);
            gates[row * H3 + 2 * H + col] = c;
            const float u = us[col];
            hn = u * h[col] + (1.0f - u) * c;
            a.out[row * 2 * H + (size_t)dir * H + col] = hn;
        }
        __syncthreads();
        for (int col = tid; col < H; col += NT) h[col] = a.out[row * 2 * H + (size_t)dir * H + col];
        __syncthreads();
    }
}

struct GruBwdArgs {
    const float* dout;       // [n_seq*T, 2H]   gradient w.r.t. the GRU output
    const float* out;        // [n_seq*T, 2H]   forward output (h_t)
    const float* gates;      // [2][n_seq*T, 3H]
    const float* Wh[2];      // [H, 3H]
    float* dpre;             // [n_seq*T, 6H]   d(pre-activation) r | u | c per direction (same layout as xproj)
    int32_t n_seq, T, H;
};

// BPTT: runs each direction's time loop in reverse.  Per step (h = h_{t-1} in that direction):
//   dh += dout_t ; du = dh (h - c) ; dc = dh (1 - u) ; dh_prev = dh u
//   dc_pre = dc (1 - c^2) ; d(rh) = Wc_h dc_pre ; dr = d(rh) h ; dh_prev += d(rh) r
//   dr_pre = dr r (1-r) ; du_pre = du u (1-u) ; dh_prev += Wg_h [dr_pre ; du_pre]
__global__ void __launch_bounds__(512)
gru_bwd_kernel(GruBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, H3 = 3 * H, NT = blockDim.x, tid = threadIdx.x;
    float* dh = reinterpret_cast<float*>(smem);     // [H] carried gradient
    float* dcp = dh + H;                            // [H] dc_pre
    float* dgp = dcp + H;                           // [2H] dr_pre | du_pre
    float* drh = dgp + 2 * H;                       // [H]
    const int seq = blockIdx.x, dir = blockIdx.y;
    const float* W = a.Wh[dir];
    const size_t MT = (size_t)a.n_seq * a.T;
    const float* gates = a.gates + (size_t)dir * MT * H3;
    for (int i = tid; i < H; i += NT) dh[i] = 0.0f;
    __syncthreads();
    // forward order was t0, t0+dt, ...; walk it backwards
    int t = dir ? 0 : a.T - 1;
    const int dtb = dir ? 1 : -1;                   // towards the forward pass's start
    for (int step = 0; step < a.T; ++step, t += dtb) {
        const size_t row = (size_t)seq * a.T + t;
        const bool first = (step == a.T - 1);       // forward's first step: h_prev = 0
        const size_t prow = row + dtb;              // row of h_{prev} in forward order
        // stage 1: elementwise through h' = u h + (1-u) c
        for (int j = tid; j < H; j += NT) {
            const float g = dh[j] + a.dout[row * 2 * H + (size_t)dir * H + j];
            const float u = gates[row * H3 + H + j], c = gates[row * H3 + 2 * H + j];
            const float hp = first ? 0.0f : a.out[prow * 2 * H + (size_t)dir * H + j];
            const float du = g * (hp - c);
            const float dc = g * (1.0f - u);
            dcp[j] = dc * (1.0f - c * c);
            dgp[H + j] = du * u * (1.0f - u);
            dh[j] = g * u;                           // dh_prev, part 1
        }
        __syncthreads();
        // stage 2: d(rh) = Wc_h @ dc_pre  (Wc_h = W[:, 2H:3H], rows = rh index)
        for (int k = tid; k < H; k += NT) {
            float acc = 0.0f;
            const float* w = W + (size_t)k * H3 + 2 * H;
            for (int j = 0; j < H; ++j) acc = fmaf(w[j], dcp[j], acc);
            drh[k] = acc;
        }
        __syncthreads();
        for (int j = tid; j < H; j += NT) {
            const float r = gates[row * H3 + j];
            const float hp = first ? 0.0f : a.out[prow * 2 * H + (size_t)dir * H + j];
            const float dr = drh[j] * hp;
            dgp[j] = dr * r * (1.0f - r);
            dh[j] += drh[j] * r;
        }
        __syncthreads();
        // stage 3: dh_prev += Wg_h @ [dr_pre ; du_pre]
        for (int k = tid; k < H; k += NT) {
            float acc = 0.0f;
            const float* w = W + (size_t)k * H3;
            for (int j = 0; j < 2 * H; ++j) acc = fmaf(w[j], dgp[j], acc);
            dh[k] += acc;
        }
        // write pre-activation gradients in xproj layout
        float* dp = a.dpre + row * 6 * (size_t)H + (size_t)dir * H3;
        for (int j = tid; j < 2 * H; j += NT) dp[j] = dgp[j];
        for (int j = tid; j < H; j += NT) dp[2 * H + j] = dcp[j];
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------- GRU (training), S sequences per WG
// The one-sequence kernels above re-read the whole recurrent matrix (3H^2 floats = 786 KB at H = 256)
// from L2 every step.  Here a workgroup advances S windows of one direction, so every weight
// load feeds S FMAs; h lives in LDS as [S][H] and is read back as float4 broadcasts.
// acc[s] += sum_k vec[s][k] * w[k * stride]  for k in [0, n): the weight column is read in batches
// of 16 independent loads (a 4-load batch per iteration left each wave waiting on L2 64-128 times
// per phase and step: the recurrences were latency-bound).  n must be a multiple of 4.
#ifdef VC_ABLATE
__device__ int g_gru_train_ablate;                  // -DVC_ABLATE builds only: set by vc_ablate_set_gru_train
#endif
template <int S>
__device__ __forceinline__ void ms_matvec(const float* __restrict__ w, size_t stride, const float* vec, int vstride, int n,
                                          float (&acc)[S]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    int k = 0;
    for (; k + 16 <= n; k += 16) {
        float wv[16];
#ifdef VC_ABLATE
        if (g_gru_train_ablate) {                            // timing only: no weight stream (tools/ab_gru_train.py --floor)
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = 0.001f * (float)(k + u);
        } else
#endif
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = w[(size_t)(k + u) * stride];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const f4 v = *reinterpret_cast<const f4*>(vec + s * vstride + k + 4 * q);
                acc[s] = fmaf(v[0], wv[4 * q], acc[s]); acc[s] = fmaf(v[1], wv[4 * q + 1], acc[s]);
                acc[s] = fmaf(v[2], wv[4 * q + 2], acc[s]); acc[s] = fmaf(v[3], wv[4 * q + 3], acc[s]);
            }
    }
    for (; k < n; k += 4) {
        const float w0 = w[(size_t)k * stride], w1 = w[(size_t)(k + 1) * stride], w2 = w[(size_t)(k + 2) * stride], w3 = w[(size_t)(k + 3) * stride];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const f4 v = *reinterpret_cast<const f4*>(vec + s * vstride + k);
            acc[s] = fmaf(v[0], w0, acc[s]); acc[s] = fmaf(v[1], w1, acc[s]);
            acc[s] = fmaf(v[2], w2, acc[s]); acc[s] = fmaf(v[3], w3, acc[s]);
        }
    }
}


// Workgroup barrier that orders LDS traffic only (__syncthreads() also waits for every outstanding global store; a
// recurrence step stores its gates / r*h / output rows right before each barrier and never reads them back).  Measured:
// it makes no difference to the step time -- what did was requesting a step's global operands one step ahead.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------- GRU (training), weights partly resident
// One window per workgroup as above, but every thread keeps the first KRG / KRC weights of its slice of a gate /
// candidate column in registers for all T steps.  The streaming form is bound by what one CU can pull through its
// L1 (786 KB per step at H = 256 is >= 5 us at 64 B/clk); here H = 128 streams nothing (all 196 KB resident: 96
// registers per thread); H = 256 keeps half (KRG / KRC below the slice length leave the rest streaming).
// Thread roles (NT = 512):
//   gates:     column cg = tid / TG (2H columns), k-slice sg = tid % TG of KG = H / TG rows      (TG = 512 / 2H)
//   candidate: column cc = tid / TC (H columns),  k-slice sc = tid % TC of KC = H / TC rows      (TC = 512 / H)
// slices of one column sit in adjacent lanes and are summed with lane shuffles (no barrier, no LDS).
template <int H, int KRG, int KRC>
__global__ void __launch_bounds__(512, 1)
gru_train_fwd_res_kernel(GruTrainArgs a) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int NT = 512, H3 = 3 * H, TG = NT / (2 * H), KG = H / TG, TC = NT / H, KC = H / TC;
    static_assert(TG >= 1 && TC >= 1 && KRG <= KG && KRC <= KC && KRG % 4 == 0 && KRC % 4 == 0 && (KG - KRG) % 16 == 0 && (KC - KRC) % 16 == 0, "slices");
    __shared__ __attribute__((aligned(16))) float h[H];
    __shared__ __attribute__((aligned(16))) float rhs[H];
    __shared__ float us[H];
    const int tid = threadIdx.x;
    const int seq = blockIdx.x, dir = blockIdx.y;
    const float* W = a.Wh[dir];
    const size_t MT = (size_t)a.n_seq * a.T;
    float* gates = a.gates + (size_t)dir * MT * H3;
    float* rhg = a.rh + (size_t)dir * MT * H;
    const int cg = tid / TG, sg = tid % TG, cc = tid / TC, sc = tid % TC;
    const float* wgp = W + (size_t)(sg * KG) * H3 + cg;            // this thread's gate column slice, row stride H3
    const float* wcp = W + (size_t)(sc * KC) * H3 + 2 * H + cc;    // ... candidate column slice
    float wg[KRG > 0 ? KRG : 1], wc[KRC > 0 ? KRC : 1];
#pragma unroll
    for (int i = 0; i < KRG; ++i) wg[i] = wgp[(size_t)i * H3];
#pragma unroll
    for (int i = 0; i < KRC; ++i) wc[i] = wcp[(size_t)i * H3];
    for (int i = tid; i < H; i += NT) h[i] = 0.0f;
    lds_barrier();
    const size_t xrow = 6 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H3;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    // the input projections of a step are requested a whole step ahead: their L2 / HBM round trip (two per step) was
    // most of what was left of a step once the weights had become resident
    float xg = sg == 0 ? xbase[(size_t)t * xrow + cg] : 0.0f, xc = sc == 0 ? xbase[(size_t)t * xrow + 2 * H + cc] : 0.0f;
    for (int step = 0; step < a.T; ++step, t += dt) {
        const size_t row = (size_t)seq * a.T + t;
        const int tn = step + 1 < a.T ? t + dt : t;
        const float xg_n = sg == 0 ? xbase[(size_t)tn * xrow + cg] : 0.0f;
        const float xc_n = sc == 0 ? xbase[(size_t)tn * xrow + 2 * H + cc] : 0.0f;
        // ---- r, u
        float acc = xg;
        {
            const float* hv = h + sg * KG;
#pragma unroll
            for (int q = 0; q < KRG / 4; ++q) {
                const f4 v = *reinterpret_cast<const f4*>(hv + 4 * q);
                acc = fmaf(v[0], wg[4 * q], acc); acc = fmaf(v[1], wg[4 * q + 1], acc);
                acc = fmaf(v[2], wg[4 * q + 2], acc); acc = fmaf(v[3], wg[4 * q + 3], acc);
            }
#pragma unroll 1
            for (int k = KRG; k < KG; k += 16) {
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = wgp[(size_t)(k + u) * H3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 v = *reinterpret_cast<const f4*>(hv + k + 4 * q);
                    acc = fmaf(v[0], wv[4 * q], acc); acc = fmaf(v[1], wv[4 * q + 1], acc);
                    acc = fmaf(v[2], wv[4 * q + 2], acc); acc = fmaf(v[3], wv[4 * q + 3], acc);
                }
            }
        }
#pragma unroll
        for (int o = 1; o < TG; o <<= 1) acc += __shfl_xor(acc, o, 64);
        {
            const float g = sigmoidf_(acc);
            if (sg == 0) {
                gates[row * H3 + cg] = g;
                if (cg < H) { const float v = g * h[cg]; rhs[cg] = v; rhg[row * H + cg] = v; }
                else us[cg - H] = g;
            }
        }
        lds_barrier();
        // ---- candidate, update
        float ac = xc;
        {
            const float* rv = rhs + sc * KC;
#pragma unroll
            for (int q = 0; q < KRC / 4; ++q) {
                const f4 v = *reinterpret_cast<const f4*>(rv + 4 * q);
                ac = fmaf(v[0], wc[4 * q], ac); ac = fmaf(v[1], wc[4 * q + 1], ac);
                ac = fmaf(v[2], wc[4 * q + 2], ac); ac = fmaf(v[3], wc[4 * q + 3], ac);
            }
#pragma unroll 1
            for (int k = KRC; k < KC; k += 16) {
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = wcp[(size_t)(k + u) * H3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 v = *reinterpret_cast<const f4*>(rv + k + 4 * q);
                    ac = fmaf(v[0], wv[4 * q], ac); ac = fmaf(v[1], wv[4 * q + 1], ac);
                    ac = fmaf(v[2], wv[4 * q + 2], ac); ac = fmaf(v[3], wv[4 * q + 3], ac);
                }
            }
        }
#pragma unroll
        for (int o = 1; o < TC; o <<= 1) ac += __shfl_xor(ac, o, 64);
        const float c = tanhf(ac);
        const float u = us[cc];
        const float hn = u * h[cc] + (1.0f - u) * c;
        if (sc == 0) {
            gates[row * H3 + 2 * H + cc] = c;
            a.out[row * 2 * H + (size_t)dir * H + cc] = hn;
        }
        lds_barrier();                               // every read of h and rhs of this step is done
        if (sc == 0) h[cc] = hn;
        lds_barrier();
        xg = xg_n;
        xc = xc_n;
    }
}

static inline int gru_group_size(int n_seq) { return 2 * n_seq <= 256 ? 1 : (n_seq <= 256 ? 2 : 4); }

template <int GS>
__global__ void __launch_bounds__(512)
gru_train_fwd_ms_kernel(GruTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, H3 = 3 * H, NT = blockDim.x, tid = threadIdx.x;
    float* h = reinterpret_cast<float*>(smem);      // [GS][H]
    float* rhs = h + GS * H;                        // [GS][H]
    float* us = rhs + GS * H;                       // [GS][H]
    float* part = us + GS * H;                      // [P][GS][H] partial sums of the candidate matvec
    // the candidate has H columns for NT threads: split its reduction over P thread groups
    // (slices stay multiples of 4: the matvec reads float4 pieces of the state vector)
    int P = (NT >= H && NT % H == 0) ? NT / H : 1;
    while (P > 1 && (H / P) % 4 != 0) P >>= 1;
    const int pk = tid % H, pp = tid / H;
    const int seq0 = blockIdx.x * GS, dir = blockIdx.y;
    const float* W = a.Wh[dir];
    const size_t MT = (size_t)a.n_seq * a.T;
    float* gates = a.gates + (size_t)dir * MT * H3;
    float* rhg = a.rh + (size_t)dir * MT * H;
    for (int i = tid; i < GS * H; i += NT) h[i] = 0.0f;
    __syncthreads();
    const size_t xrow = 6 * (size_t)H;
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    int sq[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) sq[s] = min(seq0 + s, a.n_seq - 1);
    for (int step = 0; step < a.T; ++step, t += dt) {
        for (int col = tid; col < 2 * H; col += NT) {
            float acc[GS];
#pragma unroll
            for (int s = 0; s < GS; ++s) acc[s] = a.xproj[((size_t)sq[s] * a.T + t) * xrow + (size_t)dir * H3 + col];
            ms_matvec<GS>(W + col, (size_t)H3, h, H, H, acc);
#pragma unroll
            for (int s = 0; s < GS; ++s) {
                const float g = sigmoidf_(acc[s]);
                const size_t row = (size_t)sq[s] * a.T + t;
                if (seq0 + s < a.n_seq) gates[row * H3 + col] = g;
                if (col < H) {
                    const float v = g * h[s * H + col];
                    rhs[s * H + col] = v;
                    if (seq0 + s < a.n_seq) rhg[row * H + col] = v;
                } else {
                    us[s * H + col - H] = g;
                }
            }
        }
        __syncthreads();
        float hn[GS];
        if (P > 1) {                                     // reduction index split over P thread groups
            float pacc[GS];
#pragma unroll
            for (int s = 0; s < GS; ++s) pacc[s] = 0.0f;
            const int n = H / P, k0 = pp * n;
            if (pp < P) {
                ms_matvec<GS>(W + (size_t)k0 * H3 + 2 * H + pk, (size_t)H3, rhs + k0, H, n, pacc);
#pragma unroll
                for (int s = 0; s < GS; ++s) part[(pp * GS + s) * H + pk] = pacc[s];
            }
            __syncthreads();
        }
        for (int col = tid; col < H; col += NT) {
            float acc[GS];
#pragma unroll
            for (int s = 0; s < GS; ++s) acc[s] = a.xproj[((size_t)sq[s] * a.T + t) * xrow + (size_t)dir * H3 + 2 * H + col];
            if (P > 1) {
#pragma unroll
                for (int s = 0; s < GS; ++s)
                    for (int q = 0; q < P; ++q) acc[s] += part[(q * GS + s) * H + col];
            } else {
                ms_matvec<GS>(W + 2 * H + col, (size_t)H3, rhs, H, H, acc);
            }
#pragma unroll
            for (int s = 0; s < GS; ++s) {
                const float c = tanhf(acc[s]);
                const float u = us[s * H + col];
                hn[s] = u * h[s * H + col] + (1.0f - u) * c;
                if (seq0 + s < a.n_seq) {
                    const size_t row = (size_t)sq[s] * a.T + t;
                    gates[row * H3 + 2 * H + col] = c;
                    a.out[row * 2 * H + (size_t)dir * H + col] = hn[s];
                }
            }
        }
        __syncthreads();
        // H <= NT here (one candidate column per thread at most), so hn[] of this thread is its column's
        for (int col = tid; col < H; col += NT)
#pragma unroll
            for (int s = 0; s < GS; ++s) h[s * H + col] = hn[s];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------- LSTM (use_lstm), training mode
// tf.contrib.rnn.LSTMCell(num_units, forget_bias = 1.0) under bidirectional_dynamic_rnn (/root/reference/modules.py:207-243):
//   z = [x, h] W + b, (i, j, f, o) = split(z);  c' = c sig(f + 1) + sig(i) tanh(j);  h' = tanh(c') sig(o).
// No shipped configuration sets use_lstm, so these are plain any-size kernels (one workgroup per (window, direction), the
// recurrent weights streamed from L2), not tuned ones.  Forward saves the ACTIVATED gates (i | j | f | o) and the cell state.
struct LstmTrainArgs {
    const float* xproj;      // [n_seq*T, 8H]
    const float* Wh[2];      // [H, 4H]
    float* out;              // [n_seq*T, 2H]
    float* gates;            // [2][n_seq*T, 4H]
    float* cst;              // [2][n_seq*T, H]
    int32_t n_seq, T, H;
};

__global__ void __launch_bounds__(512)
lstm_train_fwd_kernel(LstmTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, H4 = 4 * H, NT = blockDim.x, tid = threadIdx.x;
    float* h = reinterpret_cast<float*>(smem);      // [H]
    float* z = h + H;                               // [4H]
    const int seq = blockIdx.x, dir = blockIdx.y;
    const float* W = a.Wh[dir];
    const size_t MT = (size_t)a.n_seq * a.T;
    float* gates = a.gates + (size_t)dir * MT * H4;
    float* cst = a.cst + (size_t)dir * MT * H;
    for (int i = tid; i < H; i += NT) h[i] = 0.0f;
    float c = 0.0f;                                 // cell state of unit tid (H <= NT: checked by the host)
    const size_t xrow = 8 * (size_t)H;
    const float* xbase = a.xproj + (size_t)seq * a.T * xrow + (size_t)dir * H4;
    __syncthreads();
    int t = dir ? a.T - 1 : 0;
    const int dt = dir ? -1 : 1;
    for (int step = 0; step < a.T; ++step, t += dt) {
        const float* xr = xbase + (size_t)t * xrow;
        for (int col = tid; col < H4; col += NT) {
            float acc = xr[col];
            const float* w = W + col;
#pragma unroll 4
            for (int k = 0; k < H; ++k) acc = fmaf(h[k], w[(size_t)k * H4], acc);
            z[col] = acc;
        }
        __syncthreads();
        if (tid < H) {
            const float gi = sigmoidf_(z[tid]), gj = tanhf(z[H + tid]);
            const float gf = sigmoidf_(z[2 * H + tid] + 1.0f), go = sigmoidf_(z[3 * H + tid]);
            c = gf * c + gi * gj;
            const float hn = go * tanhf(c);
            h[tid] = hn;
            const size_t row = (size_t)seq * a.T + t;
            gates[row * H4 + tid] = gi; gates[row * H4 + H + tid] = gj;
            gates[row * H4 + 2 * H + tid] = gf; gates[row * H4 + 3 * H + tid] = go;
            cst[row * H + tid] = c;
            a.out[row * 2 * H + (size_t)dir * H + tid] = hn;
        }
        __syncthreads();
    }
}

struct LstmBwdArgs {
    const float* dout;       // [n_seq*T, 2H]
    const float* gates;      // [2][n_seq*T, 4H]
    const float* cst;        // [2][n_seq*T, H]
    const float* WhT[2];     // [4H, H] transposed recurrent weights
    float* dpre;             // [n_seq*T, 8H]
    int32_t n_seq, T, H;
};

// BPTT: walks the steps of a direction in reverse; dpre = gradient w.r.t. the pre-activations z (the forget bias is a
// constant), in the layout of xproj.
__global__ void __launch_bounds__(512)
lstm_bwd_kernel(LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.H, H4 = 4 * H, tid = threadIdx.x;
    float* dz = reinterpret_cast<float*>(smem);     // [4H]
    const int seq = blockIdx.x, dir = blockIdx.y;
    const float* WT = a.WhT[dir];
    const size_t MT = (size_t)a.n_seq * a.T;
    const float* gates = a.gates + (size_t)dir * MT * H4;
    const float* cst = a.cst + (size_t)dir * MT * H;
    float dh_next = 0.0f, dc_next = 0.0f;           // unit tid's, from the step after this one (forward order)
    int t = dir ? 0 : a.T - 1;                      // last step of the forward pass first
    const int dt = dir ? 1 : -1;
    for (int step = 0; step < a.T; ++step, t += dt) {
        const size_t row = (size_t)seq * a.T + t;
        if (tid < H) {
            const float gi = gates[row * H4 + tid], gj = gates[row * H4 + H + tid];
            const float gf = gates[row * H4 + 2 * H + tid], go = gates[row * H4 + 3 * H + tid];
            const float c = cst[row * H + tid];
            const bool first = step == a.T - 1;                       // the forward pass's first step: c_prev = 0
            const float cp = first ? 0.0f : cst[((size_t)seq * a.T + (t + dt)) * H + tid];
            const float tc = tanhf(c);
            const float dh = a.dout[row * 2 * H + (size_t)dir * H + tid] + dh_next;
            const float dc = dh * go * (1.0f - tc * tc) + dc_next;
            const float d_o = dh * tc * go * (1.0f - go);
            const float d_i = dc * gj * gi * (1.0f - gi);
            const float d_j = dc * gi * (1.0f - gj * gj);
            const float d_f = dc * cp * gf * (1.0f - gf);
            dc_next = dc * gf;
            dz[tid] = d_i; dz[H + tid] = d_j; dz[2 * H + tid] = d_f; dz[3 * H + tid] = d_o;
            float* dp = a.dpre + row * 8 * (size_t)H + (size_t)dir * H4;
            dp[tid] = d_i; dp[H + tid] = d_j; dp[2 * H + tid] = d_f; dp[3 * H + tid] = d_o;
        }
        __syncthreads();
        if (tid < H) {
            float acc = 0.0f;
#pragma unroll 4
            for (int col = 0; col < H4; ++col) acc = fmaf(dz[col], WT[(size_t)col * H + tid], acc);
            dh_next = acc;
        }
        __syncthreads();
    }
}

struct GruBwdMsArgs {
    GruBwdArgs b;
    const float* WhT[2];     // [3H, H] transposed recurrent weights (coalesced matvecs with W^T)
};

template <int GS>
__global__ void __launch_bounds__(512)
gru_bwd_ms_kernel(GruBwdMsArgs aa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GruBwdArgs& a = aa.b;
    const int H = a.H, H3 = 3 * H, NT = blockDim.x, tid = threadIdx.x;
    float* dh = reinterpret_cast<float*>(smem);     // [GS][H]
    float* dcp = dh + GS * H;                       // [GS][H]
    float* dgp = dcp + GS * H;                      // [GS][2H]
    float* drh = dgp + GS * 2 * H;                  // [GS][H]
    float* part = drh + GS * H;                     // [P][GS][H] partial sums of the split reductions
    // A W^T matvec has only H outputs for NT threads: the reduction index is split over P = NT / H
    // thread groups (each sums a contiguous slice), partials meet in LDS.
    int P = (NT >= H && NT % H == 0) ? NT / H : 1;
    while (P > 1 && (H / P) % 4 != 0) P >>= 1;
    const int pk = tid % H, pp = tid / H;
    const int seq0 = blockIdx.x * GS, dir = blockIdx.y;
    const float* WT = aa.WhT[dir];                  // WT[col][k] = W[k][col]
    const size_t MT = (size_t)a.n_seq * a.T;
    const float* gates = a.gates + (size_t)dir * MT * H3;
    for (int i = tid; i < GS * H; i += NT) dh[i] = 0.0f;
    __syncthreads();
    int sq[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) sq[s] = min(seq0 + s, a.n_seq - 1);
    int t = dir ? 0 : a.T - 1;
    const int dtb = dir ? 1 : -1;
    for (int step = 0; step < a.T; ++step, t += dtb) {
        const bool first = (step == a.T - 1);
        for (int i = tid; i < GS * H; i += NT) {
            const int s = i / H, j = i - s * H;
            const size_t row = (size_t)sq[s] * a.T + t;
            const float g = dh[i] + a.dout[row * 2 * H + (size_t)dir * H + j];
            const float u = gates[row * H3 + H + j], c = gates[row * H3 + 2 * H + j];
            const float hp = first ? 0.0f : a.out[(row + dtb) * 2 * H + (size_t)dir * H + j];
            dcp[i] = g * (1.0f - u) * (1.0f - c * c);
            dgp[s * 2 * H + H + j] = g * (hp - c) * u * (1.0f - u);
            dh[i] = g * u;
        }
        __syncthreads();
        // d(rh)[k] = sum_j W[k][2H + j] dc_pre[j] = sum_j WT[2H + j][k] dc_pre[j]
        if (P > 1) {
            float acc[GS];
#pragma unroll
            for (int s = 0; s < GS; ++s) acc[s] = 0.0f;
            const int n = H / P, j0 = pp * n;
            if (pp < P) {
                ms_matvec<GS>(WT + (size_t)(2 * H + j0) * H + pk, (size_t)H, dcp + j0, H, n, acc);
#pragma unroll
                for (int s = 0; s < GS; ++s) part[(pp * GS + s) * H + pk] = acc[s];
            }
            __syncthreads();
            for (int i = tid; i < GS * H; i += NT) {
                float v = part[i];
                for (int q = 1; q < P; ++q) v += part[q * GS * H + i];
                drh[i] = v;
            }
        } else {
            for (int k = tid; k < H; k += NT) {
                float acc[GS];
#pragma unroll
                for (int s = 0; s < GS; ++s) acc[s] = 0.0f;
                ms_matvec<GS>(WT + (size_t)2 * H * H + k, (size_t)H, dcp, H, H, acc);
#pragma unroll
                for (int s = 0; s < GS; ++s) drh[s * H + k] = acc[s];
            }
        }
        __syncthreads();
        for (int i = tid; i < GS * H; i += NT) {
            const int s = i / H, j = i - s * H;
            const size_t row = (size_t)sq[s] * a.T + t;
            const float r = gates[row * H3 + j];
            const float hp = first ? 0.0f : a.out[(row + dtb) * 2 * H + (size_t)dir * H + j];
            dgp[s * 2 * H + j] = drh[i] * hp * r * (1.0f - r);
            dh[i] += drh[i] * r;
        }
        __syncthreads();
        // dh_prev[k] += sum_{j < 2H} W[k][j] dg_pre[j] = sum_j WT[j][k] dg_pre[j]
        if (P > 1) {
            float acc[GS];
#pragma unroll
            for (int s = 0; s < GS; ++s) acc[s] = 0.0f;
            const int n = 2 * H / P, j0 = pp * n;
            if (pp < P) {
                ms_matvec<GS>(WT + (size_t)j0 * H + pk, (size_t)H, dgp + j0, 2 * H, n, acc);
#pragma unroll
                for (int s = 0; s < GS; ++s) part[(pp * GS + s) * H + pk] = acc[s];
            }
            __syncthreads();
            for (int i = tid; i < GS * H; i += NT) {
                float v = part[i];
                for (int q = 1; q < P; ++q) v += part[q * GS * H + i];
                dh[i] += v;
            }
        } else {
            for (int k = tid; k < H; k += NT) {
                float acc[GS];
#pragma unroll
                for (int s = 0; s < GS; ++s) acc[s] = 0.0f;
                ms_matvec<GS>(WT + k, (size_t)H, dgp, 2 * H, 2 * H, acc);
#pragma unroll
                for (int s = 0; s < GS; ++s) dh[s * H + k] += acc[s];
            }
        }
        for (int i = tid; i < GS * 3 * H; i += NT) {
            const int s = i / H3, j = i - s * H3;
            if (seq0 + s < a.n_seq) {
                const size_t row = (size_t)sq[s] * a.T + t;
                a.dpre[row * 6 * (size_t)H + (size_t)dir * H3 + j] = j < 2 * H ? dgp[s * 2 * H + j] : dcp[s * H + j - 2 * H];
            }
        }
        __syncthreads();
    }
}


// BPTT with the recurrent weights resident in registers (H = 128: 32 + 64 per thread), one window per workgroup; the
// same steps as gru_bwd_ms_kernel<1>.  Thread (k = tid / 4, slice = tid % 4) holds rows [32 slice, +32) of candidate
// column k and rows [64 slice, +64) of gate column k of W^T; the four slices of a k are adjacent lanes (shuffle sum).
template <int H>
__global__ void __launch_bounds__(512, 1)
gru_bwd_res_kernel(GruBwdMsArgs aa) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const GruBwdArgs& a = aa.b;
    constexpr int NT = 512, H3 = 3 * H, TP = NT / H, N1 = H / TP, N2 = 2 * H / TP;
    static_assert(TP == 4 && N1 % 4 == 0 && N2 % 4 == 0, "H = 128");
    __shared__ __attribute__((aligned(16))) float dh[H];
    __shared__ __attribute__((aligned(16))) float dcp[H];
    __shared__ __attribute__((aligned(16))) float dgp[2 * H];
    __shared__ float drh[H];
    const int tid = threadIdx.x, k = tid / TP, sl = tid % TP;
    const int seq = blockIdx.x, dir = blockIdx.y;
    const float* WT = aa.WhT[dir];                  // WT[col][k] = W[k][col]
    const size_t MT = (size_t)a.n_seq * a.T;
    const float* gates = a.gates + (size_t)dir * MT * H3;
    float w1[N1], w2[N2];
#pragma unroll
    for (int i = 0; i < N1; ++i) w1[i] = WT[(size_t)(2 * H + sl * N1 + i) * H + k];
#pragma unroll
    for (int i = 0; i < N2; ++i) w2[i] = WT[(size_t)(sl * N2 + i) * H + k];
    for (int i = tid; i < H; i += NT) dh[i] = 0.0f;
    lds_barrier();
    int t = dir ? 0 : a.T - 1;
    const int dtb = dir ? 1 : -1;
    // a step's operands (dout, the saved gates, the previous state) are requested a whole step ahead
    float p_do = 0.0f, p_r = 0.0f, p_u = 0.0f, p_c = 0.0f, p_hp = 0.0f;
    auto fetch = [&](int tt, bool is_first) {
        if (tid < H) {
            const size_t rw = (size_t)seq * a.T + tt;
            p_do = a.dout[rw * 2 * H + (size_t)dir * H + tid];
            p_r = gates[rw * H3 + tid]; p_u = gates[rw * H3 + H + tid]; p_c = gates[rw * H3 + 2 * H + tid];
            p_hp = is_first ? 0.0f : a.out[(rw + dtb) * 2 * H + (size_t)dir * H + tid];
        }
    };
    fetch(t, a.T == 1);
    for (int step = 0; step < a.T; ++step, t += dtb) {
        const size_t row = (size_t)seq * a.T + t;
        const float c_do = p_do, c_r = p_r, c_u = p_u, c_c = p_c, c_hp = p_hp;
        if (step + 1 < a.T) fetch(t + dtb, step + 1 == a.T - 1);
        if (tid < H) {
            const int j = tid;
            const float g = dh[j] + c_do;
            const float u = c_u, c = c_c;
            const float hp = c_hp;
            dcp[j] = g * (1.0f - u) * (1.0f - c * c);
            dgp[H + j] = g * (hp - c) * u * (1.0f - u);
            dh[j] = g * u;
        }
        lds_barrier();
        {   // d(rh)[k] = sum_j W[k][2H + j] dc_pre[j]
            float acc = 0.0f;
            const float* v = dcp + sl * N1;
#pragma unroll
            for (int q = 0; q < N1 / 4; ++q) {
                const f4 x = *reinterpret_cast<const f4*>(v + 4 * q);
                acc = fmaf(x[0], w1[4 * q], acc); acc = fmaf(x[1], w1[4 * q + 1], acc);
                acc = fmaf(x[2], w1[4 * q + 2], acc); acc = fmaf(x[3], w1[4 * q + 3], acc);
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (sl == 0) drh[k] = acc;
        }
        lds_barrier();
        if (tid < H) {
            const int j = tid;
            const float r = c_r;
            const float hp = c_hp;
            dgp[j] = drh[j] * hp * r * (1.0f - r);
            dh[j] += drh[j] * r;
        }
        lds_barrier();
        {   // dh_prev[k] += sum_{j < 2H} W[k][j] dg_pre[j]
            float acc = 0.0f;
            const float* v = dgp + sl * N2;
#pragma unroll
            for (int q = 0; q < N2 / 4; ++q) {
                const f4 x = *reinterpret_cast<const f4*>(v + 4 * q);
                acc = fmaf(x[0], w2[4 * q], acc); acc = fmaf(x[1], w2[4 * q + 1], acc);
                acc = fmaf(x[2], w2[4 * q + 2], acc); acc = fmaf(x[3], w2[4 * q + 3], acc);
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (sl == 0) dh[k] += acc;
        }
        if (tid < H3) a.dpre[row * 6 * (size_t)H + (size_t)dir * H3 + tid] = tid < 2 * H ? dgp[tid] : dcp[tid - 2 * H];
        lds_barrier();
    }
}


// ---------------------------------------------------------------------------- encoder loss / metrics
// tf.nn.softmax_cross_entropy_with_logits_v2 + reduce_mean (/root/reference/encoder.py:134-137),
// accuracy of argmax(pred) vs argmax(target) and mean squared error of the posteriors
// (encoder.py:143-150).  One wave per row; per-row values go to a workspace and are summed by
// one block in a fixed order (deterministic).  dlogits = (p * sum(t) - t) / M.
__global__ void __launch_bounds__(256)
softmax_ce_rows_kernel(const float* logits, const float* target, int M, int C, int ldl, float* dlogits, int ldd,
                       float* rowvals) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* x = logits + (size_t)row * ldl;
    const float* t = target + (size_t)row * C;
    float mx = -3.402823466e38f, tmx = -3.402823466e38f;
    int mi = 0x7fffffff, tmi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = x[c], tv = t[c];
        if (v > mx) { mx = v; mi = c; }
        if (tv > tmx) { tmx = tv; tmi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
        const float tv = __shfl_xor(tmx, o, 64); const int ti = __shfl_xor(tmi, o, 64);
        if (tv > tmx || (tv == tmx && ti < tmi)) { tmx = tv; tmi = ti; }
    }
    float se = 0.0f, st = 0.0f, stx = 0.0f;
    for (int c = lane; c < C; c += 64) {
        se += expf(x[c] - mx);
        st += t[c];
        stx = fmaf(t[c], x[c] - mx, stx);
    }
    se = vc::wave_sum(se); st = vc::wave_sum(st); stx = vc::wave_sum(stx);
    const float lse = logf(se);
    const float inv = 1.0f / se, invM = 1.0f / (float)M;
    float sq = 0.0f;
    for (int c = lane; c < C; c += 64) {
        const float pc = expf(x[c] - mx) * inv;
        const float d = pc - t[c];
        sq = fmaf(d, d, sq);
        if (dlogits) dlogits[(size_t)row * ldd + c] = (pc * st - t[c]) * invM;
    }
    sq = vc::wave_sum(sq);
    if (lane == 0) {
        rowvals[row] = st * lse - stx;                       // -sum t log p
        rowvals[(size_t)M + row] = (mi == tmi) ? 1.0f : 0.0f;
        rowvals[2 * (size_t)M + row] = sq;
    }
}
__global__ void __launch_bounds__(256)
reduce3_kernel(const float* rowvals, int M, int C, float* out) {
    __shared__ double red[3][256];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int r = threadIdx.x; r < M; r += 256) {
        a0 += (double)rowvals[r]; a1 += (double)rowvals[(size_t)M + r]; a2 += (double)rowvals[2 * (size_t)M + r];
    }
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(red[0][0] / M);                     // loss
        out[1] = (float)(red[1][0] / M);                     // accuracy
        out[2] = (float)(red[2][0] / ((double)M * C));       // mse
    }
}

inline int nblocks(size_t n) {
    size_t b = (n + TB - 1) / TB;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

// Kernel-layout copies of MANY convolution weights in one launch (the training step rebuilds them after every Adam
// update: as torch transposes / flips that was ~200 five-microsecond launches per step).  Item i: TF kernel
// src [k, cin, cout] -> mode 0: dst [cout, k*cin]  (the forward operand of vc_conv_gemm: K contiguous),
//                       mode 1: dst [cin, k*cout] with the taps reversed (the data-gradient convolution's operand).
// grid (item, tile runner); 32 x 32 tiles through LDS for mode 0, straight coalesced rows for mode 1.
__global__ void __launch_bounds__(256)
weight_layout_kernel(const vc_layout_item* items) {
    const vc_layout_item it = items[blockIdx.x];
    const float* src = it.src;
    float* dst = it.dst;
    const int k = it.k, cin = it.cin, cout = it.cout;
    if (it.mode == 0) {
        __shared__ float tile[32][33];
        const int rows = k * cin, cols = cout;                      // src [rows, cols] -> dst [cols, rows]
        const int tr = (rows + 31) / 32, tc = (cols + 31) / 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
        for (int t = blockIdx.y; t < tr * tc; t += gridDim.y) {
            const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + ty + 8 * i, c = c0 + tx;
                tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.0f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + ty + 8 * i, r = r0 + tx;
                if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[tx][ty + 8 * i];
            }
            __syncthreads();
        }
    } else {
        const size_t n = (size_t)k * cin * cout;
        for (size_t idx = (size_t)blockIdx.y * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.y * 256) {
            const int co = (int)(idx % cout);
            const size_t rest = idx / cout;
            const int jj = (int)(rest % k), ci = (int)(rest / k);   // dst[ci][jj][co]
            dst[idx] = src[((size_t)(k - 1 - jj) * cin + ci) * cout + co];
        }
    }
}

}  // namespace

extern "C" {

size_t vc_stats_workspace_floats(int32_t M, int32_t C) {
    const int rows = BN_ROWS;
    const int nblk = (M + rows - 1) / rows;
    return (size_t)nblk * 2 * C;
}

int vc_bn_train_stats(const float* d_X, int32_t M, int32_t C, int32_t ld, const float* d_gamma, const float* d_beta,
                      float* d_moving_mean, float* d_moving_var, float decay, float eps, float* d_scale,
                      float* d_shift, float* d_mean, float* d_rstd, float* d_workspace, void* stream) {
    VC_REQUIRE(d_X && d_gamma && d_beta && d_scale && d_shift && d_mean && d_rstd && d_workspace, "NULL argument");
    VC_REQUIRE(M > 0 && C > 0 && ld >= C, "bad shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = BN_ROWS, nblk = (M + rows - 1) / rows;
    hipLaunchKernelGGL(col_stats_partial_kernel, dim3((C + TB - 1) / TB, nblk), dim3(TB), 0, st, d_X, M, C, ld, rows, d_workspace);
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3((C + 31) / 32), dim3(TB), 0, st, d_workspace, nblk, M, C, d_gamma,
                       d_beta, d_moving_mean, d_moving_var, decay, eps, d_scale, d_shift, d_mean, d_rstd);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_affine_act(const float* d_X, const float* d_scale, const float* d_shift, int32_t relu, const float* d_R,
                  float* d_out, size_t n, int32_t C, void* stream) {
    VC_REQUIRE(d_X && d_out && C > 0, "bad argument");
    hipLaunchKernelGGL(affine_act_kernel, dim3(nblocks(n)), dim3(TB), 0, static_cast<hipStream_t>(stream), d_X, d_scale,
                       d_shift, relu, d_R, d_out, n, C);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_bn_backward(const float* d_G, const float* d_X, int32_t M, int32_t C, int32_t ld, int32_t T, const float* d_gamma,
                   const float* d_scale, const float* d_shift, const float* d_mean, const float* d_rstd, int32_t mode,
                   float* d_dX, float* d_dgamma, float* d_dbeta, float* d_workspace, void* stream) {
    VC_REQUIRE(d_G && d_X && d_gamma && d_scale && d_shift && d_mean && d_rstd && d_dX && d_dgamma && d_dbeta && d_workspace,
               "NULL argument");
    VC_REQUIRE(M > 0 && C > 0 && ld >= C && T > 0 && M % T == 0 && mode >= 0 && mode <= 2, "bad shape/mode");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = BN_ROWS, nblk = (M + rows - 1) / rows;
    const dim3 grid((C + TB - 1) / TB, nblk);
    hipLaunchKernelGGL(bn_bwd_pass_kernel<false>, grid, dim3(TB), 0, st, d_G, d_X, M, C, ld, T, d_scale, d_shift, d_mean, d_rstd,
                       mode, rows, d_workspace, d_gamma, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((C + 31) / 32), dim3(TB), 0, st, d_workspace, nblk, C, d_dbeta, d_dgamma);
    hipLaunchKernelGGL(bn_bwd_pass_kernel<true>, grid, dim3(TB), 0, st, d_G, d_X, M, C, ld, T, d_scale, d_shift, d_mean, d_rstd,
                       mode, rows, (float*)nullptr, d_gamma, (const float*)d_dbeta, (const float*)d_dgamma, d_dX);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_bn_post_routing(const float* d_X, int32_t M, int32_t C, int32_t ld, int32_t T, const float* d_scale,
                       const float* d_shift, uint8_t* d_bits, void* stream) {
    VC_REQUIRE(d_X && d_scale && d_shift && d_bits, "NULL argument");
    VC_REQUIRE(M > 0 && C > 0 && ld >= C && T > 0 && M % T == 0, "bad shape");
    hipLaunchKernelGGL(bn_routing_kernel, dim3(nblocks((size_t)M * C)), dim3(TB), 0, static_cast<hipStream_t>(stream), d_X,
                       M, C, ld, T, d_scale, d_shift, d_bits);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_relu_dropout_backward(const float* d_dY, const float* d_Y, float inv_keep, float* d_dZ, size_t n, void* stream) {
    VC_REQUIRE(d_dY && d_Y && d_dZ, "NULL argument");
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(nblocks(n)), dim3(TB), 0, static_cast<hipStream_t>(stream), d_dY, d_Y,
                       inv_keep, d_dZ, n);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_highway_backward(const float* d_pre, int32_t NP, const float* d_X, const float* d_dO, int32_t M, int32_t H,
                        float* d_dpre, float* d_dXd, void* stream) {
    VC_REQUIRE(d_pre && d_X && d_dO && d_dpre && d_dXd, "NULL argument");
    VC_REQUIRE(NP == 64 * ((H + 31) / 32), "NP must be the paired width 64*ceil(H/32)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (NP != 2 * H) hipLaunchKernelGGL(fill_kernel, dim3(nblocks((size_t)M * NP)), dim3(TB), 0, st, d_dpre, 0.0f, (size_t)M * NP);
    hipLaunchKernelGGL(highway_bwd_kernel, dim3(nblocks((size_t)M * H)), dim3(TB), 0, st, d_pre, NP, d_X, d_dO, M, H, d_dpre, d_dXd);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_transpose_pad(const float* d_X, int32_t M, int32_t C, int32_t ld, int32_t T, const float* d_scale,
                     const float* d_shift, int32_t relu, int32_t pool, int32_t row_shift, float* d_XT, int32_t ldt,
                     int32_t pad, int32_t slack_row, void* stream) {
    VC_REQUIRE(d_X && d_XT, "NULL argument");
    VC_REQUIRE(M > 0 && C > 0 && ld >= C && T > 0 && ldt >= M + 2 * pad && pad >= 0, "bad shape");
    VC_REQUIRE(!(d_scale == nullptr) == !(d_shift == nullptr), "scale and shift go together");
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((M + 31) / 32, (C + 31) / 32), dim3(TB), 0, static_cast<hipStream_t>(stream),
                       d_X, M, C, ld, T, d_scale, d_shift, relu, pool, row_shift, d_XT, ldt, pad, slack_row);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_col_sum(const float* d_X, int32_t M, int32_t C, int32_t ld, float* d_out, int32_t accumulate,
               float* d_workspace, void* stream) {
    VC_REQUIRE(d_X && d_out && d_workspace && M > 0 && C > 0 && ld >= C, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nrb = 64;                                   // workspace: 64 * C floats
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3((C + 63) / 64, nrb), dim3(TB), 0, st, d_X, M, C, ld, nrb, d_workspace);
    hipLaunchKernelGGL(col_sum_final_kernel, dim3((C + 31) / 32), dim3(TB), 0, st, d_workspace, nrb, C, d_out, accumulate);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_fill(float* d_p, float value, size_t n, void* stream) {
    VC_REQUIRE(d_p, "NULL argument");
    if (n) hipLaunchKernelGGL(fill_kernel, dim3(nblocks(n)), dim3(TB), 0, static_cast<hipStream_t>(stream), d_p, value, n);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_weight_layouts(const vc_layout_item* d_items, int32_t n_items, void* stream) {
    VC_REQUIRE(d_items && n_items > 0 && n_items <= 65535, "vc_weight_layouts: bad argument");
    hipLaunchKernelGGL(weight_layout_kernel, dim3((unsigned)n_items, 16), dim3(256), 0, static_cast<hipStream_t>(stream), d_items);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_axpby(const float* d_X, int32_t ldx, float a, const float* d_Y, int32_t ldy, float b, float* d_out, int32_t ldo,
             int32_t M, int32_t C, void* stream) {
    VC_REQUIRE(d_X && d_Y && d_out, "NULL argument");
    VC_REQUIRE(M > 0 && C > 0 && ldx >= C && ldy >= C && ldo >= C, "bad shape M=%d C=%d ld=%d/%d/%d", M, C, ldx, ldy, ldo);
    hipLaunchKernelGGL(axpby_kernel, dim3(nblocks((size_t)M * C)), dim3(TB), 0, static_cast<hipStream_t>(stream), d_X, ldx, a,
                       d_Y, ldy, b, d_out, ldo, M, C);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_mse_loss(const float* d_y, const float* d_t, size_t n, float weight, float* d_dY, int32_t C, int32_t ld_dy,
                float* d_loss, float* d_workspace, void* stream) {
    VC_REQUIRE(d_y && d_t && d_loss && d_workspace && n > 0 && C > 0 && ld_dy >= C, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = 256;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(TB), 0, st, d_y, d_t, n, weight, d_dY, C, ld_dy, d_workspace);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1), 0, st, d_workspace, nb, n, weight, d_loss);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_adam_step(float* d_param, const float* d_grad, float* d_m, float* d_v, size_t n, float lr_t, float beta1,
                 float beta2, float epsilon, float grad_scale, void* stream) {
    VC_REQUIRE(d_param && d_grad && d_m && d_v, "NULL argument");
    if (n) hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(TB), 0, static_cast<hipStream_t>(stream), d_param, d_grad,
                              d_m, d_v, n, lr_t, beta1, beta2, epsilon, grad_scale);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_gru_train_forward(const float* d_xproj, const float* d_Wh_fw, const float* d_Wh_bw, int32_t n_seq, int32_t T,
                         int32_t H, float* d_out, float* d_gates, float* d_rh, void* stream) {
    VC_REQUIRE(d_xproj && d_Wh_fw && d_Wh_bw && d_out && d_gates && d_rh, "NULL argument");
    VC_REQUIRE(n_seq > 0 && T > 0 && H > 0 && H <= 1024, "bad shape");
    GruTrainArgs a;
    a.xproj = d_xproj; a.Wh[0] = d_Wh_fw; a.Wh[1] = d_Wh_bw; a.out = d_out; a.gates = d_gates; a.rh = d_rh;
    a.n_seq = n_seq; a.T = T; a.H = H;
    const int nt = H >= 128 ? 512 : 256;
    if (H % 4 == 0 && H <= nt) {
        // GS windows per workgroup.  The step is latency-bound, so spreading the windows over more
        // CUs beats amortising the weight stream: one window per workgroup while 2 * n_seq workgroups
        // fit the chip once (62.4 -> 53.7 ms/step at 32 windows), more only beyond that.
        const int gs = gru_group_size(n_seq);
        const size_t lds = (3 + (size_t)(nt / H)) * gs * H * 4;
        const dim3 grid((n_seq + gs - 1) / gs, 2);
        hipStream_t st = static_cast<hipStream_t>(stream);
        // H = 128: all 196 KB of weights in registers and the step's input projections requested a step ahead: 1.38 -> 0.53 ms
        // per launch at 32 windows.  H = 256: half of the 786 KB resident (192 registers per thread): 3.14 -> 2.95 ms -- the
        // step there is mostly NOT the weight stream (2.5 ms remain with the loads removed: tools/ab_gru_train.py --floor).
        const bool resident = gs == 1 && vc::opt(vc::OPT_GRU_TRAIN_RESIDENT) != 0;
        if (resident && H == 128) hipLaunchKernelGGL((gru_train_fwd_res_kernel<128, 64, 32>), grid, dim3(512), 0, st, a);
        else if (resident && H == 256) hipLaunchKernelGGL((gru_train_fwd_res_kernel<256, 128, 64>), grid, dim3(512), 0, st, a);
        else if (gs == 1) hipLaunchKernelGGL(gru_train_fwd_ms_kernel<1>, grid, dim3(nt), lds, st, a);
        else if (gs == 2) hipLaunchKernelGGL(gru_train_fwd_ms_kernel<2>, grid, dim3(nt), lds, st, a);
        else hipLaunchKernelGGL(gru_train_fwd_ms_kernel<4>, grid, dim3(nt), lds, st, a);
    } else {
        hipLaunchKernelGGL(gru_train_fwd_kernel, dim3(n_seq, 2), dim3(nt), 3 * (size_t)H * 4, static_cast<hipStream_t>(stream), a);
    }
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

#ifdef VC_ABLATE
int vc_ablate_set_gru_train(int32_t v) {
    VC_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gru_train_ablate), &v, sizeof(int)));
    return VC_OK;
}
#endif

int vc_gru_backward(const float* d_dout, const float* d_out, const float* d_gates, const float* d_Wh_fw,
                    const float* d_Wh_bw, const float* d_WhT_fw, const float* d_WhT_bw, int32_t n_seq, int32_t T,
                    int32_t H, float* d_dpre, void* stream) {
    VC_REQUIRE(d_dout && d_out && d_gates && d_Wh_fw && d_Wh_bw && d_dpre, "NULL argument");
    VC_REQUIRE(n_seq > 0 && T > 0 && H > 0 && H <= 1024, "bad shape");
    GruBwdArgs a;
    a.dout = d_dout; a.out = d_out; a.gates = d_gates; a.Wh[0] = d_Wh_fw; a.Wh[1] = d_Wh_bw; a.dpre = d_dpre;
    a.n_seq = n_seq; a.T = T; a.H = H;
    const int nt = H >= 128 ? 512 : 256;
    if (d_WhT_fw && d_WhT_bw && H % 4 == 0) {             // multi-sequence kernel with transposed weights
        GruBwdMsArgs aa;
        aa.b = a; aa.WhT[0] = d_WhT_fw; aa.WhT[1] = d_WhT_bw;
        const int gs = gru_group_size(n_seq);
        const size_t lds = (5 + (size_t)(nt >= H ? nt / H : 1)) * gs * H * 4;
        const dim3 grid((n_seq + gs - 1) / gs, 2);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (gs == 1 && H == 128 && vc::opt(vc::OPT_GRU_TRAIN_RESIDENT) != 0)
            hipLaunchKernelGGL(gru_bwd_res_kernel<128>, grid, dim3(512), 0, st, aa);
        else if (gs == 1) hipLaunchKernelGGL(gru_bwd_ms_kernel<1>, grid, dim3(nt), lds, st, aa);
        else if (gs == 2) hipLaunchKernelGGL(gru_bwd_ms_kernel<2>, grid, dim3(nt), lds, st, aa);
        else hipLaunchKernelGGL(gru_bwd_ms_kernel<4>, grid, dim3(nt), lds, st, aa);
    } else
        hipLaunchKernelGGL(gru_bwd_kernel, dim3(n_seq, 2), dim3(nt), 5 * (size_t)H * 4, static_cast<hipStream_t>(stream), a);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_lstm_train_forward(const float* d_xproj, const float* d_Wh_fw, const float* d_Wh_bw, int32_t n_seq, int32_t T,
                          int32_t H, float* d_out, float* d_gates, float* d_cstate, void* stream) {
    VC_REQUIRE(d_xproj && d_Wh_fw && d_Wh_bw && d_out && d_gates && d_cstate, "NULL argument");
    VC_REQUIRE(n_seq > 0 && n_seq <= 65535 && T > 0 && H > 0 && H <= 512, "vc_lstm_train_forward: bad shape n_seq=%d T=%d H=%d (H <= 512)", n_seq, T, H);
    LstmTrainArgs a;
    a.xproj = d_xproj; a.Wh[0] = d_Wh_fw; a.Wh[1] = d_Wh_bw; a.out = d_out; a.gates = d_gates; a.cst = d_cstate;
    a.n_seq = n_seq; a.T = T; a.H = H;
    hipLaunchKernelGGL(lstm_train_fwd_kernel, dim3(n_seq, 2), dim3(512), 5 * (size_t)H * 4, static_cast<hipStream_t>(stream), a);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_lstm_backward(const float* d_dout, const float* d_gates, const float* d_cstate, const float* d_WhT_fw,
                     const float* d_WhT_bw, int32_t n_seq, int32_t T, int32_t H, float* d_dpre, void* stream) {
    VC_REQUIRE(d_dout && d_gates && d_cstate && d_WhT_fw && d_WhT_bw && d_dpre, "NULL argument");
    VC_REQUIRE(n_seq > 0 && n_seq <= 65535 && T > 0 && H > 0 && H <= 512, "vc_lstm_backward: bad shape n_seq=%d T=%d H=%d (H <= 512)", n_seq, T, H);
    LstmBwdArgs a;
    a.dout = d_dout; a.gates = d_gates; a.cst = d_cstate; a.WhT[0] = d_WhT_fw; a.WhT[1] = d_WhT_bw; a.dpre = d_dpre;
    a.n_seq = n_seq; a.T = T; a.H = H;
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(n_seq, 2), dim3(512), 4 * (size_t)H * 4, static_cast<hipStream_t>(stream), a);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

int vc_softmax_ce(const float* d_logits, const float* d_target, int32_t M, int32_t C, int32_t ldl, float* d_dlogits,
                  int32_t ldd, float* d_out3, float* d_workspace, void* stream) {
    VC_REQUIRE(d_logits && d_target && d_out3 && d_workspace, "NULL argument");
    VC_REQUIRE(M > 0 && C > 0 && ldl >= C && (d_dlogits == nullptr || ldd >= C), "bad shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, d_logits, d_target, M, C, ldl, d_dlogits, ldd,
                       d_workspace);
    hipLaunchKernelGGL(reduce3_kernel, dim3(1), dim3(256), 0, st, d_workspace, M, C, d_out3);
    VC_HIP_CHECK(hipGetLastError());
    return VC_OK;
}

}  // extern "C"
