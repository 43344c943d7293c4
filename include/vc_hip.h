/* vc_hip.h -- C ABI of libvc_hip.so, the MI355X (gfx950) kernels behind the voice-conversion
 * hot path of socom20/speech-cloner.
 *
 * The reference has no FFI: its boundary is the Python construct-and-run API
 * (audio_lib.calc_MFCC_input, encoder.encoder_spec_phn, decoder.decoder_specs), whose
 * arithmetic lives in librosa/scipy/TensorFlow-1.9 ops.  Each entry point below replaces the
 * third-party op(s) one reference call site lowers to; the citation is the reference file:line.
 * The Python modules in speech-cloner_amd/ (same names as the reference's) bind these with
 * ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory (hipMalloc / torch.cuda tensor data_ptr);
 *     pointers named h_* are host memory; plain sizes are element counts unless "_bytes".
 *   - `stream` is a hipStream_t passed as void*; no entry point synchronises the stream or
 *     allocates device memory except the create / destroy calls, so every launch call is
 *     hipGraph-capturable.
 *   - return value: 0 = VC_OK, otherwise a VC_ERR_* code; vc_last_error() gives the message of
 *     the calling thread's last failure.  Nothing throws across the boundary.
 *   - tensors are row-major; activations are [N, T, C] (time-major inside a window, channels
 *     contiguous) exactly as the reference's TF tensors.
 */
#ifndef VC_HIP_H
#define VC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_OK 0
#define VC_ERR_INVALID 1      /* bad argument (shape, null pointer, unsupported size) */
#define VC_ERR_HIP 2          /* a HIP runtime call failed */
#define VC_ERR_WORKSPACE 3    /* workspace too small */
#define VC_ERR_UNSUPPORTED 4

/* 2: vc_frontend_f32 / vc_frontend_stages_f32 take out_rows, vc_transpose_pad takes slack_row, vc_gemm_desc has
 *    sum_groups.  Bump on EVERY change of an exported signature or struct layout: the Python binding (_vc.py) refuses
 *    to load a library whose vc_version() differs from its own constant.
 * 3: vc_gemm_desc ends with d_workspace / workspace_bytes (vc_conv_gemm_workspace_bytes); vc_bn_post_routing added.
 * 4: vc_split16 / vc_weights16 / vc_gemm16 (training convolutions on split-float16 operands). */
#define VC_ABI_VERSION 4

int vc_version(void);
const char* vc_last_error(void);
/* Name of the gfx target the code object was built for ("gfx950"). */
const char* vc_target_arch(void);

/* Kernel-selection options.  The library never reads the process environment; the only way to steer which of
 * two equivalent HIP kernels a launch takes (A/B measurements, regression tests) is this call.  Values: -1 = the
 * library's own choice (default).  Names:
 *   "bank256"        0 = filter banks on conv_kernel instead of bank256_kernel
 *   "bank256_xcd"    0 = plain 2-D grid, 1 = whole filter-width pairs per XCD, 2 = pairs split over two XCDs
 *   "conv256"        0 = long-K single filters on conv_kernel / gemm_kernel
 *   "conv256_min_k"  shortest K that takes conv256_kernel (default 384)
 *   "conv256_wm"     2 = keep 128-row blocks for 128-column launches
 *   "proj256"        0 = the 256-channel k = 3 projection on conv256_kernel instead of the bank tiles
 *   "proj256_split"  0 = never split that projection's K over two workgroups per row tile (see d_workspace)
 *   "wgrad_xcd"      0 = weight-gradient tiles dealt round-robin to the XCDs
 *   "gru_mfma"       0 = VALU recurrence always, 1 = MFMA recurrence always (default: from 32 sequences up)
 *   "fe_fused"       0 = the shipped front-end configuration as two launches (statistics pass, feature pass) instead of ONE
 *                    (every frame transformed once; blocks wait for the summary of their own utterance)
 *   "fe_fused_spin"  polls a block of the one-launch front-end waits for the other tiles of its utterance before it
 *                    computes their records itself (default 4000, ~4 ms); 0 = never wait (tests of that path)
 *   "gru_small_mfma" 1 = the encoder's H = 40 bf16 recurrence with 16 sequences per wave on MFMA instead of one wave per
 *                    sequence (measured slower; default: off)
 *   "gru_mfma4"      1 = the four-wave MFMA recurrence with all weights in registers (bit-identical, measured slower)
 *   "prenet_lds"     0 = every wave of the fused prenet streams the weights from L2 itself (default: one stream per
 *                    block, shared through LDS)
 *   "gru_train_resident" 0 = the float32 training recurrences stream all their weights from L2 every step (default:
 *                    128 units: all of them in registers, forward and backward; 256 units: half, forward)
 *   "cbhg_front_mi"  4 = 128-row blocks in the fused encoder front
 *   "f32_f16x3"      0 = float32 INFERENCE keeps its filter banks / post-bank projections on the f32-input MFMA kernels
 *                    (default: three float16 products of exactly split operands, vc_gemm16: float32-accurate)
 *   "gru_f32_wide"   0 = float32 INFERENCE recurrences of more than 128 units stay on the streaming kernel (default: the
 *                    training forward kernel, which keeps half of the 786 KB of weights resident: 15.8 -> 3.0 ms at 64 windows)
 *   "gemm16_split"   vc_gemm16, single-pair launches: ways K is split over workgroups (1..8) + 16 * block map (0 = the splits
 *                    of a row tile on one XCD, 1 = one K range per XCD: ways must divide 8); default: chosen from the shape
 * All alternatives compute the same function (tests compare them).  Three more names, "ablate_bank256",
 * "ablate_bank256_only" and "ablate_cbhg_front", skip parts of a kernel for timing and give WRONG results: they
 * exist only in a library built with -DVC_ABLATE (tools/build_ablate.sh; vc_ablate_build() returns 1 there) and
 * are rejected with VC_ERR_INVALID by the shipped build.  Options are process-global; set them between launches. */
int vc_set_option(const char* name, int value);
int vc_get_option(const char* name, int* value);
int vc_ablate_build(void);

/* ------------------------------------------------------------------------------------------
 * Signal front-end: audio_lib.calc_MFCC_input  (/root/reference/audio_lib.py:89-244)
 *   amplitude normalisation (:125-126) -> pre-emphasis FIR (:12-28,:129-133) -> centred,
 *   reflect-padded STFT (:141-147, librosa.core.stft) -> |.|^2 -> power_to_db (:155-157) ->
 *   Slaney mel filterbank (:160-169) -> amplitude_to_db of the mel power (:172) -> DCT-II
 *   (:176-179) -> first-coefficient / scale / delta / min-shift / clip post-processing
 *   (:207-240), float32 time-major outputs (:244).
 * ------------------------------------------------------------------------------------------ */
typedef struct vc_frontend_cfg {
    int32_t sample_rate;               /* sr                       (default 16000) */
    int32_t hop_length;                /* hop_length                               */
    int32_t win_length;                /* win_length                               */
    int32_t n_fft;                     /* n_fft (None in Python -> win_length)     */
    int32_t n_mels;                    /* n_mels                                   */
    int32_t n_mfcc;                    /* n_mfcc                                   */
    float pre_emphasis;                /* 0.0 => filter skipped (audio_lib.py:129) */
    float mean_abs_amp_norm;           /* 1.0 => skipped        (audio_lib.py:125) */
    float mfcc_norm_factor;            /* 1.0 => skipped        (audio_lib.py:223) */
    float M_dB_norm_factor;            /* 1.0 => skipped        (audio_lib.py:234) */
    float P_dB_norm_factor;            /* 1.0 => skipped        (audio_lib.py:230) */
    int32_t mfcc_normaleze_first_mfcc; /* bool                  (audio_lib.py:220) */
    int32_t calc_mfcc_derivate;        /* bool                  (audio_lib.py:226) */
    int32_t clip_output;               /* bool                  (audio_lib.py:237) */
} vc_frontend_cfg;

typedef struct vc_frontend_plan vc_frontend_plan;

/* Host-only: the float64 tables a plan is built from, without touching the GPU.
 * h_mel [n_mels, 1+n_fft/2] = librosa.filters.mel(sr, n_fft, n_mels, norm=1) (audio_lib.py:160-166),
 * h_dct [n_mfcc, n_mels]    = librosa.filters.dct(n_mfcc, n_mels)            (audio_lib.py:176).
 * Either pointer may be NULL. */
int vc_frontend_host_tables(const vc_frontend_cfg* cfg, double* h_mel, double* h_dct);

/* Builds the device-side constant tables (window, DFT twiddles, sparse mel filterbank, DCT
 * basis).  h_window: host float64[win_length] analysis window (what
 * scipy.signal.get_window(name, win_length, fftbins=True) returns), or NULL for periodic hann.
 * Synchronous (small H2D copies). */
int vc_frontend_plan_create(const vc_frontend_cfg* cfg, const double* h_window,
                            vc_frontend_plan** out_plan);
void vc_frontend_plan_destroy(vc_frontend_plan* plan);

/* Number of frames for an L-sample utterance: 1 + L / hop_length  (audio_lib.py:52). */
int32_t vc_frontend_num_frames(const vc_frontend_plan* plan, int32_t n_samples);
/* Output feature widths: MFCC = n_mfcc * (1 | 2), mel = n_mels, power = 1 + n_fft/2. */
int32_t vc_frontend_mfcc_width(const vc_frontend_plan* plan);
int32_t vc_frontend_power_width(const vc_frontend_plan* plan);
/* Copies the dense float64 mel matrix [n_mels, 1+n_fft/2] / DCT basis [n_mfcc, n_mels] the
 * plan was built from into host buffers (for inspection and tests). */
int vc_frontend_get_mel(const vc_frontend_plan* plan, double* h_out);
int vc_frontend_get_dct(const vc_frontend_plan* plan, double* h_out);

size_t vc_frontend_workspace_bytes(const vc_frontend_plan* plan, int32_t batch, int32_t max_samples);

/* Batched feature extraction.
 *   d_wav      float32 [batch, wav_stride]   (utterance b = row b, first lens[b] samples)
 *   d_lens     int32   [batch] sample counts, or NULL => every utterance has max_samples
 *              (each must satisfy n_fft/2 < len <= max_samples)
 *   max_frames = 1 + max_samples / hop_length
 *   out_rows   row count of the outputs per utterance: 0 = max_frames; a smaller value stores only the first out_rows
 *              frames (the later ones still count for the utterance's normalisation statistics, as in the reference,
 *              which computes the whole utterance and then cuts windows: /root/reference/test.py:121-123, 240-241) -- e.g.
 *              800 for 4 s at hop 80, so that [batch, 800, n_mels] IS the [2 * batch, 400, n_mels] window batch of the
 *              encoder without a copy.  Needs the two-pass 400-point path when < max_frames.
 *   d_mfcc     float32 [batch, out_rows, mfcc_width]
 *   d_mel_db   float32 [batch, out_rows, n_mels]
 *   d_pow_db   float32 [batch, out_rows, 1 + n_fft/2]
 *              rows f >= 1 + lens[b]/hop of utterance b are zero-filled.
 *   d_workspace / workspace_bytes: scratch of at least vc_frontend_workspace_bytes().
 * Two launches on `stream`: STFT power + mel + raw dB with per-tile max / min / sum|x| partials;
 * finalize (amplitude normalisation as a dB offset, amin and top_db clips, min shift, DCT, delta,
 * clip).  With hop_length > n_fft/2 a third launch computes the per-utterance sum|x| first. */
int vc_frontend_f32(const vc_frontend_plan* plan, const float* d_wav, const int32_t* d_lens,
                    int32_t batch, int32_t max_samples, int32_t wav_stride, int32_t out_rows,
                    float* d_mfcc, float* d_mel_db, float* d_pow_db,
                    void* d_workspace, size_t workspace_bytes, void* stream);

/* Same call restricted to a subset of its launches (measurement hook used by bench.py to time
 * one kernel with HIP events): stage_mask bit0 = |x| partial sums (a launch only when
 * hop_length > n_fft/2), bit1 = STFT power/mel/dB, bit2 = finalize.  Later stages read what earlier ones left in the workspace. */
int vc_frontend_stages_f32(const vc_frontend_plan* plan, const float* d_wav, const int32_t* d_lens,
                           int32_t batch, int32_t max_samples, int32_t wav_stride, int32_t out_rows,
                           float* d_mfcc, float* d_mel_db, float* d_pow_db,
                           void* d_workspace, size_t workspace_bytes, void* stream,
                           int32_t stage_mask);

/* ------------------------------------------------------------------------------------------
 * Network blocks: modules.py (/root/reference/modules.py:39-356) lowered to four kernels.
 * dtype codes for activations/weights: */
#define VC_F32 0
#define VC_BF16 1
/* activation codes */
#define VC_ACT_NONE 0
#define VC_ACT_RELU 1
#define VC_ACT_SIGMOID 2
#define VC_ACT_TANH 3
/* vc_gemm_desc.mode */
#define VC_GEMM_PLAIN 0
#define VC_GEMM_HIGHWAY 1
#define VC_GEMM_MAX_GROUPS 32

/* One implicit-GEMM launch:  C[m, c_off + n] = epilogue( sum_kk A[m, kk] * Bt[n, kk] ).
 *
 * A is a Toeplitz VIEW of the activation tensor X [M = N_windows*T rows, Cin channels, row
 * stride ldx]:  A[m, j*Cin + c] = X[m + j - pad_l, c]  if 0 <= (m mod T) + j - pad_l < T else 0,
 * which is exactly tf.layers.conv1d(padding="SAME", stride 1, no bias) on [N,T,Cin] with a
 * kernel stored [taps, Cin, Cout] (modules.py:104-140; pad_l = (taps-1)/2).  taps = 1 gives
 * tf.layers.dense (modules.py:291-293, 315-317; encoder.py:109; decoder.py:127,179).
 * Bt is the kernel TRANSPOSED to [Cout, taps*Cin] (K contiguous), dtype = `dtype`.
 *
 * Grouped launch (n_groups > 1) = conv1d_banks (modules.py:144-166): group g has its own
 * taps/pad_l/K/Bt and writes columns [c_off, c_off + N) of the shared output (the concat).
 *
 * Optional prologue on A (applied per element, in this order):
 *   pro_scale/pro_shift [Cin] affine, pro_relu, pro_pool: max(A[r], A[r+1]) along time with the
 *   TF "same" rule out[T-1] = x[T-1]  (tf.layers.max_pooling1d(2,1,"same"), modules.py:331);
 *   pro_pool = 2 additionally promises the pooled values are >= 0 (integer-ordered max).
 * Epilogue: v = acc * epi_scale[c] + epi_shift[c] (NULL scale = 1, NULL shift = 0; this is the
 *   dense bias or the folded inference FusedBatchNorm of modules.py:39-102), activation,
 *   + residual R[m, n] (modules.py:340), stored as float32 (out_f32 != 0) or as `dtype`.
 * mode VC_GEMM_HIGHWAY (modules.py:297-319): Bt holds [32 rows of dense1^T | 32 rows of
 *   dense2^T] interleaved per 32 output units (N = 64*ceil(H/32) rows, zero-padded), epi_shift
 *   the biases in the same order; output [M, H]: relu(h)*sig(t) + x*(1 - sig(t)), x = X.
 * All device pointers; scale/shift vectors are float32. */
typedef struct vc_gemm_group {
    const void* d_Bt;   /* [N, K] transposed kernel, row stride K */
    int32_t K;          /* taps * Cin (multiple of 4 for f32, 8 for bf16) */
    int32_t taps;
    int32_t pad_l;
    int32_t c_off;      /* first output column of this group */
} vc_gemm_group;

typedef struct vc_gemm_desc {
    int32_t dtype;              /* VC_F32 | VC_BF16: type of X, Bt, R (and C unless out_f32) */
    int32_t mode;               /* VC_GEMM_PLAIN | VC_GEMM_HIGHWAY */
    const void* d_X;
    int32_t M, T, Cin, ldx;
    int32_t N;                  /* output columns per group */
    int32_t n_groups;
    vc_gemm_group groups[VC_GEMM_MAX_GROUPS];
    const float* d_pro_scale;   /* [Cin] or NULL */
    const float* d_pro_shift;   /* [Cin] or NULL */
    int32_t pro_relu, pro_pool;
    const float* d_epi_scale;   /* [c_off + N] or NULL */
    const float* d_epi_shift;   /* [c_off + N] or NULL */
    int32_t act;
    const void* d_R;            /* residual [M, ldr] or NULL */
    int32_t ldr;
    void* d_C;
    int32_t ldc;
    int32_t out_f32;
    /* tf.layers.dropout in training mode (modules.py:292,294), applied after the activation:
     * keep probability (0 = off) and the seed of the stateless mask (splitmix64 of the output
     * element index m*ldc + column; see drop_keep_elem in csrc/vc_gemm.hip). */
    float drop_keep;
    unsigned long long drop_seed;
    int32_t sum_groups;            /* != 0: the groups are PARTIAL SUMS of one output [M, N] (columns [0, N) of C) instead of
                                    * column blocks of it: group g convolves input channels [c_off, c_off + Cin) of X with its
                                    * own taps / pad_l / Bt, and all of them accumulate in the same tile before the epilogue
                                    * (+ residual) runs once.  This is the data gradient of conv1d_banks (the sum over the banks
                                    * of dZ_k * W_k^T, tf.gradients through modules.py:144-166) as ONE launch with the banks'
                                    * whole K.  1: one block per output tile runs every group, then the usual epilogue.
                                    * S > 1 (<= 16): the groups are dealt to S blocks per tile (g with its mirror n-1-g) and
                                    * each ADDS its bare float32 partial tile to C with atomics -- C must hold the starting
                                    * value (zeros or the residual term), no epilogue terms; for launches whose tile count
                                    * alone would leave the chip idle (summation order then varies from run to run). */
    int32_t epi_pool;              /* 1: store max(y[t], y[t+1]) inside each window of T frames (the last frame keeps
                                    * its value) = tf.layers.max_pooling1d(2, 1, 'same') of the result, fused into the
                                    * producer (modules.py:331 after :329).  Needs act = ReLU and a launch for which
                                    * vc_conv_gemm_epi_pool_supported() returns 1; vc_conv_gemm rejects it otherwise. */
    void* d_workspace;             /* optional scratch for launches that split K over workgroups (today: the 256-channel
                                    * long-K projection on the bank tiles when its row tiles alone would leave most CUs
                                    * idle): vc_conv_gemm_workspace_bytes() says how much such a launch wants (0: none).
                                    * 256-byte aligned, private to the call until it has finished on its stream (contents
                                    * need not be initialised).  NULL or too small: the unsplit form runs, same result. */
    size_t workspace_bytes;
} vc_gemm_desc;

int vc_conv_gemm(const vc_gemm_desc* desc, void* stream);
/* Bytes of d_workspace the launch described by `desc` can use (0 for nearly all launches; d_workspace itself is ignored). */
size_t vc_conv_gemm_workspace_bytes(const vc_gemm_desc* desc);
/* 1 if `desc` (with epi_pool set) can run with the pooled epilogue: the bf16 filter-bank launch that
 * maps onto the paired 256-row tiles (tiles then overlap by one frame), else 0.  Host-only. */
int vc_conv_gemm_epi_pool_supported(const vc_gemm_desc* desc);

/* Bidirectional LSTM recurrence: modules.lstm (/root/reference/modules.py:207-243 -> tf.contrib.rnn.LSTMCell with its
 * defaults: no peepholes, no projection, forget_bias 1.0, gate order i, j, f, o; bidirectional_dynamic_rnn, zero state).
 * d_xproj [n_seq*T, 8H] float32 = x W_x + b for both directions (fw | bw, 4H columns each); d_Wh_* [H, 4H] recurrent
 * halves (w_dtype); d_out [n_seq, T, 2H] (fw | bw).  H <= 512.  Reachable only with use_lstm = true, which no shipped
 * configuration sets: an any-size kernel, not a tuned one. */
int vc_lstm_bidir(const float* d_xproj, const void* d_Wh_fw, const void* d_Wh_bw, int32_t w_dtype, int32_t n_seq, int32_t T,
                  int32_t H, void* d_out, int32_t out_dtype, void* stream);

/* softmax + argmax over the last axis (encoder.py:110-111): logits float32 [M, ldl >= N] ->
 * probabilities (dtype out_dtype, row stride ldp, columns [N, ldp) zero-filled so the
 * decoder's first dense can read 16-byte rows) and int32 class ids (first maximum). */
int vc_softmax_argmax(const float* d_logits, int32_t M, int32_t N, int32_t ldl,
                      void* d_prob, int32_t ldp, int32_t out_dtype, int32_t* d_class, void* stream);
/* Same, writing the probabilities twice in one launch: float32 (the API's y_pred) and a zero-padded bf16 copy (the
 * decoder's input, decoder.py:86), identical to two calls of vc_softmax_argmax. */
int vc_softmax_argmax_dual(const float* d_logits, int32_t M, int32_t N, int32_t ldl, float* d_prob, int32_t ldp,
                           void* d_prob_bf16, int32_t ldp_bf16, int32_t* d_class, void* stream);

/* Bidirectional GRU recurrence (modules.py:168-204 -> tf.nn.bidirectional_dynamic_rnn over
 * tf.contrib.rnn.GRUCell):  g = sigmoid(xg + h Wg_h);  r,u = split(g) (r first);
 * c = tanh(xc + (r*h) Wc_h);  h' = u*h + (1-u)*c, zero initial state, backward direction runs
 * t = T-1..0.  The input halves of the cell matmuls are hoisted into one GEMM beforehand:
 *   d_xproj float32 [n_seq*T, 6H] = x @ [Wg_x^fw | Wc_x^fw | Wg_x^bw | Wc_x^bw] + biases.
 *   d_Wh[dir]: recurrent weights [H, 3H] = [Wg_h | Wc_h] (rows = h index), dtype w_dtype.
 *   d_out [n_seq*T, 2H] (dtype out_dtype): fw in columns [0,H), bw in [H,2H).
 *   d_workspace: scratch of vc_gru_workspace_bytes(H, w_dtype) bytes (the register-resident
 *   kernels re-pack the weights into their per-lane order there on every call). */
size_t vc_gru_workspace_bytes(int32_t H, int32_t w_dtype);
int vc_gru_bidir(const float* d_xproj, const void* d_Wh_fw, const void* d_Wh_bw, int32_t w_dtype,
                 int32_t n_seq, int32_t T, int32_t H, void* d_out, int32_t out_dtype,
                 void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step of decoder_specs (/root/reference/decoder.py:185-263, 327-345), float32.
 * Data gradients of dense/conv layers reuse vc_conv_gemm (dX = conv of dY with the taps flipped
 * and the kernel in TF layout as the transposed operand).  The remaining pieces: */

/* Filter gradient of tf.layers.dense / conv1d, written in TF layout [taps, Cin, N]:
 *   dW[j*Cin + c, o] = sum_m X[m + j + shift0, c] * dY[m, o]    (shift0 = -pad_l; a frame of
 *   another window contributes nothing).  Operands are TRANSPOSED, frames contiguous, built by
 *   vc_transpose_pad with a zero margin of `margin` frames on both sides of every row:
 *   d_XT [Cin, ldxt], d_dYT rows [N, ldyt]; both pointers address frame 0; the allocations carry
 *   one slack row after the last (a shifted tail read may run `margin` frames past it).
 *   Groups = the banks. */
typedef struct vc_wgrad_group {
    const void* d_dYT;
    void* d_dW;
    int32_t N, taps, shift0;
    int32_t ldw;        /* row stride of d_dW (0 = N): lets a group fill a column slice */
} vc_wgrad_group;
typedef struct vc_wgrad_desc {
    const void* d_XT;
    int32_t ldxt, ldyt, Cin, M, T, margin, n_groups;
    int32_t splits_allowed;   /* != 0: every d_dW is pre-zeroed and may be accumulated with float atomics
                                 (the frame reduction is then split over more workgroups) */
    vc_wgrad_group groups[VC_GEMM_MAX_GROUPS];
} vc_wgrad_desc;
int vc_conv_wgrad(const vc_wgrad_desc* desc, void* stream);

/* XT[c, pad + m] = pro(X)[m + row_shift, c] (zero when the shifted frame leaves the window);
 * pro = optional per-channel affine, relu, time max-pool (the forward operand prologue).
 * The launch also writes the zero margins XT[c, 0 .. pad) and XT[c, pad + M .. ldt) of every row c < C and, with
 * slack_row != 0, a zero row C (d_XT then holds (C + 1) x ldt floats): the buffer need not be initialised. */
int vc_transpose_pad(const float* d_X, int32_t M, int32_t C, int32_t ld, int32_t T, const float* d_scale,
                     const float* d_shift, int32_t relu, int32_t pool, int32_t row_shift, float* d_XT,
                     int32_t ldt, int32_t pad, int32_t slack_row, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training convolutions on split-float16 operands ("f16x3"; csrc/vc_gemm16.hip).  The reference trains in float32
 * (/root/reference/decoder.py:185-263); gfx950's f32-input MFMA runs at 1/16 of the 16-bit rate.  x * s (s a power of
 * two) splits exactly into float16 hi + lo + r, |r| <= 2^-22 |x s|, and hi*hi + hi*lo + lo*hi summed in float32
 * reproduces the float32 product to 2^-22: the error of the result against float64 equals a float32 GEMM's (tests).
 * ------------------------------------------------------------------------------------------ */
/* X [M, C] float32 (row stride ldx) -> d_out16 [M, 2C] float16 = [hi plane | lo plane] of pro(X)[m] * s_w, and
 * d_row_scale[m] = 1 / s_w, s_w the power of two that puts the largest magnitude of the row's WINDOW (T rows) in
 * [2^14, 2^15) (1 for an all-zero window): the taps of a convolution stay inside a window, whose rows must share
 * their scale.  d_row_scale holds M + M / T floats: the tail is scratch.  pro = optional per-channel affine (d_scale /
 * d_shift), relu, max-pool(2, 1, same) along time inside each window -- the operand prologue of vc_conv_gemm.
 * C: a multiple of 64 up to 4096.  Two passes over X (window maxima, then the split). */
int vc_split16(const float* d_X, int32_t M, int32_t C, int32_t ldx, int32_t T, const float* d_scale, const float* d_shift,
               int32_t relu, int32_t pool, void* d_out16, float* d_row_scale, void* stream);
/* One item of vc_weights16: a TF-layout float32 kernel src [k][cin][cout] -> float16 [hi | lo] operand rows at dst.
 * mode 0 (forward operand):       row o (cout rows), column base + j * tap_stride + plane * plane_stride + c
 * mode 1 (data-gradient operand): row c (cin rows),  column base + (k - 1 - j) * tap_stride + plane * plane_stride + o
 * row_len = elements per dst row.  Items of one `group` share ONE power-of-two scale (their largest magnitude ->
 * [2^14, 2^15)); 1 / scale is written to scale_dst[0 .. scale_n) (the GEMM's per-channel d_col_scale). */
typedef struct vc_w16_item {
    const float* src;
    void* dst;
    float* scale_dst;
    int32_t k, cin, cout, mode;
    int32_t row_len, tap_stride, plane_stride, base;
    int32_t group, scale_n;
} vc_w16_item;
/* d_items: n_items items in DEVICE memory; d_gmax: n_groups uint32 of scratch.  Three launches, no host sync. */
int vc_weights16(const vc_w16_item* d_items, int32_t n_items, uint32_t* d_gmax, int32_t n_groups, void* stream);
/* C[m, c_off + n] (+)= ( sum over taps j, channels c of  X[m + j - pad_l, c] * W[n][j][c] ) * row_scale[m] * col_scale[ch]
 *                     + col_shift[ch],   X and W the float32 values vc_split16 / vc_weights16 split, SAME padding per
 * window of T rows like vc_conv_gemm.  A PAIR is two 128-column filters over the same input: widths (taps0, taps0 +
 * extra) with a common pad_l (a filter-bank pair 2p+1 / 2p+2, /root/reference/modules.py:144-166), or the two halves
 * of one 256-column filter (extra 0).  d_Bt0 / d_Bt1: [128][taps * 2C] float16, per tap [hi plane (C) | lo plane (C)].
 * ragged != 0 is the filter bank's DATA gradient (tf.gradients through modules.py:144-166): X = dZ [M, C = 128 * K],
 * bank k = 1..K contributes its 128 channels with k taps and left padding k / 2; rows of d_Bt*: [plane][bank k][tap]
 * [128] (vc_w16_item mode 1 with tap_stride 128, plane_stride 128 * K (K + 1) / 2, base 128 * k (k - 1) / 2).
 * A single-pair launch whose row tiles would not fill the chip splits K over up to 8 workgroups per row tile when
 * given vc_gemm16_workspace_bytes() of workspace (partial sums added in a fixed order: bit-identical run to run). */
typedef struct vc_gemm16_pair {
    const void* d_Bt0;
    const void* d_Bt1;
    int32_t taps0, extra, pad_l, c_off0, c_off1;
    int32_t row0;               /* first row of X the pair reads; output row r of the pair is X row row0 + r */
    int32_t nrows0, nrows1;     /* output rows each filter stores (0 = M - row0); nrows1 = -1: the pair is ONE 128-column
                                 * filter (d_Bt1 is not read, nothing is stored for it) */
    int32_t s_off0, s_off1;     /* atomic_splits only: first entry of d_col_scale of each filter (otherwise c_off0 / c_off1) */
} vc_gemm16_pair;
typedef struct vc_gemm16_desc {
    const void* d_X16;          /* [M, ldx >= 2C] float16 from vc_split16 */
    const float* d_row_scale;   /* [M] or NULL */
    int32_t M, T, C, ldx;
    int32_t n_pairs, ragged;
    vc_gemm16_pair pairs[16];
    const float* d_col_scale;   /* per output column, or NULL */
    const float* d_col_shift;
    float* d_C;
    int32_t ldc;
    int32_t accumulate;         /* != 0: add to the contents of d_C (after the activation: a residual) */
    int32_t act;                /* VC_ACT_NONE | VC_ACT_RELU, applied to acc * scales + shift */
    int32_t atomic_splits;      /* n >= 1: every (row tile, pair) is computed by n workgroups over n ranges of K that ADD their
                                 * partial tiles to d_C with float atomics (d_C pre-initialised; summation order not fixed).
                                 * The weight-gradient form (tf.gradients w.r.t. a conv kernel): rows = (tap, input channel) of
                                 * shifted, transposed activations from vc_transpose_split16, the contraction runs over the
                                 * frames, and a pair's c_off0 / c_off1 are ELEMENT offsets of the two filters' [rows, ldc]
                                 * gradient blocks from d_C (any alignment).  No col_shift, no ragged walk. */
    void* d_workspace;          /* 256-byte aligned, or NULL */
    size_t workspace_bytes;
} vc_gemm16_desc;
size_t vc_gemm16_workspace_bytes(int32_t M, int32_t C, int32_t n_pairs);
/* Operands of the weight-gradient form of vc_gemm16 (contraction over the frames): X [M, C] float32 (+ the prologue
 * of vc_split16) -> d_out16 rows (si * C + c), si = 0 .. n_shifts-1, each [hi plane (M) | lo plane (M)] float16 of
 * pro(X)[m + shift0 + si, c] * s_c (0 where that frame leaves its window of T), s_c the power of two that puts channel
 * c's largest magnitude in [2^14, 2^15); d_row_scale[si * C + c] = 1 / s_c.  d_row_scale holds (n_shifts + 1) * C
 * floats (the tail is scratch).  M and C: multiples of 64. */
int vc_transpose_split16(const float* d_X, int32_t M, int32_t C, int32_t ldx, int32_t T, const float* d_scale,
                         const float* d_shift, int32_t relu, int32_t pool, int32_t shift0, int32_t n_shifts, void* d_out16,
                         float* d_row_scale, void* stream);
int vc_gemm16(const vc_gemm16_desc* desc, void* stream);

/* Train-mode FusedBatchNorm bookkeeping (modules.py:77-84, is_training): batch mean / biased
 * variance of X [M, C] -> scale/shift (consumed by the next launch's prologue or vc_affine_act),
 * saved mean/rstd for backward, moving statistics updated in place (decay, Bessel-corrected
 * variance).  d_workspace: vc_stats_workspace_floats(M, C) floats. */
size_t vc_stats_workspace_floats(int32_t M, int32_t C);
int vc_bn_train_stats(const float* d_X, int32_t M, int32_t C, int32_t ld, const float* d_gamma,
                      const float* d_beta, float* d_moving_mean, float* d_moving_var, float decay, float eps,
                      float* d_scale, float* d_shift, float* d_mean, float* d_rstd, float* d_workspace,
                      void* stream);
/* out = act(X * scale[c] + shift[c]) + R   (any of scale/shift/R may be NULL). */
int vc_affine_act(const float* d_X, const float* d_scale, const float* d_shift, int32_t relu, const float* d_R,
                  float* d_out, size_t n, int32_t C, void* stream);
/* BatchNorm backward fused with what follows the norm: mode 0 none, 1 relu, 2 relu + time
 * max-pool (modules.py:165,331: d_G is the gradient w.r.t. the pooled tensor).  Writes dX (raw
 * conv output gradient), dgamma, dbeta. */
int vc_bn_backward(const float* d_G, const float* d_X, int32_t M, int32_t C, int32_t ld, int32_t T,
                   const float* d_gamma, const float* d_scale, const float* d_shift, const float* d_mean,
                   const float* d_rstd, int32_t mode, float* d_dX, float* d_dgamma, float* d_dbeta,
                   float* d_workspace, void* stream);
/* The relu / max-pool routing vc_bn_backward(mode 2) applies (modules.py:165 relu, :331 max_pooling1d(2, 1, same)), one
 * byte per element of X [M, C] (output contiguous, row stride C): bit 0: bn(X) > 0; bit 1: the element receives the
 * gradient of its own frame's pool output (last frame of a window, or >= its successor); bit 2: it receives the
 * previous frame's (strictly greater than its predecessor).  Same device function as the backward pass itself; where
 * two float32 pre-activations tie to within rounding TensorFlow's float32 kernels would be equally arbitrary, so a
 * float64 restatement has to be handed these decisions to be comparable element by element (tests). */
int vc_bn_post_routing(const float* d_X, int32_t M, int32_t C, int32_t ld, int32_t T, const float* d_scale,
                       const float* d_shift, uint8_t* d_bits, void* stream);
/* dZ = (Y > 0) ? dY * inv_keep : 0 for Y = dropout(relu(Z)) (modules.py:291-294). */
int vc_relu_dropout_backward(const float* d_dY, const float* d_Y, float inv_keep, float* d_dZ, size_t n, void* stream);
/* highwaynet backward gate arithmetic (modules.py:315-318) on re-computed pre-activations in the
 * forward's paired column layout [M, NP]; writes d(pre) [M, NP] and the direct path dO*(1-T). */
int vc_highway_backward(const float* d_pre, int32_t NP, const float* d_X, const float* d_dO, int32_t M, int32_t H,
                        float* d_dpre, float* d_dXd, void* stream);
/* out[c] (+)= sum_m X[m, c]  (bias gradients); d_workspace: 64 * C floats. */
int vc_col_sum(const float* d_X, int32_t M, int32_t C, int32_t ld, float* d_out, int32_t accumulate,
               float* d_workspace, void* stream);
int vc_fill(float* d_p, float value, size_t n, void* stream);
/* out[m][c] = a * X[m][c] + b * Y[m][c] over [M, C] float32 views with row strides ldx / ldy / ldo (out may alias X or
 * Y).  decoder_specs._build_model's teacher-forced stage-2 input, /root/reference/decoder.py:148-152:
 * inputs_step2 = f_mel_pred * y_mel + (1 - f_mel_pred) * target_mel, and its gradient dY_mel += f_mel_pred * dX. */
/* Kernel-layout copies of many convolution weights in ONE launch (training: after every optimiser step).
 * d_items: DEVICE array of n_items descriptors; src = TF-layout kernel [k, cin, cout] float32 (tf.layers.conv1d /
 * dense with k = 1: /root/reference/modules.py:104-140), dst float32:
 *   mode 0: [cout, k*cin]  = the transposed operand vc_conv_gemm's groups take (vc_gemm_group.d_Bt),
 *   mode 1: [cin, k*cout] with the taps reversed = the operand of the data-gradient convolution (decoder.py:236-246's
 *           tf.gradients through conv1d). */
typedef struct vc_layout_item {
    const float* src;
    float* dst;
    int32_t k, cin, cout, mode;
} vc_layout_item;
int vc_weight_layouts(const vc_layout_item* d_items, int32_t n_items, void* stream);
int vc_axpby(const float* d_X, int32_t ldx, float a, const float* d_Y, int32_t ldy, float b, float* d_out, int32_t ldo,
             int32_t M, int32_t C, void* stream);
/* loss = weight * mean((y - t)^2) (decoder.py:187-189); optional d_dY = 2*weight/n * (y - t).
 * y/t are contiguous [n/C, C]; d_dY is written with row stride ld_dy >= C (padding columns are
 * left untouched).  d_workspace: 256 floats; d_loss: 1 float on the device (no host sync). */
int vc_mse_loss(const float* d_y, const float* d_t, size_t n, float weight, float* d_dY, int32_t C, int32_t ld_dy,
                float* d_loss, float* d_workspace, void* stream);
/* LSTM recurrence in training mode (use_lstm; /root/reference/modules.py:207-243: tf.contrib.rnn.LSTMCell, forget_bias 1.0,
 * under bidirectional_dynamic_rnn): d_xproj [n_seq*T, 8H] = per direction the input projections (i | j | f | o) incl. bias,
 * d_Wh_* [H, 4H] the recurrent rows of the cell kernel; stores the hidden states d_out [n_seq*T, 2H], the ACTIVATED gates
 * d_gates [2][n_seq*T, 4H] and the cell states d_cstate [2][n_seq*T, H].  float32, H <= 512.  (No shipped configuration
 * enables use_lstm: any-size kernels, not tuned ones.) */
int vc_lstm_train_forward(const float* d_xproj, const float* d_Wh_fw, const float* d_Wh_bw, int32_t n_seq, int32_t T,
                          int32_t H, float* d_out, float* d_gates, float* d_cstate, void* stream);
/* BPTT of the above: d_dout [n_seq*T, 2H] -> d_dpre [n_seq*T, 8H], the gradient w.r.t. the gate pre-activations in the
 * layout of d_xproj.  d_WhT_* [4H, H]: the recurrent weights transposed. */
int vc_lstm_backward(const float* d_dout, const float* d_gates, const float* d_cstate, const float* d_WhT_fw,
                     const float* d_WhT_bw, int32_t n_seq, int32_t T, int32_t H, float* d_dpre, void* stream);
/* Encoder loss and metrics (/root/reference/encoder.py:134-150): out3 = [mean softmax cross-entropy
 * with float labels, accuracy of argmax(logits) vs argmax(target), mean squared error of the
 * posteriors]; optional d_dlogits [M, ldd] = (softmax * sum(target) - target) / M.
 * d_workspace: 3 * M floats; results stay on the device. */
int vc_softmax_ce(const float* d_logits, const float* d_target, int32_t M, int32_t C, int32_t ldl, float* d_dlogits,
                  int32_t ldd, float* d_out3, float* d_workspace, void* stream);
/* tf.train.AdamOptimizer update on flat buffers (decoder.py:236-246): g is first multiplied by
 * grad_scale (1/world for data-parallel averaging), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) from the host,
 * p -= lr_t * m / (sqrt(v) + epsilon). */
int vc_adam_step(float* d_param, const float* d_grad, float* d_m, float* d_v, size_t n, float lr_t, float beta1,
                 float beta2, float epsilon, float grad_scale, void* stream);
/* GRU recurrence in training mode: like vc_gru_bidir (float32 weights) but also stores the gate
 * activations d_gates [2][n_seq*T, 3H] (r | u | c) and r*h d_rh [2][n_seq*T, H]. */
int vc_gru_train_forward(const float* d_xproj, const float* d_Wh_fw, const float* d_Wh_bw, int32_t n_seq,
                         int32_t T, int32_t H, float* d_out, float* d_gates, float* d_rh, void* stream);
/* BPTT of the above: d_dout [n_seq*T, 2H] -> d_dpre [n_seq*T, 6H] = gradient w.r.t. the gate
 * pre-activations in the layout of d_xproj (input/recurrent weight gradients follow as GEMMs).
 * d_WhT_* [3H, H]: optional transposed copies of the recurrent weights; with them several windows
 * share a workgroup and the W^T matvecs read coalesced rows. */
int vc_gru_backward(const float* d_dout, const float* d_out, const float* d_gates, const float* d_Wh_fw,
                    const float* d_Wh_bw, const float* d_WhT_fw, const float* d_WhT_bw, int32_t n_seq,
                    int32_t T, int32_t H, float* d_dpre, void* stream);

/* ---- Griffin-Lim vocoder ---------------------------------------
 * Replaces audio_lib.griffin_lim_alg / from_power_to_wav / calc_inv_preemphasis
 * (/root/reference/audio_lib.py:249-274, 278-308, 31-47; called from test.py:146-168, 253-275,
 * 346-362).  Spectrogram layout is frame-major: [batch, max_frames, 1 + n_fft/2] (the decoder's
 * y_stft layout, i.e. the TRANSPOSE of the [bins, frames] array librosa works on). */
typedef struct vc_vocoder_plan vc_vocoder_plan;

/* Device tables (padded window, DFT twiddles).  n_fft <= 0 => n_fft = win_length
 * (audio_lib.py:251-252).  h_window: host float64[win_length] or NULL for periodic hann (librosa's
 * default, the only window the reference uses here).  n_fft == 400 takes the 25 x 16 split
 * transform; other (even) sizes take a direct DFT. */
int vc_vocoder_plan_create(int32_t win_length, int32_t hop_length, int32_t n_fft, const double* h_window,
                           vc_vocoder_plan** out_plan);
void vc_vocoder_plan_destroy(vc_vocoder_plan* plan);
/* Samples librosa.istft returns for n_frames frames: hop_length * (n_frames - 1). */
int32_t vc_vocoder_num_samples(const vc_vocoder_plan* plan, int32_t n_frames);
size_t vc_vocoder_workspace_bytes(const vc_vocoder_plan* plan, int32_t batch, int32_t max_frames, int32_t trace);

/* audio_lib.py:289-298: P = max(0, P); optional P = (mean(P)/mean(P**realse)) * P**realse (means
 * over the utterance); amp = sqrt(db_to_power(P / P_dB_norm_factor - 80)).  d_P, d_amp
 * [batch, max_frames, n_bins]; rows >= n_frames[b] are written as 0.  d_n_frames may be NULL. */
int vc_power_to_amp(const float* d_P, const int32_t* d_n_frames, int32_t batch, int32_t max_frames, int32_t n_bins,
                    float P_dB_norm_factor, float realse, float* d_amp, void* stream);

/* audio_lib.py:249-274.  d_amp, d_phase0 [batch, max_frames, 1+n_fft/2] (phase0 in radians: the
 * reference draws pi * U[0,1) on the host, audio_lib.py:255 -- the caller supplies it so a seeded
 * run is reproducible); d_n_frames int32 [batch] or NULL (all max_frames); each utterance needs
 * hop*(n_frames-1) > n_fft/2.  d_wav [batch, wav_stride] receives hop*(n_frames[b]-1) samples per
 * utterance, zero beyond.  d_trace: NULL, or float32 [num_iters, batch] that receives
 * sum_n (wav_i[n] - wav_{i-1}[n])^2 for i >= 1 (the reference's verbose print is
 * sqrt(that / n_samples)); costs one extra launch per iteration.
 * num_iters launches of the fused projection kernel + 1 overlap-add on `stream`. */
int vc_griffin_lim_f32(const vc_vocoder_plan* plan, const float* d_amp, const float* d_phase0,
                       const int32_t* d_n_frames, int32_t batch, int32_t max_frames, int32_t num_iters,
                       float* d_wav, int32_t wav_stride, float* d_trace, void* d_workspace, size_t workspace_bytes,
                       void* stream);

/* audio_lib.py:301-306 in place on d_wav [batch, wav_stride] (first hop*(n_frames[b]-1) samples):
 * y[n] = x[n] + coeff*y[n-1] (skipped when coeff == 0), then y *= mean_abs_amp_norm / mean|y|
 * (skipped when mean_abs_amp_norm <= 0). */
int vc_inv_preemphasis_normalize(const vc_vocoder_plan* plan, float* d_wav, const int32_t* d_n_frames, int32_t batch,
                                 int32_t max_frames, int32_t wav_stride, float coeff, float mean_abs_amp_norm,
                                 void* stream);

/* ---- highway chain (modules.py:297-319 applied L times, modules.py:342-345) ------------------
 * All L highwaynet layers of a CBHG block in one launch (bf16, H = 128 or 256): a block keeps its
 * 128 frames in LDS across the layers, weights stream from L2 in MFMA fragment order; optionally
 * followed, on the same on-chip activations, by the GRU's input projection (modules.py:346,
 * GRUCell x-halves of both directions: a dense layer H -> n_proj with float32 output).
 * vc_highway_pack: d_Bt [n_cols, H] bf16, K contiguous (for the highway layers the paired [2H, H]
 * matrix of vc_conv_gemm's VC_GEMM_HIGHWAY mode: rows 64q..64q+31 = dense1 columns of units 32q..,
 * rows 64q+32.. = dense2) -> d_packed [n_cols*H] bf16 in fragment order.
 * vc_highway_chain: d_packed / d_bias are HOST arrays of n_layers (0..8) device pointers (bias:
 * float32 [2H], paired order).  d_Y [M, ldy] bf16 receives the last layer's output (NULL: not
 * stored; may equal d_X).  d_proj_packed (NULL: no tail) / d_proj_bias [n_proj] / d_P [M, ldp]
 * float32.  Bit-identical to n_layers launches of vc_conv_gemm(VC_GEMM_HIGHWAY) + one dense launch. */
int vc_highway_pack(const void* d_Bt, int32_t n_cols, int32_t H, void* d_packed, void* stream);
int vc_highway_chain(const void* d_X, int32_t M, int32_t H, int32_t ldx, int32_t n_layers,
                     const void* const* d_packed, const float* const* d_bias, void* d_Y, int32_t ldy,
                     const void* d_proj_packed, const float* d_proj_bias, int32_t n_proj, float* d_P, int32_t ldp,
                     void* stream);

/* ---- the encoder's pre-recurrence chain in one launch -------------------------------------------
 * encoder.py:101-107 up to the GRU: prenet (modules.py:274-295) -> conv1d_banks + bn + relu
 * (modules.py:144-166) -> max_pooling1d (modules.py:331) -> conv1d k=3 + bn + relu -> conv1d k=3 + bn
 * + residual (modules.py:334-340) -> highwaynet x n_highway (modules.py:342-345) -> the x-halves of
 * the bidirectional GRU's cell matmuls (modules.py:346, 168-204), for the SHIPPED encoder shape
 * (hp/encoder_cfg_d.json: 80 features, prenet 80 -> 40, 6 banks x 128 filters, GRU of 40 units; bf16
 * weights, inference).  vc_cbhg_front_supported() says whether a shape takes this path; callers run
 * the per-layer entry points (vc_conv_gemm, vc_highway_chain) otherwise -- same results within bf16
 * rounding (different float32 summation order).
 *
 * Weights are handed over in MFMA fragment order, built with vc_mfma_pack from row-major bf16
 * matrices W [rows, K] (row = output channel, K contiguous):
 *   packed[(tile * nks + s) * 64 + lane][e] = W[32 tile + (lane & 31)][kmap(16 s + 8 (lane >> 5) + e)],
 *   tile < ceil(rows / 32), s < nks = ceil(K / 16), zero outside W;
 *   kmap = identity (chained = 0) or, for a layer that consumes the previous layer's result tile
 *   straight from registers (chained = 1), kmap(16 s + 8 h + e) = 32 (s>>1) + 8 (2 (s&1) + (e>>2)) + 4 h + (e&3).
 * Matrices (TF kernels transposed to [out, in]):
 *   d_pk_dense1  [80, 80] plain;  d_pk_dense2 [40, 80] chained;
 *   d_pk_bank    the 6 filters [128, 40 k] (K index = tap * 40 + channel), plain, concatenated k = 1..6;
 *   d_pk_proj1   [40, 2304] with K re-ordered to (width k-1, 32-channel slice w, tap, 16-channel half s, 16):
 *                column ((((k-1) * 4 + w) * 3 + tap) * 2 + s) * 16 + j  <-  conv1d_1 kernel[tap, (k-1) * 128 + 32 w + 16 s + j, :], plain;
 *   d_pk_proj2   [40, 120] (K index = tap * 40 + channel), plain;
 *   d_pk_highway[l] the paired [128, 40] matrix (rows 64 q .. +31 dense1 of units 32 q .., rows 64 q + 32 .. dense2), chained;
 *   d_pk_gru     [240, 40] (rows: fw gates 80 | fw candidate 40 | bw gates 80 | bw candidate 40), chained.
 * d_coef: ONE float32 array of vc_cbhg_front_coef_floats() (= 3232) values, every vector zero padded to its slot:
 *   [0, 96) dense1 bias | [96, 160) dense2 bias | [160, 1184) bank scale | [1184, 2208) bank shift (folded batch
 *   norm of the 768 bank channels) | [2208, 2272) conv1d_1 scale | [2272, 2336) shift | [2336, 2400) conv1d_2 scale |
 *   [2400, 2464) shift | [2464, 2720) GRU bias (240) | [2720 + 128 l, +128) highway layer l biases, paired order.
 * d_x [n_windows * T, ldx] float32 (x_f32 = 1) or bf16; d_xproj [n_windows * T, ldp] float32 receives
 * columns [0, 240). */
#define VC_CBHG_FRONT_MAX_HIGHWAY 4
typedef struct vc_cbhg_front_desc {
    const void* d_x;
    int32_t x_f32, ldx;
    int32_t n_windows, T;
    int32_t n_features, prenet_units, width, n_banks, bank_filters, n_highway, gru_units;
    const void *d_pk_dense1, *d_pk_dense2, *d_pk_bank, *d_pk_proj1, *d_pk_proj2, *d_pk_gru;
    const void* d_pk_highway[VC_CBHG_FRONT_MAX_HIGHWAY];
    const float* d_coef;
    float* d_xproj;
    int32_t ldp;
} vc_cbhg_front_desc;
int vc_mfma_pack(const void* d_W, int32_t rows, int32_t K, int32_t ldw, int32_t chained, void* d_packed, void* stream);
int32_t vc_cbhg_front_coef_floats(void);
int vc_cbhg_front_supported(int32_t n_features, int32_t prenet_units, int32_t width, int32_t n_banks,
                            int32_t bank_filters, int32_t n_highway, int32_t gru_units, int32_t T);
int vc_cbhg_front(const vc_cbhg_front_desc* desc, void* stream);

/* ---- prenet as one launch (modules.py:274-295, inference) ----------------------------------------
 * Y = relu(relu(X W1 + b1) W2 + b2) in bf16 for the decoder stages' shapes (cin_padded -> units1 -> units2 =
 * 64 -> 256 -> 128 and 80 -> 512 -> 256; vc_prenet_chain_supported() tells, other shapes take two vc_conv_gemm
 * launches): the intermediate stays in registers.  d_pk1 = vc_mfma_pack(W1^T [units1, cin_padded], chained = 0),
 * d_pk2 = vc_mfma_pack(W2^T [units2, units1], chained = 1); d_b1 [units1], d_b2 [units2] float32;
 * d_X [M, ldx] bf16, or float32 with x_f32 = 1 (converted on load: y_mel of the previous stage), padding columns zero; d_Y [M, ldy] bf16.  Same bf16 rounding points as the two launches,
 * float32 sums in the same K order (bit-identical in practice, tested to 1e-2). */
int vc_prenet_chain_supported(int32_t cin_padded, int32_t units1, int32_t units2);
int vc_prenet_chain(const void* d_X, int32_t x_f32, int32_t M, int32_t ldx, int32_t cin_padded, int32_t units1, int32_t units2,
                    const void* d_pk1, const float* d_b1, const void* d_pk2, const float* d_b2, void* d_Y, int32_t ldy,
                    void* stream);

/* ---- on-device feature cache -----------------------------------
 * dst[r, :] = src[index[r], :] for index[r] >= 0, else pad_row (zeros when d_pad_row is NULL).
 * Rows are row_bytes wide (multiple of 4).  Replaces the h5py slicing + np.array stacking of
 * sound_ds.py:262-350, ARCTIC_reader.py:277-362, TIMIT_reader.py:474-523 (and packs front-end
 * output into the ragged cache, ARCTIC_reader.py:109-175 / TIMIT_reader.py:144-210). */
int vc_gather_rows(const void* d_src, const int64_t* d_index, const void* d_pad_row, int64_t n_rows,
                   int32_t row_bytes, void* d_dst, void* stream);

/* float32 <-> bf16 conversion of a contiguous buffer (weights preparation, I/O). */
int vc_convert(const void* d_src, int32_t src_dtype, void* d_dst, int32_t dst_dtype, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VC_HIP_H */
