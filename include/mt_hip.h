/* mt_hip.h -- C ABI of libmt_hip.so, the MI355X (gfx950) hot path of
 * cs4247/music-transcription:  30 s waveform chunk -> log-mel -> CNN-RNN -> 88-pitch logits.
 *
 * The reference has no FFI layer (it is pure Python on torch); each entry point
 * below names the reference call site whose arithmetic it replaces.  The Python
 * host (music-transcription_amd/) binds these with ctypes; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensor
 *     .data_ptr()), row-major contiguous, unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return value: 0 on success, negative MT_E* on error, never throws; the
 *     message of the last error on the calling thread is at mt_last_error();
 *   - no entry point allocates device memory or synchronises the device: scratch
 *     comes in as (workspace, workspace_bytes), sizes come from the *_bytes queries;
 *   - thread-compatible (one stream per caller thread); the only mutable global state is the mutex-protected table of
 *     persistent launches in flight (below).
 *
 * Persistent launches.  The LSTM recurrence kernels (mt_lstm_bidir_fwd*, mt_lstm_bidir_bwd*) wait on their own
 * workgroups, so all workgroups of all such launches in flight on a GPU must be resident together.  Launches on one
 * stream run in order; for launches on DIFFERENT streams the library checks at launch time that their CUs (workgroups /
 * occupancy per CU) fit the device and otherwise returns MT_EUNSUPPORTED at once -- it never queues a launch that could
 * stall the others (every in-kernel spin is bounded as well and reports through the status word of its sync workspace).
 * At hidden 512, batch <= 32: six plain forward recurrences, two with the fused input projection.
 */
#ifndef MT_HIP_H
#define MT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MT_OK            0
#define MT_EINVAL       -1   /* bad argument / unsupported shape                 */
#define MT_EWORKSPACE   -2   /* workspace too small                              */
#define MT_EHIP         -3   /* HIP runtime error (launch, memset, ...)          */
#define MT_EUNSUPPORTED -4   /* configuration outside what the kernels implement */

/* 16-bit operand type of the GEMM / convolution kernels (`_dt` entry points, mt_cnnrnn*_weights.operand_dtype).
 * Both run at the same MFMA rate with f32 accumulation.  Inference uses f16 (11 significand bits: logits within
 * ~1e-3 relative of the reference's fp32 arithmetic); the training step uses bf16 (f32's
 * exponent range for gradients; BASELINE configs[3] names bf16).  The un-suffixed entry points are the bf16 forms. */
#define MT_DT_BF16 0
#define MT_DT_F16  1
/* Flag OR-ed into `dt` of mt_gemm_lstm_gx_dt and into `flags` of mt_lstm_bidir_fwd_ex: the gate pre-activations
 * travel between the two as f16 -- same [group][t][dir][unit/8][gate][unit%8][chunk%32] layout, half the bytes of the largest
 * intermediate of the forward (492 -> 246 MB per layer at B = 32); round-to-nearest-even of W_ih x + b, added in f32 by the
 * cell update.  Inference only (the training step keeps its gates in f32).                                                  */
#define MT_GX_F16  0x10

#define MT_N_FFT 2048        /* librosa default the reference relies on (main.py:117-122) */
#define MT_N_PITCH 88

typedef void* mt_stream_t;

int         mt_version(void);
const char* mt_last_error(void);
/* Number of HIP devices visible, or negative error.  Does not create a context. */
int         mt_device_count(void);
/* Makes `device` current for the calling thread and checks that it is a gfx950 part (the only code object in the library).  No global
 * tables are built (plans and packed weights are caller-owned buffers), so calling it is optional; idempotent.  (mt_init.) */
int         mt_init(int device);
/* One size query for the path's scratch / intermediate buffers (mt_workspace_bytes(kind, dims...)): forwards to the per-buffer
 * queries below.  kind = MT_WS_*; MT_WS_MEL_PLAN takes n_mels in B; unknown kind -> 0 and mt_last_error().                          */
#define MT_WS_LSTM_GX        1
#define MT_WS_LSTM_HX        2
#define MT_WS_LSTM_CX        3
#define MT_WS_LSTM_SYNC      4
#define MT_WS_LSTM_BWD_PART  5
#define MT_WS_LSTM_DGX       6
#define MT_WS_MEL_PLAN       7
#define MT_WS_ADAM           8
size_t      mt_workspace_bytes(int kind, int B, int T, int H);
/* In-place SUM all-reduce of `count` elements over the ranks of the caller's RCCL communicator (`comm` = ncclComm_t), queued on `stream`:
 * the gradient step of data-parallel training (train/train_transcriber.py has none: single GPU; BASELINE configs[3]) for a host that is
 * not Python -- the flat gradient buffer, then mt_adam_clip_step_ex with grad_scale = 1 / world.  RCCL is opened on first use
 * (librccl.so.1), not linked.  The Python host uses torch.distributed instead.                          */
#define MT_AR_F32   0
#define MT_AR_BF16  1
#define MT_AR_F16   2
int         mt_allreduce(void* buf, size_t count, int dtype, void* comm, mt_stream_t stream);

/* ------------------------------------------------------------------ frontend
 * Replaces librosa.feature.melspectrogram + librosa.power_to_db as called at
 * main.py:117-125, data/dataset.py:155-156 and :195-196:
 *   center zero-pad n_fft/2, frames of 2048 @ hop, periodic Hann, |rFFT|^2,
 *   Slaney mel filterbank (fmin 0, fmax sr/2, norm 'slaney'), 10*log10(max(1e-10,S)),
 *   clamp to (per-chunk max - 80 dB).                                             */

/* Host-side tables (no GPU needed): dense filterbank (n_mels x 1025) as
 * librosa.filters.mel builds it; used by tests and by mt_mel_plan_init. */
int    mt_mel_filterbank_host(float* fb_host, int sr, int n_mels);
int    mt_mel_num_frames(int n_samples, int hop);           /* 1 + n_samples / hop */

/* Device-resident immutable tables for one (sr, hop, n_mels); `desc` (host memory, filled by
 * mt_mel_plan_init) carries what the launcher needs to know about them. */
typedef struct { int sr, hop, n_mels, ell_rows; } mt_mel_desc;
size_t mt_mel_plan_bytes(int n_mels);
int    mt_mel_plan_init(void* plan, size_t plan_bytes, int sr, int hop, int n_mels, mt_mel_desc* desc,
                        mt_stream_t stream);

/* wave[B][n_samples] f32 -> mel_db[B][n_mels][T] f32 (T = mt_mel_num_frames).
 * chunk_max_power[B] (f32 bit patterns, >= 0) receives each chunk's max mel POWER.
 * apply_clamp != 0: mel_db is final (clamped at max-80 dB), as power_to_db returns it.
 * apply_clamp == 0: mel_db is left unclamped; the consumer (mt_conv1_*) applies
 *                   max(x, 10*log10(max(1e-10, chunk_max_power[b])) - 80) on load.  */
int    mt_mel_db_f32(const void* plan, const mt_mel_desc* desc, const float* wave, int B, int n_samples,
                     float* mel_db, float* chunk_max_power, int apply_clamp, mt_stream_t stream);

/* ------------------------------------------------------------------ CNN blocks
 * CNNRNNModel.cnn (cnn_rnn_model.py:29-39; Large: conv1, :178-183), eval mode; BatchNorm
 * running statistics are folded into (w, bias) by the host at load_state_dict time.
 * conv1: mel[B][n_mels][T] f32 (+ optional per-chunk dB floor from chunk_max_power, may be
 *        NULL) -> act1[B][n_mels/2][T][32] bf16, channels-last.
 *        w = folded Conv2d(1,32,3x3) weights [32][9] f32, bias [32] f32.
 * conv2: act1 -> X0[(t*B+b)*ldx + fo*64 + co] bf16 (the LSTM layer-0 GEMM A matrix; the
 *        reference's feature order c*F+f, cnn_rnn_model.py:60-62, is absorbed into W_ih's
 *        column order at pack time).  w2 = folded Conv2d(32,64,3x3) [64][tap=kh*3+kw][32] bf16. */
int    mt_conv1_bn_relu_pool(const float* mel, const float* chunk_max_power, const float* w, const float* bias,
                             void* act1, int B, int n_mels, int T, mt_stream_t stream);
int    mt_conv2_bn_relu_pool(const void* act1, const void* w2, const float* bias, void* X0, int ldx,
                             int B, int F1, int T, mt_stream_t stream);
/* The same with the 16-bit operand type chosen by the caller (dt = MT_DT_BF16 | MT_DT_F16). */
int    mt_conv1_bn_relu_pool_dt(const float* mel, const float* chunk_max_power, const float* w, const float* bias,
                                void* act1, int B, int n_mels, int T, int dt, mt_stream_t stream);
/* conv1 + conv2 in one kernel (models/cnn_rnn_model.py:29-39): X0 bit-identical to mt_conv1_bn_relu_pool_dt followed by
 * mt_conv2_bn_relu_pool_dt on the same operands; act1 is computed per tile in LDS and never exists in HBM.                */
int    mt_conv12_bn_relu_pool_dt(const float* mel, const float* chunk_max_power, const float* w1, const float* b1,
                                 const void* w2, const float* b2, void* X0, int ldx, int B, int n_mels, int T, int dt,
                                 mt_stream_t stream);
int    mt_conv2_bn_relu_pool_dt(const void* act1, const void* w2, const float* bias, void* X0, int ldx,
                                int B, int F1, int T, int dt, mt_stream_t stream);

/* ------------------------------------------------------------------ GEMM (bf16 MFMA, f32 accumulate)
 * C[M][N] (f32) = A[M][K] (bf16) * W[N][K]^T (bf16) + bias[N] (f32, may be NULL).
 * lda/ldw/ldc in elements.  The caller's A buffer must be readable up to row
 * roundup(M,128)-1 and W up to row roundup(N,128)-1 (pad rows may hold anything finite or
 * not: they only feed discarded outputs); K % 64 == 0.  Replaces the nn.Linear / nn.LSTM
 * input-projection matmuls (cnn_rnn_model.py:45-52,:55,:212-228,:250-256).               */
int    mt_gemm_bf16_f32acc(const void* A, int lda, const void* W, int ldw, const float* bias,
                           float* C, int ldc, int M, int N, int K, mt_stream_t stream);
/* LSTM input projection for BOTH directions of one layer: X[(t*B+b)][K] bf16, W_ih =
 * [fwd 4H rows; reverse 4H rows][K] bf16, bias[8H] = b_ih + b_hh, output in the layout
 * the recurrence streams: gx[b/32][t][dir][j/8][gate][j%8][b%32] f32 (mt_lstm_gx_bytes). */
int    mt_gemm_lstm_gx(const void* X, int ldx, const void* W_ih, int ldw, const float* bias, float* gx,
                       int B, int T, int H, int K, mt_stream_t stream);
/* Final projection with the reference's transpose fused: logits[b][n][t], rows m = t*B+b. */
int    mt_gemm_logits(const void* X, int ldx, const void* W, int ldw, const float* bias, float* logits,
                      int B, int T, int N, int K, mt_stream_t stream);
/* The layer-to-layer input projection and the final fc with A read STRAIGHT from the previous LSTM layer's hx images (f16
 * operands: the exchanged h is the operand type already), K = 2*Hprev, column k = dir*Hprev + unit, Hprev % 64 == 0: no
 * re-layout pass between the layers.  hx_prev as mt_lstm_bidir_fwd* wrote it for the same B, T.                      */
int    mt_gemm_lstm_gx_from_hx(const float* hx_prev, const void* W_ih, int ldw, const float* bias, float* gx,
                               int B, int T, int H, int Hprev, mt_stream_t stream);
/* The same; gx_f16 != 0 stores gx as f16 (see MT_GX_F16).                                                              */
int    mt_gemm_lstm_gx_from_hx_ex(const float* hx_prev, const void* W_ih, int ldw, const float* bias, float* gx,
                                  int B, int T, int H, int Hprev, int gx_f16, mt_stream_t stream);
int    mt_gemm_logits_from_hx(const float* hx_prev, const void* W, int ldw, const float* bias, float* logits,
                              int B, int T, int N, int Hprev, mt_stream_t stream);
/* The three GEMMs above with the operand type of A and W chosen by the caller (dt = MT_DT_BF16 | MT_DT_F16). */
int    mt_gemm_f32acc_dt(const void* A, int lda, const void* W, int ldw, const float* bias,
                         float* C, int ldc, int M, int N, int K, int dt, mt_stream_t stream);
int    mt_gemm_lstm_gx_dt(const void* X, int ldx, const void* W_ih, int ldw, const float* bias, float* gx,
                          int B, int T, int H, int K, int dt, mt_stream_t stream);
int    mt_gemm_logits_dt(const void* X, int ldx, const void* W, int ldw, const float* bias, float* logits,
                         int B, int T, int N, int K, int dt, mt_stream_t stream);
/* The two input projections with MT_GEMM_SCHED_BYTES of device scratch (`sched`, 16-byte aligned; zeroed by the call on
 * `stream`; nothing else may touch it until the launch has finished: one block per GEMM call of a forward in flight).  With it,
 * and for f16 gx (MT_GX_F16 / gx_f16) with B % 32 == 0, 256 | H and an even number >= 16 of 64-wide K-tiles, the projection runs
 * as PERSISTENT tiles: a workgroup walks a dynamic sequence of 256 x 256 tiles as one K-stream and the finished tile's f16
 * output leaves under the next tile's main loop (csrc/gemm.hip, gemm256p_kernel).  sched == NULL or any other shape: the
 * one-tile-per-workgroup kernels, bit-identical results.                                                                   */
#define MT_GEMM_SCHED_BYTES 64
size_t mt_gemm_sched_bytes(void);
int    mt_gemm_lstm_gx_sched(const void* X, int ldx, const void* W_ih, int ldw, const float* bias, float* gx,
                             int B, int T, int H, int K, int dt, void* sched, mt_stream_t stream);
int    mt_gemm_lstm_gx_from_hx_sched(const float* hx_prev, const void* W_ih, int ldw, const float* bias, float* gx,
                                     int B, int T, int H, int Hprev, int gx_f16, void* sched, mt_stream_t stream);

/* ------------------------------------------------------------------ bidirectional LSTM recurrence
 * nn.LSTM(batch_first, bidirectional) as the reference runs it (cnn_rnn_model.py:45-52,
 * :69-70): gate order i,f,g,o, zero initial state; fp32 state/gates/accumulation, W_hh h on the
 * f16 MFMA (W_hh and the exchanged h rounded to f16).  H % 16 == 0, H <= 1024.
 * w_hh = [fwd; reverse] x [4H][H] f32.  hx receives every step's h as f16 in the MFMA-operand
 * layout hx[b/32][t][dir][k/16][((k/8)%2)*32 + b%32][k%8] (mt_lstm_hx_bytes; typed float* here,
 * read it through mt_lstm_relayout_* / mt_lstm_unpack_f32).  sync_ws: mt_lstm_sync_bytes(); after the
 * stream has drained its word 0 is 0 (ok), 1 + step (flag spin timed out) or
 * 0x40000000 + step (payload spin timed out).                                              */
size_t mt_lstm_gx_bytes(int B, int T, int H);
size_t mt_lstm_hx_bytes(int B, int T, int H);
size_t mt_lstm_sync_bytes(int B, int H);
int    mt_lstm_bidir_fwd(const float* gx, const float* w_hh, float* hx, void* sync_ws, size_t sync_bytes,
                         int B, int T, int H, mt_stream_t stream);
/* A layer fed by the previous LSTM layer, with its input projection fused into the recurrence (no gx buffer, no GEMM, no
 * re-layout between the layers): gate pre-activations = W_ihx x_t + bias + W_hh h_{t-1}, x_t = the previous layer's h of
 * step t (both directions) read from ITS hx.  w_ihx [2][4H][2H] f32 (column = dir'*H + unit, zero columns for padded
 * units), bias [2][4H] = b_ih + b_hh.  H <= 512.                                                                       */
int    mt_lstm_bidir_fwd_xproj(const float* hx_prev, const float* w_ihx, const float* bias, const float* w_hh, float* hx,
                               void* sync_ws, size_t sync_bytes, int B, int T, int H, mt_stream_t stream);
/* As mt_lstm_bidir_fwd, with flags: 0, or MT_GX_F16 (gx holds f16 gate pre-activations).  (The XCD-local hand-off variants
 * that this entry point used to select -- "mode 1" with 8 and "mode 2" with 16 units per workgroup -- lost to the agent-scope
 * kernel with interleaved batch groups at every schedule and are gone, and mt_xcd_census with them.)                       */
int    mt_lstm_bidir_fwd_ex(const float* gx, const float* w_hh, float* hx, void* sync_ws, size_t sync_bytes,
                            int B, int T, int H, int flags, mt_stream_t stream);
/* CUs held by persistent launches still pending on streams other than `stream` on the current device (see above). */
int    mt_persistent_cus_in_flight(mt_stream_t stream);
/* hx -> X[(t*B+b)*ldx + dir*H + j] bf16 (next GEMM's A) / y[b][t][dir*H + j] f32 (torch layout). */
int    mt_lstm_relayout_bf16(const float* hx, void* X, int ldx, int B, int T, int H, mt_stream_t stream);
int    mt_lstm_unpack_f32(const float* hx, float* y, int B, int T, int H, mt_stream_t stream);

/* ------------------------------------------------------------------ whole-model forward
 * CNNRNNModel.forward in eval mode (cnn_rnn_model.py:57-74).  All pointers are device
 * pointers to tensors the host packed once from a reference state_dict
 * (music-transcription_amd/model.py: pack_cnnrnn).                                        */
#define MT_MAX_LSTM_LAYERS 8
typedef struct {
    int n_mels;                              /* input mel bins                                    */
    int hidden;                              /* LSTM hidden size (1..1024); laid out padded to 16 */
    int layers;                              /* LSTM layers (<= MT_MAX_LSTM_LAYERS)               */
    int operand_dtype;                       /* MT_DT_BF16 / MT_DT_F16: type of every 16-bit weight below and of the */
                                             /*   activations between the kernels (model.py packs f16 for inference) */
    const float* conv1_w;                    /* [32][9]  BN-folded                                */
    const float* conv1_b;                    /* [32]                                              */
    const void*  conv2_w;                    /* 16-bit [64][9][32] BN-folded                        */
    const float* conv2_b;                    /* [64]                                              */
    const void*  w_ih[MT_MAX_LSTM_LAYERS];   /* bf16 [roundup(8Hp,128)][K_l], Hp = roundup(H,16), */
                                             /*   gate row p*Hp + j (zero for j >= H); layer 0 columns in */
                                             /*   fo*64+co order, K_0 = (n_mels/4)*64;            */
                                             /*   K_l = roundup(2H,64), zero-padded, for l > 0    */
    const float* b_gates[MT_MAX_LSTM_LAYERS];/* f32 [8Hp] = b_ih + b_hh, [fwd; reverse]           */
    const float* w_hh[MT_MAX_LSTM_LAYERS];   /* f32 [2][4Hp][Hp]                                  */
    const void*  fc_w;                       /* bf16 [128][roundup(2H,64)], rows >= 88 zero       */
    const float* fc_b;                       /* f32 [88]                                          */
    const float* w_ihx[MT_MAX_LSTM_LAYERS];  /* layers l > 0 (optional, NULL = project with the GEMM): f32 [2][4Hp][2Hp], */
                                             /*   column dir'*Hp + k: the input projection fused into the recurrence      */
                                             /*   (mt_lstm_bidir_fwd_xproj; agent-scope hand-off, Hp <= 512)               */
} mt_cnnrnn_weights;

size_t mt_cnnrnn_workspace_bytes(const mt_cnnrnn_weights* w, int B, int T);
/* 1 when mt_cnnrnn_forward* runs conv1 + conv2 as one kernel (opt-in: MT_CONV_FUSED=1 in the environment; same X0 bit for bit, act1
 * never in HBM, not faster): the conv1 stage of mt_cnnrnn_forward_ex's event list is then empty.                          */
int    mt_cnnrnn_conv_fused(void);
size_t mt_cnnrnn_status_offset(const mt_cnnrnn_weights* w, int B, int T, int layer);
/* mel[B][n_mels][T] f32 (dB; chunk_max_power may be NULL when mel is already clamped)
 * -> logits[B][88][T] f32.                                                                */
int    mt_cnnrnn_forward(const mt_cnnrnn_weights* w, const float* mel, const float* chunk_max_power,
                         int B, int T, float* logits, void* workspace, size_t workspace_bytes,
                         mt_stream_t stream);
/* Same, recording the caller's hipEvent_t handles (events[0] before the first kernel, then
 * one after each stage: conv1, conv2, (input-projection GEMM, recurrence, re-layout) x layers,
 * fc) so that a benchmark can time each kernel on the launch stream.  events may be NULL.   */
int    mt_cnnrnn_num_stages(int layers);
int    mt_cnnrnn_forward_ex(const mt_cnnrnn_weights* w, const float* mel, const float* chunk_max_power,
                            int B, int T, float* logits, void* workspace, size_t workspace_bytes,
                            void* const* events, int n_events, mt_stream_t stream);

/* ------------------------------------------------------------------ CNNRNNModelLarge pieces
 * Channels-last bf16 convolution with folded BatchNorm (ResidualBlock, freq_aware_conv;
 * cnn_rnn_model.py:76-99,:186-202):  out = act(conv_{KHx3,pad(KH/2,1)}(A) [+ conv_1x1(S)] + bias),
 * optional MaxPool2d((2,1)).  A [B][F][T][C1], S [B][F][T][C2] (residual skip input, may be NULL),
 * W [Cout][KH*3*C1 + C2] bf16 (K order: tap = kh*3+kw, ci; then the skip channels).
 * out_mode 0: [B][Fout][T][Cout] bf16;  1: GEMM-A rows X[(t*B+b)*ldx + fo*Cout + co].        */
int    mt_conv_cl_bf16(const void* A, const void* S, const void* W, const float* bias, void* out,
                       int B, int F, int T, int C1, int C2, int Cout, int KH, int relu, int pool,
                       int out_mode, int ldx, mt_stream_t stream);
/* Batched GEMMs: batch z -> (z1, z2) = (z / zdiv, z % zdiv), element offsets z1*s?1 + z2*s?2.  */
int    mt_gemm_batched_f32(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw,
                           long long sW1, long long sW2, const float* bias, float* C, int ldc,
                           long long sC1, long long sC2, int M, int N, int K, int batch, int zdiv,
                           mt_stream_t stream);
int    mt_gemm_batched_bf16out(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw,
                               long long sW1, long long sW2, const float* bias, void* C, int ldc,
                               long long sC1, long long sC2, int M, int N, int K, int batch, int zdiv,
                               int relu, mt_stream_t stream);
/* hx -> feature rows with a column offset, dropping padded hidden units (k >= Hv); X (bf16) and/or Y (f32). */
int    mt_lstm_relayout_ex(const float* hx, void* X, int ldx, float* Y, int ldy, int col_off,
                           int B, int T, int H, int Hv, mt_stream_t stream);
/* `_dt` forms of the Large model's pieces: operand type (and type of a 16-bit output) dt = MT_DT_BF16 | MT_DT_F16. */
int    mt_conv_cl_dt(const void* A, const void* S, const void* W, const float* bias, void* out,
                     int B, int F, int T, int C1, int C2, int Cout, int KH, int relu, int pool,
                     int out_mode, int ldx, int dt, mt_stream_t stream);
int    mt_gemm_batched_f32_dt(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw,
                              long long sW1, long long sW2, const float* bias, float* C, int ldc,
                              long long sC1, long long sC2, int M, int N, int K, int batch, int zdiv, int dt,
                              mt_stream_t stream);
int    mt_gemm_batched_h16out_dt(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw,
                                 long long sW1, long long sW2, const float* bias, void* C, int ldc,
                                 long long sC1, long long sC2, int M, int N, int K, int batch, int zdiv,
                                 int relu, int dt, mt_stream_t stream);
int    mt_lstm_relayout_dt(const float* hx, void* X, int ldx, float* Y, int ldy, int col_off,
                           int B, int T, int H, int Hv, int dt, mt_stream_t stream);
/* MultiHeadAttention pieces (cnn_rnn_model.py:118-139): P = softmax(clamp(S*scale, +-clip)) -> bf16 with
 * the key axis zero-padded to Tp; V^T per (chunk, head); y = LayerNorm(resid + proj) -> bf16.       */
int    mt_attn_softmax_clamped(const float* S, int lds, void* P, int Tp, int T, long long rows,
                               float scale, float clip, mt_stream_t stream);
int    mt_attn_transpose_v(const void* qkv, int ld3, int voff, void* VT, int B, int T, int Tp,
                           int heads, int dp, mt_stream_t stream);
/* Fused attention core (eval mode): ao[(t*B+b)*ldo + head*dp + d] = sum_j softmax_j(clamp(Q K^T * scale, +-clip))[t][j] V[j][d] per
 * (chunk, head), scores never written (cnn_rnn_model.py:118-139: scale, clamp BEFORE softmax -- so exp() needs no running
 * maximum).  qkv [(t*B+b)][ld3]: Q at column head*dp, K at Ca + head*dp (16-bit, dt); VT as mt_attn_transpose_v writes it for
 * Tp = roundup(T, 64).  dp in {64, 128, 192} (else MT_EUNSUPPORTED), clip <= 10.  csrc/attn_fused.hip.                       */
int    mt_attn_fused_clamped(const void* qkv, int ld3, int Ca, const void* VT, int Tp, int B, int T, int heads, int dp,
                             float scale, float clip, void* ao, int ldo, int dt, mt_stream_t stream);
int    mt_layernorm_residual(const float* resid, int ldr, const float* proj, int ldp, const float* gamma,
                             const float* beta, void* y, int ldy, long long rows, int n, float eps,
                             mt_stream_t stream);
int    mt_attn_softmax_clamped_dt(const float* S, int lds, void* P, int Tp, int T, long long rows,
                                  float scale, float clip, int dt, mt_stream_t stream);
int    mt_layernorm_residual_dt(const float* resid, int ldr, const float* proj, int ldp, const float* gamma,
                                const float* beta, void* y, int ldy, long long rows, int n, float eps, int dt,
                                mt_stream_t stream);

/* CNNRNNModelLarge.forward, eval mode (cnn_rnn_model.py:262-348).  Packed by model.py: pack_large. */
typedef struct {
    int n_mels, hidden, layers, hidden_local;     /* real sizes (hidden_local = hidden / 2)            */
    int use_attention, use_heads, heads, head_dim_pad;  /* head_dim padded to a multiple of 64         */
    float attn_scale;                             /* (real head_dim)^-1/2                              */
    int operand_dtype;                            /* MT_DT_BF16 / MT_DT_F16 (as mt_cnnrnn_weights)     */
    const float* conv1_w; const float* conv1_b;   /* [32][9], [32]                                     */
    const void*  rb1c1_w; const float* rb1c1_b;   /* bf16 [64][9*32]                                   */
    const void*  rb1c2_w; const float* rb1c2_b;   /* bf16 [64][9*64 + 32]  (conv2 + 1x1 skip)          */
    const void*  rb2c1_w; const float* rb2c1_b;   /* bf16 [128][9*64]                                  */
    const void*  rb2c2_w; const float* rb2c2_b;   /* bf16 [128][9*128 + 64]                            */
    const void*  fa_w;    const float* fa_b;      /* bf16 [256][21*128]                                */
    const void*  main_w_ih[MT_MAX_LSTM_LAYERS]; const float* main_b[MT_MAX_LSTM_LAYERS]; const float* main_w_hh[MT_MAX_LSTM_LAYERS];
    const void*  local_w_ih; const float* local_b; const float* local_w_hh;
    const void*  qkv_w;  const float* qkv_b;      /* bf16 [roundup(3*heads*dp,128)][Cp], Cp = roundup(comb,64) */
    const void*  proj_w; const float* proj_b;     /* bf16 [roundup(comb,128)][heads*dp]                */
    const float* ln_g;   const float* ln_b;       /* [comb]                                            */
    const void*  shared_w; const float* shared_b; /* bf16 [roundup(hidden,128)][Cp]                    */
    const void*  heads_w;  const float* heads_b;  /* bf16 [384][roundup(hidden,64)]: frame, onset, offset */
    const void*  fc_w;     const float* fc_b;     /* no-heads variant: bf16 [128][Cp]                  */
    const float* main_w_ihx[MT_MAX_LSTM_LAYERS];  /* main layers l > 0, optional: f32 [2][4Hp][2Hp] for the input projection */
                                                  /*   fused into the recurrence (as mt_cnnrnn_weights.w_ihx)                 */
} mt_cnnrnn_large_weights;
size_t mt_cnnrnn_large_workspace_bytes(const mt_cnnrnn_large_weights* w, int B, int T);
size_t mt_cnnrnn_large_status_offset(const mt_cnnrnn_large_weights* w, int B, int T, int idx);
/* logits3: [3][B][88][T] (frame, onset, offset) when use_heads, else [B][88][T]. */
int    mt_cnnrnn_large_forward(const mt_cnnrnn_large_weights* w, const float* mel, const float* chunk_max_power,
                               int B, int T, float* logits3, void* workspace, size_t workspace_bytes,
                               mt_stream_t stream);
/* As above; with side_stream + two caller-owned hipEvent_t (all three non-NULL) the local LSTM branch runs on side_stream
 * beside the main LSTM stack (fork after the CNN, join before the attention).                                        */
int    mt_cnnrnn_large_forward_ex(const mt_cnnrnn_large_weights* w, const float* mel, const float* chunk_max_power,
                                  int B, int T, float* logits3, void* workspace, size_t workspace_bytes,
                                  mt_stream_t stream, mt_stream_t side_stream, void* ev_fork, void* ev_join);

/* Everything on `stream`, recording the caller's hipEvent_t handles at the stage boundaries (events[0] before the first kernel,
 * then one after each of the mt_cnnrnn_large_num_stages(layers) stages: conv1, res_block1, res_block2, freq_aware_conv, local
 * LSTM (projection, recurrence, re-layout), main LSTM layers (projection, recurrence, re-layout) x layers, attention +
 * LayerNorm, heads) so that a benchmark can time each kernel group on the launch stream.                               */
int    mt_cnnrnn_large_num_stages(int layers);
int    mt_cnnrnn_large_forward_ev(const mt_cnnrnn_large_weights* w, const float* mel, const float* chunk_max_power,
                                  int B, int T, float* logits3, void* workspace, size_t workspace_bytes,
                                  void* const* events, int n_events, mt_stream_t stream);

/* ------------------------------------------------------------------ loss, prediction, F1
 * Masked BCE-with-logits (transcription_model.py:110-162,:196-217):
 *   loss[0] (=, or += when accumulate) weight * sum_{valid} bce(logit, target) / max(n_valid_frames*P, 1)
 *   grad (may be NULL) = d loss / d logits = weight * (sigmoid(x) - y) * mask / denom.
 * lengths (int64, device, may be NULL = all frames valid); n_valid_frames = sum_b min(lengths[b], T)
 * (or B*T), known to the host.  workspace: mt_bce_workspace_bytes().  Bitwise reproducible.  */
size_t mt_bce_workspace_bytes(void);
int    mt_bce_masked_fwd_bwd(const float* logits, const float* targets, const long long* lengths,
                             long long n_valid_frames, float weight, int accumulate, float* loss, float* grad,
                             void* workspace, size_t workspace_bytes, int B, int P, int T, mt_stream_t stream);
/* onset/offset targets from a roll (transcription_model.py:176-185): rows x T, diff along T. */
int    mt_onset_offset_targets(const float* roll, float* onset, float* offset, long long rows, int T,
                               mt_stream_t stream);
/* roll = (sigmoid(logits) > threshold) as float {0,1} (transcription_model.py:262-265, main.py:153-156). */
int    mt_predict_threshold(const float* logits, float* roll, long long n, float threshold, mt_stream_t stream);
/* counts[b] = {TP, FP, FN} (uint64) over the first lengths[b] frames (evaluate.py:361-373). */
int    mt_f1_counts(const float* pred, const float* target, const long long* lengths,
                    unsigned long long* counts, int B, int P, int T, mt_stream_t stream);
/* counts[b][k] = {TP, FP, FN} of (sigmoid(logits) > thresholds[k]) for K <= 16 thresholds in one pass:
 * the data side of evaluate.py's threshold tuning (:524-618) without re-running the model.          */
int    mt_f1_sweep_counts(const float* logits, const float* target, const long long* lengths,
                          const float* thresholds, int K, unsigned long long* counts, int B, int P, int T,
                          mt_stream_t stream);

/* Roll -> notes on the device: combine_piano_rolls + the run-length part of pianoroll_to_midi (main.py:164-226,
 * scripts/evaluate.py:54-88).  The NB chunks of src [NB][P][T] are one roll of NB*T frames per pitch; src_mode 0: logits
 * (active = sigmoid(x) > threshold, as mt_predict_threshold), 1: roll values (active = x > 0).  counts[p] = notes of pitch p;
 * note k of pitch p is frames [starts[i], ends[i]) at i = sum_{q<p} counts[q] + k -- the reference's note order.  Nothing is
 * written for a pitch whose notes would not fit `capacity`: the host compares sum(counts) with it.  Only the notes leave the GPU. */
int    mt_roll_to_notes(const float* src, int src_mode, float threshold, int NB, int P, int T, int* counts, int* starts, int* ends,
                        int capacity, mt_stream_t stream);

/* ------------------------------------------------------------------ optimizer step (training)
 * clip_grad_norm_(max_norm) + torch.optim.Adam with coupled L2 weight decay over flat f32 buffers
 * (train_transcriber.py:134-144, train_cnn.py:290); a NaN/Inf gradient norm skips the step (:137-142).
 * step = 1-based step count.  stats (2 floats, may be NULL) = {norm before clipping, 1 if stepped else 0}.
 * With data parallelism, all-reduce `grads` over RCCL before calling this: the mean, or (_ex) the SUM with
 * grad_scale = 1 / world -- the scale is applied on the fly in the norm and in the update, `grads` is not written.
 * _ex, keep_ranges: HOST array of n_keep <= 16 ascending disjoint [lo, hi) element ranges (NULL / 0 = everything).
 * Only those elements are counted in the norm, clipped and updated; the rest keep params and moments -- torch's
 * behaviour for parameters whose .grad is None (the onset / offset heads under the reference's frame-only loss,
 * train_transcriber.py:119 + cnn_rnn_model.py:343-349: no weight decay, no moment update).                  */
size_t mt_adam_workspace_bytes(void);
int    mt_adam_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                         float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                         int step, float* stats, void* workspace, size_t workspace_bytes, mt_stream_t stream);
int    mt_adam_clip_step_ex(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                            int step, float grad_scale, const long long* keep_ranges, int n_keep,
                            float* stats, void* workspace, size_t workspace_bytes, mt_stream_t stream);

/* ------------------------------------------------------------------ training step
 * The kernels behind `loss.backward()` of train/train_transcriber.py:130 for CNNRNNModel in train mode
 * (cnn_rnn_model.py:29-74: BatchNorm2d with BATCH statistics, ReLU, MaxPool2d((2,1)), nn.LSTM, nn.Linear),
 * orchestrated by music-transcription_amd/train_step.py.  Dense contractions of the backward pass reuse
 * mt_gemm_* and mt_conv_cl_bf16 on operands these entry points lay out.  bf16 = raw uint16 bits.          */
/* conv1 (1->32, 3x3) pre-BN statistics, recomputed from the input: sums64[c] = sum z, sums64[32+c] = sum z^2
 * over all (b, f, t); x [B][F][T] f32, w [32][9], bias [32] (raw parameters).  Zeroes sums64 first.        */
int    mt_conv1_stats(const float* x, const float* w, const float* bias, double* sums64, int B, int F, int T,
                      mt_stream_t stream);
/* sums (sum, sum of squares over `count` elements per channel) -> batch mean / rstd = 1/sqrt(var_biased+eps);
 * running_mean/var (may be NULL) updated as nn.BatchNorm2d does (momentum, unbiased variance); optionally the
 * conv parameters with the batch statistics folded in: w_folded = w*gamma*rstd [C][taps],
 * b_folded = (b-mean)*gamma*rstd + beta (pass NULL to skip).                                              */
int    mt_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* mean_out,
                      float* rstd_out, int C, const float* w, const float* b, float* w_folded, float* b_folded,
                      int taps, mt_stream_t stream);
/* z [N][C] bf16 channels-last, C in {32,64,128,256}: sums[0..C) = sum z, sums[C..2C) = sum z^2 (zeroed first). */
int    mt_bn_stats_cl(const void* z, long long N, int C, double* sums, mt_stream_t stream);
/* z [B][F][T][64] bf16 (pre-BN conv2 output) -> X[(t*B+b)*ldx + fo*64 + c] bf16 = BN + ReLU + MaxPool((2,1)). */
int    mt_bn_relu_pool_apply(const void* z, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, void* X, int ldx, int B, int F, int T, mt_stream_t stream);
/* Backward of mt_bn_relu_pool_apply: dX f32 [(t*B+b)*ldd + fo*64 + c] -> dz [B][F][T][64] bf16, dgamma[64],
 * dbeta[64] (may be NULL).  dz_lo (may be NULL): the bf16 rounding remainder of dz (dz ~= dz + dz_lo to 2^-17),
 * for the cancellation-prone conv weight gradient.  sums128: 128 doubles of scratch.                       */
int    mt_bn_pool_bwd(const float* dX, int ldd, const void* z, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, double* sums128, void* dz, void* dz_lo,
                      float* dgamma, float* dbeta, int B, int F, int T, mt_stream_t stream);
/* Pool routing that ties exactly where the f32 conv results tie (nn.MaxPool2d on f32 activations routes a tie to the first
 * row; two rows that merely round to the same bf16 value are NOT a tie): mt_conv_cl_tie is the raw convolution (bf16 in,
 * channels-last bf16 out, no activation / pool) that also writes, per frequency-row pair (2fo, 2fo+1), position and channel,
 * the order bits of its f32 accumulators -- tie[(((b*(F/2) + fo)*T + t)*(Cout/32) + co/32)*2 + {0, 1}] bit co%32 =
 * {z(2fo) > z(2fo+1), z(2fo) < z(2fo+1)} -- and mt_bn_pool_bwd_tie routes with them (tie == NULL: as mt_bn_pool_bwd).   */
int    mt_conv_cl_tie(const void* A, const void* W, const float* bias, void* out, unsigned* tie, int B, int F, int T,
                      int C1, int Cout, int KH, mt_stream_t stream);
int    mt_bn_pool_bwd_tie(const float* dX, int ldd, const void* z, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, double* sums128, void* dz, void* dz_lo,
                          float* dgamma, float* dbeta, const unsigned* tie, int B, int F, int T, mt_stream_t stream);
/* Weight and bias gradient of the second conv (Conv2d(32, 64, 3, padding=1), cnn_rnn_model.py:35) from the
 * channels-last activation a1 [B][F][T][32] and the two bf16 pieces of dz [B][F][T][64] (mt_bn_pool_bwd), with the
 * POSITION as the MFMA contraction index: no im2col, no transposed copy.  n_wg persistent workgroups
 * (mt_conv2_wgrad_workgroups()) each leave a partial P[wg][64][288] (column = tap*32 + ci) and Pb[wg][64]
 * (bias: hi piece only); mt_sum_slices_f32 adds them in a fixed order.                                          */
int    mt_conv2_wgrad_workgroups(void);
int    mt_conv2_wgrad(const void* a1, const void* dz_hi, const void* dz_lo, float* P, float* Pb, int n_wg,
                      int B, int F, int T, mt_stream_t stream);
/* Layer-0 W_ih of one direction of an nn.LSTM (f32 [4H][C*F], reference feature order c*F + f: models/cnn_rnn_model.py:60-62,
 * :292-294) -> rows row0 + p*Hp + j of the projection GEMM's 16-bit operand `out` (row pitch ldo elements) in the kernels' feature
 * order f*C + c; rows j in [H, Hp) are written as zeros.  dt = MT_DT_BF16 | MT_DT_F16.  C*(F|1)*4 bytes of LDS (<= 64 KB).
 * What _pack_bilstm (music-transcription_amd/model.py) did with an index gather + cast per direction.                     */
int    mt_pack_wih_cf(const float* w, void* out, long long ldo, int row0, int H, int Hp, int C, int F, int dt,
                      mt_stream_t stream);
/* One job of mt_pack_jobs: dst rectangle [Rp][Cp] (row pitch ld elements; dt = MT_DT_BF16 | MT_DT_F16 | MT_PACK_F32) from f32 sources.
 * With r = r1*Rn2 + r2 and c = c1*Cn2 + c2: dst[r][c] = src[r1*sr1 + r2*sr2 + c1*sc1 + c2*sc2] (+ src2[same offset] when src2 != NULL)
 * where r1 < R1v, r2 < R2v, c1 < C1v, c2 < C2v, and 0 elsewhere (strides in elements, may be negative).  tr != 0: the source is
 * contiguous along the dst rows.  tile0 = number of 32 x 32 tiles of the jobs in front of this one (ascending; job 0 has 0).
 * The table replaces the per-step torch expressions of pack_train_large (music-transcription_amd/train_step_large.py), which
 * restate what the reference keeps implicit in nn.Module parameters (/root/reference/models/cnn_rnn_model.py:186-260).          */
#define MT_PACK_F32 2
typedef struct mt_pack_job {
    const void* src; const void* src2; void* dst;
    long long ld, sr1, sr2, sc1, sc2;
    int dt, Rp, Cp, Rn2, R1v, R2v, Cn2, C1v, C2v, tr, tile0, reserved;
} mt_pack_job;
/* Runs a table of njobs jobs (DEVICE memory, ntiles = total number of 32 x 32 tiles) in one launch.                        */
int    mt_pack_jobs(const void* jobs_dev, int njobs, int ntiles, mt_stream_t stream);
/* dst[c*ldd + r] = src[r*lds + c] (bf16), r < R, c < C; every dst element with c < Cd, r < ldd is written
 * (zero outside the source).                                                                               */
int    mt_transpose_bf16(const void* src, long long lds, long long R, int C, void* dst, long long ldd, int Cd,
                         mt_stream_t stream);
/* dst[i0][i1][i2][i3] (contiguous) = alpha * src[i0*s0 + i1*s1 + i2*s2 + i3*s3]: GEMM-layout gradients ->
 * the reference's parameter shapes (state_dict order).                                                     */
int    mt_gather4_f32(const float* src, float* dst, int n0, int n1, int n2, int n3, long long s0, long long s1,
                      long long s2, long long s3, float alpha, mt_stream_t stream);
/* out[r*ldo + c] = sum_z P[z*stride + r*ldp + c]: reduction of split-K partial products (fixed order).     */
int    mt_sum_slices_f32(const float* P, long long stride, int ldp, int S, float* out, int ldo, int rows,
                         int cols, mt_stream_t stream);
/* out[r] = sum_{c<n} A[r*ld + c] (bf16 rows): bias gradients from the transposed GEMM operands.            */
int    mt_rowsum_bf16(const void* A, long long ld, long long n, float* out, int rows, mt_stream_t stream);
/* conv1 + BN + ReLU + pool backward with z1 recomputed from x: da [B][F/2][T][ldc] bf16 (channels 0..31) ->
 * dW [32][9], db [32], dgamma [32], dbeta [32].  scratch384: 384 doubles.                                  */
int    mt_conv1_bwd(const float* x, const float* w, const float* bias, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, const void* da, int ldc, double* scratch384,
                    float* dW, float* db, float* dgamma, float* dbeta, int B, int F, int T, mt_stream_t stream);
/* Train-mode recurrence: as mt_lstm_bidir_fwd; additionally the ACTIVATED gates overwrite gx in place and the
 * cell states go to cx [B/32][T][2][H/8][8][32] f32 (mt_lstm_cx_bytes).                                    */
size_t mt_lstm_cx_bytes(int B, int T, int H);
int    mt_lstm_bidir_fwd_train(float* gx_inout, const float* w_hh, float* hx, float* cx, void* sync_ws,
                               size_t sync_bytes, int B, int T, int H, mt_stream_t stream);
/* hx -> X[(t*B+b)*ldx + d*Hv + j] bf16 with nn.LSTM's inter-layer (inverted) dropout p; the mask is a
 * counter-based hash of (seed, layer, element) that mt_lstm_dh_relayout regenerates.                       */
int    mt_lstm_relayout_train(const float* hx, void* X, int ldx, int B, int T, int H, int Hv, float p,
                              unsigned seed, unsigned layer, mt_stream_t stream);
/* dX f32 [(t*B+b)*ld + d*Hv + j] (gradient of the layer output) -> dh [B/32][T][2][H/8][8][32] f32.        */
int    mt_lstm_dh_relayout(const float* dX, int ld, float* dh, int B, int T, int H, int Hv, float p,
                           unsigned seed, unsigned layer, mt_stream_t stream);
/* The same dh, straight from the GEMM that produces the gradient of a layer's output (dX = dY . W^T with
 * rows m = t*B+b, N = 2*Hv columns d*Hv + j, W [roundup(2 Hv, 128) rows][K] bf16): the layer above's input
 * gradient dG_{l+1} . W_ih_{l+1} (train/train_transcriber.py:104-131 loss.backward() through nn.LSTM,
 * cnn_rnn_model.py:45-52) or the fc layer's dL . W_fc (:54, :73) -- no f32 dX round trip, no re-layout pass.
 * Entries of padded units (j >= Hv) and chunks (b >= B) are NOT written: dh must be zero there already.       */
int    mt_gemm_lstm_dh(const void* dY, int ldy, const void* W, int ldw, float* dh, int B, int T, int H, int Hv,
                       int K, float p, unsigned seed, unsigned layer, mt_stream_t stream);
/* Backward through time (both directions of one layer; persistent kernel, lstm_bwd.hip): gates/cx from
 * mt_lstm_bidir_fwd_train, dh from mt_lstm_dh_relayout, w_hh [2][4H][H] f32 -> dgx: d(gate pre-activations)
 * as bf16 MFMA-operand images (mt_lstm_dgx_bytes).  H % 16 == 0, H <= 512.  part_ws: scratch for the per-step
 * partial products of the reduce-scatter (mt_lstm_bwd_part_bytes); sync_ws: mt_lstm_sync_bytes.             */
size_t mt_lstm_dgx_bytes(int B, int T, int H);
size_t mt_lstm_bwd_part_bytes(int B, int T, int H);
int    mt_lstm_bidir_bwd(const float* gates, const float* cx, const float* dh, const float* w_hh, void* dgx,
                         void* part_ws, size_t part_bytes, void* sync_ws, size_t sync_bytes, int B, int T, int H,
                         mt_stream_t stream);
/* The poison fill of part_ws alone (any stream), and the launch with flags bit 0 = "part_ws is already poisoned": lets the
 * caller hide the 1-GB fill of the next layer's workspace under the current layer's recurrence.                       */
int    mt_lstm_bwd_poison(void* part_ws, size_t part_bytes, int B, int T, int H, mt_stream_t stream);
int    mt_lstm_bidir_bwd_ex(const float* gates, const float* cx, const float* dh, const float* w_hh, void* dgx,
                            void* part_ws, size_t part_bytes, void* sync_ws, size_t sync_bytes, int B, int T, int H,
                            int flags, mt_stream_t stream);
/* dgx -> dG [(t*B+b)*ldg + d*4H + gate*H + j] bf16 and dGT [(d*4H + gate*H + j)*ldt + t*B + b] bf16
 * (pre-zeroed by the caller: padded rows / columns are not written).  Either output may be NULL (skipped). */
int    mt_lstm_dg_unpack(const void* dgx, void* dG, int ldg, void* dGT, long long ldt, int B, int T, int H,
                         mt_stream_t stream);
/* hx -> HT[(d*rows_per_dir + k)*ld + t*B + b] = bf16 h of the forward pass's PREVIOUS step (zero at the
 * sequence boundary): W operand of dW_hh[d] = dG_d^T . Hprev_d.                                            */
int    mt_lstm_hprev_t(const float* hx, void* HT, long long ld, int rows_per_dir, int B, int T, int H,
                       mt_stream_t stream);
/* dlogits [B][P][T] f32 -> dL [(t*B+b)*128 + p] bf16 and dLT [p*ldt + t*B + b] bf16 (p < 128, zero for p >= P). */
int    mt_dlogits_pack(const float* dlogits, void* dL, void* dLT, long long ldt, int B, int P, int T,
                       mt_stream_t stream);

/* ------------------------------------------------------------------ training step of CNNRNNModelLarge
 * The pieces between the dense contractions of `loss.backward()` for models/cnn_rnn_model.py:262-348 in train mode
 * (train/train_transcriber.py:104-131), orchestrated by music-transcription_amd/train_step_large.py.  All 16-bit
 * tensors are bf16.                                                                                                  */
/* mask[b*C + c] = keep ? 1/(1-p) : 0 -- nn.Dropout2d (cnn_rnn_model.py:188,:192,:202) zeroes whole channels of a sample. */
int    mt_dropout2d_mask(float* mask, int B, int C, float p, unsigned seed, unsigned layer, mt_stream_t stream);
/* out = Dropout2d(MaxPool2d((2,1))(ReLU(BN_a(za) [+ BN_b(zb)]))) with BATCH statistics (mean / rstd from mt_bn_finalize):
 * ResidualBlock.forward (cnn_rnn_model.py:93-99: zb = the 1x1 skip branch, no pool / pool1 behind res_block1) and
 * freq_aware_conv (:196-201).  za, zb [B][F][T][C] raw conv outputs (zb, mask2d may be NULL; relu, pool flags);
 * out_mode 0: [B][Fo][T][C], 1: GEMM-A rows X[(t*B+b)*ldx + fo*C + c].  C in {32, 64, 128, 256}.                   */
int    mt_bn_act_fwd(const void* za, const float* mean_a, const float* rstd_a, const float* gamma_a, const float* beta_a,
                     const void* zb, const float* mean_b, const float* rstd_b, const float* gamma_b, const float* beta_b,
                     const float* mask2d, void* out, int out_mode, int ldx, int B, int F, int T, int C, int relu, int pool,
                     mt_stream_t stream);
/* Backward of mt_bn_act_fwd.  Gradient of the output: dout_cl (bf16 [B][Fo][T][ldd_cl]) or dout_x (f32 GEMM-row
 * layout, ldd_x) -- exactly one.  Outputs: dza [B][F][T][pitch_a], dzb [B][F][T][pitch_b] (when zb), each with an
 * optional second bf16 piece (dza_lo, dzb_lo: the rounding remainder -- BatchNorm forces sum dz = 0 and sum dz*z = 0, so
 * the conv weight gradient is a heavily cancelling sum and runs over both pieces), parameter gradients (any may be
 * NULL).  sums: 3*C doubles of scratch.
 * Pool ties route to the first row, as nn.MaxPool2d does.                                                           */
int    mt_bn_act_bwd(const void* dout_cl, int ldd_cl, const float* dout_x, int ldd_x,
                     const void* za, const float* mean_a, const float* rstd_a, const float* gamma_a, const float* beta_a,
                     const void* zb, const float* mean_b, const float* rstd_b, const float* gamma_b, const float* beta_b,
                     const float* mask2d, double* sums, void* dza, int pitch_a, void* dza_lo, void* dzb, int pitch_b, void* dzb_lo,
                     float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                     int B, int F, int T, int C, int relu, int pool, mt_stream_t stream);
/* (r3) Weight gradient of a channels-last convolution (csrc/conv_wgrad.hip; replaces round 2's position planes + batched GEMM): what autograd computes for the
 * convolutions of ResidualBlock and freq_aware_conv (cnn_rnn_model.py:76-124,:186-196) under train_transcriber.py:130.
 *   out[co][ci][kh][kw] (f32, the reference's weight layout) = sum over (b, f, t) of
 *       (dz_hi + dz_lo)[b][f][t][co] * x[b][f + kh - KH/2][t + kw - KW/2][ci]            (zero outside the image)
 * dz_hi / dz_lo: bf16 [B][F][T][dz_pitch] (the gradient and its rounding remainder; dz_lo may be NULL), x: bf16
 * [B][F][T][x_pitch].  Cout % 64 == 0, Cin % 32 == 0, KH odd, KW = 3 or 1, pitches % 8 == 0, 16-byte aligned tensors,
 * F*T*pitch*2 < 2 GB.  ws: mt_conv_wgrad_ws_bytes() of scratch (partial sums per K split, added in a fixed order).      */
size_t mt_conv_wgrad_ws_bytes(int B, int F, int T, int Cout, int Cin, int KH, int KW);
int    mt_conv_wgrad(const void* dz_hi, const void* dz_lo, int dz_pitch, const void* x, int x_pitch, int B, int F, int T,
                     int Cout, int Cin, int KH, int KW, void* ws, size_t ws_bytes, float* out, mt_stream_t stream);
/* General form of mt_conv_cl_dt: A / S are channel slices (pitchA / pitchS elements between positions) and accum != 0
 * adds the result to `out` -- the input gradient of freq_aware_conv (256 output channels) is two calls.              */
int    mt_conv_cl_ex(const void* A, int pitchA, const void* S, int pitchS, const void* W, const float* bias, void* out,
                     int B, int F, int T, int C1, int C2, int Cout, int KH, int relu, int pool, int out_mode, int ldx,
                     int accum, int dt, mt_stream_t stream);
int    mt_transpose_bf16_batched(const void* src, long long lds, long long sstride, int R, int C, void* dst, long long ldd,
                                 long long dstride, int Cd, int batch, mt_stream_t stream);
/* MultiHeadAttention in train mode (cnn_rnn_model.py:128-133): Pd = dropout_p(softmax(clamp(S*scale, +-clip))) as the
 * bf16 GEMM operand; backward through dropout, softmax and clamp (no gradient where |S*scale| > clip):
 * dPd f32 [rows][ldp] -> dS bf16 [rows][Tp].  P is recomputed from S.                                               */
int    mt_attn_softmax_train(const float* S, int lds, void* P, int Tp, int T, long long rows, float scale, float clip,
                             float p, unsigned seed, unsigned layer, mt_stream_t stream);
int    mt_attn_clamped_bwd(const float* S, int lds, const float* dPd, int ldp, void* dS, int Tp, int T, long long rows,
                           float scale, float clip, float p, unsigned seed, unsigned layer, mt_stream_t stream);
/* LayerNorm(resid + proj) (cnn_rnn_model.py:322) with saved statistics stats[row] = {mean, rstd}, and its backward:
 * dy f32 -> dx f32 (the gradient of resid AND of proj); part = [mt_layernorm_residual_bwd_slices()][2][n] partial
 * column sums (row 0: dgamma, row 1: dbeta) for mt_sum_slices_f32.                                                  */
int    mt_layernorm_residual_train(const float* resid, int ldr, const float* proj, int ldp, const float* gamma,
                                   const float* beta, void* y, int ldy, float* stats, long long rows, int n, float eps,
                                   mt_stream_t stream);
int    mt_layernorm_residual_bwd_slices(void);
int    mt_layernorm_residual_bwd(const float* resid, int ldr, const float* proj, int ldp, const float* gamma,
                                 const float* stats, const float* dy, int ldd, float* dx, int ldx, float* part,
                                 long long rows, int n, mt_stream_t stream);
/* shared_fc backward through Dropout(ReLU(.)) given the layer's OUTPUT Y: dZ = (Y > 0) ? dY / (1-p) : 0 (bf16).      */
int    mt_heads_relu_dropout_bwd(const float* dY, int ldd, const void* Y, int ldy, void* dZ, int ldz, long long M, int N,
                                 float p, mt_stream_t stream);
/* dlogits f32 [NH][B][P][T] (frame / onset / offset heads) -> dL[(t*B+b)*ldl + head*P + p], dLT[(head*P+p)*ldt + t*B+b] bf16
 * (entries beyond NH*P are not written: the caller zeroes the buffers).                                             */
int    mt_dlogits_pack_heads(const float* dlogits, void* dL, int ldl, void* dLT, long long ldt, int NH, int B, int P, int T,
                             mt_stream_t stream);
/* Element-wise helpers: f32 rows -> bf16 GEMM operand (scaled); in-place inverted dropout (counter-based hash of the
 * element index m*N + n, or i); out = alpha*a + beta*b over rows.                                                   */
int    mt_f32_to_bf16_rows(const float* src, int lds, void* dst, int ldd, long long M, int N, float alpha, mt_stream_t stream);
int    mt_dropout_bf16_rows(void* X, int ld, long long M, int N, float p, unsigned seed, unsigned layer, mt_stream_t stream);
int    mt_dropout_f32(float* x, long long n, float p, unsigned seed, unsigned layer, mt_stream_t stream);
int    mt_axpby_rows_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo, long long M, int N,
                         float alpha, float beta, mt_stream_t stream);

/* ------------------------------------------------------------------ audio decode
 * PCM frames -> mono float at the target rate, replacing the host side of librosa.load(path, sr=16000, mono=True)
 * (main.py:76, data/dataset.py:124-130): channel mean + polyphase FIR, y[j] = sum_i x[i] h[(j+n_pre_remove)*down - i*up]
 * (= scipy.signal.resample_poly with the pre-padded filter h; the host builds h and the offsets, frontend.py).
 * src: interleaved [n_in][channels], fmt 0 = int16, 1 = int32 (left-aligned), 2 = float32.  soxr-exactness: unpinned. */
int    mt_resample_poly(const void* src, long long n_in, int channels, int fmt, const float* h, int h_len, int up, int down,
                        long long n_pre_remove, float* out, long long n_out, mt_stream_t stream);
/* The same sum with the filter in polyphase-major order hp[phase][k] = h[phase + k*up], taps_per_phase = ceil(h_len / up) taps per
 * phase (zero-padded): what the host path uses for its designed filter (pass band 0.913 of the lower Nyquist frequency, stop
 * band from that Nyquist frequency, >= 120 dB: ~500 taps per output at 44.1 -> 16 kHz; music-transcription_amd/transcribe.py). */
int    mt_resample_polyphase(const void* src, long long n_in, int channels, int fmt, const float* hp, int taps_per_phase, int up, int down,
                             long long n_pre_remove, float* out, long long n_out, mt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MT_HIP_H */
