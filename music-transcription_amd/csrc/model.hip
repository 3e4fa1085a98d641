// Whole-model forward of CNNRNNModel (cnn_rnn_model.py:57-74), eval mode, on one stream:
//   mel -> conv1 -> conv2 -> [GEMM (input projection) -> persistent recurrence -> re-layout] x layers
//       -> GEMM (fc, transposed store) -> logits (B, 88, T)
// All intermediates live in the caller's workspace; nothing is allocated or synchronised here.
#include "mt_common.h"
#include <stdlib.h>

extern "C" {
int mt_conv1_bn_relu_pool_dt(const float*, const float*, const float*, const float*, void*, int, int, int, int, mt_stream_t);
int mt_conv2_bn_relu_pool_dt(const void*, const void*, const float*, void*, int, int, int, int, int, mt_stream_t);
int mt_conv12_bn_relu_pool_dt(const float*, const float*, const float*, const float*, const void*, const float*, void*, int, int, int, int, int, mt_stream_t);
int mt_gemm_lstm_gx_dt(const void*, int, const void*, int, const float*, float*, int, int, int, int, int, mt_stream_t);
int mt_gemm_logits_dt(const void*, int, const void*, int, const float*, float*, int, int, int, int, int, mt_stream_t);
int mt_lstm_bidir_fwd_ex(const float*, const float*, float*, void*, size_t, int, int, int, int, mt_stream_t);
int mt_lstm_relayout_dt(const float*, void*, int, float*, int, int, int, int, int, int, int, mt_stream_t);
int mt_gemm_lstm_gx_from_hx_ex(const float*, const void*, int, const float*, float*, int, int, int, int, int, mt_stream_t);
int mt_gemm_logits_from_hx(const float*, const void*, int, const float*, float*, int, int, int, int, mt_stream_t);
int mt_gemm_lstm_gx_sched(const void*, int, const void*, int, const float*, float*, int, int, int, int, int, void*, mt_stream_t);
int mt_gemm_lstm_gx_from_hx_sched(const float*, const void*, int, const float*, float*, int, int, int, int, int, void*, mt_stream_t);
int mt_lstm_bidir_fwd_xproj(const float*, const float*, const float*, const float*, float*, void*, size_t, int, int, int, mt_stream_t);
size_t mt_lstm_gx_bytes(int, int, int);
size_t mt_lstm_hx_bytes(int, int, int);
size_t mt_lstm_sync_bytes(int, int);
}

namespace mt {
struct CnnRnnPlan {
    int F1, Fo2, K0, K1, M, Mpad, Hp;
    size_t act1, x0, x1, gx, hx, hx2, sync, sync_stride, sched, total;
};
static CnnRnnPlan plan(const mt_cnnrnn_weights* w, int B, int T) {
    CnnRnnPlan p;
    p.F1 = w->n_mels / 2; p.Fo2 = p.F1 / 2;
    p.K0 = p.Fo2 * 64;
    p.K1 = (int)align_up((size_t)2 * w->hidden, 64);
    p.Hp = (int)align_up((size_t)w->hidden, 16);        // recurrence layout size (zero-padded hidden units)
    p.M = T * B; p.Mpad = (int)align_up((size_t)p.M, 128);
    size_t o = 0;
    p.act1 = o; o += align_up((size_t)B * p.F1 * T * 32 * 2, 256);
    p.x0 = o;   o += align_up((size_t)p.Mpad * p.K0 * 2, 256);
    p.x1 = o;   o += align_up((size_t)p.Mpad * p.K1 * 2, 256);
    p.gx = o;   o += align_up(mt_lstm_gx_bytes(B, T, p.Hp), 256);
    p.hx = o;   o += align_up(mt_lstm_hx_bytes(B, T, p.Hp), 256);
    p.hx2 = o;  o += align_up(mt_lstm_hx_bytes(B, T, p.Hp), 256);      // ping-pong partner for the fused-projection layers
    p.sync_stride = align_up(mt_lstm_sync_bytes(B, p.Hp), 256);
    p.sync = o; o += p.sync_stride * w->layers;
    p.sched = o; o += align_up((size_t)MT_GEMM_SCHED_BYTES * w->layers, 256);      // tile queues of the persistent projection GEMMs, one block per layer
    p.total = o;
    return p;
}
}  // namespace mt

using namespace mt;

static int check_weights(const mt_cnnrnn_weights* w) {
    MT_REQUIRE(w, MT_EINVAL, "cnnrnn: null weights");
    MT_REQUIRE(w->n_mels >= 4 && w->layers >= 1 && w->layers <= MT_MAX_LSTM_LAYERS, MT_EINVAL,
               "cnnrnn: bad config n_mels=%d layers=%d", w->n_mels, w->layers);
    MT_REQUIRE(w->hidden >= 1 && w->hidden <= 1024, MT_EUNSUPPORTED, "cnnrnn: hidden size %d unsupported (1..1024)", w->hidden);
    MT_REQUIRE_DT(w->operand_dtype, "cnnrnn");
    MT_REQUIRE(w->conv1_w && w->conv1_b && w->conv2_w && w->conv2_b && w->fc_w && w->fc_b, MT_EINVAL, "cnnrnn: null weight pointer");
    for (int l = 0; l < w->layers; ++l)
        MT_REQUIRE(w->w_ih[l] && w->b_gates[l] && w->w_hh[l], MT_EINVAL, "cnnrnn: null LSTM weight pointer (layer %d)", l);
    return MT_OK;
}

extern "C" size_t mt_cnnrnn_workspace_bytes(const mt_cnnrnn_weights* w, int B, int T) {
    if (check_weights(w) != MT_OK || B <= 0 || T <= 0) return 0;
    return plan(w, B, T).total;
}

// Offset of layer l's LSTM sync block inside the workspace (word 0 = hand-off status, 0 = ok).
extern "C" size_t mt_cnnrnn_status_offset(const mt_cnnrnn_weights* w, int B, int T, int layer) {
    if (check_weights(w) != MT_OK || B <= 0 || T <= 0) return 0;
    const CnnRnnPlan p = plan(w, B, T);
    return p.sync + p.sync_stride * layer;
}

// Stage boundaries at which mt_cnnrnn_forward_ex records the caller's events (bench.py times kernels with them).
static inline int rec(void* const* events, int n_events, int& idx, hipStream_t st) {
    if (events && idx < n_events) {
        MT_CHECK_HIP(hipEventRecord((hipEvent_t)events[idx], st));
    }
    ++idx;
    return MT_OK;
}

extern "C" int mt_cnnrnn_num_stages(int layers) { return 3 + 3 * layers; }   // conv1, conv2, (gemm, rec, relayout) x L, fc
// 1: mt_cnnrnn_forward* runs conv1 + conv2 as one kernel (conv12_kernel; the conv1 stage of the event list is then empty).  OPT-IN
// (MT_CONV_FUSED=1 in the environment, read once): same X0 bit for bit and 2.4 GB less HBM traffic per forward of 128 chunks, but not
// faster -- measured 1.38 ms against 0.55 + 0.86 for the two kernels, headline 9 955 - 10 021 against 9 718 - 9 987 chunks/s at 400 steps and
// 9 082 - 9 273 against 9 364 - 9 454 at 20: conv1 is bound by vector-ALU issue (576 multiply-adds per position) and its phase does not
// hide under the other resident workgroup's MFMAs as hoped.
extern "C" int mt_cnnrnn_conv_fused(void) {
    static const int on = getenv("MT_CONV_FUSED") && atoi(getenv("MT_CONV_FUSED")) == 1;
    return on;
}

extern "C" int mt_cnnrnn_forward_ex(const mt_cnnrnn_weights* w, const float* mel, const float* chunk_max_power, int B, int T,
                                    float* logits, void* workspace, size_t workspace_bytes,
                                    void* const* events, int n_events, mt_stream_t stream) {
    int rc = check_weights(w);
    if (rc != MT_OK) return rc;
    MT_REQUIRE(mel && logits && workspace, MT_EINVAL, "mt_cnnrnn_forward: null pointer");
    MT_REQUIRE(B > 0 && T > 0, MT_EINVAL, "mt_cnnrnn_forward: bad dims B=%d T=%d", B, T);
    const CnnRnnPlan p = plan(w, B, T);
    MT_REQUIRE(workspace_bytes >= p.total, MT_EWORKSPACE, "mt_cnnrnn_forward: workspace %zu < %zu bytes", workspace_bytes, p.total);
    char* ws = (char*)workspace;
    const int H = p.Hp, Hv = w->hidden, dt = w->operand_dtype;
    hipStream_t st = (hipStream_t)stream;
    int ei = 0;
    if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;                      // event 0: start
    if (mt_cnnrnn_conv_fused()) {
        // conv1 + conv2 as ONE kernel (csrc/conv.hip, conv12_kernel): act1 never exists in HBM; the conv1 stage is empty
        if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;
        if ((rc = mt_conv12_bn_relu_pool_dt(mel, chunk_max_power, w->conv1_w, w->conv1_b, w->conv2_w, w->conv2_b, ws + p.x0, p.K0, B, w->n_mels, T, dt,
                                            stream)) != MT_OK) return rc;
    } else {
        if ((rc = mt_conv1_bn_relu_pool_dt(mel, chunk_max_power, w->conv1_w, w->conv1_b, ws + p.act1, B, w->n_mels, T, dt, stream)) != MT_OK) return rc;
        if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;
        if ((rc = mt_conv2_bn_relu_pool_dt(ws + p.act1, w->conv2_w, w->conv2_b, ws + p.x0, p.K0, B, p.F1, T, dt, stream)) != MT_OK) return rc;
    }
    if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;
    if (p.K1 != 2 * Hv) MT_CHECK_HIP(hipMemsetAsync(ws + p.x1, 0, (size_t)p.Mpad * p.K1 * 2, (hipStream_t)stream));
    char* hcur = ws + p.hx;                  // the previous layer's output images
    char* hnext = ws + p.hx2;
    // f16 operands: the exchanged h IS the next GEMM's operand type, so the GEMMs that consume a layer's output (the next
    // layer's input projection, the final fc) read their A tiles straight from the hx images -- no re-layout pass, no X1 buffer.
    // (needs un-padded hidden units in whole 64-wide K tiles; other sizes keep the re-layout.)  Layer l then writes hx buffer
    // l & 1, so the layer above reads its predecessor's while writing its own.
    const bool from_hx = dt == MT_DT_F16 && H == Hv && H % 64 == 0;
    // f16 operands + agent-scope recurrence: the gate pre-activations travel GEMM -> recurrence as f16 (MT_GX_F16, include/mt_hip.h)
    const int gx16 = (dt == MT_DT_F16) ? MT_GX_F16 : 0;
    for (int l = 0; l < w->layers; ++l) {
        const bool last = l + 1 == w->layers;
        // layers > 0 with packed W_ihx: input projection fused into the recurrence (reads the previous layer's hx directly)
        const bool fused = l > 0 && w->w_ihx[l] && H <= 512;
        if (fused) {
            if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;              // (no projection GEMM: empty stage)
            if ((rc = mt_lstm_bidir_fwd_xproj((const float*)hcur, w->w_ihx[l], w->b_gates[l], w->w_hh[l], (float*)hnext,
                                              ws + p.sync + p.sync_stride * l, p.sync_stride, B, T, H, stream)) != MT_OK) return rc;
            char* tmp = hcur; hcur = hnext; hnext = tmp;
        } else {
            const int K = l == 0 ? p.K0 : p.K1;
            if (l > 0 && from_hx) {
                if ((rc = mt_gemm_lstm_gx_from_hx_sched((const float*)hcur, w->w_ih[l], K, w->b_gates[l], (float*)(ws + p.gx), B, T, H, H, gx16,
                                                        ws + p.sched + (size_t)MT_GEMM_SCHED_BYTES * l, stream)) != MT_OK) return rc;
                char* tmp = hcur; hcur = hnext; hnext = tmp;                       // this layer writes the other buffer
            } else {
                const void* X = l == 0 ? ws + p.x0 : ws + p.x1;
                if ((rc = mt_gemm_lstm_gx_sched(X, K, w->w_ih[l], K, w->b_gates[l], (float*)(ws + p.gx), B, T, H, K, dt | gx16,
                                                ws + p.sched + (size_t)MT_GEMM_SCHED_BYTES * l, stream)) != MT_OK) return rc;
            }
            if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;
            if ((rc = mt_lstm_bidir_fwd_ex((const float*)(ws + p.gx), w->w_hh[l], (float*)hcur, ws + p.sync + p.sync_stride * l,
                                           p.sync_stride, B, T, H, gx16, stream)) != MT_OK) return rc;
        }
        if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;
        // the next consumer of feature ROWS: a GEMM-projected layer, or the final fc (none when they read hx directly)
        const bool next_fused = !last && w->w_ihx[l + 1] && H <= 512;
        if (!next_fused && !from_hx)
            if ((rc = mt_lstm_relayout_dt((const float*)hcur, ws + p.x1, p.K1, nullptr, 0, 0, B, T, H, Hv, dt, stream)) != MT_OK) return rc;
        if ((rc = rec(events, n_events, ei, st)) != MT_OK) return rc;
    }
    if (from_hx) {
        if ((rc = mt_gemm_logits_from_hx((const float*)hcur, w->fc_w, p.K1, w->fc_b, logits, B, T, MT_N_PITCH, H, stream)) != MT_OK) return rc;
    } else {
        if ((rc = mt_gemm_logits_dt(ws + p.x1, p.K1, w->fc_w, p.K1, w->fc_b, logits, B, T, MT_N_PITCH, p.K1, dt, stream)) != MT_OK) return rc;
    }
    return rec(events, n_events, ei, st);
}

extern "C" int mt_cnnrnn_forward(const mt_cnnrnn_weights* w, const float* mel, const float* chunk_max_power, int B, int T,
                                 float* logits, void* workspace, size_t workspace_bytes, mt_stream_t stream) {
    return mt_cnnrnn_forward_ex(w, mel, chunk_max_power, B, T, logits, workspace, workspace_bytes, nullptr, 0, stream);
}
