// Whole-model forward of CNNRNNModelLarge (models/cnn_rnn_model.py:262-348), eval mode, on one stream:
//   mel -> conv1 -> res_block1 -> pool -> res_block2 -> 7x3 conv + pool
//       -> bi-LSTM local (1 layer, H/2) and bi-LSTM main (L layers, H), concatenated [main | local]
//       -> LayerNorm(x + MHA(x)) -> ReLU(shared_fc) -> frame/onset/offset heads -> (3, B, 88, T)
// Hidden sizes are laid out padded to a multiple of 16 (zero weights: a padded unit's gates are 0, so its
// c and h stay 0); padded units are dropped when the LSTM output is written as feature rows.
#include "mt_common.h"

extern "C" {
int mt_conv1_bn_relu_pool_dt(const float*, const float*, const float*, const float*, void*, int, int, int, int, mt_stream_t);
int mt_conv_cl_dt(const void*, const void*, const void*, const float*, void*, int, int, int, int, int, int, int, int, int, int, int, int, mt_stream_t);
int mt_gemm_lstm_gx_dt(const void*, int, const void*, int, const float*, float*, int, int, int, int, int, mt_stream_t);
int mt_gemm_logits_dt(const void*, int, const void*, int, const float*, float*, int, int, int, int, int, mt_stream_t);
int mt_gemm_batched_f32_dt(const void*, int, long long, long long, const void*, int, long long, long long, const float*, float*, int,
                           long long, long long, int, int, int, int, int, int, mt_stream_t);
int mt_gemm_batched_h16out_dt(const void*, int, long long, long long, const void*, int, long long, long long, const float*, void*, int,
                              long long, long long, int, int, int, int, int, int, int, mt_stream_t);
int mt_lstm_bidir_fwd_ex(const float*, const float*, float*, void*, size_t, int, int, int, int, mt_stream_t);
int mt_gemm_lstm_gx_from_hx_ex(const float*, const void*, int, const float*, float*, int, int, int, int, int, mt_stream_t);
int mt_lstm_relayout_dt(const float*, void*, int, float*, int, int, int, int, int, int, int, mt_stream_t);
int mt_gemm_lstm_gx_sched(const void*, int, const void*, int, const float*, float*, int, int, int, int, int, void*, mt_stream_t);
int mt_gemm_lstm_gx_from_hx_sched(const float*, const void*, int, const float*, float*, int, int, int, int, int, void*, mt_stream_t);
int mt_lstm_bidir_fwd_xproj(const float*, const float*, const float*, const float*, float*, void*, size_t, int, int, int, mt_stream_t);
int mt_attn_softmax_clamped_dt(const float*, int, void*, int, int, long long, float, float, int, mt_stream_t);
int mt_attn_transpose_v(const void*, int, int, void*, int, int, int, int, int, mt_stream_t);
int mt_attn_fused_clamped(const void*, int, int, const void*, int, int, int, int, int, float, float, void*, int, int, mt_stream_t);
int mt_layernorm_residual_dt(const float*, int, const float*, int, const float*, const float*, void*, int, long long, int, float, int, mt_stream_t);
size_t mt_lstm_gx_bytes(int, int, int);
size_t mt_lstm_hx_bytes(int, int, int);
size_t mt_lstm_sync_bytes(int, int);
}

namespace mt {
struct LargePlan {
    int F1, F2, F3, K0, K1, M, Mpad, Tr, Tp, Hp, Hlp, comb, Cp, Hs, dp, ld3, Ca;
    size_t act1, r1a, r1, r2a, r2, x0, x1, gx, hx, gx2, hx2, hx3, sync, sync_stride, sched, rb, r32, qkv, S, P, VT, ao, proj, ln, sh, total;
};
static LargePlan lplan(const mt_cnnrnn_large_weights* w, int B, int T) {
    LargePlan p;
    p.F1 = w->n_mels / 2; p.F2 = p.F1 / 2; p.F3 = p.F2 / 2;
    p.K0 = p.F3 * 256;
    p.Hp = (int)align_up((size_t)w->hidden, 16); p.Hlp = (int)align_up((size_t)w->hidden_local, 16);
    p.K1 = (int)align_up((size_t)2 * w->hidden, 64);
    p.comb = 2 * w->hidden + 2 * w->hidden_local;
    p.Cp = (int)align_up((size_t)p.comb, 64);
    p.Hs = (int)align_up((size_t)w->hidden, 64);
    p.dp = w->head_dim_pad; p.Ca = w->heads * p.dp; p.ld3 = 3 * p.Ca;
    p.M = T * B; p.Mpad = (int)align_up((size_t)p.M, 128);
    p.Tr = (int)align_up((size_t)T, 128); p.Tp = (int)align_up((size_t)T, 64);
    const int Hmax = p.Hp > p.Hlp ? p.Hp : p.Hlp;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes, 256); return at; };
    p.act1 = take((size_t)B * p.F1 * T * 32 * 2);
    p.r1a = take((size_t)B * p.F1 * T * 64 * 2);
    p.r1 = take((size_t)B * p.F2 * T * 64 * 2);
    p.r2a = take((size_t)B * p.F2 * T * 128 * 2);
    p.r2 = take((size_t)B * p.F2 * T * 128 * 2);
    p.x0 = take((size_t)p.Mpad * p.K0 * 2);
    p.x1 = take((size_t)p.Mpad * p.K1 * 2);
    p.gx = take(mt_lstm_gx_bytes(B, T, Hmax));
    p.hx = take(mt_lstm_hx_bytes(B, T, Hmax));
    p.gx2 = take(mt_lstm_gx_bytes(B, T, p.Hlp));          // the local LSTM's own buffers: it may run beside the main stack
    p.hx2 = take(mt_lstm_hx_bytes(B, T, p.Hlp));
    p.hx3 = take(mt_lstm_hx_bytes(B, T, Hmax));            // ping-pong partner of hx for layers with the fused input projection
    p.sync_stride = align_up(mt_lstm_sync_bytes(B, Hmax), 256);
    p.sync = take(p.sync_stride * (w->layers + 1));
    p.sched = take((size_t)MT_GEMM_SCHED_BYTES * (w->layers + 1));     // tile queues of the persistent projection GEMMs (local LSTM + one per layer)
    p.rb = take((size_t)p.Mpad * p.Cp * 2);
    p.r32 = take((size_t)p.M * p.comb * 4);
    p.qkv = take((size_t)p.Tr * B * p.ld3 * 2);
    // fused attention core (csrc/attn_fused.hip; head sizes 64 / 128 / 192): the scores are never written -- no S, no P
    const bool fused_attn = p.dp == 64 || p.dp == 128 || p.dp == 192;
    p.S = take(fused_attn ? 0 : (size_t)B * w->heads * T * p.Tp * 4);
    p.P = take(fused_attn ? 0 : (size_t)B * w->heads * p.Tr * p.Tp * 2);
    p.VT = take((size_t)B * w->heads * align_up((size_t)p.dp, 128) * p.Tp * 2);
    p.ao = take((size_t)p.Mpad * p.Ca * 2);
    p.proj = take((size_t)p.M * p.comb * 4);
    p.ln = take((size_t)p.Mpad * p.Cp * 2);
    p.sh = take((size_t)p.Mpad * p.Hs * 2);
    p.total = o;
    return p;
}
}  // namespace mt

using namespace mt;

static int check_large(const mt_cnnrnn_large_weights* w) {
    MT_REQUIRE(w, MT_EINVAL, "cnnrnn_large: null weights");
    MT_REQUIRE(w->n_mels >= 8 && w->layers >= 1 && w->layers <= MT_MAX_LSTM_LAYERS && w->hidden >= 1 && w->hidden <= 1024 &&
               w->hidden_local >= 1 && w->hidden_local <= 1024, MT_EUNSUPPORTED, "cnnrnn_large: unsupported config");
    MT_REQUIRE(!w->use_attention || (w->heads > 0 && w->head_dim_pad > 0 && w->head_dim_pad % 64 == 0), MT_EINVAL, "cnnrnn_large: bad attention dims");
    MT_REQUIRE_DT(w->operand_dtype, "cnnrnn_large");
    MT_REQUIRE(2 * w->hidden + 2 * w->hidden_local <= 2048, MT_EUNSUPPORTED, "cnnrnn_large: combined feature size > 2048");
    return MT_OK;
}

extern "C" size_t mt_cnnrnn_large_workspace_bytes(const mt_cnnrnn_large_weights* w, int B, int T) {
    if (check_large(w) != MT_OK || B <= 0 || T <= 0) return 0;
    return lplan(w, B, T).total;
}
extern "C" size_t mt_cnnrnn_large_status_offset(const mt_cnnrnn_large_weights* w, int B, int T, int idx) {
    if (check_large(w) != MT_OK || B <= 0 || T <= 0) return 0;
    const LargePlan p = lplan(w, B, T);
    return p.sync + p.sync_stride * idx;       // idx 0 = local LSTM, 1.. = main layers
}

#define RUN(expr) do { int rc_ = (expr); if (rc_ != MT_OK) return rc_; } while (0)

// logits3: [3][B][88][T] f32 (frame, onset, offset) when use_heads, else [1][B][88][T].
// side_stream / ev_fork / ev_join (all three, or all NULL): the local LSTM branch (projection, recurrence, re-layout) is
// issued on side_stream between the two caller-owned events, beside the main LSTM stack -- both are latency-bound and
// together use 192 of the 256 CUs.  Without them everything runs on `stream` in order.
// events (optional, benchmarks): the caller's hipEvent_t handles, recorded on `stream` at the stage boundaries
// (mt_cnnrnn_large_num_stages): start | conv1 | res_block1 | res_block2 | freq_aware_conv | local LSTM: projection, recurrence,
// re-layout | main LSTM layer l: projection, recurrence, re-layout | attention + LayerNorm | heads.
#define REC() do { if (events && ei < n_events) MT_CHECK_HIP(hipEventRecord((hipEvent_t)events[ei], st)); ++ei; } while (0)
static int large_forward_impl(const mt_cnnrnn_large_weights* w, const float* mel, const float* chunk_max_power, int B, int T,
                              float* logits3, void* workspace, size_t workspace_bytes, mt_stream_t stream,
                              mt_stream_t side_stream, void* ev_fork, void* ev_join, void* const* events, int n_events) {
    int ei = 0;
    RUN(check_large(w));
    MT_REQUIRE(mel && logits3 && workspace && B > 0 && T > 0, MT_EINVAL, "mt_cnnrnn_large_forward: bad arguments");
    const LargePlan p = lplan(w, B, T);
    MT_REQUIRE(workspace_bytes >= p.total, MT_EWORKSPACE, "mt_cnnrnn_large_forward: workspace %zu < %zu bytes", workspace_bytes, p.total);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int Hv = w->hidden, Hl = w->hidden_local, dt = w->operand_dtype;
    // ---- CNN
    REC();
    RUN(mt_conv1_bn_relu_pool_dt(mel, chunk_max_power, w->conv1_w, w->conv1_b, ws + p.act1, B, w->n_mels, T, dt, stream));
    REC();
    RUN(mt_conv_cl_dt(ws + p.act1, nullptr, w->rb1c1_w, w->rb1c1_b, ws + p.r1a, B, p.F1, T, 32, 0, 64, 3, 1, 0, 0, 0, dt, stream));
    RUN(mt_conv_cl_dt(ws + p.r1a, ws + p.act1, w->rb1c2_w, w->rb1c2_b, ws + p.r1, B, p.F1, T, 64, 32, 64, 3, 1, 1, 0, 0, dt, stream));
    REC();
    RUN(mt_conv_cl_dt(ws + p.r1, nullptr, w->rb2c1_w, w->rb2c1_b, ws + p.r2a, B, p.F2, T, 64, 0, 128, 3, 1, 0, 0, 0, dt, stream));
    RUN(mt_conv_cl_dt(ws + p.r2a, ws + p.r1, w->rb2c2_w, w->rb2c2_b, ws + p.r2, B, p.F2, T, 128, 64, 128, 3, 1, 0, 0, 0, dt, stream));
    REC();
    RUN(mt_conv_cl_dt(ws + p.r2, nullptr, w->fa_w, w->fa_b, ws + p.x0, B, p.F2, T, 128, 0, 256, 7, 1, 1, 1, p.K0, dt, stream));
    REC();
    // ---- concatenated feature rows: bf16 GEMM operand (zero pad columns) + fp32 copy for the residual
    if (p.Cp != p.comb) MT_CHECK_HIP(hipMemsetAsync(ws + p.rb, 0, (size_t)p.Mpad * p.Cp * 2, st));
    if (p.K1 != 2 * Hv) MT_CHECK_HIP(hipMemsetAsync(ws + p.x1, 0, (size_t)p.Mpad * p.K1 * 2, st));
    // local LSTM (1 layer) -> columns [2H, 2H + 2Hl); on the side stream when one is given
    const bool fork = side_stream && ev_fork && ev_join;
    mt_stream_t ls = fork ? side_stream : stream;
    if (fork) {
        MT_CHECK_HIP(hipEventRecord((hipEvent_t)ev_fork, st));
        MT_CHECK_HIP(hipStreamWaitEvent((hipStream_t)side_stream, (hipEvent_t)ev_fork, 0));
    }
    // f16 operands + agent-scope recurrence: the gate pre-activations travel GEMM -> recurrence as f16 (MT_GX_F16, include/mt_hip.h)
    const int gx16 = (dt == MT_DT_F16) ? MT_GX_F16 : 0;
    RUN(mt_gemm_lstm_gx_sched(ws + p.x0, p.K0, w->local_w_ih, p.K0, w->local_b, (float*)(ws + p.gx2), B, T, p.Hlp, p.K0, dt | gx16, ws + p.sched, ls));
    REC();
    RUN(mt_lstm_bidir_fwd_ex((const float*)(ws + p.gx2), w->local_w_hh, (float*)(ws + p.hx2), ws + p.sync, p.sync_stride, B, T, p.Hlp, gx16, ls));
    REC();
    RUN(mt_lstm_relayout_dt((const float*)(ws + p.hx2), ws + p.rb, p.Cp, (float*)(ws + p.r32), p.comb, 2 * Hv, B, T, p.Hlp, Hl, dt, ls));
    REC();
    if (fork) MT_CHECK_HIP(hipEventRecord((hipEvent_t)ev_join, (hipStream_t)side_stream));
    // main LSTM; layers > 0 with a packed W_ihx take their input projection inside the recurrence (no GEMM, no re-layout)
    char* hcur = ws + p.hx;
    char* hnext = ws + p.hx3;
    // f16 operands: main layers l > 0 read their GEMM A tiles straight from the previous layer's hx images (see model.hip)
    const bool from_hx = dt == MT_DT_F16 && p.Hp == Hv && Hv % 64 == 0;
    for (int l = 0; l < w->layers; ++l) {
        const bool last = l + 1 == w->layers;
        const bool fused = l > 0 && w->main_w_ihx[l] && p.Hp <= 512;
        if (fused) {
            REC();                                               // (no projection GEMM: empty stage)
            RUN(mt_lstm_bidir_fwd_xproj((const float*)hcur, w->main_w_ihx[l], w->main_b[l], w->main_w_hh[l], (float*)hnext,
                                        ws + p.sync + p.sync_stride * (l + 1), p.sync_stride, B, T, p.Hp, stream));
            char* tmp = hcur; hcur = hnext; hnext = tmp;
        } else {
            const void* X = l == 0 ? ws + p.x0 : ws + p.x1;
            const int K = l == 0 ? p.K0 : p.K1;
            if (l > 0 && from_hx) {
                RUN(mt_gemm_lstm_gx_from_hx_sched((const float*)hcur, w->main_w_ih[l], K, w->main_b[l], (float*)(ws + p.gx), B, T, p.Hp, Hv, gx16,
                                                  ws + p.sched + (size_t)MT_GEMM_SCHED_BYTES * (l + 1), stream));
                char* tmp = hcur; hcur = hnext; hnext = tmp;
            } else {
                RUN(mt_gemm_lstm_gx_sched(X, K, w->main_w_ih[l], K, w->main_b[l], (float*)(ws + p.gx), B, T, p.Hp, K, dt | gx16,
                                          ws + p.sched + (size_t)MT_GEMM_SCHED_BYTES * (l + 1), stream));
            }
            REC();
            RUN(mt_lstm_bidir_fwd_ex((const float*)(ws + p.gx), w->main_w_hh[l], (float*)hcur, ws + p.sync + p.sync_stride * (l + 1),
                                  p.sync_stride, B, T, p.Hp, gx16, stream));
        }
        REC();
        const bool next_fused = !last && w->main_w_ihx[l + 1] && p.Hp <= 512;
        if (last) RUN(mt_lstm_relayout_dt((const float*)hcur, ws + p.rb, p.Cp, (float*)(ws + p.r32), p.comb, 0, B, T, p.Hp, Hv, dt, stream));
        else if (!next_fused && !from_hx) RUN(mt_lstm_relayout_dt((const float*)hcur, ws + p.x1, p.K1, nullptr, 0, 0, B, T, p.Hp, Hv, dt, stream));
        REC();
    }
    if (fork) MT_CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)ev_join, 0));      // both column ranges of rb / r32 are complete
    const void* feat = ws + p.rb;            // [Mpad][Cp] bf16
    if (w->use_attention) {
        const int heads = w->heads, dp = p.dp;
        // qkv projection (rows up to Tr*B are readable as padding of the per-head GEMMs below)
        RUN(mt_gemm_batched_h16out_dt(ws + p.rb, p.Cp, 0, 0, w->qkv_w, p.Cp, 0, 0, w->qkv_b, ws + p.qkv, p.ld3, 0, 0, p.M, p.ld3, p.Cp, 1, 1, 0, dt, stream));
        const bf16_t* qkv = (const bf16_t*)(ws + p.qkv);
        RUN(mt_attn_transpose_v(qkv, p.ld3, 2 * p.Ca, ws + p.VT, B, T, p.Tp, heads, dp, stream));
        if (dp == 64 || dp == 128 || dp == 192) {
            // QK^T -> scale -> clamp +-10 -> exp -> P V in ONE kernel per (chunk, head, 256 queries); S and P never reach memory
            RUN(mt_attn_fused_clamped(qkv, p.ld3, p.Ca, ws + p.VT, p.Tp, B, T, heads, dp, w->attn_scale, 10.0f, ws + p.ao, p.Ca, dt, stream));
        } else {
            // S[b][head] = Q K^T : A rows t -> qkv row t*B+b (lda = B*ld3), batch z = b*heads + head
            RUN(mt_gemm_batched_f32_dt(qkv, B * p.ld3, p.ld3, dp, qkv + p.Ca, B * p.ld3, p.ld3, dp, nullptr, (float*)(ws + p.S), p.Tp,
                                    (long long)heads * T * p.Tp, (long long)T * p.Tp, T, T, dp, B * heads, heads, dt, stream));
            RUN(mt_attn_softmax_clamped_dt((const float*)(ws + p.S), p.Tp, ws + p.P, p.Tp, T, (long long)B * heads * T, w->attn_scale, 10.0f, dt, stream));
            // P is [B*heads][T][Tp]; the PV GEMM may read A rows up to roundup(T,128) of a head, i.e. into the next head's
            // rows (or the buffer's tail, sized for it): those rows only feed masked outputs.
            RUN(mt_gemm_batched_h16out_dt(ws + p.P, p.Tp, (long long)heads * T * p.Tp, (long long)T * p.Tp, ws + p.VT, p.Tp,
                                        (long long)heads * align_up((size_t)dp, 128) * p.Tp, (long long)align_up((size_t)dp, 128) * p.Tp, nullptr,
                                        ws + p.ao, B * p.Ca, p.Ca, dp, T, dp, p.Tp, B * heads, heads, 0, dt, stream));
        }
        RUN(mt_gemm_batched_f32_dt(ws + p.ao, p.Ca, 0, 0, w->proj_w, p.Ca, 0, 0, w->proj_b, (float*)(ws + p.proj), p.comb, 0, 0, p.M, p.comb, p.Ca, 1, 1, dt, stream));
        if (p.Cp != p.comb) MT_CHECK_HIP(hipMemsetAsync(ws + p.ln, 0, (size_t)p.Mpad * p.Cp * 2, st));
        RUN(mt_layernorm_residual_dt((const float*)(ws + p.r32), p.comb, (const float*)(ws + p.proj), p.comb, w->ln_g, w->ln_b, ws + p.ln, p.Cp,
                                  p.M, p.comb, 1e-6f, dt, stream));
        feat = ws + p.ln;
    }
    REC();
    if (w->use_heads) {
        if (p.Hs != Hv) MT_CHECK_HIP(hipMemsetAsync(ws + p.sh, 0, (size_t)p.Mpad * p.Hs * 2, st));
        RUN(mt_gemm_batched_h16out_dt(feat, p.Cp, 0, 0, w->shared_w, p.Cp, 0, 0, w->shared_b, ws + p.sh, p.Hs, 0, 0, p.M, Hv, p.Cp, 1, 1, 1, dt, stream));
        RUN(mt_gemm_logits_dt(ws + p.sh, p.Hs, w->heads_w, p.Hs, w->heads_b, logits3, B, T, 3 * MT_N_PITCH, p.Hs, dt, stream));
    } else {
        RUN(mt_gemm_logits_dt(feat, p.Cp, w->fc_w, p.Cp, w->fc_b, logits3, B, T, MT_N_PITCH, p.Cp, dt, stream));
    }
    REC();
    return MT_OK;
}
#undef REC

extern "C" int mt_cnnrnn_large_forward_ex(const mt_cnnrnn_large_weights* w, const float* mel, const float* chunk_max_power, int B, int T,
                                          float* logits3, void* workspace, size_t workspace_bytes, mt_stream_t stream,
                                          mt_stream_t side_stream, void* ev_fork, void* ev_join) {
    return large_forward_impl(w, mel, chunk_max_power, B, T, logits3, workspace, workspace_bytes, stream, side_stream, ev_fork, ev_join, nullptr, 0);
}

// Everything on `stream`, recording the caller's events at the stage boundaries (benchmarks: per-kernel times on the launch stream).
extern "C" int mt_cnnrnn_large_num_stages(int layers) { return 9 + 3 * layers; }
extern "C" int mt_cnnrnn_large_forward_ev(const mt_cnnrnn_large_weights* w, const float* mel, const float* chunk_max_power, int B, int T,
                                          float* logits3, void* workspace, size_t workspace_bytes, void* const* events, int n_events,
                                          mt_stream_t stream) {
    return large_forward_impl(w, mel, chunk_max_power, B, T, logits3, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, events, n_events);
}

extern "C" int mt_cnnrnn_large_forward(const mt_cnnrnn_large_weights* w, const float* mel, const float* chunk_max_power, int B, int T,
                                       float* logits3, void* workspace, size_t workspace_bytes, mt_stream_t stream) {
    return mt_cnnrnn_large_forward_ex(w, mel, chunk_max_power, B, T, logits3, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr);
}
