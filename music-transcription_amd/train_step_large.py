"""Train-mode forward and backward of CNNRNNModelLarge on the HIP library.

What `loss.backward()` does in the reference's training loop (train/train_transcriber.py:104-131) for
models/cnn_rnn_model.py:262-348 in train mode -- the model the reference's canonical pipeline trains (example.sh:22):
conv1, two ResidualBlocks and the 7x3 freq_aware_conv with BatchNorm BATCH statistics, Dropout2d, the 3-layer main and
1-layer local bidirectional LSTMs, MultiHeadAttention with the +-10 clamp and probability dropout, LayerNorm(x + attn),
shared_fc + dropout + the frame / onset / offset heads -- as a torch.autograd.Function whose forward and backward are
sequences of libmt_hip.so launches (csrc/train_large.hip, train.hip, lstm.hip, lstm_bwd.hip, gemm.hip, convg.hip,
attn.hip).  torch only owns the buffers and the autograd graph edge.

Numerics: as train_step.py -- GEMM / conv operands bf16, accumulation f32, LSTM state f32, BatchNorm statistics
f64-accumulated, dropout masks from a counter-based hash seeded from torch's CPU generator.

Every dense contraction of the backward pass is an NT GEMM or a channels-last conv:
  * conv input gradients  = mt_conv_cl_* with flipped / transposed weights (a residual block's 3x3 + 1x1-skip pair in one call);
  * conv weight gradients = mt_conv_wgrad straight from the channels-last tensors (csrc/conv_wgrad.hip: both MFMA operands through
    transposed LDS reads, the contraction over positions split over workgroups, no planes, no im2col);
  * attention: dPd = dO V^T, dV = Pd^T dO, dQ = dS K, dK = dS^T Q as per-(chunk, head) batched GEMMs around
    mt_attn_clamped_bwd.
"""
from __future__ import annotations

import os

from typing import Dict, List

import torch

from . import _lib
from ._lib import lib, check, ptr
from .train_step import _ru, _st, _gemm, _gather4, BN_EPS, BN_MOMENTUM, device_key

LN_EPS, ATTN_CLIP = 1e-6, 10.0
DROPOUT2D_P = (0.1, 0.1, 0.15)          # cnn_rnn_model.py:188,:192,:202 (hard-coded in the reference)
LOCAL_LAYER_ID = 100                    # dropout / hash stream ids: main layers 0.., local LSTM, attention, heads
ATTN_LAYER_ID, HEADS_LAYER_ID, D2D_LAYER_ID = 200, 300, 400


# ---------------------------------------------------------------------------------------------------------------- packing
class _Packed:
    """The packed operands of one parameter set: destination tensors (persistent: padding is written once, the live parts every step), the job
    tables of pack_plan.PackPlan for the convolution stack (calling stream) and for everything above it (side stream), and the layer-0 W_ih
    launches that keep their own kernel (mt_pack_wih_cf + one transpose)."""

    def __init__(self, model, dev):
        from .pack_plan import PackPlan, one, two
        H, L, Hl, F = model.hidden_size, model.num_layers, model.hidden_size // 2, model.n_mels
        Hp, Hlp, K1 = _ru(H, 16), _ru(Hl, 16), _ru(2 * H, 64)
        F1, F2, F3 = F // 2, F // 4, F // 8
        K0 = F3 * 256
        comb = 2 * H + 2 * Hl
        Cp = _ru(comb, 64)
        dev = _norm_dev(dev)
        f32 = dict(device=dev, dtype=torch.float32)
        bf = dict(device=dev, dtype=torch.bfloat16)
        t: Dict[str, object] = {}
        d = dict(H=H, Hp=Hp, Hl=Hl, Hlp=Hlp, L=L, F=F, F1=F1, F2=F2, F3=F3, K0=K0, K1=K1, comb=comb, Cp=Cp)
        self.t, self.d, self.dev = t, d, dev
        self.key = _param_key(model, dev)

        def P(p):                                   # a parameter as the kernels read it: f32, on the device, contiguous -- no copy
            q = p.detach()
            if not (q.is_cuda and q.device == dev and q.dtype == torch.float32 and q.is_contiguous()):
                raise RuntimeError("train_step_large: parameters must be contiguous f32 tensors on the training device")
            return q

        # ---- convolution stack (calling stream)
        cp = self.conv_plan = PackPlan(dev)
        t["w1"], t["b1"] = P(model.conv1[0].weight).view(32, 9), P(model.conv1[0].bias)
        t["g1"], t["be1"] = P(model.conv1[1].weight), P(model.conv1[1].bias)

        def conv_cl(dst, w, cout, cin, taps):       # [Cout][Cin][KH][KW] -> [Cout][tap*Cin + ci]
            cp.add(dst, w, one(cout, cin * taps), two(taps, cin, cin, 1, taps))

        def conv_dgrad(dst, w, cout, cin, taps, base=0, at=(0, 0), shape=None):
            """[ci (rows padded)][tap'*Cout + co] = w[co][ci][taps - 1 - tap']: the flipped kernel (negative tap stride)"""
            cp.add(dst, w, one(cin, taps), two(taps, cout, cout, -1, cin * taps), base=base + taps - 1, at=at, shape=shape)

        for name, rb, cin, cout in (("rb1", model.res_block1, 32, 64), ("rb2", model.res_block2, 64, 128)):
            w1, w2, ws_ = P(rb.conv1.weight), P(rb.conv2.weight), P(rb.skip[0].weight)
            t[name + "c1_w"], t[name + "c1_b"] = torch.zeros(cout, 9 * cin, **bf), P(rb.conv1.bias)
            conv_cl(t[name + "c1_w"], w1, cout, cin, 9)
            t[name + "c2_w"], t[name + "c2_b"] = torch.zeros(cout, 9 * cout, **bf), P(rb.conv2.bias)
            conv_cl(t[name + "c2_w"], w2, cout, cout, 9)
            t[name + "s_w"], t[name + "s_b"] = torch.zeros(128, 64 if cin < 64 else cin, **bf), P(rb.skip[0].bias)   # the 1x1 skip as a GEMM (K padded to 64)
            cp.add(t[name + "s_w"], ws_, one(cout, cin), one(cin, 1))
            t[name + "c2_wd"] = torch.zeros(cout, 9 * cout, **bf)
            conv_dgrad(t[name + "c2_wd"], w2, cout, cout, 9)
            cin_p = max(cin, 64)                                                                    # conv_cl wants Cout % 64 == 0
            t[name + "c1s_wd"] = torch.zeros(cin_p, 9 * cout + cout, **bf)                          # conv1 dgrad + skip^T in one call
            conv_dgrad(t[name + "c1s_wd"], w1, cout, cin, 9, shape=(cin_p, 9 * cout))
            cp.add(t[name + "c1s_wd"], ws_, one(cin, 1), one(cout, cin), tr=True, at=(0, 9 * cout), shape=(cin_p, cout))
            for bn, tag in ((rb.bn1, "bn1"), (rb.bn2, "bn2"), (rb.skip[1], "bns")):
                t[f"{name}{tag}_g"], t[f"{name}{tag}_b"] = P(bn.weight), P(bn.bias)
        wf = P(model.freq_aware_conv[0].weight)                                                     # [256][128][7][3]
        t["fa_w"], t["fa_b"] = torch.zeros(256, 21 * 128, **bf), P(model.freq_aware_conv[0].bias)
        conv_cl(t["fa_w"], wf, 256, 128, 21)
        t["fa_wdA"], t["fa_wdB"] = torch.zeros(128, 21 * 128, **bf), torch.zeros(128, 21 * 128, **bf)   # input gradient in two halves of the 256 output channels
        conv_dgrad(t["fa_wdA"], wf, 128, 128, 21)
        conv_dgrad(t["fa_wdB"], wf, 128, 128, 21, base=128 * 128 * 21)
        t["fa_g"], t["fa_be"] = P(model.freq_aware_conv[1].weight), P(model.freq_aware_conv[1].bias)
        t["zeros256"] = torch.zeros(256, **f32)
        t["dims"] = d
        cp.finalize()
        self.conv_keys = set(t.keys())

        # ---- LSTMs, attention, heads (side stream)
        up = self.upper_plan = PackPlan(dev)
        # layer 0 of BOTH LSTMs in one 16-bit tensor [main 8 Hp rows; local 8 Hlp rows (+ tile slack)][K0] (reference feature c*F3+f -> kernel column
        # f*256+c: mt_pack_wih_cf), and ONE transpose of the whole = [W_ih_main; W_ih_local]^T for the input-gradient GEMM
        R0, R1 = 8 * Hp, 8 * Hlp
        self.R0, self.R1 = R0, R1
        wboth = self.wboth = torch.zeros(R0 + _ru(R1, 128) + 128, K0, **bf)
        self.wcat = torch.zeros(_ru(K0, 128), R0 + R1, **bf)                    # dX0 = [dG_main | dG_local] . [W_ih_main; W_ih_local]
        t["ml_wihT"] = self.wcat
        self.l0 = []                                                           # (weight, row0, H, Hp) of mt_pack_wih_cf

        def lstm(rnn, layers, Hh, Hhp, row0):
            w_ih, b_g, w_hh = [], [], []
            for l in range(layers):
                K = 2 * Hh
                if l == 0:
                    w_ih.append(wboth if row0 == 0 else wboth[row0:])
                else:
                    w_ih.append(torch.zeros(_ru(8 * Hhp, 128), K1, **bf))
                b_g.append(torch.zeros(2 * 4 * Hhp, **f32))
                w_hh.append(torch.zeros(2, 4 * Hhp, Hhp, **f32))
                for di, suf in enumerate(("", "_reverse")):
                    wi, wh = P(getattr(rnn, f"weight_ih_l{l}{suf}")), P(getattr(rnn, f"weight_hh_l{l}{suf}"))
                    bi, bh = P(getattr(rnn, f"bias_ih_l{l}{suf}")), P(getattr(rnn, f"bias_hh_l{l}{suf}"))
                    if l == 0:
                        self.l0.append((wi, row0 + di * 4 * Hhp, Hh, Hhp))
                    else:                                                      # gate row p*Hhp + j <- p*Hh + j, columns [fwd Hh | reverse Hh] padded to K1
                        up.add(w_ih[l], wi, two(4, Hhp, Hh, Hh * K, K), one(K, 1), at=(di * 4 * Hhp, 0), shape=(4 * Hhp, K1))
                    up.add(b_g[l].view(8, Hhp), bi, one(4, Hh), one(Hh, 1), src2=bh, at=(di * 4, 0), shape=(4, Hhp))
                    up.add(w_hh[l].view(2 * 4 * Hhp, Hhp), wh, two(4, Hhp, Hh, Hh * Hh, Hh), one(Hh, 1), at=(di * 4 * Hhp, 0), shape=(4 * Hhp, Hhp))
            return w_ih, b_g, w_hh

        t["m_wih"], t["m_b"], t["m_whh"] = lstm(model.rnn_main, L, H, Hp, 0)
        t["l_wih"], t["l_b"], t["l_whh"] = lstm(model.rnn_local, 1, Hl, Hlp, R0)
        t["m_wihT"] = [None]
        for l in range(1, L):                                                  # W_ih^T of the deeper layers, straight from the parameters
            wT = torch.zeros(_ru(K1, 128), 8 * Hp, **bf)
            for di, suf in enumerate(("", "_reverse")):
                up.add(wT, P(getattr(model.rnn_main, f"weight_ih_l{l}{suf}")), one(2 * H, 1), two(4, Hp, H, H * 2 * H, 2 * H), tr=True,
                       at=(0, di * 4 * Hp), shape=(wT.shape[0], 4 * Hp))
            t["m_wihT"].append(wT)
        if model.use_attention:
            heads, dh = model.attention.num_heads, model.attention.head_dim
            dp = _ru(dh, 64)
            Ca = heads * dp
            d.update(heads=heads, dh=dh, dp=dp, Ca=Ca, ld3=3 * Ca, scale=float(dh) ** -0.5)
            qw, qb = P(model.attention.qkv.weight), P(model.attention.qkv.bias)            # [3][heads][dh] rows x comb
            t["qkv_w"], t["qkv_b"] = torch.zeros(_ru(3 * Ca, 128), Cp, **bf), torch.zeros(3 * Ca, **f32)
            up.add(t["qkv_w"], qw, two(3 * heads, dp, dh, dh * comb, comb), one(comb, 1))
            up.add(t["qkv_b"].view(3 * heads, dp), qb, one(3 * heads, dh), one(dh, 1))
            t["qkv_wT"] = torch.zeros(_ru(comb, 128), 3 * Ca, **bf)
            up.add(t["qkv_wT"], qw, one(comb, 1), two(3 * heads, dp, dh, dh * comb, comb), tr=True)
            pw = P(model.attention.proj.weight)                                             # [comb] x [heads][dh]
            t["proj_w"], t["proj_b"] = torch.zeros(_ru(comb, 128), Ca, **bf), P(model.attention.proj.bias)
            up.add(t["proj_w"], pw, one(comb, comb), two(heads, dp, dh, dh, 1))
            t["proj_wT"] = torch.zeros(_ru(Ca, 128), Cp, **bf)
            up.add(t["proj_wT"], pw, two(heads, dp, dh, dh, 1), one(comb, comb), tr=True)
            t["ln_g"], t["ln_b"] = P(model.attention_norm.weight), P(model.attention_norm.bias)
        if model.use_onset_offset_heads:
            Hs = _ru(H, 64)
            d.update(Hs=Hs)
            sw = P(model.shared_fc.weight)                                                  # [H][comb]
            t["shared_w"], t["shared_b"] = torch.zeros(_ru(H, 128), Cp, **bf), P(model.shared_fc.bias)
            up.add(t["shared_w"], sw, one(H, comb), one(comb, 1))
            t["shared_wT"] = torch.zeros(_ru(comb, 128), Hs, **bf)
            up.add(t["shared_wT"], sw, one(comb, 1), one(H, comb), tr=True)
            t["heads_w"], t["heads_b"] = torch.zeros(384, Hs, **bf), torch.zeros(264, **f32)
            t["heads_wT"] = torch.zeros(_ru(Hs, 128), 384, **bf)
            for i, m in enumerate((model.frame_head, model.onset_head, model.offset_head)):
                hw, hb = P(m.weight), P(m.bias)                                             # [88][H], [88]
                up.add(t["heads_w"], hw, one(88, H), one(H, 1), at=(88 * i, 0), shape=(88, Hs))
                up.add(t["heads_b"].view(1, 264), hb, one(1, 0), one(88, 1), at=(0, 88 * i), shape=(1, 88))
                up.add(t["heads_wT"], hw, one(H, 1), one(88, H), tr=True, at=(0, 88 * i), shape=(t["heads_wT"].shape[0], 88))
        else:
            fw = P(model.fc.weight)                                                         # [88][comb]
            t["fc_w"], t["fc_b"] = torch.zeros(128, Cp, **bf), P(model.fc.bias)
            up.add(t["fc_w"], fw, one(88, comb), one(comb, 1))
            t["fc_wT"] = torch.zeros(_ru(comb, 128), 128, **bf)
            up.add(t["fc_wT"], fw, one(comb, 1), one(88, comb), tr=True)
        up.finalize()

    def run_upper(self):
        d = self.d
        for w, row0, Hh, Hhp in self.l0:
            check(lib.mt_pack_wih_cf(ptr(w), ptr(self.wboth), d["K0"], row0, Hh, Hhp, 256, d["F3"], _lib.DT_BF16, _st()), "mt_pack_wih_cf")
        check(lib.mt_transpose_bf16(ptr(self.wboth), d["K0"], self.R0 + self.R1, d["K0"], ptr(self.wcat), self.R0 + self.R1, d["K0"], _st()), "mt_transpose_bf16")
        self.upper_plan.run()


def _norm_dev(dev):
    dev = torch.device(dev)
    return torch.device(dev.type, torch.cuda.current_device()) if dev.type == "cuda" and dev.index is None else dev


def _param_key(model, dev):
    return (str(_norm_dev(dev)), bool(model.use_attention), bool(model.use_onset_offset_heads)) + tuple(p.data_ptr() for p in model.parameters())


def pack_train_large(model, dev, side=None) -> Dict[str, object]:
    """bf16 operand layouts of the CURRENT parameters (redone every step: the optimizer moves them) -- two launches of mt_pack_jobs plus layer
    0's W_ih (four mt_pack_wih_cf, one transpose); the tables and the destination tensors are built once per parameter set (round 4: as torch
    expressions this was ~250 small launches per step).  With `side` (a stream) everything above the convolution stack -- 99 % of the bytes --
    is packed on that stream beside the convolution stack's forward; t["_ready"] is the event to wait for before touching those entries.
    The returned tensors are REUSED by the next call: one step per model at a time (as the step workspace)."""
    pk = getattr(model, "_train_packed", None)
    if pk is None or pk.key != _param_key(model, dev):
        pk = model._train_packed = _Packed(model, dev)
    t = pk.t
    pk.conv_plan.run()
    if side is None:
        pk.run_upper()
        t.pop("_ready", None)
    else:
        side.wait_stream(torch.cuda.current_stream(dev))     # the optimizer's update of the parameters is ordered on the calling stream
        with torch.cuda.stream(side):
            pk.run_upper()
            ev = torch.cuda.Event()
            ev.record(side)
        t["_ready"] = ev
    return t


# ---------------------------------------------------------------------------------------------------------------- step workspace
class _StepWorkspace:
    """The training step's zero-padded scratch tensors, allocated (and zeroed) ONCE per (batch, frames): GEMM operands are read in whole
    128-row / 64-column tiles, so their rows and columns beyond the live region must hold zeros -- and no kernel ever writes there, so
    they still do in the next step.  Round 3 called torch.zeros for ~40 such tensors per step (2.3 GB of fills at B = 16, T = 937, most
    of them alone on the GPU in front of the kernel that needed the buffer).  What IS accumulated into (f64 statistic sums, status
    words) stays a per-step torch.zeros.  One step per workspace at a time: a second forward before the first one's backward gets
    fresh tensors (_FreshZeros)."""

    def __init__(self):
        self.t, self.busy = {}, False

    def zeros(self, name, *shape, **kw):
        key = (name, tuple(int(v) for v in shape), kw.get("dtype"))
        t = self.t.get(key)
        if t is None:
            t = self.t[key] = torch.zeros(*shape, **kw)
        return t


    def buffer(self, name, nbytes, device):
        """Persistent UNINITIALISED scratch (the backward recurrences' hand-off workspaces, re-poisoned before every use on a side stream:
        as per-step allocations recorded on two streams they came back to the allocator late, and every now and then a step had to
        hipMalloc a fresh 0.5 GB block -- 80 ms on this stack, the 'one step in a few hundred takes three' of round 3)."""
        key = (name, int(nbytes), "u8")
        t = self.t.get(key)
        if t is None:
            t = self.t[key] = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
        return t


class _FreshZeros:
    busy = False

    def zeros(self, name, *shape, **kw):
        return torch.zeros(*shape, **kw)

    def buffer(self, name, nbytes, device):
        return torch.empty(int(nbytes), device=device, dtype=torch.uint8)


def _workspace(model, B, T, dev):
    if os.environ.get("MT_TRAIN_WS_CACHE", "1") == "0":
        return _FreshZeros()
    pool = model.__dict__.setdefault("_train_ws", {})
    key = (int(B), int(T), str(dev))
    ws = pool.get(key)
    if ws is None:
        for k_ in [k_ for k_, v_ in pool.items() if not v_.busy][:max(0, len(pool) - 1)]:     # keep at most two shapes (ragged batches: T varies)
            del pool[k_]
        ws = pool[key] = _StepWorkspace()
    if ws.busy:
        return _FreshZeros()
    ws.busy = True
    return ws


# ---------------------------------------------------------------------------------------------------------------- small wrappers
def _conv(A, S, W, bias, out, B, F, T, C1, C2, Cout, KH, relu=0, pool=0, out_mode=0, ldx=0, pitchA=None, pitchS=None, accum=0):
    check(lib.mt_conv_cl_ex(ptr(A), pitchA or C1, ptr(S), pitchS or C2, ptr(W), ptr(bias), ptr(out), B, F, T, C1, C2, Cout, KH, relu, pool,
                            out_mode, ldx, accum, _lib.DT_BF16, _st()), "mt_conv_cl_ex")


def _gemm_bf16out(A, lda, W, ldw, bias, C, ldc, M, N, K, relu=0):
    check(lib.mt_gemm_batched_bf16out(ptr(A), lda, 0, 0, ptr(W), ldw, 0, 0, ptr(bias), ptr(C), ldc, 0, 0, M, N, K, 1, 1, relu, _st()),
          "mt_gemm_batched_bf16out")


def _bn_stats(z, N, C, bn, gamma, beta, dev):
    """Batch statistics of a channels-last bf16 tensor [N][C]; updates bn's running statistics.  -> (mean, rstd)"""
    sums = torch.empty(2 * C, device=dev, dtype=torch.float64)
    mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    check(lib.mt_bn_stats_cl(ptr(z), N, C, ptr(sums), _st()), "mt_bn_stats_cl")
    check(lib.mt_bn_finalize(ptr(sums), float(N), ptr(gamma), ptr(beta), ptr(bn.running_mean), ptr(bn.running_var), BN_MOMENTUM, BN_EPS,
                             ptr(mean), ptr(rstd), C, None, None, None, None, 0, _st()), "mt_bn_finalize")
    bn.num_batches_tracked += 1
    return mean, rstd


def _bn_act(za, sa, zb, sb, mask, out, out_mode, ldx, B, F, T, C, relu, pool):
    """sa / sb = (mean, rstd, gamma, beta)"""
    sb = sb or (None, None, None, None)
    check(lib.mt_bn_act_fwd(ptr(za), *(ptr(v) for v in sa), ptr(zb), *(ptr(v) for v in sb), ptr(mask), ptr(out), out_mode, ldx,
                            B, F, T, C, relu, pool, _st()), "mt_bn_act_fwd")


def _bn_act_bwd(dcl, ldd_cl, dx, ldd_x, za, sa, zb, sb, mask, dza, dzb, grads, B, F, T, C, relu, pool, dev, dza_lo=None, dzb_lo=None):
    sb = sb or (None, None, None, None)
    sums = torch.empty(3 * C, device=dev, dtype=torch.float64)
    dga, dba, dgb, dbb = grads
    check(lib.mt_bn_act_bwd(ptr(dcl), ldd_cl, ptr(dx), ldd_x, ptr(za), *(ptr(v) for v in sa), ptr(zb), *(ptr(v) for v in sb), ptr(mask),
                            ptr(sums), ptr(dza), C, ptr(dza_lo), ptr(dzb), C, ptr(dzb_lo), ptr(dga), ptr(dba), ptr(dgb), ptr(dbb), B, F, T, C,
                            relu, pool, _st()), "mt_bn_act_bwd")
    return sums


_SIDE2 = {}


def _side_streams(dev):
    """Two side streams per device: (weight-gradient work of the LSTM stack, the local LSTM's own chain).  The backward
    recurrences are latency-bound on 16-32 CUs: everything that does not gate the next recurrence runs beside them."""
    key = device_key(dev)
    if key not in _SIDE2:
        _SIDE2[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _SIDE2[key]


def autotune_side_streams(step, dev, candidates: int = 4, steps: int = 2):
    """Picks, by measurement, the pair of side streams the training step runs fastest with, and keeps it for the process.
    `step()` runs one full training step (they are real steps: nothing is thrown away).  Why (round 4, profiles/r04_train_large_stream_mapping.txt):
    the HIP runtime folds a process's streams onto GPU_MAX_HW_QUEUES hardware queues, and a side stream that lands on the calling stream's
    queue runs behind it instead of beside it -- 46 ms per CNNRNNModelLarge step (the single-stream time) instead of 42; with 32 queues no two
    streams share one and 42.1 against 43.6 ms by the parity of the streams' pool slots is what is left.  A process that also serves inference
    keeps 8 queues (the headline is 0.5 % faster there), so it chooses its pair by timing the step itself.  (Round 3's larger spread, 49.6 against
    55.7 ms, was mostly something else: a device allocation inside the timed steps -- the allocator's pools are per stream, a new side stream
    starts with an empty one -- which the persistent step workspace removed.)  Returns the seconds per step of every candidate."""
    import time
    dev = torch.device(dev)
    key = device_key(dev)
    seen = []
    for _ in range(max(1, candidates)):
        _forget_side_streams(key)
        step()                                          # (creates the step's side streams; first use of a stream sets its queue up)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(max(1, steps)):
            step()
        torch.cuda.synchronize(dev)
        seen.append(((time.perf_counter() - t0) / max(1, steps), _current_side_streams(key)))
    _restore_side_streams(key, min(seen, key=lambda e: e[0])[1])
    return [t for t, _ in seen]


def _forget_side_streams(key):                          # both training steps' side streams (CNNRNNModel has one, CNNRNNModelLarge two)
    from . import train_step
    _SIDE2.pop(key, None)
    train_step._SIDE.pop(key, None)


def _current_side_streams(key):
    from . import train_step
    return (_SIDE2.get(key), train_step._SIDE.get(key))


def _restore_side_streams(key, saved):
    from . import train_step
    for d_, v_ in ((_SIDE2, saved[0]), (train_step._SIDE, saved[1])):
        if v_ is not None:
            d_[key] = v_


class SideStreamTuner:
    """The same choice made inside a training loop whose steps end in a host synchronisation (train.train_one_epoch): candidate pair c
    serves steps [c (steps + 1), (c + 1)(steps + 1)), the first of them untimed; after the last candidate the fastest pair stays."""

    def __init__(self, dev, candidates: int = 4, steps: int = 2):
        self.dev, self.key = torch.device(dev), device_key(dev)
        self.candidates, self.steps = max(1, candidates), max(1, steps)
        self.i, self.seen, self.acc, self.done = 0, [], 0.0, os.environ.get("MT_TRAIN_STREAM_AUTOTUNE", "1") == "0"

    def step_begin(self):
        if self.done:
            return
        import time
        if self.i % (self.steps + 1) == 0:
            _forget_side_streams(self.key)
            self.acc = 0.0
        self.t0 = time.perf_counter()

    def step_end(self):                                 # call after the step's host synchronisation
        if self.done:
            return
        import time
        if self.i % (self.steps + 1) != 0:
            self.acc += time.perf_counter() - self.t0
        self.i += 1
        if self.i % (self.steps + 1) == 0:
            self.seen.append((self.acc / self.steps, _current_side_streams(self.key)))
            if len(self.seen) == self.candidates:
                _restore_side_streams(self.key, min(self.seen, key=lambda e: e[0])[1])
                self.done = True


def conv_wgrad_direct(dz_hi, dz_lo, dz_pitch, x, x_pitch, B, F, T, Cout, Cin, KH, KW, out):
    """out[Cout][Cin][KH][KW] (f32) = the convolution's weight gradient straight from the channels-last tensors (csrc/conv_wgrad.hip:
    no planes, both MFMA operands through transposed LDS reads, every kernel row of a K range on one XCD).  dz_hi + dz_lo = dz."""
    nws = lib.mt_conv_wgrad_ws_bytes(B, F, T, Cout, Cin, KH, KW)
    ws = torch.empty(max(nws, 16) // 4, device=out.device, dtype=torch.float32)
    check(lib.mt_conv_wgrad(ptr(dz_hi), ptr(dz_lo) if dz_lo is not None else None, dz_pitch, ptr(x), x_pitch, B, F, T, Cout, Cin, KH, KW,
                            ptr(ws), nws, ptr(out), _st()), "mt_conv_wgrad")
    return out


# ---------------------------------------------------------------------------------------------------------------- LSTM stack
def _lstm_forward(X0, K0, w_ih, b_g, w_hh, L, Hp, Hv, K1, B, T, dropout, seed, layer_id0, dev, sync_slots, ws, tag):
    """Train-mode bidirectional LSTM stack on GEMM-row input X0 [Mpad][K0].  Returns saved state; the last layer's hx is
    left for the caller to re-lay out.  Inter-layer dropout p on the outputs of layers 0..L-2."""
    M, Mpad = T * B, _ru(T * B, 128)
    bf = dict(device=dev, dtype=torch.bfloat16)
    Xs, gates, cxs, hxs = [X0], [], [], []
    for l in range(L):
        K = K0 if l == 0 else K1
        gx = torch.empty(lib.mt_lstm_gx_bytes(B, T, Hp) // 4, device=dev)
        cx = torch.empty(lib.mt_lstm_cx_bytes(B, T, Hp) // 4, device=dev)
        hx = torch.empty(lib.mt_lstm_hx_bytes(B, T, Hp) // 4, device=dev)
        sync = sync_slots.pop(0)
        check(lib.mt_gemm_lstm_gx(ptr(Xs[l]), K, ptr(w_ih[l]), K, ptr(b_g[l]), ptr(gx), B, T, Hp, K, _st()), "mt_gemm_lstm_gx")
        check(lib.mt_lstm_bidir_fwd_train(ptr(gx), ptr(w_hh[l]), ptr(hx), ptr(cx), ptr(sync), sync.numel(), B, T, Hp, _st()),
              "mt_lstm_bidir_fwd_train")
        gates.append(gx); cxs.append(cx); hxs.append(hx)
        if l < L - 1:
            Xn = ws.zeros(f"{tag}.Xn{l}", Mpad, K1, **bf)             # (rows >= M and columns >= 2 Hv stay zero: never written)
            check(lib.mt_lstm_relayout_train(ptr(hx), ptr(Xn), K1, B, T, Hp, Hv, float(dropout), seed, layer_id0 + l, _st()),
                  "mt_lstm_relayout_train")
            Xs.append(Xn)
    return dict(Xs=Xs, gates=gates, cxs=cxs, hxs=hxs)


def _lstm_backward(sv, dh, w_hh, w_ihT, L, Hp, Hv, K0, K1, B, T, dropout, seed, layer_id0, dev, sync_slots, dG0, ldg0, col0, names, g, rnn_prefix,
                   k0_gather, wg_stream=None, parts=None, zws=None, tag="", newg=None):
    """BPTT through the stack.  dh: gradient of the top layer's output in the backward recurrence's layout.  Layer 0's gate
    gradients go to dG0[:, col0 : col0 + 8 Hp] (row pitch ldg0): the caller turns them into the input gradient.  Parameter
    gradients land in g under rnn_prefix; k0_gather(gwi, di) produces layer 0's W_ih gradient in the reference layout."""
    M, Mpad = T * B, _ru(T * B, 128)
    bf = dict(device=dev, dtype=torch.bfloat16)
    f32 = dict(device=dev, dtype=torch.float32)
    Hr = _ru(Hp, 128)
    keep = []
    for l in range(L - 1, -1, -1):
        K = K0 if l == 0 else K1
        # parts: partial-product workspaces the caller has already poisoned (1 GB each at H = 512: done on a side stream, early)
        part = parts[l] if parts is not None else torch.empty(lib.mt_lstm_bwd_part_bytes(B, T, Hp), device=dev, dtype=torch.uint8)
        dgx = torch.empty(lib.mt_lstm_dgx_bytes(B, T, Hp), device=dev, dtype=torch.uint8)
        sync = sync_slots.pop(0)
        check(lib.mt_lstm_bidir_bwd_ex(ptr(sv["gates"][l]), ptr(sv["cxs"][l]), ptr(dh), ptr(w_hh[l]), ptr(dgx), ptr(part), part.numel(),
                                       ptr(sync), sync.numel(), B, T, Hp, 1 if parts is not None else 0, _st()), "mt_lstm_bidir_bwd")
        if l == 0:
            dG, ldg = dG0[:, col0:], ldg0
            dGv = dG0.reshape(-1)[col0:]
        else:
            dGfull = zws.zeros(f"{tag}.dG{l}", Mpad, 8 * Hp, **bf)
            dGv, ldg = dGfull, 8 * Hp
        dGT = zws.zeros(f"{tag}.dGT{l}", 4 * Hp + _ru(4 * Hp, 128), Mpad, **bf)
        check(lib.mt_lstm_dg_unpack(ptr(dgx), ptr(dGv), ldg, ptr(dGT), Mpad, B, T, Hp, _st()), "mt_lstm_dg_unpack")
        if l > 0:      # -> dh of layer l-1 (its output went through dropout in the forward pass)
            dh = zws.zeros(f"{tag}.dh{l}", lib.mt_lstm_cx_bytes(B, T, Hp) // 4, **f32)
            check(lib.mt_gemm_lstm_dh(ptr(dGv), 8 * Hp, ptr(w_ihT[l]), 8 * Hp, ptr(dh), B, T, Hp, Hv, 8 * Hp, float(dropout), seed,
                                      layer_id0 + l - 1, _st()), "mt_gemm_lstm_dh")
        # ---- weight gradients: dW_ih = dG^T X_l, dW_hh = dG^T H_prev, db = sum dG.  They gate nothing below: on `wg_stream`
        #      (when given) they run beside the next layer's backward recurrence.  All buffers are allocated here, on the calling
        #      stream, and stay referenced in `keep` until the caller has joined the streams.
        XT = torch.empty(_ru(K, 128) * Mpad, **bf)
        HT = zws.zeros(f"{tag}.HT{l}", 2 * Hr, Mpad, **bf)
        gb, gwi, gwh = torch.empty(8 * Hp, **f32), torch.empty(8 * Hp, K, **f32), torch.empty(2, 4 * Hp, Hp, **f32)
        outs = []
        mk = newg if newg is not None else (lambda name, *shape: torch.empty(*shape, **f32))
        for di, suf in enumerate(("", "_reverse")):        # (the reference-shaped gradients: straight into the flat gradient buffer where allowed)
            wi = mk(f"{rnn_prefix}.weight_ih_l{l}{suf}", 4 * Hv, 256 * (K0 // 256) if l == 0 else 2 * Hv)
            outs.append((wi, mk(f"{rnn_prefix}.weight_hh_l{l}{suf}", 4 * Hv, Hv), mk(f"{rnn_prefix}.bias_ih_l{l}{suf}", 4 * Hv),
                         mk(f"{rnn_prefix}.bias_hh_l{l}{suf}", 4 * Hv)))
        keep += [XT, HT, gb, gwi, gwh, dGT, dGv, outs]
        cur = torch.cuda.current_stream(dev)
        ws = wg_stream if wg_stream is not None else cur
        if wg_stream is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
        with torch.cuda.stream(ws):
            if wg_stream is not None:
                ws.wait_event(ev)
            check(lib.mt_transpose_bf16(ptr(sv["Xs"][l]), K, M, K, ptr(XT), Mpad, K, _st()), "mt_transpose_bf16")
            check(lib.mt_lstm_hprev_t(ptr(sv["hxs"][l]), ptr(HT), Mpad, Hr, B, T, Hp, _st()), "mt_lstm_hprev_t")
            check(lib.mt_rowsum_bf16(ptr(dGT), Mpad, M, ptr(gb), 8 * Hp, _st()), "mt_rowsum_bf16")
            _gemm(dGT, Mpad, XT, Mpad, gwi, K, 8 * Hp, K, Mpad)
            for di in range(2):
                _gemm(dGT[di * 4 * Hp:], Mpad, HT[di * Hr:], Mpad, gwh[di], Hp, 4 * Hp, Hp, Mpad)
            for di, suf in enumerate(("", "_reverse")):
                wi, wh, bb, bb2 = outs[di]
                if l == 0:
                    k0_gather(gwi, di, wi)
                else:
                    _gather4(gwi, di * 4 * Hp * K, wi, (4, Hv, 1, 2 * Hv), (Hp * K, K, 0, 1))
                _gather4(gwh, di * 4 * Hp * Hp, wh, (4, Hv, 1, Hv), (Hp * Hp, Hp, 0, 1))
                _gather4(gb, di * 4 * Hp, bb, (1, 1, 4, Hv), (0, 0, Hp, 1))
                _gather4(gb, di * 4 * Hp, bb2, (1, 1, 4, Hv), (0, 0, Hp, 1))
                g[f"{rnn_prefix}.weight_ih_l{l}{suf}"], g[f"{rnn_prefix}.weight_hh_l{l}{suf}"] = wi, wh
                g[f"{rnn_prefix}.bias_ih_l{l}{suf}"], g[f"{rnn_prefix}.bias_hh_l{l}{suf}"] = bb, bb2
    return keep


# ---------------------------------------------------------------------------------------------------------------- forward
def forward_train_large(model, x: torch.Tensor, p_drop: float, seed: int, p2d=DROPOUT2D_P):
    """Returns (logits [NH][B][88][T] f32 (NH = 3 with the heads, else 1), saved state).  Updates BatchNorm running statistics."""
    dev = x.device
    B, _, F, T = x.shape
    use_side = os.environ.get("MT_TRAIN_LARGE_STREAMS", "1") != "0"
    with torch.cuda.device(dev):
        pk = pack_train_large(model, dev, _side_streams(dev)[0] if use_side else None)
    d = pk["dims"]
    H, Hp, Hl, Hlp, L, F1, F2, F3, K0, K1, comb, Cp = (d[k] for k in ("H", "Hp", "Hl", "Hlp", "L", "F1", "F2", "F3", "K0", "K1", "comb", "Cp"))
    M, Mpad = T * B, _ru(T * B, 128)
    x = x.contiguous().float()
    bf = dict(device=dev, dtype=torch.bfloat16)
    f32 = dict(device=dev, dtype=torch.float32)
    ws = _workspace(model, B, T, dev)
    sv: Dict[str, object] = dict(pk=pk, x=x, B=B, T=T, p=p_drop, seed=seed, p2d=p2d, ws=ws)
    nsync = 2 * (L + 1)
    stride = _ru(max(lib.mt_lstm_sync_bytes(B, Hp), lib.mt_lstm_sync_bytes(B, Hlp)), 256)
    sync_all = torch.zeros(nsync * stride, device=dev, dtype=torch.uint8)       # one status slot per persistent launch of the step
    slots = [sync_all[i * stride:(i + 1) * stride] for i in range(nsync)]
    sv["sync_all"], sv["sync_stride"], sv["sync_free"] = sync_all, stride, slots
    with torch.cuda.device(dev):
        # ---- conv1: statistics of the recomputed pre-BN activation, folded into the inference kernel's weights
        bn1 = model.conv1[1]
        sums = torch.zeros(64, device=dev, dtype=torch.float64)
        mean1, rstd1 = torch.empty(32, **f32), torch.empty(32, **f32)
        wf1, bf1 = torch.empty(32, 9, **f32), torch.empty(32, **f32)
        check(lib.mt_conv1_stats(ptr(x), ptr(pk["w1"]), ptr(pk["b1"]), ptr(sums), B, F, T, _st()), "mt_conv1_stats")
        check(lib.mt_bn_finalize(ptr(sums), float(B * F * T), ptr(pk["g1"]), ptr(pk["be1"]), ptr(bn1.running_mean), ptr(bn1.running_var),
                                 BN_MOMENTUM, BN_EPS, ptr(mean1), ptr(rstd1), 32, ptr(pk["w1"]), ptr(pk["b1"]), ptr(wf1), ptr(bf1), 9, _st()),
              "mt_bn_finalize")
        bn1.num_batches_tracked += 1
        # (rows past the B*F1*T positions: the 1x1 skip GEMM reads whole 128-row tiles, and K = 64 from 32-channel rows)
        a1 = torch.empty(_ru(B * F1 * T, 128) + 1, 32, **bf)
        a1[B * F1 * T:].zero_()
        check(lib.mt_conv1_bn_relu_pool(ptr(x), None, ptr(wf1), ptr(bf1), ptr(a1), B, F, T, _st()), "mt_conv1_bn_relu_pool")
        sv.update(mean1=mean1, rstd1=rstd1, a1=a1)
        # ---- dropout2d mask tables
        masks = []
        for i, (C, p) in enumerate(zip((64, 128, 256), p2d)):
            if p > 0.0:
                m = torch.empty(B, C, **f32)
                check(lib.mt_dropout2d_mask(ptr(m), B, C, float(p), seed, D2D_LAYER_ID + i, _st()), "mt_dropout2d_mask")
                masks.append(m)
            else:
                masks.append(None)
        sv["masks"] = masks
        # ---- residual blocks
        xin, Fin = a1, F1
        for name, rb, cin, cout, pool, mask in (("rb1", model.res_block1, 32, 64, 1, masks[0]), ("rb2", model.res_block2, 64, 128, 0, masks[1])):
            N = B * Fin * T
            z1 = torch.empty(N, cout, **bf)
            _conv(xin, None, pk[name + "c1_w"], pk[name + "c1_b"], z1, B, Fin, T, cin, 0, cout, 3)
            s1 = _bn_stats(z1, N, cout, rb.bn1, pk[name + "bn1_g"], pk[name + "bn1_b"], dev) + (pk[name + "bn1_g"], pk[name + "bn1_b"])
            y1 = torch.empty(N, cout, **bf)
            _bn_act(z1, s1, None, None, None, y1, 0, 0, B, Fin, T, cout, 1, 0)
            z2 = torch.empty(N, cout, **bf)
            _conv(y1, None, pk[name + "c2_w"], pk[name + "c2_b"], z2, B, Fin, T, cout, 0, cout, 3)
            zs = torch.empty(N, cout, **bf)
            Ks = max(cin, 64)
            _gemm_bf16out(xin, cin, pk[name + "s_w"], Ks, pk[name + "s_b"], zs, cout, N, cout, Ks)
            s2 = _bn_stats(z2, N, cout, rb.bn2, pk[name + "bn2_g"], pk[name + "bn2_b"], dev) + (pk[name + "bn2_g"], pk[name + "bn2_b"])
            ss = _bn_stats(zs, N, cout, rb.skip[1], pk[name + "bns_g"], pk[name + "bns_b"], dev) + (pk[name + "bns_g"], pk[name + "bns_b"])
            Fo = Fin // 2 if pool else Fin
            out = torch.empty(_ru(B * Fo * T, 128) + 1, cout, **bf)
            out[B * Fo * T:].zero_()
            _bn_act(z2, s2, zs, ss, mask, out, 0, 0, B, Fin, T, cout, 1, pool)
            sv[name] = dict(xin=xin, Fin=Fin, z1=z1, s1=s1, y1=y1, z2=z2, zs=zs, s2=s2, ss=ss, cin=cin, cout=cout, pool=pool, mask=mask)
            xin, Fin = out, Fo
        r2 = xin
        # ---- freq_aware_conv: 7x3 conv -> BN -> ReLU -> pool -> dropout2d, straight into the GEMM operand rows
        N = B * F2 * T
        zf = torch.empty(N, 256, **bf)
        _conv(r2, None, pk["fa_w"], pk["fa_b"], zf, B, F2, T, 128, 0, 256, 7)
        bnf = model.freq_aware_conv[1]
        sf = _bn_stats(zf, N, 256, bnf, pk["fa_g"], pk["fa_be"], dev) + (pk["fa_g"], pk["fa_be"])
        X0 = torch.empty(Mpad, K0, **bf)
        X0[M:].zero_()
        _bn_act(zf, sf, None, None, masks[2], X0, 1, K0, B, F2, T, 256, 1, 1)
        sv.update(r2=r2, zf=zf, sf=sf)
        # ---- LSTMs
        pm = p_drop if L > 1 else 0.0
        # the local layer (an independent recurrence on the same input) runs on a side stream beside the main stack
        slots_local = [slots.pop()]
        if "_ready" in pk:
            torch.cuda.current_stream(dev).wait_event(pk["_ready"])     # the weights above the convolutions were packed on a side stream
        if use_side:
            main_st, side_b = torch.cuda.current_stream(dev), _side_streams(dev)[1]
            ev_x0 = torch.cuda.Event()
            ev_x0.record(main_st)
            with torch.cuda.stream(side_b):
                side_b.wait_event(ev_x0)
                sv["local"] = _lstm_forward(X0, K0, pk["l_wih"], pk["l_b"], pk["l_whh"], 1, Hlp, Hl, _ru(2 * Hl, 64), B, T, 0.0, seed, LOCAL_LAYER_ID, dev,
                                            slots_local, ws, "local")
                ev_loc = torch.cuda.Event()
                ev_loc.record(side_b)
            for t_ in sv["local"]["gates"] + sv["local"]["cxs"] + sv["local"]["hxs"]:
                t_.record_stream(main_st)              # allocated under the side stream, read by the backward pass on the calling one
        sv["main"] = _lstm_forward(X0, K0, pk["m_wih"], pk["m_b"], pk["m_whh"], L, Hp, H, K1, B, T, pm, seed, 0, dev, slots, ws, "main")
        if use_side:
            main_st.wait_event(ev_loc)
        else:
            sv["local"] = _lstm_forward(X0, K0, pk["l_wih"], pk["l_b"], pk["l_whh"], 1, Hlp, Hl, _ru(2 * Hl, 64), B, T, 0.0, seed, LOCAL_LAYER_ID, dev,
                                        slots_local, ws, "local")
        rb = ws.zeros("rb", Mpad, Cp, **bf)
        r32 = torch.empty(M, comb, **f32)
        check(lib.mt_lstm_relayout_ex(ptr(sv["main"]["hxs"][-1]), ptr(rb), Cp, ptr(r32), comb, 0, B, T, Hp, H, _st()), "mt_lstm_relayout_ex")
        check(lib.mt_lstm_relayout_ex(ptr(sv["local"]["hxs"][-1]), ptr(rb), Cp, ptr(r32), comb, 2 * H, B, T, Hlp, Hl, _st()), "mt_lstm_relayout_ex")
        sv.update(rb=rb, r32=r32)
        feat = rb
        # ---- attention + LayerNorm
        if model.use_attention:
            heads, dp, Ca, ld3, scale = d["heads"], d["dp"], d["Ca"], d["ld3"], d["scale"]
            Tr, Tp = _ru(T, 128), _ru(T, 64)
            dpr = _ru(dp, 128)
            qkv = ws.zeros("qkv", Tr * B, ld3, **bf)
            _gemm_bf16out(rb, Cp, pk["qkv_w"], Cp, pk["qkv_b"], qkv, ld3, M, ld3, Cp)
            S = torch.empty(B * heads, T, Tp, **f32)
            qk = qkv.reshape(-1)
            check(lib.mt_gemm_batched_f32(ptr(qk), B * ld3, ld3, dp, ptr(qk[Ca:]), B * ld3, ld3, dp, None, ptr(S), Tp, heads * T * Tp, T * Tp,
                                          T, T, dp, B * heads, heads, _st()), "mt_gemm_batched_f32 (QK^T)")
            Pd = ws.zeros("Pd", B * heads * T * Tp + Tr * Tp, **bf)           # (a head's GEMM reads whole 128-row tiles: tail slack)
            check(lib.mt_attn_softmax_train(ptr(S), Tp, ptr(Pd), Tp, T, B * heads * T, scale, ATTN_CLIP, float(p_drop), seed, ATTN_LAYER_ID, _st()),
                  "mt_attn_softmax_train")
            VT = torch.empty(B * heads, dpr, Tp, **bf)
            check(lib.mt_attn_transpose_v(ptr(qkv), ld3, 2 * Ca, ptr(VT), B, T, Tp, heads, dp, _st()), "mt_attn_transpose_v")
            ao = ws.zeros("ao", Mpad, Ca, **bf)
            check(lib.mt_gemm_batched_bf16out(ptr(Pd), Tp, heads * T * Tp, T * Tp, ptr(VT), Tp, heads * dpr * Tp, dpr * Tp, None, ptr(ao), B * Ca,
                                              Ca, dp, T, dp, Tp, B * heads, heads, 0, _st()), "mt_gemm_batched_bf16out (PV)")
            proj = torch.empty(M, comb, **f32)
            _gemm(ao, Ca, pk["proj_w"], Ca, proj, comb, M, comb, Ca, bias=pk["proj_b"])
            ln = ws.zeros("ln", Mpad, Cp, **bf)
            stats = torch.empty(M, 2, **f32)
            check(lib.mt_layernorm_residual_train(ptr(r32), comb, ptr(proj), comb, ptr(pk["ln_g"]), ptr(pk["ln_b"]), ptr(ln), Cp, ptr(stats), M, comb,
                                                  LN_EPS, _st()), "mt_layernorm_residual_train")
            sv.update(qkv=qkv, S=S, Pd=Pd, ao=ao, proj=proj, ln=ln, ln_stats=stats)
            feat = ln
        sv["feat"] = feat
        # ---- heads
        ph = 1.5 * p_drop
        if model.use_onset_offset_heads:
            Hs = d["Hs"]
            sh = ws.zeros("sh", Mpad, Hs, **bf)
            _gemm_bf16out(feat, Cp, pk["shared_w"], Cp, pk["shared_b"], sh, Hs, M, H, Cp, relu=1)
            check(lib.mt_dropout_bf16_rows(ptr(sh), Hs, M, H, float(ph), seed, HEADS_LAYER_ID, _st()), "mt_dropout_bf16_rows")
            logits = torch.empty(3, B, 88, T, **f32)
            check(lib.mt_gemm_logits(ptr(sh), Hs, ptr(pk["heads_w"]), Hs, ptr(pk["heads_b"]), ptr(logits), B, T, 264, Hs, _st()), "mt_gemm_logits")
            sv["sh"] = sh
        else:
            logits = torch.empty(1, B, 88, T, **f32)
            check(lib.mt_gemm_logits(ptr(feat), Cp, ptr(pk["fc_w"]), Cp, ptr(pk["fc_b"]), ptr(logits), B, T, 88, Cp, _st()), "mt_gemm_logits")
            check(lib.mt_dropout_f32(ptr(logits), logits.numel(), float(ph), seed, HEADS_LAYER_ID, _st()), "mt_dropout_f32")
    return logits, sv


# ---------------------------------------------------------------------------------------------------------------- backward
def backward_train_large(model, sv, dlogits: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Gradients of every parameter (reference names without the `model.` prefix, reference shapes)."""
    pk = sv["pk"]
    d = pk["dims"]
    H, Hp, Hl, Hlp, L, F, F1, F2, F3, K0, K1, comb, Cp = (d[k] for k in ("H", "Hp", "Hl", "Hlp", "L", "F", "F1", "F2", "F3", "K0", "K1", "comb", "Cp"))
    B, T, x, p_drop, seed = sv["B"], sv["T"], sv["x"], sv["p"], sv["seed"]
    dev = x.device
    M, Mpad = T * B, _ru(T * B, 128)
    bf = dict(device=dev, dtype=torch.bfloat16)
    f32 = dict(device=dev, dtype=torch.float32)
    g: Dict[str, torch.Tensor] = {}
    dlogits = dlogits.contiguous().float()
    slots = sv["sync_free"]
    ws = sv["ws"]
    # Gradient tensors: where the optimizer allows it (optim.FusedAdamClip.grad_target: single GPU, the parameter's .grad still is its
    # view of the flat gradient buffer and nothing was accumulated into it since zero_grad()), the kernel that produces a parameter's
    # gradient writes it STRAIGHT into that view -- autograd is handed None for it -- instead of into a temporary that autograd then
    # adds to the view with one elementwise kernel per parameter (82 of them per step, 1 GB of traffic at 89 M parameters).
    tgt = getattr(model, "_grad_target", None)
    direct = sv["direct_grads"] = set()

    def newg(name, *shape):
        t = tgt(name, shape) if tgt is not None else None
        if t is None:
            return torch.empty(*shape, **f32)
        direct.add(name)
        return t

    def zerog(name, *shape):
        # an analytically zero gradient (a convolution bias in front of a BatchNorm): the flat gradient buffer is zero since zero_grad()
        t = tgt(name, shape) if tgt is not None else None
        if t is None:
            return torch.zeros(*shape, **f32)
        direct.add(name)
        return t
    feat = sv["feat"]
    ph = 1.5 * p_drop
    with torch.cuda.device(dev):
        use_side = os.environ.get("MT_TRAIN_LARGE_STREAMS", "1") != "0"
        main_st = torch.cuda.current_stream(dev)
        side_a, side_b = _side_streams(dev) if use_side else (None, None)
        # the backward recurrences' partial-product workspaces are poisoned (0xFF, 1 GB per layer at H = 512) before they run: on side
        # stream A now, beside the heads' and the attention's backward, instead of in front of every recurrence
        parts_main = parts_local = ev_poison = None
        if use_side:
            parts_main = [ws.buffer(f"part.main{l_}", lib.mt_lstm_bwd_part_bytes(B, T, Hp), dev) for l_ in range(L)]
            parts_local = [ws.buffer("part.local", lib.mt_lstm_bwd_part_bytes(B, T, Hlp), dev)]
            ev0 = torch.cuda.Event()
            ev0.record(main_st)
            with torch.cuda.stream(side_a):
                side_a.wait_event(ev0)
                for p_, h_ in [(q_, Hp) for q_ in parts_main] + [(parts_local[0], Hlp)]:
                    check(lib.mt_lstm_bwd_poison(ptr(p_), p_.numel(), B, T, h_, _st()), "mt_lstm_bwd_poison")
                    p_.record_stream(side_a)
                parts_local[0].record_stream(side_b)
                ev_poison = torch.cuda.Event()
                ev_poison.record(side_a)
        featT = torch.empty(_ru(Cp, 128) * Mpad, **bf)
        check(lib.mt_transpose_bf16(ptr(feat), Cp, M, Cp, ptr(featT), Mpad, Cp, _st()), "mt_transpose_bf16")
        dfeat = torch.empty(M, comb, **f32)
        # ---- heads
        if model.use_onset_offset_heads:
            Hs = d["Hs"]
            dL, dLT = ws.zeros("dL", Mpad, 384, **bf), ws.zeros("dLT", 384, Mpad, **bf)
            check(lib.mt_dlogits_pack_heads(ptr(dlogits), ptr(dL), 384, ptr(dLT), Mpad, 3, B, 88, T, _st()), "mt_dlogits_pack_heads")
            sh = sv["sh"]
            shT = torch.empty(_ru(Hs, 128) * Mpad, **bf)
            check(lib.mt_transpose_bf16(ptr(sh), Hs, M, Hs, ptr(shT), Mpad, Hs, _st()), "mt_transpose_bf16")
            gh = torch.empty(384, Hs, **f32)
            _gemm(dLT, Mpad, shT, Mpad, gh, Hs, 264, Hs, Mpad)
            for i, n in enumerate(("frame", "onset", "offset")):
                w, hb = newg(f"{n}_head.weight", 88, H), newg(f"{n}_head.bias", 88)
                _gather4(gh, i * 88 * Hs, w, (1, 1, 88, H), (0, 0, Hs, 1))
                check(lib.mt_rowsum_bf16(ptr(dLT[i * 88:]), Mpad, M, ptr(hb), 88, _st()), "mt_rowsum_bf16")
                g[f"{n}_head.weight"], g[f"{n}_head.bias"] = w, hb
            dsh = torch.empty(M, Hs, **f32)
            _gemm(dL, 384, pk["heads_wT"], 384, dsh, Hs, M, Hs, 384)
            dzs = ws.zeros("dzs", Mpad, Hs, **bf)
            check(lib.mt_heads_relu_dropout_bwd(ptr(dsh), Hs, ptr(sh), Hs, ptr(dzs), Hs, M, H, float(ph), _st()), "mt_heads_relu_dropout_bwd")
            dzsT = torch.empty(_ru(Hs, 128) * Mpad, **bf)
            check(lib.mt_transpose_bf16(ptr(dzs), Hs, M, Hs, ptr(dzsT), Mpad, Hs, _st()), "mt_transpose_bf16")
            gs = torch.empty(_ru(H, 128), Cp, **f32)
            _gemm(dzsT, Mpad, featT, Mpad, gs, Cp, H, Cp, Mpad)
            g["shared_fc.weight"], g["shared_fc.bias"] = newg("shared_fc.weight", H, comb), newg("shared_fc.bias", H)
            _gather4(gs, 0, g["shared_fc.weight"], (1, 1, H, comb), (0, 0, Cp, 1))
            check(lib.mt_rowsum_bf16(ptr(dzsT), Mpad, M, ptr(g["shared_fc.bias"]), H, _st()), "mt_rowsum_bf16")
            _gemm(dzs, Hs, pk["shared_wT"], Hs, dfeat, comb, M, comb, Hs)
        else:
            if ph > 0.0:        # the dropout on the logits (cnn_rnn_model.py:346): same mask
                dlogits = dlogits.clone()
                check(lib.mt_dropout_f32(ptr(dlogits), dlogits.numel(), float(ph), seed, HEADS_LAYER_ID, _st()), "mt_dropout_f32")
            dL, dLT = ws.zeros("dL", Mpad, 128, **bf), ws.zeros("dLT", 128, Mpad, **bf)
            check(lib.mt_dlogits_pack(ptr(dlogits), ptr(dL), ptr(dLT), Mpad, B, 88, T, _st()), "mt_dlogits_pack")
            gf = torch.empty(128, Cp, **f32)
            _gemm(dLT, Mpad, featT, Mpad, gf, Cp, 88, Cp, Mpad)
            g["fc.weight"], g["fc.bias"] = newg("fc.weight", 88, comb), newg("fc.bias", 88)
            _gather4(gf, 0, g["fc.weight"], (1, 1, 88, comb), (0, 0, Cp, 1))
            check(lib.mt_rowsum_bf16(ptr(dLT), Mpad, M, ptr(g["fc.bias"]), 88, _st()), "mt_rowsum_bf16")
            _gemm(dL, 128, pk["fc_wT"], 128, dfeat, comb, M, comb, 128)
        # ---- LayerNorm + attention
        if model.use_attention:
            heads, dh_, dp, Ca, ld3, scale = d["heads"], d["dh"], d["dp"], d["Ca"], d["ld3"], d["scale"]
            Tr, Tp = _ru(T, 128), _ru(T, 64)
            dpr = _ru(dp, 128)
            nsl = lib.mt_layernorm_residual_bwd_slices()
            dxln = torch.empty(M, comb, **f32)
            part = torch.zeros(nsl, 2, comb, **f32)
            check(lib.mt_layernorm_residual_bwd(ptr(sv["r32"]), comb, ptr(sv["proj"]), comb, ptr(pk["ln_g"]), ptr(sv["ln_stats"]), ptr(dfeat), comb,
                                                ptr(dxln), comb, ptr(part), M, comb, _st()), "mt_layernorm_residual_bwd")
            g["attention_norm.weight"], g["attention_norm.bias"] = newg("attention_norm.weight", comb), newg("attention_norm.bias", comb)
            for i_, k_ in enumerate(("attention_norm.weight", "attention_norm.bias")):
                check(lib.mt_sum_slices_f32(ptr(part[0, i_]), 2 * comb, comb, nsl, ptr(g[k_]), comb, 1, comb, _st()), "mt_sum_slices_f32")
            # proj
            dpb = ws.zeros("dpb", Mpad, Cp, **bf)
            check(lib.mt_f32_to_bf16_rows(ptr(dxln), comb, ptr(dpb), Cp, M, comb, 1.0, _st()), "mt_f32_to_bf16_rows")
            dpT = torch.empty(_ru(Cp, 128) * Mpad, **bf)
            check(lib.mt_transpose_bf16(ptr(dpb), Cp, M, Cp, ptr(dpT), Mpad, Cp, _st()), "mt_transpose_bf16")
            aoT = torch.empty(_ru(Ca, 128) * Mpad, **bf)
            check(lib.mt_transpose_bf16(ptr(sv["ao"]), Ca, M, Ca, ptr(aoT), Mpad, Ca, _st()), "mt_transpose_bf16")
            gp = torch.empty(_ru(comb, 128), Ca, **f32)
            _gemm(dpT, Mpad, aoT, Mpad, gp, Ca, comb, Ca, Mpad)
            g["attention.proj.weight"], g["attention.proj.bias"] = newg("attention.proj.weight", comb, comb), newg("attention.proj.bias", comb)
            _gather4(gp, 0, g["attention.proj.weight"], (1, comb, heads, dh_), (0, Ca, dp, 1))
            check(lib.mt_rowsum_bf16(ptr(dpT), Mpad, M, ptr(g["attention.proj.bias"]), comb, _st()), "mt_rowsum_bf16")
            dO = ws.zeros("dO", Tr * B, Ca, **bf)
            _gemm_bf16out(dpb, Cp, pk["proj_wT"], Cp, None, dO, Ca, M, Ca, Cp)
            # dPd = dO V^T per (chunk, head)
            qkv = sv["qkv"]
            qk = qkv.reshape(-1)
            dOf = dO.reshape(-1)
            dPd = torch.empty(B * heads, T, Tp, **f32)
            check(lib.mt_gemm_batched_f32(ptr(dOf), B * Ca, Ca, dp, ptr(qk[2 * Ca:]), B * ld3, ld3, dp, None, ptr(dPd), Tp, heads * T * Tp, T * Tp,
                                          T, T, dp, B * heads, heads, _st()), "mt_gemm_batched_f32 (dO V^T)")
            dS = ws.zeros("dS", B * heads * T * Tp + Tr * Tp, **bf)
            check(lib.mt_attn_clamped_bwd(ptr(sv["S"]), Tp, ptr(dPd), Tp, ptr(dS), Tp, T, B * heads * T, scale, ATTN_CLIP, float(p_drop), seed,
                                          ATTN_LAYER_ID, _st()), "mt_attn_clamped_bwd")
            dqkv = ws.zeros("dqkv", Mpad, ld3, **bf)
            dq = dqkv.reshape(-1)
            # dV = Pd^T dO:  A = Pd^T [t'][t] per (chunk, head), W = dO^T [d][t]
            PdT = torch.empty(B * heads, Tr, Tp, **bf)
            check(lib.mt_transpose_bf16_batched(ptr(sv["Pd"]), Tp, T * Tp, T, Tp, ptr(PdT), Tp, Tr * Tp, Tr, B * heads, _st()), "mt_transpose_bf16_batched")
            dOT = torch.empty(B * heads, dpr, Tp, **bf)
            check(lib.mt_attn_transpose_v(ptr(dO), Ca, 0, ptr(dOT), B, T, Tp, heads, dp, _st()), "mt_attn_transpose_v")
            check(lib.mt_gemm_batched_bf16out(ptr(PdT), Tp, heads * Tr * Tp, Tr * Tp, ptr(dOT), Tp, heads * dpr * Tp, dpr * Tp, None, ptr(dq[2 * Ca:]),
                                              B * ld3, ld3, dp, T, dp, Tp, B * heads, heads, 0, _st()), "mt_gemm_batched_bf16out (dV)")
            # dQ = dS K:  W = K^T [d][t']
            KT = torch.empty(B * heads, dpr, Tp, **bf)
            check(lib.mt_attn_transpose_v(ptr(qkv), ld3, Ca, ptr(KT), B, T, Tp, heads, dp, _st()), "mt_attn_transpose_v")
            check(lib.mt_gemm_batched_bf16out(ptr(dS), Tp, heads * T * Tp, T * Tp, ptr(KT), Tp, heads * dpr * Tp, dpr * Tp, None, ptr(dq), B * ld3,
                                              ld3, dp, T, dp, Tp, B * heads, heads, 0, _st()), "mt_gemm_batched_bf16out (dQ)")
            # dK = dS^T Q:  A = dS^T [t'][t], W = Q^T [d][t]
            dST = torch.empty(B * heads, Tr, Tp, **bf)
            check(lib.mt_transpose_bf16_batched(ptr(dS), Tp, T * Tp, T, Tp, ptr(dST), Tp, Tr * Tp, Tr, B * heads, _st()), "mt_transpose_bf16_batched")
            QT = torch.empty(B * heads, dpr, Tp, **bf)
            check(lib.mt_attn_transpose_v(ptr(qkv), ld3, 0, ptr(QT), B, T, Tp, heads, dp, _st()), "mt_attn_transpose_v")
            check(lib.mt_gemm_batched_bf16out(ptr(dST), Tp, heads * Tr * Tp, Tr * Tp, ptr(QT), Tp, heads * dpr * Tp, dpr * Tp, None, ptr(dq[Ca:]),
                                              B * ld3, ld3, dp, T, dp, Tp, B * heads, heads, 0, _st()), "mt_gemm_batched_bf16out (dK)")
            # qkv projection
            dqT = torch.empty(_ru(ld3, 128) * Mpad, **bf)
            check(lib.mt_transpose_bf16(ptr(dqkv), ld3, M, ld3, ptr(dqT), Mpad, ld3, _st()), "mt_transpose_bf16")
            rbT = torch.empty(_ru(Cp, 128) * Mpad, **bf)
            check(lib.mt_transpose_bf16(ptr(sv["rb"]), Cp, M, Cp, ptr(rbT), Mpad, Cp, _st()), "mt_transpose_bf16")
            gq = torch.empty(_ru(ld3, 128), Cp, **f32)
            _gemm(dqT, Mpad, rbT, Mpad, gq, Cp, ld3, Cp, Mpad)
            g["attention.qkv.weight"] = newg("attention.qkv.weight", 3 * comb, comb)
            for w3 in range(3):
                _gather4(gq, w3 * Ca * Cp, g["attention.qkv.weight"][w3 * comb:(w3 + 1) * comb], (1, heads, dh_, comb), (0, dp * Cp, Cp, 1))
            gqb = torch.empty(ld3, **f32)
            check(lib.mt_rowsum_bf16(ptr(dqT), Mpad, M, ptr(gqb), ld3, _st()), "mt_rowsum_bf16")
            g["attention.qkv.bias"] = newg("attention.qkv.bias", 3 * comb)
            _gather4(gqb, 0, g["attention.qkv.bias"], (1, 3, heads, dh_), (0, Ca, dp, 1))
            dra = torch.empty(M, comb, **f32)
            _gemm(dqkv, ld3, pk["qkv_wT"], ld3, dra, comb, M, comb, ld3)
            dr = torch.empty(M, comb, **f32)
            check(lib.mt_axpby_rows_f32(ptr(dxln), comb, ptr(dra), comb, ptr(dr), comb, M, comb, 1.0, 1.0, _st()), "mt_axpby_rows_f32")
        else:
            dr = dfeat
        # ---- LSTMs: main stack and the local layer; layer 0's gate gradients side by side -> one input-gradient GEMM
        ldg = 8 * Hp + 8 * Hlp
        dG0 = ws.zeros("dG0", Mpad, ldg, **bf)
        drf = dr.reshape(-1)
        dh_m = ws.zeros("dh_m", lib.mt_lstm_cx_bytes(B, T, Hp) // 4, **f32)
        check(lib.mt_lstm_dh_relayout(ptr(drf), comb, ptr(dh_m), B, T, Hp, H, 0.0, seed, 0, _st()), "mt_lstm_dh_relayout")
        dh_l = ws.zeros("dh_l", lib.mt_lstm_cx_bytes(B, T, Hlp) // 4, **f32)
        check(lib.mt_lstm_dh_relayout(ptr(drf[2 * H:]), comb, ptr(dh_l), B, T, Hlp, Hl, 0.0, seed, 0, _st()), "mt_lstm_dh_relayout")

        def k0_gather(Hx, Hxp):
            def f(gwi, di, wi):
                _gather4(gwi, di * 4 * Hxp * K0, wi, (4, Hx, 256, F3), (Hxp * K0, K0, 1, 256))
            return f
        pm = p_drop if L > 1 else 0.0
        # Stream plan: the main stack's recurrences on the calling stream, its weight-gradient GEMMs on side stream A beside the
        # next layer's recurrence, the local layer's whole chain (an independent recurrence writing its own columns of dG0)
        # on side stream B beside the main stack.  MT_TRAIN_LARGE_STREAMS=0: everything on the calling stream.
        if use_side:
            main_st.wait_event(ev_poison)
        ev_in = torch.cuda.Event()
        ev_in.record(main_st)
        slots_local = [slots.pop()]                    # (the local layer's status slot: taken now, the streams pop independently)
        keep_all = []
        if use_side:
            side_a.wait_event(ev_in)
            with torch.cuda.stream(side_b):
                side_b.wait_event(ev_in)
                keep_all += _lstm_backward(sv["local"], dh_l, pk["l_whh"], [None], 1, Hlp, Hl, K0, _ru(2 * Hl, 64), B, T, 0.0, seed, LOCAL_LAYER_ID, dev,
                                           slots_local, dG0, ldg, 8 * Hp, None, g, "rnn_local", k0_gather(Hl, Hlp), parts=parts_local, zws=ws, tag="local", newg=newg)
        keep_all += _lstm_backward(sv["main"], dh_m, pk["m_whh"], pk["m_wihT"], L, Hp, H, K0, K1, B, T, pm, seed, 0, dev, slots, dG0, ldg, 0, None, g,
                                   "rnn_main", k0_gather(H, Hp), wg_stream=side_a, parts=parts_main, zws=ws, tag="main", newg=newg)
        if not use_side:
            keep_all += _lstm_backward(sv["local"], dh_l, pk["l_whh"], [None], 1, Hlp, Hl, K0, _ru(2 * Hl, 64), B, T, 0.0, seed, LOCAL_LAYER_ID, dev,
                                       slots_local, dG0, ldg, 8 * Hp, None, g, "rnn_local", k0_gather(Hl, Hlp), zws=ws, tag="local", newg=newg)
        else:
            for st_ in (side_a, side_b):                # join: dG0 is complete, every LSTM gradient is final
                evj = torch.cuda.Event()
                evj.record(st_)
                main_st.wait_event(evj)
            for t_ in [v for k_, v in g.items() if k_.startswith("rnn_local.")]:
                t_.record_stream(main_st)              # allocated under side stream B, consumed by autograd on the calling stream
        sv["_keep_lstm_bwd"] = keep_all                # (referenced until the backward pass returns)
        # data parallel: every parameter ABOVE the convolution stack -- both LSTMs, attention, LayerNorm, shared_fc, the heads:
        # 99 % of the 89 M parameters and one contiguous tail of the flat gradient buffer -- is final here, while the whole
        # convolution backward (about half of the step) is still to run: their all-reduce starts now, under it
        # (optim.EarlyBucket; the onset / offset heads' zero gradients under a frame-only loss ride along so that the tail
        # stays contiguous -- the optimizer still skips those parameters, see CnnRnnLargeTrainFn.backward).
        model._early_taken = set()
        if getattr(model, "_grad_sync", None) is not None:
            conv_pfx = ("conv1.", "res_block1.", "res_block2.", "freq_aware_conv.")
            early = {k_: v_ for k_, v_ in g.items() if v_ is not None and not k_.startswith(conv_pfx)}
            for k_ in model._grad_sync.reduce_early(early, main_st):
                keep_all.append(g[k_])
                g[k_] = None
                model._early_taken.add(k_)
        dX0 = torch.empty(M, K0, **f32)
        _gemm(dG0, ldg, pk["ml_wihT"], ldg, dX0, K0, M, K0, ldg)
        # ---- convolution stack.  Weight gradients (csrc/conv_wgrad.hip: matrix-pipe bound, at most one workgroup per CU) go to side
        #      stream A as soon as their dz exists, beside the chain of BatchNorm backward passes (HBM bound) and input-gradient
        #      convolutions on the calling stream; joined at the end.
        # Second bf16 piece of every dz (round 2: dz = hi + lo through the weight-gradient sums)?  Round 4: no.  Rounding dz to ONE bf16 adds
        # zero-mean noise of 2^-9 per term to sums over 10^5 - 10^6 positions: ~1e-4 of a gradient entry by the estimate in DESIGN.md 2, four
        # orders of magnitude under the gradient's own sensitivity to last-bit forward differences (profiles/r04_oracle_noise_floor.txt) -- and
        # it cost 2.5 GB of writes in the BatchNorm backward kernels, the same in reads and HALF of the 5.3 TFLOP of the seven weight-gradient
        # launches.  MT_TRAIN_DZ_LO=1 brings the second piece back (A/B).
        use_lo = os.environ.get("MT_TRAIN_DZ_LO", "0") == "1"
        lo_like = (lambda t_: torch.empty_like(t_)) if use_lo else (lambda t_: None)

        def wgrad(key, dz_hi, dz_lo, dzp, xt, xp, Fx, co, ci, KH, KW):
            if not use_side:
                g[key] = conv_wgrad_direct(dz_hi, dz_lo, dzp, xt, xp, B, Fx, T, co, ci, KH, KW, newg(key, co, ci, KH, KW))
                return
            ev = torch.cuda.Event()
            ev.record(main_st)
            with torch.cuda.stream(side_a):
                side_a.wait_event(ev)
                g[key] = conv_wgrad_direct(dz_hi, dz_lo, dzp, xt, xp, B, Fx, T, co, ci, KH, KW, newg(key, co, ci, KH, KW))
                g[key].record_stream(main_st)
                for t_ in (dz_hi, dz_lo, xt):           # allocated on the calling stream, read here: not to be reused before this is done
                    if t_ is not None:
                        t_.record_stream(side_a)
        # ---- freq_aware_conv
        masks = sv["masks"]
        z256 = pk["zeros256"]
        dzf = torch.empty(B * F2 * T, 256, **bf)
        dzf_lo = lo_like(dzf)
        g["freq_aware_conv.1.weight"], g["freq_aware_conv.1.bias"] = newg("freq_aware_conv.1.weight", 256), newg("freq_aware_conv.1.bias", 256)
        _bn_act_bwd(None, 0, dX0, K0, sv["zf"], sv["sf"], None, None, masks[2], dzf, None,
                    (g["freq_aware_conv.1.weight"], g["freq_aware_conv.1.bias"], None, None), B, F2, T, 256, 1, 1, dev, dza_lo=dzf_lo)
        dr2 = torch.empty(B * F2 * T, 128, **bf)
        _conv(dzf, None, pk["fa_wdA"], z256, dr2, B, F2, T, 128, 0, 128, 7, pitchA=256)
        _conv(dzf.reshape(-1)[128:], None, pk["fa_wdB"], z256, dr2, B, F2, T, 128, 0, 128, 7, pitchA=256, accum=1)
        wgrad("freq_aware_conv.0.weight", dzf, dzf_lo, 256, sv["r2"], 128, F2, 256, 128, 7, 3)
        g["freq_aware_conv.0.bias"] = zerog("freq_aware_conv.0.bias", 256)          # a conv bias in front of a BatchNorm: analytically zero
        # ---- residual blocks, top down
        dout = dr2
        for name, mask_i in (("rb2", 1), ("rb1", 0)):
            st = sv[name]
            cin, cout, pool, Fin, xin = st["cin"], st["cout"], st["pool"], st["Fin"], st["xin"]
            N = B * Fin * T
            dz2, dzs = torch.empty(N, cout, **bf), torch.empty(N, cout, **bf)
            dz2_lo, dzs_lo = lo_like(dz2), lo_like(dzs)
            pfx = "res_block1" if name == "rb1" else "res_block2"
            gr = {k: newg(f"{pfx}.{k}", cout) for k in ("bn2.weight", "bn2.bias", "skip.1.weight", "skip.1.bias", "bn1.weight", "bn1.bias")}
            _bn_act_bwd(dout, cout, None, 0, st["z2"], st["s2"], st["zs"], st["ss"], st["mask"], dz2, dzs,
                        (gr["bn2.weight"], gr["bn2.bias"], gr["skip.1.weight"], gr["skip.1.bias"]), B, Fin, T, cout, 1, pool, dev,
                        dza_lo=dz2_lo, dzb_lo=dzs_lo)
            wgrad(pfx + ".conv2.weight", dz2, dz2_lo, cout, st["y1"], cout, Fin, cout, cout, 3, 3)
            wgrad(pfx + ".skip.0.weight", dzs, dzs_lo, cout, xin, cin, Fin, cout, cin, 1, 1)
            dy1 = torch.empty(N, cout, **bf)
            _conv(dz2, None, pk[name + "c2_wd"], z256, dy1, B, Fin, T, cout, 0, cout, 3)
            dz1 = torch.empty(N, cout, **bf)
            dz1_lo = lo_like(dz1)
            _bn_act_bwd(dy1, cout, None, 0, st["z1"], st["s1"], None, None, None, dz1, None, (gr["bn1.weight"], gr["bn1.bias"], None, None),
                        B, Fin, T, cout, 1, 0, dev, dza_lo=dz1_lo)
            cin_p = max(cin, 64)
            dxin = torch.empty(N, cin_p, **bf)
            _conv(dz1, dzs, pk[name + "c1s_wd"], z256, dxin, B, Fin, T, cout, cout, cin_p, 3)
            wgrad(pfx + ".conv1.weight", dz1, dz1_lo, cout, xin, cin, Fin, cout, cin, 3, 3)
            for k in ("conv1.bias", "conv2.bias", "skip.0.bias"):
                g[f"{pfx}.{k}"] = zerog(f"{pfx}.{k}", cout)
            for k, v in gr.items():
                g[f"{pfx}.{k}"] = v
            dout = dxin
        # ---- conv1 (z1 recomputed from the input); dout = d a1, [B][F1][T][64] with channels 0..31 valid
        g["conv1.0.weight"], g["conv1.0.bias"] = newg("conv1.0.weight", 32, 1, 3, 3), newg("conv1.0.bias", 32)
        g["conv1.1.weight"], g["conv1.1.bias"] = newg("conv1.1.weight", 32), newg("conv1.1.bias", 32)
        sums = torch.zeros(512, device=dev, dtype=torch.float64)
        check(lib.mt_conv1_bwd(ptr(x), ptr(pk["w1"]), ptr(pk["b1"]), ptr(sv["mean1"]), ptr(sv["rstd1"]), ptr(pk["g1"]), ptr(pk["be1"]),
                               ptr(dout), 64, ptr(sums[128:]), ptr(g["conv1.0.weight"]), ptr(g["conv1.0.bias"]), ptr(g["conv1.1.weight"]),
                               ptr(g["conv1.1.bias"]), B, F, T, _st()), "mt_conv1_bwd")
        if use_side:
            evj = torch.cuda.Event()
            evj.record(side_a)
            main_st.wait_event(evj)
    return g


class CnnRnnLargeTrainFn(torch.autograd.Function):
    """logits [NH][B][88][T] = CNNRNNModelLarge(x) in train mode; backward returns the gradient of every parameter."""

    @staticmethod
    def forward(ctx, model, x, p_drop, seed, p2d, names, frame_only, *params):
        logits, sv = forward_train_large(model, x, p_drop, seed, p2d)
        ctx.model, ctx.sv, ctx.names, ctx.frame_only = model, sv, names, frame_only
        model._train_sync = (sv["sync_all"], sv["sync_stride"])
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        g = backward_train_large(ctx.model, ctx.sv, dlogits)
        direct = ctx.sv.get("direct_grads", set())
        ctx.sv["ws"].busy = False                       # the step's workspace may serve the next forward (stream order: the backward
        ctx.sv = None                                   # pass joined its side streams into the calling stream before it returned)
        if ctx.frame_only:
            # model(mel) of the reference's loop returns the frame logits only (cnn_rnn_model.py:343-349,
            # train_transcriber.py:119): the onset / offset heads are not part of the graph, their .grad stays None and
            # torch.optim.Adam never touches them (no weight decay either).  Exact-zero gradients would not be the same
            # thing: Adam's normalisation turns g' = wd * p into a step of ~lr * sign(p).
            for n in ctx.names:
                if n.startswith(("onset_head.", "offset_head.")):
                    g[n] = None
        # (FusedAdamClip reads this: p.grad is a view of its flat buffer there, never None.  Parameters whose gradient the
        #  early all-reduce took over are None here too, but they DO have a gradient -- only the frame-only heads do not.)
        early = getattr(ctx.model, "_early_taken", set())
        no_grad = {n for n in ctx.names if g[n] is None and n not in early}
        if ctx.frame_only:
            no_grad |= {n for n in ctx.names if n.startswith(("onset_head.", "offset_head."))}
        # a parameter is left out of the step only if NO backward pass since zero_grad() produced a gradient for it (two backward
        # passes before one step: a frame-only pass and a return_all_heads pass -> the heads DO have a gradient): intersection
        from .optim import note_params_without_grad
        note_params_without_grad(ctx.model, no_grad)
        for n in direct:                                # already in the flat gradient buffer (see newg in backward_train_large)
            g[n] = None
        return (None, None, None, None, None, None, None) + tuple(g[n] for n in ctx.names)


def train_forward_large(model, x: torch.Tensor, return_all_heads: bool = False):
    names = [n for n, _ in model.named_parameters()]
    params = [p for _, p in model.named_parameters()]
    p = float(getattr(model, "dropout_p", 0.0))
    p2d = tuple(float(v) for v in getattr(model, "dropout2d_p", DROPOUT2D_P))
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (p > 0.0 or any(v > 0.0 for v in p2d)) else 0
    if torch.is_grad_enabled():
        frame_only = bool(model.use_onset_offset_heads and not return_all_heads)
        out = CnnRnnLargeTrainFn.apply(model, x, p, seed, p2d, names, frame_only, *params)
    else:
        out, sv = forward_train_large(model, x, p, seed, p2d)
        sv["ws"].busy = False                           # no backward pass will follow
        model._train_sync = (sv["sync_all"], sv["sync_stride"])
    if model.use_onset_offset_heads and return_all_heads:
        return {"frame": out[0], "onset": out[1], "offset": out[2]}
    return out[0]
