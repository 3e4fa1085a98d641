"""Train-mode forward and backward of CNNRNNModel on the HIP library.

What `loss.backward()` does in the reference's training loop (train/train_transcriber.py:104-131) for
models/cnn_rnn_model.py:57-74 in train mode -- BatchNorm2d with batch statistics (running statistics updated),
ReLU, MaxPool2d((2,1)), the 3-layer bidirectional nn.LSTM with inter-layer dropout, nn.Linear -- as a
torch.autograd.Function whose forward and backward are sequences of libmt_hip.so launches (csrc/train.hip,
lstm_bwd.hip, lstm.hip, gemm.hip, convg.hip, conv.hip).  torch only owns the buffers and the autograd graph edge.

Numerics: GEMM/conv operands bf16, accumulation f32, LSTM state/gates f32 (the reference: fp16 autocast around
convs/fc, fp32 LSTM), BatchNorm statistics f64-accumulated.  Dropout masks come from a counter-based hash seeded
from torch's CPU generator (reproducible under torch.manual_seed; not bit-identical to cuDNN's Philox stream).

One training step may be in flight per model: scratch buffers are cached per (B, T) and reused by the next call.
"""
from __future__ import annotations

from typing import Dict, List

import os

import torch

from . import _lib
from ._lib import lib, check, ptr

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _ru(v: int, a: int) -> int:
    return (v + a - 1) // a * a


def _st():
    return _lib.stream_ptr()


class _PackedSmall:
    """The packed operands of one parameter set of CNNRNNModel: persistent destination tensors and the two job tables of pack_plan.PackPlan
    (convolutions: what the step needs first; LSTM + fc: packed on the side stream beside the convolutions).  Layer 0's W_ih keeps its own
    kernel (mt_pack_wih_cf: the feature re-ordering) and its transpose."""

    def __init__(self, model, dev):
        from .pack_plan import PackPlan, one, two
        dev = _norm_dev(dev)
        H, L, F = model.hidden_size, model.num_layers, model.n_mels
        Hp, K1, Fo2 = _ru(H, 16), _ru(2 * H, 64), (F // 2) // 2
        K0 = Fo2 * 64
        f32 = dict(device=dev, dtype=torch.float32)
        bf = dict(device=dev, dtype=torch.bfloat16)
        self.key = _param_key(model, dev)
        self.conv: Dict[str, object] = {}
        self.rnn: Dict[str, object] = {}
        self.dims = dict(H=H, Hp=Hp, L=L, F=F, F1=F // 2, Fo2=Fo2, K0=K0, K1=K1)

        def P(p):                                   # a parameter as the kernels read it: f32, on the device, contiguous -- no copy
            q = p.detach()
            if not (q.is_cuda and q.device == dev and q.dtype == torch.float32 and q.is_contiguous()):
                raise RuntimeError("train_step: parameters must be contiguous f32 tensors on the training device")
            return q

        t, cp = self.conv, PackPlan(dev)
        c1, bn1, c2, bn2 = model.cnn[0], model.cnn[1], model.cnn[4], model.cnn[5]
        t["w1"], t["b1"] = P(c1.weight).view(32, 9), P(c1.bias)
        w2 = P(c2.weight)                                                                            # [64][32][3][3]
        t["w2"], t["b2"] = torch.zeros(64, 288, **bf), P(c2.bias)                                    # [co][tap*32 + ci]
        cp.add(t["w2"], w2, one(64, 288), two(9, 32, 32, 1, 9))
        t["w2d"] = torch.zeros(64, 576, **bf)                                                        # dgrad: [ci (pad 64)][tap'*64 + co] = w2[co][ci][8 - tap']
        cp.add(t["w2d"], w2, one(32, 9), two(9, 64, 64, -1, 288), base=8)
        t["zero64"] = torch.zeros(64, **f32)
        for i, bn in ((1, bn1), (2, bn2)):
            t[f"g{i}"], t[f"be{i}"] = P(bn.weight), P(bn.bias)
        self.conv_plan = cp.finalize()

        t, up = self.rnn, PackPlan(dev)
        self.K0, self.Hp, self.H = K0, Hp, H
        t["w_ih"], t["b_g"], t["w_hh"], t["w_ihT"] = [], [], [], []
        self.l0 = []
        for l in range(L):
            K, Kp = (K0, K0) if l == 0 else (2 * H, K1)
            t["w_ih"].append(torch.zeros(_ru(8 * Hp, 128), Kp, **bf))
            t["b_g"].append(torch.zeros(2 * 4 * Hp, **f32))
            t["w_hh"].append(torch.zeros(2, 4 * Hp, Hp, **f32))
            t["w_ihT"].append(torch.zeros(_ru(Kp, 128), 8 * Hp, **bf))
            for di, suf in enumerate(("", "_reverse")):
                wi, wh = P(getattr(model.rnn, f"weight_ih_l{l}{suf}")), P(getattr(model.rnn, f"weight_hh_l{l}{suf}"))
                bi, bh = P(getattr(model.rnn, f"bias_ih_l{l}{suf}")), P(getattr(model.rnn, f"bias_hh_l{l}{suf}"))
                if l == 0:                                                                           # reference column c*Fo2+f -> kernel column f*64+c
                    self.l0.append((wi, di * 4 * Hp))
                else:
                    up.add(t["w_ih"][l], wi, two(4, Hp, H, H * K, K), one(K, 1), at=(di * 4 * Hp, 0), shape=(4 * Hp, Kp))
                    up.add(t["w_ihT"][l], wi, one(K, 1), two(4, Hp, H, H * K, K), tr=True, at=(0, di * 4 * Hp), shape=(t["w_ihT"][l].shape[0], 4 * Hp))
                up.add(t["b_g"][l].view(8, Hp), bi, one(4, H), one(H, 1), src2=bh, at=(di * 4, 0), shape=(4, Hp))
                up.add(t["w_hh"][l].view(2 * 4 * Hp, Hp), wh, two(4, Hp, H, H * H, H), one(H, 1), at=(di * 4 * Hp, 0), shape=(4 * Hp, Hp))
        fw = P(model.fc.weight)                                                                      # [88][2H]
        t["fc_w"], t["fc_b"] = torch.zeros(128, K1, **bf), P(model.fc.bias)
        up.add(t["fc_w"], fw, one(88, 2 * H), one(2 * H, 1))
        t["fc_wT"] = torch.zeros(_ru(K1, 128), 128, **bf)
        up.add(t["fc_wT"], fw, one(2 * H, 1), one(88, 2 * H), tr=True)
        self.rnn_plan = up.finalize()

    def run(self, part):
        if part in ("all", "conv"):
            self.conv_plan.run()
        if part in ("all", "rnn"):
            w0, wT0 = self.rnn["w_ih"][0], self.rnn["w_ihT"][0]
            for w, row0 in self.l0:
                check(lib.mt_pack_wih_cf(ptr(w), ptr(w0), self.K0, row0, self.H, self.Hp, 64, self.K0 // 64, _lib.DT_BF16, _st()), "mt_pack_wih_cf")
            check(lib.mt_transpose_bf16(ptr(w0), self.K0, 8 * self.Hp, self.K0, ptr(wT0), 8 * self.Hp, self.K0, _st()), "mt_transpose_bf16")
            self.rnn_plan.run()


def _norm_dev(dev):
    dev = torch.device(dev)
    return torch.device(dev.type, torch.cuda.current_device()) if dev.type == "cuda" and dev.index is None else dev


def _param_key(model, dev):
    return (str(_norm_dev(dev)),) + tuple(p.data_ptr() for p in model.parameters())


def pack_train(model, dev, part: str = "all") -> Dict[str, object]:
    """Device-side operand layouts of the CURRENT parameters (re-done every step: the optimizer moves them) on the current stream: one
    mt_pack_jobs launch per part (round 4: ~60 torch launches before), layer 0's W_ih by mt_pack_wih_cf + one transpose.
    part = "conv" (what the step needs first), "rnn" (LSTM + fc, packed on the side stream beside the convolutions) or "all".
    The returned tensors are REUSED by the next call (one step per model at a time)."""
    pk = getattr(model, "_train_packed", None)
    if pk is None or pk.key != _param_key(model, dev):
        pk = model._train_packed = _PackedSmall(model, dev)
    pk.run(part)
    t: Dict[str, object] = {"dims": pk.dims}
    if part in ("all", "conv"):
        t.update(pk.conv)
    if part in ("all", "rnn"):
        t.update(pk.rnn)
    return t


class _Prof:
    """Optional per-launch timing for benchmarks: when `model._profile` is a list, `with prof("name"):` brackets the launches
    inside with two timing events on the current stream and appends (name, start, end).  Free otherwise."""

    def __init__(self, model):
        self.rec = getattr(model, "_profile", None)

    def __call__(self, name):
        return _ProfSpan(self.rec, name)


class _ProfSpan:
    def __init__(self, rec, name):
        self.rec, self.name = rec, name

    def __enter__(self):
        if self.rec is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.rec is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.rec.append((self.name, self.a, b))
        return False


_SIDE = {}


def device_key(dev) -> str:
    """One spelling per device for the side-stream tables: 'cuda', torch.device('cuda') and 'cuda:0' are the same GPU (a tuner built
    with the caller's device='cuda' must find the streams the step created under x.device = cuda:0)."""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return f"{dev.type}:{idx}"


def _side_stream(dev):
    key = device_key(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _split_k(M, N, K):
    """K split of a weight-gradient GEMM: its output (M x N = a parameter) is small, its contraction (K = T B positions) long, so one
    tile per workgroup fills a fraction of the chip (dW_hh at H = 512: 16 tiles of 256 x 256 for 256 CUs, each walking 236 K-tiles).
    -> number of K slices (1 = no split): the largest of 8, 4, 2 that divides K into whole 64-wide K-tiles and keeps the launch under
    ~2 workgroups per CU.  MT_GEMM_SPLITK=0 turns it off."""
    if K < 4096 or K % 64 or os.environ.get("MT_GEMM_SPLITK", "1") == "0":
        return 1
    big = M >= 1024 and N >= 512 and N % 128 == 0                      # (gemm.hip launch_dt: the shapes that take the 256-tile kernel)
    tiles = (-(-M // 256)) * (-(-N // 256)) if big else (-(-M // 128)) * (-(-N // 128))
    for S in (8, 4, 2):
        if (K // 64) % S == 0 and tiles * S <= 512 and tiles * (S // 2) < 256:
            return S
    return 1


def _gemm(A, lda, W, ldw, C, ldc, M, N, K, bias=None):
    # the kernel reads whole 128-row tiles of both operands: the views handed in must cover them
    assert A.numel() >= (_ru(M, 128) - 1) * lda + K and W.numel() >= (_ru(N, 128) - 1) * ldw + K, "GEMM operand smaller than its tiles"
    assert C.numel() >= (M - 1) * ldc + N
    S = _split_k(M, N, K) if bias is None else 1
    if S > 1:       # K slices as a batch into partial products, summed in a fixed order (bitwise reproducible)
        Kc = K // S
        part = torch.empty(S * M * N, device=C.device, dtype=torch.float32)
        check(lib.mt_gemm_batched_f32(ptr(A), lda, 0, Kc, ptr(W), ldw, 0, Kc, None, ptr(part), N, 0, M * N, M, N, Kc, S, S, _st()), "mt_gemm_batched_f32 (split K)")
        check(lib.mt_sum_slices_f32(ptr(part), M * N, N, S, ptr(C), ldc, M, N, _st()), "mt_sum_slices_f32")
        return
    check(lib.mt_gemm_bf16_f32acc(ptr(A), lda, ptr(W), ldw, ptr(bias), ptr(C), ldc, M, N, K, _st()), "mt_gemm_bf16_f32acc")


def _gather4(src, src_off, dst, n, s, alpha=1.0):
    """dst (contiguous, prod(n) elements) = alpha * src.flatten()[src_off + sum i_k s_k]"""
    base = src.reshape(-1)[src_off:]
    check(lib.mt_gather4_f32(ptr(base), ptr(dst), n[0], n[1], n[2], n[3], s[0], s[1], s[2], s[3], float(alpha), _st()), "mt_gather4_f32")


def forward_train(model, x: torch.Tensor, dropout: float, seed: int):
    """Returns (logits (B,88,T) f32, saved state for backward_train).  Updates the BatchNorm running statistics."""
    dev = x.device
    B, _, F, T = x.shape
    prof = _Prof(model)
    # The LSTM / fc operands are packed on the side stream while the main stream runs the convolutions, and the first
    # BPTT workspace gets its poison fill there as well (the side stream first waits for everything already queued on
    # the main stream: the previous step's optimizer update, and the last users of whatever the allocator hands back).
    with torch.cuda.device(dev):
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        ev_start = torch.cuda.Event()
        ev_start.record(main)
        pk = pack_train(model, dev, "conv")
        d = pk["dims"]
        Hp_, L_ = d["Hp"], d["L"]
        # (persistent per (B, T): as per-step allocations used on two streams they came back to the allocator late, and a step that has to
        #  hipMalloc a fresh 0.5 GB block waits ~80 ms for it -- see train_step_large._StepWorkspace)
        pool = model.__dict__.setdefault("_bptt_parts", {})
        pkey = (int(B), int(T), Hp_, str(dev))
        if pkey not in pool:
            if len(pool) >= 2:
                pool.pop(next(iter(pool)))
            pool[pkey] = [torch.empty(lib.mt_lstm_bwd_part_bytes(B, T, Hp_), device=dev, dtype=torch.uint8) for _ in range(min(2, L_))]
        parts = pool[pkey]
    H, Hp, L, F1, K0, K1 = d["H"], d["Hp"], d["L"], d["F1"], d["K0"], d["K1"]
    M, Mpad = T * B, _ru(T * B, 128)
    x = x.contiguous().float()
    bf = dict(device=dev, dtype=torch.bfloat16)
    f32 = dict(device=dev, dtype=torch.float32)
    sums = torch.zeros(512, device=dev, dtype=torch.float64)
    sv: Dict[str, object] = {"pk": pk, "x": x, "B": B, "T": T, "dropout": dropout, "seed": seed}
    bn1, bn2 = model.cnn[1], model.cnn[5]
    with torch.cuda.device(dev):
        # ---- conv1: batch statistics of the recomputed pre-BN activation, folded into the inference kernel's weights
        mean1, rstd1 = torch.empty(32, **f32), torch.empty(32, **f32)
        wf1, bf1 = torch.empty(32, 9, **f32), torch.empty(32, **f32)
        check(lib.mt_conv1_stats(ptr(x), ptr(pk["w1"]), ptr(pk["b1"]), ptr(sums), B, F, T, _st()), "mt_conv1_stats")
        check(lib.mt_bn_finalize(ptr(sums), float(B * F * T), ptr(pk["g1"]), ptr(pk["be1"]), ptr(bn1.running_mean), ptr(bn1.running_var),
                                 BN_MOMENTUM, BN_EPS, ptr(mean1), ptr(rstd1), 32, ptr(pk["w1"]), ptr(pk["b1"]), ptr(wf1), ptr(bf1), 9, _st()),
              "mt_bn_finalize")
        a1 = torch.empty(B, F1, T, 32, **bf)
        check(lib.mt_conv1_bn_relu_pool(ptr(x), None, ptr(wf1), ptr(bf1), ptr(a1), B, F, T, _st()), "mt_conv1_bn_relu_pool")
        # ---- conv2: raw bf16 output -> statistics -> BN + ReLU + pool straight into the GEMM operand
        z2 = torch.empty(B, F1, T, 64, **bf)
        # (+ the order of each pooled row pair's f32 results before their bf16 rounding: the backward pass routes with it)
        tie2 = torch.empty(B * (F1 // 2) * T * 2 * 2, device=dev, dtype=torch.int32)
        check(lib.mt_conv_cl_tie(ptr(a1), ptr(pk["w2"]), ptr(pk["b2"]), ptr(z2), ptr(tie2), B, F1, T, 32, 64, 3, _st()), "mt_conv_cl_tie")
        sv["tie2"] = tie2
        sums2 = sums[128:]
        mean2, rstd2 = torch.empty(64, **f32), torch.empty(64, **f32)
        check(lib.mt_bn_stats_cl(ptr(z2), B * F1 * T, 64, ptr(sums2), _st()), "mt_bn_stats_cl")
        check(lib.mt_bn_finalize(ptr(sums2), float(B * F1 * T), ptr(pk["g2"]), ptr(pk["be2"]), ptr(bn2.running_mean), ptr(bn2.running_var),
                                 BN_MOMENTUM, BN_EPS, ptr(mean2), ptr(rstd2), 64, None, None, None, None, 0, _st()), "mt_bn_finalize")
        X0 = torch.empty(Mpad, K0, **bf)               # every column of the M valid rows is written below; pad rows only feed
        X0[M:].zero_()                                 # discarded GEMM outputs but must be finite for the transposed (K = rows) use
        check(lib.mt_bn_relu_pool_apply(ptr(z2), ptr(mean2), ptr(rstd2), ptr(pk["g2"]), ptr(pk["be2"]), ptr(X0), K0, B, F1, T, _st()),
              "mt_bn_relu_pool_apply")
        # (issued only now, so that the host has the convolutions queued before it spends its time on the packing launches)
        with torch.cuda.stream(side):
            side.wait_event(ev_start)
            pk.update(pack_train(model, dev, "rnn"))
            ev_pack = torch.cuda.Event()
            ev_pack.record(side)
            check(lib.mt_lstm_bwd_poison(ptr(parts[0]), parts[0].numel(), B, T, Hp_, _st()), "mt_lstm_bwd_poison")
            ev_part0 = torch.cuda.Event()
            ev_part0.record(side)
        sv.update(mean1=mean1, rstd1=rstd1, a1=a1, z2=z2, mean2=mean2, rstd2=rstd2, parts=parts, ev_part0=ev_part0)
        main.wait_event(ev_pack)                       # LSTM / fc operands are packed
        # ---- LSTM layers
        Xs: List[torch.Tensor] = [X0]
        gates, cxs, hxs = [], [], []
        # one status slot per persistent launch of the step (L forward + L backward recurrences): a launch zeroes its own
        # slot only, so the host can read every launch's hand-off status after the step (model.raise_on_train_handoff_timeout)
        sstride = _ru(lib.mt_lstm_sync_bytes(B, Hp), 256)
        sync_all = torch.zeros(2 * L * sstride, device=dev, dtype=torch.uint8)
        model._train_sync = (sync_all, sstride)
        sv["sync_all"], sv["sync_stride"] = sync_all, sstride
        for l in range(L):
            sync = sync_all[l * sstride:(l + 1) * sstride]
            K = K0 if l == 0 else K1
            gx = torch.empty(lib.mt_lstm_gx_bytes(B, T, Hp) // 4, **f32)
            cx = torch.empty(lib.mt_lstm_cx_bytes(B, T, Hp) // 4, **f32)
            hx = torch.empty(lib.mt_lstm_hx_bytes(B, T, Hp) // 4, **f32)
            with prof(f"gemm_lstm_gx_l{l}"):
                check(lib.mt_gemm_lstm_gx(ptr(Xs[l]), K, ptr(pk["w_ih"][l]), K, ptr(pk["b_g"][l]), ptr(gx), B, T, Hp, K, _st()), "mt_gemm_lstm_gx")
            with prof(f"lstm_rec_train_l{l}"):
                check(lib.mt_lstm_bidir_fwd_train(ptr(gx), ptr(pk["w_hh"][l]), ptr(hx), ptr(cx), ptr(sync), sync.numel(), B, T, Hp, _st()),
                      "mt_lstm_bidir_fwd_train")
            if K1 == 2 * H:
                Xn = torch.empty(Mpad, K1, **bf)
                Xn[M:].zero_()
            else:
                Xn = torch.zeros(Mpad, K1, **bf)       # padding columns stay zero
            p = dropout if l < L - 1 else 0.0
            check(lib.mt_lstm_relayout_train(ptr(hx), ptr(Xn), K1, B, T, Hp, H, float(p), seed, l, _st()), "mt_lstm_relayout_train")
            gates.append(gx); cxs.append(cx); hxs.append(hx); Xs.append(Xn)
        logits = torch.empty(B, 88, T, **f32)
        check(lib.mt_gemm_logits(ptr(Xs[L]), K1, ptr(pk["fc_w"]), K1, ptr(pk["fc_b"]), ptr(logits), B, T, 88, K1, _st()), "mt_gemm_logits")
        # the side stream's poison fill of parts[0] finishes under the recurrences above: joining here is free, and it
        # keeps the allocator's stream ordering sound when backward_train never runs (no_grad, a skipped batch)
        main.wait_event(ev_part0)
    sv.update(Xs=Xs, gates=gates, cxs=cxs, hxs=hxs)
    for bn in (bn1, bn2):
        bn.num_batches_tracked += 1
    return logits, sv


def backward_train(model, sv, dlogits: torch.Tensor, debug: dict = None) -> Dict[str, torch.Tensor]:
    """Gradients of every parameter (reference names without the `model.` prefix, reference shapes)."""
    pk = sv["pk"]
    prof = _Prof(model)
    d = pk["dims"]
    H, Hp, L, F, F1, Fo2, K0, K1 = d["H"], d["Hp"], d["L"], d["F"], d["F1"], d["Fo2"], d["K0"], d["K1"]
    B, T, x = sv["B"], sv["T"], sv["x"]
    dev = x.device
    M, Mpad = T * B, _ru(T * B, 128)
    bf = dict(device=dev, dtype=torch.bfloat16)
    f32 = dict(device=dev, dtype=torch.float32)
    g: Dict[str, torch.Tensor] = {}
    dlogits = dlogits.contiguous().float()
    Xs = sv["Xs"]
    # single GPU: the LSTM / fc weight gradients (99.9 % of the 36 M values) are written straight into the parameters' views of the flat gradient
    # buffer (optim.FusedAdamClip.make_grad_target; see train_step_large.backward_train_large) instead of into temporaries autograd then adds
    tgt = getattr(model, "_grad_target", None)
    direct = sv["direct_grads"] = set()

    def newg(name, *shape):
        t = tgt(name, shape) if tgt is not None else None
        if t is None:
            return torch.empty(*shape, **f32)
        direct.add(name)
        return t
    with torch.cuda.device(dev):
        # ---- fc: dW = dL^T X_L, db = sum dL, dX_L = dL W
        dL, dLT = torch.zeros(Mpad, 128, **bf), torch.zeros(128, Mpad, **bf)
        check(lib.mt_dlogits_pack(ptr(dlogits), ptr(dL), ptr(dLT), Mpad, B, 88, T, _st()), "mt_dlogits_pack")
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        XT = torch.empty(_ru(K1, 128) * Mpad, **bf)
        gfc = torch.empty(128, K1, **f32)
        g["fc.weight"], g["fc.bias"] = newg("fc.weight", 88, 2 * H), newg("fc.bias", 88)
        keep = [dL, dLT, XT, gfc]                    # side-stream operands stay referenced until the streams have joined
        ev0 = torch.cuda.Event()
        ev0.record(main)

        def enqueue_fc_wgrad():                      # weight gradients never gate the layers below: side stream
            with torch.cuda.stream(side):
                side.wait_event(ev0)
                check(lib.mt_transpose_bf16(ptr(Xs[L]), K1, M, K1, ptr(XT), Mpad, K1, _st()), "mt_transpose_bf16")
                _gemm(dLT, Mpad, XT, Mpad, gfc, K1, 88, K1, Mpad)
                _gather4(gfc, 0, g["fc.weight"], (1, 1, 88, 2 * H), (0, 0, K1, 1))
                check(lib.mt_rowsum_bf16(ptr(dLT), Mpad, M, ptr(g["fc.bias"]), 88, _st()), "mt_rowsum_bf16")
        # ---- LSTM layers, top to bottom.  dh (gradient of a layer's output in the backward recurrence's layout, with the
        #      layer's dropout mask) is written by the GEMM that produces it: the fc layer's dL W here, the layer above's
        #      dG W_ih below.  Padded units / chunks are never written and stay zero.
        sync_all, sstride = sv["sync_all"], sv["sync_stride"]
        dh = torch.zeros(lib.mt_lstm_cx_bytes(B, T, Hp) // 4, **f32)
        check(lib.mt_gemm_lstm_dh(ptr(dL), 128, ptr(pk["fc_wT"]), 128, ptr(dh), B, T, Hp, H, 128, 0.0, sv["seed"], L - 1, _st()),
              "mt_gemm_lstm_dh (fc)")
        # two dgate buffers: the side stream still unpacks layer l's into dGT while the main stream's recurrence of layer l-1 runs
        dgxs = [torch.empty(lib.mt_lstm_dgx_bytes(B, T, Hp), device=dev, dtype=torch.uint8) for _ in range(min(2, L))]
        ev_unp = [None, None]
        # two partial-product workspaces: the first was filled with the poison pattern during the forward pass, the fill of
        # the next layer's (1 GB at H = 512) runs on the side stream under the current layer's recurrence
        parts = sv["parts"]
        ev_part = [sv["ev_part0"], None]
        Hr = _ru(Hp, 128)
        dX0 = None
        # Every weight-gradient GEMM contracts over positions, so it wants the forward activations transposed (X_l^T,
        # h_{t-1}^T).  None of that depends on the backward pass: it is all produced now, on the side stream,
        # under the top layer's backward recurrence, instead of after the bottom layer's where nothing is left to hide it.
        Npos = B * F1 * T
        XTs = [torch.empty(_ru(K0 if l == 0 else K1, 128) * Mpad, **bf) for l in range(L)]
        HTs = [torch.zeros(2 * Hr, Mpad, **bf) for l in range(L)]
        keep += XTs + HTs

        def enqueue_precompute():
            with torch.cuda.stream(side):
                for l in range(L - 1, -1, -1):
                    K = K0 if l == 0 else K1
                    check(lib.mt_transpose_bf16(ptr(Xs[l]), K, M, K, ptr(XTs[l]), Mpad, K, _st()), "mt_transpose_bf16")
                    check(lib.mt_lstm_hprev_t(ptr(sv["hxs"][l]), ptr(HTs[l]), Mpad, Hr, B, T, Hp, _st()), "mt_lstm_hprev_t")
        # dW_hh of both directions as one split-K launch when T*B splits evenly (4 Hp x Hp outputs are only 64 tiles)
        nkt = Mpad // 64
        Sh = next((c for c in (8, 7, 6, 5, 4, 3, 2) if nkt % c == 0), 1)
        # The weight gradients of a layer (dW_ih, dW_hh, db: transposes + GEMMs with K = T*B) are not needed by the layers
        # below it: they run on a side stream under the next layer's backward recurrence, which is latency-bound on 32 CUs.
        for l in range(L - 1, -1, -1):
            K = K0 if l == 0 else K1
            it = (L - 1 - l) % 2
            part = parts[it % len(parts)]
            if ev_part[it] is not None:
                main.wait_event(ev_part[it])             # this workspace's poison fill (issued on the side stream) is done
            if l > 0 and len(parts) == 2:                # fill the other workspace for the next layer down, beside this recurrence
                evp = torch.cuda.Event()
                evp.record(main)                         # (its previous user, two layers up, has finished in main-stream order)
                with torch.cuda.stream(side):
                    side.wait_event(evp)
                    check(lib.mt_lstm_bwd_poison(ptr(parts[1 - it]), parts[1 - it].numel(), B, T, Hp, _st()), "mt_lstm_bwd_poison")
                    ev_part[1 - it] = torch.cuda.Event()
                    ev_part[1 - it].record(side)
            sync = sync_all[(L + l) * sstride:(L + l + 1) * sstride]
            dgx = dgxs[it % len(dgxs)]
            if ev_unp[it % len(dgxs)] is not None:
                main.wait_event(ev_unp[it % len(dgxs)])  # the side stream's unpack of this buffer's previous user is done
            with prof(f"lstm_bptt_l{l}"):
                check(lib.mt_lstm_bidir_bwd_ex(ptr(sv["gates"][l]), ptr(sv["cxs"][l]), ptr(dh), ptr(pk["w_hh"][l]), ptr(dgx), ptr(part), part.numel(),
                                               ptr(sync), sync.numel(), B, T, Hp, 1, _st()), "mt_lstm_bidir_bwd")
            ev = torch.cuda.Event()
            ev.record(main)                              # dgates of this layer are complete
            if l == L - 1:                           # the host queues the side-stream work only once the top recurrence is running
                enqueue_fc_wgrad()
                enqueue_precompute()
            # dG (rows = positions) feeds the input-gradient GEMM on the main stream; dGT (rows = gate units) only the weight
            # gradients, so its half of the unpack runs on the side stream
            # (dGT is also read as a GEMM A operand from row 4Hp (reverse direction): whole 128-row tiles must stay inside it)
            dG, dGT = torch.empty(Mpad, 8 * Hp, **bf), torch.empty(4 * Hp + _ru(4 * Hp, 128), Mpad, **bf)
            dG[M:].zero_()                           # the unpack writes every column of the M valid rows / every row's M valid columns;
            check(lib.mt_lstm_dg_unpack(ptr(dgx), ptr(dG), 8 * Hp, None, 0, B, T, Hp, _st()), "mt_lstm_dg_unpack")
            # buffers of the side-stream work are allocated here, on the main stream (stream-ordered allocator)
            XTl, HT = XTs[l], HTs[l]
            gb, gwi, gwh = torch.empty(8 * Hp, **f32), torch.empty(8 * Hp, K, **f32), torch.empty(2, 4 * Hp, Hp, **f32)
            Ph = torch.empty(2, Sh, 4 * Hp, Hp, **f32) if Sh > 1 else None
            outs = []
            for di, suf in enumerate(("", "_reverse")):
                outs.append((newg(f"rnn.weight_ih_l{l}{suf}", 4 * H, 64 * Fo2 if l == 0 else 2 * H), newg(f"rnn.weight_hh_l{l}{suf}", 4 * H, H),
                             newg(f"rnn.bias_ih_l{l}{suf}", 4 * H), newg(f"rnn.bias_hh_l{l}{suf}", 4 * H)))
            keep += [dG, dGT, gb, gwi, gwh, Ph]
            # ---- input gradient: the only product the next layer down waits for
            if l > 0:                                # -> dh of layer l-1, whose output went through dropout in the forward pass
                check(lib.mt_gemm_lstm_dh(ptr(dG), 8 * Hp, ptr(pk["w_ihT"][l]), 8 * Hp, ptr(dh), B, T, Hp, H, 8 * Hp,
                                          float(sv["dropout"]), sv["seed"], l - 1, _st()), "mt_gemm_lstm_dh")
            else:
                dX0 = torch.empty(M, K0, **f32)
                _gemm(dG, 8 * Hp, pk["w_ihT"][0], 8 * Hp, dX0, K0, M, K0, 8 * Hp)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dGT[:, M:].zero_()                   # the K-padding (columns M.. of dGT) and the tile rows past 8 Hp must be zero
                dGT[8 * Hp:].zero_()
                check(lib.mt_lstm_dg_unpack(ptr(dgx), None, 0, ptr(dGT), Mpad, B, T, Hp, _st()), "mt_lstm_dg_unpack")
                ev_unp[it % len(dgxs)] = torch.cuda.Event()
                ev_unp[it % len(dgxs)].record(side)
                check(lib.mt_rowsum_bf16(ptr(dGT), Mpad, M, ptr(gb), 8 * Hp, _st()), "mt_rowsum_bf16")
                _gemm(dGT, Mpad, XTl, Mpad, gwi, K, 8 * Hp, K, Mpad)
                if Sh > 1:                               # batch z = direction * Sh + K-slice
                    Kh = Mpad // Sh
                    check(lib.mt_gemm_batched_f32(ptr(dGT), Mpad, 4 * Hp * Mpad, Kh, ptr(HT), Mpad, Hr * Mpad, Kh, None, ptr(Ph), Hp,
                                                  Sh * 4 * Hp * Hp, 4 * Hp * Hp, 4 * Hp, Hp, Kh, 2 * Sh, Sh, _st()), "mt_gemm_batched_f32 (dW_hh)")
                    for di in range(2):
                        check(lib.mt_sum_slices_f32(ptr(Ph[di]), 4 * Hp * Hp, Hp, Sh, ptr(gwh[di]), Hp, 4 * Hp, Hp, _st()), "mt_sum_slices_f32")
                else:
                    for di in range(2):
                        _gemm(dGT[di * 4 * Hp:], Mpad, HT[di * Hr:], Mpad, gwh[di], Hp, 4 * Hp, Hp, Mpad)
                for di, suf in enumerate(("", "_reverse")):
                    wi, wh, bb, bb2 = outs[di]
                    if l == 0:
                        _gather4(gwi, di * 4 * Hp * K, wi, (4, H, 64, Fo2), (Hp * K, K, 1, 64))
                    else:
                        _gather4(gwi, di * 4 * Hp * K, wi, (4, H, 1, 2 * H), (Hp * K, K, 0, 1))
                    _gather4(gwh, di * 4 * Hp * Hp, wh, (4, H, 1, H), (Hp * Hp, Hp, 0, 1))
                    _gather4(gb, di * 4 * Hp, bb, (1, 1, 4, H), (0, 0, Hp, 1))
                    _gather4(gb, di * 4 * Hp, bb2, (1, 1, 4, H), (0, 0, Hp, 1))
                    g[f"rnn.weight_ih_l{l}{suf}"], g[f"rnn.weight_hh_l{l}{suf}"] = wi, wh
                    g[f"rnn.bias_ih_l{l}{suf}"], g[f"rnn.bias_hh_l{l}{suf}"] = bb, bb2
            if l == 1 and getattr(model, "_grad_sync", None) is not None:
                # data parallel: everything above layer 0 (LSTM layers >= 1 and the fc: the tail of the flat gradient) is complete
                # in the side stream's order -- its all-reduce starts now, under layer 0's recurrence, weight-gradient GEMM and the
                # convolution backward (optim.EarlyBucket)
                early = {k: v for k, v in g.items() if k.startswith("fc.") or (k.startswith("rnn.") and not k.split("_l")[-1].startswith("0"))}
                for k in model._grad_sync.reduce_early(early, side):
                    keep.append(g[k])                    # (read on the side stream: stays referenced until the streams have joined)
                    g[k] = None
        # ---- conv2: BN + ReLU + pool backward, dgrad (flipped-weight conv), wgrad (split-K GEMM over positions)
        sums = torch.zeros(512, device=dev, dtype=torch.float64)
        dz2, dz2lo = torch.empty(Npos, 64, **bf), torch.empty(Npos, 64, **bf)
        g["cnn.5.weight"], g["cnn.5.bias"] = torch.empty(64, **f32), torch.empty(64, **f32)
        check(lib.mt_bn_pool_bwd_tie(ptr(dX0), K0, ptr(sv["z2"]), ptr(sv["mean2"]), ptr(sv["rstd2"]), ptr(pk["g2"]), ptr(pk["be2"]), ptr(sums),
                                     ptr(dz2), ptr(dz2lo), ptr(g["cnn.5.weight"]), ptr(g["cnn.5.bias"]), ptr(sv["tie2"]), B, F1, T, _st()),
              "mt_bn_pool_bwd_tie")
        da1 = torch.empty(B, F1, T, 64, **bf)
        check(lib.mt_conv_cl_bf16(ptr(dz2), None, ptr(pk["w2d"]), ptr(pk["zero64"]), ptr(da1), B, F1, T, 64, 0, 64, 3, 0, 0, 0, 0, _st()),
              "mt_conv_cl_bf16 (dgrad)")
        nwg = lib.mt_conv2_wgrad_workgroups()
        P, Pb = torch.empty(nwg, 64, 288, **f32), torch.empty(nwg, 64, **f32)      # per-workgroup partial sums
        gw2 = torch.empty(64, 288, **f32)
        g["cnn.4.weight"], g["cnn.4.bias"] = torch.empty(64, 32, 3, 3, **f32), torch.empty(64, **f32)
        keep += [dz2, dz2lo, P, Pb, gw2]
        ev2 = torch.cuda.Event()
        ev2.record(main)
        with torch.cuda.stream(side):                    # conv2 weight gradient beside the conv1 backward: positions are the MFMA
            side.wait_event(ev2)                         # contraction index, no im2col and no transposed dz (csrc/train.hip)
            check(lib.mt_conv2_wgrad(ptr(sv["a1"]), ptr(dz2), ptr(dz2lo), ptr(P), ptr(Pb), nwg, B, F1, T, _st()), "mt_conv2_wgrad")
            check(lib.mt_sum_slices_f32(ptr(P), 64 * 288, 288, nwg, ptr(gw2), 288, 64, 288, _st()), "mt_sum_slices_f32")
            _gather4(gw2, 0, g["cnn.4.weight"], (1, 64, 32, 9), (0, 288, 1, 32))
            check(lib.mt_sum_slices_f32(ptr(Pb), 64, 64, nwg, ptr(g["cnn.4.bias"]), 64, 1, 64, _st()), "mt_sum_slices_f32")
        if debug is not None:
            debug.update(dX0=dX0, dz2=dz2, dz2lo=dz2lo, da1=da1, gw2=gw2, P=P)
        # ---- conv1 (z1 recomputed from the input)
        g["cnn.0.weight"], g["cnn.0.bias"] = torch.empty(32, 1, 3, 3, **f32), torch.empty(32, **f32)
        g["cnn.1.weight"], g["cnn.1.bias"] = torch.empty(32, **f32), torch.empty(32, **f32)
        check(lib.mt_conv1_bwd(ptr(x), ptr(pk["w1"]), ptr(pk["b1"]), ptr(sv["mean1"]), ptr(sv["rstd1"]), ptr(pk["g1"]), ptr(pk["be1"]),
                               ptr(da1), 64, ptr(sums[128:]), ptr(g["cnn.0.weight"]), ptr(g["cnn.0.bias"]), ptr(g["cnn.1.weight"]),
                               ptr(g["cnn.1.bias"]), B, F, T, _st()), "mt_conv1_bwd")
        main.wait_stream(side)                       # every gradient is complete in the caller's stream order
    del keep
    return g


class CnnRnnTrainFn(torch.autograd.Function):
    """logits = CNNRNNModel(x) in train mode; backward returns the gradient of every parameter."""

    @staticmethod
    def forward(ctx, model, x, dropout, seed, names, *params):
        logits, sv = forward_train(model, x, dropout, seed)
        ctx.model, ctx.sv, ctx.names = model, sv, names
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        g = backward_train(ctx.model, ctx.sv, dlogits)
        for n in ctx.sv.get("direct_grads", ()):        # already in the flat gradient buffer
            g[n] = None
        ctx.sv = None
        return (None, None, None, None, None) + tuple(g[n] for n in ctx.names)


def train_forward(model, x: torch.Tensor) -> torch.Tensor:
    names = [n for n, _ in model.named_parameters()]
    params = [p for _, p in model.named_parameters()]
    p = float(model.rnn.dropout) if model.num_layers > 1 else 0.0
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0.0 else 0
    return CnnRnnTrainFn.apply(model, x, p, seed, names, *params)
