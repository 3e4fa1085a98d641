"""Host-side mirror of the reference model surface over the HIP library.

`TranscriptionModel`, `CNNRNNModel` and `CNNRNNModelLarge` keep the reference's
constructor arguments, attribute names and `state_dict` key names/shapes
(models/transcription_model.py:26-89, models/cnn_rnn_model.py:16-55,:154-256), so
reference checkpoints load with `load_state_dict(strict=True)` and the same
`torch.manual_seed` yields the same initial weights (the parameter containers are
created in the reference's registration order).  The torch modules below only HOLD
parameters: every forward pass runs in libmt_hip.so (csrc/*.hip).  There is no CPU
or eager-torch fallback -- a CPU tensor, a missing library or a mode the kernels do
not implement yet raises.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import lib, check, ptr, CnnRnnWeights, CnnRnnLargeWeights

BN_EPS = 1e-5


def _round_up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


def _fold_bn(conv_w, conv_b, bn):
    """Conv2d followed by eval-mode BatchNorm2d == Conv2d with w*s, (b-mu)*s+beta, s = gamma/sqrt(var+eps)."""
    s = bn.weight.detach().float().cpu() / torch.sqrt(bn.running_var.detach().float().cpu() + bn.eps)
    w = conv_w.detach().float().cpu() * s[:, None, None, None]
    b = (conv_b.detach().float().cpu() - bn.running_mean.detach().float().cpu()) * s + bn.bias.detach().float().cpu()
    return w, b


def _bf16(t):
    return t.to(torch.bfloat16).contiguous()


def operand_dtype(model=None) -> int:
    """16-bit operand type of the inference GEMM / conv kernels (MT_DT_*): f16 by default -- 11 significand bits against
    bf16's 8 at the same MFMA rate, which is what keeps the logits (and the framewise F1) on the fp32 reference's
    (tests/test_gpu_f1_parity.py).  `model.operand_dtype = "bf16"` or MT_OPERAND_DTYPE=bf16 select
    bf16 (A/B comparisons).  The training step always runs bf16 (train_step.py)."""
    v = os.environ.get("MT_OPERAND_DTYPE") or getattr(model, "operand_dtype", "f16")
    if v not in ("f16", "bf16"):
        raise ValueError(f"operand dtype must be 'f16' or 'bf16', got {v!r}")
    return _lib.DT_F16 if v == "f16" else _lib.DT_BF16


def _h16(t, dt: int):
    """f32 -> the 16-bit operand type (f16 saturates instead of overflowing, like the kernels' conversion)."""
    if dt == _lib.DT_F16:
        return t.clamp(-65504.0, 65504.0).to(torch.float16).contiguous()
    return t.to(torch.bfloat16).contiguous()


def _pad2(t, rows, cols):
    out = torch.zeros(rows, cols, dtype=t.dtype)
    out[:t.shape[0], :t.shape[1]] = t
    return out


def _pack_bilstm(rnn: nn.LSTM, layers: int, H: int, k0_cols, dev, dt: int = _lib.DT_BF16, k0_cf=None, wih0_out=None):
    """Pack a bidirectional nn.LSTM for mt_gemm_lstm_gx + mt_lstm_bidir_fwd (computed on `dev`: the training
    step re-packs after every optimizer step).  The hidden size is laid out padded to Hp = roundup(H, 16): a
    padded unit has zero weights and bias, so its gates are 0 and its c, h stay 0.  Gate row p*Hp + j;
    directions stacked [fwd; reverse]; layer 0's columns are re-ordered by `k0_cols` (index tensor: kernel
    column -> reference column); deeper layers take the compact [fwd H | reverse H] rows padded to
    roundup(2H, 64) columns.  Returns (w_ih[], b_gates[], w_hh[]).
    k0_cf = (C, F): the re-ordering is kernel column f*C + c <- reference column c*F + f and layer 0's W_ih is packed by
    mt_pack_wih_cf (one launch per direction instead of an index gather, a copy and a cast); wih0_out: a 16-bit tensor
    [>= roundup(8 Hp, 128) rows][K] to pack it into (the caller owns the rows beyond 8 Hp)."""
    Hp = _round_up(H, 16)
    K1 = _round_up(2 * H, 64)
    k0 = k0_cols.to(dev)
    w_ih, b_g, w_hh = [], [], []
    for l in range(layers):
        K = k0.numel() if l == 0 else 2 * H
        Kp = K if l == 0 else K1
        fast0 = l == 0 and k0_cf is not None and torch.device(dev).type == "cuda" and k0_cf[0] * (k0_cf[1] | 1) * 4 <= 64 * 1024
        if fast0:
            C_, F_ = k0_cf
            assert C_ * F_ == K
            w16 = wih0_out
            if w16 is None:
                w16 = torch.empty(_round_up(8 * Hp, 128), K, dtype=torch.float16 if dt == _lib.DT_F16 else torch.bfloat16, device=dev)
                w16[8 * Hp:].zero_()
            with torch.cuda.device(dev):
                for di, suf in enumerate(("", "_reverse")):
                    w = getattr(rnn, f"weight_ih_l0{suf}").detach()
                    w = w if (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()) else w.to(dev, torch.float32).contiguous()
                    _lib.check(_lib.lib.mt_pack_wih_cf(_lib.ptr(w), _lib.ptr(w16), K, di * 4 * Hp, H, Hp, C_, F_, dt, _lib.stream_ptr()), "mt_pack_wih_cf")
        wcat = None if fast0 else torch.zeros(_round_up(8 * Hp, 128), Kp, dtype=torch.float32, device=dev)
        bcat = torch.zeros(2, 4, Hp, dtype=torch.float32, device=dev)
        hcat = torch.zeros(2, 4, Hp, Hp, dtype=torch.float32, device=dev)
        for di, suf in enumerate(("", "_reverse")):
            if not fast0:
                w = getattr(rnn, f"weight_ih_l{l}{suf}").detach().to(dev, torch.float32)
                w = w[:, k0] if l == 0 else w
                wcat[di * 4 * Hp:(di + 1) * 4 * Hp].view(4, Hp, Kp)[:, :H, :K] = w.reshape(4, H, K)
            b = (getattr(rnn, f"bias_ih_l{l}{suf}") + getattr(rnn, f"bias_hh_l{l}{suf}")).detach().to(dev, torch.float32)
            bcat[di, :, :H] = b.reshape(4, H)
            hcat[di, :, :H, :H] = getattr(rnn, f"weight_hh_l{l}{suf}").detach().to(dev, torch.float32).reshape(4, H, H)
        w_ih.append(w16 if fast0 else _h16(wcat, dt))
        b_g.append(bcat.reshape(-1).contiguous())
        w_hh.append(hcat.reshape(2, 4 * Hp, Hp).contiguous())
    return w_ih, b_g, w_hh


class _HipForward:
    """Mixin: packed-weight cache keyed on parameter versions + per-shape workspaces."""

    def _param_signature(self):
        from .optim import WEIGHTS_EPOCH
        return (WEIGHTS_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _ensure_packed(self, device):
        sig = (str(device), operand_dtype(self), self._param_signature())
        if getattr(self, "_pack_sig", None) != sig:
            # Once per weight change.  Forwards may be in flight on several caller streams (bench.py keeps three): the old
            # packed tensors and workspaces must not go back to the allocator while a stream still reads them, and the new
            # ones (packed by kernels on THIS stream) must be complete before another stream's forward uses them.
            if torch.cuda.is_available():
                torch.cuda.synchronize(device)
            self._packed = self._pack(device)
            self._pack_sig = sig
            self._ws = {}
            if torch.cuda.is_available():
                torch.cuda.synchronize(device)
        return self._packed

    # ---- hand-off status of the persistent recurrence launches (csrc/lstm.hip): every spin is bounded and reports
    #      through a status word per layer; a timed-out layer leaves NaN poison in its output, so whoever syncs checks.
    def _status_offsets(self, B, T):
        raise NotImplementedError

    def raise_on_handoff_timeout(self, B=None, T=None, sync: bool = True):
        """Raises MtError if a recurrence hand-off spin of any forward issued so far hit its bound.  sync=False: the
        caller has already synchronised with the forwards it cares about (e.g. by copying their result to the host);
        the check then costs one 4-byte-per-layer copy."""
        if not getattr(self, "_ws", None):
            return
        if sync:
            torch.cuda.synchronize()
        words, names = [], []
        for key, ws in self._ws.items():
            if (B is not None and key[0] != B) or (T is not None and key[1] != T):
                continue
            offs = self._status_offsets(key[0], key[1])
            idx = torch.tensor([o // 4 for _, o in offs], dtype=torch.int64, device=ws.device)
            words.append(ws.view(torch.int32)[idx])
            names += [n for n, _ in offs]
        if not words:
            return
        vals = torch.cat(words).cpu().tolist()
        for n, st in zip(names, vals):
            if st != 0:
                kind = "payload" if st & 0x40000000 else "flag"
                raise _lib.MtError(f"LSTM {n}: inter-workgroup hand-off ({kind} spin) timed out at step {(st & 0x3fffffff) - (0 if st & 0x40000000 else 1)}: "
                                   "the launch was not fully resident (too many persistent launches in flight on this GPU?)")

    def raise_on_train_handoff_timeout(self):
        """The same for the persistent launches of the last training step (forward and backward recurrences); the
        caller has synchronised with the step (train.train_one_epoch reads the loss and the optimizer's statistics)."""
        ts = getattr(self, "_train_sync", None)
        if ts is None:
            return
        buf, stride = ts
        vals = buf.view(torch.int32)[:: stride // 4].cpu().tolist()
        for i, st in enumerate(vals):
            if st != 0:
                raise _lib.MtError(f"training step: persistent recurrence launch {i} timed out in its inter-workgroup hand-off (status {st:#x})")

    def _check_inflight_bound(self, key, bound, what):
        """Co-residency rule of the persistent recurrence kernels: at most `bound` forwards of this
        model in flight per GPU, i.e. at most `bound` caller streams."""
        streams = {k[2] for k in self._ws} | {key[2]}
        if len(streams) > bound:                       # forget the streams that have drained: nothing of theirs is in flight
            for k in list(self._ws):
                if k[2] != key[2] and torch.cuda.ExternalStream(k[2], device=self._ws[k].device).query():
                    self._ws.pop(k)
            streams = {k[2] for k in self._ws} | {key[2]}
        if len(streams) > bound:
            raise _lib.MtError(f"{what}: {len(streams)} caller streams > {bound}: the persistent recurrence launches of that many "
                               "forwards cannot all be resident on one GPU (they would stall on each other)")

    @staticmethod
    def _require_cuda(x):
        if not x.is_cuda:
            raise RuntimeError("music_transcription_amd runs on the MI355X only: got a CPU tensor "
                               "(there is no CPU fallback in the product path)")


class CNNRNNModel(nn.Module, _HipForward):
    """CNN + bi-LSTM + FC transcriber (reference models/cnn_rnn_model.py:5-74).
    Input (B, 1, n_mels, T) float32 dB mel on the GPU -> logits (B, 88, T)."""

    def __init__(self, n_mels: int = 229, hidden_size: int = 256, num_layers: int = 2, dropout: float = 0.3):
        super().__init__()
        self.n_mels, self.hidden_size, self.num_layers, self.output_dim = n_mels, hidden_size, num_layers, 88
        self.cnn = nn.Sequential(
            nn.Conv2d(1, 32, kernel_size=(3, 3), padding=(1, 1)), nn.BatchNorm2d(32), nn.ReLU(), nn.MaxPool2d((2, 1)),
            nn.Conv2d(32, 64, kernel_size=(3, 3), padding=(1, 1)), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d((2, 1)))
        self.rnn = nn.LSTM(input_size=64 * (n_mels // 4), hidden_size=hidden_size, num_layers=num_layers,
                           dropout=dropout, batch_first=True, bidirectional=True)
        self.fc = nn.Linear(hidden_size * 2, self.output_dim)

    # ---- weight packing (once per load_state_dict): layouts documented in include/mt_hip.h
    def _pack(self, device) -> Dict[str, object]:
        H, L, Fo2 = self.hidden_size, self.num_layers, self.n_mels // 4
        if H > 1024:
            raise NotImplementedError(f"hidden_size={H}: the recurrence kernel keeps W_hh slices in registers, H <= 1024")
        if L > _lib.MAX_LSTM_LAYERS:
            raise NotImplementedError(f"num_layers={L} > {_lib.MAX_LSTM_LAYERS}")
        dev = dict(device=device)
        dt = operand_dtype(self)
        w1, b1 = _fold_bn(self.cnn[0].weight, self.cnn[0].bias, self.cnn[1])
        w2, b2 = _fold_bn(self.cnn[4].weight, self.cnn[4].bias, self.cnn[5])
        t = {"conv1_w": w1.reshape(32, 9).contiguous().to(**dev), "conv1_b": b1.contiguous().to(**dev),
             "conv2_w": _h16(w2.permute(0, 2, 3, 1).reshape(64, 9, 32), dt).to(**dev),   # [co][tap][ci]
             "conv2_b": b2.contiguous().to(**dev)}
        K1 = _round_up(2 * H, 64)
        # reference feature index c*Fo2+f (cnn_rnn_model.py:60-62) -> kernel column f*64+c
        cols = (torch.arange(64)[None, :] * Fo2 + torch.arange(Fo2)[:, None]).reshape(-1)
        wi, bg, wh = _pack_bilstm(self.rnn, L, H, cols, device, dt, k0_cf=(64, Fo2))
        for l in range(L):
            t[f"w_ih{l}"], t[f"b_g{l}"], t[f"w_hh{l}"] = wi[l], bg[l], wh[l]
        # layers > 0: W_ih in the layout of the fused input projection (mt_lstm_bidir_fwd_xproj): f32 [2][4Hp][2Hp],
        # column dir'*Hp + k of the previous layer's (padded) hidden units
        Hp = _round_up(H, 16)
        for l in range(1, L):
            wx = torch.zeros(2, 4, Hp, 2, Hp, dtype=torch.float32, device=device)
            for di, suf in enumerate(("", "_reverse")):
                wsrc = getattr(self.rnn, f"weight_ih_l{l}{suf}").detach().to(device, torch.float32).reshape(4, H, 2, H)
                wx[di, :, :H, :, :H] = wsrc
            t[f"w_ihx{l}"] = wx.reshape(2, 4 * Hp, 2 * Hp).contiguous()
        fw = torch.zeros(128, K1)
        fw[:88, :2 * H] = self.fc.weight.detach().float().cpu()
        t["fc_w"] = _h16(fw, dt).to(**dev)
        t["fc_b"] = self.fc.bias.detach().float().contiguous().to(**dev)
        w = CnnRnnWeights()
        w.n_mels, w.hidden, w.layers, w.operand_dtype = self.n_mels, H, L, dt
        w.conv1_w, w.conv1_b, w.conv2_w, w.conv2_b = (ptr(t[k]) for k in ("conv1_w", "conv1_b", "conv2_w", "conv2_b"))
        for l in range(L):
            w.w_ih[l], w.b_gates[l], w.w_hh[l] = ptr(t[f"w_ih{l}"]), ptr(t[f"b_g{l}"]), ptr(t[f"w_hh{l}"])
            w.w_ihx[l] = None                        # set per call from self.fuse_input_projection (forward)
        w.fc_w, w.fc_b = ptr(t["fc_w"]), ptr(t["fc_b"])
        return {"tensors": t, "struct": w}

    def forward(self, x, chunk_max_power: Optional[torch.Tensor] = None, check_status: bool = False, events=None):
        self._require_cuda(x)
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.n_mels:
            raise ValueError(f"expected (B, 1, {self.n_mels}, T), got {tuple(x.shape)}")
        B, _, _, T = x.shape
        if T == 0:   # the reference's guard (cnn_rnn_model.py:65-66); its own conv raises before reaching it
            return torch.zeros(B, self.output_dim, 1, device=x.device)
        if x.requires_grad:
            raise NotImplementedError("gradients w.r.t. the input mel are not implemented (the reference never asks for them)")
        if self.training:
            # train mode: BatchNorm batch statistics, LSTM dropout, activations saved, autograd edge to the HIP
            # backward pass (train_step.py); under torch.no_grad() the same kernels run without the graph edge
            if self.hidden_size > 512:
                raise NotImplementedError("training: the backward recurrence kernel supports hidden_size <= 512")
            from .train_step import train_forward
            return train_forward(self, x)
        pk = self._ensure_packed(x.device)
        w = pk["struct"]
        x = x.contiguous().float()
        logits = torch.empty(B, self.output_dim, T, dtype=torch.float32, device=x.device)
        # one workspace per (shape, stream): forwards issued on different streams may overlap on the GPU
        key = (B, T, torch.cuda.current_stream(x.device).cuda_stream)
        env = os.environ.get("MT_LSTM_XPROJ")
        fuse = (env == "1") if env in ("0", "1") else bool(getattr(self, "fuse_input_projection", False))
        if key not in self._ws:
            self._check_inflight_bound(key, 2 if fuse else 6, "CNNRNNModel.forward")
            nbytes = lib.mt_cnnrnn_workspace_bytes(w, B, T)
            if nbytes == 0:
                raise _lib.MtError("mt_cnnrnn_workspace_bytes: " + _lib.last_error())
            for k in [k for k in self._ws if k[:2] != (B, T)] + list(self._ws)[: max(0, len(self._ws) - 7)]:
                self._ws.pop(k, None)                       # keep one shape resident, at most 8 streams
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        ws = self._ws[key]
        ev_arr, n_ev = None, 0
        if events is not None:   # torch.cuda.Event(enable_timing=True) objects, already created (recorded once)
            import ctypes
            n_ev = len(events)
            ev_arr = (ctypes.c_void_p * n_ev)(*[e.cuda_event for e in events])
        # Layers > 0 can take their input projection inside the recurrence (no GEMM, no gx buffer, no re-layout between LSTM
        # layers; csrc/lstm.hip, XP).  It lengthens the latency-bound recurrence and removes GEMM work: a loss with one batch in
        # flight, a gain with several (the GEMMs are the shared resource then) -- so the caller decides.  MT_LSTM_XPROJ=0/1 forces it.
        # A fused recurrence workgroup fills a CU's register file (one per CU, 256 per GPU = two launches): at most TWO forwards
        # of this model in flight per GPU with it, at most six without (the library refuses more: csrc/residency.hip).
        for l in range(1, self.num_layers):
            w.w_ihx[l] = ptr(pk["tensors"][f"w_ihx{l}"]) if (fuse and self.hidden_size <= 512) else None
        with torch.cuda.device(x.device):
            check(lib.mt_cnnrnn_forward_ex(w, ptr(x), ptr(chunk_max_power), B, T, ptr(logits), ptr(ws), ws.numel(),
                                           ev_arr, n_ev, _lib.stream_ptr()), "mt_cnnrnn_forward")
        if check_status:
            self.raise_on_handoff_timeout(B, T)
        return logits

    def _status_offsets(self, B, T):
        w = self._packed["struct"]
        return [(f"layer {l}", lib.mt_cnnrnn_status_offset(w, B, T, l)) for l in range(self.num_layers)]


class CNNRNNModelLarge(nn.Module, _HipForward):
    """Residual CNN + dual bi-LSTM + clamped 8-head attention + frame/onset/offset heads
    (reference models/cnn_rnn_model.py:142-348).  Input (B, 1, n_mels, T) -> logits (B, 88, T), or a dict
    of three with return_all_heads=True."""

    class _Res(nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, cout, (3, 3), 1, (1, 1))
            self.bn1 = nn.BatchNorm2d(cout)
            self.conv2 = nn.Conv2d(cout, cout, (3, 3), 1, (1, 1))
            self.bn2 = nn.BatchNorm2d(cout)
            self.skip = nn.Sequential()
            if cin != cout:
                self.skip = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=1), nn.BatchNorm2d(cout))

    class _Attn(nn.Module):
        def __init__(self, dim, heads):
            super().__init__()
            self.num_heads, self.head_dim = heads, dim // heads
            self.qkv = nn.Linear(dim, dim * 3)
            self.proj = nn.Linear(dim, dim)

    def __init__(self, n_mels=229, hidden_size=512, num_layers=3, dropout=0.2, use_attention=True,
                 use_onset_offset_heads=True, num_attention_heads=8):
        super().__init__()
        self.n_mels, self.hidden_size, self.num_layers, self.output_dim = n_mels, hidden_size, num_layers, 88
        self.use_attention, self.use_onset_offset_heads = use_attention, use_onset_offset_heads
        # train-mode dropouts of the reference: LSTM inter-layer / attention probabilities p, heads 1.5 p
        # (cnn_rnn_model.py:216,:238,:251), and its hard-coded spatial dropouts (:188,:192,:202)
        self.dropout_p, self.dropout2d_p = float(dropout), (0.1, 0.1, 0.15)
        self.conv1 = nn.Sequential(nn.Conv2d(1, 32, (3, 3), padding=(1, 1)), nn.BatchNorm2d(32), nn.ReLU(), nn.MaxPool2d((2, 1)))
        self.res_block1 = self._Res(32, 64)
        self.res_block2 = self._Res(64, 128)
        self.freq_aware_conv = nn.Sequential(nn.Conv2d(128, 256, (7, 3), padding=(3, 1)), nn.BatchNorm2d(256), nn.ReLU(),
                                             nn.MaxPool2d((2, 1)))
        k = 256 * (n_mels // 8)
        self.rnn_main = nn.LSTM(k, hidden_size, num_layers, dropout=dropout if num_layers > 1 else 0,
                                batch_first=True, bidirectional=True)
        self.rnn_local = nn.LSTM(k, hidden_size // 2, 1, batch_first=True, bidirectional=True)
        comb = hidden_size * 2 + (hidden_size // 2) * 2
        if use_attention:
            self.attention = self._Attn(comb, num_attention_heads)
            self.attention_norm = nn.LayerNorm(comb, eps=1e-6)
        if use_onset_offset_heads:
            self.shared_fc = nn.Linear(comb, hidden_size)
            self.frame_head = nn.Linear(hidden_size, 88)
            self.onset_head = nn.Linear(hidden_size, 88)
            self.offset_head = nn.Linear(hidden_size, 88)
        else:
            self.fc = nn.Linear(comb, 88)

    # ---- weight packing: layouts in include/mt_hip.h (mt_cnnrnn_large_weights)
    def _pack(self, device):
        H, L, Hl = self.hidden_size, self.num_layers, self.hidden_size // 2
        if H > 1024 or L > _lib.MAX_LSTM_LAYERS or Hl < 1:
            raise NotImplementedError(f"hidden_size={H}, num_layers={L}: outside what the kernels implement")
        F3 = self.n_mels // 8
        comb = 2 * H + 2 * Hl
        Cp = _round_up(comb, 64)
        t = {}

        def put(name, tensor):
            t[name] = tensor.contiguous().to(device)
            return ptr(t[name])

        def conv_cl(wf):        # [Cout][Cin][KH][KW] -> [Cout][(kh*KW + kw)*Cin + ci]
            return wf.permute(0, 2, 3, 1).reshape(wf.shape[0], -1)

        w = CnnRnnLargeWeights()
        w.n_mels, w.hidden, w.layers, w.hidden_local = self.n_mels, H, L, Hl
        dt = w.operand_dtype = operand_dtype(self)
        w.use_attention, w.use_heads = int(self.use_attention), int(self.use_onset_offset_heads)
        w1, b1 = _fold_bn(self.conv1[0].weight, self.conv1[0].bias, self.conv1[1])
        w.conv1_w, w.conv1_b = put("conv1_w", w1.reshape(32, 9)), put("conv1_b", b1)
        for name, rb in (("rb1", self.res_block1), ("rb2", self.res_block2)):
            wa, ba = _fold_bn(rb.conv1.weight, rb.conv1.bias, rb.bn1)
            wb, bb = _fold_bn(rb.conv2.weight, rb.conv2.bias, rb.bn2)
            if len(rb.skip):
                wsk, bsk = _fold_bn(rb.skip[0].weight, rb.skip[0].bias, rb.skip[1])
                wsk = wsk.reshape(wsk.shape[0], -1)
            else:           # identity skip == 1x1 conv with the identity matrix (exact in bf16)
                wsk, bsk = torch.eye(wb.shape[0]), torch.zeros(wb.shape[0])
            setattr(w, name + "c1_w", put(name + "c1_w", _h16(conv_cl(wa), dt)))
            setattr(w, name + "c1_b", put(name + "c1_b", ba))
            setattr(w, name + "c2_w", put(name + "c2_w", _h16(torch.cat([conv_cl(wb), wsk], 1), dt)))
            setattr(w, name + "c2_b", put(name + "c2_b", bb + bsk))
        wf, bf_ = _fold_bn(self.freq_aware_conv[0].weight, self.freq_aware_conv[0].bias, self.freq_aware_conv[1])
        w.fa_w, w.fa_b = put("fa_w", _h16(conv_cl(wf), dt)), put("fa_b", bf_)
        # reference feature index c*F3+f (cnn_rnn_model.py:292-294) -> kernel column f*256+c
        cols = (torch.arange(256)[None, :] * F3 + torch.arange(F3)[:, None]).reshape(-1)
        wi, bg, wh = _pack_bilstm(self.rnn_main, L, H, cols, device, dt, k0_cf=(256, F3))
        Hp = _round_up(H, 16)
        for l in range(L):
            t[f"m_wi{l}"], t[f"m_b{l}"], t[f"m_wh{l}"] = wi[l], bg[l], wh[l]
            w.main_w_ih[l], w.main_b[l], w.main_w_hh[l] = ptr(wi[l]), ptr(bg[l]), ptr(wh[l])
            w.main_w_ihx[l] = None                   # set per call from self.fuse_input_projection (forward)
            if l > 0:                                # W_ih in the layout of the fused input projection (see CNNRNNModel._pack)
                wx = torch.zeros(2, 4, Hp, 2, Hp, dtype=torch.float32, device=device)
                for di, suf in enumerate(("", "_reverse")):
                    wx[di, :, :H, :, :H] = getattr(self.rnn_main, f"weight_ih_l{l}{suf}").detach().to(device, torch.float32).reshape(4, H, 2, H)
                t[f"m_wix{l}"] = wx.reshape(2, 4 * Hp, 2 * Hp).contiguous()
        wi, bg, wh = _pack_bilstm(self.rnn_local, 1, Hl, cols, device, dt, k0_cf=(256, F3))
        t["l_wi"], t["l_b"], t["l_wh"] = wi[0], bg[0], wh[0]
        w.local_w_ih, w.local_b, w.local_w_hh = ptr(wi[0]), ptr(bg[0]), ptr(wh[0])
        if self.use_attention:
            heads, d = self.attention.num_heads, self.attention.head_dim
            dp = _round_up(d, 64)
            Ca = heads * dp
            w.heads, w.head_dim_pad, w.attn_scale = heads, dp, float(d) ** -0.5
            qw = self.attention.qkv.weight.detach().float().cpu().reshape(3, heads, d, comb)
            qb = self.attention.qkv.bias.detach().float().cpu().reshape(3, heads, d)
            qwp = torch.zeros(3, heads, dp, Cp); qwp[:, :, :d, :comb] = qw
            qbp = torch.zeros(3, heads, dp); qbp[:, :, :d] = qb
            w.qkv_w = put("qkv_w", _h16(_pad2(qwp.reshape(3 * Ca, Cp), _round_up(3 * Ca, 128), Cp), dt))
            w.qkv_b = put("qkv_b", qbp.reshape(-1))
            pw = self.attention.proj.weight.detach().float().cpu().reshape(comb, heads, d)
            pwp = torch.zeros(comb, heads, dp); pwp[:, :, :d] = pw
            w.proj_w = put("proj_w", _h16(_pad2(pwp.reshape(comb, Ca), _round_up(comb, 128), Ca), dt))
            w.proj_b = put("proj_b", self.attention.proj.bias.detach().float().cpu())
            w.ln_g = put("ln_g", self.attention_norm.weight.detach().float().cpu())
            w.ln_b = put("ln_b", self.attention_norm.bias.detach().float().cpu())
        if self.use_onset_offset_heads:
            Hs = _round_up(H, 64)
            w.shared_w = put("shared_w", _h16(_pad2(self.shared_fc.weight.detach().float().cpu(), _round_up(H, 128), Cp), dt))
            w.shared_b = put("shared_b", self.shared_fc.bias.detach().float().cpu())
            hw = torch.cat([m.weight.detach().float().cpu() for m in (self.frame_head, self.onset_head, self.offset_head)], 0)
            hb = torch.cat([m.bias.detach().float().cpu() for m in (self.frame_head, self.onset_head, self.offset_head)], 0)
            w.heads_w, w.heads_b = put("heads_w", _h16(_pad2(hw, 384, Hs), dt)), put("heads_b", hb)
        else:
            w.fc_w = put("fc_w", _h16(_pad2(self.fc.weight.detach().float().cpu(), 128, Cp), dt))
            w.fc_b = put("fc_b", self.fc.bias.detach().float().cpu())
        return {"tensors": t, "struct": w}

    def forward(self, x, return_all_heads=False, chunk_max_power: Optional[torch.Tensor] = None, check_status: bool = False, events=None):
        self._require_cuda(x)
        if x.requires_grad:
            raise NotImplementedError("gradients w.r.t. the input mel are not implemented (the reference never asks for them)")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.n_mels:
            raise ValueError(f"expected (B, 1, {self.n_mels}, T), got {tuple(x.shape)}")
        B, _, _, T = x.shape
        heads_out = self.use_onset_offset_heads
        if T == 0:
            z = torch.zeros(B, self.output_dim, 1, device=x.device)
            return {"frame": z, "onset": z.clone(), "offset": z.clone()} if (heads_out and return_all_heads) else z
        if self.training:
            # train mode: BatchNorm batch statistics, every dropout of the reference, activations saved, autograd edge to
            # the HIP backward pass (train_step_large.py); under torch.no_grad() the same kernels run without the edge
            if self.hidden_size > 512:
                raise NotImplementedError("training: the backward recurrence kernel supports hidden_size <= 512")
            if len(self.res_block1.skip) == 0 or len(self.res_block2.skip) == 0:
                raise NotImplementedError("training: identity skips are not implemented (the reference's blocks change width)")
            from .train_step_large import train_forward_large
            return train_forward_large(self, x, return_all_heads=return_all_heads)
        pk = self._ensure_packed(x.device)
        w = pk["struct"]
        x = x.contiguous().float()
        out = torch.empty(3 if heads_out else 1, B, self.output_dim, T, dtype=torch.float32, device=x.device)
        key = (B, T, torch.cuda.current_stream(x.device).cuda_stream)
        if key not in self._ws:
            self._check_inflight_bound(key, 3, "CNNRNNModelLarge.forward")      # main + local recurrence per forward
            nbytes = lib.mt_cnnrnn_large_workspace_bytes(w, B, T)
            if nbytes == 0:
                raise _lib.MtError("mt_cnnrnn_large_workspace_bytes: " + _lib.last_error())
            for k in [k for k in self._ws if k[:2] != (B, T)] + list(self._ws)[: max(0, len(self._ws) - 3)]:
                self._ws.pop(k, None)
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        ws = self._ws[key]
        # a side stream + fork/join events per caller stream: the local LSTM branch runs beside the main LSTM stack
        if not hasattr(self, "_side"):
            self._side = {}
        if key[2] not in self._side:
            with torch.cuda.device(x.device):
                side = torch.cuda.Stream(device=x.device)
                evs = (torch.cuda.Event(), torch.cuda.Event())
                for e in evs:
                    e.record()                       # creates the underlying hipEvent_t
            self._side[key[2]] = (side, evs)
        side, evs = self._side[key[2]]
        env = os.environ.get("MT_LSTM_XPROJ")
        fuse = (env == "1") if env in ("0", "1") else bool(getattr(self, "fuse_input_projection", False))
        for l in range(1, self.num_layers):
            w.main_w_ihx[l] = ptr(pk["tensors"][f"m_wix{l}"]) if (fuse and self.hidden_size <= 512) else None
        with torch.cuda.device(x.device):
            if events is not None:     # benchmark path: everything on the caller's stream, events at the stage boundaries
                import ctypes
                ev_arr = (ctypes.c_void_p * len(events))(*[e.cuda_event for e in events])
                check(lib.mt_cnnrnn_large_forward_ev(w, ptr(x), ptr(chunk_max_power), B, T, ptr(out), ptr(ws), ws.numel(), ev_arr, len(events),
                                                     _lib.stream_ptr()), "mt_cnnrnn_large_forward_ev")
            else:
                check(lib.mt_cnnrnn_large_forward_ex(w, ptr(x), ptr(chunk_max_power), B, T, ptr(out), ptr(ws), ws.numel(),
                                                     _lib.stream_ptr(), side.cuda_stream, evs[0].cuda_event, evs[1].cuda_event),
                      "mt_cnnrnn_large_forward")
        if check_status:
            self.raise_on_handoff_timeout(B, T)
        if heads_out and return_all_heads:
            return {"frame": out[0], "onset": out[1], "offset": out[2]}
        return out[0]

    def _status_offsets(self, B, T):
        w = self._packed["struct"]
        return [("local" if i == 0 else f"main layer {i - 1}", lib.mt_cnnrnn_large_status_offset(w, B, T, i))
                for i in range(self.num_layers + 1)]


class TranscriptionModel(nn.Module):
    """Drop-in for models/transcription_model.py:16-266 on the CNN-RNN path."""

    def __init__(self, model_type: str = "cnn_rnn", n_mels: int = 229, hidden_size: int = 256, num_layers: int = 2,
                 dropout: float = 0.3, device: str = "cpu", use_attention: bool = True,
                 use_onset_offset_heads: bool = True, **kwargs):
        super().__init__()
        self.model_type = model_type.lower()
        self.device = device
        self.use_onset_offset_heads = use_onset_offset_heads
        if self.model_type in ("cnn_rnn", "cnn+rnn"):
            self.model = CNNRNNModel(n_mels=n_mels, hidden_size=hidden_size, num_layers=num_layers, dropout=dropout)
        elif self.model_type in ("cnn_rnn_large", "large"):
            self.model = CNNRNNModelLarge(n_mels=n_mels, hidden_size=hidden_size, num_layers=num_layers, dropout=dropout,
                                          use_attention=use_attention, use_onset_offset_heads=use_onset_offset_heads)
        elif self.model_type in ("ast", "transformer", "audio_transformer"):
            raise NotImplementedError("the AST experiment (models/transformer_model.py) is outside the MI355X hot path")
        else:
            raise ValueError(f"Unknown model type: {model_type}")
        self.criterion = nn.BCEWithLogitsLoss()
        self.to(device)

    def forward(self, x, return_all_heads=False, **kwargs):
        if self.model_type in ("cnn_rnn_large", "large") and self.use_onset_offset_heads:
            return self.model(x, return_all_heads=return_all_heads)
        return self.model(x)

    def compute_loss(self, logits, targets, lengths=None):
        """Masked BCE-with-logits (single tensor or frame/onset/offset dict); see ops.compute_loss."""
        from . import ops
        return ops.compute_loss(logits, targets, lengths)

    @torch.no_grad()
    def predict(self, x, threshold: float = 0.5, **kwargs):
        from . import ops
        return ops.predict_from_logits(self.forward(x), threshold)
