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
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import lib, check, ptr, CnnRnnWeights

BN_EPS = 1e-5


def _round_up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


def _fold_bn(conv_w, conv_b, bn):
    """Conv2d followed by eval-mode BatchNorm2d == Conv2d with w*s, (b-mu)*s+beta, s = gamma/sqrt(var+eps)."""
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    w = conv_w.detach().float() * s[:, None, None, None]
    b = (conv_b.detach().float() - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
    return w, b


def _bf16(t):
    return t.to(torch.bfloat16).contiguous()


class _HipForward:
    """Mixin: packed-weight cache keyed on parameter versions + per-shape workspaces."""

    def _param_signature(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def _ensure_packed(self, device):
        sig = (str(device), self._param_signature())
        if getattr(self, "_pack_sig", None) != sig:
            self._packed = self._pack(device)
            self._pack_sig = sig
            self._ws = {}
        return self._packed

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
        if H % 16 or H > 1024:
            raise NotImplementedError(f"hidden_size={H}: the recurrence kernel needs a multiple of 16, <= 1024")
        if L > _lib.MAX_LSTM_LAYERS:
            raise NotImplementedError(f"num_layers={L} > {_lib.MAX_LSTM_LAYERS}")
        dev = dict(device=device)
        w1, b1 = _fold_bn(self.cnn[0].weight, self.cnn[0].bias, self.cnn[1])
        w2, b2 = _fold_bn(self.cnn[4].weight, self.cnn[4].bias, self.cnn[5])
        t = {"conv1_w": w1.reshape(32, 9).contiguous().to(**dev), "conv1_b": b1.contiguous().to(**dev),
             "conv2_w": _bf16(w2.permute(0, 2, 3, 1).reshape(64, 9, 32)).to(**dev),   # [co][tap][ci]
             "conv2_b": b2.contiguous().to(**dev)}
        K1 = _round_up(2 * H, 64)
        npad = _round_up(8 * H, 128)
        for l in range(L):
            wi = torch.cat([getattr(self.rnn, f"weight_ih_l{l}").detach().float(),
                            getattr(self.rnn, f"weight_ih_l{l}_reverse").detach().float()], 0)     # (8H, K)
            if l == 0:   # reference feature index c*Fo2+f (cnn_rnn_model.py:60-62) -> kernel's f*64+c
                wi = wi.reshape(8 * H, 64, Fo2).permute(0, 2, 1).reshape(8 * H, Fo2 * 64)
                K = Fo2 * 64
            else:
                K = K1
            wp = torch.zeros(npad, K)
            wp[:8 * H, :wi.shape[1]] = wi
            t[f"w_ih{l}"] = _bf16(wp).to(**dev)
            t[f"b_g{l}"] = torch.cat([
                getattr(self.rnn, f"bias_ih_l{l}").detach().float() + getattr(self.rnn, f"bias_hh_l{l}").detach().float(),
                getattr(self.rnn, f"bias_ih_l{l}_reverse").detach().float() + getattr(self.rnn, f"bias_hh_l{l}_reverse").detach().float(),
            ]).contiguous().to(**dev)
            t[f"w_hh{l}"] = torch.stack([getattr(self.rnn, f"weight_hh_l{l}").detach().float(),
                                         getattr(self.rnn, f"weight_hh_l{l}_reverse").detach().float()]).contiguous().to(**dev)
        fw = torch.zeros(128, K1)
        fw[:88, :2 * H] = self.fc.weight.detach().float()
        t["fc_w"] = _bf16(fw).to(**dev)
        t["fc_b"] = self.fc.bias.detach().float().contiguous().to(**dev)
        w = CnnRnnWeights()
        w.n_mels, w.hidden, w.layers = self.n_mels, H, L
        w.conv1_w, w.conv1_b, w.conv2_w, w.conv2_b = (ptr(t[k]) for k in ("conv1_w", "conv1_b", "conv2_w", "conv2_b"))
        for l in range(L):
            w.w_ih[l], w.b_gates[l], w.w_hh[l] = ptr(t[f"w_ih{l}"]), ptr(t[f"b_g{l}"]), ptr(t[f"w_hh{l}"])
        w.fc_w, w.fc_b = ptr(t["fc_w"]), ptr(t["fc_b"])
        return {"tensors": t, "struct": w}

    def forward(self, x, chunk_max_power: Optional[torch.Tensor] = None, check_status: bool = False, events=None):
        self._require_cuda(x)
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            raise NotImplementedError("the HIP path implements the eval-mode forward; the training step "
                                      "(backward kernels) is not built yet -- call model.eval() / torch.no_grad()")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.n_mels:
            raise ValueError(f"expected (B, 1, {self.n_mels}, T), got {tuple(x.shape)}")
        B, _, _, T = x.shape
        if T == 0:   # the reference's guard (cnn_rnn_model.py:65-66); its own conv raises before reaching it
            return torch.zeros(B, self.output_dim, 1, device=x.device)
        pk = self._ensure_packed(x.device)
        w = pk["struct"]
        x = x.contiguous().float()
        logits = torch.empty(B, self.output_dim, T, dtype=torch.float32, device=x.device)
        # one workspace per (shape, stream): forwards issued on different streams may overlap on the GPU
        key = (B, T, torch.cuda.current_stream(x.device).cuda_stream)
        if key not in self._ws:
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
        with torch.cuda.device(x.device):
            check(lib.mt_cnnrnn_forward_ex(w, ptr(x), ptr(chunk_max_power), B, T, ptr(logits), ptr(ws), ws.numel(),
                                           ev_arr, n_ev, _lib.stream_ptr()), "mt_cnnrnn_forward")
        if check_status:
            self.raise_on_handoff_timeout(B, T)
        return logits

    def raise_on_handoff_timeout(self, B, T):
        """Synchronises; raises if a recurrence hand-off spin hit its bound (see csrc/lstm.hip)."""
        w = self._packed["struct"]
        torch.cuda.synchronize()
        for key, ws in self._ws.items():
            if key[:2] != (B, T):
                continue
            for l in range(self.num_layers):
                off = lib.mt_cnnrnn_status_offset(w, B, T, l)
                st = int(ws[off:off + 4].view(torch.int32).item())
                if st != 0:
                    raise _lib.MtError(f"LSTM layer {l}: inter-workgroup hand-off timed out at step {st - 1}")


class CNNRNNModelLarge(nn.Module, _HipForward):
    """Parameter container with the reference's names (models/cnn_rnn_model.py:142-256).
    Its HIP forward (residual blocks, 7x3 conv, dual LSTM, clamped attention, heads) is the
    next SURVEY-8 row and is not built yet: forward raises instead of falling back."""

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

    def forward(self, x, return_all_heads=False):
        self._require_cuda(x)
        raise NotImplementedError("CNNRNNModelLarge: HIP forward not built yet (SURVEY 8 rows a4-a6)")


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
