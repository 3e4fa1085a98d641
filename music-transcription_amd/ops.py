"""Loss, prediction and framewise-F1 on the GPU (csrc/post.hip) behind the reference's semantics:
compute_loss (models/transcription_model.py:110-217), predict (:219-266) and the evaluation loop's
F1 (scripts/evaluate.py:361-378)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import lib, check, ptr


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("music_transcription_amd ops run on the GPU only (got a CPU tensor)")


class _MaskedBCE(torch.autograd.Function):
    """weight * sum_valid bce / max(n_valid*P, 1); forward and d/dlogits come from ONE kernel pass."""

    @staticmethod
    def forward(ctx, logits, targets, lengths, weight):
        B, P, T = logits.shape
        x = logits.detach().contiguous().float()
        y = targets.detach().contiguous().float()
        if lengths is not None:
            len_host = lengths.detach().to("cpu", torch.int64)
            n_valid = int(torch.clamp(len_host, 0, T).sum())
            len_dev = len_host.to(x.device)
        else:
            n_valid, len_dev = B * T, None
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if logits.requires_grad else None
        ws = torch.empty(lib.mt_bce_workspace_bytes(), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.mt_bce_masked_fwd_bwd(ptr(x), ptr(y), ptr(len_dev), n_valid, float(weight), 0, ptr(loss), ptr(grad),
                                            ptr(ws), ws.numel(), B, P, T, _lib.stream_ptr()), "mt_bce_masked_fwd_bwd")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None


def masked_bce(logits, targets, lengths=None, weight: float = 1.0):
    _need_cuda(logits, targets)
    if logits.shape != targets.shape or logits.dim() != 3:
        raise ValueError(f"logits {tuple(logits.shape)} and targets {tuple(targets.shape)} must both be (B, 88, T)")
    return _MaskedBCE.apply(logits, targets, lengths, weight)


def onset_offset_targets(targets):
    _need_cuda(targets)
    y = targets.contiguous().float()
    on, off = torch.empty_like(y), torch.empty_like(y)
    with torch.cuda.device(y.device):
        check(lib.mt_onset_offset_targets(ptr(y), ptr(on), ptr(off), y.numel() // y.shape[-1], y.shape[-1], _lib.stream_ptr()),
              "mt_onset_offset_targets")
    return on, off


def compute_loss(logits, targets, lengths=None):
    """Single-head or dict (frame/onset/offset, weights .5/.25/.25) masked BCE."""
    if isinstance(logits, dict):
        on, off = onset_offset_targets(targets)
        return (masked_bce(logits["frame"], targets, lengths, 0.5) + masked_bce(logits["onset"], on, lengths, 0.25)
                + masked_bce(logits["offset"], off, lengths, 0.25))
    if logits.shape[-1] != targets.shape[-1]:
        raise NotImplementedError("time-axis interpolation (transcription_model.py:140-142) is never reached on the "
                                  "CNN-RNN path (logits and targets share T) and is not implemented")
    return masked_bce(logits, targets, lengths, 1.0)


def predict_from_logits(logits, threshold: float = 0.5):
    _need_cuda(logits)
    x = logits.detach().contiguous().float()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib.mt_predict_threshold(ptr(x), ptr(out), x.numel(), float(threshold), _lib.stream_ptr()), "mt_predict_threshold")
    return out


def f1_counts(pred, target, lengths: Optional[torch.Tensor] = None):
    """(B, 3) int64 {TP, FP, FN} over each sample's valid frames."""
    _need_cuda(pred, target)
    p, t = pred.contiguous().float(), target.contiguous().float()
    B, P, T = p.shape
    ld = None if lengths is None else lengths.to(p.device, torch.int64).contiguous()
    counts = torch.empty(B, 3, dtype=torch.int64, device=p.device)
    with torch.cuda.device(p.device):
        check(lib.mt_f1_counts(ptr(p), ptr(t), ptr(ld), ptr(counts), B, P, T, _lib.stream_ptr()), "mt_f1_counts")
    return counts


def framewise_f1(pred, target, lengths=None):
    """Per-sample binary F1 with zero_division=0 (evaluate.py:369-373) -> list of floats."""
    c = f1_counts(pred, target, lengths).cpu().tolist()
    return [0.0 if (2 * tp + fp + fn) == 0 else 2.0 * tp / (2 * tp + fp + fn) for tp, fp, fn in c]


def mean_f1(pred, target, lengths=None) -> float:
    v = framewise_f1(pred, target, lengths)
    return float(sum(v) / len(v)) if v else 0.0
