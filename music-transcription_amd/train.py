"""Training loop of the reference over the HIP training step.

`train_one_epoch` / `evaluate` keep the reference's signatures and control flow
(train/train_transcriber.py:90-158,:161-191): NaN/Inf-loss batches are skipped, more than 10 raise, a
non-finite gradient norm skips the optimizer step, the epoch returns (mean loss, step losses).  Differences, all
on the device side: no GradScaler (bf16 operands with f32 accumulation need no loss scaling), clip + Adam are
ONE fused pass over a flat buffer (optim.FusedAdamClip), and with torch.distributed initialised the flat
gradient is all-reduced (mean) over RCCL before the clip -- the data-parallel step of BASELINE config 4.
"""
from __future__ import annotations

import math
import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist

from .optim import EarlyBucket, FusedAdamClip, flatten_parameters


def _world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def make_optimizer(model, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-5,
                   max_grad_norm: float = 1.0) -> FusedAdamClip:
    """optim.Adam(model.parameters(), lr, eps=1e-8, weight_decay=1e-5) of scripts/train_cnn.py:290, fused with
    clip_grad_norm_ (train_transcriber.py:134).  Parameters become views of one flat buffer."""
    flat, grads = flatten_parameters(model.parameters())
    opt = FusedAdamClip(flat, grads, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_norm=max_grad_norm)
    opt.net = getattr(model, "model", model)       # the training step records there which parameters got no gradient (optim._keep_ranges)
    if _world_size() == 1 and os.environ.get("MT_DIRECT_GRADS", "1") != "0":
        # single GPU: the step's backward pass writes gradients straight into the flat buffer (optim.FusedAdamClip.make_grad_target)
        opt.net._grad_target = opt.make_grad_target(dict(opt.net.named_parameters()))
    if _world_size() > 1 and os.environ.get("MT_DP_EARLY_BUCKET", "1") != "0":
        # data parallel: the gradients of the upper LSTM layers and the fc are all-reduced under the rest of the backward pass
        net = getattr(model, "model", model)
        by_id = {id(p): (p, o, k) for p, o, k in grads._mt_views}
        named = {n: by_id[id(p)] for n, p in net.named_parameters() if id(p) in by_id}
        opt.early = EarlyBucket(grads, named)
        net._grad_sync = opt.early
    return opt


def train_one_epoch(model, dataloader: Iterable, optimizer: FusedAdamClip, device, max_grad_norm: float = 1.0,
                    log=None) -> Tuple[float, List[float]]:
    if not isinstance(optimizer, FusedAdamClip):
        raise TypeError("train_one_epoch drives the fused HIP optimizer: build it with make_optimizer(model, ...)")
    model.train()
    optimizer.max_norm = float(max_grad_norm)
    total, step_losses, nan_count, n_batches = 0.0, [], 0, 0
    net0 = getattr(model, "model", model)
    tuner = None
    if type(net0).__name__ in ("CNNRNNModel", "CNNRNNModelLarge"):   # the step's side streams are chosen by measurement over the first dozen steps
        from .train_step_large import SideStreamTuner
        tuner = getattr(net0, "_side_stream_tuner", None)
        if tuner is None:
            tuner = net0._side_stream_tuner = SideStreamTuner(device)
    for batch in dataloader:
        n_batches += 1
        if tuner is not None:
            tuner.step_begin()
        optimizer.zero_grad()
        mel, roll, lengths = batch
        mel, roll = mel.to(device, non_blocking=True), roll.to(device, non_blocking=True)
        logits = model(mel)
        loss = model.compute_loss(logits, roll, lengths)
        step_loss = float(loss.item())
        bad = math.isnan(step_loss) or math.isinf(step_loss)
        if _world_size() > 1:
            # Data parallel: the skip must be COLLECTIVE.  optimizer.step() holds the gradient all-reduce, so a rank that
            # skipped alone would leave the others waiting in it: every rank skips when any rank's loss is non-finite.
            flag = torch.tensor([1.0 if bad else 0.0], device=loss.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            bad = bool(flag.item() > 0.0)
        if bad:
            nan_count += 1
            if nan_count > 10:
                raise RuntimeError("Too many NaN losses - training unstable!")
            continue
        loss.backward()
        stats = optimizer.step().tolist()              # {grad norm before clipping, 1.0 if the step was taken}
        if tuner is not None:
            tuner.step_end()
        net = getattr(model, "model", model)
        if hasattr(net, "raise_on_train_handoff_timeout"):
            net.raise_on_train_handoff_timeout()       # (the .tolist() above synchronised with the whole step)
        if stats[1] == 0.0:                            # non-finite gradient norm (identical on every rank: it is taken after
            nan_count += 1                             # the all-reduce): step skipped on the device, Adam's step count not advanced
            optimizer.unskip()
            continue
        total += step_loss
        step_losses.append(step_loss)
        if log is not None:
            log(len(step_losses), step_loss, stats[0])
    avg = total / n_batches if step_losses else float("nan")
    return avg, step_losses


@torch.no_grad()
def evaluate(model, dataloader: Iterable, device) -> float:
    """Mean validation loss (train_transcriber.py:161-191)."""
    model.eval()
    total, n = 0.0, 0
    for mel, roll, lengths in dataloader:
        logits = model(mel.to(device, non_blocking=True))
        total += float(model.compute_loss(logits, roll.to(device, non_blocking=True), lengths).item())
        n += 1
    return total / max(n, 1)
