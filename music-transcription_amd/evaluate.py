"""Evaluation on the GPU: the hot loops of scripts/evaluate.py restructured.

  * `evaluate_dataset` = run_evaluation's headless loop (evaluate.py:335-379): per-sample framewise F1 over the valid
    frames (zero_division = 0), unweighted mean.  The reference runs batch 1; samples of EQUAL length are batched
    here (no padding arises, so results are per-sample identical) and the F1 counts are integer sums on the device.
  * `tune_threshold` = run_threshold_tuning (evaluate.py:556-618) with the same coarse-to-fine schedule, but the
    model runs ONCE: logits stay on the device and every candidate threshold is an integer-count pass
    (mt_f1_sweep_counts, up to 16 thresholds per pass).
  * recordings / chunks shard over ranks with no data-path collective (parallel.py); per-sample F1 values are
    gathered with one small all-reduce.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import lib, check, ptr
from .parallel import gather_values, shard_range


def _f1(tp, fp, fn) -> float:
    d = 2 * tp + fp + fn
    return 0.0 if d == 0 else 2.0 * tp / d


@torch.no_grad()
def collect_logits(model, dataset, indices: Sequence[int], device="cuda", max_batch: int = 128):
    """Forward every sample once; returns [(index, logits (88, T) on device, roll (88, T) on device)]."""
    by_len = defaultdict(list)
    for i in indices:
        mel, roll = dataset[i]
        by_len[int(mel.shape[-1])].append((i, mel, roll))
    out = []
    for T, items in by_len.items():
        for s in range(0, len(items), max_batch):
            grp = items[s:s + max_batch]
            mel = torch.stack([m for _, m, _ in grp]).to(device)                    # (b, 1, n_mels, T): equal T, no padding
            logits = model(mel)
            for (i, _, roll), lg in zip(grp, logits):
                out.append((i, lg.contiguous(), roll.to(device).float().contiguous()))
    net = getattr(model, "model", model)
    if hasattr(net, "raise_on_handoff_timeout"):
        net.raise_on_handoff_timeout(sync=True)        # a timed-out recurrence would have left NaN logits: fail loudly, once per pass
    out.sort(key=lambda x: x[0])
    return out


def f1_at_thresholds(logits_rolls, thresholds: Sequence[float]) -> np.ndarray:
    """mean-over-samples F1 for each threshold; one counts pass per 16 thresholds per sample length group."""
    thresholds = [float(t) for t in thresholds]
    per_sample = np.zeros((len(logits_rolls), len(thresholds)))
    by_len = defaultdict(list)
    for n, (_, lg, roll) in enumerate(logits_rolls):
        by_len[lg.shape[-1]].append(n)
    for T, idxs in by_len.items():
        lg = torch.stack([logits_rolls[n][1] for n in idxs])
        rl = torch.stack([logits_rolls[n][2] for n in idxs])
        B, P, _ = lg.shape
        for k0 in range(0, len(thresholds), 16):
            th = torch.tensor(thresholds[k0:k0 + 16], dtype=torch.float32, device=lg.device)
            K = th.numel()
            counts = torch.empty(B, K, 3, dtype=torch.int64, device=lg.device)
            with torch.cuda.device(lg.device):
                check(lib.mt_f1_sweep_counts(ptr(lg), ptr(rl), None, ptr(th), K, ptr(counts), B, P, T, _lib.stream_ptr()), "mt_f1_sweep_counts")
            c = counts.cpu().numpy()
            for bi, n in enumerate(idxs):
                for k in range(K):
                    per_sample[n, k0 + k] = _f1(*[int(v) for v in c[bi, k]])
    return per_sample


def evaluate_dataset(model, dataset, threshold: float = 0.5, device="cuda", subset: Optional[int] = None,
                     rank: int = 0, world: int = 1) -> Tuple[float, List[float]]:
    """-> (mean F1 over ALL samples, per-sample F1 list), identical on every rank."""
    n = len(dataset) if subset is None else min(subset, len(dataset))
    mine = list(shard_range(n, rank, world))
    lr = collect_logits(model, dataset, mine, device)
    vals = f1_at_thresholds(lr, [threshold])[:, 0] if lr else np.zeros(0)
    allv = gather_values(mine, vals.tolist(), n)
    return (float(np.mean(allv)) if allv else 0.0), allv


def tune_threshold(model, dataset, device="cuda", subset: Optional[int] = None, tune_range=(0.05, 0.95), tune_step=0.1,
                   tune_min_step=0.01, tune_rounds=6, rank: int = 0, world: int = 1, log=print):
    """Coarse-to-fine search of evaluate.py:556-618 (same candidate grids, same strict-improvement rule, same window
    and stopping rule); returns (best_threshold, best_mean_f1)."""
    n = len(dataset) if subset is None else min(subset, len(dataset))
    mine = list(shard_range(n, rank, world))
    lr = collect_logits(model, dataset, mine, device)          # the only forward passes
    tune_min, tune_max = tune_range
    step = tune_step
    best_t, best_f1 = 0.5, -1.0
    for rnd in range(1, tune_rounds + 1):
        ths = np.arange(tune_min, tune_max + step / 2, step)
        local = f1_at_thresholds(lr, ths) if lr else np.zeros((0, len(ths)))
        means = []
        for k in range(len(ths)):
            allv = gather_values(mine, local[:, k].tolist(), n)
            means.append(float(np.mean(allv)) if allv else 0.0)
        rb_t, rb_f = best_t, best_f1
        for t, f in zip(ths, means):
            if f > rb_f:
                rb_f, rb_t = f, float(t)
        best_t, best_f1 = rb_t, rb_f
        if log:
            log(f"=== Round {rnd}/{tune_rounds} | range=[{tune_min:.4f}, {tune_max:.4f}] step={step:.4f} -> t={best_t:.4f} f1={best_f1:.6f}")
        tune_min = max(0.01, best_t - 2 * step)
        tune_max = min(0.99, best_t + 2 * step)
        step = step / 2
        if step < tune_min_step:
            break
    return best_t, best_f1
