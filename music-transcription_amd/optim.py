"""Fused clip + Adam step and the data-parallel gradient reduction of the training loop:
reference semantics = clip_grad_norm_(1.0) then torch.optim.Adam(lr, eps=1e-8, weight_decay=1e-5) with a
NaN-norm skip (train/train_transcriber.py:130-150, scripts/train_cnn.py:290).

All parameters live in ONE flat fp32 buffer (parameters become views of it), so the step is two kernel
launches and the data-parallel reduction is ONE RCCL all-reduce of the flat gradient."""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist

from . import _lib
from ._lib import lib, check, ptr


# Bumped by every in-place update of parameter memory that bypasses torch's version counters (the fused optimizer
# writes through raw pointers): model._param_signature() includes it, so packed inference weights are rebuilt.
WEIGHTS_EPOCH = [0]


def flatten_parameters(params: Iterable[torch.nn.Parameter]):
    """Re-home parameters as views of one flat fp32 tensor; returns (flat_params, flat_grads) with p.grad views."""
    params = [p for p in params if p.requires_grad]
    n = sum(p.numel() for p in params)
    dev = params[0].device
    flat = torch.empty(n, dtype=torch.float32, device=dev)
    grads = torch.zeros(n, dtype=torch.float32, device=dev)
    o = 0
    views = []
    for p in params:
        k = p.numel()
        flat[o:o + k].copy_(p.detach().reshape(-1))
        p.data = flat[o:o + k].view_as(p)
        p.grad = grads[o:o + k].view_as(p)
        views.append((p, o, k))
        o += k
    grads._mt_views = views          # FusedAdamClip.step() re-attaches p.grad views a caller has detached
    return flat, grads


def _world() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_mean_(flat_grads: torch.Tensor) -> torch.Tensor:
    """Gradient mean over ranks: one all-reduce of the flat buffer (RCCL on GPUs, gloo on CPU tensors).  (Host-side helper
    and what the CPU tests check; FusedAdamClip.step() all-reduces the SUM and folds 1 / world into the fused kernel.)"""
    if _world() > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
        flat_grads.div_(dist.get_world_size())
    return flat_grads


class EarlyBucket:
    """Data-parallel overlap of the gradient all-reduce with the backward pass (the reference has none: single GPU).  The
    training step's backward produces the gradients top-down -- fc, then the LSTM layers L-1 .. 1, then layer 0 and the
    convolutions -- and parameters sit in the flat buffer in model order, so the gradients of [LSTM layers >= 1, fc] are one
    contiguous TAIL of the flat gradient, complete while layer 0's backward recurrence (the longest kernel of the step), its
    84-MB weight-gradient GEMM and the convolution backward still run.  `reduce_early` folds those gradients into the flat buffer
    on the stream that produced them and starts their all-reduce on a communication stream; FusedAdamClip.step() then reduces
    only the head of the buffer and waits for the tail's.  Used by train_step.backward_train when the optimizer attached one
    to the model (world size > 1).

    Contract: ONE backward per optimizer step, with optimizer.zero_grad() (or model.zero_grad()) in front of it, as the
    reference's loop does (train_transcriber.py:113-144).  A second backward before step() (gradient accumulation) is not
    taken over: its gradients go through autograd as usual and step() raises, because the tail would then hold a
    rank-summed microbatch plus a local one."""

    def __init__(self, flat_grads: torch.Tensor, named_views):
        self.g = flat_grads
        self.where = {name: (o, k, p) for name, (p, o, k) in named_views.items()}
        self.comm = torch.cuda.Stream(device=flat_grads.device)
        self.pending = None                       # (start, end, work, event) of the tail reduce in flight
        self.early_params = set()
        self.stale = False                        # a backward ran while a tail reduce was pending

    def reduce_early(self, grads: dict, stream) -> set:
        """grads: name -> gradient tensor (complete in `stream`'s order).  Returns the names taken over (their gradient now lives
        in the flat buffer; autograd must be given None for them).  Nothing is taken unless the names form the buffer's tail."""
        if _world() <= 1:
            return set()
        if self.pending is not None:
            self.stale = True                     # a second backward before step(): see the class docstring
            return set()
        items = [(self.where[n], t) for n, t in grads.items() if n in self.where and t is not None]
        if not items:
            return set()
        start = min(o for (o, k, p), _ in items)
        if sum(k for (o, k, p), _ in items) != self.g.numel() - start:
            return set()                          # not a contiguous tail of the flat buffer: leave everything to step()
        with torch.cuda.stream(stream):
            for (o, k, p), t in items:
                if p.grad is None:                # zero_grad(set_to_none=True) detached the view: the tail may hold last
                    self.g[o:o + k].copy_(t.reshape(-1))       # step's reduced gradient -> overwrite, never add
                else:
                    self.g[o:o + k].add_(t.reshape(-1))        # accumulate, as autograd would have
            ev = torch.cuda.Event()
            ev.record(stream)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ev)
            work = dist.all_reduce(self.g[start:], op=dist.ReduceOp.SUM, async_op=True)
            done = torch.cuda.Event()
            done.record(self.comm)
        self.pending = (start, work, done)
        self.early_params = {id(p) for (o, k, p), _ in items}
        return {n for n in grads if n in self.where and grads[n] is not None}

    def finish(self, stream) -> int:
        """Wait (in `stream`'s order) for the tail reduce; returns where the un-reduced head ends (numel if nothing was early)."""
        if self.stale:
            self.stale = False
            raise RuntimeError("EarlyBucket: two backward passes before one optimizer.step() -- the early all-reduce of the "
                               "gradient tail supports one backward per step (set MT_DP_EARLY_BUCKET=0 for gradient accumulation)")
        if self.pending is None:
            return self.g.numel()
        start, work, done = self.pending
        work.wait()
        stream.wait_event(done)
        self.pending = None
        return start


def note_params_without_grad(net, no_grad) -> None:
    """Called by a training step's backward pass with the names of the parameters it produced NO gradient for.  A parameter is
    left out of the next optimizer step only if no backward pass since zero_grad() / step() reached it (torch.optim.Adam skips
    `grad is None`, and a second backward that does reach the parameter makes its grad a tensor): the intersection over the
    passes, not the union.  `net._params_without_grad is None` = no backward pass yet."""
    prev = getattr(net, "_params_without_grad", None)
    net._params_without_grad = set(no_grad) if prev is None else (set(prev) & set(no_grad))


class FusedAdamClip:
    """clip_grad_norm_ + Adam (coupled L2) in libmt_hip.so over flat buffers.  step() returns a (2,) device tensor
    {grad norm before clipping, 1.0 if the step was taken / 0.0 if skipped for a non-finite norm}: no host sync."""

    def __init__(self, flat_params: torch.Tensor, flat_grads: torch.Tensor, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-5, max_norm=1.0):
        if not flat_params.is_cuda:
            raise RuntimeError("FusedAdamClip runs on the GPU only")
        self.p, self.g = flat_params, flat_grads
        self.m, self.v = torch.zeros_like(flat_params), torch.zeros_like(flat_params)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.t = 0                    # steps actually taken (a skipped step does not advance it: unskip())
        self._views = getattr(flat_grads, "_mt_views", None)
        self.ws = torch.empty(lib.mt_adam_workspace_bytes(), dtype=torch.uint8, device=flat_params.device)
        self.stats = torch.zeros(2, dtype=torch.float32, device=flat_params.device)
        self.early = None                 # EarlyBucket, attached by train.make_optimizer for data-parallel runs
        # torch.optim.Adam and clip_grad_norm_ skip parameters whose .grad is None: after the loop's zero_grad() those are the
        # parameters the backward pass did not reach (the onset / offset heads under the reference's frame-only loss,
        # train_transcriber.py:119).  Here p.grad is always a view of the flat buffer, so the training step's autograd function
        # records which parameters it returned no gradient for (`net._params_without_grad`, names) and the fused kernel gets the
        # flat ranges of all the others (keep ranges).  `net` is attached by train.make_optimizer.
        self.net = None
        self.skip_untouched = True
        self._touched = set()             # parameters whose flat-gradient view a backward pass wrote directly since zero_grad()

    def zero_grad(self):
        self.g.zero_()
        self._touched = set()
        if self.net is not None:
            self.net._params_without_grad = None        # None = no backward pass since (the training step intersects)

    def make_grad_target(self, named_params):
        """-> target(name, shape): the flat-gradient view of parameter `name` for a training step's backward pass to WRITE its gradient
        into (the step then hands autograd None for it), or None when the gradient must go through autograd's accumulation instead:
        the parameter's .grad no longer is its view of the flat buffer (a caller detached it), or something was already written for
        it since zero_grad() (a second backward pass before step(): gradients accumulate, as in torch).  A view is handed out once
        per zero_grad()."""
        off = {id(p): (o, k) for p, o, k in (self._views or [])}

        def target(name, shape):
            p = named_params.get(name)
            if p is None or id(p) not in off or id(p) in self._touched:
                return None
            o, k = off[id(p)]
            n = 1
            for v in shape:
                n *= int(v)
            if n != k or p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * o:
                return None
            self._touched.add(id(p))
            return self.g[o:o + k].view(*[int(v) for v in shape])
        return target

    def _keep_ranges(self):
        """Ascending merged [lo, hi) ranges of the parameters that take part in this step: all of them, minus those the last
        backward pass produced no gradient for; None = everything."""
        skip_names = getattr(self.net, "_params_without_grad", None) if self.net is not None else None
        if not self._views or not self.skip_untouched or not skip_names:
            return None
        skip = {id(p) for n, p in self.net.named_parameters() if n in skip_names}
        out = []
        for p, o, k in self._views:
            if id(p) not in skip:
                if out and out[-1][1] == o:
                    out[-1][1] = o + k
                else:
                    out.append([o, o + k])
        return out

    def _reattach_grad_views(self):
        """`model.zero_grad()` (set_to_none=True, as the reference's loop calls it) or an assignment to p.grad detaches a
        parameter from the flat gradient buffer; autograd then accumulates into a fresh tensor the fused step would never
        see.  Fold such gradients back into the flat buffer and restore the views."""
        if not self._views:
            return
        base = self.g.data_ptr()
        for p, o, k in self._views:
            view = self.g[o:o + k].view_as(p)
            if p.grad is None and self.early is not None and id(p) in self.early.early_params:
                pass                                  # its gradient already sits (reduced) in the flat buffer
            elif p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != base + 4 * o:
                view.copy_(p.grad)
            else:
                continue
            p.grad = view

    def unskip(self):
        """The device skipped the last step (non-finite gradient norm, stats[1] == 0): torch's Adam would not have been
        called at all (train/train_transcriber.py:137-142), so its step count must not advance either."""
        self.t = max(0, self.t - 1)

    def step(self, sync_grads: bool = True):
        """One clip + Adam step.  With torch.distributed initialised (and sync_grads) the flat gradient is all-reduced as a
        SUM and the 1 / world of the mean is applied inside the fused kernel (no extra pass over the buffer): after step()
        the flat gradient buffer holds the rank SUM, not the mean."""
        import ctypes as C
        self._reattach_grad_views()
        keep = self._keep_ranges()
        scale = 1.0
        if sync_grads and _world() > 1:
            head = self.g.numel()
            if self.early is not None:
                head = self.early.finish(torch.cuda.current_stream(self.g.device))      # the tail was reduced under the backward pass
                self.early.early_params = set()
            if head > 0:
                dist.all_reduce(self.g[:head], op=dist.ReduceOp.SUM)
            scale = 1.0 / _world()
        if keep is None:
            kr, nk = None, 0
        else:
            flat_ranges = [v for r in keep for v in r]
            kr, nk = (C.c_longlong * len(flat_ranges))(*flat_ranges), len(keep)
        self.t += 1
        with torch.cuda.device(self.p.device):
            check(lib.mt_adam_clip_step_ex(ptr(self.p), ptr(self.g), ptr(self.m), ptr(self.v), self.p.numel(), self.lr, self.betas[0],
                                           self.betas[1], self.eps, self.wd, self.max_norm, self.t, scale, kr, nk, ptr(self.stats),
                                           ptr(self.ws), self.ws.numel(), _lib.stream_ptr()), "mt_adam_clip_step_ex")
        if self.net is not None:
            self.net._params_without_grad = None        # None = no backward pass since (the training step intersects)
        WEIGHTS_EPOCH[0] += 1
        return self.stats
