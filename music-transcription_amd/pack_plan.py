"""Job tables for mt_pack_jobs (include/mt_hip.h): the padded / transposed / re-ordered operand copies of the f32 parameters that a training
step rebuilds after every optimizer update, described ONCE per parameter set and run as one launch per step.

The reference has no such step -- its parameters are used by torch's own kernels as they are (/root/reference/models/cnn_rnn_model.py:186-260,
/root/reference/train/train_transcriber.py:117-134); the layouts here are the operands of the HIP kernels of train_step_large.py."""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

JOB_DTYPE = np.dtype([("src", "u8"), ("src2", "u8"), ("dst", "u8"), ("ld", "i8"), ("sr1", "i8"), ("sr2", "i8"), ("sc1", "i8"), ("sc2", "i8"),
                      ("dt", "i4"), ("Rp", "i4"), ("Cp", "i4"), ("Rn2", "i4"), ("R1v", "i4"), ("R2v", "i4"), ("Cn2", "i4"), ("C1v", "i4"),
                      ("C2v", "i4"), ("tr", "i4"), ("tile0", "i4"), ("reserved", "i4")])
assert JOB_DTYPE.itemsize == 112          # sizeof(mt_pack_job)
_BIG = 1 << 30
_DT = {torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16, torch.float32: 2}

Level = Tuple[int, int, int, int, int]    # (n1 valid, n2 padded, n2 valid, stride1, stride2)


def one(valid: int, stride: int) -> Level:
    """index i < valid -> i * stride"""
    return (1, _BIG, valid, 0, stride)


def two(n1: int, n2_padded: int, n2_valid: int, stride1: int, stride2: int) -> Level:
    """index i = i1 * n2_padded + i2 with i1 < n1, i2 < n2_valid -> i1 * stride1 + i2 * stride2"""
    return (n1, n2_padded, n2_valid, stride1, stride2)


class PackPlan:
    def __init__(self, dev):
        self.dev = torch.device(dev)
        if self.dev.type == "cuda" and self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.jobs = []
        self.keep = []          # sources and destinations stay alive with the table that holds their addresses
        self.ntiles = 0
        self.table: Optional[torch.Tensor] = None

    def add(self, dst: torch.Tensor, src: torch.Tensor, rows: Level, cols: Level, *, src2: Optional[torch.Tensor] = None, base: int = 0,
            tr: bool = False, at: Sequence[int] = (0, 0), shape: Optional[Sequence[int]] = None):
        """dst: 2-D tensor with unit column stride; the job writes the rectangle `shape` (default: to the end of dst) at offset `at`.
        src (+ src2): contiguous f32 tensors on the device, read at element base + the offset the two index levels give."""
        assert self.table is None, "plan is final"
        assert dst.dim() == 2 and dst.stride(1) == 1 and dst.device == self.dev and dst.dtype in _DT
        Rp, Cp = shape if shape is not None else (dst.shape[0] - at[0], dst.shape[1] - at[1])
        assert 0 <= at[0] and 0 <= at[1] and at[0] + Rp <= dst.shape[0] and at[1] + Cp <= dst.shape[1] and Rp > 0 and Cp > 0
        for s_ in (src, src2):
            if s_ is not None and not (s_.is_cuda and s_.device == self.dev and s_.dtype == torch.float32 and s_.is_contiguous()):
                raise RuntimeError("pack plan: parameters must be contiguous f32 tensors on the training device")
        assert src2 is None or src2.numel() == src.numel()
        lo = hi = base
        for n1, n2p, n2v, s1, s2 in (rows, cols):       # every reachable source offset stays inside the tensor
            for n, s in ((n1, s1), (min(n2v, n2p), s2)):
                lo += min(0, (n - 1) * s)
                hi += max(0, (n - 1) * s)
        assert 0 <= lo and hi < src.numel(), (lo, hi, src.numel())
        ld = dst.stride(0)
        job = np.zeros((), dtype=JOB_DTYPE)
        job["src"] = src.data_ptr() + 4 * base
        job["src2"] = 0 if src2 is None else src2.data_ptr() + 4 * base
        job["dst"] = dst.data_ptr() + (at[0] * ld + at[1]) * dst.element_size()
        job["ld"], job["dt"], job["Rp"], job["Cp"], job["tr"] = ld, _DT[dst.dtype], Rp, Cp, int(bool(tr))
        job["R1v"], job["Rn2"], job["R2v"], job["sr1"], job["sr2"] = rows
        job["C1v"], job["Cn2"], job["C2v"], job["sc1"], job["sc2"] = cols
        job["tile0"] = self.ntiles
        self.ntiles += ((Rp + 31) // 32) * ((Cp + 31) // 32)
        self.jobs.append(job)
        self.keep += [dst, src] + ([src2] if src2 is not None else [])

    def finalize(self):
        if self.jobs:
            arr = np.stack(self.jobs)
            self.table = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(self.dev)
        return self

    def run(self):
        """One launch on the current stream of the plan's device."""
        if self.jobs:
            _lib.check(_lib.lib.mt_pack_jobs(_lib.ptr(self.table), len(self.jobs), self.ntiles, _lib.stream_ptr()), "mt_pack_jobs")
