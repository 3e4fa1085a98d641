"""Log-mel frontend on the GPU: the drop-in for `audio_to_mel` (reference main.py:103-130,
data/dataset.py:155-156,:195-196) plus the batched device variant used by the fast path.
All arithmetic runs in csrc/mel.hip (mt_mel_db_f32); torch only owns the buffers."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ptr

SR = 16000
N_MELS = 320
HOP_LENGTH = 512
N_FFT = 2048


def num_frames(n_samples: int, hop_length: int = HOP_LENGTH) -> int:
    return int(lib.mt_mel_num_frames(int(n_samples), int(hop_length)))


def mel_filterbank(sr: int = SR, n_mels: int = N_MELS) -> np.ndarray:
    """Dense (n_mels, 1025) Slaney filterbank as the kernel's tables are built from (host only)."""
    fb = np.empty((n_mels, N_FFT // 2 + 1), dtype=np.float32)
    check(lib.mt_mel_filterbank_host(fb.ctypes.data, sr, n_mels), "mt_mel_filterbank_host")
    return fb


class MelFrontend:
    """Device-resident tables for one (sr, hop, n_mels) + the kernel launcher."""

    def __init__(self, sr: int = SR, n_mels: int = N_MELS, hop_length: int = HOP_LENGTH, device="cuda"):
        self.sr, self.n_mels, self.hop = int(sr), int(n_mels), int(hop_length)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MelFrontend runs on the GPU only (no CPU fallback in the product path)")
        nbytes = lib.mt_mel_plan_bytes(self.n_mels)
        self.plan = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.desc = _lib.MelDesc()
        with torch.cuda.device(self.device):
            check(lib.mt_mel_plan_init(ptr(self.plan), nbytes, self.sr, self.hop, self.n_mels, self.desc, _lib.stream_ptr()),
                  "mt_mel_plan_init")

    def __call__(self, wave: torch.Tensor, clamp: bool = True, out=None, chunk_max=None):
        """wave (B, N) float32 on the device -> (mel (B, 1, n_mels, T) float32 dB, chunk_max_power (B,)).
        clamp=False leaves the top_db clamp to the consumer (conv1 applies it on load)."""
        if wave.dim() == 1:
            wave = wave[None]
        if not wave.is_cuda:
            raise RuntimeError("MelFrontend expects a CUDA tensor")
        wave = wave.contiguous().float()
        B, N = wave.shape
        T = num_frames(N, self.hop)
        if out is None:
            out = torch.empty(B, 1, self.n_mels, T, dtype=torch.float32, device=wave.device)
        if chunk_max is None:
            chunk_max = torch.empty(B, dtype=torch.float32, device=wave.device)
        with torch.cuda.device(wave.device):
            check(lib.mt_mel_db_f32(ptr(self.plan), self.desc, ptr(wave), B, N, ptr(out), ptr(chunk_max),
                                    1 if clamp else 0, _lib.stream_ptr()), "mt_mel_db_f32")
        return out, chunk_max


_frontends = {}


def get_frontend(sr=SR, n_mels=N_MELS, hop_length=HOP_LENGTH, device="cuda") -> MelFrontend:
    key = (int(sr), int(n_mels), int(hop_length), str(torch.device(device)))
    if key not in _frontends:
        _frontends[key] = MelFrontend(sr, n_mels, hop_length, device)
    return _frontends[key]


def audio_to_mel(audio_chunk, sr: int = SR, n_mels: int = N_MELS, hop_length: int = HOP_LENGTH, device="cuda"):
    """Drop-in for main.py:103-130: (N,) samples -> (1, 1, n_mels, T) float32 CPU tensor."""
    wave = torch.as_tensor(np.asarray(audio_chunk, dtype=np.float32)).to(device)
    mel, _ = get_frontend(sr, n_mels, hop_length, device)(wave[None], clamp=True)
    return mel.cpu()
