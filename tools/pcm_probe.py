"""Where the from-PCM corpus figure's time goes (GPU box): H2D rate from pinned memory, the polyphase resampler (tiled kernel against the
thread-per-output one: MT_RESAMPLE_TILED=0 in a second process), and their agreement.  Usage: python tools/pcm_probe.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from music_transcription_amd import transcribe as tr  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
rate = 44100
n = int(secs * rate)
g = torch.Generator(device="cuda").manual_seed(1)
pcm = (torch.randn(n, 2, device="cuda", generator=g).clamp_(-3, 3) * 8000).to(torch.int16)
host = torch.empty(n, 2, dtype=torch.int16, pin_memory=True)
host.copy_(pcm)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    d = host.to("cuda", non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"H2D {host.numel() * 2 / 1e6:.0f} MB pinned: {dt * 1e3:.2f} ms = {host.numel() * 2 / dt / 1e9:.1f} GB/s")
y = tr.resample_pcm_device(pcm, rate, 16000)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    y = tr.resample_pcm_device(pcm, rate, 16000)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 5
tiled = os.environ.get("MT_RESAMPLE_TILED", "1") != "0"
print(f"resample {secs:g} s of 44.1 kHz stereo int16 -> {y.numel()} samples, {'tiled' if tiled else 'thread-per-output'} kernel: {ms:.3f} ms "
      f"({y.numel() * 495 * 2 / ms / 1e9:.2f} TFLOP/s of filter arithmetic; 20 h would take {ms * 72000 / secs:.0f} ms)")
np.save(os.path.join(ROOT, "gpurun_out", f"pcm_probe_{'tiled' if tiled else 'plain'}.npy"), y[:2000000].cpu().numpy())
other = os.path.join(ROOT, "gpurun_out", f"pcm_probe_{'plain' if tiled else 'tiled'}.npy")
if os.path.exists(other):
    a, b = y[:2000000].cpu().numpy(), np.load(other)
    print(f"tiled vs thread-per-output kernel: max |d| {np.abs(a - b).max():.3g} (signal max {np.abs(b).max():.3g})")
