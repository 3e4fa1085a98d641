"""Time mt_mel_db_f32 alone (B = 32 chunks of 30 s) and check it against the oracle on two chunks."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import music_transcription_amd as mta
from oracle import frontend_ref as FR
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
base = FR.synth_audio(2, 480000, seed=1)
wave = torch.from_numpy(np.concatenate([base] * (B // 2 + 1))[:B].copy()).cuda()
fe = mta.MelFrontend(16000, 320, 512, "cuda")
mel, cm = fe(wave, clamp=True)
ref = FR.audio_to_mel_batch(base)
d = np.abs(mel[:2].cpu().numpy() - ref)
print("max|d| dB", d.max(), "mean", d.mean())
for clamp in (False, True):
    for _ in range(3): fe(wave, clamp=clamp, out=mel, chunk_max=cm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): fe(wave, clamp=clamp, out=mel, chunk_max=cm)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    gb = B * (4 * 480000 + 4 * 320 * 938) / 1e9
    print(f"clamp={clamp}: {ms*1e3:.1f} us per launch, {gb/ms*1e3:.0f} GB/s algorithmic, {B/ms*1e3:.0f} chunks/s")
