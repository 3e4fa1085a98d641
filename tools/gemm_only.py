import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import music_transcription_amd as mta
from music_transcription_amd._lib import lib, check, ptr
B, T, H, K = 32, 938, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 5120
M = B * T; Mp = (M + 127) // 128 * 128
X = (torch.randn(Mp, K, device="cuda") * 0.5).bfloat16()
W = (torch.randn(8 * H, K, device="cuda") * 0.02).bfloat16()
bias = torch.zeros(8 * H, device="cuda")
g = torch.empty(lib.mt_lstm_gx_bytes(B, T, H) // 4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): check(lib.mt_gemm_lstm_gx(ptr(X), K, ptr(W), K, ptr(bias), ptr(g), B, T, H, K, st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): check(lib.mt_gemm_lstm_gx(ptr(X), K, ptr(W), K, ptr(bias), ptr(g), B, T, H, K, st))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"K={K}: {ms:.3f} ms  {2.0 * M * 8 * H * K / ms / 1e9:.0f} TFLOP/s")
