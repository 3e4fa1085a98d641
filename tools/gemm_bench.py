"""Time the bf16 GEMM entry points at the model's projection shapes (B = 32, T = 938)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_transcription_amd._lib import lib, check, ptr, stream_ptr
torch.manual_seed(0)
for (M, N, K, name) in ((30016, 4096, 5120, "l0"), (30016, 4096, 1024, "l1"), (8192, 8192, 8192, "8k")):
    Mp = (M + 255) // 256 * 256
    A = (torch.rand(Mp, K, device="cuda") * 2 - 1).bfloat16()
    W = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    C = torch.empty(M, N, device="cuda")
    st = stream_ptr()
    for _ in range(3):
        check(lib.mt_gemm_bf16_f32acc(ptr(A), K, ptr(W), K, None, ptr(C), N, M, N, K, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        check(lib.mt_gemm_bf16_f32acc(ptr(A), K, ptr(W), K, None, ptr(C), N, M, N, K, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: M={M} N={N} K={K}  {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s")

# the model's own entry point (gx epilogue), on uniform and on post-ReLU-like operands
B, T, H = 32, 938, 512
for K in (5120, 1024):
    for kind in ("uniform", "relu"):
        M = B * T
        Mp = (M + 255) // 256 * 256
        X = torch.rand(Mp, K, device="cuda") * 2 - 1
        if kind == "relu":
            X = torch.relu(X) * 0.3
        X = X.bfloat16()
        W = ((torch.rand(8 * H, K, device="cuda") * 2 - 1) * 0.03).bfloat16()
        bias = torch.zeros(8 * H, device="cuda")
        gx = torch.empty(M * 8 * H, device="cuda")
        st = stream_ptr()
        for _ in range(3):
            check(lib.mt_gemm_lstm_gx(ptr(X), K, ptr(W), K, ptr(bias), ptr(gx), B, T, H, K, st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            check(lib.mt_gemm_lstm_gx(ptr(X), K, ptr(W), K, ptr(bias), ptr(gx), B, T, H, K, st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"gx K={K} {kind}: {ms:.3f} ms  {2.0 * M * 8 * H * K / ms / 1e9:.0f} TF/s")
