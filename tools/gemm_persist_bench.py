"""Persistent-tile projection GEMM (csrc/gemm.hip, gemm256p_kernel) against the one-tile-per-workgroup kernel:
bit-exact comparison and timing at the model's projection shapes (f16 operands, f16 gx output).
    python tools/gemm_persist_bench.py [B ...]        (default B = 32 128)
MT_GEMM_PARK / MT_GEMM_TPW select the kernel variant / tiles per workgroup (read once per process)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from music_transcription_amd._lib import lib, check, ptr, stream_ptr, DT_F16, GX_F16

torch.manual_seed(0)
T, H = 938, 512
dev = "cuda"


def run(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in [int(v) for v in sys.argv[1:]] or [32, 128]:
    M = B * T
    Mp = (M + 255) // 256 * 256
    sched = torch.zeros(64, dtype=torch.uint8, device=dev)
    st = stream_ptr()
    gx_bytes = lib.mt_lstm_gx_bytes(B, T, H)
    for K in (1024, 5120):
        X = (torch.relu(torch.rand(Mp, K, device=dev) * 2 - 1) * 0.3).half()
        W = ((torch.rand(8 * H, K, device=dev) * 2 - 1) * 0.03).half()
        bias = torch.randn(8 * H, device=dev) * 0.1
        g0 = torch.full((gx_bytes // 2,), float("nan"), dtype=torch.float16, device=dev)
        g1 = torch.full((gx_bytes // 2,), float("nan"), dtype=torch.float16, device=dev)
        f0 = lambda: check(lib.mt_gemm_lstm_gx_dt(ptr(X), K, ptr(W), K, ptr(bias), ptr(g0), B, T, H, K, DT_F16 | GX_F16, st))
        f1 = lambda: check(lib.mt_gemm_lstm_gx_sched(ptr(X), K, ptr(W), K, ptr(bias), ptr(g1), B, T, H, K, DT_F16 | GX_F16, ptr(sched), st))
        t0, t1 = run(f0), run(f1)
        t0b, t1b = run(f0), run(f1)
        same = bool(torch.equal(g0.view(torch.int16), g1.view(torch.int16)))
        fl = 2.0 * M * 8 * H * K
        print(f"B={B} K={K} rows: one-tile {min(t0, t0b):.3f} ms ({fl / min(t0, t0b) / 1e9:.0f} TF/s)  persistent {min(t1, t1b):.3f} ms "
              f"({fl / min(t1, t1b) / 1e9:.0f} TF/s)  bit-identical={same} nan={bool(torch.isnan(g1).any())}", flush=True)
        del X, W, g0, g1
    # A read straight from hx images (layers > 0): Hprev = 512 -> K = 1024
    hx_bytes = lib.mt_lstm_hx_bytes(B, T, H)
    hx = (torch.rand(hx_bytes // 2, device=dev) * 2 - 1).half()
    W = ((torch.rand(8 * H, 2 * H, device=dev) * 2 - 1) * 0.03).half()
    bias = torch.randn(8 * H, device=dev) * 0.1
    g0 = torch.full((gx_bytes // 2,), float("nan"), dtype=torch.float16, device=dev)
    g1 = torch.full((gx_bytes // 2,), float("nan"), dtype=torch.float16, device=dev)
    f0 = lambda: check(lib.mt_gemm_lstm_gx_from_hx_ex(ptr(hx), ptr(W), 2 * H, ptr(bias), ptr(g0), B, T, H, H, 1, st))
    f1 = lambda: check(lib.mt_gemm_lstm_gx_from_hx_sched(ptr(hx), ptr(W), 2 * H, ptr(bias), ptr(g1), B, T, H, H, 1, ptr(sched), st))
    t0, t1 = run(f0), run(f1)
    t0b, t1b = run(f0), run(f1)
    same = bool(torch.equal(g0.view(torch.int16), g1.view(torch.int16)))
    fl = 2.0 * M * 8 * H * 2 * H
    print(f"B={B} K=1024 from hx: one-tile {min(t0, t0b):.3f} ms ({fl / min(t0, t0b) / 1e9:.0f} TF/s)  persistent {min(t1, t1b):.3f} ms "
          f"({fl / min(t1, t1b) / 1e9:.0f} TF/s)  bit-identical={same} nan={bool(torch.isnan(g1).any())}", flush=True)
    del hx, W, g0, g1
