"""How much does a resident recurrence kernel slow the MFMA GEMM (and vice versa)?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import music_transcription_amd as mta
from music_transcription_amd._lib import lib, check, ptr
B, T, H, K = 32, 938, 512, 5120
M = B * T; Mp = (M + 127) // 128 * 128
X = (torch.randn(Mp, K, device="cuda") * 0.5).bfloat16()
W = (torch.randn(8 * H, K, device="cuda") * 0.02).bfloat16()
bias = torch.zeros(8 * H, device="cuda")
def bufs():
    return (torch.empty(lib.mt_lstm_gx_bytes(B, T, H) // 4, device="cuda"), torch.empty(lib.mt_lstm_hx_bytes(B, T, H) // 4, device="cuda"),
            torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device="cuda"))
whh = ((torch.rand(2, 4 * H, H, device="cuda") * 2 - 1) / np.sqrt(H)).contiguous()
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
g1, h1, y1 = bufs(); g2, h2, y2 = bufs(); g3, h3, y3 = bufs()
def gemm(st, g): check(lib.mt_gemm_lstm_gx(ptr(X), K, ptr(W), K, ptr(bias), ptr(g), B, T, H, K, st.cuda_stream))
def rec(st, g, h, y): check(lib.mt_lstm_bidir_fwd(ptr(g), ptr(whh), ptr(h), ptr(y), y.numel(), B, T, H, st.cuda_stream))
gemm(s1, g1); gemm(s2, g2); gemm(s3, g3); torch.cuda.synchronize()
def timeit(fn, st, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(n): fn()
        e1.record(st)
    return e0, e1, n
def report(tag, *ts):
    torch.cuda.synchronize()
    print(tag, " ".join(f"{e0.elapsed_time(e1) / n:.3f} ms" for e0, e1, n in ts))
report("gemm solo      :", timeit(lambda: gemm(s1, g1), s1, 10))
report("rec solo       :", timeit(lambda: rec(s2, g2, h2, y2), s2, 3))
report("gemm | rec     :", timeit(lambda: gemm(s1, g1), s1, 20), timeit(lambda: rec(s2, g2, h2, y2), s2, 6))
report("gemm | rec rec :", timeit(lambda: gemm(s1, g1), s1, 30), timeit(lambda: rec(s2, g2, h2, y2), s2, 6), timeit(lambda: rec(s3, g3, h3, y3), s3, 6))
report("rec | rec      :", timeit(lambda: rec(s2, g2, h2, y2), s2, 4), timeit(lambda: rec(s3, g3, h3, y3), s3, 4))
report("gemm | gemm    :", timeit(lambda: gemm(s1, g1), s1, 10), timeit(lambda: gemm(s2, g2), s2, 10))
for st in (y1, y2, y3): assert int(st[:4].view(torch.int32).item()) == 0
# how many recurrence launches really overlap?
NS = 6
ss = [torch.cuda.Stream() for _ in range(NS)]
bb = [bufs() for _ in range(NS)]
for i in range(NS): gemm(ss[i], bb[i][0])
torch.cuda.synchronize()
import time
for n in (1, 2, 3, 4, 6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(3):
        for i in range(n): rec(ss[i], *bb[i])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"{n} streams x 3 rec launches each: wall {dt:.2f} ms -> {dt / 3:.2f} ms per round, {3 * n / dt:.3f} rec/ms")
