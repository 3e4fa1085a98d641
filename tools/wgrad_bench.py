"""Convolution weight gradients of CNNRNNModelLarge at training size (B = 16, n_mels = 229, T = 937): the direct kernel
(csrc/conv_wgrad.hip) per shape -- ms, TFLOP/s over both pieces of dz -- and, with --planes, round 2's planes + batched GEMM path."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import music_transcription_amd as mta  # noqa: F401
from music_transcription_amd import train_step_large as TL

ap = argparse.ArgumentParser()
ap.add_argument("--planes", action="store_true", help="(needs a checkout that still has the planes path: git show 26866a0)")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--frames", type=int, default=937)
args = ap.parse_args()
B, T = args.batch, args.frames
dev = torch.device("cuda", 0)
shapes = [("rb1.conv1", 114, 32, 64, 3, 3), ("rb1.conv2", 114, 64, 64, 3, 3), ("rb1.skip", 114, 32, 64, 1, 1),
          ("rb2.conv1", 57, 64, 128, 3, 3), ("rb2.conv2", 57, 128, 128, 3, 3), ("rb2.skip", 57, 64, 128, 1, 1),
          ("freq_aware", 57, 128, 256, 7, 3)]          # (res_block2 does not pool: 57 rows)
tot_new = tot_old = 0.0
for name, F, Cin, Cout, KH, KW in shapes:
    x = torch.randn(B, F, T, Cin, device=dev).bfloat16()
    hi = torch.randn(B, F, T, Cout, device=dev).bfloat16()
    lo = (torch.randn(B, F, T, Cout, device=dev) * 2 ** -9).bfloat16()
    out = torch.empty(Cout, Cin, KH, KW, device=dev)
    flops = 2 * 2.0 * B * F * T * Cout * Cin * KH * KW

    def new():
        TL.conv_wgrad_direct(hi, lo, Cout, x, Cin, B, F, T, Cout, Cin, KH, KW, out)

    def old():
        pl = TL._Planes(B, F, T, KH // 2, dev)
        xP = pl.make(x, Cin, Cin, (2, 1, 0))
        TL._conv_wgrad(pl, [pl.make(hi, Cout, Cout, (1,)), pl.make(lo, Cout, Cout, (1,))], xP, Cout, Cin, KH, (0, 1, 2) if KW == 3 else (1,), out)

    res = {}
    for tag, fn in (("direct", new),) + ((("planes", old),) if args.planes else ()):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[tag] = e0.elapsed_time(e1) / 5
    keep = out.clone()
    if args.planes:
        new()
        torch.cuda.synchronize()
        d = float((out - keep).abs().max()) / float(keep.abs().max())
    else:
        d = float("nan")
    tot_new += res["direct"]
    tot_old += res.get("planes", 0.0)
    print(f"{name:11s} F={F:3d} {Cin:3d}->{Cout:3d} {KH}x{KW}: direct {res['direct']:7.3f} ms = {flops / res['direct'] * 1e-9:7.1f} TFLOP/s"
          + (f"   planes+GEMM {res['planes']:7.3f} ms   max rel diff {d:.2e}" if args.planes else ""), flush=True)
print(f"sum: direct {tot_new:.3f} ms" + (f", planes+GEMM {tot_old:.3f} ms" if args.planes else ""))
