"""Diagnostic (not shipped): a -DMT_CONVG_DIAG copy of the generic convolution, and where a tile's wall time goes for the convolutions of
CNNRNNModelLarge at inference size (B = 16, T = 938, f16 operands).   python tools/convg_diag.py"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "music-transcription_amd", "csrc")
so = "/tmp/libmt_convg_diag.so"
srcs = [os.path.join(csrc, f) for f in ("api.hip", "convg.hip")]
extra = " ".join(a for a in sys.argv[1:] if a.startswith("-D"))
subprocess.check_call(f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMT_CONVG_DIAG {extra} -I{ROOT}/include -shared {' '.join(srcs)} -o {so}", shell=True)
lib = C.CDLL(so)
vp, i32 = C.c_void_p, C.c_int
lib.mt_conv_cl_ex.argtypes = [vp, i32, vp, i32, vp, vp, vp] + [i32] * 13 + [vp]
lib.mt_convg_diag_read.argtypes = [vp, i32]
B, T = 16, 938
st = torch.cuda.current_stream().cuda_stream
shapes = [("rb1.conv1", 114, 32, 0, 64, 3, 0), ("rb1.conv2+skip", 114, 64, 32, 64, 3, 1), ("rb2.conv1", 57, 64, 0, 128, 3, 0),
          ("rb2.conv2+skip", 57, 128, 64, 128, 3, 0), ("freq_aware 7x3", 57, 128, 0, 256, 7, 1)]
for name, F, C1, C2, Cout, KH, pool in shapes:
    A = torch.randn(B * F * T + 64, C1, device="cuda").half()
    S = torch.randn(B * F * T + 64, max(C2, 8), device="cuda").half() if C2 else None
    Ktot = KH * 3 * C1 + C2
    W = (torch.randn(Cout, Ktot, device="cuda") * 0.05).half()
    bias = torch.zeros(Cout, device="cuda")
    Fo = F // 2 if pool else F
    out = torch.empty(B * Fo * T + 64, Cout, device="cuda", dtype=torch.float16)
    buf = np.zeros(8, dtype=np.uint64)

    def run():
        rc = lib.mt_conv_cl_ex(A.data_ptr(), C1, S.data_ptr() if C2 else None, C2, W.data_ptr(), bias.data_ptr(), out.data_ptr(),
                               B, F, T, C1, C2, Cout, KH, 1, pool, 0, 0, 0, 1, st)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    lib.mt_convg_diag_read(buf.ctypes.data, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    lib.mt_convg_diag_read(buf.ctypes.data, 1)
    tiles = float(buf[4])
    ph = buf[:4].astype(np.float64) * 10.0 / tiles                      # ns per tile
    flops = 2.0 * B * F * T * Cout * Ktot
    print(f"{name:16s} {e0.elapsed_time(e1) / 3:6.3f} ms ({flops / (e0.elapsed_time(e1) / 3) * 1e-9:6.1f} TFLOP/s, stamps included)  per tile: "
          f"input staging {ph[0] / 1e3:5.2f} us, first weights {ph[1] / 1e3:5.2f}, main loop {ph[2] / 1e3:6.2f}, epilogue {ph[3] / 1e3:5.2f}  "
          f"(MFMA time of the main loop at peak: {2.0 * 256 * min(Cout, 256 if Cout % 256 == 0 else 128 if Cout % 128 == 0 else 64) * Ktot / 4069 / 2.4e3:5.2f} us)", flush=True)
