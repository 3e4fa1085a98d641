"""Diagnostic (not shipped): build mel.hip with -DMT_MEL_DIAG and print per-phase wall time of wave 0."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "music-transcription_amd", "csrc")
so = "/tmp/libmt_mel_diag.so"
subprocess.check_call(f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMT_MEL_DIAG -I{ROOT}/include -shared {csrc}/api.hip {csrc}/mel.hip -o {so}", shell=True)
lib = C.CDLL(so)
vp = C.c_void_p
class Desc(C.Structure): _fields_ = [(n, C.c_int) for n in ("sr", "hop", "n_mels", "ell_rows")]
lib.mt_mel_plan_bytes.restype = C.c_size_t
lib.mt_mel_plan_init.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(Desc), vp]
lib.mt_mel_db_f32.argtypes = [vp, C.POINTER(Desc), vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]
lib.mt_mel_diag_read.argtypes = [vp]
B, N = 32, 480000
wave = (torch.randn(B, N, device="cuda") * 0.1).contiguous()
plan = torch.empty(lib.mt_mel_plan_bytes(320), dtype=torch.uint8, device="cuda")
d = Desc(); st = torch.cuda.current_stream().cuda_stream
assert lib.mt_mel_plan_init(plan.data_ptr(), plan.numel(), 16000, 512, 320, d, st) == 0
mel = torch.empty(B, 320, 938, device="cuda"); cm = torch.empty(B, device="cuda")
for _ in range(3):
    assert lib.mt_mel_db_f32(plan.data_ptr(), d, wave.data_ptr(), B, N, mel.data_ptr(), cm.data_ptr(), 0, st) == 0
torch.cuda.synchronize()
out = np.zeros((1024, 12), dtype=np.uint64)
lib.mt_mel_diag_read(out.ctypes.data)
x = out[:256].astype(np.float64) * 10.0 / 7.5   # ns per (tile iteration): 3.75 tiles x 2 iterations per block
names = ["wait at loop top", "window (load wait)", "fft A", "twiddle", "transpose", "fft B", "split+power", "mel+dB", "tile store (per tile/2)"]
for i, n in enumerate(names):
    print(f"{n:26s} mean {x[:, i].mean():9.0f} ns  min {x[:, i].min():9.0f}  max {x[:, i].max():9.0f}")
print("sum", x[:, :9].sum(1).mean())
