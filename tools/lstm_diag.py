"""Diagnostic (not shipped): build libmt_hip_diag.so with -DMT_LSTM_DIAG and print where a recurrence
step spends its wall time.  Usage on the GPU box:  python tools/lstm_diag.py [B T H]"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "music-transcription_amd", "csrc")
so = "/tmp/libmt_hip_diag.so"
extra = os.environ.get("MT_DIAG_FLAGS", "")
# the library's objects with the recurrence rebuilt under -DMT_LSTM_DIAG (run `make -C music-transcription_amd/csrc` first)
objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "lstm.o"]
subprocess.check_call(f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMT_LSTM_DIAG {extra} -I{ROOT}/include -c {csrc}/lstm.hip -o /tmp/lstm_diag.o "
                      f"2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/lstm_diag.o {' '.join(objs)} -o {so}", shell=True)
lib = C.CDLL(so)
B, T, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (32, 938, 512)))
vp = C.c_void_p
lib.mt_lstm_gx_bytes.restype = lib.mt_lstm_hx_bytes.restype = lib.mt_lstm_sync_bytes.restype = C.c_size_t
lib.mt_lstm_bidir_fwd_ex.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, vp]
MODE = 0
lib.mt_lstm_diag_read.argtypes = [vp]
gx = torch.randn(lib.mt_lstm_gx_bytes(B, T, H) // 4, device="cuda") * 0.5
whh = ((torch.rand(2, 4 * H, H, device="cuda") * 2 - 1) / np.sqrt(H)).contiguous()
hx = torch.empty(lib.mt_lstm_hx_bytes(B, T, H) // 4, device="cuda")
sync = torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.mt_lstm_bidir_fwd_ex(gx.data_ptr(), whh.data_ptr(), hx.data_ptr(), sync.data_ptr(), sync.numel(), B, T, H, MODE, st)
    e1.record(); torch.cuda.synchronize()
    print("rc", rc, "status", hex(int(sync[:4].view(torch.int32).item())), "tickets", sync[32:64].view(torch.int32).tolist())
    print(f"launch {it}: {e0.elapsed_time(e1):.3f} ms  = {1e3 * e0.elapsed_time(e1) / T:.2f} us/step")
out = np.zeros((1024, 8), dtype=np.uint64)
lib.mt_lstm_diag_read(out.ctypes.data)
nwg = 2 * (H // 8) * ((B + 31) // 32) if MODE == 0 else 8 * (H // 8)
d = out[:nwg].astype(np.float64) * 10.0 / T      # ns per step
d = d[d.sum(1) > 0]                               # XCD mode: workgroups that found no lane exit at once
names = ["poll wait", "barrier", "h load + MFMA", "LDS reduce (+barrier)", "cell", "store + drain", "barrier + flag", "-"]
for i, n in enumerate(names[:7]):
    print(f"{n:24s} mean {d[:, i].mean():8.0f} ns   min {d[:, i].min():8.0f}   max {d[:, i].max():8.0f}")
print(f"{'sum':24s} mean {d[:, :7].sum(1).mean():8.0f} ns")
print(f"poisoned polls per step (wave 0): mean {out[:nwg][out[:nwg].sum(1) > 0][:, 7].mean() / T:.3f}")
