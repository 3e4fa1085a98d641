"""Diagnostic (not shipped): build a -DMT_BPTT_DIAG copy of the backward recurrence and print where a step of
lstm_bptt_kernel (one direction per workgroup, MT_BPTT_MODE=1) spends its wall time.   python tools/bptt_diag.py [B T H]"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "music-transcription_amd", "csrc")
so = "/tmp/libmt_bptt_diag.so"
# the library's objects with the backward recurrence rebuilt under -DMT_BPTT_DIAG (run `make -C music-transcription_amd/csrc` first)
objs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".o") and f != "lstm_bwd.o"]
subprocess.check_call(f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMT_BPTT_DIAG -I{ROOT}/include -c {csrc}/lstm_bwd.hip -o /tmp/lstm_bwd_diag.o "
                      f"2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/lstm_bwd_diag.o {' '.join(objs)} -o {so}", shell=True)
os.environ["MT_BPTT_MODE"] = "1"
lib = C.CDLL(so)
B, T, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (16, 937, 512)))
vp = C.c_void_p
for f in ("mt_lstm_cx_bytes", "mt_lstm_dgx_bytes", "mt_lstm_bwd_part_bytes"):
    getattr(lib, f).restype = C.c_size_t
nkb, NG = H // 8, (B + 31) // 32
gates = torch.rand(NG * T * 2 * nkb * 1024, device="cuda")
cx = torch.randn(lib.mt_lstm_cx_bytes(B, T, H) // 4, device="cuda")
dh = torch.randn(lib.mt_lstm_cx_bytes(B, T, H) // 4, device="cuda") * 0.1
whh = ((torch.rand(2, 4 * H, H, device="cuda") * 2 - 1) / np.sqrt(H)).contiguous()
dgx = torch.empty(lib.mt_lstm_dgx_bytes(B, T, H), dtype=torch.uint8, device="cuda")
part = torch.empty(lib.mt_lstm_bwd_part_bytes(B, T, H), dtype=torch.uint8, device="cuda")
sync = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
lib.mt_lstm_bidir_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.mt_lstm_bidir_bwd(gates.data_ptr(), cx.data_ptr(), dh.data_ptr(), whh.data_ptr(), dgx.data_ptr(), part.data_ptr(), part.numel(),
                               sync.data_ptr(), sync.numel(), B, T, H, st)
    e1.record(); torch.cuda.synchronize()
    print("rc", rc, "status", hex(int(sync[:4].view(torch.int32).item())), f"launch {it}: {e0.elapsed_time(e1):.3f} ms (incl. the 0xFF fill of the "
          f"partial-product buffer) = {1e3 * e0.elapsed_time(e1) / T:.2f} us/step")
out = np.zeros((1024, 8), dtype=np.uint64)
lib.mt_lstm_bwd_diag_read.argtypes = [vp]
lib.mt_lstm_bwd_diag_read(out.ctypes.data)
nwg = 2 * ((H + 31) // 32) * NG
d = out[:nwg].astype(np.float64) * 10.0 / T      # ns per step (s_memrealtime ticks of 10 ns)
names = ["sleep + gather (poll)", "fetch issue + cell + img write", "barrier A", "LDS read + MFMA + publish", "dgx store + barrier B"]
for i, n in enumerate(names):
    print(f"{n:34s} mean {d[:, i].mean():8.0f} ns   min {d[:, i].min():8.0f}   max {d[:, i].max():8.0f}")
print(f"{'sum':34s} mean {d[:, :5].sum(1).mean():8.0f} ns")
print(f"failed polls per step (wave 0): mean {out[:nwg, 7].mean() / T:.3f}")
