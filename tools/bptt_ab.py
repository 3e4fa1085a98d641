"""Diagnostic (not shipped): time the in-tree backward recurrence under the env given on the command line and print a checksum of its
output, so that variants (MT_BPTT_POLL2, MT_BPTT_POLL_GAP, MT_BPTT_POLL_FIRST) can be compared launch for launch.
python tools/bptt_ab.py [B T H]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("MT_BPTT_AB_LIB") or os.path.join(ROOT, "music-transcription_amd", "libmt_hip.so"))   # a variant build to compare
B, T, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (16, 937, 512)))
vp = C.c_void_p
for f in ("mt_lstm_cx_bytes", "mt_lstm_dgx_bytes", "mt_lstm_bwd_part_bytes"):
    getattr(lib, f).restype = C.c_size_t
nkb, NG = H // 8, (B + 31) // 32
torch.manual_seed(0)
gates = torch.rand(NG * T * 2 * nkb * 1024, device="cuda")
cx = torch.randn(lib.mt_lstm_cx_bytes(B, T, H) // 4, device="cuda")
dh = torch.randn(lib.mt_lstm_cx_bytes(B, T, H) // 4, device="cuda") * 0.1
whh = ((torch.rand(2, 4 * H, H, device="cuda") * 2 - 1) / np.sqrt(H)).contiguous()
dgx = torch.zeros(lib.mt_lstm_dgx_bytes(B, T, H), dtype=torch.uint8, device="cuda")
part = torch.empty(lib.mt_lstm_bwd_part_bytes(B, T, H), dtype=torch.uint8, device="cuda")
sync = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
lib.mt_lstm_bidir_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
st = torch.cuda.current_stream().cuda_stream
ts = []
for it in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.mt_lstm_bidir_bwd(gates.data_ptr(), cx.data_ptr(), dh.data_ptr(), whh.data_ptr(), dgx.data_ptr(), part.data_ptr(), part.numel(),
                               sync.data_ptr(), sync.numel(), B, T, H, st)
    e1.record(); torch.cuda.synchronize()
    assert rc == 0 and int(sync[:4].view(torch.int32).item()) == 0, (rc, hex(int(sync[:4].view(torch.int32).item())))
    ts.append(e0.elapsed_time(e1))
env = {k: os.path.basename(v) for k, v in os.environ.items() if k.startswith("MT_BPTT")}
chk = int(dgx.view(torch.int32).to(torch.int64).sum().item())
print(f"{env}  B={B} T={T} H={H}: min {min(ts):.3f} ms  median {sorted(ts)[len(ts) // 2]:.3f} ms (incl. poison fill) = {1e3 * min(ts) / T:.2f} us/step  checksum {chk}")
