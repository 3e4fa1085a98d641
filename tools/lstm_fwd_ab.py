"""Diagnostic (not shipped): time the in-tree forward recurrence (train mode and inference, f32 gx) and print a checksum of hx, so that builds
(MT_LSTM_AB_LIB=<variant .so>) and knobs (MT_LSTM_C16) can be compared launch for launch.   python tools/lstm_fwd_ab.py [B T H]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("MT_LSTM_AB_LIB") or os.path.join(ROOT, "music-transcription_amd", "libmt_hip.so"))
B, T, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (16, 937, 512)))
vp = C.c_void_p
for f in ("mt_lstm_gx_bytes", "mt_lstm_hx_bytes", "mt_lstm_cx_bytes", "mt_lstm_sync_bytes"):
    getattr(lib, f).restype = C.c_size_t
torch.manual_seed(0)
gx0 = torch.randn(lib.mt_lstm_gx_bytes(B, T, H) // 4, device="cuda") * 0.5
whh = ((torch.rand(2, 4 * H, H, device="cuda") * 2 - 1) / np.sqrt(H)).contiguous()
hx = torch.empty(lib.mt_lstm_hx_bytes(B, T, H) // 4, device="cuda")
cx = torch.empty(lib.mt_lstm_cx_bytes(B, T, H) // 4, device="cuda")
sync = torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device="cuda")
lib.mt_lstm_bidir_fwd_train.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
lib.mt_lstm_bidir_fwd.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
st = torch.cuda.current_stream().cuda_stream
env = {k: os.path.basename(v) for k, v in os.environ.items() if k.startswith("MT_LSTM")}
for train in (1, 0):
    ts = []
    for it in range(8):
        gx = gx0.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if train:
            rc = lib.mt_lstm_bidir_fwd_train(gx.data_ptr(), whh.data_ptr(), hx.data_ptr(), cx.data_ptr(), sync.data_ptr(), sync.numel(), B, T, H, st)
        else:
            rc = lib.mt_lstm_bidir_fwd(gx.data_ptr(), whh.data_ptr(), hx.data_ptr(), sync.data_ptr(), sync.numel(), B, T, H, st)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0 and int(sync[:4].view(torch.int32).item()) == 0, (rc, hex(int(sync[:4].view(torch.int32).item())))
        ts.append(e0.elapsed_time(e1))
    chk = int(hx.view(torch.int32).to(torch.int64).sum().item()) + (int(cx.view(torch.int32).to(torch.int64).sum().item()) + int(gx.view(torch.int32).to(torch.int64).sum().item()) if train else 0)
    print(f"{env}  {'train' if train else 'infer'} B={B} T={T} H={H}: min {min(ts):.3f} ms  median {sorted(ts)[len(ts) // 2]:.3f} ms (incl. the poison fill) = {1e3 * min(ts) / T:.2f} us/step  checksum {chk}")
