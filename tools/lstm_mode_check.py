import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import music_transcription_amd as mta
from music_transcription_amd._lib import lib, check, ptr
B, T, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4])
gx = torch.randn(lib.mt_lstm_gx_bytes(B, T, H) // 4, device="cuda") * 0.5
whh = ((torch.rand(2, 4 * H, H, device="cuda") * 2 - 1) / np.sqrt(H)).contiguous()
hx = torch.empty(lib.mt_lstm_hx_bytes(B, T, H) // 4, device="cuda")
sync = torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.mt_lstm_bidir_fwd_ex(ptr(gx), ptr(whh), ptr(hx), ptr(sync), sync.numel(), B, T, H, mode, st))
    e1.record(); torch.cuda.synchronize()
    w = sync.view(torch.int32)
    print(f"mode {mode} launch {it}: {e0.elapsed_time(e1):.3f} ms = {1e3 * e0.elapsed_time(e1) / T:.2f} us/step status {hex(int(w[0]))} tickets {w[8:16].tolist()} flags[0:8] {w[64:72].tolist()}")
