import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import music_transcription_amd as mta
from oracle import model_ref as R
def mel(B, nm, T, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=g))
for (nm, hs, nl, B, T, att, heads) in ((32, 16, 2, 2, 50, False, False), (32, 16, 2, 2, 50, False, True), (32, 16, 2, 2, 50, True, False),
                                       (32, 16, 2, 2, 50, True, True), (48, 32, 3, 1, 41, True, True), (64, 64, 1, 2, 40, True, True)):
    sd = R.make_state_dict("large", nm, hs, nl, 11, use_attention=att, use_heads=heads)
    m = mta.TranscriptionModel("large", n_mels=nm, hidden_size=hs, num_layers=nl, device="cuda", use_attention=att, use_onset_offset_heads=heads).eval()
    m.load_state_dict(sd, strict=True)
    x = mel(B, nm, T, 5)
    with torch.no_grad():
        got = m(x.cuda()).cpu()
        m.model.raise_on_handoff_timeout(B, T)
        ref = R.cnnrnn_large_forward(sd, x)
        emu = R.cnnrnn_large_forward(sd, x, o=R.Opts(gemm_bf16=True))
    print(f"nm={nm} H={hs} L={nl} att={att} heads={heads}: |got-ref|={float((got-ref).abs().max()):.4f} |got-emu|={float((got-emu).abs().max()):.4f} |emu-ref|={float((emu-ref).abs().max()):.4f} |ref|max={float(ref.abs().max()):.3f}")
if len(sys.argv) > 1:
    c = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "canonical_models.npz"))
    nm, hs, nl, B, T, wseed, xseed = [int(v) for v in c["large_937_cfg"]]
    sd = R.set_bn_flat(R.make_state_dict("cnn_rnn_large", nm, hs, nl, wseed), c["large_937_bn"])
    m = mta.TranscriptionModel("cnn_rnn_large", n_mels=nm, hidden_size=hs, num_layers=nl, device="cuda").eval()
    m.load_state_dict(sd, strict=True)
    x = mel(B, nm, T, xseed)
    with torch.no_grad():
        d = m(x.cuda(), return_all_heads=True)
        m.model.raise_on_handoff_timeout(B, T)
        import time; t0 = time.time()
        emu = R.cnnrnn_large_forward(sd, x, return_all_heads=True, o=R.Opts(gemm_bf16=True))
        print("emu time", time.time() - t0)
    for k, key in (("frame", "large_937_sample"), ("onset", "large_937_onset_sample"), ("offset", "large_937_offset_sample")):
        g = d[k].cpu().numpy()
        print(k, "vs golden", np.abs(g[:, ::5, ::7] - c[key]).max(), "vs emu", np.abs(g - emu[k].numpy()).max(), "emu vs golden",
              np.abs(emu[k].numpy()[:, ::5, ::7] - c[key]).max(), "max|ref|", np.abs(c[key]).max(), "nan", np.isnan(g).sum())
