"""Diagnostic (GPU box): intermediates of the HIP train-mode forward of CNNRNNModelLarge against the CPU oracle with the
same bf16 rounding points, stage by stage.  Usage: python tests/tools/large_train_debug.py [n_mels H L B T]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_ref as R  # noqa: E402
import music_transcription_amd as mta  # noqa: E402
from music_transcription_amd import train_step_large as TL  # noqa: E402

nm, H, L, B, T = [int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (32, 16, 2, 3, 40))]
g = torch.Generator().manual_seed(41)
mel = (torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=g))
sd = R.make_state_dict("cnn_rnn_large", nm, H, L, 23)
m = mta.TranscriptionModel("cnn_rnn_large", n_mels=nm, hidden_size=H, num_layers=L, dropout=0.0, device="cuda")
m.load_state_dict(sd, strict=True)
m.model.dropout2d_p = (0.0, 0.0, 0.0)
m.train()
with torch.no_grad():
    logits, sv = TL.forward_train_large(m.model, mel.cuda(), 0.0, 0, (0.0, 0.0, 0.0))
torch.cuda.synchronize()

o = R.Opts(gemm_bf16=True)
p = "model."
sdo = {k: v.clone() for k, v in sd.items()}
F1, F2, F3 = nm // 2, nm // 4, nm // 8


def cl(t, C, Fq):            # HIP channels-last [B*Fq*T (+pad)][C] -> [B][C][Fq][T]
    return t.float().cpu()[:B * Fq * T].reshape(B, Fq, T, C).permute(0, 3, 1, 2)


def rep(name, a, b):
    d = (a - b).abs()
    print(f"{name:28s} max|d| {float(d.max()):.4g}  mean|d| {float(d.mean()):.3g}  max|ref| {float(b.abs().max()):.3g}  frac>1e-2 {float((d > 1e-2).float().mean()):.3g}")


with torch.no_grad():
    h = R.pool_f2(torch.relu(R.conv_bn(mel, sdo, p + "conv1.0", p + "conv1.1", (1, 1), o, quant=False, train=True)))
    rep("a1 (conv1 block)", cl(sv["a1"], 32, F1), h)
    x = h
    for name, pre, cout, pool in (("rb1", p + "res_block1", 64, True), ("rb2", p + "res_block2", 128, False)):
        st = sv[name]
        Fin = st["Fin"]
        z1 = R.conv_bn.__wrapped__ if hasattr(R.conv_bn, "__wrapped__") else None
        # raw conv outputs (pre-BN), as the oracle rounds them
        w, bb = sdo[pre + ".conv1.weight"], sdo[pre + ".conv1.bias"]
        y1 = R._bf16_round(F.conv2d(R._bf16_round(x), R._bf16_round(w), bb, padding=(1, 1)))
        rep(name + " z1 (raw conv1)", cl(st["z1"], cout, Fin), y1)
        a = torch.relu(R.conv_bn(x, sdo, pre + ".conv1", pre + ".bn1", (1, 1), o, quant=True, train=True))
        rep(name + " y1 (bn1+relu)", cl(st["y1"], cout, Fin), a)
        ws, bs = sdo[pre + ".skip.0.weight"], sdo[pre + ".skip.0.bias"]
        ys = R._bf16_round(F.conv2d(R._bf16_round(x), R._bf16_round(ws), bs))
        rep(name + " zs (raw skip)", cl(st["zs"], cout, Fin), ys)
        w2, b2 = sdo[pre + ".conv2.weight"], sdo[pre + ".conv2.bias"]
        y2 = R._bf16_round(F.conv2d(R._bf16_round(a), R._bf16_round(w2), b2, padding=(1, 1)))
        rep(name + " z2 (raw conv2)", cl(st["z2"], cout, Fin), y2)
        out = R.res_block(x, sdo, pre, o, True)
        if pool:
            out = R.pool_f2(out)
        nxt = sv["rb2"]["xin"] if name == "rb1" else sv["r2"]
        rep(name + " out", cl(nxt, cout, Fin // 2 if pool else Fin), out)
        x = out
    hf = R.pool_f2(torch.relu(R.conv_bn(x, sdo, p + "freq_aware_conv.0", p + "freq_aware_conv.1", (3, 1), o, quant=True, train=True)))
    X0 = sv["main"]["Xs"][0].float().cpu()[:T * B].reshape(T, B, F3, 256).permute(1, 3, 2, 0)
    rep("X0 (fa conv block)", X0, hf)
    feats = hf.permute(0, 3, 1, 2).reshape(B, T, 256 * F3)
    main = R.bilstm(feats, sdo, p + "rnn_main", L, o)
    local = R.bilstm(feats, sdo, p + "rnn_local", 1, o)
    r = torch.cat([main, local], -1)
    comb = r.shape[-1]
    r32 = sv["r32"].cpu().reshape(T, B, comb).permute(1, 0, 2)
    rep("r (lstm concat)", r32, r)
    att = R.attention(r, sdo, p + "attention", 8, o)
    rep("proj (attention out)", sv["proj"].cpu().reshape(T, B, comb).permute(1, 0, 2), att)
    ln = F.layer_norm(r + att, (comb,), sdo[p + "attention_norm.weight"], sdo[p + "attention_norm.bias"], 1e-6)
    Cp = sv["ln"].shape[1]
    rep("ln", sv["ln"].float().cpu()[:T * B, :comb].reshape(T, B, comb).permute(1, 0, 2), ln)
    lo = R.cnnrnn_large_forward({k: v.clone() for k, v in sd.items()}, mel, return_all_heads=True, o=o, train=True)
    for i, k in enumerate(("frame", "onset", "offset")):
        rep("logits " + k, logits[i].cpu(), lo[k])
