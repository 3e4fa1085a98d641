"""Test reference (not shipped): the operand packing of the CNNRNNModel training step as plain torch expressions -- what pack_train did until
round 4, kept to check the job tables of music-transcription_amd/pack_plan.py bit for bit."""
from typing import Dict

import torch

from music_transcription_amd.train_step import _ru


def pack_train_torch(model, dev, part: str = "all") -> Dict[str, object]:
    """Device-side operand layouts of the CURRENT parameters (re-done every step: the optimizer moves them).
    part = "conv" (what the step needs first), "rnn" (LSTM + fc, packed on the side stream beside the convolutions) or "all"."""
    from music_transcription_amd.model import _pack_bilstm
    H, L, F = model.hidden_size, model.num_layers, model.n_mels
    Hp, K1, Fo2 = _ru(H, 16), _ru(2 * H, 64), (F // 2) // 2
    K0 = Fo2 * 64
    f32 = dict(device=dev, dtype=torch.float32)
    t: Dict[str, object] = {}
    t["dims"] = dict(H=H, Hp=Hp, L=L, F=F, F1=F // 2, Fo2=Fo2, K0=K0, K1=K1)
    if part in ("all", "conv"):
        c1, bn1, c2, bn2 = model.cnn[0], model.cnn[1], model.cnn[4], model.cnn[5]
        t["w1"] = c1.weight.detach().to(**f32).reshape(32, 9).contiguous()
        t["b1"] = c1.bias.detach().to(**f32).contiguous()
        w2 = c2.weight.detach().to(**f32)
        t["w2"] = w2.permute(0, 2, 3, 1).reshape(64, 288).to(torch.bfloat16).contiguous()           # [co][tap*32 + ci]
        t["b2"] = c2.bias.detach().to(**f32).contiguous()
        wd = torch.zeros(64, 576, **f32)                                                             # dgrad: [ci (pad 64)][tap'*64 + co]
        wd[:32] = w2.flip(2, 3).permute(1, 2, 3, 0).reshape(32, 576)
        t["w2d"] = wd.to(torch.bfloat16)
        t["zero64"] = torch.zeros(64, **f32)
        for i, bn in ((1, bn1), (2, bn2)):
            t[f"g{i}"] = bn.weight.detach().to(**f32).contiguous()
            t[f"be{i}"] = bn.bias.detach().to(**f32).contiguous()
    if part in ("all", "rnn"):
        cols = (torch.arange(64)[None, :] * Fo2 + torch.arange(Fo2)[:, None]).reshape(-1)             # kernel col f*64+c -> ref col c*Fo2+f
        t["w_ih"], t["b_g"], t["w_hh"] = _pack_bilstm(model.rnn, L, H, cols, dev, k0_cf=(64, K0 // 64))
        t["w_ihT"] = []
        for l in range(L):
            K = K0 if l == 0 else K1
            wT = torch.zeros(_ru(K, 128), 8 * Hp, device=dev, dtype=torch.bfloat16)
            wT[:K] = t["w_ih"][l][:8 * Hp].t()
            t["w_ihT"].append(wT)
        fw = torch.zeros(128, K1, **f32)
        fw[:88, :2 * H] = model.fc.weight.detach().to(**f32)
        t["fc_w"] = fw.to(torch.bfloat16)
        fwT = torch.zeros(_ru(K1, 128), 128, device=dev, dtype=torch.bfloat16)
        fwT[:K1] = t["fc_w"].t()
        t["fc_wT"] = fwT
        t["fc_b"] = model.fc.bias.detach().to(**f32).contiguous()
    return t


