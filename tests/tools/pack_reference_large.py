"""Test reference (not shipped): the operand packing of the CNNRNNModelLarge training step written as plain torch expressions -- what
pack_train_large did until round 4, kept to check the job tables of music-transcription_amd/pack_plan.py bit for bit."""
from typing import Dict

import torch

from music_transcription_amd import _lib
from music_transcription_amd._lib import lib, check, ptr
from music_transcription_amd.train_step import _ru, _st


# ---------------------------------------------------------------------------------------------------------------- packing
def _conv_cl(w):            # [Cout][Cin][KH][KW] -> [Cout][(kh*KW + kw)*Cin + ci]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def _conv_dgrad_w(w, rows_pad=None):
    """Input-gradient weights of a KH x 3 convolution: [Cin (padded rows)][(kh'*3 + kw')*Cout + co] = w[co][ci][KH-1-kh'][2-kw']."""
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1)
    if rows_pad and rows_pad > wd.shape[0]:
        wd = torch.cat([wd, wd.new_zeros(rows_pad - wd.shape[0], wd.shape[1])], 0)
    return wd


def pack_train_large_torch(model, dev, side=None) -> Dict[str, object]:
    """bf16 operand layouts of the CURRENT parameters (redone every step: the optimizer moves them).  With `side` (a stream) everything
    above the convolution stack -- 99 % of the bytes: LSTM, attention and head weights, the layer-0 column permutation -- is packed on
    that stream beside the convolution stack's forward; t["_ready"] is the event to wait for before touching those entries."""
    from music_transcription_amd.model import _pack_bilstm
    H, L, Hl, F = model.hidden_size, model.num_layers, model.hidden_size // 2, model.n_mels
    Hp, Hlp, K1 = _ru(H, 16), _ru(Hl, 16), _ru(2 * H, 64)
    F1, F2, F3 = F // 2, F // 4, F // 8
    K0 = F3 * 256
    comb = 2 * H + 2 * Hl
    Cp = _ru(comb, 64)
    f32 = dict(device=dev, dtype=torch.float32)
    bf = torch.bfloat16
    t: Dict[str, object] = {}
    d = dict(H=H, Hp=Hp, Hl=Hl, Hlp=Hlp, L=L, F=F, F1=F1, F2=F2, F3=F3, K0=K0, K1=K1, comb=comb, Cp=Cp)
    g = lambda p: p.detach().to(**f32)
    t["w1"], t["b1"] = g(model.conv1[0].weight).reshape(32, 9).contiguous(), g(model.conv1[0].bias).contiguous()
    t["g1"], t["be1"] = g(model.conv1[1].weight).contiguous(), g(model.conv1[1].bias).contiguous()
    for name, rb, cin, cout in (("rb1", model.res_block1, 32, 64), ("rb2", model.res_block2, 64, 128)):
        w1, w2, ws = g(rb.conv1.weight), g(rb.conv2.weight), g(rb.skip[0].weight).reshape(cout, cin)
        t[name + "c1_w"], t[name + "c1_b"] = _conv_cl(w1).to(bf).contiguous(), g(rb.conv1.bias).contiguous()
        t[name + "c2_w"], t[name + "c2_b"] = _conv_cl(w2).to(bf).contiguous(), g(rb.conv2.bias).contiguous()
        wsp = torch.zeros(128, 64 if cin < 64 else cin, **f32)             # the 1x1 skip as a GEMM (K padded to 64: see mt_gemm)
        wsp[:cout, :cin] = ws
        t[name + "s_w"], t[name + "s_b"] = wsp.to(bf), g(rb.skip[0].bias).contiguous()
        t[name + "c2_wd"] = _conv_dgrad_w(w2).to(bf).contiguous()                                   # [cout][9*cout]
        cin_p = max(cin, 64)                                                                        # conv_cl wants Cout % 64 == 0
        wd = torch.zeros(cin_p, 9 * cout + cout, **f32)                                             # conv1 dgrad + skip^T in one call
        wd[:cin, :9 * cout] = _conv_dgrad_w(w1)
        wd[:cin, 9 * cout:] = ws.t()
        t[name + "c1s_wd"] = wd.to(bf)
        for bn, tag in ((rb.bn1, "bn1"), (rb.bn2, "bn2"), (rb.skip[1], "bns")):
            t[f"{name}{tag}_g"], t[f"{name}{tag}_b"] = g(bn.weight).contiguous(), g(bn.bias).contiguous()
    wf = g(model.freq_aware_conv[0].weight)
    t["fa_w"], t["fa_b"] = _conv_cl(wf).to(bf).contiguous(), g(model.freq_aware_conv[0].bias).contiguous()
    t["fa_wdA"] = _conv_dgrad_w(wf[:128]).to(bf).contiguous()             # input gradient in two halves of the 256 output channels
    t["fa_wdB"] = _conv_dgrad_w(wf[128:]).to(bf).contiguous()
    t["fa_g"], t["fa_be"] = g(model.freq_aware_conv[1].weight).contiguous(), g(model.freq_aware_conv[1].bias).contiguous()
    t["zeros256"] = torch.zeros(256, **f32)
    t["dims"] = d
    if side is None:
        _pack_upper(model, dev, t, d)
    else:
        main_st = torch.cuda.current_stream(dev)
        conv_keys = set(t.keys())
        side.wait_stream(main_st)                       # the optimizer's update of the parameters is ordered on the calling stream
        with torch.cuda.stream(side):
            _pack_upper(model, dev, t, d)
            ev = torch.cuda.Event()
            ev.record(side)
        for k_, v_ in t.items():                        # allocated under the side stream, used (and freed) under the calling one
            if k_ in conv_keys:
                continue
            for t_ in (v_ if isinstance(v_, (list, tuple)) else [v_]):
                if isinstance(t_, torch.Tensor):
                    t_.record_stream(main_st)
        t["_ready"] = ev
    return t


def _pack_upper(model, dev, t, d):
    from music_transcription_amd.model import _pack_bilstm
    H, L, Hl, Hp, Hlp, K0, K1, F3, comb, Cp = (d[k] for k in ("H", "L", "Hl", "Hp", "Hlp", "K0", "K1", "F3", "comb", "Cp"))
    f32 = dict(device=dev, dtype=torch.float32)
    bf = torch.bfloat16
    g = lambda p: p.detach().to(**f32)
    # LSTMs (layer-0 columns re-ordered: reference feature c*F3+f -> kernel column f*256+c)
    cols = (torch.arange(256)[None, :] * F3 + torch.arange(F3)[:, None]).reshape(-1)
    # layer 0 of BOTH LSTMs into one 16-bit tensor [main 8 Hp rows; local 8 Hlp rows (+ tile slack)][K0]: the two forward projections read
    # their own rows, and ONE transpose of the whole gives [W_ih_main; W_ih_local]^T for the input-gradient GEMM (was: two index gathers
    # and casts per LSTM, a zero fill of the 126-MB transposed operand and two strided copies into it, every step)
    R0, R1 = 8 * Hp, 8 * Hlp
    wboth = torch.empty(R0 + _ru(R1, 128) + 128, K0, device=dev, dtype=bf)
    wboth[R0 + R1:].zero_()
    t["m_wih"], t["m_b"], t["m_whh"] = _pack_bilstm(model.rnn_main, L, H, cols, dev, k0_cf=(256, F3), wih0_out=wboth)
    t["l_wih"], t["l_b"], t["l_whh"] = _pack_bilstm(model.rnn_local, 1, Hl, cols, dev, k0_cf=(256, F3), wih0_out=wboth[R0:])
    t["m_wihT"] = [None]
    for l in range(1, L):
        wT = torch.zeros(_ru(K1, 128), 8 * Hp, device=dev, dtype=bf)
        wT[:K1] = t["m_wih"][l][:8 * Hp].t()
        t["m_wihT"].append(wT)
    wcat = torch.empty(_ru(K0, 128), R0 + R1, device=dev, dtype=bf)             # dX0 = [dG_main | dG_local] . [W_ih_main; W_ih_local]
    if _ru(K0, 128) > K0:
        wcat[K0:].zero_()
    check(lib.mt_transpose_bf16(ptr(wboth), K0, R0 + R1, K0, ptr(wcat), R0 + R1, K0, _st()), "mt_transpose_bf16")
    t["ml_wihT"] = wcat
    if model.use_attention:
        heads, dh = model.attention.num_heads, model.attention.head_dim
        dp = _ru(dh, 64)
        Ca = heads * dp
        d.update(heads=heads, dh=dh, dp=dp, Ca=Ca, ld3=3 * Ca, scale=float(dh) ** -0.5)
        qw = g(model.attention.qkv.weight).reshape(3, heads, dh, comb)
        qwp = torch.zeros(3, heads, dp, Cp, **f32); qwp[:, :, :dh, :comb] = qw
        qbp = torch.zeros(3, heads, dp, **f32); qbp[:, :, :dh] = g(model.attention.qkv.bias).reshape(3, heads, dh)
        qfull = torch.zeros(_ru(3 * Ca, 128), Cp, **f32); qfull[:3 * Ca] = qwp.reshape(3 * Ca, Cp)
        t["qkv_w"], t["qkv_b"] = qfull.to(bf), qbp.reshape(-1).contiguous()
        qT = torch.zeros(_ru(comb, 128), 3 * Ca, **f32); qT[:Cp] = qfull[:3 * Ca].t()
        t["qkv_wT"] = qT.to(bf)
        pw = g(model.attention.proj.weight).reshape(comb, heads, dh)
        pwp = torch.zeros(_ru(comb, 128), heads, dp, **f32); pwp[:comb, :, :dh] = pw
        t["proj_w"], t["proj_b"] = pwp.reshape(-1, Ca).to(bf), g(model.attention.proj.bias).contiguous()
        pT = torch.zeros(_ru(Ca, 128), Cp, **f32); pT[:Ca, :comb] = pwp.reshape(-1, Ca)[:comb].t()
        t["proj_wT"] = pT.to(bf)
        t["ln_g"], t["ln_b"] = g(model.attention_norm.weight).contiguous(), g(model.attention_norm.bias).contiguous()
    if model.use_onset_offset_heads:
        Hs = _ru(H, 64)
        d.update(Hs=Hs)
        sw = torch.zeros(_ru(H, 128), Cp, **f32); sw[:H, :comb] = g(model.shared_fc.weight)
        t["shared_w"], t["shared_b"] = sw.to(bf), g(model.shared_fc.bias).contiguous()
        swT = torch.zeros(_ru(comb, 128), Hs, **f32); swT[:comb, :H] = g(model.shared_fc.weight).t()
        t["shared_wT"] = swT.to(bf)
        hw = torch.zeros(384, Hs, **f32)
        hw[:264, :H] = torch.cat([g(m.weight) for m in (model.frame_head, model.onset_head, model.offset_head)], 0)
        t["heads_w"] = hw.to(bf)
        t["heads_b"] = torch.cat([g(m.bias) for m in (model.frame_head, model.onset_head, model.offset_head)], 0).contiguous()
        hwT = torch.zeros(_ru(Hs, 128), 384, **f32); hwT[:Hs] = hw.t()
        t["heads_wT"] = hwT.to(bf)
    else:
        fw = torch.zeros(128, Cp, **f32); fw[:88, :comb] = g(model.fc.weight)
        t["fc_w"], t["fc_b"] = fw.to(bf), g(model.fc.bias).contiguous()
        fwT = torch.zeros(_ru(comb, 128), 128, **f32); fwT[:comb] = fw[:, :comb].t()
        t["fc_wT"] = fwT.to(bf)


