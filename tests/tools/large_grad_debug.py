"""Diagnostic (GPU box): where does the convolution stack's BACKWARD pass of the CNNRNNModelLarge training step leave the oracle?
Every stage of backward_train_large below the LSTMs (BatchNorm + activation backward, input-gradient convolutions, weight
gradients) is captured and compared twice against torch autograd on the CPU oracle with the same bf16 rounding points:

  isolated   -- the oracle's stage is fed the HIP stage's own INPUT gradient (f64 autograd): the stage's own error;
  cumulative -- against the oracle's end-to-end backward (what the parity tests see).

Usage: python tests/tools/large_grad_debug.py [n_mels H L B T]      (default 320 64 2 2 200: the realistic-count test's shape)"""
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

nm, H, L, B, T = [int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (320, 64, 2, 2, 200))]
torch.set_num_threads(16)
g = torch.Generator().manual_seed(5)
gm = torch.Generator().manual_seed(31)
mel = (torch.rand(B, 1, nm, T, generator=gm) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=gm))
roll = (torch.rand(B, 88, T, generator=g) < 0.1).float()
lengths = torch.tensor([T] + [max(1, T - 37)] * (B - 1), dtype=torch.int64)
for b in range(1, B):
    mel[b, :, :, int(lengths[b]):] = 0.0
    roll[b, :, int(lengths[b]):] = 0.0
sd = R.make_state_dict("cnn_rnn_large", nm, H, L, 21)
m = mta.TranscriptionModel("cnn_rnn_large", n_mels=nm, hidden_size=H, num_layers=L, dropout=0.0, device="cuda")
m.load_state_dict(sd, strict=True)
m.model.dropout2d_p = (0.0, 0.0, 0.0)
m.train()
F1, F2, F3 = nm // 2, nm // 4, nm // 8

# ---------------------------------------------------------------------------------------------------- capture the HIP stages
cap = []
_bn, _cv, _wg = TL._bn_act_bwd, TL._conv, TL.conv_wgrad_direct
in_bwd = [False]


def bn_wrap(dcl, ldd_cl, dx, ldd_x, za, sa, zb, sb, mask, dza, dzb, grads, B_, F_, T_, C, relu, pool, dev, dza_lo=None, dzb_lo=None):
    r = _bn(dcl, ldd_cl, dx, ldd_x, za, sa, zb, sb, mask, dza, dzb, grads, B_, F_, T_, C, relu, pool, dev, dza_lo=dza_lo, dzb_lo=dzb_lo)
    torch.cuda.synchronize()
    cap.append(("bn", dict(dcl=None if dcl is None else dcl.float().cpu(), dx=None if dx is None else dx.cpu().clone(), F=F_, C=C, pool=pool,
                           dza=dza.float().cpu(), dza_lo=None if dza_lo is None else dza_lo.float().cpu(),
                           dzb=None if dzb is None else dzb.float().cpu(), dzb_lo=None if dzb_lo is None else dzb_lo.float().cpu(),
                           grads=[None if v is None else v.cpu().clone() for v in grads], sa=[v.cpu().clone() for v in sa[:2]])))
    return r


def conv_wrap(A, S, W, bias, out, B_, F_, T_, C1, C2, Cout, KH, relu=0, pool=0, out_mode=0, ldx=0, pitchA=None, pitchS=None, accum=0):
    _cv(A, S, W, bias, out, B_, F_, T_, C1, C2, Cout, KH, relu, pool, out_mode, ldx, pitchA, pitchS, accum)
    if in_bwd[0]:
        torch.cuda.synchronize()
        cap.append(("conv", dict(out=out.float().cpu(), F=F_, Cout=Cout, accum=accum)))


def wg_wrap(dz_hi, dz_lo, dz_pitch, x, x_pitch, B_, F_, T_, Cout, Cin, KH, KW, out):
    r = _wg(dz_hi, dz_lo, dz_pitch, x, x_pitch, B_, F_, T_, Cout, Cin, KH, KW, out)
    torch.cuda.synchronize()
    cap.append(("wgrad", dict(out=out.cpu().clone(), shape=(Cout, Cin, KH, KW))))
    return r


TL._bn_act_bwd, TL._conv, TL.conv_wgrad_direct = bn_wrap, conv_wrap, wg_wrap

# every library call goes check(lib.mt_x(ptr(a), ..., ptr(b)), "name"): ptr() sees the tensors, check() the name.  Snapshot the tensor
# arguments of the calls of interest (after the call is queued: clone() is ordered behind it on the stream)
from music_transcription_amd import train_step as TS  # noqa: E402
WANT = ("mt_layernorm_residual_bwd", "mt_axpby_rows_f32", "mt_lstm_bidir_bwd", "mt_lstm_dg_unpack", "mt_gemm_lstm_dh", "mt_lstm_dh_relayout")
calls, cur = [], []
_ptr, _check = TL.ptr, TL.check


def ptr_wrap(t):
    if torch.is_tensor(t):
        cur.append(t)
    return _ptr(t)


def check_wrap(rc, name=""):
    _check(rc, name)
    if in_bwd[0] and name in WANT:
        calls.append((name, [t.detach().clone() for t in cur]))
    cur.clear()


TL.ptr, TL.check, TS.ptr, TS.check = ptr_wrap, check_wrap, ptr_wrap, check_wrap
os.environ["MT_TRAIN_LARGE_STREAMS"] = "0"
logits = m(mel.cuda())
loss = m.compute_loss(logits, roll.cuda(), lengths)
in_bwd[0] = True
loss.backward()
torch.cuda.synchronize()
m.model.raise_on_train_handoff_timeout()
hip_grads = {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in m.model.named_parameters()}

# ---------------------------------------------------------------------------------------------------- the oracle, node by node
o = R.Opts(gemm_bf16=True)
p = "model."
sdo = {k: v.clone() for k, v in sd.items()}
for k, v in sdo.items():
    if v.dtype.is_floating_point and "running_" not in k:
        v.requires_grad_(True)
bf = R._bf16_round


def conv_raw(x, w, b, pad):
    y = F.conv2d(bf(x), bf(w), b, padding=pad)
    return bf(y.detach()) + (y - y.detach())


def bnorm(y, pre):
    return F.batch_norm(y, None, None, sdo[pre + ".weight"], sdo[pre + ".bias"], True, 0.1, R.BN_EPS)


nodes = {}


def keep(name, t):
    t.retain_grad()
    nodes[name] = t
    return t


h = keep("a1", R.pool_f2(torch.relu(R.conv_bn(mel, sdo, p + "conv1.0", p + "conv1.1", (1, 1), o, quant=False, train=True))))
x = h
for name, pre, pool in (("rb1", p + "res_block1", True), ("rb2", p + "res_block2", False)):
    z1 = keep(name + ".z1", conv_raw(x, sdo[pre + ".conv1.weight"], sdo[pre + ".conv1.bias"], (1, 1)))
    y1 = keep(name + ".y1", torch.relu(bnorm(z1, pre + ".bn1")))
    z2 = keep(name + ".z2", conv_raw(y1, sdo[pre + ".conv2.weight"], sdo[pre + ".conv2.bias"], (1, 1)))
    zs = keep(name + ".zs", conv_raw(x, sdo[pre + ".skip.0.weight"], sdo[pre + ".skip.0.bias"], (0, 0)))
    out = torch.relu(bnorm(z2, pre + ".bn2") + bnorm(zs, pre + ".skip.1"))
    if pool:
        out = R.pool_f2(out)
    x = keep(name + ".out", out)
zf = keep("zf", conv_raw(x, sdo[p + "freq_aware_conv.0.weight"], sdo[p + "freq_aware_conv.0.bias"], (3, 1)))
hf = keep("hf", R.pool_f2(torch.relu(bnorm(zf, p + "freq_aware_conv.1"))))
feats = hf.permute(0, 3, 1, 2).reshape(B, T, 256 * F3)


def lstm_dir_nodes(x, pre, l, suf, reverse, tag):
    """oracle.model_ref.lstm_dir with the gate pre-activation input (gx) kept as a node"""
    w_ih, w_hh = sdo[f"{pre}.weight_ih_l{l}{suf}"], sdo[f"{pre}.weight_hh_l{l}{suf}"]
    Hh = w_hh.shape[1]
    gx = keep(tag + ".gx", R._rq(x, o) @ R._rq(w_ih, o).t() + (sdo[f"{pre}.bias_ih_l{l}{suf}"] + sdo[f"{pre}.bias_hh_l{l}{suf}"]))
    q16 = lambda v: v + (v.half().float() - v).detach()
    w_hh_t = q16(w_hh).t().contiguous()
    h, c = x.new_zeros(B, Hh), x.new_zeros(B, Hh)
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        gg = gx[:, t] + h @ w_hh_t
        i_, f_, g_, o_ = gg.split(Hh, dim=1)
        c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
        h = q16(torch.sigmoid(o_) * torch.tanh(c))
        outs[t] = h
    return torch.stack(outs, 1)


def bilstm_nodes(x, pre, layers, tag):
    for l in range(layers):
        x = keep(f"{tag}.out{l}", torch.cat([lstm_dir_nodes(x, pre, l, "", False, f"{tag}.l{l}.f"), lstm_dir_nodes(x, pre, l, "_reverse", True, f"{tag}.l{l}.r")], -1))
    return x


main = bilstm_nodes(feats, p + "rnn_main", L, "main")
local = bilstm_nodes(feats, p + "rnn_local", 1, "local")
r = keep("r", torch.cat([main, local], dim=-1))
pre_ln = keep("pre_ln", r + R.attention(r, sdo, p + "attention", 8, o))
feat = keep("feat", F.layer_norm(pre_ln, (r.shape[-1],), sdo[p + "attention_norm.weight"], sdo[p + "attention_norm.bias"], R.LN_EPS))
shared = torch.relu(R._rq(feat, o) @ R._rq(sdo[p + "shared_fc.weight"], o).t() + sdo[p + "shared_fc.bias"])
lo = (R._rq(shared, o) @ R._rq(sdo[p + "frame_head.weight"], o).t() + sdo[p + "frame_head.bias"]).transpose(1, 2)
R.compute_loss(lo, roll, lengths).backward(retain_graph=True)
print(f"logits HIP vs oracle: max|d| {float((logits.detach().cpu() - lo.detach()).abs().max()):.4g}")


def cl(t, C, Fq):            # HIP channels-last [B*Fq*T (+pad)][C(+pad)] -> [B][C][Fq][T]
    return t[:B * Fq * T].reshape(B, Fq, T, -1)[..., :C].permute(0, 3, 1, 2)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), float((a - b).norm() / b.norm().clamp_min(1e-30))


def rep(tag, a, b):
    mx, l2 = rel(a, b)
    print(f"  {tag:58s} max|d|/max|ref| {mx:9.3g}   |d|2/|ref|2 {l2:9.3g}")


def agrad(out_node, in_nodes, gout):
    return torch.autograd.grad(out_node, in_nodes, grad_outputs=gout.to(out_node.dtype), retain_graph=True, allow_unused=True)


# ---------------------------------------------------------------------------------------------------- upper stack, in backward order
Hl = H // 2
comb = 2 * H + 2 * Hl
rows = lambda t_: t_.float().cpu()[:T * B].reshape(T, B, -1).permute(1, 0, 2)          # [(t*B+b)][C] -> [B][T][C]


def dh_layout(t_, Hp_, Hv):            # BPTT layout [g][t][d][Hp/8][8][32] f32 -> [B][T][2*Hv]
    v = t_.float().cpu().reshape(-1, T, 2, Hp_ // 8, 8, 32)[0]                          # one batch group (B <= 32)
    v = v.permute(4, 0, 1, 2, 3).reshape(32, T, 2, Hp_)[:B, :, :, :Hv]
    return v.reshape(B, T, 2 * Hv)


def dg_rows(t_, ld, col0, Hp_, Hv, d):  # dG [(t*B+b)][ld] bf16, column col0 + d*4Hp + p*Hp + j -> [B][T][4*Hv]
    v = t_.float().cpu().reshape(-1, ld)[:T * B, col0 + d * 4 * Hp_: col0 + (d + 1) * 4 * Hp_].reshape(T, B, 4, Hp_)[..., :Hv]
    return v.permute(1, 0, 2, 3).reshape(B, T, 4 * Hv)


print("== upper stack (cumulative = against the oracle's end-to-end backward; isolated = the oracle's stage fed the HIP stage's input)")
byname = {}
for nm_, ts in calls:
    byname.setdefault(nm_, []).append(ts)
ln_c = byname["mt_layernorm_residual_bwd"][0]        # r32, proj, ln_g, ln_stats, dfeat, dxln, part
dfeat_h, dxln_h = rows(ln_c[4])[..., :comb], rows(ln_c[5])[..., :comb]
rep("dfeat (into LayerNorm) cumulative", dfeat_h, nodes["feat"].grad)
rep("dxln cumulative", dxln_h, nodes["pre_ln"].grad)
(gi,) = agrad(nodes["feat"], [nodes["pre_ln"]], dfeat_h)
rep("dxln isolated", dxln_h, gi)
dr_h = rows(byname["mt_axpby_rows_f32"][0][2])[..., :comb]
rep("dr (into the LSTMs) cumulative", dr_h, nodes["r"].grad)
(gi,) = agrad(nodes["pre_ln"], [nodes["r"]], dxln_h)
rep("dr isolated (attention backward + residual)", dr_h, gi)
Hp_m, Hp_l = ((H + 15) // 16) * 16, ((Hl + 15) // 16) * 16
bw = byname["mt_lstm_bidir_bwd"]                      # args: gates, cx, dh, w_hh, dgx, part, sync
un = byname["mt_lstm_dg_unpack"]                      # args: dgx, dG, dGT
# call order with MT_TRAIN_LARGE_STREAMS=0: main layers L-1 .. 0, then the local layer
order = [("main", l, Hp_m, H) for l in range(L - 1, -1, -1)] + [("local", 0, Hp_l, Hl)]
for (tag, l, Hp_, Hv), b_args, u_args in zip(order, bw, un):
    dh_h = dh_layout(b_args[2], Hp_, Hv)
    out_node = nodes[f"{tag}.out{l}"]
    rep(f"{tag} layer {l}: dh (into BPTT) cumulative", dh_h, out_node.grad)
    dG = u_args[1]
    if l == 0:        # layer 0 writes its columns of the shared dG0 (row pitch 8 Hp_main + 8 Hp_local): the captured view starts at its first column
        ld = 8 * Hp_m + 8 * Hp_l
        flat = dG.float().cpu().reshape(-1)
        flat = flat[:(flat.numel() // ld) * ld] if flat.numel() >= T * B * ld else torch.cat([flat, flat.new_zeros(T * B * ld - flat.numel())])
        dg_of = lambda d, flat=flat, ld=ld: dg_rows(flat, ld, 0, Hp_, Hv, d)
    else:
        dg_of = lambda d, dG=dG: dg_rows(dG, 8 * Hp_, 0, Hp_, Hv, d)
    for d, sfx in ((0, "f"), (1, "r")):
        gx_node = nodes[f"{tag}.l{l}.{sfx}.gx"]
        dgh = dg_of(d)
        rep(f"{tag} layer {l} dir {sfx}: dgates cumulative", dgh, gx_node.grad)
        (gi,) = agrad(out_node, [gx_node], dh_h)
        rep(f"{tag} layer {l} dir {sfx}: dgates isolated (BPTT alone)", dgh, gi)
print("== convolution stack")
it = iter(cap)


def nxt(kind):
    k, d_ = next(it)
    assert k == kind, (k, kind)
    return d_


W = lambda k: sdo[p + k]
print("== freq_aware_conv")
s = nxt("bn")
dX0 = s["dx"][:T * B].reshape(T, B, F3, 256).permute(1, 3, 2, 0).contiguous()
rep("dX0 (into the conv stack) cumulative", dX0, nodes["hf"].grad)
gz, gg, gb = agrad(nodes["hf"], [nodes["zf"], W("freq_aware_conv.1.weight"), W("freq_aware_conv.1.bias")], dX0)
dzf = cl(s["dza"] + s["dza_lo"], 256, F2)
rep("dzf hi+lo isolated", dzf, gz); rep("dzf hi only isolated", cl(s["dza"], 256, F2), gz); rep("dzf cumulative", dzf, nodes["zf"].grad)
rep("bn gamma isolated", s["grads"][0], gg); rep("bn beta isolated", s["grads"][1], gb)
rep("bn gamma cumulative", s["grads"][0], W("freq_aware_conv.1.weight").grad)
c1 = nxt("conv"); c2 = nxt("conv")
(gr2,) = agrad(nodes["zf"], [nodes["rb2.out"]], dzf)
rep("dr2 (dgrad from hi) isolated vs autograd(hi+lo)", cl(c2["out"], 128, F2), gr2)
(gr2h,) = agrad(nodes["zf"], [nodes["rb2.out"]], cl(s["dza"], 256, F2))
rep("dr2 isolated vs autograd(hi only)", cl(c2["out"], 128, F2), gr2h)
rep("dr2 cumulative", cl(c2["out"], 128, F2), nodes["rb2.out"].grad)
w = nxt("wgrad")
(gw,) = agrad(nodes["zf"], [W("freq_aware_conv.0.weight")], dzf)
rep("fa weight grad isolated (dz = HIP hi+lo)", w["out"], gw); rep("fa weight grad cumulative", w["out"], W("freq_aware_conv.0.weight").grad)
dout = cl(c2["out"], 128, F2)
for name, pfx, cin, cout, pool, Fin in (("rb2", "res_block2", 64, 128, False, F2), ("rb1", "res_block1", 32, 64, True, F1)):
    print("==", pfx)
    xin_node = nodes["rb1.out"] if name == "rb2" else nodes["a1"]
    s = nxt("bn")
    g2, gs, gg2, gb2, ggs, gbs = agrad(nodes[name + ".out"], [nodes[name + ".z2"], nodes[name + ".zs"], W(pfx + ".bn2.weight"), W(pfx + ".bn2.bias"),
                                                          W(pfx + ".skip.1.weight"), W(pfx + ".skip.1.bias")], dout)
    dz2, dzs = cl(s["dza"] + s["dza_lo"], cout, Fin), cl(s["dzb"] + s["dzb_lo"], cout, Fin)
    rep("dz2 hi+lo isolated", dz2, g2); rep("dzs hi+lo isolated", dzs, gs)
    rep("dz2 cumulative", dz2, nodes[name + ".z2"].grad); rep("dzs cumulative", dzs, nodes[name + ".zs"].grad)
    rep("bn2 gamma isolated", s["grads"][0], gg2); rep("bn2 beta isolated", s["grads"][1], gb2)
    rep("skip bn gamma isolated", s["grads"][2], ggs); rep("skip bn beta isolated", s["grads"][3], gbs)
    rep("bn2 gamma cumulative", s["grads"][0], W(pfx + ".bn2.weight").grad); rep("skip bn gamma cumulative", s["grads"][2], W(pfx + ".skip.1.weight").grad)
    w2 = nxt("wgrad"); wsk = nxt("wgrad")
    (gw2,) = agrad(nodes[name + ".z2"], [W(pfx + ".conv2.weight")], dz2)
    (gws,) = agrad(nodes[name + ".zs"], [W(pfx + ".skip.0.weight")], dzs)
    rep("conv2 weight grad isolated", w2["out"], gw2); rep("conv2 weight grad cumulative", w2["out"], W(pfx + ".conv2.weight").grad)
    rep("skip weight grad isolated", wsk["out"], gws); rep("skip weight grad cumulative", wsk["out"], W(pfx + ".skip.0.weight").grad)
    c = nxt("conv")
    (gy1,) = agrad(nodes[name + ".z2"], [nodes[name + ".y1"]], cl(s["dza"], cout, Fin))
    dy1 = cl(c["out"], cout, Fin)
    rep("dy1 isolated vs autograd(hi only)", dy1, gy1); rep("dy1 cumulative", dy1, nodes[name + ".y1"].grad)
    s1 = nxt("bn")
    g1, gg1, gb1 = agrad(nodes[name + ".y1"], [nodes[name + ".z1"], W(pfx + ".bn1.weight"), W(pfx + ".bn1.bias")], dy1)
    dz1 = cl(s1["dza"] + s1["dza_lo"], cout, Fin)
    rep("dz1 hi+lo isolated", dz1, g1); rep("dz1 cumulative", dz1, nodes[name + ".z1"].grad)
    rep("bn1 gamma isolated", s1["grads"][0], gg1); rep("bn1 gamma cumulative", s1["grads"][0], W(pfx + ".bn1.weight").grad)
    c = nxt("conv")
    gx_a = agrad(nodes[name + ".z1"], [xin_node], cl(s1["dza"], cout, Fin))[0] + agrad(nodes[name + ".zs"], [xin_node], cl(s["dzb"], cout, Fin))[0]
    dxin = cl(c["out"], cin, Fin)
    rep("dxin isolated vs autograd(hi only)", dxin, gx_a); rep("dxin cumulative", dxin, xin_node.grad)
    w1 = nxt("wgrad")
    (gw1,) = agrad(nodes[name + ".z1"], [W(pfx + ".conv1.weight")], dz1)
    rep("conv1 weight grad isolated", w1["out"], gw1); rep("conv1 weight grad cumulative", w1["out"], W(pfx + ".conv1.weight").grad)
    dout = dxin
print("== conv1 (1 -> 32)")
for k in ("conv1.0.weight", "conv1.1.weight", "conv1.1.bias"):
    rep(k + " cumulative", hip_grads[k], W(k).grad)
