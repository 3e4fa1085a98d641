"""GPU parity of the CNNRNNModelLarge training step: the HIP train-mode forward / backward (through the
C ABI) against the reference's own gradients (tests/golden/train_step_large.npz, written by make_golden_train.py from the
reference's modules and its train_one_epoch), against torch autograd on the CPU oracle at other shapes / variants, and
unit tests of the pieces (BatchNorm + activation forward / backward, convolution weight gradients over position planes,
attention softmax backward with the clamp, LayerNorm backward) against torch autograd.

Tolerances as tests/test_gpu_train.py: GEMM / conv operands are bf16 (f32 accumulate), so a gradient tensor is compared
relative to its own largest entry and the whole flat gradient by its cosine to the reference."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model_ref as R

# Against the oracle with the HIP path's bf16 rounding points.  Which distance is a kernel error and which is the model's own
# sensitivity?  MEASURED (round 4, tests/tools/large_grad_debug.py + the oracle alone on the CPU):
#   * every backward stage fed the HIP path's OWN input gradient agrees with f64 autograd on that stage: convolution weight
#     gradients 0.2 - 0.4 %, input-gradient convolutions 0.3 %, attention + residual 0.04 %, LayerNorm 1 %, BPTT 0.4 - 0.7 % per layer;
#   * the gradient entering the top of the backward pass (d feat, behind the heads) is already 7.6 % (L2) from the oracle's, and every
#     stage below inherits that: it is ReLU / max-pool DECISIONS that differ where the two forward passes differ in a last bit, and
#     BatchNorm's projections (sum dz = 0, sum dz xhat = 0) remove most of the upstream gradient's norm but none of that noise;
#   * the oracle is that sensitive to ITSELF: perturbing its 16-bit roundings by 2^-20 relative -- the size of an f32 re-association --
#     moves its train-mode logits by 0.03 and its conv weight gradients by 10 - 19 % of their largest entry (cosine 0.9977); at 2^-24 still
#     7.5 - 10 % (cosine 0.9986 - 0.9992).  The HIP path sits at logits 0.04, conv weight gradients 9 - 11 %, cosine 0.9977 at that shape.
# So the per-tensor bound is NOT a fixed table: it is the oracle's own noise floor at the test's shape (_oracle_noise_floor: three
# perturbed runs of the oracle, per-tensor maximum), with GRAD_REL as the bound wherever the floor is below it.
GRAD_REL = 6e-2        # per tensor: max |g - g_ref| / max |g_ref|
FLOOR_AMP, FLOOR_SEEDS = 2.0 ** -20, (1, 2, 3)
FLOOR_FACTOR, FLOOR_MARGIN = 1.5, 0.02     # HIP within 1.5 x the floor + 2 % (the floor itself is three random draws)
GRAD_COS = 0.998
# Against the fp32 reference golden the bar is set by bf16 itself, not by the kernels: at the golden's toy shapes (a few
# hundred positions per BatchNorm channel, random labels) the CPU oracle with bf16-rounded activations and EXACT f32
# autograd is already cos 0.992 / up to 45 % per tensor away from the fp32 reference (ReLU / max-pool decisions flip, and
# BatchNorm makes the conv weight gradients heavily cancelling sums) -- measured in the test and printed.  The HIP path
# must be no further from fp32 than that oracle plus GRAD_REL, tensor by tensor.
GRAD_COS_FP32 = 0.99
LOGIT_TOL = 3e-2       # eval-mode logits against the fp32 oracle
LOGIT_TOL_TRAIN_FP32 = 6e-2   # train-mode logits (bf16 operands through 8 convolutions with batch statistics over a few hundred
LOGIT_TOL_TRAIN_EMU = 4e-2    # positions at these tiny shapes) against the fp32 reference / the oracle with the same rounding points
# a conv bias in front of a BatchNorm has an analytically zero gradient (the reference's is rounding noise)
ZERO_GRAD = ("conv1.0.bias", "res_block1.conv1.bias", "res_block1.conv2.bias", "res_block1.skip.0.bias", "res_block2.conv1.bias",
             "res_block2.conv2.bias", "res_block2.skip.0.bias", "freq_aware_conv.0.bias")


@pytest.fixture(scope="module")
def mta():
    import music_transcription_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def _lib():
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    return lib, check, ptr, stream_ptr


def _mel_in(B, nm, T, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=g))


def _roll_in(B, T, seed, p=0.04):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 88, T, generator=g) < p).float()


def _bf(x):
    return x.to(torch.bfloat16).float()


# ------------------------------------------------------------------------------------------------ kernel unit tests
@pytest.mark.parametrize("C,F_,T,B,two,relu,pool,p2d,xrows", [(64, 6, 21, 2, True, 1, 1, 0.0, False), (128, 5, 17, 3, True, 1, 0, 0.3, False),
                                                                (256, 7, 19, 2, False, 1, 1, 0.2, True), (32, 4, 70, 1, False, 1, 0, 0.0, False),
                                                                (64, 8, 33, 2, False, 0, 0, 0.0, False)])
def test_bn_act_forward_backward_vs_autograd(mta, C, F_, T, B, two, relu, pool, p2d, xrows):
    lib, check, ptr, st = _lib()
    g = torch.Generator().manual_seed(C + F_ + T)
    za = _bf(torch.randn(B, F_, T, C, generator=g) * 1.5 + 0.3)
    zb = _bf(torch.randn(B, F_, T, C, generator=g)) if two else None
    par = lambda: (torch.randn(C, generator=g) * 0.2, 1.0 / (0.5 + torch.rand(C, generator=g)), 1.0 + 0.3 * torch.randn(C, generator=g),
                   0.2 * torch.randn(C, generator=g))
    # batch statistics (what makes the BN backward's mean terms right): mean / rstd of the tensors themselves
    def stats(z):
        m = z.mean(dim=(0, 1, 2))
        v = z.var(dim=(0, 1, 2), unbiased=False)
        return m, 1.0 / torch.sqrt(v + 1e-5)
    ma, ra = stats(za)
    _, _, ga, ba = par()
    if two:
        mb, rb_ = stats(zb)
        _, _, gb, bb = par()
    Fo = F_ // 2 if pool else F_
    mask = None
    if p2d > 0:
        mask = torch.empty(B, C, device="cuda")
        check(lib.mt_dropout2d_mask(ptr(mask), B, C, p2d, 123, 7, st()))
        frac = float((mask == 0).float().mean())
        assert abs(frac - p2d) < 0.15 and torch.all((mask == 0) | ((mask - 1 / (1 - p2d)).abs() < 1e-6))
    dev = lambda t: None if t is None else t.cuda().contiguous()
    zad, zbd = dev(za.bfloat16()), dev(zb.bfloat16() if two else None)
    sa = [dev(v) for v in (ma, ra, ga, ba)]
    sb = [dev(v) for v in (mb, rb_, gb, bb)] if two else [None] * 4
    ldx = Fo * C + 8
    out = torch.zeros(B * T if xrows else B * Fo * T, ldx if xrows else C, dtype=torch.bfloat16, device="cuda")
    check(lib.mt_bn_act_fwd(ptr(zad), *(ptr(v) for v in sa), ptr(zbd), *(ptr(v) for v in sb), ptr(mask), ptr(out), 1 if xrows else 0, ldx,
                            B, F_, T, C, relu, pool, st()), "fwd")
    # reference with autograd (channels-first torch ops on the same bf16-rounded inputs)
    zar = za.clone().requires_grad_(True)
    zbr = zb.clone().requires_grad_(True) if two else None
    gar, bar = ga.clone().requires_grad_(True), ba.clone().requires_grad_(True)
    def bn(z, gam, bet):
        m = z.mean(dim=(0, 1, 2)); v = z.var(dim=(0, 1, 2), unbiased=False)
        return (z - m) / torch.sqrt(v + 1e-5) * gam + bet
    y = bn(zar, gar, bar)
    if two:
        gbr, bbr = gb.clone().requires_grad_(True), bb.clone().requires_grad_(True)
        y = y + bn(zbr, gbr, bbr)
    if relu:
        y = torch.relu(y)
    if pool:
        y = F.max_pool2d(y.permute(0, 3, 1, 2), kernel_size=(2, 1)).permute(0, 2, 3, 1)
    if mask is not None:
        y = y * mask.cpu()[:, None, None, :]
    if xrows:
        got = out.float().cpu()[:, :Fo * C].reshape(T, B, Fo, C).permute(1, 2, 0, 3)
    else:
        got = out.float().cpu().reshape(B, Fo, T, C)
    assert (got - y.detach()).abs().max() < 2e-2 * max(1.0, float(y.abs().max()))
    # backward
    dout = torch.randn(B, Fo, T, C, generator=g)
    y.backward(dout)
    sums = torch.empty(3 * C, dtype=torch.float64, device="cuda")
    dza = torch.empty(B * F_ * T, C, dtype=torch.bfloat16, device="cuda")
    dzb = torch.empty_like(dza) if two else None
    gr = [torch.empty(C, device="cuda") for _ in range(4)]
    if xrows:
        dx = torch.zeros(T * B, ldx, device="cuda")
        dx[:, :Fo * C] = dout.permute(2, 0, 1, 3).reshape(T * B, Fo * C).cuda()
        dcl = None
    else:
        dcl, dx = dout.bfloat16().cuda().contiguous(), None
        dout = dcl.float().cpu()
        # (the bf16 rounding of the incoming gradient is part of the contract: redo the reference backward with it)
        for t_ in (zar, zbr, gar, bar) + ((gbr, bbr) if two else ()):
            if t_ is not None:
                t_.grad = None
        y2 = bn(zar, gar, bar) + (bn(zbr, gbr, bbr) if two else 0.0)
        y2 = torch.relu(y2) if relu else y2
        y2 = F.max_pool2d(y2.permute(0, 3, 1, 2), kernel_size=(2, 1)).permute(0, 2, 3, 1) if pool else y2
        y2 = y2 * mask.cpu()[:, None, None, :] if mask is not None else y2
        y2.backward(dout)
    check(lib.mt_bn_act_bwd(ptr(dcl), C, ptr(dx), ldx, ptr(zad), *(ptr(v) for v in sa), ptr(zbd), *(ptr(v) for v in sb), ptr(mask), ptr(sums),
                            ptr(dza), C, None, ptr(dzb), C, None, ptr(gr[0]), ptr(gr[1]), ptr(gr[2]) if two else None, ptr(gr[3]) if two else None,
                            B, F_, T, C, relu, pool, st()), "bwd")
    sc = float(zar.grad.abs().max())
    assert (dza.float().cpu().reshape(B, F_, T, C) - zar.grad).abs().max() < 1.5e-2 * sc
    assert (gr[0].cpu() - gar.grad).abs().max() < 2e-3 * max(1.0, float(gar.grad.abs().max()))
    assert (gr[1].cpu() - bar.grad).abs().max() < 2e-3 * max(1.0, float(bar.grad.abs().max()))
    if two:
        assert (dzb.float().cpu().reshape(B, F_, T, C) - zbr.grad).abs().max() < 1.5e-2 * float(zbr.grad.abs().max())
        assert (gr[2].cpu() - gbr.grad).abs().max() < 2e-3 * max(1.0, float(gbr.grad.abs().max()))
        assert (gr[3].cpu() - bbr.grad).abs().max() < 2e-3 * max(1.0, float(bbr.grad.abs().max()))


@pytest.mark.parametrize("B,F_,T,Cin,Cout,KH,xp,dp", [(2, 6, 21, 32, 64, 3, 32, 64), (1, 9, 70, 64, 128, 3, 64, 136), (2, 8, 133, 128, 256, 7, 128, 256),
                                                      (3, 5, 17, 64, 64, 3, 72, 64), (1, 3, 64, 128, 128, 3, 128, 128), (2, 4, 65, 32, 128, 3, 64, 128),
                                                      (1, 1, 1, 128, 64, 3, 128, 64), (4, 30, 200, 64, 128, 3, 64, 128)])
def test_conv_weight_gradient_direct(mta, B, F_, T, Cin, Cout, KH, xp, dp):
    """mt_conv_wgrad (transposed LDS reads of the channels-last tensors, two pieces of dz into one accumulator) == torch's conv2d weight
    gradient for every channel tiling of the kernel (64 / 128 / 256 output, 32 / 64 / 128 input channels), ragged T (a partial last
    64-frame tile, T = 64 exactly, T = 1), position pitches wider than the channel count, with and without the remainder piece, and
    for the 1 x 1 skip; twice: bitwise reproducible."""
    from music_transcription_amd import train_step_large as TL
    g = torch.Generator().manual_seed(B * 100 + T + Cin)
    x = _bf(torch.randn(B, F_, T, xp, generator=g))
    dzf = torch.randn(B, F_, T, dp, generator=g)
    hi = _bf(dzf)
    lo = _bf(dzf - hi)
    xd, hid, lod = x.bfloat16().cuda().contiguous(), hi.bfloat16().cuda().contiguous(), lo.bfloat16().cuda().contiguous()
    ph = KH // 2
    xc = x[..., :Cin].permute(0, 3, 1, 2)
    for pieces in (2, 1):
        dz = (hi + lo) if pieces == 2 else hi
        w = torch.zeros(Cout, Cin, KH, 3, requires_grad=True)
        F.conv2d(xc.double(), w.double(), padding=(ph, 1)).backward(dz[..., :Cout].permute(0, 3, 1, 2).double())
        with torch.cuda.device(0):
            out = TL.conv_wgrad_direct(hid, lod if pieces == 2 else None, dp, xd, xp, B, F_, T, Cout, Cin, KH, 3, torch.empty(Cout, Cin, KH, 3, device="cuda"))
            out2 = TL.conv_wgrad_direct(hid, lod if pieces == 2 else None, dp, xd, xp, B, F_, T, Cout, Cin, KH, 3, torch.empty(Cout, Cin, KH, 3, device="cuda"))
        sc = float(w.grad.abs().max())
        err = float((out.cpu() - w.grad).abs().max())
        assert err < 1e-4 * sc + 1e-5, (pieces, err, sc)                # exact bf16 x bf16 products, f32 accumulation
        assert torch.equal(out, out2)
    w1 = torch.zeros(Cout, Cin, 1, 1, requires_grad=True)
    F.conv2d(xc.double(), w1.double()).backward((hi + lo)[..., :Cout].permute(0, 3, 1, 2).double())
    with torch.cuda.device(0):
        out1 = TL.conv_wgrad_direct(hid, lod, dp, xd, xp, B, F_, T, Cout, Cin, 1, 1, torch.empty(Cout, Cin, 1, 1, device="cuda"))
    assert float((out1.cpu() - w1.grad).abs().max()) < 1e-4 * float(w1.grad.abs().max()) + 1e-5


def test_conv_cl_channel_slices_and_accumulate(mta):
    """mt_conv_cl_ex: the 256-channel input gradient of freq_aware_conv as two accumulating calls over channel halves."""
    lib, check, ptr, st = _lib()
    from music_transcription_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    B, F_, T = 2, 9, 37
    dz = _bf(torch.randn(B, F_, T, 256, generator=g))
    w = _bf(torch.randn(256, 128, 7, 3, generator=g) * 0.05)           # forward weights [co][ci][kh][kw]
    def _conv_dgrad_w(w_):          # [ci][(kh'*3 + kw')*Cout + co] = w[co][ci][KH-1-kh'][2-kw']
        return w_.flip(2, 3).permute(1, 2, 3, 0).reshape(w_.shape[1], -1)
    wa, wb = _conv_dgrad_w(w[:128]).bfloat16().cuda().contiguous(), _conv_dgrad_w(w[128:]).bfloat16().cuda().contiguous()
    dzd = dz.bfloat16().cuda().contiguous()
    out = torch.empty(B * F_ * T, 128, dtype=torch.bfloat16, device="cuda")
    zero = torch.zeros(256, device="cuda")
    check(lib.mt_conv_cl_ex(ptr(dzd), 256, None, 0, ptr(wa), ptr(zero), ptr(out), B, F_, T, 128, 0, 128, 7, 0, 0, 0, 0, 0, L.DT_BF16, st()))
    check(lib.mt_conv_cl_ex(ptr(dzd.reshape(-1)[128:]), 256, None, 0, ptr(wb), ptr(zero), ptr(out), B, F_, T, 128, 0, 128, 7, 0, 0, 0, 0, 1,
                            L.DT_BF16, st()))
    xin = torch.zeros(B, 128, F_, T, requires_grad=True)
    F.conv2d(xin, w, padding=(3, 1)).backward(dz.permute(0, 3, 1, 2))
    ref = xin.grad.permute(0, 2, 3, 1)
    assert (out.float().cpu().reshape(B, F_, T, 128) - ref).abs().max() < 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_attention_softmax_forward_backward_vs_autograd(mta, p):
    lib, check, ptr, st = _lib()
    g = torch.Generator().manual_seed(3)
    rows, T, Tp = 10, 100, 128
    S = torch.randn(rows, Tp, generator=g) * 40.0                       # scaled scores reach well beyond the +-10 clamp
    scale, clip = 0.2, 10.0
    Sd = S.cuda()
    P = torch.empty(rows, Tp, dtype=torch.bfloat16, device="cuda")
    check(lib.mt_attn_softmax_train(ptr(Sd), Tp, ptr(P), Tp, T, rows, scale, clip, p, 77, 200, st()))
    Sr = S[:, :T].clone().requires_grad_(True)
    A = torch.softmax(torch.clamp(Sr * scale, -clip, clip), dim=-1)
    got = P.float().cpu()
    assert torch.all(got[:, T:] == 0)
    if p == 0.0:
        assert (got[:, :T] - A.detach()).abs().max() < 4e-3 * float(A.max())
        keep = torch.ones(rows, T)
    else:
        keep = (got[:, :T] != 0).float()                                # the kernel's mask (regenerated by the backward kernel)
        assert abs(float(keep.mean()) - (1 - p)) < 0.08
        assert (got[:, :T] - A.detach() * keep / (1 - p)).abs().max() < 8e-3 * float(A.max()) / (1 - p)
    dPd = torch.randn(rows, Tp, generator=g)
    (A * keep / (1 - p)).backward(dPd[:, :T])
    dS = torch.empty(rows, Tp, dtype=torch.bfloat16, device="cuda")
    check(lib.mt_attn_clamped_bwd(ptr(Sd), Tp, ptr(dPd.cuda()), Tp, ptr(dS), Tp, T, rows, scale, clip, p, 77, 200, st()))
    gd = dS.float().cpu()
    assert torch.all(gd[:, T:] == 0)
    assert (gd[:, :T] - Sr.grad).abs().max() < 1e-2 * float(Sr.grad.abs().max())
    outside = (Sr.detach() * scale).abs() > clip
    assert outside.float().mean() > 0.1 and torch.all(gd[:, :T][outside] == 0)        # the clamp's zero-gradient region is exercised


def test_layernorm_residual_forward_backward_vs_autograd(mta):
    lib, check, ptr, st = _lib()
    g = torch.Generator().manual_seed(9)
    rows, n, ld = 300, 48, 64
    a, pj = torch.randn(rows, n, generator=g), torch.randn(rows, n, generator=g) * 0.5
    gam, bet = 1.0 + 0.2 * torch.randn(n, generator=g), 0.1 * torch.randn(n, generator=g)
    y = torch.zeros(rows, ld, dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(rows, 2, device="cuda")
    ad, pd_, gd, bd = a.cuda(), pj.cuda(), gam.cuda(), bet.cuda()
    check(lib.mt_layernorm_residual_train(ptr(ad), n, ptr(pd_), n, ptr(gd), ptr(bd), ptr(y), ld, ptr(stats), rows, n, 1e-6, st()))
    ar, gr, br = a.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = F.layer_norm(ar + pj, (n,), gr, br, 1e-6)
    assert (y.float().cpu()[:, :n] - ref.detach()).abs().max() < 2e-2
    dy = torch.randn(rows, n, generator=g)
    ref.backward(dy)
    nsl = lib.mt_layernorm_residual_bwd_slices()
    dx = torch.empty(rows, n, device="cuda")
    part = torch.zeros(nsl, 2, n, device="cuda")
    check(lib.mt_layernorm_residual_bwd(ptr(ad), n, ptr(pd_), n, ptr(gd), ptr(stats), ptr(dy.cuda()), n, ptr(dx), n, ptr(part), rows, n, st()))
    assert (dx.cpu() - ar.grad).abs().max() < 1e-4 * max(1.0, float(ar.grad.abs().max()))
    red = part.sum(0).cpu()
    assert (red[0] - gr.grad).abs().max() < 1e-3 and (red[1] - br.grad).abs().max() < 1e-3


# ------------------------------------------------------------------------------------------------ whole training step
def _golden_batches(g):
    nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]
    out = []
    for k in range(nb):
        mel, roll = _mel_in(B, nm, T, sx + k), _roll_in(B, T, sx + 100 + k, 0.1)
        lengths = torch.tensor([T, T - 7, T - 15][:B], dtype=torch.int64)
        for b in range(B):
            mel[b, :, :, lengths[b]:] = 0.0
            roll[b, :, lengths[b]:] = 0.0
        out.append((mel, roll, lengths))
    return out


def _hip_large(mta, nm, H, L, seed, dropout=0.0, p2d=(0.0, 0.0, 0.0), **kw):
    m = mta.TranscriptionModel(model_type="cnn_rnn_large", n_mels=nm, hidden_size=H, num_layers=L, dropout=dropout, device="cuda", **kw)
    sd = R.make_state_dict("cnn_rnn_large", nm, H, L, seed, use_attention=kw.get("use_attention", True),
                           use_heads=kw.get("use_onset_offset_heads", True))
    m.load_state_dict(sd, strict=True)
    m.model.dropout2d_p = p2d
    return m, sd


def _compare_grads(named_grads, ref, prefix="model."):
    flat_a, flat_b, worst = [], [], {}
    for k, gr in ref.items():
        b = np.asarray(gr, dtype=np.float32)
        if named_grads[k] is None:             # no gradient path (torch: .grad stays None): the reference's must be all zero
            assert np.abs(b).max() == 0.0, k
            continue
        a = named_grads[k].detach().float().cpu().numpy()
        assert a.shape == b.shape, (k, a.shape, b.shape)
        short = k[len(prefix):] if k.startswith(prefix) else k
        if short in ZERO_GRAD:
            continue
        scale = np.abs(b).max()
        if scale == 0.0:                       # a head without gradient (frame-only loss): must be exactly zero here too
            assert np.abs(a).max() == 0.0, k
            continue
        worst[k] = float(np.abs(a - b).max() / scale)
        flat_a.append(a.ravel()); flat_b.append(b.ravel())
    fa, fb = np.concatenate(flat_a), np.concatenate(flat_b)
    return worst, float(fa @ fb / (np.linalg.norm(fa) * np.linalg.norm(fb)))


def _oracle_grads(sd, mel, roll, lengths, emulate_bf16, all_heads=False):
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = [k for k, v in sdo.items() if v.dtype.is_floating_point and "running_" not in k]
    for k in keys:
        sdo[k].requires_grad_(True)
    lo = R.cnnrnn_large_forward(sdo, mel, return_all_heads=all_heads, o=R.Opts(gemm_bf16=emulate_bf16), train=True)
    R.compute_loss(lo, roll, lengths).backward()
    grads = {k: (sdo[k].grad.numpy() if sdo[k].grad is not None else np.zeros(tuple(sdo[k].shape), np.float32)) for k in keys}
    return lo, grads


def _oracle_noise_floor(sd, mel, roll, lengths, ref, all_heads=False, amp=FLOOR_AMP, seeds=FLOOR_SEEDS):
    """The emulating oracle's distance from ITSELF when every 16-bit rounding of an activation is preceded by a relative perturbation of
    `amp` (2^-20: an f32 re-association; parameters are not perturbed): {tensor: max over the seeds of max |g' - g| / max |g|}, the smallest
    cosine, the largest logit distance.  `ref` = the unperturbed oracle's gradients."""
    floor, cos_min, dl_max = {}, 1.0, 0.0
    orig = R._bf16_round
    lo0 = None
    for seed in (None,) + tuple(seeds):
        gen = torch.Generator().manual_seed(seed or 0)

        def dithered(x, gen=gen, on=seed is not None):
            if on and not (x.is_leaf and x.requires_grad):
                x = x + x.detach() * (amp * (2.0 * torch.rand(x.shape, generator=gen) - 1.0))
            return orig(x)
        R._bf16_round = dithered
        try:
            lo, g1 = _oracle_grads(sd, mel, roll, lengths, True, all_heads)
        finally:
            R._bf16_round = orig
        lo = lo["frame"] if isinstance(lo, dict) else lo
        if seed is None:
            lo0 = lo.detach()
            continue
        w, c = _compare_grads({k: torch.from_numpy(v) for k, v in g1.items()}, ref)
        for k, v in w.items():
            floor[k] = max(floor.get(k, 0.0), v)
        cos_min, dl_max = min(cos_min, c), max(dl_max, float((lo.detach() - lo0).abs().max()))
    return floor, cos_min, dl_max


def _bound(floor, k):
    return max(GRAD_REL, FLOOR_FACTOR * floor.get(k, 0.0) + FLOOR_MARGIN)


def _report(tag, worst, cos):
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:12]
    print(f"\n[{tag}] cos={cos:.6f} worst: " + ", ".join(f"{k.replace('model.', '')}={v:.3g}" for k, v in top))


def test_large_train_step_matches_reference_golden(mta, golden_dir):
    g = np.load(os.path.join(golden_dir, "train_step_large.npz"))
    nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]
    data = _golden_batches(g)
    m, sd = _hip_large(mta, nm, H, L, sw)
    m.train()
    mel, roll, lengths = data[0]
    logits = m(mel.cuda())
    assert logits.requires_grad and logits.shape == (B, 88, T)
    assert np.abs(logits.detach().cpu().numpy() - g["logits0"]).max() < LOGIT_TOL_TRAIN_FP32
    loss = m.compute_loss(logits, roll.cuda(), lengths)
    assert abs(loss.item() - float(g["loss0"])) < 2e-3
    loss.backward()
    m.model.raise_on_train_handoff_timeout()
    grads = {k: p.grad for k, p in m.named_parameters()}
    ref = {k[len("grad::"):]: g[k] for k in g.files if k.startswith("grad::")}
    assert set(ref) == set(grads)
    worst32, cos32 = _compare_grads(grads, ref)
    _report("large vs fp32 reference golden", worst32, cos32)
    lo_emu, ref_emu = _oracle_grads(sd, mel, roll, lengths, True)
    print("logits vs emulating oracle:", float((logits.detach().cpu() - lo_emu.detach()).abs().max()))
    worst, cos = _compare_grads(grads, ref_emu)
    _report("large vs bf16-emulating oracle", worst, cos)
    we, ce = _compare_grads({k: torch.from_numpy(v) for k, v in ref_emu.items()}, ref)
    _report("(bf16-emulating oracle vs fp32 reference golden)", we, ce)
    floor, cos_floor, dl_floor = _oracle_noise_floor(sd, mel, roll, lengths, ref_emu)
    _report(f"(emulating oracle vs ITSELF under 2^-20 perturbations of its roundings: logits {dl_floor:.3f})", floor, cos_floor)
    bad = {k: (v, we[k]) for k, v in worst32.items() if v > we[k] + _bound(floor, k)}
    assert not bad and cos32 > GRAD_COS_FP32 and cos32 > ce - 2e-3, (bad, cos32, ce)
    dd = (logits.detach().cpu() - lo_emu.detach()).abs()
    assert float(dd.mean()) < 4e-3 and float(dd.max()) < 2.5 * LOGIT_TOL_TRAIN_EMU
    bad = {k: (v, floor.get(k)) for k, v in worst.items() if v > _bound(floor, k)}
    assert not bad and cos > min(GRAD_COS, cos_floor - 1e-3), (bad, cos, cos_floor)
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    # the frame-only loss leaves the onset / offset heads out of the graph (train_transcriber.py:119): .grad None, as in torch
    assert all(p.grad is None for k, p in m.named_parameters() if "onset_head" in k or "offset_head" in k)
    assert abs(gn - float(g["gradnorm0"])) < 3e-2 * float(g["gradnorm0"])
    sdm = m.state_dict()
    for k in g.files:                        # BatchNorm running statistics after the step's forward
        if k.startswith("bn0::") and "num_batches" not in k:
            a, b = sdm[k[len("bn0::"):]].cpu().numpy(), g[k]
            assert np.abs(a - b).max() <= 3e-3 * max(np.abs(b).max(), 1.0), k
        elif k.startswith("bn0::"):
            assert int(sdm[k[len("bn0::"):]]) == int(g[k])


def test_large_gradients_at_a_realistic_position_count(mta):
    """n_mels = 320, B = 2, T = 200 -- 64 000 positions per BatchNorm channel in the first block, 32 000 in the 7x3 conv -- against the
    oracle with the same bf16 rounding points.  The per-tensor distance does not shrink to GRAD_REL with the position count: conv weight
    gradients 9 - 11 %, BatchNorm affine gradients 10 - 18 %, cosine 0.9977.  Round 4 found why (header of this file): that IS the oracle's
    distance from itself under 2^-20 perturbations of its roundings (10 - 19 %, cosine 0.9977), while every backward stage alone is within
    0.2 - 1 % of autograd (tests/tools/large_grad_debug.py, profiles/r04_large_grad_stage_errors.txt).  Asserted: the HIP path is within the
    measured floor, tensor by tensor, at least as close in cosine, and its logits are no further away than the perturbed oracle's."""
    nm, H, L, B, T = 320, 64, 2, 2, 200
    m, sd = _hip_large(mta, nm, H, L, 21)
    m.train()
    g = torch.Generator().manual_seed(5)
    mel = _mel_in(B, nm, T, 31)
    roll = (torch.rand(B, 88, T, generator=g) < 0.1).float()
    lengths = torch.tensor([T, T - 37], dtype=torch.int64)
    mel[1, :, :, T - 37:] = 0.0
    roll[1, :, T - 37:] = 0.0
    logits = m(mel.cuda())
    loss = m.compute_loss(logits, roll.cuda(), lengths)
    loss.backward()
    m.model.raise_on_train_handoff_timeout()
    grads = {k: p.grad for k, p in m.named_parameters()}
    lo_emu, ref_emu = _oracle_grads(sd, mel, roll, lengths, True)
    dl = float((logits.detach().cpu() - lo_emu.detach()).abs().max())
    worst, cos = _compare_grads(grads, ref_emu)
    _report(f"large 320/64/2, B = 2, T = 200 vs bf16-emulating oracle (logits {dl:.3f})", worst, cos)
    floor, cos_floor, dl_floor = _oracle_noise_floor(sd, mel, roll, lengths, ref_emu)
    _report(f"(emulating oracle vs ITSELF under 2^-20 perturbations of its roundings: logits {dl_floor:.3f})", floor, cos_floor)
    assert dl < 2.5 * LOGIT_TOL_TRAIN_EMU and dl < 2.0 * dl_floor + 0.01, (dl, dl_floor)
    bad = {k: (v, floor.get(k)) for k, v in worst.items() if v > _bound(floor, k)}
    assert not bad and cos > cos_floor - 1e-3 and cos > 0.997, (bad, cos, cos_floor)
    assert max(worst.values()) < 0.25                      # (and an absolute cap, whatever the floor's draws say)


def test_large_training_loop_matches_reference_losses(mta, golden_dir):
    """train_one_epoch of this package over the golden's 3 batches, CNNRNNModelLarge (what example.sh:22 trains)."""
    g = np.load(os.path.join(golden_dir, "train_step_large.npz"))
    nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]
    data = [(a.cuda(), b.cuda(), c) for a, b, c in _golden_batches(g)]
    m, _ = _hip_large(mta, nm, H, L, sw)
    opt = mta.make_optimizer(m, lr=float(g["lr"]))
    avg, losses = mta.train_one_epoch(m, data, opt, torch.device("cuda"), max_grad_norm=1.0)
    assert np.abs(np.array(losses) - g["losses"]).max() < 3e-3, (losses, g["losses"])
    assert abs(avg - float(g["avg_loss"])) < 3e-3
    # Parameters after the reference's own train_one_epoch (`post::`): the onset / offset heads never enter its graph, so
    # torch.optim.Adam leaves them bit-for-bit alone (no weight decay either) -- and so must the fused step; every other
    # parameter moved by at most a few Adam steps of lr from the reference's.
    sd0 = R.make_state_dict("cnn_rnn_large", nm, H, L, sw)
    sdm = m.state_dict()
    lr, nsteps = float(g["lr"]), len(losses)
    for k in g.files:
        if not k.startswith("post::") or "running_" in k or "num_batches" in k:
            continue
        name = k[len("post::"):]
        got, want = sdm[name].detach().float().cpu().numpy(), g[k]
        if "onset_head" in name or "offset_head" in name:
            assert np.array_equal(want, sd0[name].numpy()), name            # the reference did not touch them ...
            assert np.array_equal(got, want), name                          # ... and neither did the fused optimizer
        else:
            assert np.abs(got - want).max() <= 2.0 * nsteps * lr + 1e-7, (name, float(np.abs(got - want).max()))
    # ... and every trained tensor MOVED the way the reference's did (the bound above alone passes for weights that never changed)
    from test_gpu_train import _update_cosines, UPDATE_COS_ALL_LARGE as UPDATE_COS_ALL, UPDATE_COS_TENSOR_LARGE as UPDATE_COS_TENSOR
    cos_t, cos_all = _update_cosines({k[len("post::"):]: g[k] for k in g.files if k.startswith("post::") and "running_" not in k and "num_batches" not in k
                                      and "onset_head" not in k and "offset_head" not in k
                                      # (convolution biases in front of a BatchNorm: analytically zero gradient, Adam turns rounding noise into steps)
                                      and not re.search(r"(conv1\.0|freq_aware_conv\.0|res_block\d\.(conv1|conv2|skip\.0))\.bias$", k)}, sdm, sd0)
    print("\n[update direction, CNNRNNModelLarge] all=%.4f worst: " % cos_all + ", ".join(f"{k.replace('model.', '')}={v:.3f}" for k, v in sorted(cos_t.items(), key=lambda kv: kv[1])[:6]))
    assert cos_all >= UPDATE_COS_ALL and min(cos_t.values()) >= UPDATE_COS_TENSOR, (cos_all, {k: v for k, v in cos_t.items() if v < UPDATE_COS_TENSOR})
    # the packed inference weights follow the optimizer, and eval mode agrees with the oracle on the trained weights
    m.eval()
    with torch.no_grad():
        after = m(data[0][0]).cpu()
        sdo = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        ref = R.cnnrnn_large_forward(sdo, data[0][0].cpu())
    assert (after - ref).abs().max() < LOGIT_TOL


@pytest.mark.parametrize("kw,all_heads", [(dict(), True), (dict(use_attention=False), False), (dict(use_onset_offset_heads=False), False)])
def test_large_variants_and_dict_loss_vs_oracle_autograd(mta, kw, all_heads):
    """The dict loss path (frame / onset / offset, transcription_model.py:164-194) and the --no_attention / single-head
    variants against torch autograd on the CPU oracle; other shapes than the golden (padded hidden sizes, odd T)."""
    nm, H, L, B, T = 64, 24, 2, 4, 93          # (several thousand positions per BatchNorm channel: flips of single ReLU / pool decisions weigh less)
    m, sd = _hip_large(mta, nm, H, L, seed=31, **kw)
    m.train()
    mel, roll = _mel_in(B, nm, T, 3), _roll_in(B, T, 4, 0.1)
    lengths = torch.tensor([T, T - 6, T - 20, T - 41], dtype=torch.int64)
    out = m(mel.cuda(), return_all_heads=all_heads)
    loss = m.compute_loss(out, roll.cuda(), lengths)
    loss.backward()
    m.model.raise_on_train_handoff_timeout()
    grads = {"model." + k: p.grad for k, p in m.model.named_parameters()}
    lo, ref = _oracle_grads(sd, mel, roll, lengths, True, all_heads)
    def close(a, b):      # 1-ulp differences flip a few ReLU / pool decisions: bound the bulk tightly, the outliers loosely
        dd = (a.detach().cpu() - b.detach()).abs()
        return float(dd.mean()) < 6e-3 and float(dd.max()) < 4 * LOGIT_TOL_TRAIN_EMU
    if all_heads:
        for k in ("frame", "onset", "offset"):
            assert close(out[k], lo[k]), k
    else:
        assert close(out, lo)
    worst, cos = _compare_grads(grads, ref)
    _report(f"variant {kw} all_heads={all_heads}", worst, cos)
    floor, cos_floor, _ = _oracle_noise_floor(sd, mel, roll, lengths, ref, all_heads)
    bad = {k: (v, floor.get(k)) for k, v in worst.items() if v > _bound(floor, k)}
    assert not bad and cos > min(GRAD_COS, cos_floor - 1e-3), (bad, cos, cos_floor)


def test_large_train_with_dropout_runs_and_is_seeded(mta):
    nm, H, L, B, T = 32, 16, 2, 3, 24
    m, _ = _hip_large(mta, nm, H, L, seed=5, dropout=0.2, p2d=(0.1, 0.1, 0.15))
    m.train()
    mel, roll = _mel_in(B, nm, T, 1).cuda(), _roll_in(B, T, 2, 0.1).cuda()
    outs = []
    for seed in (1, 1, 2):
        torch.manual_seed(seed)
        for p in m.parameters():
            p.grad = None
        lg = m(mel)
        m.compute_loss(lg, roll).backward()
        gflat = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])
        assert torch.isfinite(lg).all() and torch.isfinite(gflat).all()
        outs.append((lg.detach().clone(), gflat.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-7)     # same seed, same masks
    assert not torch.equal(outs[0][0], outs[2][0])                                                                # another seed, other masks
    m.eval()
    with torch.no_grad():
        ev = m(mel)
    assert not torch.equal(ev, outs[0][0])


def test_large_full_size_training_step(mta):
    """CNNRNNModelLarge 320/512/3 (what example.sh:22 trains), cached-format batch of 4 ragged chunks (T <= 937), every
    dropout of the reference active: logits and every gradient finite, BatchNorm buffers updated, the train-mode forward
    under torch.no_grad() equals the one with the autograd edge for the same seed, and two fused clip + Adam steps at a
    large learning rate lower the loss on the same batch."""
    nm, H, L, B, T = 320, 512, 3, 4, 937
    m = mta.TranscriptionModel(model_type="cnn_rnn_large", n_mels=nm, hidden_size=H, num_layers=L, dropout=0.2, device="cuda")
    m.load_state_dict(R.make_state_dict("cnn_rnn_large", nm, H, L, 9), strict=True)
    g = torch.Generator().manual_seed(3)
    lengths = torch.tensor([T, 700, 469, 900], dtype=torch.int64)
    mel = torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0
    roll = (torch.rand(B, 88, T, generator=g) < 0.04).float()
    for b in range(B):
        mel[b, :, :, lengths[b]:] = 0.0
        roll[b, :, lengths[b]:] = 0.0
    meld, rolld = mel.cuda(), roll.cuda()
    m.train()
    rm0 = m.model.freq_aware_conv[1].running_mean.clone()
    torch.manual_seed(5)
    logits = m(meld)
    loss = m.compute_loss(logits, rolld, lengths)
    loss.backward()
    m.model.raise_on_train_handoff_timeout()
    assert logits.shape == (B, 88, T) and torch.isfinite(logits).all()
    for k, p in m.named_parameters():
        if "onset_head" in k or "offset_head" in k:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k        # frame-only loss (train_transcriber.py:119)
        elif k[len("model."):] in ZERO_GRAD:
            assert p.grad is not None and float(p.grad.abs().max()) < 1e-6, k   # conv bias in front of a BatchNorm: zero up to the rounding of sum(dz)
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, k
    assert not torch.equal(rm0, m.model.freq_aware_conv[1].running_mean)
    torch.manual_seed(5)
    with torch.no_grad():
        again = m(meld)
    assert torch.equal(again, logits.detach())
    opt = mta.make_optimizer(m, lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        ls = m.compute_loss(m(meld), rolld, lengths)
        ls.backward()
        st = opt.step(sync_grads=False).tolist()
        assert st[1] == 1.0 and np.isfinite(st[0])
        losses.append(float(ls.item()))
    assert losses[-1] < losses[0], losses


def test_train_cnn_script_trains_the_large_model(mta, tmp_path):
    """scripts/train_cnn.py --model cnn_rnn_large (what the reference's example.sh:22 trains) on a tiny cache: one epoch runs through the HIP
    training step, the three checkpoint names hold the reference's CNNRNNModelLarge keys, --no_onset_offset_heads drops the head keys."""
    import json
    import subprocess
    import sys
    nm, T = 32, 40
    cache = str(tmp_path / "cache")
    for split, n in (("train", 4), ("validation", 2)):
        for i in range(n):
            mel = _mel_in(1, nm, T - (i % 2) * 6, 400 + i)[0]
            mta.write_cache_chunk(cache, split, i, mel, _roll_in(1, mel.shape[-1], 500 + i, 0.1)[0])
        mta.write_cache_metadata(cache, split, [{} for _ in range(n)], n_mels=nm)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra, heads in (([], True), (["--no_onset_offset_heads"], False)):
        run = str(tmp_path / ("run_h" if heads else "run_nh"))
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "train_cnn.py"), "--cached_dir", cache, "--model", "cnn_rnn_large", "--batch_size", "2",
                            "--epochs", "1", "--lr", "1e-3", "--n_mels", str(nm), "--hidden_size", "16", "--num_layers", "2", "--dropout", "0.1",
                            "--run_dir", run, "--num_workers", "0"] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        hist = json.load(open(os.path.join(run, "history.json")))
        assert len(hist) == 1 and np.isfinite(hist[0]["train_loss"]) and np.isfinite(hist[0]["val_loss"]) and hist[0]["steps"] == 2
        man = R.make_state_dict("cnn_rnn_large", nm, 16, 2, 0, use_heads=heads)
        for name in ("model_epoch_1.pth", "model_best.pth", "model_final.pth"):
            ck = torch.load(os.path.join(run, "checkpoints", name))
            assert set(ck) == set(man) and all(ck[k].shape == man[k].shape for k in man), (name, set(ck) ^ set(man))


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(use_attention=False), dict(use_onset_offset_heads=False)])
def test_pack_job_tables_equal_the_torch_expressions(mta, kw):
    """The operands of the training step as mt_pack_jobs writes them (pack_plan.py, one launch per stream and step) against the torch
    expressions they replace (tests/tools/pack_reference_large.py): every tensor bit for bit, twice -- the second call re-packs moved
    parameters into the SAME destination tensors."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from pack_reference_large import pack_train_large_torch
    from music_transcription_amd.train_step_large import pack_train_large
    m, _ = _hip_large(mta, 64, 24, 3, seed=5, **kw)
    net = m.model

    def flat(t):
        out = {}
        for k, v in t.items():
            if isinstance(v, torch.Tensor):
                out[k] = v
            elif isinstance(v, (list, tuple)):
                for i, u in enumerate(v):
                    if isinstance(u, torch.Tensor):
                        out[f"{k}[{i}]"] = u
        return out

    for rnd in range(2):
        got, want = flat(pack_train_large(net, "cuda")), flat(pack_train_large_torch(net, torch.device("cuda")))
        torch.cuda.synchronize()
        assert set(want) <= set(got), set(want) - set(got)
        for k, w in want.items():
            g = got[k]
            if k == "m_wih[0]" or k == "l_wih[0]":          # (views of one tensor: compare the rows the reference returns)
                g = g[:w.shape[0]]
            assert g.shape == w.shape and g.dtype == w.dtype, (k, g.shape, w.shape, g.dtype, w.dtype)
            assert torch.equal(g, w), (rnd, k, float((g.float() - w.float()).abs().max()))
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn_like(p) * 0.01)
