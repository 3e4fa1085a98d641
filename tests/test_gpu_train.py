"""GPU parity of the training step: the HIP train-mode forward / backward (through the C ABI)
against the reference's own gradients (tests/golden/train_step.npz, written by make_golden_train.py from the
reference's modules and its train_one_epoch) and against torch autograd on the CPU oracle at other shapes.

Tolerances: GEMM/conv operands are bf16 (f32 accumulate) and the backward recurrence exchanges bf16 gate
gradients, so a gradient tensor is compared relative to its own largest entry (GRAD_REL) and the whole flat
gradient by its cosine to the reference (GRAD_COS)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model_ref as R

GRAD_REL = 3e-2        # max |g - g_ref| / max |g_ref| per tensor, against the oracle with the HIP path's bf16 rounding points
GRAD_REL_FP32 = 1e-1   # the same against the fp32 reference (bf16 activations flip a few ReLU / max-pool decisions)
# conv2's weight gradient is a sum with heavy cancellation (BatchNorm makes sum dz = 0 and sum dz*z = 0).  The ~0.07 % of pool
# pairs whose two rows round to the SAME bf16 value used to move it by ~14 % of its largest entry (the gradient went to the
# first row, as on a true tie); the pool now routes by the order of the conv's f32 results before their rounding
# (mt_conv_cl_tie), and conv2's weight gradient meets the common 10 % bound against the fp32 reference (observed 6 %).
GRAD_REL_FP32_BY_KEY = {}                                   # the reference golden: every tensor within GRAD_REL_FP32
GRAD_REL_FP32_BY_KEY_SWEEP = {"model.cnn.4.weight": 1.5e-1}   # the shape sweep below (fewer positions still): observed <= 0.14
GRAD_COS = 0.9995      # cosine of the flat gradient against the oracle with the same rounding points
GRAD_COS_FP32 = 0.998  # ... against the fp32 reference (tiny shapes: a few hundred positions per channel)
LOGIT_TOL = 3e-2
ZERO_GRAD_KEYS = ("cnn.0.bias", "cnn.4.bias")   # conv bias in front of BatchNorm: analytically zero gradient


@pytest.fixture(scope="module")
def mta():
    import music_transcription_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def _mel_in(B, nm, T, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=g))


def _roll_in(B, T, seed, p=0.04):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 88, T, generator=g) < p).float()


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


def _hip_model(mta, nm, H, L, seed, dropout=0.0):
    m = mta.TranscriptionModel(model_type="cnn_rnn", n_mels=nm, hidden_size=H, num_layers=L, dropout=dropout, device="cuda")
    sd = R.make_state_dict("cnn_rnn", nm, H, L, seed)
    m.load_state_dict(sd, strict=True)
    return m, sd


def _compare_grads(named_grads, ref, prefix="model."):
    flat_a, flat_b, worst = [], [], {}
    for k, gr in ref.items():
        a = named_grads[k].detach().float().cpu().numpy()
        b = np.asarray(gr, dtype=np.float32)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        short = k[len(prefix):] if k.startswith(prefix) else k
        if short in ZERO_GRAD_KEYS:
            continue
        scale = np.abs(b).max()
        worst[k] = float(np.abs(a - b).max() / max(scale, 1e-12))
        flat_a.append(a.ravel()); flat_b.append(b.ravel())
    fa, fb = np.concatenate(flat_a), np.concatenate(flat_b)
    cos = float(fa @ fb / (np.linalg.norm(fa) * np.linalg.norm(fb)))
    return worst, cos


def _oracle_grads(sd, mel, roll, lengths, emulate_bf16):
    """torch autograd through the CPU oracle's train-mode forward; optionally with the HIP path's rounding points."""
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = [k for k, v in sdo.items() if v.dtype.is_floating_point and "running_" not in k]
    for k in keys:
        sdo[k].requires_grad_(True)
    lo = R.cnnrnn_forward(sdo, mel, R.Opts(gemm_bf16=emulate_bf16), train=True)
    R.compute_loss(lo, roll, lengths).backward()
    return lo.detach(), {k: sdo[k].grad.numpy() for k in keys}


def test_train_step_matches_reference_golden(mta, golden_dir):
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]
    data = _golden_batches(g)
    m, _ = _hip_model(mta, nm, H, L, sw)
    m.train()
    mel, roll, lengths = data[0]
    logits = m(mel.cuda())
    assert logits.requires_grad
    assert np.abs(logits.detach().cpu().numpy() - g["logits0"]).max() < LOGIT_TOL
    loss = m.compute_loss(logits, roll.cuda(), lengths)
    assert abs(loss.item() - float(g["loss0"])) < 2e-3
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters()}
    ref = {k[len("grad::"):]: g[k] for k in g.files if k.startswith("grad::")}
    assert set(ref) == set(grads)
    worst, cos = _compare_grads(grads, ref)
    print("\n[small vs fp32 golden] cos=%.6f " % cos + ", ".join(f"{k.replace('model.', '')}={v:.3g}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]))
    bad = {k: v for k, v in worst.items() if v > GRAD_REL_FP32_BY_KEY.get(k, GRAD_REL_FP32)}
    assert not bad and cos > GRAD_COS_FP32, (bad, cos)
    _, ref_emu = _oracle_grads(R.make_state_dict("cnn_rnn", nm, H, L, sw), mel, roll, lengths, True)
    worst, cos = _compare_grads(grads, ref_emu)
    bad = {k: v for k, v in worst.items() if v > GRAD_REL}
    assert not bad and cos > GRAD_COS, (bad, cos)
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    assert abs(gn - float(g["gradnorm0"])) < 2e-2 * float(g["gradnorm0"])
    # BatchNorm running statistics after the step's forward
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("bn0::") and "num_batches" not in k:
            a, b = sd[k[len("bn0::"):]].cpu().numpy(), g[k]
            assert np.abs(a - b).max() <= 2e-3 * max(np.abs(b).max(), 1.0), k
        elif k.startswith("bn0::"):
            assert int(sd[k[len("bn0::"):]]) == int(g[k])


# Adam's first steps are ~lr * sign(g) per element, so an element whose gradient is small against the 16-bit rounding of the
# backward pass may step the other way: the per-tensor cosine of the UPDATE is lower than the gradient's.  0.0 = did not move.
# Measured (round 4): CNNRNNModel 0.9992 over all tensors, worst tensor 0.993; CNNRNNModelLarge 0.979 / 0.922 (its conv-stack
# gradients carry the LSTM input gradient's noise through the BatchNorm projections: tests/tools/large_grad_debug.py).
UPDATE_COS_ALL, UPDATE_COS_TENSOR = 0.995, 0.98
UPDATE_COS_ALL_LARGE, UPDATE_COS_TENSOR_LARGE = 0.97, 0.90


def _update_cosines(post_ref, sd_got, sd_init):
    """{tensor: cos(got - w0, want - w0)}, and the cosine over all tensors together."""
    out, num, na, nb = {}, 0.0, 0.0, 0.0
    for name, want in post_ref.items():
        w0 = sd_init[name].double().numpy().ravel()
        da = sd_got[name].detach().double().cpu().numpy().ravel() - w0
        db = np.asarray(want, dtype=np.float64).ravel() - w0
        d = float(np.linalg.norm(da) * np.linalg.norm(db))
        out[name] = float(da @ db) / d if d > 0 else 0.0
        num += float(da @ db); na += float(da @ da); nb += float(db @ db)
    return out, (num / np.sqrt(na * nb) if na > 0 and nb > 0 else 0.0)


def test_training_loop_matches_reference_losses(mta, golden_dir):
    """train_one_epoch of this package (NaN guards, clip 1.0, Adam) over the golden's 3 batches."""
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]
    data = [(a.cuda(), b.cuda(), c) for a, b, c in _golden_batches(g)]
    m, _ = _hip_model(mta, nm, H, L, sw)
    opt = mta.make_optimizer(m, lr=float(g["lr"]))
    avg, losses = mta.train_one_epoch(m, data, opt, torch.device("cuda"), max_grad_norm=1.0)
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-3, (losses, g["losses"])
    assert abs(avg - float(g["avg_loss"])) < 2e-3
    sd = m.state_dict()
    for k in g.files:
        if not k.startswith("post::") or "num_batches" in k:
            continue
        name = k[len("post::"):]
        if name[len("model."):] in ZERO_GRAD_KEYS:
            continue            # zero-gradient parameters: Adam turns rounding noise into +-lr steps (in the reference too)
        a, b = sd[name].float().cpu().numpy(), g[k]
        # 3 Adam steps of lr move a weight by <= 3 lr: agreement to a fraction of that
        assert np.abs(a - b).max() <= 3.2 * float(g["lr"]) + 2e-3 * np.abs(b).max(), name
    # ... and the model MOVED the way the reference's did (the bound above alone would pass for weights that never changed): the
    # update (post - initial) against the reference's own, per tensor and over all of them
    sd0 = R.make_state_dict("cnn_rnn", nm, H, L, sw)
    cos_t, cos_all = _update_cosines({k[len("post::"):]: g[k] for k in g.files if k.startswith("post::") and "num_batches" not in k and "running_" not in k
                                      and k[len("post::model."):] not in ZERO_GRAD_KEYS}, sd, sd0)
    print("\n[update direction, CNNRNNModel] all=%.4f worst: " % cos_all + ", ".join(f"{k.replace('model.', '')}={v:.3f}" for k, v in sorted(cos_t.items(), key=lambda kv: kv[1])[:5]))
    assert cos_all >= UPDATE_COS_ALL and min(cos_t.values()) >= UPDATE_COS_TENSOR, (cos_all, {k: v for k, v in cos_t.items() if v < UPDATE_COS_TENSOR})


def test_eval_after_training_uses_updated_weights(mta, golden_dir):
    """The fused optimizer writes parameters through raw pointers: the packed inference weights must follow."""
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]
    data = [(a.cuda(), b.cuda(), c) for a, b, c in _golden_batches(g)]
    m, _ = _hip_model(mta, nm, H, L, sw)
    x = data[0][0]
    m.eval()
    with torch.no_grad():
        before = m(x).clone()
    opt = mta.make_optimizer(m, lr=1e-2)
    mta.train_one_epoch(m, data, opt, torch.device("cuda"))
    m.eval()
    with torch.no_grad():
        after = m(x).clone()
    assert (after - before).abs().max() > 1e-3
    fresh = mta.TranscriptionModel(model_type="cnn_rnn", n_mels=nm, hidden_size=H, num_layers=L, device="cuda")
    fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    fresh.eval()
    with torch.no_grad():
        again = fresh(x)
    assert torch.equal(after, again)
    # and the oracle agrees on the trained weights
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = R.cnnrnn_forward(sd, x.cpu())
    assert (after.cpu() - ref).abs().max() < LOGIT_TOL


def test_train_cnn_script_two_epochs(mta, tmp_path):
    """scripts/train_cnn.py on a tiny cache: runs, writes a reference-loadable checkpoint, loss goes down."""
    import json
    import subprocess
    import sys
    nm, T = 32, 40
    cache = str(tmp_path / "cache")
    for split, n in (("train", 8), ("validation", 3)):
        for i in range(n):
            mel = _mel_in(1, nm, T - (i % 3) * 5, 200 + i)[0]
            roll = _roll_in(1, mel.shape[-1], 300 + i, 0.1)[0]
            mta.write_cache_chunk(cache, split, i, mel, roll)
        mta.write_cache_metadata(cache, split, [{} for _ in range(n)], n_mels=nm)
    run = str(tmp_path / "run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "train_cnn.py"), "--cached_dir", cache, "--batch_size", "4",
                        "--epochs", "3", "--lr", "3e-3", "--n_mels", str(nm), "--hidden_size", "16", "--num_layers", "2",
                        "--dropout", "0.1", "--run_dir", run, "--save_every", "3", "--num_workers", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    hist = json.load(open(os.path.join(run, "history.json")))
    assert len(hist) == 3 and hist[-1]["train_loss"] < hist[0]["train_loss"]
    # the reference's three checkpoint names (scripts/train_cnn.py:345-358), each a plain state_dict with the reference's keys
    man = R.make_state_dict("cnn_rnn", nm, 16, 2, 0)
    cks = {}
    for name in ("model_epoch_3.pth", "model_best.pth", "model_final.pth"):
        ck = cks[name] = torch.load(os.path.join(run, "checkpoints", name))
        assert set(ck) == set(man) and all(ck[k].shape == man[k].shape and ck[k].device.type == "cpu" for k in man), name
    assert all(torch.equal(cks["model_final.pth"][k], cks["model_epoch_3.pth"][k]) for k in man)      # final = the last epoch's weights
    best_epoch = min(hist, key=lambda h: h["val_loss"])["epoch"]                                      # best = lowest validation loss
    assert "New best model saved" in r.stdout and (best_epoch == 3) == all(torch.equal(cks["model_best.pth"][k], cks["model_final.pth"][k]) for k in man)
    # scripts/evaluate.py --headless on that checkpoint: exactly one line, EVAL_MEAN_F1=<the oracle's mean framewise F1>
    e = subprocess.run([sys.executable, os.path.join(root, "scripts", "evaluate.py"), "--model", os.path.join(run, "checkpoints", "model_best.pth"),
                        "--cache_dir", cache, "--split", "validation", "--model_type", "cnn_rnn", "--hidden_size", "16", "--num_layers", "2",
                        "--threshold", "0.3", "--headless"], capture_output=True, text=True, timeout=300)
    assert e.returncode == 0, e.stderr[-2000:]
    out = e.stdout.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("EVAL_MEAN_F1=") and len(out[0].split("=")[1].split(".")[1]) == 6, e.stdout
    sd = cks["model_best.pth"]
    ds = mta.CachedMaestroDataset(cache, "validation")
    f1s, near = [], 0
    for i in range(len(ds)):
        mel, roll = ds[i]
        with torch.no_grad():
            lg = R.cnnrnn_forward(sd, mel[None])
        near += int(((lg - float(np.log(0.3 / 0.7))).abs() < 3e-2).sum())
        f1s.append(R.f1_binary(roll.numpy(), R.predict(lg, 0.3)[0].numpy()))
    got = float(out[0].split("=")[1])
    # (cells whose fp32 logit sits within the f16 path's tolerance of the threshold may fall on either side)
    assert abs(got - float(np.mean(f1s))) <= (1e-6 if near == 0 else 0.02), (got, float(np.mean(f1s)), near)
    from music_transcription_amd import evaluate as E                                               # the library call behind the script: same number
    hm = mta.TranscriptionModel("cnn_rnn", n_mels=nm, hidden_size=16, num_layers=2, device="cuda").eval()
    hm.load_state_dict(sd)
    assert abs(E.evaluate_dataset(hm, ds, 0.3, "cuda")[0] - got) < 1e-6
    e2 = subprocess.run([sys.executable, os.path.join(root, "scripts", "evaluate.py"), "--model", str(tmp_path / "missing.pth"), "--cache_dir", cache,
                         "--headless"], capture_output=True, text=True, timeout=120)
    assert e2.returncode == 1 and "not found" in e2.stdout


@pytest.mark.parametrize("nm,H,L,B,T", [(40, 32, 3, 2, 33), (64, 48, 2, 5, 21), (32, 24, 1, 34, 12),
                                        (37, 16, 2, 2, 19), (38, 16, 1, 3, 21)])      # odd n_mels / odd n_mels//2: un-pooled tail rows
def test_train_grads_match_oracle_autograd(mta, nm, H, L, B, T):
    """Other shapes (padded hidden sizes, > 1 batch group, odd T) against torch autograd on the CPU oracle."""
    m, sd = _hip_model(mta, nm, H, L, seed=77)
    m.train()
    mel, roll = _mel_in(B, nm, T, 9), _roll_in(B, T, 10, 0.1)
    lengths = torch.tensor([max(1, T - 3 * (b % 4)) for b in range(B)], dtype=torch.int64)
    logits = m(mel.cuda())
    loss = m.compute_loss(logits, roll.cuda(), lengths)
    loss.backward()
    grads = {"model." + k: p.grad for k, p in m.model.named_parameters()}
    lo, ref = _oracle_grads(sd, mel, roll, lengths, False)
    assert np.abs(logits.detach().cpu().numpy() - lo.numpy()).max() < LOGIT_TOL
    worst, cos = _compare_grads(grads, ref)
    bad = {k: v for k, v in worst.items() if v > GRAD_REL_FP32_BY_KEY_SWEEP.get(k, GRAD_REL_FP32)}
    assert not bad and cos > GRAD_COS_FP32, (bad, cos)
    lo, ref = _oracle_grads(sd, mel, roll, lengths, True)
    assert np.abs(logits.detach().cpu().numpy() - lo.numpy()).max() < 5e-3
    worst, cos = _compare_grads(grads, ref)
    bad = {k: v for k, v in worst.items() if v > GRAD_REL}
    assert not bad and cos > GRAD_COS, (bad, cos)


@pytest.mark.parametrize("B", [16, 9, 24, 40])
def test_lstm_bptt_canonical_width(mta, B):
    """The backward recurrence at the canonical hidden size (H = 512: 16 workgroups per direction) on a short
    sequence, against torch autograd through the oracle's explicit LSTM loop.  B <= 16 runs the one-cell-per-thread variant of the
    kernel, 24 the two-cell one with a ragged batch group, 40 two batch groups."""
    from music_transcription_amd import _lib
    from music_transcription_amd._lib import lib, check, ptr
    torch.manual_seed(3)
    T, H, K = 24, 512, 64
    x = torch.randn(B, T, K) * 0.5
    w_ih = (torch.rand(2, 4 * H, K) - 0.5) * 0.08
    w_hh = (torch.rand(2, 4 * H, H) - 0.5) * 0.08
    bias = (torch.rand(2, 4 * H) - 0.5) * 0.08
    dy = torch.randn(B, T, 2 * H) * 0.1
    # oracle
    xo = x.clone().requires_grad_(True)
    who = w_hh.clone().requires_grad_(True)
    wio = w_ih.clone().requires_grad_(True)
    outs = [R.lstm_dir(xo, wio[d], who[d], bias[d], torch.zeros(4 * H), bool(d), R.Opts()) for d in range(2)]
    y = torch.cat(outs, -1)
    (y * dy).sum().backward()
    # HIP: forward (train) + backward through the C ABI
    dev = "cuda"
    M, Mpad = T * B, (T * B + 127) // 128 * 128
    X = torch.zeros(Mpad, K, dtype=torch.bfloat16, device=dev)
    X[:M] = x.permute(1, 0, 2).reshape(M, K).to(torch.bfloat16).to(dev)
    W = w_ih.reshape(8 * H, K).to(torch.bfloat16).to(dev).contiguous()
    gx = torch.empty(lib.mt_lstm_gx_bytes(B, T, H) // 4, device=dev)
    cx = torch.empty(lib.mt_lstm_cx_bytes(B, T, H) // 4, device=dev)
    hx = torch.empty(lib.mt_lstm_hx_bytes(B, T, H) // 4, device=dev)
    sync = torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device=dev)
    bg = bias.reshape(-1).to(dev).contiguous()
    whd = w_hh.to(dev).contiguous()
    st = _lib.stream_ptr()
    check(lib.mt_gemm_lstm_gx(ptr(X), K, ptr(W), K, ptr(bg), ptr(gx), B, T, H, K, st))
    check(lib.mt_lstm_bidir_fwd_train(ptr(gx), ptr(whd), ptr(hx), ptr(cx), ptr(sync), sync.numel(), B, T, H, st))
    yh = torch.empty(B, T, 2 * H, device=dev)
    check(lib.mt_lstm_unpack_f32(ptr(hx), ptr(yh), B, T, H, st))
    assert (yh.cpu() - y.detach()).abs().max() < 5e-3
    dX = dy.permute(1, 0, 2).reshape(M, 2 * H).contiguous().to(dev)
    dh = torch.empty(lib.mt_lstm_cx_bytes(B, T, H) // 4, device=dev)
    dgx = torch.empty(lib.mt_lstm_dgx_bytes(B, T, H), dtype=torch.uint8, device=dev)
    check(lib.mt_lstm_dh_relayout(ptr(dX), 2 * H, ptr(dh), B, T, H, H, 0.0, 0, 0, st))
    part = torch.empty(lib.mt_lstm_bwd_part_bytes(B, T, H), dtype=torch.uint8, device=dev)
    check(lib.mt_lstm_bidir_bwd(ptr(gx), ptr(cx), ptr(dh), ptr(whd), ptr(dgx), ptr(part), part.numel(), ptr(sync), sync.numel(), B, T, H, st))
    dG = torch.zeros(Mpad, 8 * H, dtype=torch.bfloat16, device=dev)
    dGT = torch.zeros(8 * H, Mpad, dtype=torch.bfloat16, device=dev)
    check(lib.mt_lstm_dg_unpack(ptr(dgx), ptr(dG), 8 * H, ptr(dGT), Mpad, B, T, H, st))
    torch.cuda.synchronize()
    assert int(sync[:4].view(torch.int32).item()) == 0, "backward hand-off timed out"
    assert torch.equal(dG[:M].t().contiguous(), dGT[:, :M].contiguous())
    # d(gate pre-activations): reference via dL/dx = dG W_ih  and dL/dW_ih = dG^T x  (both linear in dG)
    dGf = dG[:M].float().cpu()
    dx_h = (dGf @ w_ih.reshape(8 * H, K)).reshape(T, B, K).permute(1, 0, 2)
    assert (dx_h - xo.grad).abs().max() < GRAD_REL * xo.grad.abs().max()
    dwi_h = (dGf.t() @ x.permute(1, 0, 2).reshape(M, K)).reshape(2, 4 * H, K)
    assert (dwi_h - wio.grad).abs().max() < GRAD_REL * wio.grad.abs().max()
    # dW_hh through mt_lstm_hprev_t + GEMM
    Hr = H
    HT = torch.zeros(2 * Hr, Mpad, dtype=torch.bfloat16, device=dev)
    check(lib.mt_lstm_hprev_t(ptr(hx), ptr(HT), Mpad, Hr, B, T, H, st))
    gwh = torch.empty(2, 4 * H, H, device=dev)
    for d in range(2):
        check(lib.mt_gemm_bf16_f32acc(ptr(dGT[d * 4 * H:]), Mpad, ptr(HT[d * Hr:]), Mpad, None, ptr(gwh[d]), H, 4 * H, H, Mpad, st))
    assert (gwh.cpu() - who.grad).abs().max() < GRAD_REL * who.grad.abs().max()


def test_dropout_mask_statistics_and_backward_consistency(mta):
    from music_transcription_amd import _lib
    from music_transcription_amd._lib import lib, check, ptr
    B, T, H = 3, 17, 32
    dev = "cuda"
    hx = torch.zeros(lib.mt_lstm_hx_bytes(B, T, H) // 4, device=dev)
    hx.view(torch.int16).fill_(0x3C00)                 # every h = 1.0 (f16)
    M, K1 = T * B, 64
    st = _lib.stream_ptr()
    X = torch.zeros(M, K1, dtype=torch.bfloat16, device=dev)
    p = 0.3
    check(lib.mt_lstm_relayout_train(ptr(hx), ptr(X), K1, B, T, H, H, p, 1234, 0, st))
    Xf = X.float().cpu()
    kept = (Xf != 0)
    assert torch.allclose(Xf[kept], torch.tensor(1.0 / (1.0 - p)), rtol=1e-2)
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - p) < 0.05
    # same mask in the backward re-layout
    dX = torch.ones(M, K1, device=dev)
    dh = torch.empty(lib.mt_lstm_cx_bytes(B, T, H) // 4, device=dev)
    check(lib.mt_lstm_dh_relayout(ptr(dX), K1, ptr(dh), B, T, H, H, p, 1234, 0, st))
    d5 = dh.view(1, T, 2, H // 8, 8, 32).cpu()         # [g][t][d][kb][jl][b]
    back = torch.zeros(M, 2 * H)
    for t in range(T):
        for b in range(B):
            back[t * B + b] = d5[0, t, :, :, :, b].reshape(2 * H)
    assert torch.equal(back != 0, kept)
    # another seed gives another mask; dropout 0 is the identity
    X2 = torch.zeros(M, K1, dtype=torch.bfloat16, device=dev)
    check(lib.mt_lstm_relayout_train(ptr(hx), ptr(X2), K1, B, T, H, H, p, 99, 0, st))
    assert not torch.equal(X2, X)
    check(lib.mt_lstm_relayout_train(ptr(hx), ptr(X2), K1, B, T, H, H, 0.0, 99, 0, st))
    assert torch.all(X2.float() == 1.0)


def test_train_forward_with_dropout_runs_and_differs(mta):
    m, _ = _hip_model(mta, 32, 16, 2, seed=5, dropout=0.5)
    m.train()
    x = _mel_in(2, 32, 30, 1).cuda()
    torch.manual_seed(0)
    a = m(x).detach().clone()
    torch.manual_seed(0)
    b = m(x).detach().clone()
    torch.manual_seed(1)
    c = m(x).detach().clone()
    assert (a - b).abs().max() < 1e-5 and (a - c).abs().max() > 1e-4
    m.eval()
    with torch.no_grad():
        e = m(x)
    assert torch.isfinite(e).all()


def test_preprocess_and_cache_end_to_end(mta, tmp_path):
    """scripts/preprocess_dataset.py path: synthetic MAESTRO-like tree (44.1 kHz stereo WAV + MIDI + csv) ->
    cache -> the reader; mel equals the frontend on the resampled slice, labels come from the own MIDI reader."""
    from scipy.io import wavfile
    from music_transcription_amd import preprocess as P, transcribe as TR, midi as MD
    from oracle import frontend_ref as FR
    root = tmp_path / "maestro"
    (root / "2004").mkdir(parents=True)
    rng = np.random.default_rng(0)
    rows = ["canonical_composer,canonical_title,split,year,midi_filename,audio_filename,duration"]
    durs = {"a": 47.0, "b": 31.5, "c": 12.0}
    for name, split in (("a", "train"), ("b", "train"), ("c", "validation")):
        n = int(durs[name] * 44100)
        t = np.arange(n) / 44100.0
        sig = 0.3 * np.sin(2 * np.pi * 440.0 * t) * np.exp(-0.5 * (t % 2.0)) + 0.01 * rng.standard_normal(n)
        wavfile.write(str(root / "2004" / f"{name}.wav"), 44100, (np.stack([sig, 0.5 * sig], 1) * 32767).astype(np.int16))
        notes = [(69, s, s + 1.5) for s in np.arange(0.0, durs[name] - 2.0, 2.0)]
        TR.write_midi(notes, str(root / "2004" / f"{name}.midi"))
        rows.append(f"X,Y,{split},2004,2004/{name}.midi,2004/{name}.wav,{durs[name]}")
    (root / "maestro-v3.0.0.csv").write_text("\n".join(rows) + "\n")
    cache = str(tmp_path / "cache")
    st = P.preprocess_and_cache(str(root), cache, 30.0, 0.0, 64, 16000, 512, "train")
    # a: 0-30 and 30-47 (17 s >= 15 s); b: 0-30 only (1.5 s tail dropped)
    assert st == {"cached": 3, "skipped": 0, "failed": 0}
    assert P.preprocess_and_cache(str(root), cache, 30.0, 0.0, 64, 16000, 512, "train")["skipped"] == 3
    assert P.preprocess_and_cache(str(root), cache, 30.0, 0.0, 64, 16000, 512, "validation")["cached"] == 0   # 12 s < 15 s
    ds = mta.CachedMaestroDataset(cache, "train")
    assert len(ds) == 3 and ds.metadata["n_mels"] == 64 and ds.metadata["chunks"][1]["start_sample"] == 480000
    mel0, roll0 = ds[0]
    mel1, roll1 = ds[1]
    assert mel0.shape == (1, 64, 937) and roll0.shape == (88, 937)            # min(938, int(30*31.25)) frames
    assert mel1.shape == (1, 64, 531) and roll1.shape == (88, 531)            # 17 s: min(1 + 272000//512, int(17*31.25))
    y = TR.load_audio(str(root / "2004" / "a.wav"), 16000)
    ref = FR.audio_to_mel_batch(y[None, :480000], 16000, 64, 512)[0, 0, :, :937]
    assert np.abs(mel0[0].numpy() - ref).max() < 5e-2
    # A4 (MIDI 69 -> row 48) sounds for 1.5 s out of every 2 s; nothing else does
    on = roll0[48].numpy()
    assert 0.70 < on[:-1].mean() < 0.80 and roll0.sum() == on.sum() and on[-1] == 0
    m = MD.MidiFile(str(root / "2004" / "a.midi"))
    assert np.array_equal(roll1.numpy(), MD.chunk_roll(m, 30.0, 47.0)[:, :531])


@pytest.mark.parametrize("mtype", ["cnn_rnn", "cnn_rnn_large"])
def test_data_parallel_training_two_ranks(mta, mtype):
    """Two ranks (torch.distributed.run, gloo between processes that share this box's GPU), different data per rank:
    after three steps of all-reduce(mean) + fused clip/Adam every rank holds bit-identical parameters -- the same ones whether the
    gradients of the upper LSTM layers and the fc are reduced early, under the rest of the backward pass, or all at once."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for early in ("1", "0"):          # with the tail of the flat gradient all-reduced under the backward pass (optim.EarlyBucket), and without
        env = dict(os.environ, MT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MT_DP_EARLY_BUCKET=early, MT_DP_MODEL=mtype)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29533", os.path.join(root, "tests", "tools", "dp_check.py")], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        assert out["world"] == 2 and out["identical_parameters"] and out["model"] == mtype, out
        assert all(np.isfinite(out["losses_rank0"])) and len(out["losses_rank0"]) == 3
        if mtype == "cnn_rnn_large":       # CNNRNNModelLarge: everything above the convolutions is reduced under the convolution backward,
            assert out["onset_offset_heads_untouched"] is True, out       # and the heads without a gradient path stay as they were
        outs[early] = out
    assert outs["1"]["early_bucket_reduces"] == 3 and outs["0"]["early_bucket_reduces"] == 0
    # the same sums in a different order of collectives: identical parameters
    assert outs["1"]["checksum"] == outs["0"]["checksum"] and outs["1"]["losses_rank0"] == outs["0"]["losses_rank0"]


def test_full_size_training_step_configs3(mta):
    """BASELINE configs[3] shape on one GPU: CNNRNNModel 320/512/3, batch 16 cached-format chunks with ragged
    T in [469, 937] right-padded with 0.0 (collate_fn), Bernoulli(0.04) rolls.  The train-mode forward equals the CPU
    oracle's (same bf16 rounding points; BatchNorm batch statistics), every gradient is finite, the step is run-to-run
    deterministic, and one fused clip + Adam step changes the eval-mode output."""
    nm, H, L, B, T = 320, 512, 3, 16, 937
    m, sd = _hip_model(mta, nm, H, L, seed=4, dropout=0.0)
    g = torch.Generator().manual_seed(12)
    lengths = torch.randint(469, T + 1, (B,), generator=g)
    lengths[0] = T
    mel = torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0
    roll = (torch.rand(B, 88, T, generator=g) < 0.04).float()
    for b in range(B):
        mel[b, :, :, lengths[b]:] = 0.0
        roll[b, :, lengths[b]:] = 0.0
    m.train()
    runs = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        logits = m(mel.cuda())
        loss = m.compute_loss(logits, roll.cuda(), lengths)
        loss.backward()
        m.model.raise_on_train_handoff_timeout()
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        assert torch.isfinite(flat).all() and float(flat.abs().max()) > 0
        runs.append((logits.detach().clone(), float(loss.item()), flat.clone()))
        for mod in m.modules():                               # the second run must see the same BatchNorm buffers
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    # weight-gradient partial sums are reduced in a fixed order; the BatchNorm sums use f64 atomics (order-dependent in the
    # last f64 bits, invisible after the f32 rounding in practice): allow 1e-6 relative
    assert (runs[0][2] - runs[1][2]).abs().max() <= 1e-6 * float(runs[0][2].abs().max())
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        lo = R.cnnrnn_forward({k: v.clone() for k, v in sd.items()}, mel, R.Opts(gemm_bf16=True, fast_lstm=False), train=True)
    assert (runs[0][0].cpu() - lo).abs().max() < 1e-2
    assert abs(runs[0][1] - float(R.compute_loss(lo, roll, lengths))) < 1e-3
    opt = mta.make_optimizer(m, lr=1e-3)
    m.eval()
    with torch.no_grad():
        before = m(mel[:2, :, :, :469].contiguous().cuda()).clone()
    m.train()
    opt.zero_grad()
    m.compute_loss(m(mel.cuda()), roll.cuda(), lengths).backward()
    stats = opt.step(sync_grads=False).tolist()
    assert stats[1] == 1.0 and np.isfinite(stats[0]) and stats[0] > 0
    m.eval()
    with torch.no_grad():
        after = m(mel[:2, :, :, :469].contiguous().cuda())
    assert (after - before).abs().max() > 1e-4


@pytest.mark.parametrize("R,C,lds,ldd,Cd,off", [(300, 64, 64, 320, 64, 0), (1000, 72, 72, 1024, 128, 0), (257, 50, 56, 264, 50, 0),
                                              (130, 64, 64, 136, 64, 3), (64, 24, 30, 70, 24, 0)])
def test_transpose_bf16_helper(mta, R, C, lds, ldd, Cd, off):
    """mt_transpose_bf16 (train.hip): dst[c*ldd + r] = src[r*lds + c], zeros in the padding; the 16-byte path, its ragged
    chunks, and the scalar kernel it falls back to for unaligned operands (off / lds / ldd not multiples of 8)."""
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    torch.manual_seed(R + C)
    buf = torch.randn(R * lds + 8, device="cuda").bfloat16()
    src = buf[off:off + R * lds]
    dst = torch.full((Cd * ldd,), 7.0, device="cuda").bfloat16()
    check(lib.mt_transpose_bf16(ptr(src), lds, R, C, ptr(dst), ldd, Cd, stream_ptr()), "mt_transpose_bf16")
    torch.cuda.synchronize()
    want = torch.zeros(Cd, ldd, device="cuda").bfloat16()
    want[:C, :R] = src.view(R, lds)[:, :C].t()
    assert torch.equal(dst.view(Cd, ldd), want)


@pytest.mark.parametrize("n,s", [((4, 24, 64, 80), (24 * 5120, 5120, 1, 64)), ((2, 5, 7, 3), (200, 30, 1, 7)),
                                 ((3, 4, 1, 16), (100, 20, 0, 1)), ((1, 1, 8, 9), (0, 0, 9, 1))])
def test_gather4_helper(mta, n, s):
    """mt_gather4_f32: dst[i0][i1][i2][i3] = alpha * src[sum i_k s_k]; the LDS-transposing kernel (s2 = 1, s3 = n2) and the
    generic one."""
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    need = sum((nk - 1) * sk for nk, sk in zip(n, s)) + 1
    src = torch.randn(need, device="cuda")
    dst = torch.empty(n, device="cuda")
    check(lib.mt_gather4_f32(ptr(src), ptr(dst), n[0], n[1], n[2], n[3], s[0], s[1], s[2], s[3], 0.5, stream_ptr()), "mt_gather4_f32")
    torch.cuda.synchronize()
    want = 0.5 * torch.as_strided(src, n, s)
    assert torch.equal(dst, want)


@pytest.mark.parametrize("B,T,H,Hv,K,p", [(16, 60, 512, 512, 256, 0.0), (16, 512, 512, 512, 256, 0.25), (6, 2800, 256, 256, 128, 0.25),
                                        (5, 37, 48, 40, 64, 0.5), (40, 33, 64, 64, 128, 0.0)])
def test_gemm_lstm_dh_matches_gemm_plus_relayout(mta, B, T, H, Hv, K, p):
    """mt_gemm_lstm_dh (the dh layout + dropout mask as a GEMM epilogue) against the two-pass path it replaces,
    mt_gemm_bf16_f32acc + mt_lstm_dh_relayout: bit-identical, on the 128-tile kernel, the 256-tile kernel's hoisted path
    (4 | B) and its generic one, with padded units (Hv < H) and a ragged last batch group."""
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    torch.manual_seed(B * T + H)
    M, N = T * B, 2 * Hv
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    A = torch.randn(Mp, K, device="cuda").bfloat16()
    W = (torch.randn(Np, K, device="cuda") * 0.1).bfloat16()
    n_dh = lib.mt_lstm_cx_bytes(B, T, H) // 4
    dX = torch.empty(M, N, device="cuda")
    check(lib.mt_gemm_bf16_f32acc(ptr(A), K, ptr(W), K, None, ptr(dX), N, M, N, K, stream_ptr()), "gemm")
    want = torch.empty(n_dh, device="cuda")
    check(lib.mt_lstm_dh_relayout(ptr(dX), N, ptr(want), B, T, H, Hv, p, 77, 1, stream_ptr()), "relayout")
    got = torch.zeros(n_dh, device="cuda")
    check(lib.mt_gemm_lstm_dh(ptr(A), K, ptr(W), K, ptr(got), B, T, H, Hv, K, p, 77, 1, stream_ptr()), "mt_gemm_lstm_dh")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    if p > 0:
        frac = (got == 0).float().mean().item()
        assert frac > 0.5 * p                      # the mask is really applied


@pytest.mark.parametrize("B,F,T", [(1, 16, 16), (2, 20, 37), (3, 33, 50), (1, 5, 7)])
def test_conv2_wgrad_direct_kernel(mta, B, F, T):
    """mt_conv2_wgrad (position-contracted MFMA, transposed LDS staging, 3 pre-shifted activation copies) against the fp64
    weight / bias gradient of Conv2d(32, 64, 3, padding=1) computed from the SAME bf16 operands: exact products, so only
    the f32 accumulation order differs.  Ragged tiles in both axes, batch > 1, a tile smaller than 16 x 16."""
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    import torch.nn.functional as Fn
    torch.manual_seed(B * 100 + F + T)
    a1 = torch.randn(B, F, T, 32, device="cuda").bfloat16()
    dz = torch.randn(B, F, T, 64, device="cuda") * 0.1
    hi = dz.bfloat16()
    lo = (dz - hi.float()).bfloat16()
    nwg = lib.mt_conv2_wgrad_workgroups()
    P, Pb = torch.empty(nwg, 64, 288, device="cuda"), torch.empty(nwg, 64, device="cuda")
    check(lib.mt_conv2_wgrad(ptr(a1), ptr(hi), ptr(lo), ptr(P), ptr(Pb), nwg, B, F, T, stream_ptr()), "mt_conv2_wgrad")
    torch.cuda.synchronize()
    nt = B * ((F + 15) // 16) * ((T + 15) // 16)
    got = P[:min(nwg, nt)].double().sum(0).reshape(64, 9, 32).permute(0, 2, 1).reshape(64, 32, 3, 3).cpu()
    gotb = Pb[:min(nwg, nt)].double().sum(0).cpu()
    assert torch.all(P[min(nwg, nt):] == 0) and torch.all(Pb[min(nwg, nt):] == 0)      # idle workgroups still write their (zero) partial
    x = a1.double().cpu().permute(0, 3, 1, 2)                         # NCHW
    g = (hi.double() + lo.double()).cpu().permute(0, 3, 1, 2)
    w = torch.zeros(64, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    Fn.conv2d(x, w, padding=1).backward(g)
    scale = w.grad.abs().max().item()
    assert (got - w.grad).abs().max().item() <= 2e-5 * scale + 1e-6
    wantb = hi.double().cpu().sum((0, 1, 2))
    assert (gotb - wantb).abs().max().item() <= 1e-4 * wantb.abs().max().item() + 1e-5


@pytest.mark.parametrize("S,rows,cols,ldp,ldo", [(1, 3, 5, 5, 5), (7, 64, 288, 288, 288), (256, 1, 64, 64, 64), (19, 10, 33, 40, 36)])
def test_sum_slices_helper(mta, S, rows, cols, ldp, ldo):
    """mt_sum_slices_f32: out[r][c] = sum_z P[z][r][c] (split-K / per-workgroup partials), fixed summation tree."""
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    torch.manual_seed(S + rows)
    P = torch.randn(S, rows, ldp, device="cuda")
    out = torch.full((rows, ldo), 9.0, device="cuda")
    check(lib.mt_sum_slices_f32(ptr(P), rows * ldp, ldp, S, ptr(out), ldo, rows, cols, stream_ptr()), "mt_sum_slices_f32")
    out2 = torch.full((rows, ldo), 9.0, device="cuda")
    check(lib.mt_sum_slices_f32(ptr(P), rows * ldp, ldp, S, ptr(out2), ldo, rows, cols, stream_ptr()), "mt_sum_slices_f32")
    torch.cuda.synchronize()
    want = P[:, :, :cols].double().sum(0)
    assert (out[:, :cols].double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    assert torch.all(out[:, cols:] == 9.0) and torch.equal(out, out2)


def test_pack_wih_cf_matches_index_gather(mta):
    """mt_pack_wih_cf (layer 0's W_ih -> the projection GEMM's 16-bit operand rows in the kernels' feature order) against the torch index
    gather + cast it replaces (model._pack_bilstm): bit-exact for bf16 and f16, zero rows for padded hidden units, both directions, row offset."""
    from music_transcription_amd import _lib
    from music_transcription_amd.model import _pack_bilstm, _h16
    for H, C, F, dt in ((24, 64, 10, _lib.DT_BF16), (16, 256, 5, _lib.DT_F16), (512, 256, 40, _lib.DT_BF16)):
        Hp = (H + 15) // 16 * 16
        K = C * F
        rnn = torch.nn.LSTM(K, H, 1, bidirectional=True).cuda()
        with torch.no_grad():
            rnn.weight_ih_l0[3, 5] = 1e6                       # (f16 saturates at 65504 instead of overflowing)
        cols = (torch.arange(C)[None, :] * F + torch.arange(F)[:, None]).reshape(-1)
        slow, _, _ = _pack_bilstm(rnn, 1, H, cols, "cuda", dt)
        fast, _, _ = _pack_bilstm(rnn, 1, H, cols, "cuda", dt, k0_cf=(C, F))
        assert fast[0].dtype == slow[0].dtype and fast[0].shape == slow[0].shape and torch.equal(fast[0].view(torch.int16), slow[0].view(torch.int16)), (H, C, F)
        out = torch.full((8 * Hp + 256, K), 7.0, device="cuda", dtype=fast[0].dtype)
        _pack_bilstm(rnn, 1, H, cols, "cuda", dt, k0_cf=(C, F), wih0_out=out[64:])
        assert torch.equal(out[64:64 + 8 * Hp].view(torch.int16), slow[0][:8 * Hp].view(torch.int16)) and bool((out[:64] == 7.0).all()) and bool((out[64 + 8 * Hp:] == 7.0).all())


def test_mt_allreduce_with_a_single_rank_rccl_communicator(mta):
    """mt_allreduce (the C ABI's gradient all-reduce for a host that is not Python) drives RCCL's ncclAllReduce with the CALLER's
    communicator: here a one-rank communicator made through RCCL's own C API (ncclGetUniqueId / ncclCommInitRank), so the sum over ranks is the
    identity -- in place, on the caller's stream, f32 and bf16; bad arguments are refused without touching RCCL."""
    import ctypes as C
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr, last_error
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl.so.1 is not on the loader path of this box")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        x = torch.randn(1 << 20, device="cuda", generator=g)
        want = x.clone()
        check(lib.mt_allreduce(ptr(x), x.numel(), 0, comm, stream_ptr()), "mt_allreduce")
        xb = want.bfloat16()
        wantb = xb.clone()
        check(lib.mt_allreduce(ptr(xb), xb.numel(), 1, comm, stream_ptr()), "mt_allreduce")
        torch.cuda.synchronize()
        assert torch.equal(x, want) and torch.equal(xb, wantb)
        assert lib.mt_allreduce(ptr(x), x.numel(), 7, comm, stream_ptr()) < 0 and "dtype" in last_error()
        assert lib.mt_allreduce(ptr(x), 0, 0, comm, stream_ptr()) < 0 and lib.mt_allreduce(ptr(x), 4, 0, None, stream_ptr()) < 0
    finally:
        rccl.ncclCommDestroy(comm)
    assert lib.mt_init(0) == 0 and lib.mt_init(99) < 0
    assert lib.mt_workspace_bytes(5, 16, 937, 512) == lib.mt_lstm_bwd_part_bytes(16, 937, 512) == 937 * 2 * 16 * 16 * 1024


@pytest.mark.gpu
def test_pack_job_tables_equal_the_torch_expressions_small_model(mta):
    """CNNRNNModel's training operands as mt_pack_jobs writes them (train_step.pack_train) against the torch expressions they replace
    (tests/tools/pack_reference_small.py): bit for bit, before and after the parameters move."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from pack_reference_small import pack_train_torch
    from music_transcription_amd.train_step import pack_train
    m = mta.TranscriptionModel(model_type="cnn_rnn", n_mels=64, hidden_size=24, num_layers=3, dropout=0.0, device="cuda")
    net = m.model
    for rnd in range(2):
        got, want = pack_train(net, "cuda"), pack_train_torch(net, torch.device("cuda"))
        torch.cuda.synchronize()
        for k, w in want.items():
            for i, (g_, w_) in enumerate(zip(got[k], w) if isinstance(w, (list, tuple)) else [(got[k], w)]):
                if not isinstance(w_, torch.Tensor):
                    continue
                assert g_.shape == w_.shape and g_.dtype == w_.dtype, (k, i, g_.shape, w_.shape)
                assert torch.equal(g_, w_), (rnd, k, i, float((g_.float() - w_.float()).abs().max()))
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn_like(p) * 0.01)
