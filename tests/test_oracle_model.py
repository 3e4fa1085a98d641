"""The CPU oracle (oracle/model_ref.py) against outputs of the reference itself
(tests/golden/*.npz, written by tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as R

TOL = 2e-5  # fp32 noise floor between two fp32 evaluation orders (~1e-6 observed)


def _sd(mt, cfg, bn=None, **kw):
    """Seeded weights + the fixture's calibrated BatchNorm running statistics."""
    nm, hs, nl, B, T, wseed, xseed = [int(v) for v in cfg]
    sd = R.make_state_dict(mt, nm, hs, nl, wseed, **kw)
    if bn is not None:
        R.set_bn_flat(sd, bn)
    return sd, (nm, hs, nl, B, T, xseed)


def _mel(B, nm, T, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=g))


def _wsum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values() if v.dtype.is_floating_point))


@pytest.fixture(scope="module")
def small(golden_dir):
    return np.load(os.path.join(golden_dir, "small_models.npz"))


@pytest.mark.parametrize("tag,mt", [("small_a", "cnn_rnn"), ("small_b", "cnn_rnn"),
                                    ("large_a", "cnn_rnn_large"), ("large_b", "cnn_rnn_large")])
def test_small_configs_full_logits(small, tag, mt):
    sd, (nm, hs, nl, B, T, xs) = _sd(mt, small[f"{tag}_cfg"], small[f"{tag}_bn"])
    assert abs(_wsum(sd) - float(small[f"{tag}_wsum"])) < 1e-6 * float(small[f"{tag}_wsum"]), "weight RNG drift"
    x = _mel(B, nm, T, xs)
    y = R.forward(sd, x, mt).numpy()
    assert y.shape == small[f"{tag}_logits"].shape == (B, 88, T)
    assert np.abs(y - small[f"{tag}_logits"]).max() < TOL
    if mt == "cnn_rnn_large":
        d = R.cnnrnn_large_forward(sd, x, return_all_heads=True)
        for k in ("frame", "onset", "offset"):
            assert np.abs(d[k].numpy() - small[f"{tag}_{k}"]).max() < TOL
    for th in (0.3, 0.5, 0.7):
        want = np.unpackbits(small[f"{tag}_pred{int(th * 10)}"])[: B * 88 * T].reshape(B, 88, T)
        got = R.predict(torch.from_numpy(small[f"{tag}_logits"]), th).numpy()
        assert (got == want).all()
    assert int(small[f"{tag}_zero_raises"]) == 1  # the reference cannot run T == 0 at all


def test_fast_lstm_matches_loop(small):
    sd, (nm, hs, nl, B, T, xs) = _sd("cnn_rnn", small["small_b_cfg"], small["small_b_bn"])
    x = _mel(B, nm, T, xs)
    a = R.cnnrnn_forward(sd, x).numpy()
    b = R.cnnrnn_forward(sd, x, R.Opts(fast_lstm=True)).numpy()
    assert np.abs(a - b).max() < TOL


def test_large_variants(small):
    x = _mel(2, 32, 30, 6)
    sd = R.set_bn_flat(R.make_state_dict("large", 32, 16, 2, 12, use_attention=False, use_heads=True), small["large_noattn_bn"])
    assert np.abs(R.forward(sd, x, "large").numpy() - small["large_noattn_logits"]).max() < TOL
    sd = R.set_bn_flat(R.make_state_dict("large", 32, 16, 2, 13, use_attention=True, use_heads=False), small["large_noheads_bn"])
    assert np.abs(R.forward(sd, x, "large").numpy() - small["large_noheads_logits"]).max() < TOL


def test_padding_leaks_into_valid_frames(small):
    sd = R.set_bn_flat(R.make_state_dict("cnn_rnn", 32, 16, 2, 11), small["pad_bn"])
    xa, xb = _mel(1, 32, 50, 7), _mel(1, 32, 30, 8)
    xp = torch.cat([xa, torch.nn.functional.pad(xb, (0, 20))], 0)
    yb = R.cnnrnn_forward(sd, xp).numpy()
    assert np.abs(yb - small["pad_logits_batch"]).max() < TOL
    solo = R.cnnrnn_forward(sd, xb).numpy()
    assert np.abs(solo - small["pad_logits_solo_b"]).max() < TOL
    # Appendix A: zero (0 dB) padding is not masked, so valid frames differ from the solo run
    assert np.abs(yb[1, :, :30] - solo[0]).max() > 1e-3


@pytest.mark.parametrize("tag,mt", [("small_937", "cnn_rnn"), ("small_938", "cnn_rnn"),
                                    ("large_937", "cnn_rnn_large")])
def test_canonical_config_samples(golden_dir, tag, mt):
    c = np.load(os.path.join(golden_dir, "canonical_models.npz"))
    sd, (nm, hs, nl, B, T, xs) = _sd(mt, c[f"{tag}_cfg"], c[f"{tag}_bn"])
    assert abs(_wsum(sd) - float(c[f"{tag}_wsum"])) < 1e-6 * float(c[f"{tag}_wsum"])
    x = _mel(B, nm, T, xs)
    with torch.no_grad():
        if mt == "cnn_rnn_large":
            d = R.cnnrnn_large_forward(sd, x, return_all_heads=True, o=R.Opts(fast_lstm=True))
            y = d["frame"].numpy()
            assert np.abs(d["onset"].numpy()[:, ::5, ::7] - c[f"{tag}_onset_sample"]).max() < 5e-5
            assert np.abs(d["offset"].numpy()[:, ::5, ::7] - c[f"{tag}_offset_sample"]).max() < 5e-5
        else:
            y = R.cnnrnn_forward(sd, x, R.Opts(fast_lstm=True)).numpy()
    assert np.abs(y[:, ::5, ::7] - c[f"{tag}_sample"]).max() < 5e-5
    st = c[f"{tag}_stats"]
    assert abs(y.mean() - st[0]) < 1e-5 and abs(np.abs(y).max() - st[2]) < 5e-5


def test_losses(golden_dir):
    g = np.load(os.path.join(golden_dir, "loss.npz"))
    logits = torch.from_numpy(g["logits"])
    B, P, T = logits.shape
    targets = torch.from_numpy(np.unpackbits(g["targets"])[: B * P * T].reshape(B, P, T).astype(np.float32))
    lengths = torch.from_numpy(g["lengths"])
    assert abs(float(R.compute_loss(logits, targets)) - float(g["loss_nolen"])) < 1e-6
    assert abs(float(R.compute_loss(logits, targets, lengths)) - float(g["loss_len"])) < 1e-6
    d = {"frame": logits, "onset": logits * 0.5 - 1.0, "offset": -logits + 0.25}
    assert abs(float(R.compute_loss(d, targets)) - float(g["loss_dict_nolen"])) < 1e-6
    assert abs(float(R.compute_loss(d, targets, lengths)) - float(g["loss_dict_len"])) < 1e-6
    assert float(R.compute_loss(logits, targets, torch.tensor([0, 0, 0]))) == float(g["loss_len_zero"]) == 0.0
    lg = logits.clone().requires_grad_(True)
    R.compute_loss(lg, targets, lengths).backward()
    assert np.abs(lg.grad.numpy() - g["grad_len"]).max() < 1e-8


def test_collate(golden_dir):
    g = np.load(os.path.join(golden_dir, "collate.npz"))
    gen = torch.Generator().manual_seed(int(g["seed"]))
    batch = [(torch.randn(1, 8, int(t), generator=gen), (torch.rand(88, int(t), generator=gen) < 0.1).float())
             for t in g["Ts"]]
    mel, roll, lens = R.collate(batch)
    assert (mel.numpy() == g["mel"]).all() and (roll.numpy() == g["roll"]).all()
    assert (lens.numpy() == g["lengths"]).all()
    assert mel[1, 0, :, 9:].abs().max() == 0.0  # pad value is 0.0 in the dB domain


def test_f1_matches_sklearn_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "f1.npz"))
    for yt, yp, want in zip(g["y_true"], g["y_pred"], g["f1"]):
        assert abs(R.f1_binary(yt, yp) - want) < 1e-12
    assert R.f1_binary(np.zeros(10), np.zeros(10)) == 0.0  # zero_division=0, not 1


def test_manifest_key_names(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    for key, ent in man.items():
        parts = key.split(":")
        mt, nm, hs, nl = parts[0], int(parts[1]), int(parts[2]), int(parts[3])
        sd = R.make_state_dict(mt, nm, hs, nl, 0, use_attention="noattn" not in parts, use_heads="noheads" not in parts)
        assert {k: list(v.shape) for k, v in sd.items()} == ent["keys"], key
    assert man["cnn_rnn:320:512:3"]["n_params"] == 35785368
    assert man["cnn_rnn_large:320:512:3"]["n_params"] == 89494088


@pytest.mark.parametrize("model_type,fname", [("cnn_rnn", "train_step.npz"), ("cnn_rnn_large", "train_step_large.npz")])
def test_oracle_training_step_pinned_by_reference_golden(golden_dir, model_type, fname):
    """oracle.model_ref.train_steps (train-mode forward, autograd, clip, Adam) against the REFERENCE's own gradients,
    losses and post-step weights (tests/golden/train_step*.npz, written by make_golden_train.py)."""
    g = np.load(os.path.join(golden_dir, fname))
    nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]

    def mel_in(seed):
        gen = torch.Generator().manual_seed(seed)
        return (torch.rand(B, 1, nm, T, generator=gen) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=gen))

    def roll_in(seed):
        gen = torch.Generator().manual_seed(seed)
        return (torch.rand(B, 88, T, generator=gen) < 0.1).float()
    data = []
    for k in range(nb):
        mel, roll = mel_in(sx + k), roll_in(sx + 100 + k)
        lengths = torch.tensor([T, T - 7, T - 15][:B], dtype=torch.int64)
        for b in range(B):
            mel[b, :, :, lengths[b]:] = 0.0
            roll[b, :, lengths[b]:] = 0.0
        data.append((mel, roll, lengths))
    sd = R.make_state_dict(model_type, nm, H, L, sw)
    losses, grads, gn, logits0 = R.train_steps(sd, data, lr=float(g["lr"]), model_type=model_type)
    assert np.abs(np.array(losses) - g["losses"]).max() < 2e-6
    assert abs(gn - float(g["gradnorm0"])) < 5e-6 * max(1.0, float(g["gradnorm0"]))      # (the golden's norm is a float64 sum)
    assert np.abs(logits0.numpy() - g["logits0"]).max() < 1e-5
    for k, v in grads.items():
        assert np.abs(v.numpy() - g["grad::" + k]).max() < 2e-7 * max(1.0, float(np.abs(g["grad::" + k]).max())), k
    for k, v in sd.items():
        if not v.dtype.is_floating_point:
            assert int(v) == int(g["post::" + k]), k
            continue
        # conv biases in front of a BatchNorm have an analytically zero gradient: Adam amplifies their rounding noise
        zero_grad_bias = k.endswith(".bias") and any(t in k for t in ("cnn.0.", "cnn.4.", "conv1.0.", ".conv1.bias", ".conv2.bias", "skip.0.", "freq_aware_conv.0."))
        # (three Adam steps of lr 1e-4 move a weight by up to 3e-4; entries whose gradient is ~0 amplify rounding noise)
        tol = 3.5e-4 if zero_grad_bias else (5e-6 if model_type == "cnn_rnn" else 2e-5)
        assert np.abs(v.numpy() - g["post::" + k]).max() < tol, k
