"""F1 parity at canonical size on trained-scale weights (BASELINE north star: framewise F1 within +-0.002).

The HIP path computes its GEMMs on bf16 operands and the recurrence on f16 operands (f32 accumulate / state); the
reference's inference arithmetic is fp32.  These tests license that: >= 4 full 30 s chunks (T = 938) go through
waveform -> mel -> model on the HIP path and through the fp32 CPU oracle (oracle frontend + oracle model, no
rounding emulation), with weights rescaled into a trained model's regime (oracle.model_ref.trained_scale_state_dict:
saturating LSTM gates, clamped attention scores, frame logits ~ N(-5, 3) -> ~5 % positive cells, |logit| > 6), and
the reference's metric (scripts/evaluate.py:361-378: per-chunk binary F1 over the flattened roll, unweighted mean)
is compared at threshold 0.5 and at the threshold tuned on the fp32 side.

Labels: no MAESTRO data exists here, so the ground-truth roll is synthetic and correlated with the model output:
y = 1 where fp32_logit + N(0,1) > 0 (the fp32 model then scores a realistic F1 of ~0.7-0.8 against it).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import frontend_ref as FR
from oracle import model_ref as R

F1_TOL = 0.002            # the north star's tolerance on the mean framewise F1
N_CHUNKS = 4


@pytest.fixture(scope="module")
def mta():
    import music_transcription_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def _f1_at(logits, rolls, thr):
    preds = (torch.sigmoid(logits) > thr).float().numpy()
    return R.mean_f1(preds, rolls, [rolls.shape[-1]] * rolls.shape[0])


def _tune(logits, rolls):
    """Threshold maximising the fp32 model's mean F1 on a 0.05 grid (the data side of evaluate.py:556-618)."""
    grid = [round(0.05 * k, 2) for k in range(1, 20)]
    vals = [_f1_at(logits, rolls, t) for t in grid]
    return grid[int(np.argmax(vals))]


def _run(mta, model_type, seed, dtype="f16"):
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    wave = FR.synth_audio(N_CHUNKS, 480000, seed=seed)
    mel_ref = torch.from_numpy(FR.audio_to_mel_batch(wave))
    sd0 = R.make_state_dict(model_type, 320, 512, 3, seed=seed + 1)
    sd, cal = R.trained_scale_state_dict(sd0, model_type, mel_ref[:1])
    with torch.no_grad():
        ref = R.forward(sd, mel_ref, model_type, o=R.Opts(fast_lstm=True))            # fp32 end to end
    g = torch.Generator().manual_seed(seed + 2)
    rolls = ((ref + 1.0 * torch.randn(ref.shape, generator=g)) > 0).float().numpy()

    model = mta.TranscriptionModel(model_type, n_mels=320, hidden_size=512, num_layers=3, device="cuda")
    model.load_state_dict(sd, strict=True)
    model.eval()
    model.model.operand_dtype = dtype                 # "f16" is the shipped default; "bf16" is measured for DESIGN.md
    fe = mta.MelFrontend(16000, 320, 512, "cuda")
    with torch.no_grad():
        mel, cmax = fe(torch.from_numpy(wave).cuda(), clamp=True)
        got = model.model(mel, check_status=True).float().cpu()
    assert got.shape == ref.shape == (N_CHUNKS, 88, 938)
    dl = float((got - ref).abs().max())
    span = float(ref.abs().max())
    pos = float((ref > 0).float().mean())
    thr = _tune(ref, rolls)
    out = {"max_dlogit": dl, "max_logit": span, "rel": dl / span, "pos_frac": pos, "thr": thr, "cal_gain": cal["gain"]}
    for name, t in (("0.5", 0.5), ("tuned", thr)):
        out["f1_ref_" + name] = _f1_at(ref, rolls, t)
        out["f1_hip_" + name] = _f1_at(got, rolls, t)
    flips = int(((got > 0) != (ref > 0)).sum())
    out["flips"] = flips
    print(f"\n[{model_type} {dtype}] " + " ".join(f"{k}={v:.5g}" if isinstance(v, float) else f"{k}={v}" for k, v in out.items()))
    # the regime is the one the test claims
    assert span >= 6.0 and 0.02 <= pos <= 0.10, (span, pos)
    assert 0.4 <= out["f1_ref_0.5"] <= 0.97, out["f1_ref_0.5"]
    tol = F1_TOL if dtype == "f16" else 5 * F1_TOL     # bf16 operands: reported, not the shipped inference path
    for name in ("0.5", "tuned"):
        assert abs(out["f1_hip_" + name] - out["f1_ref_" + name]) <= tol, out
    return out


REL_TOL = {"f16": 1e-2, "bf16": 8e-2}     # max |dlogit| / max |logit| against the fp32 oracle


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_f1_parity_cnnrnn_canonical_trained_scale(mta, dtype):
    out = _run(mta, "cnn_rnn", seed=101, dtype=dtype)
    assert out["rel"] < REL_TOL[dtype], out


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_f1_parity_cnnrnn_large_canonical_trained_scale(mta, dtype):
    out = _run(mta, "cnn_rnn_large", seed=202, dtype=dtype)
    assert out["rel"] < REL_TOL[dtype], out


# ------------------------------------------------------------------ hardened evidence (round 3)
# The licence for f16 operands must not rest on one seed, one label-noise level, one recurrence gain and unpadded T = 938:
#   * 2 seeds x 8 chunks, W_hh gain in {3, 4} (gain 4 makes the LSTM's own dynamics amplify input perturbations: DESIGN.md 2);
#   * labels y = [fp32 logit + N(0, sigma) > 0] with sigma in {1.0, 0.3}: with sigma = 0.3 the F1 is decided by the cells near the
#     threshold, exactly the ones a rounding difference can flip;
#   * a cached-format batch (T = 937) of RAGGED chunks right-padded with 0.0 dB as collate_fn does
#     (train/train_transcriber.py:23-39): the padding leaks into valid frames through the bi-LSTM and the attention
#      and the metric is per-sample F1 over the VALID frames only (scripts/evaluate.py:361-378).
# Every row asserts |dF1| <= 0.002 at threshold 0.5 and at the fp32-tuned threshold, and the table is printed.
N_HARD = 8


def _hip_logits(mta, model_type, sd, mel_dev, cmax=None):
    model = mta.TranscriptionModel(model_type, n_mels=320, hidden_size=512, num_layers=3, device="cuda")
    model.load_state_dict(sd, strict=True)
    model.eval()
    with torch.no_grad():
        out = model.model(mel_dev, check_status=True) if cmax is None else model.model(mel_dev, chunk_max_power=cmax, check_status=True)
    return out.float().cpu()


def _f1_valid(logits, rolls, lengths, thr):
    preds = (torch.sigmoid(logits) > thr).float().numpy()
    return R.mean_f1(preds, rolls, list(lengths))


@pytest.mark.parametrize("model_type", ["cnn_rnn", "cnn_rnn_large"])
def test_f1_parity_matrix_seeds_noise_gain(mta, model_type):
    rows, worst = [], 0.0
    fe = mta.MelFrontend(16000, 320, 512, "cuda")
    for seed in (303, 404):
        wave = FR.synth_audio(N_HARD, 480000, seed=seed)
        mel_ref = torch.from_numpy(FR.audio_to_mel_batch(wave))
        with torch.no_grad():
            mel_dev, _ = fe(torch.from_numpy(wave).cuda(), clamp=True)
        sd0 = R.make_state_dict(model_type, 320, 512, 3, seed=seed + 1)
        for gain in (3.0, 4.0):
            sd, _ = R.trained_scale_state_dict(sd0, model_type, mel_ref[:1], w_hh_gain=gain)
            with torch.no_grad():
                ref = R.forward(sd, mel_ref, model_type, o=R.Opts(fast_lstm=True))
            got = _hip_logits(mta, model_type, sd, mel_dev)
            rel = float((got - ref).abs().max() / ref.abs().max())
            for sigma in (1.0, 0.3):
                g = torch.Generator().manual_seed(seed + int(10 * sigma) + int(gain))
                rolls = ((ref + sigma * torch.randn(ref.shape, generator=g)) > 0).float().numpy()
                thr = _tune(ref, rolls)
                for name, t in (("0.5", 0.5), ("tuned", thr)):
                    fr, fh = _f1_at(ref, rolls, t), _f1_at(got, rolls, t)
                    rows.append((seed, gain, sigma, name, t, fr, fh, fh - fr, rel))
                    worst = max(worst, abs(fh - fr))
    print(f"\n[{model_type} f16] seed gain sigma thr  F1 fp32 -> HIP (delta)  max|dlogit|/max|logit|")
    for r_ in rows:
        print(f"  {r_[0]} {r_[1]:.0f} {r_[2]:.1f} {r_[3]:>5s}={r_[4]:.2f}  {r_[5]:.5f} -> {r_[6]:.5f} ({r_[7]:+.5f})  {r_[8]:.4f}")
    assert worst <= F1_TOL, (worst, rows)
    assert max(r_[8] for r_ in rows) < 2.5e-2          # (gain 4: the recurrence amplifies the operand rounding; F1 is what is licensed)


@pytest.mark.parametrize("model_type", ["cnn_rnn", "cnn_rnn_large"])
def test_f1_parity_ragged_batch_padded_with_zero_db(mta, model_type):
    """Cached-format chunks (mel dB already clamped, T = 937), ragged lengths >= 50 % (data/dataset.py:82), right-padded with 0.0
    exactly as collate_fn pads: the same padded tensor goes through the HIP model and the fp32 oracle; per-sample F1 over the
    valid frames."""
    from music_transcription_amd import collate_fn
    seed, T = 505, 937
    wave = FR.synth_audio(N_HARD, 480000, seed=seed)
    mel_full = FR.audio_to_mel_batch(wave)[:, 0, :, :T]                      # (B, 320, 937): what the cache stores per chunk
    lengths = [937, 470, 800, 937, 600, 512, 700, 900]
    g = torch.Generator().manual_seed(seed)
    batch = [(torch.from_numpy(mel_full[i][None, :, :lengths[i]].copy()), torch.zeros(88, lengths[i])) for i in range(N_HARD)]
    mel_pad, _, lens = collate_fn(batch)                                     # (B, 1, 320, 937), pad value 0.0
    assert mel_pad.shape == (N_HARD, 1, 320, T) and float(mel_pad[1, 0, :, 500:].abs().max()) == 0.0
    sd0 = R.make_state_dict(model_type, 320, 512, 3, seed=seed + 1)
    sd, _ = R.trained_scale_state_dict(sd0, model_type, mel_pad[:1])
    with torch.no_grad():
        ref = R.forward(sd, mel_pad, model_type, o=R.Opts(fast_lstm=True))
    got = _hip_logits(mta, model_type, sd, mel_pad.cuda())
    assert got.shape == ref.shape == (N_HARD, 88, T)
    lens = [int(v) for v in lens]
    worst = 0.0
    for sigma in (1.0, 0.3):
        rolls = ((ref + sigma * torch.randn(ref.shape, generator=g)) > 0).float().numpy()
        for i, L in enumerate(lens):
            rolls[i, :, L:] = 0.0
        for t in (0.5, 0.3):
            fr, fh = _f1_valid(ref, rolls, lens, t), _f1_valid(got, rolls, lens, t)
            print(f"\n[{model_type} ragged 0-dB-padded T=937] sigma={sigma} thr={t}: F1 fp32 {fr:.5f} -> HIP {fh:.5f} ({fh - fr:+.5f})")
            worst = max(worst, abs(fh - fr))
    valid = torch.zeros(ref.shape, dtype=torch.bool)
    for i, L in enumerate(lens):
        valid[i, :, :L] = True
    print(f"max|dlogit| on valid frames {float((got - ref).abs()[valid].max()):.4f} of max|logit| {float(ref.abs()[valid].max()):.2f}")
    assert worst <= F1_TOL, worst
