#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

The reference model code is imported unchanged from /root/reference (nothing is
copied); weights come from oracle.model_ref.make_state_dict(seed) loaded with
load_state_dict(strict=True), inputs from seeded torch generators.  The fixtures
hold inputs' seeds, outputs (full for small configs, strided samples + summary
statistics for canonical ones) and a weight checksum so that RNG drift is caught.
`train.train_transcriber` has top-level imports of librosa / pretty_midi (absent
here); inert empty module objects are registered for those two names only so that
`collate_fn` can be imported -- no functionality of either library is emulated.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import model_ref  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def ref_model(model_type, n_mels, hidden, layers, seed, calib_T=64, **kw):
    from models.transcription_model import TranscriptionModel
    m = TranscriptionModel(model_type=model_type, n_mels=n_mels, hidden_size=hidden,
                           num_layers=layers, dropout=0.2, device="cpu", **kw)
    sd = model_ref.make_state_dict(model_type, n_mels, hidden, layers, seed,
                                   use_attention=kw.get("use_attention", True),
                                   use_heads=kw.get("use_onset_offset_heads", True))
    m.load_state_dict(sd, strict=True)
    # calibrate the BatchNorm running statistics on seeded inputs (reference in train mode, cumulative
    # average), as a trained checkpoint's would be: activations then stay O(1) through the network
    if calib_T:
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
                mod.momentum = None
        m.train()
        with torch.no_grad():
            for k in range(3):
                m(mel_input(2, n_mels, calib_T, seed=100 + k))
        for k, v in m.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                sd[k] = v.detach().clone()
    m.eval()
    return m, sd


def checksum(sd):
    return float(sum(v.double().abs().sum().item() for k, v in sd.items() if v.dtype.is_floating_point))


def mel_input(B, n_mels, T, seed):
    g = torch.Generator().manual_seed(seed)
    # dB-like range [-80, 0] with structure
    return (torch.rand(B, 1, n_mels, T, generator=g) * 60.0 - 70.0
            + 10.0 * torch.randn(B, 1, n_mels, 1, generator=g))


def roll_input(B, T, seed, p=0.04):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 88, T, generator=g) < p).float()


def main():
    out = {}

    # (i) state_dict manifests -------------------------------------------------
    from models.transcription_model import TranscriptionModel
    manifest = {}
    for mt in ("cnn_rnn", "cnn_rnn_large"):
        for (nm, hs, nl) in ((320, 512, 3), (229, 256, 2)):
            m = TranscriptionModel(model_type=mt, n_mels=nm, hidden_size=hs, num_layers=nl, device="cpu")
            manifest[f"{mt}:{nm}:{hs}:{nl}"] = {
                "keys": {k: list(v.shape) for k, v in m.state_dict().items()},
                "n_params": int(sum(p.numel() for p in m.parameters())),
            }
    m = TranscriptionModel(model_type="large", n_mels=64, hidden_size=32, num_layers=2, device="cpu",
                           use_attention=False, use_onset_offset_heads=False)
    manifest["large:64:32:2:noattn:noheads"] = {"keys": {k: list(v.shape) for k, v in m.state_dict().items()},
                                                "n_params": int(sum(p.numel() for p in m.parameters()))}
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)

    # (ii)/(iii) small configs, full outputs --------------------------------------
    small = {}
    with torch.no_grad():
        for tag, mt, nm, hs, nl, B, T in (("small_a", "cnn_rnn", 32, 16, 2, 2, 50),
                                          ("small_b", "cnn_rnn", 40, 32, 3, 3, 37),
                                          ("large_a", "cnn_rnn_large", 32, 16, 2, 2, 50),
                                          ("large_b", "cnn_rnn_large", 48, 32, 3, 1, 41)):
            m, sd = ref_model(mt, nm, hs, nl, seed=11)
            x = mel_input(B, nm, T, seed=5)
            small[f"{tag}_cfg"] = np.array([nm, hs, nl, B, T, 11, 5])
            small[f"{tag}_wsum"] = np.array(checksum(sd))
            small[f"{tag}_bn"] = model_ref.get_bn_flat(sd)
            small[f"{tag}_logits"] = m(x).numpy()
            if mt == "cnn_rnn_large":
                d = m(x, return_all_heads=True)
                for k in ("frame", "onset", "offset"):
                    small[f"{tag}_{k}"] = d[k].numpy()
            # (viii) predict at thresholds
            for th in (0.3, 0.5, 0.7):
                small[f"{tag}_pred{int(th * 10)}"] = np.packbits(m.predict(x, threshold=th).numpy().astype(np.uint8))
            # zero-length input: the reference's T==0 guard (cnn_rnn_model.py:65-66) is
            # unreachable -- its first Conv2d raises RuntimeError for T=0.  Record that.
            try:
                m(torch.zeros(B, 1, nm, 0))
                small[f"{tag}_zero_raises"] = np.array(0)
            except RuntimeError:
                small[f"{tag}_zero_raises"] = np.array(1)
        # variants without attention / heads
        m, sd = ref_model("large", 32, 16, 2, seed=12, use_attention=False, use_onset_offset_heads=True)
        x = mel_input(2, 32, 30, seed=6)
        small["large_noattn_bn"] = model_ref.get_bn_flat(sd)
        small["large_noattn_logits"] = m(x).numpy()
        m, sd = ref_model("large", 32, 16, 2, seed=13, use_attention=True, use_onset_offset_heads=False)
        small["large_noheads_bn"] = model_ref.get_bn_flat(sd)
        small["large_noheads_logits"] = m(x).numpy()

        # (vii) padded batch: collate semantics leak padding into valid frames
        m, sd = ref_model("cnn_rnn", 32, 16, 2, seed=11)
        small["pad_bn"] = model_ref.get_bn_flat(sd)
        xa = mel_input(1, 32, 50, seed=7)
        xb = mel_input(1, 32, 30, seed=8)
        xp = torch.cat([xa, torch.nn.functional.pad(xb, (0, 20))], dim=0)
        small["pad_logits_batch"] = m(xp).numpy()
        small["pad_logits_solo_b"] = m(xb).numpy()
    np.savez_compressed(os.path.join(HERE, "small_models.npz"), **small)

    # canonical configs: strided samples -------------------------------------------
    canon = {}
    with torch.no_grad():
        for tag, mt, B, T in (("small_937", "cnn_rnn", 1, 937), ("small_938", "cnn_rnn", 2, 938),
                              ("large_937", "cnn_rnn_large", 1, 937), ("large_938", "cnn_rnn_large", 2, 938)):
            m, sd = ref_model(mt, 320, 512, 3, seed=21)
            x = mel_input(B, 320, T, seed=9)
            y = m(x).numpy()
            canon[f"{tag}_cfg"] = np.array([320, 512, 3, B, T, 21, 9])
            canon[f"{tag}_wsum"] = np.array(checksum(sd))
            canon[f"{tag}_bn"] = model_ref.get_bn_flat(sd)
            canon[f"{tag}_sample"] = y[:, ::5, ::7].copy()
            canon[f"{tag}_stats"] = np.array([y.mean(), y.std(), np.abs(y).max(), y.min(), y.max()], dtype=np.float64)
            if mt == "cnn_rnn_large" and B == 1:
                d = m(x, return_all_heads=True)
                canon[f"{tag}_onset_sample"] = d["onset"].numpy()[:, ::5, ::7].copy()
                canon[f"{tag}_offset_sample"] = d["offset"].numpy()[:, ::5, ::7].copy()
    np.savez_compressed(os.path.join(HERE, "canonical_models.npz"), **canon)

    # (iv) losses -------------------------------------------------------------------
    loss = {}
    m, _ = ref_model("cnn_rnn_large", 32, 16, 2, seed=11, calib_T=0)
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, 88, 40, generator=g) * 2.0
    targets = roll_input(3, 40, seed=4, p=0.1)
    lengths = torch.tensor([40, 25, 1])
    loss["logits"] = logits.numpy()
    loss["targets"] = np.packbits(targets.numpy().astype(np.uint8))
    loss["lengths"] = lengths.numpy()
    loss["loss_nolen"] = m.compute_loss(logits, targets).numpy()
    loss["loss_len"] = m.compute_loss(logits, targets, lengths).numpy()
    d = {"frame": logits, "onset": logits * 0.5 - 1.0, "offset": -logits + 0.25}
    loss["loss_dict_nolen"] = m.compute_loss(d, targets).numpy()
    loss["loss_dict_len"] = m.compute_loss(d, targets, lengths).numpy()
    loss["loss_len_zero"] = m.compute_loss(logits, targets, torch.tensor([0, 0, 0])).numpy()
    lg = logits.clone().requires_grad_(True)
    m.compute_loss(lg, targets, lengths).backward()
    loss["grad_len"] = lg.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "loss.npz"), **loss)

    # (v) collate_fn (inert module objects for the two absent top-level imports) ------
    for name in ("librosa", "pretty_midi"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from train.train_transcriber import collate_fn
    g = torch.Generator().manual_seed(17)
    batch = [(torch.randn(1, 8, t, generator=g), (torch.rand(88, t, generator=g) < 0.1).float()) for t in (13, 9, 13, 4)]
    mel, roll, lens = collate_fn(batch)
    np.savez_compressed(os.path.join(HERE, "collate.npz"), mel=mel.numpy(), roll=roll.numpy(), lengths=lens.numpy(),
                        seed=np.array(17), Ts=np.array([13, 9, 13, 4]))

    # (ix) F1: sklearn semantics used by scripts/evaluate.py:369-373 ---------------------
    from sklearn.metrics import f1_score
    rng = np.random.default_rng(5)
    cases_t, cases_p, vals = [], [], []
    for k in range(8):
        n = 88 * 20
        yt = (rng.random(n) < (0.0 if k == 0 else 0.1)).astype(np.float32)
        yp = (rng.random(n) < (0.0 if k in (0, 1) else 0.12)).astype(np.float32)
        if k == 2:
            yp = yt.copy()
        if k == 3:
            yt[:] = 0
        cases_t.append(yt); cases_p.append(yp)
        vals.append(f1_score(yt, yp, zero_division=0))
    np.savez_compressed(os.path.join(HERE, "f1.npz"), y_true=np.stack(cases_t), y_pred=np.stack(cases_p),
                        f1=np.array(vals, dtype=np.float64))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
