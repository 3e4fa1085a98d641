#!/usr/bin/env python3
"""Golden vectors for the training step: the REFERENCE's own modules and its own
train_one_epoch (train/train_transcriber.py:90-158) run on CPU in this container, dropout = 0.

    python tests/golden/make_golden_train.py        # needs /root/reference (build container only)

Writes tests/golden/train_step.npz (CNNRNNModel) and tests/golden/train_step_large.npz (CNNRNNModelLarge):
  * cfg, seeds                       -- everything needed to regenerate weights (oracle.make_state_dict) and inputs
  * logits0, loss0, gradnorm0        -- train-mode forward / masked loss / global grad norm of step 1
  * grad::<key>                      -- every parameter's gradient of step 1 (before clipping)
  * bn0::<key>                       -- BatchNorm running statistics after step 1's forward
  * losses                           -- step losses returned by the reference's train_one_epoch over 3 batches
  * post::<key>                      -- every parameter / BN buffer after those 3 Adam steps
Only data is stored; no reference source text.
"""
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
from tests.golden.make_golden import mel_input, roll_input  # noqa: E402

torch.set_num_threads(8)

CFG = dict(n_mels=32, hidden=16, layers=2, B=3, T=40, seed_w=21, seed_x=31, lr=1e-4, n_batches=3)
MODEL_TYPE = "cnn_rnn"
OUT_NAME = "train_step.npz"


def batches():
    out = []
    for k in range(CFG["n_batches"]):
        mel = mel_input(CFG["B"], CFG["n_mels"], CFG["T"], seed=CFG["seed_x"] + k)
        roll = roll_input(CFG["B"], CFG["T"], seed=CFG["seed_x"] + 100 + k, p=0.1)
        lengths = torch.tensor([CFG["T"], CFG["T"] - 7, CFG["T"] - 15][:CFG["B"]], dtype=torch.int64)
        for b in range(CFG["B"]):            # collate_fn semantics: right-pad with 0.0 beyond each length
            mel[b, :, :, lengths[b]:] = 0.0
            roll[b, :, lengths[b]:] = 0.0
        out.append((mel, roll, lengths))
    return out


def fresh_model():
    from models.transcription_model import TranscriptionModel
    m = TranscriptionModel(model_type=MODEL_TYPE, n_mels=CFG["n_mels"], hidden_size=CFG["hidden"],
                           num_layers=CFG["layers"], dropout=0.0, device="cpu")
    sd = model_ref.make_state_dict(MODEL_TYPE, CFG["n_mels"], CFG["hidden"], CFG["layers"], CFG["seed_w"])
    m.load_state_dict(sd, strict=True)
    # CNNRNNModelLarge hard-codes its spatial dropouts (cnn_rnn_model.py:188,:192,:202); the golden step is
    # deterministic, so their probabilities are configured to 0 like every other dropout (dropout=0.0 above)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    return m


def main():
    for name in ("librosa", "pretty_midi"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    from train.train_transcriber import train_one_epoch

    out = {"cfg": np.array([CFG[k] for k in ("n_mels", "hidden", "layers", "B", "T", "seed_w", "seed_x", "n_batches")]),
           "lr": np.array(CFG["lr"])}
    data = batches()

    # step 1 by hand on the reference modules: logits, loss, gradients
    m = fresh_model()
    m.train()
    mel, roll, lengths = data[0]
    logits = m(mel)
    loss = m.compute_loss(logits, roll, lengths)
    loss.backward()
    out["logits0"] = logits.detach().numpy()
    out["loss0"] = np.array(loss.item())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))
    out["gradnorm0"] = np.array(gn.item())
    for k, p in m.named_parameters():                     # (the Large model's onset / offset heads get no gradient
        out["grad::" + k] = (p.grad.detach() if p.grad is not None else torch.zeros_like(p)).numpy()   # from the reference's loop)
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["bn0::" + k] = v.detach().numpy()

    # the reference's loop over all batches (Adam as scripts/train_cnn.py:290)
    m = fresh_model()
    opt = torch.optim.Adam(m.parameters(), lr=CFG["lr"], eps=1e-8, weight_decay=1e-5)
    avg, losses = train_one_epoch(m, data, opt, torch.device("cpu"), max_grad_norm=1.0)
    out["losses"] = np.array(losses)
    out["avg_loss"] = np.array(avg)
    for k, v in m.state_dict().items():
        out["post::" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, OUT_NAME), **out)
    print("%s: loss0 %.6f gradnorm0 %.4f losses %s" % (OUT_NAME, out["loss0"], out["gradnorm0"], losses))


if __name__ == "__main__":
    main()
    # the same through CNNRNNModelLarge (what the reference's canonical pipeline trains: example.sh:22)
    MODEL_TYPE, OUT_NAME = "cnn_rnn_large", "train_step_large.npz"
    CFG = dict(n_mels=32, hidden=16, layers=2, B=3, T=40, seed_w=23, seed_x=41, lr=1e-4, n_batches=3)
    main()
