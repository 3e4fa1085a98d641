import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.getcwd())
import music_transcription_amd as mta
from oracle import model_ref as R
g = np.load('tests/golden/train_step_large.npz')
nm, H, L, B, T, sw, sx, nb = [int(v) for v in g["cfg"]]
m = mta.TranscriptionModel(model_type="cnn_rnn_large", n_mels=nm, hidden_size=H, num_layers=L, dropout=0.0, device="cuda")
m.load_state_dict(R.make_state_dict("cnn_rnn_large", nm, H, L, sw), strict=True)
m.model.dropout2d_p = (0.0, 0.0, 0.0)
opt = mta.make_optimizer(m, lr=1e-4)
m.train()
mel = torch.randn(B, 1, nm, T).cuda(); roll = (torch.rand(B, 88, T) < 0.1).float().cuda()
opt.zero_grad()
loss = m.compute_loss(m(mel), roll, None)
loss.backward()
print("skip names", sorted(getattr(m.model, "_params_without_grad", ())))
print("none grads", [k for k, p in m.named_parameters() if p.grad is None])
print("keep", opt._keep_ranges(), opt.g.numel())
w0 = m.model.onset_head.weight.detach().clone()
opt.step(sync_grads=False)
print("onset changed", float((m.model.onset_head.weight.detach() - w0).abs().max()))
