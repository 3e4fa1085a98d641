"""Data-parallel training rehearsal: N ranks (torch.distributed.run), each with its own batches, one all-reduce (mean) of
the flat gradient per step; prints a checksum of the parameters per rank -- they must agree.  Backend: RCCL ("nccl") with
one GPU per rank, or MT_BENCH_BACKEND=gloo with all ranks on the visible GPU(s) (rehearsal on a one-GPU box)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
backend = os.environ.get("MT_BENCH_BACKEND", "nccl")
dev_index = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
import music_transcription_amd as mta
from oracle import model_ref

nm, H, L, B, T = 32, 16, 2, 3, 40
mtype = os.environ.get("MT_DP_MODEL", "cnn_rnn")                    # cnn_rnn | cnn_rnn_large (what example.sh trains)
model = mta.TranscriptionModel(mtype, n_mels=nm, hidden_size=H, num_layers=L, dropout=0.0, device=str(dev))
sd0 = model_ref.make_state_dict(mtype, nm, H, L, seed=21)
model.load_state_dict(sd0)
if mtype == "cnn_rnn_large":
    model.model.dropout2d_p = (0.0, 0.0, 0.0)
opt = mta.make_optimizer(model, lr=1e-3)
g = torch.Generator().manual_seed(1000 + rank)                      # different data on every rank
batches = []
for _ in range(3):
    mel = torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0
    roll = (torch.rand(B, 88, T, generator=g) < 0.1).float()
    batches.append((mel, roll, torch.tensor([T, T - 5, T - 11])))
n_early = [0]
if opt.early is not None:                                           # count the early reduces actually taken
    _orig = opt.early.reduce_early
    def _counting(grads, stream):
        took = _orig(grads, stream)
        n_early[0] += 1 if took else 0
        return took
    opt.early.reduce_early = _counting
avg, losses = mta.train_one_epoch(model, batches, opt, dev)
flat = torch.cat([p.detach().reshape(-1).double() for p in model.parameters()])
mine = torch.tensor([flat.sum().item(), flat.abs().sum().item(), (flat * torch.arange(flat.numel(), device=dev)).sum().item()], dtype=torch.float64)
where = dev if backend == "nccl" else torch.device("cpu")
mine = mine.to(where)
allv = [torch.zeros(3, dtype=torch.float64, device=where) for _ in range(world)]
dist.all_gather(allv, mine)
heads_untouched = None
if mtype == "cnn_rnn_large":                                        # the frame-only loss never reaches them: bit-for-bit the initial values
    sdm = model.state_dict()
    heads_untouched = all(torch.equal(sdm[k].cpu(), sd0[k]) for k in sd0 if "onset_head" in k or "offset_head" in k)
if rank == 0:
    same = all(torch.equal(allv[0].cpu(), v.cpu()) for v in allv)
    print(json.dumps({"world": world, "model": mtype, "identical_parameters": bool(same), "losses_rank0": losses, "checksum": allv[0].tolist(),
                      "early_bucket_reduces": n_early[0], "onset_offset_heads_untouched": heads_untouched}))
dist.destroy_process_group()
