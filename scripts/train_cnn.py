#!/usr/bin/env python3
"""BASELINE.json configs[3]: chunked training of the CNN-RNN transcriber from a preprocessed cache, data-parallel
over the GPUs of one node (the training half of the reference's scripts/train_cnn.py:86-372, same flag names).

    python scripts/train_cnn.py --cached_dir cached_dataset --batch_size 16 --epochs 25
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_cnn.py ...

One process per GPU.  Every step is the HIP training step (train-mode forward, backward, fused clip + Adam); with more
than one rank each rank draws its own shard of the shuffled chunk indices (DistributedSampler) and the flat gradient
is all-reduced (mean) over RCCL before the clip, so all ranks hold identical weights.  Checkpoints are plain
`state_dict` files that the reference's TranscriptionModel loads unchanged, under the reference's names
(scripts/train_cnn.py:345-358): `checkpoints/model_epoch_N.pth` every --save_every epochs and after the last one,
`checkpoints/model_best.pth` whenever the validation loss improves, `checkpoints/model_final.pth` at the end.  Background
re-execution, run-directory bookkeeping and loss plots of the reference script are out of scope.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# The training step overlaps its weight-gradient work on two side streams; streams that share a hardware queue run one after the other.  With 32
# queues (the runtime's default is 4) no two of a training process's streams share one (profiles/r04_train_large_stream_mapping.txt).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cached_dir", default="cached_dataset")
    ap.add_argument("--subset_size", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=8, help="per GPU")
    ap.add_argument("--epochs", type=int, default=25)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--save_every", type=int, default=10)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--start_epoch", type=int, default=1)
    ap.add_argument("--model", default="cnn_rnn")
    ap.add_argument("--n_mels", type=int, default=320)
    ap.add_argument("--hidden_size", type=int, default=512)
    ap.add_argument("--num_layers", type=int, default=3)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--use_attention", action="store_true", default=True, help="attention block (cnn_rnn_large only)")
    ap.add_argument("--no_attention", action="store_false", dest="use_attention")
    ap.add_argument("--use_onset_offset_heads", action="store_true", default=True, help="onset / offset heads (cnn_rnn_large only)")
    ap.add_argument("--no_onset_offset_heads", action="store_false", dest="use_onset_offset_heads")
    ap.add_argument("--run_dir", default="outputs/train_cnn")
    ap.add_argument("--num_workers", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from torch.utils.data import DataLoader, Subset
    from torch.utils.data.distributed import DistributedSampler
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("MT_BENCH_BACKEND", "nccl")
    dev_index = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import music_transcription_amd as mta
    from music_transcription_amd import train as T

    if args.model not in ("cnn_rnn", "cnn+rnn", "cnn_rnn_large", "large"):
        raise SystemExit(f"model={args.model}: the HIP training step exists for cnn_rnn and cnn_rnn_large")
    torch.manual_seed(args.seed)                       # same initial weights on every rank
    train_ds = mta.CachedMaestroDataset(args.cached_dir, "train")
    val_ds = mta.CachedMaestroDataset(args.cached_dir, "validation")
    if args.subset_size:
        train_ds = Subset(train_ds, range(min(args.subset_size, len(train_ds))))
        val_ds = Subset(val_ds, range(min(max(1, args.subset_size // 4), len(val_ds))))
    sampler = DistributedSampler(train_ds, num_replicas=world, rank=rank, shuffle=True, seed=args.seed, drop_last=True) if world > 1 else None
    kw = dict(collate_fn=mta.collate_fn, num_workers=args.num_workers, pin_memory=True)
    train_loader = DataLoader(train_ds, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler, drop_last=world > 1, **kw)
    val_loader = DataLoader(val_ds, batch_size=args.batch_size, shuffle=False, **kw)

    model = mta.TranscriptionModel(model_type=args.model, n_mels=args.n_mels, hidden_size=args.hidden_size, num_layers=args.num_layers,
                                   dropout=args.dropout, device=str(dev), use_attention=args.use_attention,
                                   use_onset_offset_heads=args.use_onset_offset_heads)
    start_epoch = args.start_epoch
    if args.resume:
        model.load_state_dict(torch.load(args.resume, map_location=dev))
        m = re.search(r"epoch_(\d+)", os.path.basename(args.resume))
        if m and args.start_epoch == 1:
            start_epoch = int(m.group(1)) + 1
    opt = mta.make_optimizer(model, lr=args.lr, eps=1e-8, weight_decay=1e-5)
    ckpt_dir = os.path.join(args.run_dir, "checkpoints")
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
    history = []
    best_val_loss = float("inf")

    def save(name):                                    # plain state_dict with the reference's keys (rank 0 only)
        path = os.path.join(ckpt_dir, name)
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
        return path
    for epoch in range(start_epoch, args.epochs + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        t0 = time.perf_counter()
        train_loss, step_losses = T.train_one_epoch(model, train_loader, opt, dev, max_grad_norm=1.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        val_loss = T.evaluate(model, val_loader, dev) if rank == 0 else float("nan")
        if rank == 0:
            rec = {"epoch": epoch, "train_loss": train_loss, "val_loss": val_loss, "steps": len(step_losses),
                   "chunks_per_s": round(len(step_losses) * args.batch_size * world / max(dt, 1e-9), 2)}
            history.append(rec)
            print(json.dumps(rec), flush=True)
            if epoch % args.save_every == 0 or epoch == args.epochs:
                print(f"Checkpoint saved to {save(f'model_epoch_{epoch}.pth')}", flush=True)
            if val_loss < best_val_loss:               # best model = lowest validation loss (reference scripts/train_cnn.py:349-354)
                best_val_loss = val_loss
                save("model_best.pth")
                print(f"New best model saved! Val loss: {val_loss:.4f}", flush=True)
        if world > 1:
            dist.barrier()
    if rank == 0:
        print(f"Final model saved to {save('model_final.pth')}", flush=True)       # reference scripts/train_cnn.py:180,:357
        with open(os.path.join(args.run_dir, "history.json"), "w") as f:
            json.dump(history, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
