#!/usr/bin/env python3
"""Framewise-F1 evaluation of a checkpoint over a cached split: the evaluation half of the reference's scripts/evaluate.py
(:335-379 headless loop, :524-618 threshold tuning; same flag names and defaults for what is kept).

    python scripts/evaluate.py --model outputs/.../checkpoints/model_best.pth --cache_dir cached_dataset_mels320 --headless
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/evaluate.py --model ... --headless

`--headless` prints exactly one line, `EVAL_MEAN_F1=<%.6f>` (what the reference's example.sh greps for).  The metric is the
reference's: per-sample binary F1 over the valid frames of the flattened (88, L) roll with zero_division = 0, unweighted mean
over the samples.  On the device: the model runs once per sample (samples of equal length batched -- no padding arises, so each
sample's logits are what batch 1 gives), thresholding and the TP / FP / FN counts are one integer pass (mt_f1_sweep_counts);
with `--tune_threshold` every candidate threshold of the reference's coarse-to-fine schedule is another counts pass over the SAME
logits instead of another run of the model.  With more than one rank the samples are sharded contiguously (no data-path
collective) and the per-sample values gathered by one small all-reduce.  Full-file evaluation (`--data_source full`), MIDI /
plot outputs, background mode and the results browser are out of scope.
"""
import argparse
import json
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser(description="Evaluate a transcription checkpoint (framewise F1) on a cached split")
    ap.add_argument("--model", type=str, required=True, help="Path to model checkpoint (.pth file)")
    ap.add_argument("--split", type=str, default="test", choices=["train", "validation", "test"])
    ap.add_argument("--threshold", type=float, default=0.5, help="Sigmoid threshold for binary prediction (default: 0.5)")
    ap.add_argument("--subset", type=int, default=None, help="Limit number of samples (for quick eval)")
    ap.add_argument("--batch_size", type=int, default=1, help="accepted for compatibility: samples of equal length are batched on the device")
    ap.add_argument("--data_source", type=str, default="auto", choices=["auto", "cache", "full"])
    ap.add_argument("--cache_dir", type=str, default="cached_dataset_mels320")
    ap.add_argument("--n_mels", type=int, default=None, help="Number of mel bins (auto-detected from cache if not specified)")
    ap.add_argument("--model_type", type=str, default="cnn_rnn_large")
    ap.add_argument("--hidden_size", type=int, default=512)
    ap.add_argument("--num_layers", type=int, default=3)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--out_dir", type=str, default="eval_outputs", help="results.json is written here unless --headless")
    ap.add_argument("--headless", action="store_true", help="Headless mode: only print EVAL_MEAN_F1=<value>")
    ap.add_argument("--tune_threshold", action="store_true")
    ap.add_argument("--tune_rounds", type=int, default=6)
    ap.add_argument("--tune_range", type=float, nargs=2, default=[0.05, 0.95])
    ap.add_argument("--tune_step", type=float, default=0.1)
    ap.add_argument("--tune_min_step", type=float, default=0.01)
    args = ap.parse_args()
    say = (lambda *a, **k: None) if args.headless else print

    if not os.path.exists(args.model):
        print(f"Error: Model checkpoint not found: {args.model}")
        return 1
    meta_path = os.path.join(args.cache_dir, f"{args.split}_metadata.pkl")
    if args.data_source == "full" or not os.path.exists(meta_path):
        print(f"Error: no cached split at {meta_path} (full-file evaluation is not part of this build: run scripts/preprocess_dataset.py first)")
        return 1
    n_mels = args.n_mels
    if n_mels is None:                                  # evaluate.py:151-156: n_mels from the cache metadata
        with open(meta_path, "rb") as f:
            n_mels = pickle.load(f).get("n_mels", 320)
        say(f"Auto-detected n_mels={n_mels} from cache metadata")

    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("MT_BENCH_BACKEND", "nccl")
    if not torch.cuda.is_available():
        print("Error: music_transcription_amd evaluates on the GPU only")
        return 1
    dev_index = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import music_transcription_amd as mta
    from music_transcription_amd import evaluate as E

    say(f"Using device: {dev}")
    model = mta.TranscriptionModel(model_type=args.model_type, device=dev, n_mels=n_mels, hidden_size=args.hidden_size,
                                   num_layers=args.num_layers, dropout=args.dropout)
    model.load_state_dict(torch.load(args.model, map_location=dev))
    model.eval()
    say(f"Loading cached dataset from: {args.cache_dir}")
    ds = mta.CachedMaestroDataset(args.cache_dir, args.split)
    threshold = args.threshold
    if args.tune_threshold:
        threshold, tuned_f1 = E.tune_threshold(model, ds, dev, subset=args.subset, tune_range=tuple(args.tune_range), tune_step=args.tune_step,
                                               tune_min_step=args.tune_min_step, tune_rounds=args.tune_rounds, rank=rank, world=world,
                                               log=say if rank == 0 else None)
        say(f"Best threshold: {threshold:.4f} (mean F1 {tuned_f1:.6f})")
    mean_f1, per_sample = E.evaluate_dataset(model, ds, threshold, dev, subset=args.subset, rank=rank, world=world)
    if rank == 0:
        if args.headless:
            print(f"EVAL_MEAN_F1={mean_f1:.6f}")
        else:
            print(f"\nMean framewise F1 over {len(per_sample)} samples at threshold {threshold:.4f}: {mean_f1:.6f}")
            os.makedirs(args.out_dir, exist_ok=True)
            with open(os.path.join(args.out_dir, "results.json"), "w") as f:
                json.dump({"mean_f1": mean_f1, "threshold": threshold, "per_sample_f1": per_sample, "split": args.split,
                           "num_samples": len(per_sample), "model": args.model, "model_type": args.model_type}, f)
            print(f"Results written to {os.path.join(args.out_dir, 'results.json')}")
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
