#!/usr/bin/env python3
"""MAESTRO -> preprocessed cache with the HIP mel frontend (the reference's scripts/preprocess_dataset.py, same flags
for the mel path; waveform/token caches belong to the AST experiment and are out of scope).

    python scripts/preprocess_dataset.py --root_dir maestro-v3.0.0 --cache_dir cached_dataset --n_mels 320 --chunk_length 30
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/preprocess_dataset.py ...   # recordings sharded over GPUs
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_dir", default="maestro-v3.0.0")
    ap.add_argument("--cache_dir", default="cached_dataset")
    ap.add_argument("--chunk_length", type=float, default=30.0)
    ap.add_argument("--overlap", type=float, default=0.0)
    ap.add_argument("--n_mels", type=int, default=229)
    ap.add_argument("--sr", type=int, default=16000)
    ap.add_argument("--hop_length", type=int, default=512)
    ap.add_argument("--splits", nargs="+", default=["train", "validation", "test"])
    ap.add_argument("--subset_size", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    import torch
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    from music_transcription_amd.preprocess import preprocess_and_cache
    for split in args.splits:
        st = preprocess_and_cache(args.root_dir, args.cache_dir, args.chunk_length, args.overlap, args.n_mels, args.sr, args.hop_length,
                                  split, args.force, device=f"cuda:{torch.cuda.current_device()}", subset_size=args.subset_size,
                                  rank=rank, world=world)
        print(f"[rank {rank}] {split}: {st}", flush=True)


if __name__ == "__main__":
    main()
