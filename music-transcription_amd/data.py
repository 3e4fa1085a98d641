"""The reference's on-disk cache format and batch collation.

Cache (scripts/preprocess_dataset.py:66-69,:138-154,:172; reader data/cached_dataset.py:66-88):
    {cache_dir}/{split}/chunk_%06d.pt   = torch.save({'mel': f32 (1, n_mels, T), 'roll': f32 (88, T)})
    {cache_dir}/{split}_metadata.pkl    = pickle of a dict (keys below)
These run on the host (DataLoader workers in the reference), so they are plain CPU code.
"""
from __future__ import annotations

import os
import pickle
from typing import Iterable, List, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

METADATA_KEYS = ("root_dir", "chunk_length", "overlap", "split", "num_chunks", "chunks", "sr", "n_mels",
                 "hop_length", "return_waveform", "tokenize", "data_type")


def collate_fn(batch: Sequence[Tuple[torch.Tensor, torch.Tensor]]):
    """train/train_transcriber.py:23-39: right-pad mel (1,n_mels,Ti) and roll (88,Ti) with 0.0 to the
    batch max T (0.0 is 0 dB in the mel domain -- the reference does not mask it); lengths as int64."""
    lengths = [int(m.shape[-1]) for m, _ in batch]
    t_max = max(lengths)
    mel = torch.stack([F.pad(m, (0, t_max - m.shape[-1])) for m, _ in batch])
    roll = torch.stack([F.pad(r, (0, t_max - r.shape[-1])) for _, r in batch])
    return mel, roll, torch.tensor(lengths, dtype=torch.long)


def chunk_path(cache_dir: str, split: str, idx: int) -> str:
    return os.path.join(cache_dir, split, f"chunk_{idx:06d}.pt")


def write_cache_chunk(cache_dir: str, split: str, idx: int, mel: torch.Tensor, roll: torch.Tensor) -> str:
    """One record of the cache: mel (1, n_mels, T) f32 and roll (88, T) f32, trimmed to a common T
    as data/dataset.py:159-161 does."""
    mel, roll = mel.detach().to("cpu", torch.float32), roll.detach().to("cpu", torch.float32)
    if mel.dim() == 2:
        mel = mel[None]
    t = min(mel.shape[-1], roll.shape[-1])
    os.makedirs(os.path.join(cache_dir, split), exist_ok=True)
    path = chunk_path(cache_dir, split, idx)
    torch.save({"mel": mel[..., :t].contiguous(), "roll": roll[..., :t].contiguous()}, path)
    return path


def write_cache_metadata(cache_dir: str, split: str, chunks: List[dict], *, root_dir: str = "", chunk_length=30.0,
                         overlap=0.0, sr=16000, n_mels=320, hop_length=512) -> str:
    meta = {"root_dir": root_dir, "chunk_length": chunk_length, "overlap": overlap, "split": split,
            "num_chunks": len(chunks), "chunks": list(chunks), "sr": sr, "n_mels": n_mels, "hop_length": hop_length,
            "return_waveform": False, "tokenize": False, "data_type": "mel"}
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"{split}_metadata.pkl")
    with open(path, "wb") as f:
        pickle.dump(meta, f)
    return path


class CachedMaestroDataset(Dataset):
    """Reader with the reference's behaviour (data/cached_dataset.py:12-88): FileNotFoundError when the
    metadata, the split directory or a chunk is missing; items are (mel, roll) CPU tensors."""

    def __init__(self, cache_dir: str = "cached_dataset", split: str = "train"):
        self.cache_dir, self.split = cache_dir, split
        self.split_cache_dir = os.path.join(cache_dir, split)
        meta_path = os.path.join(cache_dir, f"{split}_metadata.pkl")
        if not os.path.exists(meta_path):
            raise FileNotFoundError(f"Cache not found at {meta_path}. Run preprocess_dataset.py first!")
        with open(meta_path, "rb") as f:
            self.metadata = pickle.load(f)
        self.num_chunks = self.metadata["num_chunks"]
        if not os.path.exists(self.split_cache_dir):
            raise FileNotFoundError(f"Cache directory not found: {self.split_cache_dir}. Run preprocess_dataset.py first!")

    def __len__(self):
        return self.num_chunks

    def __getitem__(self, idx):
        path = chunk_path(self.cache_dir, self.split, idx)
        if not os.path.exists(path):
            raise FileNotFoundError(f"Cached chunk not found: {path}. Re-run preprocess_dataset.py")
        rec = torch.load(path, weights_only=False)
        if "mel" not in rec:
            raise NotImplementedError("waveform / token caches belong to the AST experiment (out of scope)")
        return rec["mel"], rec["roll"]
