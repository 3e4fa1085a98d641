#!/usr/bin/env python3
"""Cache-format compatibility fixture: a tiny cache written by THIS package's writer
(music_transcription_amd.data.write_cache_*) and read back by the REFERENCE's own readers
(data/cached_dataset.py:66-88 CachedMaestroDataset, :97-120 HybridMaestroDataset), imported unchanged from
/root/reference on CPU in the build container.

    python tests/golden/make_golden_cache.py        # needs /root/reference and the built libmt_hip.so

Writes tests/golden/cache_fixture/ (the cache: two splits, 3 + 2 records, a few KB) and
tests/golden/cache_fixture.json: what the reference's readers returned for it (lengths, shapes, dtypes, float64
checksums, metadata keys, and HybridMaestroDataset's decision to use the cache).  tests/test_host_cpu.py then checks
this package's reader against that record.  Only data is stored; no reference source text.
"""
import json
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

N_MELS, SPLITS = 16, (("train", (13, 9, 13)), ("validation", (7, 13)))


def main():
    import music_transcription_amd as mta
    cache = os.path.join(HERE, "cache_fixture")
    shutil.rmtree(cache, ignore_errors=True)
    g = torch.Generator().manual_seed(29)
    for split, Ts in SPLITS:
        chunks = []
        for i, t in enumerate(Ts):
            mel = torch.randn(1, N_MELS, t + 1, generator=g) * 20.0 - 40.0          # one frame longer than the roll: trimmed to min_len
            roll = (torch.rand(88, t, generator=g) < 0.1).float()
            mta.write_cache_chunk(cache, split, i, mel, roll)
            chunks.append(dict(file_idx=i, start_sample=0, end_sample=480000, start_time=0.0, end_time=30.0))
        mta.write_cache_metadata(cache, split, chunks, root_dir="maestro-v3.0.0", chunk_length=30.0, overlap=0.0, sr=16000,
                                 n_mels=N_MELS, hop_length=512)

    from data.cached_dataset import CachedMaestroDataset, HybridMaestroDataset     # the reference's readers, unchanged
    rec = {"n_mels": N_MELS, "splits": {}}
    for split, Ts in SPLITS:
        ds = CachedMaestroDataset(cache_dir=cache, split=split)
        items = []
        for i in range(len(ds)):
            mel, roll = ds[i]
            items.append({"mel_shape": list(mel.shape), "roll_shape": list(roll.shape), "mel_dtype": str(mel.dtype), "roll_dtype": str(roll.dtype),
                          "mel_sum": float(mel.double().sum()), "mel_abs_sum": float(mel.double().abs().sum()),
                          "roll_sum": float(roll.double().sum()),
                          "roll_checksum": float((roll.double() * torch.arange(roll.numel()).reshape(roll.shape)).sum())})
        hy = HybridMaestroDataset(root_dir="does-not-exist", cache_dir=cache, split=split, chunk_length=30.0, overlap=0.0)
        rec["splits"][split] = {"len": len(ds), "items": items, "metadata_keys": sorted(ds.metadata.keys()),
                                "metadata": {k: v for k, v in ds.metadata.items() if k != "chunks"},
                                "chunk_keys": sorted(ds.metadata["chunks"][0].keys()),
                                "hybrid_uses_cache": bool(hy.use_cache), "hybrid_len": len(hy)}
        assert hy.use_cache and len(hy) == len(Ts)
    with open(os.path.join(HERE, "cache_fixture.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("cache fixture written:", cache)


if __name__ == "__main__":
    main()
