"""Cache writer: MAESTRO recordings -> the reference's preprocessed cache, end to end
(scripts/preprocess_dataset.py:77-257 driving data/dataset.py:57-167).

  * chunk index exactly as `MaestroDataset._build_chunk_index` (dataset.py:57-95): chunk = int(len*sr) samples,
    hop = int(chunk*(1-overlap)), a chunk is kept if it has >= 50 % of the chunk length, stop once a chunk
    reaches the end of the file; total_samples = int(duration*sr);
  * mel: the HIP frontend (csrc/mel.hip), all equal-length chunks of a recording in one batched launch -- the
    reference runs librosa per chunk on a host thread; labels: midi.chunk_roll (own SMF parser);
  * record = {'mel': (1, n_mels, T), 'roll': (88, T)} trimmed to the common T (dataset.py:159-161), files
    `{cache}/{split}/chunk_%06d.pt`, `{cache}/{split}_metadata.pkl` with the reference's keys.
Audio decoding is transcribe.load_audio_device (WAV, channel mean + polyphase resampling on the GPU: row f3, not
soxr-exact); the whole recording is resampled once and sliced, where librosa.load(offset, duration) resamples each slice
on its own.
"""
from __future__ import annotations

import csv
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import data as D
from .frontend import get_frontend
from .midi import MidiFile, chunk_roll
from .transcribe import load_audio_device


def build_chunk_index(durations: Sequence[float], chunk_length: float = 30.0, overlap: float = 0.0, sr: int = 16000) -> List[dict]:
    """Chunk positions for recordings of the given durations (seconds): list of the reference's chunk dicts."""
    chunk_samples = int(chunk_length * sr)
    hop_samples = int(chunk_samples * (1.0 - overlap))
    if hop_samples <= 0:
        raise ValueError("overlap must be < 1")
    chunks = []
    for file_idx, duration in enumerate(durations):
        total = int(duration * sr)
        start = 0
        while start < total:
            end = min(start + chunk_samples, total)
            if (end - start) >= chunk_samples * 0.5:
                chunks.append({"file_idx": file_idx, "start_sample": start, "end_sample": end,
                               "start_time": start / sr, "end_time": end / sr})
            start += hop_samples
            if end >= total:
                break
    return chunks


def read_maestro_csv(root_dir: str, split: Optional[str] = None, year=None, subset_size: Optional[int] = None,
                     csv_path: Optional[str] = None) -> List[Dict[str, str]]:
    """Rows of maestro-v3.0.0.csv filtered as data/dataset.py:34-50 does (year, official split, head(subset_size))."""
    csv_path = csv_path or os.path.join(root_dir, "maestro-v3.0.0.csv")
    with open(csv_path, newline="", encoding="utf-8") as f:
        rows = list(csv.DictReader(f))
    if year is not None:
        rows = [r for r in rows if int(r["year"]) == int(year)]
    if split is not None:
        rows = [r for r in rows if r["split"] == split]
    return rows[:subset_size] if subset_size else rows


def wav_duration(path: str) -> float:
    """Duration from the header (librosa.get_duration(path=...)): frames / native rate.  A missing `.wav` falls back to the `.mp3`
    beside it (data/dataset.py:68-70), whose duration comes from a host decode (transcribe.decode_compressed_host)."""
    import wave
    from .transcribe import decode_compressed_host, resolve_audio_path
    path = resolve_audio_path(path)
    with open(path, "rb") as fh:
        if fh.read(4) not in (b"RIFF", b"RIFX", b"RF64"):
            rate, data = decode_compressed_host(path)
            return data.shape[0] / float(rate)
    try:
        with wave.open(path, "rb") as w:
            return w.getnframes() / float(w.getframerate())
    except wave.Error:                                   # float / extensible WAVs: fall back to a full read
        from scipy.io import wavfile
        rate, data = wavfile.read(path, mmap=True)
        return data.shape[0] / float(rate)


def preprocess_and_cache(root_dir: str = "maestro-v3.0.0", cache_dir: str = "cached_dataset", chunk_length: float = 30.0,
                         overlap: float = 0.0, n_mels: int = 229, sr: int = 16000, hop_length: int = 512, split: str = "train",
                         force: bool = False, device="cuda", subset_size: Optional[int] = None, year=None, max_batch: int = 32,
                         rank: int = 0, world: int = 1, log=print) -> Dict[str, int]:
    """The reference's `preprocess_and_cache` (same leading arguments).  With world > 1 every rank writes the chunks of
    its own recordings (file_idx % world == rank; no collective: recordings are independent) and rank 0 the metadata."""
    rows = read_maestro_csv(root_dir, split, year, subset_size)
    paths = [os.path.join(root_dir, r["audio_filename"]) for r in rows]
    durations = [wav_duration(p) for p in paths]
    chunks = build_chunk_index(durations, chunk_length, overlap, sr)
    if rank == 0:
        D.write_cache_metadata(cache_dir, split, chunks, root_dir=root_dir, chunk_length=chunk_length, overlap=overlap, sr=sr,
                               n_mels=n_mels, hop_length=hop_length)
    os.makedirs(os.path.join(cache_dir, split), exist_ok=True)
    fe = get_frontend(sr, n_mels, hop_length, device)
    by_file: Dict[int, List[int]] = {}
    for idx, c in enumerate(chunks):
        by_file.setdefault(c["file_idx"], []).append(idx)
    stats = {"cached": 0, "skipped": 0, "failed": 0}
    for file_idx, idxs in by_file.items():
        if file_idx % world != rank:
            continue
        todo = [i for i in idxs if force or not os.path.exists(D.chunk_path(cache_dir, split, i))]
        stats["skipped"] += len(idxs) - len(todo)
        if not todo:
            continue
        try:
            y = load_audio_device(paths[file_idx], sr, device)          # stays on the GPU: decode, resample, mel
            midi = MidiFile(os.path.join(root_dir, rows[file_idx]["midi_filename"]))
            by_len: Dict[int, List[int]] = {}
            for i in todo:
                by_len.setdefault(chunks[i]["end_sample"] - chunks[i]["start_sample"], []).append(i)
            for n, group in by_len.items():
                for g0 in range(0, len(group), max_batch):
                    part = group[g0:g0 + max_batch]
                    wave = torch.zeros(len(part), n, dtype=torch.float32, device=y.device)
                    for k, i in enumerate(part):
                        seg = y[chunks[i]["start_sample"]:chunks[i]["end_sample"]]
                        wave[k, :seg.numel()] = seg
                    mel, _ = fe(wave, clamp=True)                                         # (len, 1, n_mels, T) dB, per-chunk top_db
                    mel = mel.cpu()
                    for k, i in enumerate(part):
                        roll = torch.from_numpy(chunk_roll(midi, chunks[i]["start_time"], chunks[i]["end_time"], sr, hop_length))
                        D.write_cache_chunk(cache_dir, split, i, mel[k], roll)
                        stats["cached"] += 1
        except Exception as e:                            # the reference logs and counts failures, and goes on
            log(f"Error processing recording {file_idx} ({paths[file_idx]}): {e}")
            stats["failed"] += len(todo)
    return stats
