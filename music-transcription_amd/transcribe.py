"""Audio file -> MIDI with the HIP hot path: the pipeline of the reference's main.py (:60-100 chunking,
:103-161 mel + predict, :164-226 roll -> notes, :229-287 driver), restructured for the GPU:

  * all 30 s chunks of a recording go through ONE batched mel launch and ONE batched forward
    (the reference loops batch-1 on the host, main.py:258-266); chunks are exactly 480000 samples,
    so no time padding arises and batching is parity-safe;
  * the thresholded roll comes back once, notes are run-length decoded per pitch on the host
    (as the reference does) and written by a dependency-free Standard MIDI File writer.

Audio decoding: librosa/soundfile are not available in this image, so only WAV (PCM 16/24/32-bit or
float) is read, channels are averaged (librosa mono=True) and other sample rates are resampled with
scipy's polyphase filter -- NOT sample-exact with librosa's soxr_hq.
"""
from __future__ import annotations

import os
import struct
from pathlib import Path
from typing import List, Tuple

import numpy as np
import torch

from .frontend import get_frontend, num_frames
from .model import TranscriptionModel
from . import ops

MODEL_TYPE, N_MELS, HIDDEN_SIZE, NUM_LAYERS, DROPOUT = "cnn_rnn_large", 320, 512, 3, 0.2     # main.py:16-20
SR, HOP_LENGTH, CHUNK_LENGTH, THRESHOLD = 16000, 512, 30.0, 0.5                                # main.py:21-24


def load_audio(path: str, sr: int = SR) -> np.ndarray:
    """Mono float32 at `sr`.  WAV only (see module docstring)."""
    from scipy.io import wavfile
    try:
        rate, data = wavfile.read(path)
    except ValueError as e:
        raise ValueError(f"{path}: only WAV input is supported in this build ({e})")
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    else:
        data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1)
    if rate != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(rate), int(sr))
        data = resample_poly(data, sr // g, rate // g).astype(np.float32)
    return np.ascontiguousarray(data, dtype=np.float32)


def split_into_chunks(y: np.ndarray, chunk_length: float = CHUNK_LENGTH, sr: int = SR) -> Tuple[np.ndarray, float]:
    """main.py:60-100: ceil(len / chunk) chunks, the last one zero-padded in the waveform domain.  -> (n, chunk_samples)."""
    chunk = int(chunk_length * sr)
    n = max(1, int(np.ceil(len(y) / chunk)))
    out = np.zeros((n, chunk), dtype=np.float32)
    out.reshape(-1)[:len(y)] = y[: n * chunk]
    return out, len(y) / sr


def pianoroll_to_notes(roll: np.ndarray, fs: float, min_midi: int = 21) -> List[Tuple[int, float, float]]:
    """main.py:189-226: per pitch, runs of active frames -> (pitch, start_s, end_s)."""
    notes = []
    for i in range(roll.shape[0]):
        active = (roll[i] > 0).astype(np.int8)
        d = np.diff(np.concatenate([[0], active, [0]]))
        for s, e in zip(np.where(d == 1)[0], np.where(d == -1)[0]):
            if e > s:
                notes.append((min_midi + i, s / fs, e / fs))
    return notes


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append(0x80 | (n & 0x7F))
        n >>= 7
    return bytes(reversed(out))


def write_midi(notes: List[Tuple[int, float, float]], path: str, velocity: int = 100, resolution: int = 220, tempo_bpm: float = 120.0):
    """Format-1 SMF: a tempo track and one Acoustic Grand Piano track (what pretty_midi.PrettyMIDI().write emits
    for one Instrument(program=0) at its default resolution 220 / 120 bpm)."""
    tick = lambda t: int(round(t * resolution * tempo_bpm / 60.0))
    ev = []
    for pitch, s, e in notes:
        ev.append((tick(s), 1, 0x90, pitch, velocity))
        ev.append((tick(e), 0, 0x80, pitch, 0))          # note-offs first at equal ticks
    ev.sort(key=lambda x: (x[0], x[1], x[3]))
    trk = bytearray(b"\x00\xC0\x00")                     # program change: channel 0, program 0
    last = 0
    for t, _, status, pitch, vel in ev:
        trk += _vlq(t - last) + bytes([status, pitch, vel])
        last = t
    trk += b"\x01\xFF\x2F\x00"
    tempo = int(round(60e6 / tempo_bpm))
    meta = b"\x00\xFF\x51\x03" + struct.pack(">I", tempo)[1:] + b"\x00\xFF\x58\x04\x04\x02\x18\x08" + b"\x01\xFF\x2F\x00"
    with open(path, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, 1, 2, resolution))
        f.write(b"MTrk" + struct.pack(">I", len(meta)) + meta)
        f.write(b"MTrk" + struct.pack(">I", len(trk)) + bytes(trk))


def load_model(model_path: str, device: str = "cuda", model_type: str = MODEL_TYPE, n_mels: int = N_MELS,
               hidden_size: int = HIDDEN_SIZE, num_layers: int = NUM_LAYERS) -> TranscriptionModel:
    """main.py:27-57: plain state_dict checkpoint, eval mode."""
    model = TranscriptionModel(model_type=model_type, n_mels=n_mels, hidden_size=hidden_size, num_layers=num_layers,
                               dropout=DROPOUT, device=device)
    model.load_state_dict(torch.load(model_path, map_location=device))
    model.eval()
    return model


@torch.no_grad()
def transcribe_chunks(model: TranscriptionModel, chunks: np.ndarray, threshold: float = THRESHOLD, batch: int = 32,
                      n_mels: int = N_MELS, device: str = "cuda") -> np.ndarray:
    """(n, 480000) waveform chunks -> (88, n * T) {0,1} roll; mel + forward + threshold all on the GPU."""
    fe = get_frontend(SR, n_mels, HOP_LENGTH, device)
    rolls = []
    for i in range(0, len(chunks), batch):
        wave = torch.from_numpy(chunks[i:i + batch]).to(device)
        mel, cmax = fe(wave, clamp=False)
        net = model.model
        logits = net(mel, chunk_max_power=cmax)
        roll = ops.predict_from_logits(logits, threshold)                      # (b, 88, T)
        rolls.append(roll.permute(1, 0, 2).reshape(88, -1).cpu().numpy())      # combine_piano_rolls: concat along time
    return np.concatenate(rolls, axis=1)


def transcribe_audio(audio_path: str, model_path: str, output_path=None, device=None, threshold: float = THRESHOLD, **model_kw):
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    if device != "cuda":
        raise RuntimeError("music_transcription_amd runs on the GPU only (-d cuda)")
    print(f"Using device: {device}")
    model = load_model(model_path, device, **model_kw)
    y = load_audio(audio_path, SR)
    chunks, duration = split_into_chunks(y)
    print(f"Audio duration: {duration:.2f} seconds; {len(chunks)} chunks of {CHUNK_LENGTH}s")
    roll = transcribe_chunks(model, chunks, threshold, n_mels=model_kw.get("n_mels", N_MELS), device=device)
    notes = pianoroll_to_notes(roll, SR / HOP_LENGTH)
    if output_path is None:
        p = Path(audio_path)
        output_path = p.parent / f"{p.stem}_transcription.mid"
    write_midi(notes, str(output_path))
    print(f"MIDI file saved to: {output_path}")
    return output_path
