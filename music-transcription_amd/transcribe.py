"""Audio file -> MIDI with the HIP hot path: the pipeline of the reference's main.py (:60-100 chunking,
:103-161 mel + predict, :164-226 roll -> notes, :229-287 driver), restructured for the GPU:

  * all 30 s chunks of a recording go through ONE batched mel launch and ONE batched forward
    (the reference loops batch-1 on the host, main.py:258-266); chunks are exactly 480000 samples,
    so no time padding arises and batching is parity-safe;
  * the thresholded roll comes back once, notes are run-length decoded per pitch on the host
    (as the reference does) and written by a dependency-free Standard MIDI File writer.

Audio decoding: librosa/soundfile are not available in this image, so only WAV (PCM 16/24/32-bit or
float) is read; channel mean, PCM scaling and polyphase resampling run on the GPU (csrc/resample.hip,
scipy.signal.resample_poly's filter) -- NOT sample-exact with librosa's soxr_hq.
"""
from __future__ import annotations

import os
import struct
from pathlib import Path
from typing import Optional, List, Tuple

import numpy as np
import torch

from .frontend import get_frontend, num_frames
from .model import TranscriptionModel
from . import ops, _lib
from ._lib import lib, check, ptr

MODEL_TYPE, N_MELS, HIDDEN_SIZE, NUM_LAYERS, DROPOUT = "cnn_rnn_large", 320, 512, 3, 0.2     # main.py:16-20
SR, HOP_LENGTH, CHUNK_LENGTH, THRESHOLD = 16000, 512, 30.0, 0.5                                # main.py:21-24


# Anti-alias / anti-image filter of the GPU resampler.  librosa.load(sr=16000) resamples with soxr_hq (main.py:76,
# data/dataset.py:124-130); soxr is not available here, so the filter is DESIGNED to soxr's published HQ figures instead of
# copied from it: pass band up to 0.913 of the lower Nyquist frequency, stop band from that Nyquist frequency on (no aliasing
# into the band at all), >= 120 dB of rejection (HQ is specified as 20-bit precision).  A Kaiser-windowed sinc on the
# up-sampled grid meets that by construction (scipy.signal.kaiserord); tests/test_host_cpu.py checks the realised frequency
# response.  mel's fmax = 8000 Hz = the Nyquist frequency of the 16 kHz signal, so the roll-off between 7.3 and 8 kHz does land
# in the top mel bins -- as soxr's does; tests/test_gpu_parity.py bounds what a 4x longer filter would change there.
RESAMPLE_PASSBAND = 0.913        # of the lower of the two Nyquist frequencies
RESAMPLE_STOPBAND = 1.0
RESAMPLE_REJECTION_DB = 120.0


def resample_fir(up: int, down: int, passband: float = RESAMPLE_PASSBAND, stopband: float = RESAMPLE_STOPBAND,
                 rejection_db: float = RESAMPLE_REJECTION_DB) -> np.ndarray:
    """Prototype low-pass on the up-sampled grid (rate_in * up), odd length, unit DC gain (scipy.signal.resample_poly scales it
    by `up`): edges relative to the grid's Nyquist frequency are passband / max(up, down) and stopband / max(up, down)."""
    from scipy.signal import firwin, kaiserord
    m = float(max(up, down))
    width = (stopband - passband) / m
    n, beta = kaiserord(rejection_db, width)
    n |= 1
    return firwin(n, 0.5 * (passband + stopband) / m, window=("kaiser", beta))


def resample_plan(rate_in: int, rate_out: int, n_in: int, fir: Optional[np.ndarray] = None):
    """Polyphase plan with scipy.signal.resample_poly's conventions for a given prototype filter (default: resample_fir): the
    filter scaled by `up` and pre-padded so that output sample j sits at input time j*down/up.  Returns
    (up, down, h float32, n_pre_remove, n_out); y[j] = sum_i x[i] h[(j + n_pre_remove) * down - i * up], which is
    scipy.signal.resample_poly(x, up, down, window=fir) -- the structural oracle of the tests.  up == down == 1 -> a one-tap
    identity plan."""
    from math import gcd
    g = gcd(int(rate_in), int(rate_out))
    up, down = int(rate_out) // g, int(rate_in) // g
    if up == down == 1:
        return 1, 1, np.ones(1, np.float32), 0, n_in
    n_out = (n_in * up + down - 1) // down
    proto = resample_fir(up, down) if fir is None else np.asarray(fir, dtype=np.float64)
    half_len = (len(proto) - 1) // 2
    h = proto * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down

    def out_len(len_h):
        nt = (n_in + (len_h + (-len_h % up)) // up - 1) * up
        return nt // down + (1 if nt % down else 0)
    n_post_pad = 0
    while out_len(len(h) + n_pre_pad + n_post_pad) < n_out + n_pre_remove:
        n_post_pad += 1
    h = np.concatenate([np.zeros(n_pre_pad), h, np.zeros(n_post_pad)]).astype(np.float32)
    return up, down, h, n_pre_remove, n_out


def polyphase_table(h: np.ndarray, up: int) -> np.ndarray:
    """h[(phase) + k*up] -> table[phase][k] (zero-padded): the taps one output sample needs are then contiguous."""
    L = (len(h) + up - 1) // up
    t = np.zeros(L * up, dtype=np.float32)
    t[:len(h)] = h
    return np.ascontiguousarray(t.reshape(L, up).T)


_PLAN_FILTERS = {}


def resolve_audio_path(path: str) -> str:
    """The reference's fallback for corpora distributed as mp3 (data/dataset.py:68-70,:119-121,:175-177): a missing `x.wav` is
    looked up as `x.mp3`."""
    if not os.path.exists(path) and path.endswith(".wav") and os.path.exists(path[:-4] + ".mp3"):
        return path[:-4] + ".mp3"
    return path


def decode_compressed_host(path: str):
    """Compressed audio (mp3 / flac / ogg ...) -> (sample rate, (frames, channels) float32 / int16 PCM) on the host, through
    whichever decoder is installed -- soundfile, audioread (what librosa.load falls back to, main.py:76), torchaudio; none of them
    ships in this image, where the call raises.  The PCM then takes the same path as a WAV file: one H2D copy, channel mean and
    polyphase resampling on the GPU (csrc/resample.hip)."""
    errors = []
    try:
        import soundfile
        data, rate = soundfile.read(path, dtype="float32", always_2d=True)
        return int(rate), np.ascontiguousarray(data, dtype=np.float32)
    except ImportError as e:
        errors.append(f"soundfile: {e}")
    except Exception as e:                                  # (libsndfile builds without mp3 support)
        errors.append(f"soundfile: {e}")
    try:
        import audioread
        with audioread.audio_open(path) as f:
            rate, ch = int(f.samplerate), int(f.channels)
            pcm = np.frombuffer(b"".join(f), dtype="<i2")
        return rate, np.ascontiguousarray(pcm.reshape(-1, ch))
    except ImportError as e:
        errors.append(f"audioread: {e}")
    except Exception as e:
        errors.append(f"audioread: {e}")
    try:
        import torchaudio
        wav, rate = torchaudio.load(path)                   # (channels, frames) float32
        return int(rate), np.ascontiguousarray(wav.t().numpy(), dtype=np.float32)
    except ImportError as e:
        errors.append(f"torchaudio: {e}")
    except Exception as e:
        errors.append(f"torchaudio: {e}")
    raise ValueError(f"{path}: not a WAV file and no host decoder could read it ({'; '.join(errors)})")


def load_audio_device(path: str, sr: int = SR, device="cuda") -> torch.Tensor:
    """Audio file -> mono float32 at `sr` ON THE DEVICE (librosa.load(path, sr=sr, mono=True) of main.py:76): the PCM frames
    go to the GPU as they are (memory-mapped read, one H2D copy) and csrc/resample.hip does the channel mean, the PCM
    scaling and the polyphase resampling.  WAV (PCM 8/16/24/32-bit or float32) is read directly; anything else goes through
    decode_compressed_host; a missing `.wav` falls back to the `.mp3` beside it as the reference does.  Not soxr-exact."""
    from scipy.io import wavfile
    from . import _lib
    path = resolve_audio_path(path)
    with open(path, "rb") as fh:
        is_wav = fh.read(4) in (b"RIFF", b"RIFX", b"RF64")
    if is_wav:
        try:
            rate, data = wavfile.read(path, mmap=True)
        except ValueError as e:
            raise ValueError(f"{path}: unreadable WAV file ({e})")
    else:
        rate, data = decode_compressed_host(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        fmt = 0
    elif data.dtype == np.int32:
        fmt = 1
    elif data.dtype == np.float32:
        fmt = 2
    elif data.dtype == np.uint8:
        data, fmt = ((data.astype(np.float32) - 128.0) / 128.0), 2
    else:
        data, fmt = data.astype(np.float32), 2
    n_in, ch = data.shape
    if n_in == 0:
        return torch.zeros(0, dtype=torch.float32, device=device)
    dev = torch.device(device)
    src = torch.from_numpy(np.ascontiguousarray(data)).to(dev, non_blocking=True)
    return resample_pcm_device(src, int(rate), sr)


_PCM_FMT = {torch.int16: 0, torch.int32: 1, torch.float32: 2}


def resample_pcm_device(src: torch.Tensor, rate: int, sr: int = SR) -> torch.Tensor:
    """(frames, channels) interleaved PCM ON THE DEVICE (int16 / int32 / float32, as a WAV file holds it) at `rate` -> mono
    float32 at `sr`: channel mean, PCM scaling and polyphase resampling in one kernel (mt_resample_polyphase, csrc/resample.hip).
    The filter table of a (rate, sr) pair is built once per device (resample_fir)."""
    if not src.is_cuda or src.dim() != 2 or src.dtype not in _PCM_FMT:
        raise ValueError("resample_pcm_device expects a (frames, channels) int16 / int32 / float32 CUDA tensor")
    src = src.contiguous()
    n_in, ch = int(src.shape[0]), int(src.shape[1])
    dev = src.device
    if n_in == 0:
        return torch.zeros(0, dtype=torch.float32, device=dev)
    pkey = (int(rate), int(sr), str(dev))
    if pkey not in _PLAN_FILTERS:
        # (the plan's trailing zero taps depend on n_in, the filter, `up`, `down` and n_pre_remove do not: taps beyond the table are zeros)
        up, down, h, n_pre, _ = resample_plan(int(rate), sr, n_in)
        _PLAN_FILTERS[pkey] = (up, down, torch.from_numpy(polyphase_table(h, up)).to(dev), n_pre)
    up, down, hd, n_pre_remove = _PLAN_FILTERS[pkey]
    n_out = n_in if up == down == 1 else (n_in * up + down - 1) // down
    out = torch.empty(n_out, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib.mt_resample_polyphase(_lib.ptr(src), n_in, ch, _PCM_FMT[src.dtype], _lib.ptr(hd), hd.shape[1], up, down, n_pre_remove,
                                                  _lib.ptr(out), n_out, _lib.stream_ptr()), "mt_resample_polyphase")
    return out


def load_audio(path: str, sr: int = SR) -> np.ndarray:
    """Mono float32 at `sr` as a numpy array (decoded and resampled on the GPU, see load_audio_device)."""
    return load_audio_device(path, sr, "cuda").cpu().numpy()


def split_into_chunks_device(y: torch.Tensor, chunk_length: float = CHUNK_LENGTH, sr: int = SR):
    """main.py:60-100 on the device: ceil(len / chunk) chunks, the last one zero-padded.  -> ((n, chunk_samples), seconds)."""
    chunk = int(chunk_length * sr)
    n = max(1, -(-y.numel() // chunk))
    out = torch.zeros(n * chunk, dtype=torch.float32, device=y.device)
    out[:y.numel()] = y[: n * chunk]
    return out.view(n, chunk), y.numel() / sr


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


def notes_from_logits_device(logits: torch.Tensor, threshold: float = THRESHOLD, fs: float = SR / HOP_LENGTH, min_midi: int = 21,
                             is_roll: bool = False) -> List[Tuple[int, float, float]]:
    """(n_chunks, 88, T) logits (or {0,1} roll values with is_roll=True) ON THE DEVICE -> notes, without the roll ever
    leaving the GPU: threshold + combine_piano_rolls + the run-length of pianoroll_to_midi (main.py:153-226) in
    mt_roll_to_notes; only 88 counts and two ints per note are copied to the host."""
    if not logits.is_cuda:
        raise RuntimeError("notes_from_logits_device expects a CUDA tensor")
    x = logits.detach().contiguous().float()
    NB, P, T = x.shape
    dev = x.device
    counts = torch.empty(P, dtype=torch.int32, device=dev)
    cap = max(1024, NB * 64)
    while True:
        starts, ends = torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(lib.mt_roll_to_notes(ptr(x), 1 if is_roll else 0, float(threshold), NB, P, T, ptr(counts), ptr(starts), ptr(ends), cap,
                                       _lib.stream_ptr()), "mt_roll_to_notes")
        c = counts.cpu().numpy()
        total = int(c.sum())
        if total <= cap:
            break
        cap = total
    s, e = starts[:total].cpu().numpy(), ends[:total].cpu().numpy()
    pitches = np.repeat(np.arange(P) + min_midi, c)
    return [(int(pp), float(a) / fs, float(b) / fs) for pp, a, b in zip(pitches, s, e) if b > a]


@torch.no_grad()
def transcribe_chunks_to_notes(model: "TranscriptionModel", chunks, threshold: float = THRESHOLD, batch: int = 128, n_mels: int = N_MELS,
                               device: str = "cuda") -> List[Tuple[int, float, float]]:
    """(n, 480000) waveform chunks -> notes; mel, forward, threshold, concatenation and run-length all on the GPU."""
    fe = get_frontend(SR, n_mels, HOP_LENGTH, device)
    if not torch.is_tensor(chunks):
        chunks = torch.from_numpy(np.ascontiguousarray(chunks, dtype=np.float32))
    chunks = chunks.to(device)
    net = model.model
    outs = []
    for i in range(0, len(chunks), batch):
        mel, cmax = fe(chunks[i:i + batch].contiguous(), clamp=False)
        outs.append(net(mel, chunk_max_power=cmax))
    notes = notes_from_logits_device(torch.cat(outs), threshold, SR / HOP_LENGTH)
    net.raise_on_handoff_timeout(sync=False)               # (the copies above synchronised with every forward)
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
def transcribe_chunks(model: TranscriptionModel, chunks, threshold: float = THRESHOLD, batch: int = 128,
                      n_mels: int = N_MELS, device: str = "cuda") -> np.ndarray:
    """(n, 480000) waveform chunks (device tensor, or numpy) -> (88, n * T) {0,1} roll; mel + forward + threshold all on the GPU."""
    fe = get_frontend(SR, n_mels, HOP_LENGTH, device)
    if not torch.is_tensor(chunks):
        chunks = torch.from_numpy(np.ascontiguousarray(chunks, dtype=np.float32))
    chunks = chunks.to(device)
    rolls = []
    for i in range(0, len(chunks), batch):
        wave = chunks[i:i + batch].contiguous()
        mel, cmax = fe(wave, clamp=False)
        net = model.model
        logits = net(mel, chunk_max_power=cmax)
        roll = ops.predict_from_logits(logits, threshold)                      # (b, 88, T)
        rolls.append(roll.permute(1, 0, 2).reshape(88, -1).cpu().numpy())      # combine_piano_rolls: concat along time
        net.raise_on_handoff_timeout(sync=False)       # (the copy above synchronised with this batch's forward)
    return np.concatenate(rolls, axis=1)


def transcribe_audio(audio_path: str, model_path: str, output_path=None, device=None, threshold: float = THRESHOLD, **model_kw):
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    if device != "cuda":
        raise RuntimeError("music_transcription_amd runs on the GPU only (-d cuda)")
    print(f"Using device: {device}")
    model = load_model(model_path, device, **model_kw)
    y = load_audio_device(audio_path, SR, device)       # decode + resample on the GPU; the waveform never visits the host
    chunks, duration = split_into_chunks_device(y)
    print(f"Audio duration: {duration:.2f} seconds; {len(chunks)} chunks of {CHUNK_LENGTH}s")
    notes = transcribe_chunks_to_notes(model, chunks, threshold, n_mels=model_kw.get("n_mels", N_MELS), device=device)
    if output_path is None:
        p = Path(audio_path)
        output_path = p.parent / f"{p.stem}_transcription.mid"
    write_midi(notes, str(output_path))
    print(f"MIDI file saved to: {output_path}")
    return output_path
