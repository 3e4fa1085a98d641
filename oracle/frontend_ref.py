"""CPU ORACLE (test infrastructure, NOT product code) -- log-mel frontend.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  The shipped path is the HIP kernel
`mt_mel_db_f32` (music-transcription_amd/csrc/mel.hip).

PARITY UNPINNED.  The arithmetic restated here lives in a third-party dependency
of the reference that is absent from /root/reference and from this image:
`librosa` (requirements.txt:9, `librosa>=0.10.0` -- a floor, no lockfile).  The
reference holds no test or golden vector for this boundary, so this restatement
follows librosa >= 0.10's published algorithm as the reference calls it:

    main.py:117-125          melspectrogram(y, sr, n_mels, hop_length) -> power_to_db
    data/dataset.py:155-156  same two calls (cache writer)
    data/dataset.py:195-196  same two calls (full-file mode)

with librosa defaults n_fft=2048, win_length=2048, window='hann' (periodic),
center=True, pad_mode='constant', power=2.0, fmin=0, fmax=sr/2, htk=False,
norm='slaney'; power_to_db(ref=1.0, amin=1e-10, top_db=80.0) with the max taken
over the whole array passed in (one chunk).  It is cross-checked in
tests/test_oracle_frontend.py against two independent local implementations
(transformers.audio_utils and torch.stft).
"""
from __future__ import annotations

import numpy as np

N_FFT = 2048
AMIN = 1e-10
TOP_DB = 80.0


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True), float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_t = min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_t, mels)


def mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr: int = 16000, n_fft: int = N_FFT, n_mels: int = 320,
                   fmin: float = 0.0, fmax: float | None = None) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm='slaney', dtype=float32) -> (n_mels, 1+n_fft//2)."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins, dtype=np.float64)
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2)
    mel_f = mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def num_frames(n_samples: int, hop_length: int = 512) -> int:
    return 1 + n_samples // hop_length


def stft_power(y: np.ndarray, hop_length: int = 512, n_fft: int = N_FFT) -> np.ndarray:
    """|STFT|^2 as librosa computes it: float64 window*frame product, rfft, result
    rounded to complex64, abs in float32, squared in float32.  -> (1+n_fft//2, T)."""
    y = np.asarray(y, dtype=np.float32)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    T = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    frames = yp[idx]                                   # (n_fft, T) float32
    win = hann_periodic(n_fft)[:, None]                # float64
    spec = np.fft.rfft(win * frames, axis=0).astype(np.complex64)
    mag = np.abs(spec)                                 # float32
    return mag ** 2.0


def melspectrogram(y: np.ndarray, sr: int = 16000, n_mels: int = 320,
                   hop_length: int = 512) -> np.ndarray:
    S = stft_power(y, hop_length)
    fb = mel_filterbank(sr, N_FFT, n_mels)
    return np.einsum("ft,mf->mt", S, fb, optimize=True).astype(np.float32)


def power_to_db(S: np.ndarray, amin: float = AMIN, top_db: float = TOP_DB) -> np.ndarray:
    S = np.asarray(S)
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec -= 10.0 * np.log10(np.maximum(amin, 1.0))
    return np.maximum(log_spec, log_spec.max() - top_db)


def audio_to_mel(audio_chunk: np.ndarray, sr: int = 16000, n_mels: int = 320,
                 hop_length: int = 512) -> np.ndarray:
    """main.py:103-130 without the torch wrapping: (N,) -> (n_mels, T) float32 dB."""
    mel = melspectrogram(np.asarray(audio_chunk, dtype=np.float32), sr, n_mels, hop_length)
    return power_to_db(mel).astype(np.float32)


def audio_to_mel_batch(wave: np.ndarray, sr: int = 16000, n_mels: int = 320,
                       hop_length: int = 512) -> np.ndarray:
    """(B, N) -> (B, 1, n_mels, T); the dB clamp max is per chunk (Appendix A quirk 2)."""
    return np.stack([audio_to_mel(w, sr, n_mels, hop_length)[None] for w in wave])


def synth_audio(batch: int, n_samples: int = 480000, seed: int = 1234, sr: int = 16000) -> np.ndarray:
    """SURVEY 8(d) synthetic input: 0.1*N(0,1) noise + 1-6 decaying sinusoids at piano
    fundamentals, clipped to [-1, 1].  Shared by tests and bench.py."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.empty((batch, n_samples), dtype=np.float32)
    for b in range(batch):
        y = 0.1 * rng.standard_normal(n_samples)
        for _ in range(int(rng.integers(1, 7))):
            key = int(rng.integers(0, 88))
            f0 = 27.5 * 2.0 ** (key / 12.0)
            onset = float(rng.uniform(0.0, 0.8)) * n_samples / sr
            amp = float(rng.uniform(0.1, 0.6))
            env = np.where(t >= onset, np.exp(-(t - onset) * float(rng.uniform(0.3, 3.0))), 0.0)
            y += amp * env * np.sin(2 * np.pi * f0 * (t - onset))
        out[b] = np.clip(y, -1.0, 1.0).astype(np.float32)
    return out
