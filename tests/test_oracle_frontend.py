"""oracle/frontend_ref.py (restatement of the librosa calls at main.py:117-125).
PARITY UNPINNED: librosa is absent and the reference holds no fixture for this
boundary; these are cross-checks against two independent local implementations."""
import numpy as np
import pytest
import torch

from oracle import frontend_ref as FR


@pytest.fixture(scope="module")
def chunk():
    return FR.synth_audio(1, 480000, seed=99)[0]


def test_shapes_and_frame_count(chunk):
    m = FR.audio_to_mel(chunk)
    assert m.shape == (320, 938) and m.dtype == np.float32          # 1 + 480000 // 512
    assert FR.num_frames(480000) == 938 and FR.num_frames(479744) == 938 and FR.num_frames(1000) == 2
    assert m.max() - m.min() <= 80.0 + 1e-4


def test_filterbank_vs_transformers():
    from transformers.audio_utils import mel_filter_bank
    for n_mels in (229, 320):
        ours = FR.mel_filterbank(16000, 2048, n_mels)
        theirs = mel_filter_bank(1025, n_mels, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
        assert np.abs(ours - theirs).max() < 1e-7
        assert (ours.sum(axis=1) > 0).all()                         # no empty filters
        assert ((ours > 0).sum(axis=0) <= 2).all()                  # each bin feeds <= 2 adjacent filters
    fb = FR.mel_filterbank(16000, 2048, 320)
    assert int((fb > 0).sum()) == 2036                              # 0.62 % dense


def test_db_vs_transformers_spectrogram(chunk):
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    fb = mel_filter_bank(1025, 320, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney")
    ref = spectrogram(chunk.astype(np.float64), window_function(2048, "hann", periodic=True), 2048, 512,
                      fft_length=2048, power=2.0, center=True, pad_mode="constant", mel_filters=fb,
                      log_mel="dB", reference=1.0, min_value=1e-10, db_range=80.0)
    ours = FR.audio_to_mel(chunk)
    assert ref.shape == ours.shape
    assert np.abs(ref - ours).max() < 1e-3


def test_db_vs_torch_stft(chunk):
    y = torch.from_numpy(chunk)
    win = torch.hann_window(2048, periodic=True, dtype=torch.float32)
    S = torch.stft(y, 2048, 512, 2048, win, center=True, pad_mode="constant", return_complex=True).abs() ** 2
    mel = torch.from_numpy(FR.mel_filterbank()) @ S
    db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    db = torch.maximum(db, db.max() - 80.0).numpy()
    assert np.abs(db - FR.audio_to_mel(chunk)).max() < 2e-3


def test_edge_signals():
    z = FR.audio_to_mel(np.zeros(480000, np.float32))
    assert np.abs(z + 100.0).max() < 2e-5 and (z == z[0, 0]).all()  # silence -> -100 dB (float32 log10 of float32(1e-10))
    imp = np.zeros(480000, np.float32); imp[12345] = 1.0
    m = FR.audio_to_mel(imp)
    assert np.isfinite(m).all() and m.max() - m.min() == pytest.approx(80.0, abs=1e-4)
    half = FR.synth_audio(1, 240000, seed=3)[0]
    padded = np.concatenate([half, np.zeros(240000, np.float32)])   # main.py:93-95 zero-pads the waveform
    m = FR.audio_to_mel(padded)
    assert np.allclose(m[:, 600:], m.max() - 80.0)                  # floor clamped to max-80, not -100


def test_batch_uses_per_chunk_max():
    w = FR.synth_audio(2, 48000, seed=5)
    w[1] *= 1e-3
    mb = FR.audio_to_mel_batch(w)
    assert mb.shape == (2, 1, 320, 94)
    assert np.array_equal(mb[1, 0], FR.audio_to_mel(w[1]))
    assert mb[1].max() < mb[0].max() - 40
