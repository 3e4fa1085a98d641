"""CPU-side tests (no GPU): the C ABI loads and exports everything include/mt_hip.h declares, host
tables match the oracle, host logic (collate, cache format, sharding) matches the reference goldens."""
import json
import os
import pickle
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mta():
    import __graft_entry__ as ge
    ge.build()
    import music_transcription_amd as m
    return m


def test_header_and_library_agree(mta):
    hdr = open(os.path.join(ROOT, "include", "mt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    from music_transcription_amd import _lib
    for name in declared:
        assert hasattr(_lib.lib, name), f"{name} declared in mt_hip.h but not exported by libmt_hip.so"
    assert set(_lib.EXPORTS) <= declared, set(_lib.EXPORTS) - declared
    assert _lib.lib.mt_version() >= 100


def test_host_entry_points_without_gpu(mta):
    from music_transcription_amd import _lib
    lib = _lib.lib
    assert lib.mt_mel_num_frames(480000, 512) == 938 and lib.mt_mel_num_frames(479744, 512) == 938
    assert lib.mt_mel_num_frames(10, 0) == _lib_code("MT_EINVAL")
    assert lib.mt_mel_plan_bytes(320) > 3 * 8192 and lib.mt_mel_plan_bytes(0) == 0
    assert lib.mt_lstm_gx_bytes(32, 938, 512) == 938 * 2 * 64 * 4096
    assert lib.mt_lstm_hx_bytes(33, 10, 256) == 2 * 10 * 2 * 32 * 512
    fb = np.zeros((4, 1025), np.float32)
    assert lib.mt_mel_filterbank_host(fb.ctypes.data, -1, 4) == _lib_code("MT_EINVAL")
    assert "bad arguments" in _lib.last_error()
    with pytest.raises(_lib.MtError):
        _lib.check(-1, "x")


def _lib_code(name):
    hdr = open(os.path.join(ROOT, "include", "mt_hip.h")).read()
    return int(re.search(rf"#define {name}\s+(-?\d+)", hdr).group(1))


def test_filterbank_tables_match_oracle(mta):
    from oracle import frontend_ref as FR
    for sr, nm in ((16000, 320), (16000, 229), (22050, 128), (16000, 32)):
        assert np.array_equal(mta.mel_filterbank(sr, nm), FR.mel_filterbank(sr, 2048, nm))


def test_state_dict_manifest(mta, golden_dir):
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    for key, ent in man.items():
        parts = key.split(":")
        m = mta.TranscriptionModel(parts[0], n_mels=int(parts[1]), hidden_size=int(parts[2]), num_layers=int(parts[3]),
                                   device="cpu", use_attention="noattn" not in parts,
                                   use_onset_offset_heads="noheads" not in parts)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == ent["keys"], key
        assert sum(p.numel() for p in m.parameters()) == ent["n_params"]


def test_constructor_surface(mta):
    m = mta.TranscriptionModel()                       # reference defaults: cnn_rnn, 229, 256, 2, 0.3, cpu
    assert (m.model_type, m.device, m.use_onset_offset_heads) == ("cnn_rnn", "cpu", True)
    assert isinstance(m.criterion, torch.nn.BCEWithLogitsLoss)
    assert mta.TranscriptionModel("CNN+RNN").model_type == "cnn+rnn"
    assert isinstance(mta.TranscriptionModel("large", n_mels=64, hidden_size=32).model, mta.CNNRNNModelLarge)
    with pytest.raises(ValueError, match="Unknown model type"):
        mta.TranscriptionModel("nope")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 229, 8))                   # the product path never computes on the CPU
    with pytest.raises(RuntimeError):
        mta.compute_loss(torch.zeros(1, 88, 4), torch.zeros(1, 88, 4))


def test_same_seed_same_init_as_reference_order(mta):
    # parameter containers are registered in the reference's order, so one seed gives the same tensors
    torch.manual_seed(123); a = mta.TranscriptionModel("cnn_rnn", n_mels=32, hidden_size=16, num_layers=2)
    torch.manual_seed(123)
    c1 = torch.nn.Conv2d(1, 32, 3, padding=1)
    assert torch.equal(a.state_dict()["model.cnn.0.weight"], c1.weight)


def test_collate_matches_reference_golden(mta, golden_dir):
    g = np.load(os.path.join(golden_dir, "collate.npz"))
    gen = torch.Generator().manual_seed(int(g["seed"]))
    batch = [(torch.randn(1, 8, int(t), generator=gen), (torch.rand(88, int(t), generator=gen) < 0.1).float()) for t in g["Ts"]]
    mel, roll, lens = mta.collate_fn(batch)
    assert np.array_equal(mel.numpy(), g["mel"]) and np.array_equal(roll.numpy(), g["roll"])
    assert np.array_equal(lens.numpy(), g["lengths"]) and lens.dtype == torch.long


def test_cache_format_roundtrip(mta, tmp_path):
    cache = str(tmp_path / "cached_dataset_mels320")
    g = torch.Generator().manual_seed(0)
    chunks = []
    for i, T in enumerate((937, 937, 600)):
        mel = torch.randn(1, 320, T + 1, generator=g)              # mel one frame longer: trimmed to min_len
        roll = (torch.rand(88, T, generator=g) < 0.04).float()
        p = mta.write_cache_chunk(cache, "test", i, mel, roll)
        assert os.path.basename(p) == f"chunk_{i:06d}.pt"
        chunks.append({"file_idx": 0, "start_sample": i * 480000, "end_sample": (i + 1) * 480000,
                       "start_time": 30.0 * i, "end_time": 30.0 * (i + 1)})
    mta.write_cache_metadata(cache, "test", chunks, root_dir="maestro-v3.0.0", chunk_length=30.0, overlap=0.0)
    meta = pickle.load(open(os.path.join(cache, "test_metadata.pkl"), "rb"))
    from music_transcription_amd.data import METADATA_KEYS
    assert tuple(meta.keys()) == METADATA_KEYS and meta["num_chunks"] == 3 and meta["n_mels"] == 320
    rec = torch.load(os.path.join(cache, "test", "chunk_000000.pt"), weights_only=False)   # the reference reader's call
    assert set(rec) == {"mel", "roll"} and rec["mel"].shape == (1, 320, 937) and rec["roll"].shape == (88, 937)
    ds = mta.CachedMaestroDataset(cache, "test")
    assert len(ds) == 3 and ds[2][0].shape == (1, 320, 600) and ds[2][0].dtype == torch.float32
    with pytest.raises(FileNotFoundError):
        mta.CachedMaestroDataset(cache, "train")
    os.remove(os.path.join(cache, "test", "chunk_000001.pt"))
    with pytest.raises(FileNotFoundError):
        ds[1]


def test_cache_written_here_is_what_the_reference_reader_reads(mta, golden_dir):
    """tests/golden/cache_fixture/ was written by this package's writer and read by the REFERENCE's CachedMaestroDataset /
    HybridMaestroDataset (make_golden_cache.py, build container); cache_fixture.json records what they returned.  This
    package's reader must return the same, and the writer must still produce byte-identical tensors."""
    rec = json.load(open(os.path.join(golden_dir, "cache_fixture.json")))
    cache = os.path.join(golden_dir, "cache_fixture")
    for split, ent in rec["splits"].items():
        assert ent["hybrid_uses_cache"] and ent["hybrid_len"] == ent["len"]          # the reference's own acceptance test passed
        ds = mta.CachedMaestroDataset(cache_dir=cache, split=split)
        assert len(ds) == ent["len"] and sorted(ds.metadata.keys()) == ent["metadata_keys"]
        assert {k: v for k, v in ds.metadata.items() if k != "chunks"} == ent["metadata"]
        assert sorted(ds.metadata["chunks"][0].keys()) == ent["chunk_keys"]
        for i, it in enumerate(ent["items"]):
            mel, roll = ds[i]
            assert list(mel.shape) == it["mel_shape"] and list(roll.shape) == it["roll_shape"]
            assert str(mel.dtype) == it["mel_dtype"] and str(roll.dtype) == it["roll_dtype"]
            assert float(mel.double().sum()) == it["mel_sum"] and float(mel.double().abs().sum()) == it["mel_abs_sum"]
            assert float(roll.double().sum()) == it["roll_sum"]
            assert float((roll.double() * torch.arange(roll.numel()).reshape(roll.shape)).sum()) == it["roll_checksum"]
            assert mel.shape[-1] == roll.shape[-1]                                    # trimmed to min_len (data/dataset.py:159-161)


def test_shard_and_lpt():
    from music_transcription_amd.parallel import shard_range, lpt_assign
    for n in (0, 1, 7, 8, 177):
        for w in (1, 2, 3, 8):
            parts = [list(shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    rng = np.random.default_rng(0)
    dur = rng.uniform(60, 1200, size=177)
    a = lpt_assign(dur, 8)
    assert sorted(sum(a, [])) == list(range(177))
    loads = [dur[i].sum() for i in a]
    assert max(loads) - min(loads) < dur.max()          # LPT bound


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from importlib import import_module
import importlib.util as u
spec = u.spec_from_file_location("par", os.path.join(sys.argv[1], "music-transcription_amd", "parallel.py"))
par = u.module_from_spec(spec); spec.loader.exec_module(par)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
dur = [float((7 * i) % 13 + 1) for i in range(23)]
mine = par.lpt_assign(dur, 2)[rank]
vals = [dur[i] * 0.5 + i for i in mine]                # stands for a per-recording F1 computed on this rank
full = par.gather_values(mine, vals, len(dur))
assert full == [dur[i] * 0.5 + i for i in range(len(dur))], full
sl = par.shard_range(10, rank, 2)
assert list(sl) == ([0, 1, 2, 3, 4] if rank == 0 else [5, 6, 7, 8, 9])
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 500)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
        assert "ok" in o


def test_chunking_notes_and_midi_writer(mta, tmp_path):
    from music_transcription_amd import transcribe as tr
    y = np.arange(480000 + 1234, dtype=np.float32)
    chunks, dur = tr.split_into_chunks(y)
    assert chunks.shape == (2, 480000) and dur == pytest.approx((480000 + 1234) / 16000)
    assert chunks[1, 1233] == y[-1] and chunks[1, 1234:].max() == 0.0          # main.py:93-95 zero pad
    assert tr.split_into_chunks(np.zeros(480000, np.float32))[0].shape == (1, 480000)
    roll = np.zeros((88, 100), np.float32)
    roll[0, 0:5] = 1; roll[39, 10:11] = 1; roll[87, 95:100] = 1; roll[39, 20:30] = 1
    notes = tr.pianoroll_to_notes(roll, 31.25)
    assert notes == [(21, 0.0, 5 / 31.25), (60, 10 / 31.25, 11 / 31.25), (60, 20 / 31.25, 30 / 31.25), (108, 95 / 31.25, 100 / 31.25)]
    p = tmp_path / "o.mid"
    tr.write_midi(notes, str(p))
    raw = p.read_bytes()
    assert raw[:4] == b"MThd" and raw[8:14] == bytes([0, 1, 0, 2, 0, 220]) and raw.count(b"MTrk") == 2
    assert raw.count(bytes([0x90, 60, 100])) == 2 and raw.endswith(b"\xFF\x2F\x00")
    # the resampling plan handed to the GPU kernel reproduces scipy.signal.resample_poly with the SAME prototype filter
    # (evaluated here in numpy, natural and polyphase-major tap order)
    from scipy.signal import resample_poly
    rng = np.random.default_rng(0)
    for rate, n_in in ((44100, 5000), (48000, 3001), (22050, 777), (8000, 500)):
        x = rng.standard_normal(n_in)
        up, down, h, npr, n_out = tr.resample_plan(rate, 16000, n_in)
        ref = resample_poly(x, up, down, window=tr.resample_fir(up, down))
        assert n_out == len(ref)
        j = np.arange(n_out)[:, None]
        i = np.arange(n_in)[None, :]
        idx = (j + npr) * down - i * up
        ok = (idx >= 0) & (idx < len(h))
        y = (np.where(ok, h.astype(np.float64)[np.clip(idx, 0, len(h) - 1)], 0.0) * x[None, :]).sum(1)
        assert np.abs(y - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())
        tab = tr.polyphase_table(h, up)
        assert tab.shape[0] == up and all(tab[ph, k] == h[ph + k * up] for ph in (0, up - 1) for k in (0, 1, tab.shape[1] // 2) if ph + k * up < len(h))
    assert tr.resample_plan(16000, 16000, 123)[:2] == (1, 1)


def test_cli_errors_like_reference(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), str(tmp_path / "nope.wav"), str(tmp_path / "m.pth")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Audio file not found" in r.stdout
    (tmp_path / "a.wav").write_bytes(b"RIFF")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), str(tmp_path / "a.wav"), str(tmp_path / "m.pth")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Model file not found" in r.stdout


DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
import importlib.util as u
root = sys.argv[1]
sys.path.insert(0, root)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
import music_transcription_amd as mta
from music_transcription_amd.optim import allreduce_mean_
rank = dist.get_rank()
g = torch.arange(10, dtype=torch.float32) * (rank + 1)          # rank 0: x, rank 1: 2x  -> mean 1.5x
allreduce_mean_(g)
assert torch.allclose(g, torch.arange(10, dtype=torch.float32) * 1.5), g
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_two_rank_gloo_gradient_mean(tmp_path, mta):
    script = tmp_path / "dp.py"
    script.write_text(DP_WORKER)
    port = str(29000 + os.getpid() % 500)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]


def test_compressed_audio_goes_through_a_host_decoder(mta, tmp_path, monkeypatch):
    """data/dataset.py:68-70: a missing x.wav is looked up as x.mp3; anything that is not a WAV container is decoded on the host by
    whichever decoder is installed (none in this image: a stub stands in for `soundfile`), and without one the error says so."""
    import sys
    import types
    from music_transcription_amd import transcribe as tr
    from music_transcription_amd import preprocess as pp
    mp3 = tmp_path / "rec.mp3"
    mp3.write_bytes(b"ID3\x03" + bytes(64))
    assert tr.resolve_audio_path(str(tmp_path / "rec.wav")) == str(mp3)
    assert tr.resolve_audio_path(str(tmp_path / "other.wav")) == str(tmp_path / "other.wav")
    for name in ("soundfile", "audioread", "torchaudio"):
        monkeypatch.setitem(sys.modules, name, None)              # import -> ImportError
    with pytest.raises(ValueError, match="no host decoder"):
        tr.decode_compressed_host(str(mp3))
    pcm = (np.arange(44100 * 2, dtype=np.float32).reshape(-1, 2) % 7) / 7.0
    stub = types.ModuleType("soundfile")
    stub.read = lambda path, dtype="float32", always_2d=True: (pcm, 44100)
    monkeypatch.setitem(sys.modules, "soundfile", stub)
    rate, data = tr.decode_compressed_host(str(mp3))
    assert rate == 44100 and data.shape == (44100, 2) and data.dtype == np.float32
    assert abs(pp.wav_duration(str(tmp_path / "rec.wav")) - 1.0) < 1e-9          # through the .wav -> .mp3 fallback


def test_resampling_filter_meets_its_stated_spec():
    """The designed anti-alias filter (transcribe.resample_fir; librosa's soxr_hq is not available: SURVEY 8 f3): realised
    frequency response for the rates MAESTRO ships (44.1 and 48 kHz) and for an up-sampling case -- pass band flat to 0.01 dB
    up to 0.913 of the lower Nyquist frequency, >= 110 dB down from that Nyquist frequency on (design: 120 dB), unit DC gain."""
    from math import gcd
    from music_transcription_amd import transcribe as tr
    for rate_in, rate_out in ((44100, 16000), (48000, 16000), (8000, 16000)):
        g = gcd(rate_in, rate_out)
        up, down = rate_out // g, rate_in // g
        h = tr.resample_fir(up, down)
        assert len(h) % 2 == 1 and abs(h.sum() - 1.0) < 1e-9 and np.allclose(h, h[::-1])
        nfft = 1 << 22
        H = np.abs(np.fft.rfft(h, nfft))
        f = np.arange(len(H)) / nfft * rate_in * up                     # Hz on the up-sampled grid
        nyq = min(rate_in, rate_out) / 2.0
        pb = H[f <= tr.RESAMPLE_PASSBAND * nyq]
        sb = H[f >= tr.RESAMPLE_STOPBAND * nyq]
        assert np.abs(20 * np.log10(pb)).max() < 0.01, (rate_in, np.abs(20 * np.log10(pb)).max())
        assert 20 * np.log10(sb.max()) < -110.0, (rate_in, 20 * np.log10(sb.max()))
        taps_per_output = len(h) / up
        assert taps_per_output < 700                                    # cost bound: ~500 taps per output sample


def test_lpt_balances_a_maestro_sized_corpus_over_eight_ranks():
    """SURVEY 8e / BASELINE configs[4]: 177 recordings with gamma-distributed lengths (20 h in all) LPT-sharded over 8 ranks --
    every rank's load within 2 % of the mean for 20 different corpora, every recording on exactly one rank, the assignment
    identical whoever computes it (ties broken by index)."""
    from music_transcription_amd.parallel import lpt_assign
    from music_transcription_amd.corpus import synthetic_corpus
    for seed in range(20):
        dur = synthetic_corpus(177, 20.0, seed=seed)
        shards = lpt_assign(dur, 8)
        assert sorted(i for s_ in shards for i in s_) == list(range(177))
        loads = np.array([sum(dur[i] for i in s_) for s_ in shards])
        assert loads.max() <= 1.02 * loads.mean() and loads.min() >= 0.98 * loads.mean(), (seed, loads / loads.mean())
        assert shards == lpt_assign(list(dur), 8)
    # degenerate shapes: fewer recordings than ranks, one rank
    assert [len(s_) for s_ in lpt_assign([3.0, 1.0], 4)] == [1, 1, 0, 0] and lpt_assign([1.0, 2.0, 3.0], 1) == [[2, 1, 0]]


def test_side_stream_tuner_keeps_the_fastest_candidate(mta, monkeypatch):
    """train_step_large.SideStreamTuner (the choice of the training step's side streams by measurement, made inside a loop whose steps end in
    a host synchronisation): candidate c serves steps [c (steps + 1), (c + 1)(steps + 1)), the first of them untimed; after the last candidate
    the fastest one is restored into BOTH training steps' stream tables and the tuner goes quiet.  Streams and clock are faked: host logic only."""
    from music_transcription_amd import train_step, train_step_large as TL
    import time as _time
    made = []

    def fake_pair(dev):
        key = train_step.device_key(dev)
        if key not in TL._SIDE2:
            made.append(("pair", len(made)))
            TL._SIDE2[key] = made[-1]
            train_step._SIDE[key] = ("single", len(made) - 1)
        return TL._SIDE2[key]

    monkeypatch.setattr(TL, "_side_streams", fake_pair)
    monkeypatch.setenv("MT_TRAIN_STREAM_AUTOTUNE", "1")
    clock = [0.0]
    monkeypatch.setattr(_time, "perf_counter", lambda: clock[0])
    TL._SIDE2.clear(); train_step._SIDE.clear()
    tuner = TL.SideStreamTuner("cpu", candidates=3, steps=2)
    cost = {0: 5.0, 1: 3.0, 2: 4.0}                    # seconds per step under candidate 0, 1, 2
    for i in range(12):
        tuner.step_begin()
        c = fake_pair(torch.device("cpu"))[1] if not tuner.done else None      # (the step itself creates the streams)
        clock[0] += 100.0 if (not tuner.done and i % 3 == 0) else (cost[c] if c is not None else 1.0)   # first step of a candidate: warm-up, any length
        tuner.step_end()
    assert tuner.done and [round(t, 6) for t, _ in tuner.seen] == [5.0, 3.0, 4.0]
    assert TL._SIDE2["cpu:0"] == ("pair", 1) and train_step._SIDE["cpu:0"] == ("single", 1)
    n = len(made)
    tuner.step_begin(); tuner.step_end()               # quiet afterwards: no further candidates
    assert len(made) == n == 3
    TL._SIDE2.clear(); train_step._SIDE.clear()
    off = TL.SideStreamTuner("cpu")
    monkeypatch.setenv("MT_TRAIN_STREAM_AUTOTUNE", "0")
    assert TL.SideStreamTuner("cpu").done and not off.done


def test_bench_compact_line_survives_a_bounded_tail():
    """bench.py's LAST stdout line is the driver's record: round 3's full record (21.7 KB, kept as profiles/r03_bench_default_line.json)
    must compact to <= 4 KB with the contract's fields, and must still parse when only the last 8 KB of stdout + stderr are kept."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default_line.json")))
    full["configs1_literal_b32"] = {"workload": "x", "value": 7568.05, "unit": "chunks/s", "ms_per_step": 4.228, "streams": 3, "one_in_flight": 4990.0}
    full["configs4_corpus_from_pcm"] = dict(full["configs4_corpus"], value=1500.0, wall_s=1.66)
    line = bench.compact_line(full)
    assert len(line) <= bench.LINE_LIMIT == 4096 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["value"] == full["value"] and d["metric"] == full["metric"] and d["dtype"] == "f16" and set(d["config"]) >= {"workload", "batch_per_gpu"}
    rf = d["roofline"]
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "all"} <= set(rf) and rf["bound"] in ("hbm", "mfma")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and len(rf["all"]) >= 9
    assert {"value", "unit", "cores", "kind"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert d["configs1_literal_b32"]["value"] == 7568.05 and d["configs1_literal_b32"]["streams"] == 3
    assert set(d["sections"]) == {"configs2_large_b16", "configs3_train_b16", "train_large_b16", "configs4_corpus", "configs4_corpus_from_pcm"}
    for o in d["sections"].values():
        assert o["value"] > 0 and ("ms_per_step" in o or "wall_s" in o)
    # the way the record's tail is cut: last 8 KB of everything the process wrote, the line is the last one that starts with '{'
    noise = "".join(f"[bench] section {i}: 1.0 s\n" for i in range(400))
    tail = (noise + line + "\n")[-8192:]
    last = [l for l in tail.splitlines() if l.startswith("{")][-1]
    assert json.loads(last) == d
    # a pathological record (huge stage maps) still comes out under the limit, contract fields intact
    full["roofline"]["all_large"] = {f"kernel_number_{i}": 0.123456 for i in range(400)}
    line2 = bench.compact_line(full)
    d2 = json.loads(line2)
    assert len(line2) <= 4096 and d2["value"] == full["value"] and d2["roofline"]["frac"] == rf["frac"] and "cpu_baseline" in d2


def test_side_stream_tables_use_one_key_per_device(mta):
    """a tuner built with the caller's device='cuda' must forget / restore the streams the step created under
    x.device = cuda:0 (the tables were keyed by str(device): 'cuda' != 'cuda:0' and the tuner measured one pair four times)."""
    from music_transcription_amd import train_step, train_step_large as L
    assert train_step.device_key("cuda") == train_step.device_key(torch.device("cuda")) == train_step.device_key("cuda:0") == "cuda:0"
    assert train_step.device_key(torch.device("cuda", 3)) == "cuda:3"
    tuner = L.SideStreamTuner("cuda", candidates=2, steps=1)
    assert tuner.key == "cuda:0"
    a, b = object(), object()
    L._SIDE2["cuda:0"], train_step._SIDE["cuda:0"] = (a, a), b          # what a step on cuda:0 creates
    try:
        saved = L._current_side_streams(tuner.key)
        assert saved == ((a, a), b)
        L._forget_side_streams(tuner.key)
        assert "cuda:0" not in L._SIDE2 and "cuda:0" not in train_step._SIDE
        L._restore_side_streams(tuner.key, saved)
        assert L._SIDE2["cuda:0"] == (a, a) and train_step._SIDE["cuda:0"] is b
    finally:
        L._SIDE2.pop("cuda:0", None); train_step._SIDE.pop("cuda:0", None)


def test_params_without_grad_is_an_intersection_over_backward_passes(mta):
    """two backward passes before one optimizer step (a frame-only pass, then a return_all_heads pass): the onset /
    offset heads received a gradient in the second, so the fused step must NOT skip them (torch.optim.Adam would update them)."""
    from music_transcription_amd.optim import FusedAdamClip, note_params_without_grad

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body, self.onset_head = torch.nn.Linear(3, 2), torch.nn.Linear(2, 2)
    net = Net()
    opt = FusedAdamClip.__new__(FusedAdamClip)                         # (host logic only: no device buffers)
    views, off = [], 0
    for _, p in net.named_parameters():
        views.append((p, off, p.numel())); off += p.numel()
    opt._views, opt.net, opt.skip_untouched = views, net, True
    net._params_without_grad = None
    assert opt._keep_ranges() is None                                  # no backward yet: everything takes part
    heads = {"onset_head.weight", "onset_head.bias"}
    note_params_without_grad(net, heads)                               # frame-only backward
    assert opt._keep_ranges() == [[0, 8]]                              # body only (6 + 2 values)
    note_params_without_grad(net, set())                               # second backward reaches the heads
    assert net._params_without_grad == set() and opt._keep_ranges() is None
    net._params_without_grad = None
    note_params_without_grad(net, heads); note_params_without_grad(net, heads)     # two frame-only passes: still skipped
    assert opt._keep_ranges() == [[0, 8]]


def test_direct_gradient_targets_follow_torch_accumulation_semantics(mta):
    """optim.FusedAdamClip.make_grad_target: a backward pass may write a parameter's gradient straight into its view of the flat gradient
    buffer ONCE per zero_grad(); a second backward pass before step() gets None (its gradient goes through autograd and is added, as torch
    accumulates), and so does a parameter whose .grad a caller detached (zero_grad(set_to_none=True)) or whose shape does not match."""
    from music_transcription_amd.optim import FusedAdamClip
    a, b = torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))
    g = torch.zeros(17)
    a.grad, b.grad = g[0:12].view(3, 4), g[12:17].view(5)
    opt = FusedAdamClip.__new__(FusedAdamClip)
    opt.g, opt._views, opt._touched, opt.net = g, [(a, 0, 12), (b, 12, 5)], set(), None
    target = opt.make_grad_target({"a": a, "b": b})
    t = target("a", (3, 4))
    assert t is not None and t.data_ptr() == g.data_ptr() and t.shape == (3, 4)
    t.fill_(2.0)
    assert float(a.grad.sum()) == 24.0                              # the view IS the parameter's gradient
    assert target("a", (3, 4)) is None                               # second backward before step(): accumulate through autograd
    assert target("b", (4,)) is None and target("c", (1,)) is None   # wrong size / unknown parameter
    b.grad = None                                                    # a caller's zero_grad(set_to_none=True)
    assert target("b", (5,)) is None
    b.grad = g[12:17].view(5)
    opt.g.zero_(); opt._touched = set()                              # = zero_grad()
    assert target("a", (12,)) is not None and target("b", (5,)) is not None


def test_workspace_query_and_allreduce_argument_checks(mta):
    """SURVEY 8(b)'s mt_workspace_bytes(kind, dims...) forwards to the per-buffer queries; mt_allreduce refuses bad arguments before it
    looks for RCCL (no GPU here)."""
    from music_transcription_amd import _lib
    lib = _lib.lib
    assert lib.mt_workspace_bytes(1, 32, 938, 512) == lib.mt_lstm_gx_bytes(32, 938, 512)
    assert lib.mt_workspace_bytes(2, 33, 10, 256) == lib.mt_lstm_hx_bytes(33, 10, 256)
    assert lib.mt_workspace_bytes(5, 16, 937, 512) * 2 == lib.mt_workspace_bytes(5, 17, 937, 512)       # 16-column hand-off slices for B <= 16
    assert lib.mt_workspace_bytes(7, 320, 0, 0) == lib.mt_mel_plan_bytes(320) and lib.mt_workspace_bytes(8, 0, 0, 0) == lib.mt_adam_workspace_bytes()
    assert lib.mt_workspace_bytes(99, 1, 1, 1) == 0 and "unknown kind" in _lib.last_error()
    assert lib.mt_allreduce(None, 4, 0, None, None) == _lib_code("MT_EINVAL")


def test_pack_job_record_matches_the_c_struct(tmp_path):
    """pack_plan.JOB_DTYPE is the byte layout of mt_pack_job (include/mt_hip.h): size and every field offset, read from the C compiler."""
    import subprocess
    from music_transcription_amd.pack_plan import JOB_DTYPE, one, two
    fields = [n for n in JOB_DTYPE.names]
    src = tmp_path / "o.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mt_hip.h"\nint main(void) { printf("%zu", sizeof(mt_pack_job));\n'
                   + "".join(f'printf(" %zu", offsetof(mt_pack_job, {n}));\n' for n in fields) + "return 0; }\n")
    exe = tmp_path / "o"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == JOB_DTYPE.itemsize
    assert out[1:] == [JOB_DTYPE.fields[n][1] for n in fields]
    assert one(5, 3) == (1, 1 << 30, 5, 0, 3) and two(4, 32, 24, 7, 1) == (4, 32, 24, 7, 1)
