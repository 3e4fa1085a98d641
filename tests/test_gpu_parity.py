"""GPU parity tests: the HIP path (through the C ABI of libmt_hip.so) against the CPU oracle
on the same seeded inputs, and against the committed reference-generated goldens."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import frontend_ref as FR
from oracle import model_ref as R


@pytest.fixture(scope="module")
def mta():
    import music_transcription_amd as m
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return m


def _mel_in(B, nm, T, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 1, nm, T, generator=g) * 60.0 - 70.0 + 10.0 * torch.randn(B, 1, nm, 1, generator=g))


# ------------------------------------------------------------------ frontend
MEL_MAX_TOL_DB = 5e-2    # worst bin (fp32 FFT cancellation noise on bins ~80 dB below the chunk peak)
MEL_MEAN_TOL_DB = 2e-4


@pytest.mark.parametrize("B,N,n_mels,hop", [(2, 480000, 320, 512), (3, 48000, 229, 512), (1, 16000, 64, 256),
                                            (2, 1000, 32, 512), (1, 479744, 320, 512), (1, 123457, 80, 512)])
def test_mel_matches_oracle(mta, B, N, n_mels, hop):
    wave = FR.synth_audio(B, N, seed=N % 97)
    fe = mta.MelFrontend(16000, n_mels, hop, "cuda")
    mel, cmax = fe(torch.from_numpy(wave).cuda(), clamp=True)
    ref = FR.audio_to_mel_batch(wave, 16000, n_mels, hop)
    got = mel.cpu().numpy()
    assert got.shape == ref.shape
    d = np.abs(got - ref)
    assert d.max() < MEL_MAX_TOL_DB and d.mean() < MEL_MEAN_TOL_DB, (d.max(), d.mean())
    # per-chunk max of the mel power
    pmax = np.array([FR.melspectrogram(w, 16000, n_mels, hop).max() for w in wave])
    assert np.allclose(cmax.cpu().numpy(), pmax, rtol=2e-5)


def test_mel_edge_signals(mta):
    fe = mta.MelFrontend(16000, 320, 512, "cuda")
    z, _ = fe(torch.zeros(1, 480000, device="cuda"))
    assert np.abs(z.cpu().numpy() + 100.0).max() < 2e-5                      # silence
    imp = np.zeros((1, 480000), np.float32); imp[0, 12345] = 1.0
    m, _ = fe(torch.from_numpy(imp).cuda())
    assert np.abs(m.cpu().numpy() - FR.audio_to_mel_batch(imp)).max() < MEL_MAX_TOL_DB
    half = np.concatenate([FR.synth_audio(1, 240000, seed=3)[0], np.zeros(240000, np.float32)])[None]
    m, _ = fe(torch.from_numpy(half).cuda())
    ref = FR.audio_to_mel_batch(half)
    assert np.abs(m.cpu().numpy() - ref).max() < MEL_MAX_TOL_DB
    assert np.allclose(m.cpu().numpy()[0, 0, :, 600:], ref.max() - 80.0, atol=1e-3)   # clamp floor active


def test_mel_unclamped_plus_chunk_max_equals_clamped(mta):
    wave = torch.from_numpy(FR.synth_audio(2, 48000, seed=11)).cuda()
    wave[1] *= 1e-3                                                          # per-CHUNK max, not per batch
    fe = mta.MelFrontend(16000, 320, 512, "cuda")
    a, cm = fe(wave, clamp=True)
    b, cm2 = fe(wave, clamp=False)
    floor = 10.0 * torch.log10(torch.clamp(cm2, min=1e-10)) - 80.0
    assert torch.equal(torch.maximum(b, floor[:, None, None, None]), a)
    assert float(a[1].max()) < float(a[0].max()) - 40


def test_audio_to_mel_dropin(mta):
    w = FR.synth_audio(1, 480000, seed=5)[0]
    m = mta.audio_to_mel(w)
    assert m.shape == (1, 1, 320, 938) and m.dtype == torch.float32 and not m.is_cuda
    assert np.abs(m.numpy()[0, 0] - FR.audio_to_mel(w)).max() < MEL_MAX_TOL_DB


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 192), (1000, 88, 1024), (4096, 4096, 512),
                                   (17000, 640, 192), (33000, 1024, 64)])     # the last three take the 256 x 256 tile
def test_gemm_bf16(mta, M, N, K):
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(M + N + K)
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    A = torch.randn(Mp, K, generator=g).bfloat16().cuda()
    W = torch.randn(Np, K, generator=g).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    Cc = torch.zeros(M, N, device="cuda")
    check(lib.mt_gemm_bf16_f32acc(ptr(A), K, ptr(W), K, ptr(bias), ptr(Cc), N, M, N, K, stream_ptr()))
    ref = A[:M].float().double().cpu() @ W[:N].float().double().cpu().t() + bias.double().cpu()
    err = (Cc.double().cpu() - ref).abs().max().item()
    assert err < 2e-4 * np.sqrt(K), err          # fp32 accumulation of exact bf16 products


# ------------------------------------------------------------------ LSTM layer (input projection + recurrence)
@pytest.mark.parametrize("B,T,H,K", [(2, 20, 16, 64), (5, 33, 32, 128), (32, 40, 512, 1024), (33, 12, 256, 192), (1, 50, 64, 64),
                                     (150, 9, 64, 64), (70, 11, 32, 64), (32, 300, 512, 128)])    # 5 / 3 batch groups: interleaved in one launch; the last: input projection on the 256 x 256 tile
def test_lstm_layer_matches_oracle(mta, B, T, H, K):
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + H)
    bound = 1.0 / np.sqrt(H)
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * bound
    x = torch.randn(B, T, K, generator=g).bfloat16().float()          # exactly representable inputs
    w_ih = [u(4 * H, K).bfloat16().float() for _ in range(2)]
    w_hh = [u(4 * H, H) for _ in range(2)]
    b_ih = [u(4 * H) for _ in range(2)]
    b_hh = [u(4 * H) for _ in range(2)]
    ref = torch.cat([R.lstm_dir(x, w_ih[d], w_hh[d], b_ih[d], b_hh[d], bool(d), R.Opts()) for d in range(2)], -1)
    ref16 = torch.cat([R.lstm_dir(x, w_ih[d], w_hh[d], b_ih[d], b_hh[d], bool(d), R.Opts(lstm_f16=True)) for d in range(2)], -1)
    M = T * B
    Mp, Np = (M + 127) // 128 * 128, (8 * H + 127) // 128 * 128
    X = torch.zeros(Mp, K); X[:M] = x.transpose(0, 1).reshape(M, K)    # row m = t*B + b
    Wp = torch.zeros(Np, K); Wp[:8 * H] = torch.cat(w_ih, 0)
    X, Wp = X.bfloat16().cuda(), Wp.bfloat16().cuda()
    bg = torch.cat([b_ih[0] + b_hh[0], b_ih[1] + b_hh[1]]).cuda()
    whh = torch.stack(w_hh).contiguous().cuda()
    gx = torch.empty(lib.mt_lstm_gx_bytes(B, T, H) // 4, device="cuda")
    hx = torch.full((lib.mt_lstm_hx_bytes(B, T, H) // 4,), float("nan"), device="cuda")
    sync = torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device="cuda")
    y = torch.empty(B, T, 2 * H, device="cuda")
    s = stream_ptr()
    check(lib.mt_gemm_lstm_gx(ptr(X), K, ptr(Wp), K, ptr(bg), ptr(gx), B, T, H, K, s))
    check(lib.mt_lstm_bidir_fwd(ptr(gx), ptr(whh), ptr(hx), ptr(sync), sync.numel(), B, T, H, s))
    check(lib.mt_lstm_unpack_f32(ptr(hx), ptr(y), B, T, H, s))
    torch.cuda.synchronize()
    assert int(sync[:4].view(torch.int32).item()) == 0, "hand-off timeout"
    # against the oracle with the kernel's rounding points (W_hh, exchanged h -> f16): an f16 ulp of h (2^-11) at most
    # where a rounding decision flips; against the fp32 LSTM of the reference: f16-operand noise, ~1e-3
    err16, err = (y.cpu() - ref16).abs().max().item(), (y.cpu() - ref).abs().max().item()
    assert err16 < 6e-4 and (y.cpu() - ref16).abs().mean().item() < 2e-5 and err < 2e-3, (err16, err)
    # next layer's A matrix
    K1 = (2 * H + 63) // 64 * 64
    X1 = torch.zeros(Mp, K1, dtype=torch.bfloat16, device="cuda")
    check(lib.mt_lstm_relayout_bf16(ptr(hx), ptr(X1), K1, B, T, H, s))
    # X1 = bf16(published f16 h), round-to-nearest-even
    want = y.transpose(0, 1).reshape(M, 2 * H)
    assert torch.equal(X1[:M, :2 * H], want.bfloat16())


@pytest.mark.parametrize("H,T", [(512, 40), (256, 33), (48, 21), (16, 9), (64, 50)])
def test_lstm_16_column_kernel_equals_the_32_column_kernel(mta, H, T):
    """B <= 16 runs lstm_rec16_kernel (16x16x32 MFMAs, K split in 32-wide steps, cells in two waves), B = 17 .. 32 lstm_rec_kernel: the SAME 16
    sequences, alone (B = 16) and with a 17th beside them, must give the same h for every step up to the f32 summation order of the K split
    (the 17th chunk cannot influence the others) -- inference and train mode (activated gates and cell states saved for the backward pass)."""
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(7 * H + T)
    nkb = H // 8
    whh = ((torch.rand(2, 4 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).cuda()
    gx17 = torch.randn(T, 2, nkb, 4, 8, 32, generator=g) * 1.5                 # [t][dir][unit block][gate][unit][batch slot]
    gx17[..., 17:] = 0.0
    gx16 = gx17.clone(); gx16[..., 16:] = 0.0
    s = stream_ptr()
    outs = {}
    for B, gx0 in ((16, gx16), (17, gx17)):
        for train in (False, True):
            gx = gx0.clone().reshape(-1).cuda()
            hx = torch.empty(lib.mt_lstm_hx_bytes(B, T, H) // 4, device="cuda")
            cx = torch.zeros(lib.mt_lstm_cx_bytes(B, T, H) // 4, device="cuda")
            sync = torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device="cuda")
            y = torch.empty(B, T, 2 * H, device="cuda")
            if train:
                check(lib.mt_lstm_bidir_fwd_train(ptr(gx), ptr(whh), ptr(hx), ptr(cx), ptr(sync), sync.numel(), B, T, H, s))
            else:
                check(lib.mt_lstm_bidir_fwd(ptr(gx), ptr(whh), ptr(hx), ptr(sync), sync.numel(), B, T, H, s))
            check(lib.mt_lstm_unpack_f32(ptr(hx), ptr(y), B, T, H, s))
            torch.cuda.synchronize()
            assert int(sync[:4].view(torch.int32).item()) == 0, "hand-off timeout"
            outs[(B, train)] = (y[:16].cpu(), gx.reshape(T, 2, nkb, 4, 8, 32)[..., :16].cpu(), cx.reshape(T, 2, nkb, 8, 32)[..., :16].cpu())
    for train in (False, True):
        a, b = outs[(16, train)], outs[(17, train)]
        assert torch.isfinite(a[0]).all() and float(a[0].abs().max()) > 0.05
        assert float((a[0] - b[0]).abs().max()) < 1.5e-3                       # h is published as f16: an ulp (2^-11 at |h| < 1) where a rounding flips
        assert float((a[0] - b[0]).abs().mean()) < 2e-5
    assert torch.equal(outs[(16, False)][0], outs[(16, True)][0])                 # train mode changes what is SAVED, not h
    ga, gb = outs[(16, True)][1], outs[(17, True)][1]                             # activated gates i, f, g, o overwrite gx in place
    assert float((ga - gb).abs().max()) < 2e-3 and float(ga.min()) >= -1.0 and float(ga.max()) <= 1.0
    assert float((outs[(16, True)][2] - outs[(17, True)][2]).abs().max()) < 5e-3  # cell states


@pytest.mark.parametrize("B,T,H,K", [(3, 20, 16, 64), (32, 40, 512, 1024), (96, 24, 512, 1024), (70, 11, 32, 64), (150, 9, 64, 64), (64, 36, 256, 512)])
def test_f16_gate_preactivations_equal_the_rounded_f32_ones(mta, B, T, H, K):
    """MT_GX_F16 (include/mt_hip.h): the projection GEMM stores W_ih x + b as f16 -- bit for bit the round-to-nearest-even of what
    it stores as f32, in the same layout, through every epilogue (per-element, 16-B rows, the LDS-staged whole-tile path at
    B % 32 == 0 and H % 256 == 0) -- and the recurrence fed with them (one to four interleaved batch groups, its loader wave
    streaming 2-KB blocks) publishes exactly the h of the f32 recurrence fed with those rounded values."""
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr, DT_F16, GX_F16
    g = torch.Generator().manual_seed(B * 7 + T * 3 + H)
    bound = 1.0 / np.sqrt(H)
    M = T * B
    Mp, Np = (M + 255) // 256 * 256, (8 * H + 127) // 128 * 128
    X = torch.zeros(Mp, K); X[:M] = torch.randn(M, K, generator=g)
    Wp = torch.zeros(Np, K); Wp[:8 * H] = (torch.rand(8 * H, K, generator=g) * 2 - 1) * bound
    X, Wp = X.half().cuda(), Wp.half().cuda()
    bg = ((torch.rand(8 * H, generator=g) * 2 - 1) * bound).cuda()
    whh = ((torch.rand(2, 4 * H, H, generator=g) * 2 - 1) * bound).contiguous().cuda()
    n_gx = lib.mt_lstm_gx_bytes(B, T, H) // 4
    gx32 = torch.zeros(n_gx, device="cuda")
    gx16 = torch.zeros(n_gx, device="cuda")                        # (the f16 form fills the first half)
    s = stream_ptr()
    check(lib.mt_gemm_lstm_gx_dt(ptr(X), K, ptr(Wp), K, ptr(bg), ptr(gx32), B, T, H, K, DT_F16, s))
    check(lib.mt_gemm_lstm_gx_dt(ptr(X), K, ptr(Wp), K, ptr(bg), ptr(gx16), B, T, H, K, DT_F16 | GX_F16, s))
    torch.cuda.synchronize()
    got16 = gx16.view(torch.float16)[:n_gx]
    assert torch.equal(got16, gx32.half())                         # (padded batch rows: both untouched zeros)
    assert float(gx16.view(torch.float16)[n_gx:].abs().max()) == 0.0      # nothing written past the f16 image
    outs = []
    for gxb, mode in ((gx32.half().float(), 0), (gx16, GX_F16)):
        hx = torch.full((lib.mt_lstm_hx_bytes(B, T, H) // 4,), float("nan"), device="cuda")
        sync = torch.empty(lib.mt_lstm_sync_bytes(B, H), dtype=torch.uint8, device="cuda")
        y = torch.empty(B, T, 2 * H, device="cuda")
        check(lib.mt_lstm_bidir_fwd_ex(ptr(gxb), ptr(whh), ptr(hx), ptr(sync), sync.numel(), B, T, H, mode, s))
        check(lib.mt_lstm_unpack_f32(ptr(hx), ptr(y), B, T, H, s))
        torch.cuda.synchronize()
        assert int(sync[:4].view(torch.int32).item()) == 0, "hand-off timeout"
        outs.append(y)
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0.01


# ------------------------------------------------------------------ whole model
@pytest.mark.parametrize("tag", ["small_a", "small_b"])
def test_cnnrnn_small_vs_reference_golden(mta, golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "small_models.npz"))
    nm, hs, nl, B, T, wseed, xseed = [int(v) for v in z[f"{tag}_cfg"]]
    sd = R.set_bn_flat(R.make_state_dict("cnn_rnn", nm, hs, nl, wseed), z[f"{tag}_bn"])
    model = mta.TranscriptionModel("cnn_rnn", n_mels=nm, hidden_size=hs, num_layers=nl, device="cuda")
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = _mel_in(B, nm, T, xseed)
    with torch.no_grad():
        got = model.model(x.cuda(), check_status=True).cpu()
        emu = R.cnnrnn_forward(sd, x, R.Opts(gemm_f16=True))
    golden = torch.from_numpy(z[f"{tag}_logits"])
    assert got.shape == golden.shape
    assert (got - emu).abs().max().item() < 2e-3          # same bf16 input rounding: only fp32 ordering differs
    assert (got - golden).abs().max().item() < 3e-2       # vs the fp32 reference itself (bf16 GEMM inputs)
    pred = model.predict(x.cuda(), threshold=0.5).cpu()
    flips = (pred != R.predict(golden, 0.5)).float().mean().item()
    assert flips < 0.02


@pytest.mark.parametrize("hidden,B", [(48, 70), (64, 100), (80, 33)])
def test_cnnrnn_odd_sizes_with_interleaved_batch_groups(mta, hidden, B):
    """Hidden sizes that are not whole 64-wide K tiles (the layer-to-layer projections then read re-laid-out rows, not the hx
    images) and batches of 2 - 4 batch groups with a ragged last one (the recurrence interleaves them; its padded rows carry no
    data): against the oracle with the same rounding points, and chunk for chunk against a forward over a slice of the batch."""
    nm, L, T = 32, 2, 24
    sd = R.make_state_dict("cnn_rnn", nm, hidden, L, seed=hidden)
    model = mta.TranscriptionModel("cnn_rnn", n_mels=nm, hidden_size=hidden, num_layers=L, device="cuda").eval()
    model.load_state_dict(sd, strict=True)
    x = _mel_in(B, nm, T, seed=B)
    with torch.no_grad():
        got = model.model(x.cuda(), check_status=True).cpu()
        emu = R.cnnrnn_forward(sd, x, R.Opts(gemm_f16=True))
        part = model.model(x[40:40 + 20].contiguous().cuda(), check_status=True).cpu() if B >= 60 else model.model(x[:5].contiguous().cuda(), check_status=True).cpu()
    assert got.shape == (B, 88, T) and torch.isfinite(got).all()
    assert (got - emu).abs().max().item() < 2e-3
    ref_slice = got[40:60] if B >= 60 else got[:5]
    assert (part - ref_slice).abs().max().item() < 1e-3            # (GEMM tile order differs with M: f32 summation order only)


def test_cnnrnn_canonical_vs_reference_golden(mta, golden_dir):
    c = np.load(os.path.join(golden_dir, "canonical_models.npz"))
    for tag in ("small_937", "small_938"):
        nm, hs, nl, B, T, wseed, xseed = [int(v) for v in c[f"{tag}_cfg"]]
        sd = R.set_bn_flat(R.make_state_dict("cnn_rnn", nm, hs, nl, wseed), c[f"{tag}_bn"])
        model = mta.TranscriptionModel("cnn_rnn", n_mels=nm, hidden_size=hs, num_layers=nl, device="cuda")
        model.load_state_dict(sd, strict=True)
        model.eval()
        x = _mel_in(B, nm, T, xseed)
        with torch.no_grad():
            got = model.model(x.cuda(), check_status=True).cpu().numpy()
        d = np.abs(got[:, ::5, ::7] - c[f"{tag}_sample"])
        assert d.max() < 3e-2, d.max()
        assert abs(got.mean() - c[f"{tag}_stats"][0]) < 2e-3


def test_errors_are_loud(mta):
    model = mta.TranscriptionModel("cnn_rnn", n_mels=32, hidden_size=16, num_layers=1, device="cuda").eval()
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 1, 32, 10))                  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        mta.TranscriptionModel("nope")
    with pytest.raises(ValueError):
        model(torch.zeros(1, 1, 31, 10, device="cuda"))
    with torch.no_grad():
        assert model(torch.zeros(2, 1, 32, 0, device="cuda")).shape == (2, 88, 1)


# ------------------------------------------------------------------ loss / predict / F1
def test_loss_matches_reference_golden(mta, golden_dir):
    g = np.load(os.path.join(golden_dir, "loss.npz"))
    logits = torch.from_numpy(g["logits"]).cuda()
    B, P, T = logits.shape
    targets = torch.from_numpy(np.unpackbits(g["targets"])[: B * P * T].reshape(B, P, T).astype(np.float32)).cuda()
    lengths = torch.from_numpy(g["lengths"])
    model = mta.TranscriptionModel("cnn_rnn", n_mels=32, hidden_size=16, num_layers=1, device="cuda")
    assert abs(float(model.compute_loss(logits, targets)) - float(g["loss_nolen"])) < 2e-6
    assert abs(float(model.compute_loss(logits, targets, lengths)) - float(g["loss_len"])) < 2e-6
    d = {"frame": logits, "onset": logits * 0.5 - 1.0, "offset": -logits + 0.25}
    assert abs(float(model.compute_loss(d, targets)) - float(g["loss_dict_nolen"])) < 2e-6
    assert abs(float(model.compute_loss(d, targets, lengths)) - float(g["loss_dict_len"])) < 2e-6
    assert float(model.compute_loss(logits, targets, torch.tensor([0, 0, 0]))) == 0.0
    lg = logits.clone().requires_grad_(True)
    model.compute_loss(lg, targets, lengths).backward()
    assert np.abs(lg.grad.cpu().numpy() - g["grad_len"]).max() < 1e-8
    on, off = mta.ops.onset_offset_targets(targets)
    ron, roff = R.onset_offset_targets(targets.cpu())
    assert torch.equal(on.cpu(), ron) and torch.equal(off.cpu(), roff)
    # bitwise reproducible
    assert float(model.compute_loss(logits, targets, lengths)) == float(model.compute_loss(logits, targets, lengths))


def test_predict_and_f1(mta, golden_dir):
    z = np.load(os.path.join(golden_dir, "small_models.npz"))
    logits = torch.from_numpy(z["small_a_logits"])
    B, P, T = logits.shape
    for th in (0.3, 0.5, 0.7):
        want = np.unpackbits(z[f"small_a_pred{int(th * 10)}"])[: B * P * T].reshape(B, P, T)
        got = mta.predict_from_logits(logits.cuda(), th).cpu().numpy()
        assert set(np.unique(got)) <= {0.0, 1.0}
        # the reference compares sigmoid(x) > t in fp32, the kernel x > log(t / (1 - t)): a cell may differ only where fp32's
        # sigmoid cannot tell x from the threshold (|x - logit(t)| of a few ulp of sigmoid = 6e-8 / (t (1 - t)))
        bad = got != want
        assert bad.mean() < 1e-4 and np.abs(logits.numpy()[bad] - np.log(th / (1 - th))).max(initial=0.0) < 2e-6, (th, int(bad.sum()))
    g = np.load(os.path.join(golden_dir, "f1.npz"))
    yt = torch.from_numpy(g["y_true"]).reshape(-1, 88, 20).cuda()
    yp = torch.from_numpy(g["y_pred"]).reshape(-1, 88, 20).cuda()
    got = mta.framewise_f1(yp, yt)
    assert np.abs(np.array(got) - g["f1"]).max() < 1e-12
    assert got[0] == 0.0                                             # zero_division=0
    lens = torch.tensor([20, 7, 0, 13, 20, 1, 20, 19])
    want = [R.f1_binary(yt[i, :, :int(L)].cpu(), yp[i, :, :int(L)].cpu()) for i, L in enumerate(lens)]
    assert np.allclose(mta.framewise_f1(yp, yt, lens), want, atol=1e-12)
    assert abs(mta.mean_f1(yp, yt, lens) - R.mean_f1(yp.cpu(), yt.cpu(), lens)) < 1e-12


# ------------------------------------------------------------------ end-to-end CLI (main.py surface)
def test_main_cli_end_to_end(mta, tmp_path):
    import subprocess, sys
    from scipy.io import wavfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wave = FR.synth_audio(1, 16000 * 40, seed=21)[0]                       # 40 s -> 2 chunks, the second zero-padded
    wavfile.write(str(tmp_path / "a.wav"), 16000, wave)
    sd = R.make_state_dict("cnn_rnn", 64, 32, 1, seed=2)
    sd["model.fc.bias"] += 0.2                                             # make some cells fire
    torch.save(sd, str(tmp_path / "m.pth"))
    out = tmp_path / "o.mid"
    r = subprocess.run([sys.executable, os.path.join(root, "main.py"), str(tmp_path / "a.wav"), str(tmp_path / "m.pth"), "-o", str(out),
                        "-d", "cuda", "-t", "0.5", "--model-type", "cnn_rnn", "--n-mels", "64", "--hidden-size", "32", "--num-layers", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Transcription completed successfully!" in r.stdout and out.read_bytes()[:4] == b"MThd"
    # same roll as the oracle pipeline on the same audio (chunk -> mel -> model -> threshold -> concat)
    from music_transcription_amd import transcribe as tr
    chunks, _ = tr.split_into_chunks(wave)
    model = tr.load_model(str(tmp_path / "m.pth"), "cuda", "cnn_rnn", 64, 32, 1)
    roll = tr.transcribe_chunks(model, chunks, 0.5, n_mels=64)
    ref = []
    for c in chunks:
        m = torch.from_numpy(FR.audio_to_mel(c, 16000, 64, 512))[None, None]
        ref.append(R.predict(R.cnnrnn_forward(sd, m), 0.5)[0].numpy())
    ref = np.concatenate(ref, axis=1)
    assert roll.shape == ref.shape == (88, 2 * 938)
    assert (roll != ref).mean() < 5e-3 and roll.sum() > 0                   # only |logit| ~ 0 cells may flip
    # the CLI's notes come from the device run-length: the same notes as the reference's host loop over the HIP roll
    notes_dev = tr.transcribe_chunks_to_notes(model, chunks, 0.5, n_mels=64)
    assert notes_dev == R.pianoroll_to_notes(roll, 16000 / 512) and len(notes_dev) > 0
    from music_transcription_amd import midi as midi_mod
    back = sorted((n.pitch, n.start, n.end) for n in midi_mod.MidiFile(str(out)).instruments[0].notes)     # the file the CLI wrote
    assert len(back) == len(notes_dev)
    for (p1, s1, e1), (p2, s2, e2) in zip(back, sorted(notes_dev)):
        assert p1 == p2 and abs(s1 - s2) < 3e-3 and abs(e1 - e2) < 3e-3              # SMF ticks: 1 / 440 s


# ------------------------------------------------------------------ roll -> notes on the device (row f2)
def test_roll_to_notes_device_matches_oracle(mta):
    """mt_roll_to_notes (threshold + chunk concatenation + run-length on the GPU) against the oracle's restatement of
    pianoroll_to_midi (main.py:189-226) -- same notes, same order; edge cases: silence, all-on, notes at the very first /
    last frame, notes crossing a chunk boundary, single-frame notes, a note buffer that has to grow."""
    from music_transcription_amd import transcribe as tr
    fs = 31.25
    g = torch.Generator().manual_seed(4)
    for NB, T, dens in ((1, 40, 0.3), (3, 57, 0.5), (5, 938, 0.04), (2, 300, 0.9), (40, 64, 0.5)):
        roll = (torch.rand(NB, 88, T, generator=g) < dens).float()
        roll[:, 0] = 0.0                                            # a silent pitch
        roll[:, 1] = 1.0                                            # one note over the whole recording
        roll[:, 2] = 0.0; roll[0, 2, 0] = 1.0; roll[-1, 2, -1] = 1.0   # single frames at both ends
        if NB > 1:
            roll[:, 3] = 0.0; roll[0, 3, -3:] = 1.0; roll[1, 3, :2] = 1.0      # crosses the chunk boundary: ONE note
        flat = roll.permute(1, 0, 2).reshape(88, NB * T).numpy()   # combine_piano_rolls: concatenation along time
        want = R.pianoroll_to_notes(flat, fs)
        got = tr.notes_from_logits_device(roll.cuda(), fs=fs, is_roll=True)
        assert got == want, (NB, T, len(got), len(want))
        assert tr.pianoroll_to_notes(flat, fs) == want             # (the host helper kept for callers that hold a roll)
        # from logits, with the predict threshold fused in
        logits = torch.randn(NB, 88, T, generator=g) * 2.0
        for thr in (0.3, 0.5, 0.7):
            pr = R.predict(logits, thr).permute(1, 0, 2).reshape(88, NB * T).numpy()
            assert tr.notes_from_logits_device(logits.cuda(), thr, fs) == R.pianoroll_to_notes(pr, fs)
    if NB > 1:
        n3 = [n for n in want if n[0] == 21 + 3]
        assert len(n3) == 1 and abs(n3[0][2] - n3[0][1] - 5 / fs) < 1e-9


# ------------------------------------------------------------------ optimizer step (training row a11)
def test_fused_adam_clip_matches_torch(mta):
    from music_transcription_amd.optim import FusedAdamClip, flatten_parameters
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 32, 3, 3), (64,), (2048, 300), (88, 1024), (7,)]
    ref_params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    my_params = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_params]
    ref_opt = torch.optim.Adam(ref_params, lr=1e-4, eps=1e-8, weight_decay=1e-5)       # scripts/train_cnn.py:290
    flat, grads = flatten_parameters(my_params)
    opt = FusedAdamClip(flat, grads, lr=1e-4, eps=1e-8, weight_decay=1e-5, max_norm=1.0)
    for step in range(4):
        scale = [5.0, 0.01, 1.0, 30.0][step]                                            # clipped and unclipped steps
        for rp, mp in zip(ref_params, my_params):
            gr = torch.randn(rp.shape, generator=g) * scale
            rp.grad = gr.clone()
            mp.grad.copy_(gr)
        norm = torch.nn.utils.clip_grad_norm_(ref_params, 1.0)                          # train_transcriber.py:134
        ref_opt.step()
        st = opt.step(sync_grads=False).cpu()
        assert abs(float(st[0]) - float(norm)) < 1e-4 * float(norm) and float(st[1]) == 1.0
        for rp, mp in zip(ref_params, my_params):
            assert (mp.detach().cpu() - rp.detach()).abs().max().item() < 5e-7          # 1 ulp at |p| ~ 4
    # NaN guard: a non-finite norm leaves parameters untouched (train_transcriber.py:137-142)
    before = flat.clone()
    my_params[0].grad[0, 0, 0, 0] = float("nan")
    st = opt.step(sync_grads=False).cpu()
    assert float(st[1]) == 0.0 and torch.equal(flat, before)


# ------------------------------------------------------------------ evaluation loop + one-pass threshold sweep
def test_evaluate_and_tune_threshold(mta, tmp_path):
    from music_transcription_amd import evaluate as ev
    nm, hs, nl = 32, 16, 1
    sd = R.make_state_dict("cnn_rnn", nm, hs, nl, seed=4)
    sd["model.fc.bias"] += 0.3
    model = mta.TranscriptionModel("cnn_rnn", n_mels=nm, hidden_size=hs, num_layers=nl, device="cuda").eval()
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(1)
    cache = str(tmp_path / "cache")
    Ts = [60, 60, 45, 60, 45]
    for i, T in enumerate(Ts):
        mel = _mel_in(1, nm, T, 50 + i)[0]
        with torch.no_grad():
            ref_logits = R.cnnrnn_forward(sd, mel[None])[0]
        roll = ((ref_logits + 0.3 * torch.randn(88, T, generator=g)) > 0.1).float()     # correlated labels
        mta.write_cache_chunk(cache, "test", i, mel, roll)
    mta.write_cache_metadata(cache, "test", [{} for _ in Ts], n_mels=nm)
    ds = mta.CachedMaestroDataset(cache, "test")
    mean_f1, per = ev.evaluate_dataset(model, ds, threshold=0.5)
    # reference semantics on the CPU oracle: batch 1, sigmoid > t, sklearn-style F1, unweighted mean
    want = []
    for i in range(len(ds)):
        mel, roll = ds[i]
        with torch.no_grad():
            pred = R.predict(R.cnnrnn_forward(sd, mel[None]), 0.5)[0]
        want.append(R.f1_binary(roll, pred))
    assert np.allclose(per, want, atol=2e-2) and abs(mean_f1 - np.mean(want)) < 1e-2   # only |logit| ~ 0 cells can flip
    # one-pass sweep == per-threshold predict + F1 on the same logits (exact)
    lr = ev.collect_logits(model, ds, range(len(ds)))
    ths = [0.05, 0.3, 0.5, 0.7, 0.95]
    sweep = ev.f1_at_thresholds(lr, ths)
    for n, (_, lg, roll) in enumerate(lr):
        for k, t in enumerate(ths):
            direct = mta.framewise_f1(mta.predict_from_logits(lg[None], t), roll[None])[0]
            assert abs(sweep[n, k] - direct) < 1e-12
    best_t, best_f1 = ev.tune_threshold(model, ds, log=None)
    grid = ev.f1_at_thresholds(lr, np.arange(0.01, 0.995, 0.005)).mean(0)
    assert best_f1 >= grid.max() - 5e-3 and 0.01 <= best_t <= 0.99                       # coarse-to-fine finds the plateau


def test_full_size_coscheduled_batches_equal_separate_forwards(mta):
    """Three batches of 32 chunks in ONE forward (B = 96: the recurrence interleaves the three batch groups inside one
    persistent launch, csrc/lstm.hip NG) give, chunk for chunk, the logits of three separate B = 32 forwards."""
    from oracle import frontend_ref as FR
    sd = R.make_state_dict("cnn_rnn", 320, 512, 3, seed=1)
    model = mta.TranscriptionModel("cnn_rnn", n_mels=320, hidden_size=512, num_layers=3, device="cuda").eval()
    model.load_state_dict(sd, strict=True)
    wave = torch.from_numpy(FR.synth_audio(6, 480000, seed=31)).cuda()
    wave = torch.cat([wave * s for s in (1.0, 0.5, 0.25, 0.9, 0.7, 0.6, 0.8, 0.3, 0.45, 0.55, 0.65, 0.75, 0.85, 0.95, 0.35, 0.15)], 0)      # 96 distinct chunks
    mel, _ = mta.MelFrontend(16000, 320, 512, "cuda")(wave, clamp=True)
    with torch.no_grad():
        big = model(mel).clone()
        model.model.raise_on_handoff_timeout()
        for k in range(3):
            part = model(mel[32 * k:32 * (k + 1)].contiguous())
            assert (part - big[32 * k:32 * (k + 1)]).abs().max().item() < 2e-3, k      # (GEMM tile order differs with M)
        model.model.raise_on_handoff_timeout()
    assert torch.isfinite(big).all() and float(big.std()) > 1e-3


# ------------------------------------------------------------------ BASELINE-size properties (no oracle needed at this size)
def test_full_size_batch_independence_and_determinism(mta):
    """configs[1] shape (B = 32 x 30 s, n_mels 320, hidden 512, 3 layers): chunks are independent data-parallel work, so
    (i) a chunk's logits do not depend on what else is in the batch or where it sits in it, (ii) the forward is
    run-to-run deterministic, (iii) mel + forward from waveforms equals forward of the separately computed mel."""
    from oracle import frontend_ref as FR
    sd = R.make_state_dict("cnn_rnn", 320, 512, 3, seed=0)
    model = mta.TranscriptionModel("cnn_rnn", n_mels=320, hidden_size=512, num_layers=3, device="cuda").eval()
    model.load_state_dict(sd, strict=True)
    wave = torch.from_numpy(FR.synth_audio(4, 480000, seed=77)).cuda()
    wave = torch.cat([wave * s for s in (1.0, 0.5, 0.25, 0.9, 0.7, 0.6, 0.8, 0.3)], 0)          # 32 distinct chunks
    fe = mta.MelFrontend(16000, 320, 512, "cuda")
    mel, _ = fe(wave, clamp=True)
    assert mel.shape == (32, 1, 320, 938)
    with torch.no_grad():
        full = model(mel).clone()
        again = model(mel).clone()
        assert torch.equal(full, again)                                               # (ii)
        perm = torch.randperm(32, generator=torch.Generator().manual_seed(1)).cuda()
        shuffled = model(mel[perm].contiguous())
        assert torch.equal(shuffled, full[perm])                                      # (i) position in the batch
        for k in (0, 13, 31):
            alone = model(mel[k:k + 1].contiguous())
            assert (alone[0] - full[k]).abs().max().item() < 1e-5                     # (i) batch composition (tile shapes differ)
        sub = model(mel[:7].contiguous())
        assert (sub - full[:7]).abs().max().item() < 1e-5
    assert torch.isfinite(full).all() and full.shape == (32, 88, 938)
    model.model.raise_on_handoff_timeout(32, 938)


@pytest.mark.parametrize("B,n_mels,T,dt", [(2, 320, 937, "f16"), (3, 229, 100, "f16"), (1, 64, 50, "bf16"), (2, 38, 33, "f16"), (5, 32, 17, "bf16")])
def test_fused_conv1_conv2_is_bit_identical_to_the_two_kernels(mta, B, n_mels, T, dt):
    """conv12_kernel (act1 computed per tile in LDS from the mel tile, never written to HBM) against mt_conv1_bn_relu_pool_dt followed by
    mt_conv2_bn_relu_pool_dt on the same operands: the same X0, bit for bit -- with the 80-dB clamp floor taken from the chunk maxima,
    image borders, an odd number of pooled rows (n_mels = 38: F1 = 19), tiles that overhang T and F."""
    from music_transcription_amd import _lib
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(B * 1000 + n_mels + T)
    d = _lib.DT_F16 if dt == "f16" else _lib.DT_BF16
    t16 = torch.float16 if dt == "f16" else torch.bfloat16
    mel = (torch.rand(B, n_mels, T, generator=g) * 90.0 - 100.0).cuda()
    cmax = (10.0 ** ((torch.rand(B, generator=g) * 20.0 - 10.0) / 10.0)).cuda()          # floors between -90 and -70 dB: part of the mel is clamped
    w1, b1 = (torch.randn(32, 9, generator=g) * 0.05).cuda(), (torch.randn(32, generator=g) * 0.5 + 1.0).cuda()
    w2 = (torch.randn(64, 288, generator=g) * 0.05).to(t16).cuda()
    b2 = (torch.randn(64, generator=g) * 0.3).cuda()
    F1, Fo2 = n_mels // 2, n_mels // 4
    ldx, Mp = Fo2 * 64 + 64, (T * B + 127) // 128 * 128
    s = stream_ptr()
    outs = []
    for fused in (False, True):
        X0 = torch.full((Mp, ldx), 7.0, dtype=t16, device="cuda")
        if fused:
            check(lib.mt_conv12_bn_relu_pool_dt(ptr(mel), ptr(cmax), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(X0), ldx, B, n_mels, T, d, s), "conv12")
        else:
            act1 = torch.empty(B, F1, T, 32, dtype=t16, device="cuda")
            check(lib.mt_conv1_bn_relu_pool_dt(ptr(mel), ptr(cmax), ptr(w1), ptr(b1), ptr(act1), B, n_mels, T, d, s), "conv1")
            check(lib.mt_conv2_bn_relu_pool_dt(ptr(act1), ptr(w2), ptr(b2), ptr(X0), ldx, B, F1, T, d, s), "conv2")
        torch.cuda.synchronize()
        outs.append(X0)
    a, b = outs
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    live = a[:T * B, :Fo2 * 64].float()
    assert torch.isfinite(live).all() and float(live.abs().max()) > 0.1 and bool((a[:, Fo2 * 64:] == 7.0).all())     # and nothing written beside the tile
    # without chunk maxima (no clamp) as well
    outs = []
    for fused in (False, True):
        X0 = torch.zeros(Mp, ldx, dtype=t16, device="cuda")
        if fused:
            check(lib.mt_conv12_bn_relu_pool_dt(ptr(mel), None, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(X0), ldx, B, n_mels, T, d, s), "conv12")
        else:
            act1 = torch.empty(B, F1, T, 32, dtype=t16, device="cuda")
            check(lib.mt_conv1_bn_relu_pool_dt(ptr(mel), None, ptr(w1), ptr(b1), ptr(act1), B, n_mels, T, d, s), "conv1")
            check(lib.mt_conv2_bn_relu_pool_dt(ptr(act1), ptr(w2), ptr(b2), ptr(X0), ldx, B, F1, T, d, s), "conv2")
        outs.append(X0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert lib.mt_cnnrnn_conv_fused() in (0, 1)                                     # (opt-in for the whole-model forward: MT_CONV_FUSED=1)


# ------------------------------------------------------------------ audio decode (row f3): GPU resampler vs scipy
@pytest.mark.parametrize("rate,ch,dtype,n", [(44100, 2, "int16", 200000), (48000, 1, "float32", 150001), (16000, 2, "int16", 50000),
                                             (44100, 2, "int32", 44100), (22050, 1, "int16", 33333),
                                             # up-sampling (P = 2 phases per tile) / fewer than 4096 outputs (the thread-per-output kernel)
                                             (8000, 1, "int16", 30000), (44100, 2, "int16", 5000)])
def test_load_audio_device_matches_scipy_resample_poly(mta, tmp_path, rate, ch, dtype, n):
    from math import gcd
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    from music_transcription_amd import transcribe as tr
    rng = np.random.default_rng(n)
    t = np.arange(n) / rate
    sig = np.stack([0.4 * np.sin(2 * np.pi * 440.0 * t + c) + 0.05 * rng.standard_normal(n) for c in range(ch)], 1)
    if dtype == "int16":
        data = (sig * 32767).astype(np.int16); mono = data.astype(np.float64).mean(1) / 32768.0
    elif dtype == "int32":
        data = (sig * 2147483647).astype(np.int32); mono = data.astype(np.float64).mean(1) / 2147483648.0
    else:
        data = sig.astype(np.float32); mono = data.astype(np.float64).mean(1)
    path = str(tmp_path / "x.wav")
    wavfile.write(path, rate, data if ch > 1 else data[:, 0])
    y = tr.load_audio_device(path, 16000, "cuda")
    g = gcd(rate, 16000)
    up, down = 16000 // g, rate // g
    ref = resample_poly(mono, up, down, window=tr.resample_fir(up, down)) if rate != 16000 else mono       # scipy, the SAME designed taps
    assert y.is_cuda and y.dtype == torch.float32 and y.numel() == len(ref)
    assert np.abs(y.cpu().numpy() - ref).max() < 5e-6
    if rate != 16000:                                           # the natural-order entry point computes the same sum
        from music_transcription_amd._lib import lib, check, ptr, stream_ptr
        _, _, h, npr, n_out = tr.resample_plan(rate, 16000, n)
        src = torch.from_numpy(np.ascontiguousarray(data.reshape(n, ch))).cuda()
        y2 = torch.empty(n_out, device="cuda")
        check(lib.mt_resample_poly(ptr(src), n, ch, {"int16": 0, "int32": 1, "float32": 2}[dtype], ptr(torch.from_numpy(h).cuda()), len(h), up, down, npr,
                                   ptr(y2), n_out, stream_ptr()))
        assert float((y2 - y).abs().max()) < 2e-6
    chunks, dur = tr.split_into_chunks_device(y)
    assert chunks.shape == (max(1, -(-len(ref) // 480000)), 480000) and abs(dur - len(ref) / 16000.0) < 1e-9
    assert torch.equal(chunks.reshape(-1)[:len(ref)], y) and float(chunks.reshape(-1)[len(ref):].abs().sum()) == 0.0


def test_resampler_filter_is_converged_for_the_mel_and_the_rolls(mta, tmp_path):
    """f3 bound (the reference resamples with soxr_hq, which is not available here).  44.1 kHz stereo audio with strong energy
    ABOVE 8 kHz -- noise up to 22 kHz, partials at 9 / 12 / 15 kHz louder than the piano-range tones -- is resampled with the
    shipped filter (pass band 0.913 Nyquist, stop band from Nyquist, 120 dB) and with a 4x sharper one (0.978 Nyquist, 150 dB,
    four times the taps).
      A. Signal with a spectral gap over both filters' transition bands (7.0-8.7 kHz): everything the two filters are SPECIFIED to
         treat alike.  mel dB within 0.15 dB on every bin within 50 dB of the chunk maximum (1 dB within 60 dB in the pass band;
         down to the 80 dB clamp floor the bins hold residues and are reported), the thresholded rolls of a trained-scale
         model differ in < 1 % of the cells: no aliasing, no imaging, no pass-band ripple reaches the model.
      B. The same with the gap filled (full-band noise + tones at 7.5 / 7.7 kHz): the filters now differ BY DESIGN between 7.3
         and 8 kHz (the sharper one passes up to 7.8 kHz; soxr_hq rolls off from 7.3 kHz like the shipped one).  Reported: the
         dB difference of the bins inside the pass band (STFT leakage of the transition-band content) and of the ten bins that
         reach into 7.3-8 kHz, and how many roll cells move."""
    from music_transcription_amd import transcribe as tr
    from music_transcription_amd._lib import lib, check, ptr, stream_ptr
    rate, secs = 44100, 30
    n = rate * secs
    rng = np.random.default_rng(11)
    t = np.arange(n) / rate
    white = rng.standard_normal(n)
    spec = np.fft.rfft(white)
    f = np.fft.rfftfreq(n, 1.0 / rate)
    gap = (f > 7000.0) & (f < 8700.0)
    noise_gap = np.fft.irfft(np.where(gap, 0.0, spec), n)
    env = 0.6 + 0.4 * np.sin(2 * np.pi * 0.7 * t)                         # smooth: an abrupt envelope would splatter the loud 9 kHz partial into the gap

    def tones(fs):
        return sum(a * env * np.sin(2 * np.pi * f0 * t) for f0, a in fs)
    base = tones(((220.0, 0.12), (1760.0, 0.08), (6500.0, 0.04), (9000.0, 0.16), (12000.0, 0.16), (15000.0, 0.12)))   # peak sum 0.68: nothing clips
    sig_a = 0.02 * noise_gap + base
    sig_b = 0.02 * white + base + tones(((7500.0, 0.04), (7700.0, 0.04)))
    assert np.abs(sig_a).max() < 0.95 and np.abs(sig_b).max() < 0.95
    fe = mta.MelFrontend(16000, 320, 512, "cuda")
    fir_sharp = tr.resample_fir(160, 441, passband=0.978, stopband=1.0, rejection_db=150.0)
    assert len(fir_sharp) > 3.5 * len(tr.resample_fir(160, 441))

    def mels(sig):
        data = np.stack([sig, 0.8 * sig], 1).astype(np.float32)
        src = torch.from_numpy(data).cuda()
        out = []
        for fir in (None, fir_sharp):
            up, down, h, npr, n_out = tr.resample_plan(rate, 16000, n, fir=fir)
            tab = torch.from_numpy(tr.polyphase_table(h, up)).cuda()
            y = torch.empty(n_out, device="cuda")
            check(lib.mt_resample_polyphase(ptr(src), n, 2, 2, ptr(tab), tab.shape[1], up, down, npr, ptr(y), n_out, stream_ptr()))
            assert y.numel() == 480000
            with torch.no_grad():
                out.append(fe(y[None], clamp=True)[0])
        return out

    def mel_to_hz(m):                                                     # Slaney scale (librosa.mel_frequencies, htk=False)
        return np.where(m >= 15.0, 1000.0 * np.exp(np.log(6.4) / 27.0 * (m - 15.0)), 200.0 / 3 * m)
    mmax = 15.0 + np.log(8000.0 / 1000.0) / (np.log(6.4) / 27.0)
    centres = mel_to_hz(np.linspace(0.0, mmax, 322))
    inband = centres[2:] <= 0.913 * 8000.0                                # bins whose whole triangle lies in the pass band

    sd0 = R.make_state_dict("cnn_rnn", 320, 512, 3, seed=77)
    model = mta.TranscriptionModel("cnn_rnn", n_mels=320, hidden_size=512, num_layers=3, device="cuda").eval()
    res = {}
    for tag, sig in (("A", sig_a), ("B", sig_b)):
        mel_s, mel_i = mels(sig)
        ms, mi = mel_s[0, 0].cpu().numpy(), mel_i[0, 0].cpu().numpy()
        d = np.abs(ms - mi)
        lev = {}
        for L in (50, 60, 80):                                            # bins within L dB of the chunk maximum (80 = the clamp floor)
            above = (mi > float(mi.max()) - L + 1.0) & (ms > float(mi.max()) - L + 1.0)
            lev[L] = (float(d[inband][above[inband]].max()), float(d[~inband][above[~inband]].max()) if above[~inband].any() else 0.0)
        sd, _ = R.trained_scale_state_dict(sd0, "cnn_rnn", mel_i.cpu())
        model.load_state_dict(sd, strict=True)
        with torch.no_grad():
            la, lb = model.model(mel_s, check_status=True).cpu(), model.model(mel_i, check_status=True).cpu()
        flips = (la > 0) != (lb > 0)
        res[tag] = dict(lev=lev, dlogit=float((la - lb).abs().max()), flips=int(flips.sum()), far=float(lb[flips].abs().max()) if flips.any() else 0.0)
        print(f"\n[{tag}] mel dB shipped vs 4x sharper filter, max |d| over (pass-band bins, the {int((~inband).sum())} bins reaching into 7.3-8 kHz) among bins "
              f"within 50 / 60 / 80 dB of the chunk maximum: {lev[50][0]:.3f}, {lev[50][1]:.3f} / {lev[60][0]:.3f}, {lev[60][1]:.3f} / {lev[80][0]:.3f}, {lev[80][1]:.3f}; "
              f"logits max |d| {res[tag]['dlogit']:.4f}; roll cells flipped {res[tag]['flips']} of {flips.numel()}, farthest from the threshold {res[tag]['far']:.3f}")
    # A: what carries the signal (within 50 dB of the maximum) agrees to 0.15 dB everywhere; nearer the clamp floor a bin holds 1e-6
    #    of the maximum's power and a dB difference there is a difference of residues (reported above)
    assert max(res["A"]["lev"][50]) <= 0.15 and res["A"]["lev"][60][0] <= 1.0, res["A"]
    assert res["A"]["flips"] < 0.01 * 88 * 938 and res["A"]["far"] <= 1.0, res["A"]
    # B is a report of the by-design difference; these bounds only catch a broken filter
    assert res["B"]["lev"][50][0] <= 0.5 and res["B"]["flips"] < 0.1 * 88 * 938, res["B"]


@pytest.mark.parametrize("tag", ["small_a", "small_b"])
def test_fused_input_projection_matches_goldens(mta, golden_dir, tag):
    """CNNRNNModel.fuse_input_projection: layers > 0 project their input inside the recurrence (csrc/lstm.hip, XP) -- same
    tolerances against the reference goldens and the emulating oracle as the GEMM path, and close to it."""
    z = np.load(os.path.join(golden_dir, "small_models.npz"))
    nm, hs, nl, B, T, seed_w, seed_x = [int(v) for v in z[f"{tag}_cfg"]]
    sd = R.make_state_dict("cnn_rnn", nm, hs, nl, seed_w)
    R.set_bn_flat(sd, z[f"{tag}_bn"])
    model = mta.TranscriptionModel("cnn_rnn", n_mels=nm, hidden_size=hs, num_layers=nl, device="cuda")
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = _mel_in(B, nm, T, seed_x)
    with torch.no_grad():
        plain = model.model(x.cuda(), check_status=True).cpu()
        model.model.fuse_input_projection = True
        fused = model.model(x.cuda(), check_status=True).cpu()
        emu = R.cnnrnn_forward(sd, x, R.Opts(gemm_f16=True, gx_f16=False))     # (layers > 0 take no gx buffer: nothing rounded there)
    # a different code path with the same answer up to the f16 rounding of the plain path's gate pre-activations (layers > 0)
    assert (fused - plain).abs().max().item() < 2e-3
    assert (fused - emu).abs().max().item() < 2e-3
    assert np.abs(fused.numpy() - z[f"{tag}_logits"]).max() < 3e-2


def test_fused_input_projection_full_size(mta):
    from oracle import frontend_ref as FR
    sd = R.make_state_dict("cnn_rnn", 320, 512, 3, seed=0)
    model = mta.TranscriptionModel("cnn_rnn", n_mels=320, hidden_size=512, num_layers=3, device="cuda").eval()
    model.load_state_dict(sd, strict=True)
    wave = torch.from_numpy(FR.synth_audio(4, 480000, seed=5)).cuda()
    mel, _ = mta.MelFrontend(16000, 320, 512, "cuda")(torch.cat([wave, 0.5 * wave, 0.25 * wave, 0.7 * wave]), clamp=True)
    with torch.no_grad():
        plain = model(mel).clone()
        model.model.fuse_input_projection = True
        fused = model(mel).clone()
        again = model(mel).clone()
    model.model.raise_on_handoff_timeout(16, 938)
    assert torch.equal(fused, again) and (fused - plain).abs().max().item() < 3e-3


# ------------------------------------------------------------------ BASELINE configs[4]: whole-corpus transcription, sharded over ranks
def test_transcribe_corpus_two_ranks_matches_oracle_pipeline(mta, tmp_path):
    """scripts/transcribe_corpus.py on a 6-recording corpus of WAV files (44.1 kHz stereo int16 and 16 kHz mono float), two
    ranks (torch.distributed.run, gloo between processes that share this box's GPU): recordings are LPT-sharded with no
    data-path collective, every recording is transcribed exactly once, per-recording F1 values are gathered, and the rolls
    equal the CPU oracle pipeline's (decode -> chunk -> mel -> model -> threshold -> concatenate) up to |logit| ~ 0 cells."""
    import json
    import subprocess
    import sys
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wav_dir, dump = tmp_path / "corpus", tmp_path / "rolls"
    wav_dir.mkdir()
    nm, H, L = 64, 32, 1
    sd = R.make_state_dict("cnn_rnn", nm, H, L, seed=2)
    sd["model.fc.bias"] += 0.2
    torch.save(sd, str(tmp_path / "m.pth"))
    durs = [41.0, 12.5, 65.0, 30.0, 33.3, 7.0]                          # seconds: 1-3 chunks each, the last one zero-padded
    waves16 = {}
    rng = np.random.default_rng(0)
    for i, d in enumerate(durs):
        name = f"rec_{i}"
        if i % 2 == 0:                                                    # 16 kHz mono float: no resampling -> sample-exact oracle input
            w = FR.synth_audio(1, int(16000 * d), seed=50 + i)[0]
            wavfile.write(str(wav_dir / (name + ".wav")), 16000, w)
            waves16[name] = w
        else:                                                             # 44.1 kHz stereo int16: decode + channel mean + resample on the GPU
            w = FR.synth_audio(1, int(44100 * d), seed=50 + i, sr=44100)[0]
            st = np.stack([w, 0.5 * w], 1)
            wavfile.write(str(wav_dir / (name + ".wav")), 44100, (st * 32767.0).astype(np.int16))
            waves16[name] = None
        T_total = -(-int(16000 * d) // 480000) * 938
        np.save(str(wav_dir / (name + ".roll.npy")), (rng.random((88, T_total)) < 0.05).astype(np.float32))
    env = dict(os.environ, MT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(root, "scripts", "transcribe_corpus.py"), "--wav-dir", str(wav_dir), "--model", str(tmp_path / "m.pth"), "--model-type", "cnn_rnn",
           "--n-mels", str(nm), "--hidden-size", str(H), "--num-layers", str(L), "--batch", "4", "--dump-rolls", str(dump)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["recordings"] == 6 and out["n_gpus"] == 2 and out["chunks"] == sum(-(-int(16000 * d) // 480000) for d in durs)
    assert len(out["per_recording_f1"]) == 6 and all(0.0 <= v <= 1.0 for v in out["per_recording_f1"])
    assert sorted(os.listdir(dump)) == sorted(f"rec_{i}.roll.bits.npy" for i in range(6))          # each recording exactly once
    # the 16 kHz recordings against the oracle pipeline (main.py:60-100 chunking, :103-130 mel, model, :153-159 threshold, :164-186 concat)
    for i in (0, 2, 4):
        name = f"rec_{i}"
        w = waves16[name]
        n_ch = -(-len(w) // 480000)
        wp = np.zeros(n_ch * 480000, np.float32); wp[:len(w)] = w
        ref = []
        for c in wp.reshape(n_ch, 480000):
            mel = torch.from_numpy(FR.audio_to_mel(c, 16000, nm, 512))[None, None]
            ref.append(R.predict(R.cnnrnn_forward(sd, mel), 0.5)[0].numpy())
        ref = np.concatenate(ref, axis=1)
        got = np.unpackbits(np.load(str(dump / (name + ".roll.bits.npy"))), axis=1)[:, :ref.shape[1]]
        assert got.shape == ref.shape and (got != ref).mean() < 5e-3 and got.sum() > 0, (name, (got != ref).mean())
        truth = np.load(str(wav_dir / (name + ".roll.npy")))
        assert abs(out["per_recording_f1"][i] - R.f1_binary(truth, got)) < 1e-9                      # the gathered F1 is this recording's


def test_pcm_source_equals_the_wav_path_and_prefetches(mta, tmp_path):
    """corpus.PcmSource (pinned host PCM -> H2D on a copy stream, two recordings ahead -> mt_resample_polyphase -> chunks): the same
    samples as transcribe.load_audio_device on the WAV file holding that PCM, for recordings taken in shard order; the filter
    table of a rate pair is built once."""
    from scipy.io import wavfile
    from music_transcription_amd import corpus, transcribe as tr
    pcm, rates = {}, {}
    for i, (rate, secs, ch) in enumerate(((44100, 31.0, 2), (48000, 3.0, 1), (44100, 0.7, 2), (16000, 2.0, 2))):
        w = FR.synth_audio(1, int(rate * secs), seed=30 + i, sr=rate)[0]
        x = np.stack([w, 0.4 * w][:ch], 1)
        p16 = (x * 32767.0).astype(np.int16)
        wavfile.write(str(tmp_path / f"r{i}.wav"), rate, p16)
        pcm[i], rates[i] = torch.from_numpy(p16).pin_memory(), rate
    ids = [2, 0, 3, 1]
    src = corpus.PcmSource(lambda i: pcm[i], lambda i: rates[i], ids, "cuda", ahead=2)
    for k, i in enumerate(ids):
        got = src(i)
        assert len(src.inflight) == min(2, len(ids) - 1 - k)                         # the next two recordings are already on their way
        want = tr.split_into_chunks_device(tr.load_audio_device(str(tmp_path / f"r{i}.wav"), 16000, "cuda"))[0]
        assert got.shape == want.shape and got.shape[1] == 480000 and torch.equal(got, want), i
    assert src.bytes_h2d == sum(p.numel() * 2 for p in pcm.values())
    with pytest.raises(ValueError):
        tr.resample_pcm_device(torch.zeros(10, 2, dtype=torch.float64, device="cuda"), 44100)


def test_oversubscribed_persistent_launches_fail_fast(mta):
    """The recurrence kernels wait on their own workgroups, so all launches in flight must be co-resident.  The library keeps
    the persistent launches pending per stream (csrc/residency.hip) and refuses one that would not fit -- immediately, with
    MT_EUNSUPPORTED, instead of a 2-second spin timeout per layer.  Four forwards with the fused input projection (one
    workgroup per CU each, 128 CUs per launch) on four streams: the third is refused while two are pending."""
    from music_transcription_amd import _lib
    from oracle import frontend_ref as FR
    sd = R.make_state_dict("cnn_rnn", 320, 512, 3, seed=0)
    model = mta.TranscriptionModel("cnn_rnn", n_mels=320, hidden_size=512, num_layers=3, device="cuda").eval()
    model.load_state_dict(sd, strict=True)
    net = model.model
    net.fuse_input_projection = True
    wave = torch.from_numpy(FR.synth_audio(2, 480000, seed=5)).cuda()
    mel, _ = mta.MelFrontend(16000, 320, 512, "cuda")(torch.cat([wave] * 16), clamp=True)        # B = 32, T = 938
    with torch.no_grad():
        ref = net(mel).clone()                                                                    # packs, warms up
        torch.cuda.synchronize()
        assert _lib.lib.mt_persistent_cus_in_flight(None) == 0
        # the host-side bound of the model refuses a third stream outright ...
        streams = [torch.cuda.Stream() for _ in range(4)]
        nbytes = _lib.lib.mt_cnnrnn_workspace_bytes(net._ensure_packed(mel.device)["struct"], 32, 938)
        for s_ in streams:                                  # a cached block per stream: the workspace of a new stream is then handed out
            with torch.cuda.stream(s_):                     # without a hipMalloc (milliseconds: the forwards before it would have drained)
                warm = [torch.empty(nbytes, dtype=torch.uint8, device="cuda"), torch.empty(32, 88, 938, device="cuda")]
                del warm
        torch.cuda.synchronize()
        outs = []
        with pytest.raises(_lib.MtError, match="caller streams"):
            for s_ in streams:
                with torch.cuda.stream(s_):
                    outs.append(net(mel))
        torch.cuda.synchronize()
        net.raise_on_handoff_timeout()
        # (a stream that has already drained no longer counts, so the refusal comes with the third or the fourth forward)
        assert 2 <= len(outs) <= 3 and all(torch.equal(o, ref) for o in outs)
        # ... and so does the library when that bound is bypassed (another process, a caller of the C ABI)
        net._ws.clear()
        net._check_inflight_bound = lambda *a, **k: None
        t0 = __import__("time").perf_counter()
        outs = []
        with pytest.raises(_lib.MtError, match="persistent"):
            for s_ in streams:
                with torch.cuda.stream(s_):
                    outs.append(net(mel))
        dt = __import__("time").perf_counter() - t0
        torch.cuda.synchronize()
        assert dt < 1.0                                                          # refused at launch time, not after a spin bound
        assert 2 <= len(outs) <= 3 and all(torch.equal(o, ref) for o in outs)    # the admitted forwards are unharmed
        assert _lib.lib.mt_persistent_cus_in_flight(None) == 0
        del net._check_inflight_bound
        net._ws.clear()
        assert torch.equal(net(mel), ref)                                        # and the model keeps working
        net.raise_on_handoff_timeout()


# ------------------------------------------------------------------ the driver's command
def test_bench_line_contract(tmp_path):
    """`python bench.py --steps K --warmup W` prints ONE JSON line with the contract's fields; K steps of 32 chunks are timed
    whether or not K is a multiple of the batches per forward (the left-over steps run as one smaller forward)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail_path = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, MT_BENCH_DETAIL=detail_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]     # ONE line, and it is the last thing on stdout
    assert len(lines[0]) < 4096                                                   # the driver keeps a bounded tail (round 3: 21.7 KB, unparsed)
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "configs1_literal_b32", "sections"):
        assert k in d, k
    assert d["steps"] == 10 and d["warmup"] == 4 and d["n_gpus"] == 1 and d["unit"] == "chunks/s" and d["value"] > 1000
    assert abs(d["value"] - 32 * 10 / (d["ms_per_step"] * 10 / 1e3)) / d["value"] < 1e-3
    c = d["config"]
    assert c["batch_per_gpu"] == 32 and c["coscheduled_batches_per_forward"] == 4 and c["streams_per_gpu"] == 4 and "workload" in c
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["avg_launch_ms"] > 0
    assert "traffic" in rf
    # every kernel's roofline fraction in one compact object, for the three workloads
    assert {"all", "all_large", "all_train"} <= set(rf) and len(rf["all"]) >= 9 and all(0.0 < v < 1.0 for v in rf["all"].values())
    lit = d["configs1_literal_b32"]                                               # BASELINE configs[1] as written: one batch of 32 per forward
    assert lit["value"] > 1000 and lit["streams"] in (3, 4) and lit["one_in_flight"] > 1000
    assert {"configs2_large_b16", "configs3_train_b16", "train_large_b16", "configs4_corpus", "configs4_corpus_from_pcm"} <= set(d["sections"])
    for sec, o in d["sections"].items():
        assert "error" not in o and o["value"] > 0, (sec, o)
    # the full record (stage tables, schedules) sits beside it
    full = json.load(open(detail_path))
    assert full["value"] == d["value"] and len(full["stages"]) >= 10 and "stages_one_stream" in full["configs2_large_b16"]
    fc = full["config"]
    assert sum(fc["forwards_in_timed_region"].values()) >= 3 and fc["distinct_chunks_per_forward"] is True
    cz = full["configs4_corpus"]
    assert cz["chunks"] > 2000 and cz["finite"] and cz["notes"] > 1000 and cz["value"] > 500
    cp = full["configs4_corpus_from_pcm"]
    assert cp["finite"] and cp["notes"] > 1000 and cp["value"] > 100 and cp["chunks"] == cz["chunks"]


def test_bench_two_ranks_rehearsal(tmp_path):
    """The driver's multi-GPU command line (torch.distributed.run, one rank per GPU, barrier + MAX of the elapsed time, rank 0
    prints the line), rehearsed with two ranks folded onto this box's one GPU over gloo (MT_BENCH_BACKEND): `value` counts both
    ranks' chunks."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "2", "--streams", "1", "--cosched", "2",
           "--no-sections", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 9 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 32 * 9 / (d["ms_per_step"] * 9 / 1e3)) / d["value"] < 1e-3
    assert d["config"]["coscheduled_batches_per_forward"] == 2 and d["config"]["streams_per_gpu"] == 1


def test_compressed_audio_takes_the_wav_device_path(mta, tmp_path, monkeypatch):
    """A file that is not a WAV container is decoded on the host (stub decoder: none ships in this image) and then resampled on the
    GPU exactly like the WAV holding the same PCM (channel mean + polyphase resampling, csrc/resample.hip)."""
    import types
    from scipy.io import wavfile
    from music_transcription_amd import transcribe as tr
    w = FR.synth_audio(1, 44100 * 2, seed=9, sr=44100)[0]
    pcm = np.stack([w, 0.25 * w], 1).astype(np.float32)
    wavfile.write(str(tmp_path / "a.wav"), 44100, pcm)
    (tmp_path / "b.mp3").write_bytes(b"ID3\x03" + bytes(128))
    stub = types.ModuleType("soundfile")
    stub.read = lambda path, dtype="float32", always_2d=True: (pcm, 44100)
    monkeypatch.setitem(sys.modules, "soundfile", stub)
    ya = tr.load_audio_device(str(tmp_path / "a.wav"), 16000, "cuda")
    yb = tr.load_audio_device(str(tmp_path / "b.wav"), 16000, "cuda")          # (missing .wav -> the .mp3 beside it)
    assert ya.shape == yb.shape and ya.numel() == 32000 and torch.equal(ya, yb) and float(ya.abs().max()) > 0.01
