#!/usr/bin/env python3
"""bench.py -- 30 s audio chunks/s through the MI355X hot path (mel frontend + CNNRNNModel forward).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of `--batch` synthetic 30 s / 16 kHz chunks
already resident in HBM (BASELINE.json configs[1]: CNNRNNModel (36M) inference, batch = 32):
    waveform (B, 480000) f32 -> mt_mel_db_f32 -> conv1 -> conv2 -> 3 x (bf16 MFMA input projection
    -> persistent fp32 bi-LSTM recurrence -> re-layout) -> fc -> logits (B, 88, 938) f32.
Chunks are independent (main.py:258-266 keeps no cross-chunk state), so N GPUs run N independent
batches with no data-path collective (weak scaling); the only collectives are the timing barrier
and a MAX over ranks of the elapsed time.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the kernel that dominates the step, timed with HIP events on the launch stream
                  inside the timed region (events recorded natively by mt_cnnrnn_forward_ex);
  stages       -- the same for every kernel of the step;
  cpu_baseline -- the CPU oracle (a port of the reference path, oracle/*.py) timed on this node's
                  host cores on a bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# chip peaks from /opt/skills/guides/MI355X_MICROARCH.md (dense; spec)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0
PEAK_F32_MATRIX_TFLOPS = 157.3

N_SAMPLES, SR, HOP, N_MELS, HIDDEN, LAYERS = 480000, 16000, 512, 320, 512, 3


def stage_table(B, T, n_mels, H, L, fused=False):
    """(name, bound, work per launch, unit of work) for each timed stage, in launch order.
    Algorithmic figures: SURVEY 8(d) / DESIGN.md 'Kernels'."""
    F1, Fo2 = n_mels // 2, n_mels // 4
    M = B * T
    st = [("mel_kernel", "hbm", B * (4 * N_SAMPLES + 4 * n_mels * T), "B"),
          ("conv1_kernel", "hbm", B * (4 * n_mels * T + 2 * 32 * F1 * T), "B"),
          ("conv2_kernel", "mfma", 2.0 * B * (2 * Fo2) * T * 64 * 288, "FLOP")]
    for l in range(L):
        K = Fo2 * 64 if l == 0 else 2 * H
        proj = 2.0 * M * 8 * H * K
        in_rec = fused and l > 0                     # layers > 0 project inside the recurrence: their GEMM stage is empty
        st.append((f"gemm_lstm_gx_l{l}", "mfma", 0.0 if in_rec else proj, "FLOP"))
        # algorithmic FLOPs of W_hh h (+ W_ih x when fused); f16 MFMA; the kernel is bound by the per-step
        # inter-workgroup hand-off latency, not by the matrix pipe
        st.append((f"lstm_rec_l{l}", "mfma_f16", 2.0 * M * 8 * H * H + (proj if in_rec else 0.0), "FLOP"))
        # (the re-layout pass is skipped when the next layer reads hx directly)
        st.append((f"lstm_relayout_l{l}", "hbm", 0 if (fused and l + 1 < L) else M * 2 * H * (4 + 2), "B"))
    st.append(("gemm_logits", "mfma", 2.0 * M * 88 * 2 * H, "FLOP"))
    return st


def host_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota and by the
    GPU box's per-GPU CPU share (16).  os.cpu_count() alone oversubscribes a quota'd container."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MT_BENCH_CPU_THREADS", "16"))))


def synth_audio(batch, n_samples=480000, seed=1234, sr=16000):
    """SURVEY 8(d) synthetic input: 0.1*N(0,1) noise + 1-6 decaying sinusoids at piano fundamentals, clipped to [-1, 1]."""
    import numpy as np
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.empty((batch, n_samples), dtype=np.float32)
    for b in range(batch):
        y = 0.1 * rng.standard_normal(n_samples)
        for _ in range(int(rng.integers(1, 7))):
            f0 = 27.5 * 2.0 ** (int(rng.integers(0, 88)) / 12.0)
            onset = float(rng.uniform(0.0, 0.8)) * n_samples / sr
            amp = float(rng.uniform(0.1, 0.6))
            env = np.where(t >= onset, np.exp(-(t - onset) * float(rng.uniform(0.3, 3.0))), 0.0)
            y += amp * env * np.sin(2 * np.pi * f0 * (t - onset))
        out[b] = np.clip(y, -1.0, 1.0).astype(np.float32)
    return out


def seeded_model(mta, model_type, device, seed=0, **kw):
    """Random-init weights of the architecture (there is no checkpoint): torch's default initialisation under a fixed seed,
    BatchNorm running statistics drawn around (0, 1) so that folding them is exercised."""
    import torch
    torch.manual_seed(seed)
    model = mta.TranscriptionModel(model_type, n_mels=N_MELS, hidden_size=HIDDEN, num_layers=LAYERS, device="cpu", **kw)
    g = torch.Generator().manual_seed(seed + 1)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return model.to(device)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def bench_large(args):
    """BASELINE.json configs[2]: CNNRNNModelLarge (89M), batch 16, 1 GPU -- informational (not the bench line)."""
    import numpy as np
    import torch
    import music_transcription_amd as mta
    dev = torch.device("cuda", 0)
    B, K, W, NS = args.batch, args.steps, args.warmup, max(1, args.streams)
    T = mta.num_frames(N_SAMPLES, HOP)
    base = synth_audio(min(B, 4), N_SAMPLES, seed=1234)
    wave = torch.from_numpy(np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B].copy()).to(dev)
    model = seeded_model(mta, "cnn_rnn_large", str(dev)).eval()
    fe = mta.MelFrontend(SR, N_MELS, HOP, dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    mel = [torch.empty(B, 1, N_MELS, T, device=dev) for _ in range(NS)]
    cmax = [torch.empty(B, device=dev) for _ in range(NS)]

    def step(j):
        s = j % NS
        with torch.cuda.stream(streams[s]), torch.no_grad():
            fe(wave, clamp=False, out=mel[s], chunk_max=cmax[s])
            return model.model(mel[s], return_all_heads=True, chunk_max_power=cmax[s])
    for j in range(max(W, NS)):
        step(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(K):
        out = step(j)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    model.model.raise_on_handoff_timeout(B, T)
    flops = 326.47e9 * B * T / 938.0
    print(json.dumps({"metric": "30 s audio chunks/sec (mel+CNNRNNModelLarge forward)", "value": round(B * K / el, 2), "unit": "chunks/s",
                      "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": round(1e3 * el / K, 3), "higher_is_better": True,
                      "data": "synthetic", "config": {"workload": "CNNRNNModelLarge inference, batch=16 (BASELINE.json configs[2])",
                                                      "batch_per_gpu": B, "streams_per_gpu": NS},
                      "model_tflops_per_s": round(flops * K / el / 1e12, 1), "finite": bool(torch.isfinite(out["frame"]).all())}))


def bench_train(args):
    """BASELINE.json configs[3]: chunked training of CNNRNNModel, batch 16 per GPU, data-parallel -- informational (not
    the bench line).  Step = train-mode forward + masked BCE + backward (HIP kernels) + ONE all-reduce (mean) of the flat
    gradient over RCCL + fused clip/Adam.  Inputs: cached-format batches, ragged T in [469, 937] right-padded with 0.0
    (collate_fn semantics), Bernoulli(0.04) rolls; resident in HBM before the timed region."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    backend = os.environ.get("MT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import music_transcription_amd as mta
    B, K, W, T = args.batch, args.steps, args.warmup, 937
    g = torch.Generator().manual_seed(1234 + rank)
    model = seeded_model(mta, "cnn_rnn", str(dev), dropout=0.3)      # the same initial weights on every rank
    opt = mta.make_optimizer(model, lr=1e-4)
    batches = []
    for _ in range(2):
        lengths = torch.randint(469, T + 1, (B,), generator=g)
        lengths[0] = T
        mel = torch.rand(B, 1, N_MELS, T, generator=g) * 60.0 - 70.0
        roll = (torch.rand(B, 88, T, generator=g) < 0.04).float()
        for b in range(B):
            mel[b, :, :, lengths[b]:] = 0.0
            roll[b, :, lengths[b]:] = 0.0
        batches.append((mel.to(dev), roll.to(dev), lengths))
    model.train()

    def step(j):
        mel, roll, lengths = batches[j % len(batches)]
        opt.zero_grad()
        loss = model.compute_loss(model(mel), roll, lengths)
        loss.backward()
        opt.step()                                   # all-reduce (mean) of the flat gradient, then clip + Adam
        return loss
    for j in range(W):
        step(j)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(K):
        loss = step(j)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    if rank == 0:
        flops = 3.0 * 72.76e9 * B * world * T / 938.0
        print(json.dumps({"metric": "30 s audio chunks/sec (CNNRNNModel training step)", "value": round(B * world * K / el, 2),
                          "unit": "chunks/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * el / K, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16 MFMA operands, f32 accumulate / LSTM state / master weights", "data": "synthetic",
                          "config": {"workload": "CNNRNNModel training, batch=16/GPU cached-format chunks, data-parallel "
                                                 "(BASELINE.json configs[3])", "batch_per_gpu": B, "frames": T,
                                     "parallelism": f"dp{world} (one RCCL all-reduce of the flat gradient per step)"},
                          "model_tflops_per_s": round(flops * K / el / 1e12, 1), "final_loss": round(float(loss.item()), 5)}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x 5 ms: long enough that the first and last rounds of the streams in flight do not weigh)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches in flight per GPU (each step is issued whole on stream i %% streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", choices=["cnn_rnn", "cnn_rnn_large"], default="cnn_rnn",
                    help="cnn_rnn = the bench line (BASELINE configs[1]); cnn_rnn_large = informational run of configs[2] "
                         "(use --batch 16): whole-step timing only")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer = the bench line; train = informational run of configs[3] (use --batch 16 --steps 5)")
    args = ap.parse_args()
    if args.mode == "train":
        return bench_train(args)
    if args.model == "cnn_rnn_large":
        return bench_large(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU.  MT_BENCH_BACKEND=gloo (+ ranks folded onto the visible devices) exists only to rehearse
    # the multi-rank control flow on a one-GPU box; the driver's runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("MT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import music_transcription_amd as mta

    B, K, W = args.batch, args.steps, args.warmup
    T = mta.num_frames(N_SAMPLES, HOP)
    # seeded synthetic input: noise + decaying piano-range sinusoids; seed = 1234 + rank.
    # Four distinct chunks tiled to the batch keep host-side synthesis short; the kernels see B chunks.
    base = synth_audio(min(B, 4), N_SAMPLES, seed=1234 + rank)
    wave = torch.from_numpy(np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B].copy()).to(dev)
    model = seeded_model(mta, "cnn_rnn", str(dev))
    model.eval()
    net = model.model
    fe = mta.MelFrontend(SR, N_MELS, HOP, dev)
    NS = max(1, args.streams)
    # Persistent recurrence launches wait on their own workgroups, so every launch in flight must eventually be fully
    # resident.  A plain recurrence workgroup needs a third of a CU's registers (768 slots per GPU = 6 launches of 128),
    # one with the fused projection a whole CU's (256 slots = 2 launches): with at most 3 launches in flight the third
    # always gets the slots the first frees (deficits sum to 128 = one launch); 4+ fused launches could starve each other
    # until the 2 s spin bound trips.  Hence: fusion only for 2-3 streams, never more than 6 streams.
    if NS > 6:
        raise SystemExit("--streams: at most 6 forwards in flight per GPU (co-residency of the persistent recurrence launches)")
    # with several batches in flight the projection GEMMs are the shared resource: let layers 1.. project inside the recurrence
    net.fuse_input_projection = NS == 2
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    mel = [torch.empty(B, 1, N_MELS, T, device=dev) for _ in range(NS)]
    cmax = [torch.empty(B, device=dev) for _ in range(NS)]

    nst = 3 + 3 * LAYERS
    ev_mel = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(K)]
    ev_net = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K)]
    for row in ev_mel + ev_net:          # create the underlying hipEvent_t handles before the timed region
        for e in row:
            e.record()

    def step(j, i=None):
        """One whole pass (mel + forward) over one batch, issued on stream j % NS."""
        s = j % NS
        with torch.cuda.stream(streams[s]), torch.no_grad():
            if i is not None:
                ev_mel[i][0].record()
            fe(wave, clamp=False, out=mel[s], chunk_max=cmax[s])   # unclamped dB + per-chunk max; conv1 clamps on load
            if i is not None:
                ev_mel[i][1].record()
            return net(mel[s], chunk_max_power=cmax[s], events=None if i is None else ev_net[i])

    log(f"rank {rank}/{world}: setup done, B={B} T={T}; warmup {W}, steps {K}")
    for j in range(max(W, NS if W else 0)):
        step(j)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        logits = step(i, i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    net.raise_on_handoff_timeout(B, T)
    log(f"timed region: {elapsed:.3f} s for {K} steps")
    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        # ---- per-kernel times from the events recorded inside the timed region
        table = stage_table(B, T, N_MELS, HIDDEN, LAYERS, fused=bool(net.fuse_input_projection))
        ms = [float(np.mean([ev_mel[i][0].elapsed_time(ev_mel[i][1]) for i in range(K)]))]
        for s in range(nst):
            ms.append(float(np.mean([ev_net[i][s].elapsed_time(ev_net[i][s + 1]) for i in range(K)])))
        stages = []
        for (name, bound, work, unit), t_ms in zip(table, ms):
            if unit == "B":
                ach, peak, u = work / (max(t_ms, 1e-6) * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            else:
                ach = work / (max(t_ms, 1e-6) * 1e-3) / 1e12
                peak, u = (PEAK_F32_MATRIX_TFLOPS if bound == "mfma_f32" else PEAK_BF16_TFLOPS), "TFLOP/s"    # f16 peak = bf16 peak
            stages.append({"kernel": name, "bound": "hbm" if bound == "hbm" else "mfma", "ms": round(t_ms, 4),
                           "achieved": round(ach, 2), "peak": peak, "unit": u, "frac": round(ach / peak, 4),
                           "work_per_launch": work, "work_unit": unit,
                           "mfma_dtype": {"mfma": "bf16", "mfma_f32": "f32", "mfma_f16": "f16"}.get(bound)})
        # dominant kernel = largest share of the step; the three recurrence launches are one kernel
        groups = {}
        for s in stages:
            key = s["kernel"].rsplit("_l", 1)[0] if "_l" in s["kernel"] else s["kernel"]
            g = groups.setdefault(key, {"ms": 0.0, "work": 0.0, "n": 0, "ref": s})
            g["ms"] += s["ms"]; g["work"] += s["work_per_launch"]; g["n"] += 1
        dom_key = max(groups, key=lambda k: groups[k]["ms"])
        g = groups[dom_key]
        avg_ms, avg_work = g["ms"] / g["n"], g["work"] / g["n"]
        ref = g["ref"]
        ach = avg_work / (avg_ms * 1e-3) / (1e9 if ref["work_unit"] == "B" else 1e12)
        # HBM bytes per launch from the committed PMC profile of this same command (separate --pmc passes,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): profiles/r01_pmc_traffic.json
        traffic = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            hit = [v for k, v in prof.items() if dom_key.replace("_l", "").split("_gx")[0] in k.replace("::", "_")]
            if dom_key == "lstm_rec":
                hit = [v for k, v in prof.items() if "lstm_rec_kernel" in k]
            if hit and B == 32:      # launch-weighted mean over the kernel's variants (plain / fused-projection recurrence)
                traffic = round(sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit))
        except Exception:
            traffic = None
        roofline = {"kernel": dom_key, "bound": ref["bound"], "achieved": round(ach, 2), "peak": ref["peak"],
                    "unit": ref["unit"], "frac": round(ach / ref["peak"], 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": g["n"],
                    "share_of_step": round(g["ms"] / sum(ms), 3), "mfma_dtype": ref["mfma_dtype"]}

        # ---- CPU baseline: the oracle (port of the reference path) on this node's host cores
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import frontend_ref, model_ref      # the CPU port of the reference path: used ONLY in this leg
            cores = host_cores()
            log(f"cpu baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
            torch.set_num_threads(cores)
            w_np = wave[:1].cpu().numpy()
            sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
            def cpu_chunk():
                m = frontend_ref.audio_to_mel_batch(w_np, SR, N_MELS, HOP)          # main.py:117-125
                with torch.no_grad():
                    return model_ref.cnnrnn_forward(sd, torch.from_numpy(m), model_ref.Opts(fast_lstm=True))
            t1 = time.perf_counter(); ref_logits = cpu_chunk(); first = time.perf_counter() - t1
            log(f"cpu baseline: first chunk {first:.2f} s")
            n = int(max(2, min(24, 15.0 / max(first, 1e-3))))
            t1 = time.perf_counter()
            for _ in range(n):
                cpu_chunk()
            dt = time.perf_counter() - t1
            cpu = {"value": round(n / dt, 3), "unit": "chunks/s", "cores": cores, "kind": "port",
                   "sample": f"{n} chunks, batch 1 (the reference's main.py:258 loop): numpy STFT/mel/dB + fp32 torch-CPU "
                             f"CNNRNNModel forward (oracle/frontend_ref.py + oracle/model_ref.py), {cores} threads"}
            # the timed GPU logits for chunk 0 must agree with the oracle (same weights, same audio)
            err = float((logits[0].cpu() - ref_logits[0]).abs().max())
            cpu["max_abs_logit_diff_vs_gpu"] = round(err, 5)

        out = {"metric": "30 s audio chunks/sec (mel+CNNRNN forward)", "value": round(world * B * K / elapsed, 2),
               "unit": "chunks/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16 MFMA (f32 accumulate) for conv2 / layer-0 projection / fc, f16 MFMA for the recurrence and the "
                        "projections fused into it (f32 state, gates and accumulation), f32 FFT and conv1",
               "data": "synthetic",
               "config": {"workload": "CNNRNNModel inference, batch=32x30 s synthetic 16 kHz audio, mel+CNN-RNN HIP path "
                                      "(BASELINE.json configs[1])", "batch_per_gpu": B, "n_samples": N_SAMPLES,
                          "n_mels": N_MELS, "hidden": HIDDEN, "layers": LAYERS, "frames": T, "parallelism": f"dp{world} (independent chunks)", "streams_per_gpu": NS, "fused_input_projection": bool(net.fuse_input_projection)},
               "roofline": roofline, "cpu_baseline": cpu, "stages": stages}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
