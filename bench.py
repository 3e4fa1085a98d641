#!/usr/bin/env python3
"""bench.py -- 30 s audio chunks/s through the MI355X hot path (mel frontend + CNNRNNModel forward).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--cosched 4] [--streams 4] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of `--batch` synthetic 30 s / 16 kHz chunks
already resident in HBM (BASELINE.json configs[1]: CNNRNNModel (36M) inference, batch = 32):
    waveform (B, 480000) f32 -> mt_mel_db_f32 -> conv1 -> conv2 -> 3 x (f16 MFMA input projection
    -> persistent bi-LSTM recurrence, f32 state) -> fc -> logits (B, 88, 938) f32.
The K timed steps are issued `--cosched` at a time as one forward (the recurrence interleaves the batch
groups of the co-scheduled batches inside one persistent launch), `--streams` forwards in flight.
Chunks are independent (main.py:258-266 keeps no cross-chunk state), so N GPUs run N independent
schedules with no data-path collective (weak scaling); the only collectives are the timing barrier
and a MAX over ranks of the elapsed time.

Rank 0 prints ONE JSON line of at most 4 KB as the LAST line of stdout (compact_line): the contract fields,
  roofline     -- the kernel that dominates the step, timed with HIP events on the launch stream (recorded
                  natively by mt_cnnrnn_forward_ex) in an un-overlapped pass of this run: ONE forward of the
                  timed region's shape in flight; roofline.all = {stage: fraction of its roofline};
  cpu_baseline -- the CPU oracle (a port of the reference path, oracle/*.py) timed on this node's
                  host cores on a bounded sample of the same workload, rank 0, N = 1 only;
  configs1_literal_b32 -- BASELINE configs[1] as written: ONE batch of 32 chunks per forward;
  sections     -- one short object per neighbouring BASELINE config (Large inference, both training steps, corpus).
Everything else (stage tables of every pass, schedules, autotune lists, samples) goes to bench_detail.json
(repo root and gpurun_out/; MT_BENCH_DETAIL overrides the path).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise); two streams
# on one queue run their kernels one after the other.  The default schedule keeps 4 forwards in flight on 4 streams next to
# torch's own stream, so the bench asks for 8 queues (must be in the environment before the runtime initialises).  Measured:
# 8 600 chunks/s with 4 queues, 10 500 with 8, same kernels.
_USER_SET_QUEUES = "GPU_MAX_HW_QUEUES" in os.environ
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# chip peaks from /opt/skills/guides/MI355X_MICROARCH.md (dense; spec)
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0
PEAK_F32_MATRIX_TFLOPS = 157.3

N_SAMPLES, SR, HOP, N_MELS, HIDDEN, LAYERS = 480000, 16000, 512, 320, 512, 3


def stage_table(B, T, n_mels, H, L, fused=False, conv_fused=None):
    """(name, bound, work per launch, unit of work) for each timed stage, in launch order.
    Algorithmic figures: SURVEY 8(d) / DESIGN.md 'Kernels'."""
    F1, Fo2 = n_mels // 2, n_mels // 4
    M = B * T
    if conv_fused is None:
        from music_transcription_amd._lib import lib
        conv_fused = bool(lib.mt_cnnrnn_conv_fused())
    st = [("mel_kernel", "hbm", B * (4 * N_SAMPLES + 4 * n_mels * T), "B"),
          # conv1 + conv2 as one kernel (conv12_kernel: act1 never exists in HBM): the conv1 stage is empty, the fused kernel is charged
          # conv2's matrix-pipe work (conv1's 576 multiply-adds per position run on the vector ALU beside it)
          ("conv1_kernel", "hbm", 0 if conv_fused else B * (4 * n_mels * T + 2 * 32 * F1 * T), "B"),
          ("conv12_kernel" if conv_fused else "conv2_kernel", "mfma", 2.0 * B * (2 * Fo2) * T * 64 * 288, "FLOP")]        # "mfma": f16 operands (inference)
    for l in range(L):
        K = Fo2 * 64 if l == 0 else 2 * H
        proj = 2.0 * M * 8 * H * K
        in_rec = fused and l > 0                     # layers > 0 project inside the recurrence: their GEMM stage is empty
        st.append((f"gemm_lstm_gx_l{l}", "mfma", 0.0 if in_rec else proj, "FLOP"))
        # algorithmic FLOPs of W_hh h (+ W_ih x when fused); f16 MFMA; the kernel is bound by the per-step
        # inter-workgroup hand-off latency, not by the matrix pipe
        st.append((f"lstm_rec_l{l}", "mfma_f16", 2.0 * M * 8 * H * H + (proj if in_rec else 0.0), "FLOP"))
        # (no re-layout pass: with f16 operands the consuming GEMMs read their A tiles straight from the hx images; the stage is empty)
        st.append((f"lstm_relayout_l{l}", "hbm", 0, "B"))
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
    B, K, W, NS = args.batch, args.steps, args.warmup, max(1, min(args.streams, 3))   # (two recurrence launches per forward: at most 3 forwards in flight)
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
    mtype = args.model
    model = seeded_model(mta, mtype, str(dev), dropout=0.3 if mtype == "cnn_rnn" else 0.2)      # the same initial weights on every rank
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
    # the side streams the step runs fastest with (every rank the same number of steps)
    from music_transcription_amd.train_step_large import autotune_side_streams
    autotune_side_streams(lambda: step(0), dev, candidates=4, steps=1)
    # untimed steps on the chosen streams, queued back to back as the timed region queues them (the host runs a step ahead of the device, so
    # blocks that a side stream still holds are not back yet when the next step asks: the allocator's pools must have grown to THAT pattern
    # -- a device allocation inside the timed region costs up to 80 ms)
    for _ in range(2):
        for j in range(3):
            step(j)
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    segs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    host = []
    for j in range(K):
        th = time.perf_counter()
        loss = step(j)
        host.append(time.perf_counter() - th)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    log(f"train: host enqueue {1e3 * t_host / K:.2f} ms per step ({', '.join(f'{1e3 * v:.1f}' for v in host)}), device allocations inside the timed region: "
        f"{torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - segs0}, reserved {torch.cuda.memory_reserved(dev) / 2 ** 30:.1f} GiB")
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    if rank == 0:
        flops = 3.0 * (72.76e9 if mtype == "cnn_rnn" else 326.47e9) * B * world * T / 938.0
        print(json.dumps({"metric": f"30 s audio chunks/sec ({'CNNRNNModel' if mtype == 'cnn_rnn' else 'CNNRNNModelLarge'} training step)", "value": round(B * world * K / el, 2),
                          "unit": "chunks/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * el / K, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16 MFMA operands, f32 accumulate / LSTM state / master weights", "data": "synthetic",
                          "config": {"workload": (f"CNNRNNModel training, batch={B}/GPU cached-format chunks, data-parallel (BASELINE.json configs[3] is batch=16/GPU)"
                                                  if mtype == "cnn_rnn" else f"CNNRNNModelLarge training (what example.sh:22 trains), batch={B}/GPU cached-format chunks, data-parallel"),
                                     "batch_per_gpu": B, "frames": T,
                                     "parallelism": f"dp{world} (one RCCL all-reduce of the flat gradient per step)"},
                          "model_tflops_per_s": round(flops * K / el / 1e12, 1), "final_loss": round(float(loss.item()), 5)}))
    if world > 1:
        dist.destroy_process_group()


def _roofline_from_stages(stages, share_of=None):
    """Dominant kernel = largest total time; launches of one kernel (its per-layer stages) are averaged."""
    groups = {}
    for s in stages:
        key = s["kernel"].rsplit("_l", 1)[0] if ("_l" in s["kernel"] and s["kernel"].rsplit("_l", 1)[1].isdigit()) else s["kernel"]
        g = groups.setdefault(key, {"ms": 0.0, "work": 0.0, "n": 0, "ref": s})
        if s["ms"] > 0 and s["work_per_launch"] > 0:
            g["ms"] += s["ms"]; g["work"] += s["work_per_launch"]; g["n"] += 1
    groups = {k: g for k, g in groups.items() if g["n"]}
    dom_key = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[dom_key]
    avg_ms, avg_work = g["ms"] / g["n"], g["work"] / g["n"]
    ref = g["ref"]
    ach = avg_work / (avg_ms * 1e-3) / (1e9 if ref["work_unit"] == "B" else 1e12)
    total = share_of if share_of else sum(s["ms"] for s in stages)
    return dom_key, {"kernel": dom_key, "bound": ref["bound"], "achieved": round(ach, 2), "peak": ref["peak"], "unit": ref["unit"],
                     "frac": round(ach / ref["peak"], 4), "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": g["n"],
                     "share_of_step": round(g["ms"] / total, 3), "mfma_dtype": ref["mfma_dtype"]}


# dominant-stage key of a section -> substring of the kernel's name in the committed PMC profile of that section's command
_PMC_KERNEL_OF = {"lstm_bptt": "lstm_bptt_kernel", "lstm_rec_train": "lstm_rec_kernel", "lstm_rec": "lstm_rec_kernel", "gemm_lstm_gx": "gemm256",
                  "convg_freq_aware_7x3": "convg_kernel", "convg_res_block1": "convg_kernel", "convg_res_block2": "convg_kernel",
                  "attention_layernorm": "attn", "conv1_kernel": "conv1_kernel", "heads": "gemm"}


def _pmc_traffic(fnames, dom_key):
    """HBM bytes per launch of the section's dominant kernel from a committed rocprofv3 --pmc summary (tools/refresh_pmc.sh):
    PMC counters cannot be read from inside the process.  Where one kernel NAME covers several stages (convg_kernel), the
    launches of that name with the most bytes are taken (the 7x3 conv is the largest convg launch)."""
    sub = _PMC_KERNEL_OF.get(dom_key, dom_key)
    for fname in fnames:
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", fname)))["kernels"]
        except Exception:
            continue
        hit = [v for k, v in prof.items() if sub in k]
        if hit:
            best = max(hit, key=lambda v: v["hbm_bytes_per_launch_corrected"])
            return round(best["hbm_bytes_per_launch_corrected"]), f"profiles/{fname} (committed rocprofv3 --pmc passes of this section's command; not re-measured in this run)"
    return None, None


def _stage_rows(table, ms):
    stages = []
    for (name, bound, work, unit), t_ms in zip(table, ms):
        if unit == "B":
            ach, peak, u = work / (max(t_ms, 1e-6) * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        else:
            ach = work / (max(t_ms, 1e-6) * 1e-3) / 1e12
            peak, u = PEAK_BF16_TFLOPS, "TFLOP/s"                         # f16 peak = bf16 peak
        stages.append({"kernel": name, "bound": "hbm" if bound == "hbm" else "mfma", "ms": round(t_ms, 4), "achieved": round(ach, 2),
                       "peak": peak, "unit": u, "frac": round(ach / peak, 4), "work_per_launch": work, "work_unit": unit,
                       "mfma_dtype": {"mfma": "f16", "mfma_bf16": "bf16", "mfma_f16": "f16"}.get(bound)})
    return stages


def large_stage_table(B, T, n_mels, H, L):
    """Stages of mt_cnnrnn_large_forward_ev in order, with their algorithmic work (SURVEY 8d / BASELINE.md section 4)."""
    F1, F2, F3 = n_mels // 2, n_mels // 4, n_mels // 8
    N1, N2, M, Hl = B * F1 * T, B * F2 * T, B * T, H // 2
    K0, comb, heads = F3 * 256, 2 * H + 2 * (H // 2), 8
    dh = comb // heads
    st = [("conv1_kernel", "hbm", B * (4 * n_mels * T + 2 * 32 * F1 * T), "B"),
          ("convg_res_block1", "mfma", 2.0 * N1 * 64 * (9 * 32) + 2.0 * N1 * 64 * (9 * 64 + 32), "FLOP"),
          ("convg_res_block2", "mfma", 2.0 * N2 * 128 * (9 * 64) + 2.0 * N2 * 128 * (9 * 128 + 64), "FLOP"),
          ("convg_freq_aware_7x3", "mfma", 2.0 * N2 * 256 * (21 * 128), "FLOP"),
          ("gemm_lstm_gx_local", "mfma", 2.0 * M * 8 * Hl * K0, "FLOP"),
          ("lstm_rec_local", "mfma_f16", 2.0 * M * 8 * Hl * Hl, "FLOP"),
          ("lstm_relayout_local", "hbm", M * 2 * Hl * (2 + 2 + 4), "B")]
    for l in range(L):
        K = K0 if l == 0 else 2 * H
        st += [(f"gemm_lstm_gx_l{l}", "mfma", 2.0 * M * 8 * H * K, "FLOP"), (f"lstm_rec_l{l}", "mfma_f16", 2.0 * M * 8 * H * H, "FLOP"),
               (f"lstm_relayout_l{l}", "hbm", M * 2 * H * (2 + 2 + 4) if l == L - 1 else 0, "B")]      # (layers below the top: the next GEMM reads hx directly)
    st += [("attention_layernorm", "mfma", 2.0 * M * comb * 3 * comb + 2 * 2.0 * B * heads * T * T * dh + 2.0 * M * comb * comb, "FLOP"),
           ("heads", "mfma", 2.0 * M * comb * H + 2.0 * M * H * 264, "FLOP")]
    return st


def cpu_baseline_small(model, wave, logits, cores):
    """The CPU oracle (port of the reference path) on this node's host cores: batch 1 (the reference's main.py:258 loop) on a
    bounded sample, and batch 8 on one batch."""
    import numpy as np
    import torch
    from oracle import frontend_ref, model_ref      # the CPU port of the reference path: used ONLY in this leg
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    w1 = wave[:1].cpu().numpy()

    def run(w_np):
        m = frontend_ref.audio_to_mel_batch(w_np, SR, N_MELS, HOP)          # main.py:117-125
        with torch.no_grad():
            return model_ref.cnnrnn_forward(sd, torch.from_numpy(m), model_ref.Opts(fast_lstm=True))
    t1 = time.perf_counter(); ref_logits = run(w1); first = time.perf_counter() - t1
    log(f"cpu baseline: first chunk {first:.2f} s on {cores} threads")
    n = int(max(2, min(16, 8.0 / max(first, 1e-3))))
    t1 = time.perf_counter()
    for _ in range(n):
        run(w1)
    dt = time.perf_counter() - t1
    cpu = {"value": round(n / dt, 3), "unit": "chunks/s", "cores": cores, "kind": "port",
           "sample": f"{n} chunks, batch 1 (the reference's main.py:258 loop): numpy STFT/mel/dB + fp32 torch-CPU CNNRNNModel forward "
                     f"(oracle/frontend_ref.py + oracle/model_ref.py), {cores} threads",
           "max_abs_logit_diff_vs_gpu": round(float((logits[0].cpu() - ref_logits[0]).abs().max()), 5)}
    w8 = wave[:8].cpu().numpy()
    t1 = time.perf_counter(); run(w8); dt8 = time.perf_counter() - t1
    cpu["batch8"] = {"value": round(8 / dt8, 3), "unit": "chunks/s", "sample": "1 batch of 8 chunks, same path"}
    return cpu


def section_coscheduled(mta, dev, net, fe, wave32):
    """The same workload with several batches of 32 chunks CO-SCHEDULED in one forward (B = 64 / 96 / 128): the recurrence
    interleaves the batch groups inside one persistent launch (csrc/lstm.hip, NG) -- while one group's h travels to its consumers
    the workgroup computes the others' steps -- and the GEMMs see M = B T rows.  Informational: the headline keeps BASELINE's
    batch = 32 per forward.  MT_BENCH_COSCHED = "96x1,96x2" selects (chunks per forward) x (streams)."""
    import torch
    T = mta.num_frames(N_SAMPLES, HOP)
    out = {}
    for combo in os.environ.get("MT_BENCH_COSCHED", "32x3,32x4,96x4,128x1").split(","):
        B, NS = (int(v) for v in combo.split("x"))
        K = int(os.environ.get("MT_BENCH_COSCHED_FORWARDS", max(6, 1152 // B)))
        wave = torch.cat([wave32] * (B // 32))
        streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
        mel = [torch.empty(B, 1, N_MELS, T, device=dev) for _ in range(NS)]
        cmax = [torch.empty(B, device=dev) for _ in range(NS)]

        def step(j):
            s = j % NS
            with torch.cuda.stream(streams[s]), torch.no_grad():
                fe(wave, clamp=False, out=mel[s], chunk_max=cmax[s])
                return net(mel[s], chunk_max_power=cmax[s])
        try:
            for j in range(NS + 1):
                step(j)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for j in range(K):
                lg = step(j)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            net.raise_on_handoff_timeout(B, T)
            out[f"b{B}_streams_{NS}"] = {"value": round(B * K / el, 2), "unit": "chunks/s", "ms_per_forward": round(1e3 * el / K, 3),
                                         "finite": bool(torch.isfinite(lg).all())}
        except Exception as e:                      # (e.g. the library's residency check refusing that many launches in flight)
            torch.cuda.synchronize()
            out[f"b{B}_streams_{NS}"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        del mel, cmax
        net._ws.clear()
    out["workload"] = "CNNRNNModel inference, several batches of 32 chunks co-scheduled in one forward"
    return out


def section_large(mta, dev, cores, do_cpu):
    """BASELINE.json configs[2]: CNNRNNModelLarge (89M), batch 16, 1 GPU: throughput with 3 batches in flight, an un-overlapped
    1-stream pass for per-kernel efficiencies, the CPU oracle beside it."""
    import numpy as np
    import torch
    B, NS, K, K1 = 16, 3, 30, 4
    T = mta.num_frames(N_SAMPLES, HOP)
    base = synth_audio(4, N_SAMPLES, seed=1234)
    wave = torch.from_numpy(np.concatenate([base] * 4)[:B].copy()).to(dev)
    model = seeded_model(mta, "cnn_rnn_large", str(dev)).eval()
    net = model.model
    fe = mta.MelFrontend(SR, N_MELS, HOP, dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    mel = [torch.empty(B, 1, N_MELS, T, device=dev) for _ in range(NS)]
    cmax = [torch.empty(B, device=dev) for _ in range(NS)]

    def step(j, events=None, s=None):
        s = j % NS if s is None else s
        with torch.cuda.stream(streams[s]), torch.no_grad():
            fe(wave, clamp=False, out=mel[s], chunk_max=cmax[s])
            return net(mel[s], chunk_max_power=cmax[s], events=events)
    for j in range(NS + 1):
        step(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(K):
        out = step(j)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    net.raise_on_handoff_timeout(B, T)
    nst = lib_stages = None
    from music_transcription_amd._lib import lib
    nst = lib.mt_cnnrnn_large_num_stages(LAYERS)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K1)]
    with torch.cuda.stream(streams[0]):
        for row in evs:
            for e in row:
                e.record()
    torch.cuda.synchronize()
    for i in range(K1):
        step(i, events=evs[i], s=0)
    torch.cuda.synchronize()
    ms = [float(np.mean([evs[i][k].elapsed_time(evs[i][k + 1]) for i in range(K1)])) for k in range(nst)]
    stages = _stage_rows(large_stage_table(B, T, N_MELS, HIDDEN, LAYERS), ms)
    dom, roof = _roofline_from_stages(stages)
    roof["traffic"], roof["traffic_source"] = _pmc_traffic(("r04_pmc_traffic_large.json", "r03_pmc_traffic_large.json"), dom)
    # the same with 8 batches of 16 per forward (four batch groups of 32 interleaved in each persistent recurrence launch)
    cos = None
    try:
        C = 8
        wave_c = torch.cat([wave] * C)
        mel_c = [torch.empty(C * B, 1, N_MELS, T, device=dev) for _ in range(NS)]
        cmax_c = [torch.empty(C * B, device=dev) for _ in range(NS)]

        def step_c(j):
            s = j % NS
            with torch.cuda.stream(streams[s]), torch.no_grad():
                fe(wave_c, clamp=False, out=mel_c[s], chunk_max=cmax_c[s])
                return net(mel_c[s], chunk_max_power=cmax_c[s])
        for j in range(NS + 1):
            step_c(j)
        torch.cuda.synchronize()
        KC = 9
        t0 = time.perf_counter()
        for j in range(KC):
            oc = step_c(j)
        torch.cuda.synchronize()
        elc = time.perf_counter() - t0
        net.raise_on_handoff_timeout(C * B, T)
        cos = {"value": round(C * B * KC / elc, 2), "unit": "chunks/s", "ms_per_step": round(1e3 * elc / (KC * C), 3),
               "coscheduled_batches_per_forward": C, "streams_per_gpu": NS, "finite": bool(torch.isfinite(oc).all())}
        del mel_c, cmax_c
    except Exception as e:
        torch.cuda.synchronize()
        cos = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    one = {"value": round(B * K / el, 2), "unit": "chunks/s", "ms_per_step": round(1e3 * el / K, 3), "steps": K, "streams_per_gpu": NS}
    best = cos if (cos and "value" in cos and cos["value"] > one["value"]) else None
    sec = {"workload": "CNNRNNModelLarge inference, batch=16 (BASELINE.json configs[2]); a step = one batch of 16 chunks",
           "value": best["value"] if best else one["value"], "unit": "chunks/s",
           "ms_per_step": best["ms_per_step"] if best else one["ms_per_step"],
           "scheduling": (f"{best['coscheduled_batches_per_forward']} steps per forward (their batch groups of 32 interleaved in each persistent recurrence "
                          f"launch), {NS} forwards in flight") if best else f"one step per forward, {NS} forwards in flight",
           "one_batch_per_forward": one, "coscheduled": cos,
           "dtype": "f16 MFMA operands, f32 accumulate / LSTM state",
           "model_tflops_per_s": round(326.47e9 * T / 938.0 * (best["value"] if best else one["value"]) / 1e12, 1), "one_stream_ms_per_step": round(sum(ms), 3),
           "roofline": roof, "stages_one_stream": stages, "finite": bool(torch.isfinite(out).all())}
    if do_cpu:
        from oracle import frontend_ref, model_ref
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        w1 = wave[:1].cpu().numpy()
        n = 2
        t1 = time.perf_counter()
        for _ in range(n):
            m = frontend_ref.audio_to_mel_batch(w1, SR, N_MELS, HOP)
            with torch.no_grad():
                ref = model_ref.cnnrnn_large_forward(sd, torch.from_numpy(m), o=model_ref.Opts(fast_lstm=True))
        dt = time.perf_counter() - t1
        with torch.no_grad():
            got = step(0, s=0)
        torch.cuda.synchronize()
        sec["cpu_baseline"] = {"value": round(n / dt, 3), "unit": "chunks/s", "cores": cores, "kind": "port",
                               "sample": f"{n} chunks, batch 1, numpy frontend + fp32 torch-CPU CNNRNNModelLarge forward (oracle), {cores} threads",
                               "max_abs_logit_diff_vs_gpu": round(float((got[0].cpu() - ref[0]).abs().max()), 5)}
    return sec


def section_train(mta, dev, cores, do_cpu):
    """BASELINE.json configs[3] on one GPU: CNNRNNModel training step, batch 16 cached-format chunks (ragged T in [469, 937],
    Bernoulli(0.04) rolls): train-mode forward + masked BCE + backward + fused clip/Adam.  (The 8-GPU form adds one all-reduce of the
    flat gradient per step: `bench.py --mode train` under torchrun.)"""
    import numpy as np
    import torch
    B, K, W, T = 16, 6, 2, 937
    g = torch.Generator().manual_seed(1234)
    model = seeded_model(mta, "cnn_rnn", str(dev), dropout=0.3)
    opt = mta.make_optimizer(model, lr=1e-4)
    lengths = torch.randint(469, T + 1, (B,), generator=g)
    lengths[0] = T
    mel = torch.rand(B, 1, N_MELS, T, generator=g) * 60.0 - 70.0
    roll = (torch.rand(B, 88, T, generator=g) < 0.04).float()
    for b in range(B):
        mel[b, :, :, lengths[b]:] = 0.0
        roll[b, :, lengths[b]:] = 0.0
    meld, rolld = mel.to(dev), roll.to(dev)
    model.train()

    def step():
        opt.zero_grad()
        loss = model.compute_loss(model(meld), rolld, lengths)
        loss.backward()
        opt.step()
        return loss
    for _ in range(W):
        step()
    from music_transcription_amd.train_step_large import autotune_side_streams
    tuned = autotune_side_streams(step, dev, candidates=4, steps=2)     # (see section_train_large)
    for _ in range(2):                                 # (back to back, as the timed region queues them: see bench_train)
        for _j in range(3):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    model.model.raise_on_train_handoff_timeout()
    # per-launch times of the persistent recurrences and the projection GEMMs (events on the launch stream)
    model.model._profile = []
    step()
    torch.cuda.synchronize()
    spans = {}
    for name, a, b in model.model._profile:
        spans.setdefault(name, []).append(a.elapsed_time(b))
    model.model._profile = None
    M = B * T
    table, ms = [], []
    for l in range(LAYERS):
        Kl = (N_MELS // 4) * 64 if l == 0 else 2 * HIDDEN
        for nm_, bound, work in ((f"gemm_lstm_gx_l{l}", "mfma_bf16", 2.0 * M * 8 * HIDDEN * Kl), (f"lstm_rec_train_l{l}", "mfma_f16", 2.0 * M * 8 * HIDDEN * HIDDEN),
                                 (f"lstm_bptt_l{l}", "mfma_bf16", 2.0 * M * 8 * HIDDEN * HIDDEN)):
            if nm_ in spans:
                table.append((nm_, bound, work, "FLOP")); ms.append(float(np.mean(spans[nm_])))
    stages = _stage_rows(table, ms)
    dom, roof = _roofline_from_stages(stages, share_of=1e3 * el / K)
    roof["traffic"], roof["traffic_source"] = _pmc_traffic(("r04_pmc_traffic_train.json", "r03_pmc_traffic_train.json"), dom)
    sec = {"workload": "CNNRNNModel training step, batch=16 cached-format chunks, 1 GPU (BASELINE.json configs[3] per-GPU shape)",
           "value": round(B * K / el, 2), "unit": "chunks/s", "ms_per_step": round(1e3 * el / K, 3), "steps": K,
           "dtype": "bf16 MFMA operands, f32 accumulate / LSTM state / master weights", "model_tflops_per_s": round(3.0 * 72.76e9 * B * T / 938.0 * K / el / 1e12, 1),
           "roofline": roof, "stages_timed": stages, "final_loss": round(float(loss.item()), 5),
           "side_stream_autotune_ms_per_step": [round(1e3 * t, 2) for t in tuned]}
    if do_cpu:
        from oracle import model_ref
        torch.set_num_threads(cores)
        sd = {k: (v.detach().float() if v.dtype.is_floating_point else v.detach()).cpu().clone() for k, v in model.state_dict().items()}
        nb = 2
        batch = [(mel[:nb, :, :, :].clone(), roll[:nb].clone(), lengths[:nb].clone())]
        t1 = time.perf_counter()
        model_ref.train_steps(sd, batch, o=model_ref.Opts(fast_lstm=True))
        dt = time.perf_counter() - t1
        sec["cpu_baseline"] = {"value": round(nb / dt, 3), "unit": "chunks/s", "cores": cores, "kind": "port",
                               "sample": f"1 step on {nb} chunks: fp32 torch-CPU train-mode forward + autograd backward + clip + Adam (oracle.train_steps), {cores} threads"}
    return sec


def section_train_large(mta, dev, cores, do_cpu):
    """CNNRNNModelLarge training step (what the reference's example.sh:22 trains), batch 16 cached-format chunks, 1 GPU: train-mode
    forward (batch-statistic BatchNorm, Dropout2d, dual LSTM, clamped attention with probability dropout, heads) + masked BCE +
    backward + fused clip/Adam, with the step's stream plan (LSTM weight gradients beside the next backward recurrence, the local
    layer's chain beside the main stack).  The roofline object is the WHOLE step against the bf16 matrix peak (3 x forward FLOPs)."""
    import torch
    B, K, W, T = 16, 8, 2, 937
    g = torch.Generator().manual_seed(1234)
    model = seeded_model(mta, "cnn_rnn_large", str(dev), dropout=0.2)
    opt = mta.make_optimizer(model, lr=1e-4)
    lengths = torch.randint(469, T + 1, (B,), generator=g)
    lengths[0] = T
    mel = torch.rand(B, 1, N_MELS, T, generator=g) * 60.0 - 70.0
    roll = (torch.rand(B, 88, T, generator=g) < 0.04).float()
    for b in range(B):
        mel[b, :, :, lengths[b]:] = 0.0
        roll[b, :, lengths[b]:] = 0.0
    meld, rolld = mel.to(dev), roll.to(dev)
    model.train()

    def step():
        opt.zero_grad()
        loss = model.compute_loss(model(meld), rolld, lengths)
        loss.backward()
        opt.step()
        return loss
    for _ in range(W):
        step()
    # the step's two side streams: the pair it runs fastest with (train_step_large.autotune_side_streams: which streams of torch's pool
    # they are decides whether their work overlaps the calling stream's; by now this process has used a dozen streams)
    from music_transcription_amd.train_step_large import autotune_side_streams
    tuned = autotune_side_streams(step, dev, candidates=4, steps=1)
    for _ in range(2):                                 # (untimed steps on the chosen streams, queued back to back as the timed region queues them: the
        for _j in range(3):                            #  allocator's per-stream pools settle to the pattern of a host that runs a step ahead -- a device
            step()                                     #  allocation inside the timed region costs up to 80 ms)
        torch.cuda.synchronize()
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    marks = []
    for _ in range(K):
        loss = step()
        e_ = torch.cuda.Event(enable_timing=True)
        e_.record()
        marks.append(e_)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per_step = [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(K - 1)]
    log(f"train_large_b16: steps (ms, event to event) {per_step}, device allocations in the timed region "
        f"{torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - allocs0}")
    model.model.raise_on_train_handoff_timeout()
    tf = 3.0 * 326.47e9 * B * T / 938.0 * K / el / 1e12
    sec = {"workload": "CNNRNNModelLarge training step (example.sh:22's model), batch=16 cached-format chunks, 1 GPU",
           "value": round(B * K / el, 2), "unit": "chunks/s", "ms_per_step": round(1e3 * el / K, 3), "steps": K,
           "dtype": "bf16 MFMA operands, f32 accumulate / LSTM state / master weights", "model_tflops_per_s": round(tf, 1),
           "roofline": {"kernel": "whole training step (3 x forward FLOPs)", "bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "traffic": None,
                        "per_kernel": "profiles/r04_train_large_b16_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --mode train --model cnn_rnn_large`)"},
           "final_loss": round(float(loss.item()), 5), "steps_ms_event_to_event": per_step,
           "side_stream_autotune_ms_per_step": [round(1e3 * t, 2) for t in tuned]}
    if do_cpu:
        from oracle import model_ref
        torch.set_num_threads(cores)
        sd = {k: (v.detach().float() if v.dtype.is_floating_point else v.detach()).cpu().clone() for k, v in model.state_dict().items()}
        batch = [(mel[:1].clone(), roll[:1].clone(), lengths[:1].clone())]
        t1 = time.perf_counter()
        model_ref.train_steps(sd, batch, o=model_ref.Opts(fast_lstm=True), model_type="cnn_rnn_large")
        dt = time.perf_counter() - t1
        sec["cpu_baseline"] = {"value": round(1 / dt, 3), "unit": "chunks/s", "cores": cores, "kind": "port",
                               "sample": f"1 step on 1 chunk: fp32 torch-CPU train-mode forward + autograd backward + clip + Adam (oracle.train_steps), {cores} threads"}
    return sec


def section_corpus(mta, dev, cores, do_cpu):
    """BASELINE.json configs[4] on one GPU: the whole synthetic "MAESTRO test split" (177 recordings, 20 h, gamma-distributed
    lengths) through music_transcription_amd.corpus.transcribe_shard -- the code path of scripts/transcribe_corpus.py --
    with CNNRNNModelLarge 320/512/3 (main.py:16-20's model): slabs of 128 chunks cut across recordings, 3 forwards in flight,
    notes extracted on the device (only the note lists reach the host), framewise F1 against a Bernoulli(0.04) reference roll.
    Weights are random (no checkpoint exists): the frame head's bias is shifted so that ~0.3 % of the cells are active, which
    gives a trained model's note count (a few thousand per recording) instead of millions of one-frame notes."""
    import torch
    from music_transcription_amd import corpus
    n_rec, hours = 177, 20.0
    durations = corpus.synthetic_corpus(n_rec, hours, seed=0)
    model = seeded_model(mta, "cnn_rnn_large", str(dev)).eval()
    t1 = time.perf_counter()
    chunks = {i: corpus.synth_recording(i, durations[i], dev) for i in range(n_rec)}        # 4.6 GB resident before the timed region
    torch.cuda.synchronize()
    t_synth = time.perf_counter() - t1
    fe = mta.get_frontend(SR, N_MELS, HOP, str(dev))
    with torch.no_grad():                                                                   # calibrate the output bias on one slab
        mel, cmax = fe(torch.cat([chunks[i] for i in range(6)])[:64], clamp=False)
        lg = model.model(mel, chunk_max_power=cmax)
        shift = float(torch.quantile(lg.flatten()[::97].float(), 0.997))
        model.model.frame_head.bias.sub_(shift)              # (in place through autograd's version counter: the packed weights are rebuilt)
    gens = {}

    def ref_roll(i, T_total):
        g = gens.setdefault("g", torch.Generator(device=dev))
        g.manual_seed(i)
        return (torch.rand(88, T_total, device=dev, generator=g) < 0.04).float()
    res = corpus.transcribe_shard(model, list(range(n_rec)), lambda i: chunks[i], n_mels=N_MELS, device=dev, batch=128, streams=3,
                                  threshold=0.5, want_notes=True, reference_roll_of=ref_roll)
    n, wall = res["chunks"], res["wall_s"]
    f1 = list(res["f1"].values())
    sec = {"workload": f"offline corpus transcription, {n_rec} synthetic recordings / {hours:g} h of 16 kHz audio, CNNRNNModelLarge, 1 GPU "
                       "(BASELINE.json configs[4]; on 8 GPUs the recordings are LPT-sharded over the ranks, no data-path collective)",
           "value": round(n / wall, 1), "unit": "chunks/s", "wall_s": round(wall, 3), "chunks": n, "slabs_of_128": res["slabs"],
           "audio_hours_per_wall_second": round(hours / wall, 2), "notes": res["n_notes"], "finite": res["finite"],
           "mean_f1_vs_random_reference": round(float(sum(f1) / max(len(f1), 1)), 5),
           "d2h": "note lists only (two ints per note + 88 counts per recording); rolls and logits stay on the GPU",
           "timed": "slab assembly across recordings, mel + forward of every slab (3 streams), per-recording notes (mt_roll_to_notes) and F1 counts "
                    "as soon as a recording's last slab is done (logits held per slab); "
                    f"not timed: synthesis of the audio ({t_synth:.1f} s, resident in HBM), weight packing"}
    # ---- the same corpus END TO END from what a MAESTRO .wav holds: 44.1 kHz stereo int16 PCM in pinned host memory.  Timed: H2D of
    #      every recording (copy stream, two recordings ahead), channel mean + PCM scaling + polyphase resampling to 16 kHz
    #      (mt_resample_polyphase = librosa.load(sr=16000, mono=True), main.py:76), chunking (main.py:60-100), then as above.
    try:
        del chunks
        torch.cuda.empty_cache()
        rate = 44100
        t1 = time.perf_counter()
        pcm, off = {}, 0
        frames = [int(durations[i] * rate) for i in range(n_rec)]
        big = torch.empty(sum(frames), 2, dtype=torch.int16, pin_memory=True)      # ONE pinned region (12.7 GB), a view per recording
        for i in range(n_rec):
            g = torch.Generator(device=dev).manual_seed(777 + i)
            m = frames[i]
            t = torch.arange(m, device=dev, dtype=torch.float32) / rate
            y = 0.1 * torch.randn(m, device=dev, generator=g)
            for k in range(4):
                f0 = 27.5 * 2.0 ** (float(torch.randint(0, 88, (1,), device=dev, generator=g)) / 12.0)
                y += 0.3 * torch.exp(-((t * (0.5 + k)) % 3.0)) * torch.sin(2 * torch.pi * f0 * t)
            st = torch.stack([y, 0.8 * y + 0.02 * torch.randn(m, device=dev, generator=g)], 1).clamp_(-1, 1)
            pcm[i] = big[off:off + m]
            pcm[i].copy_((st * 32767.0).to(torch.int16))
            off += m
            del t, y, st
        torch.cuda.synchronize()
        t_pcm = time.perf_counter() - t1
        src = corpus.PcmSource(lambda i: pcm[i], lambda i: rate, list(range(n_rec)), dev, ahead=2)
        src(0)                                                 # the resampler's filter table for (44100, 16000): built once, not timed
        src = corpus.PcmSource(lambda i: pcm[i], lambda i: rate, list(range(n_rec)), dev, ahead=2)
        torch.cuda.synchronize()
        res2 = corpus.transcribe_shard(model, list(range(n_rec)), src, n_mels=N_MELS, device=dev, batch=128, streams=3,
                                       threshold=0.5, want_notes=True, reference_roll_of=ref_roll)
        n2, wall2 = res2["chunks"], res2["wall_s"]
        sec["_extra"] = {"configs4_corpus_from_pcm": {
            "workload": f"the same corpus end to end from {rate / 1000:g} kHz stereo int16 PCM in pinned host memory ({src.bytes_h2d / 1e9:.1f} GB): H2D + "
                        "channel mean + polyphase resampling to 16 kHz + chunking + mel + CNNRNNModelLarge forward + notes + F1, 1 GPU",
            "value": round(n2 / wall2, 1), "unit": "chunks/s", "wall_s": round(wall2, 3), "chunks": n2, "notes": res2["n_notes"], "finite": res2["finite"],
            "audio_hours_per_wall_second": round(hours / wall2, 2), "h2d_gb": round(src.bytes_h2d / 1e9, 2),
            "h2d_gb_per_s_needed": round(src.bytes_h2d / 1e9 / wall2, 1),
            "timed": "everything from the pinned host PCM to the note lists; not timed: synthesis of the PCM "
                     f"({t_pcm:.1f} s), the resampler's filter table (316 KB, built once per rate pair), weight packing"}}
        del pcm, big
    except Exception as e:
        import traceback
        sec["_extra"] = {"configs4_corpus_from_pcm": {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}}
    return sec


LINE_LIMIT = 4096          # bytes: the driver keeps a bounded tail of stdout; round 3's 21.7 KB line was cut and could not be parsed


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(d):
    """The ONE JSON line of the contract, <= LINE_LIMIT bytes: the contract's scalar fields, `roofline` (dominant kernel + every
    stage's fraction as {kernel: frac}), `cpu_baseline`, the literal configs[1] schedule (one batch of 32 per forward) and one
    short object per neighbouring BASELINE config.  Stage tables, schedules and autotune lists live in bench_detail.json."""
    rf = d.get("roofline") or {}
    roof = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "share_of_step", "mfma_dtype"))
    roof.setdefault("traffic", None)
    roof["all"] = {k: round(v, 3) for k, v in (rf.get("all") or {}).items()}
    cfg = d.get("config") or {}
    config = _pick(cfg, ("workload", "batch_per_gpu", "frames", "coscheduled_batches_per_forward", "streams_per_gpu", "batches_in_flight", "parallelism"))
    cpu = d.get("cpu_baseline")
    if isinstance(cpu, dict):
        c2 = _pick(cpu, ("value", "unit", "cores", "kind", "error"))
        if "sample" in cpu:
            c2["sample"] = str(cpu["sample"])[:120]
        if isinstance(cpu.get("batch8"), dict):
            c2["batch8_value"] = cpu["batch8"].get("value")
        cpu = c2
    out = _pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    out["vs_baseline"] = d.get("vs_baseline")
    out.update({"dtype": "f16", "data": d.get("data", "synthetic"), "config": config, "roofline": roof, "cpu_baseline": cpu})
    if isinstance(d.get("roofline_one_batch"), dict):
        out["roofline_one_batch"] = _pick(d["roofline_one_batch"], ("kernel", "frac", "avg_launch_ms"))
    lit = d.get("configs1_literal_b32")
    if isinstance(lit, dict):
        out["configs1_literal_b32"] = _pick(lit, ("value", "unit", "ms_per_step", "streams", "one_in_flight", "error"))
    secs = {}
    for name in ("configs2_large_b16", "configs3_train_b16", "train_large_b16", "configs4_corpus", "configs4_corpus_from_pcm"):
        sec = d.get(name)
        if not isinstance(sec, dict):
            continue
        if "error" in sec:
            secs[name] = {"error": str(sec["error"])[:160]}
            continue
        o = _pick(sec, ("value", "unit", "ms_per_step", "wall_s", "chunks", "notes"))
        r_ = sec.get("roofline")
        if isinstance(r_, dict):
            o["roofline_kernel"], o["roofline_frac"] = str(r_.get("kernel"))[:40], r_.get("frac")
        cb = sec.get("cpu_baseline")
        if isinstance(cb, dict):
            o["cpu_baseline"] = cb.get("value")
        secs[name] = o
    out["sections"] = secs
    for key in ("all_large", "all_train"):
        if rf.get(key):
            out["roofline"][key] = {k: round(v, 3) for k, v in rf[key].items()}
    out["detail"] = "bench_detail.json"
    line = json.dumps(out, separators=(",", ":"))
    for drop in (("roofline", "all_train"), ("roofline", "all_large"), ("roofline_one_batch",), ("sections",)):     # never over the limit
        if len(line) <= LINE_LIMIT:
            break
        tgt = out
        for k in drop[:-1]:
            tgt = tgt.get(k, {})
        tgt.pop(drop[-1], None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def emit(detail):
    """Full record -> bench_detail.json (repo root, and gpurun_out/ when it exists so that it travels back from a GPU box);
    compact record -> the LAST line of stdout."""
    paths = ([os.environ["MT_BENCH_DETAIL"]] if os.environ.get("MT_BENCH_DETAIL") else
             [os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")])
    for path in paths:
        try:
            if os.path.isdir(os.path.dirname(os.path.abspath(path))):
                with open(path, "w") as f:
                    json.dump(detail, f)
        except OSError as e:
            log(f"could not write {path}: {e}")
    sys.stderr.flush()
    print(compact_line(detail), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (400 x 3 ms: long enough that the first and last rounds of the forwards in flight do not weigh)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--streams", type=int, default=4,
                    help="forwards in flight per GPU (forward f is issued whole on stream f %% streams)")
    ap.add_argument("--cosched", type=int, default=4,
                    help="batches (steps) co-scheduled into ONE forward: the recurrence interleaves their batch groups inside one "
                         "persistent launch (csrc/lstm.hip, NG).  1 = one batch per forward, as in round 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sections", action="store_true", help="only the headline line: skip the 1-stream pass and the configs[2] / configs[3] sections")
    ap.add_argument("--model", choices=["cnn_rnn", "cnn_rnn_large"], default="cnn_rnn",
                    help="cnn_rnn = the bench line (BASELINE configs[1]); cnn_rnn_large = informational run of configs[2] "
                         "(use --batch 16): whole-step timing only")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer = the bench line; train = informational run of configs[3] (use --batch 16 --steps 5)")
    args = ap.parse_args()
    if args.mode == "train":
        # a training-only process: 32 hardware queues, so that the step's side streams never share one with the calling stream (a shared
        # queue serialises them: 46 instead of 42 ms per CNNRNNModelLarge step; tools/train_large_queue_probe.py, profiles/r04_train_large_stream_mapping.txt).
        # Inference keeps 8 (9 996 / 9 430 chunks/s at 400 / 20 steps against 9 925 - 9 956 / 9 207 - 9 393 with 32).
        if not _USER_SET_QUEUES:
            os.environ["GPU_MAX_HW_QUEUES"] = "32"
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
    C = max(1, min(args.cosched, 4, K))               # batches per forward (the recurrence interleaves at most 4 batch groups of 32)
    BF = C * B                                        # chunks per forward
    T = mta.num_frames(N_SAMPLES, HOP)
    # A STEP is one pass of the hot path over one batch of B = 32 chunks (BASELINE configs[1]).  The steps of the timed region are
    # issued C at a time as one forward over C x 32 chunks (K % C left-over steps as one smaller forward at the end), and --streams
    # such forwards are in flight: C x streams batches of 32 at once, against round 1's 3 (one per stream).
    # seeded synthetic input: noise + decaying piano-range sinusoids; seed = 1234 + rank.
    # Four distinct chunks tiled to the batch keep host-side synthesis short; the kernels see B chunks.
    # Eight chunks are synthesised on the host (that is what takes time there); the other chunks of a forward are DISTINCT
    # signals derived from them on the device: a circular time shift, a gain and fresh seeded noise per copy, so that no two of
    # the C x B chunks of a forward are equal (every kernel works on data, not on repeats a cache could serve).
    base = torch.from_numpy(synth_audio(min(B, 8), N_SAMPLES, seed=1234 + rank)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    parts = []
    for c in range((C * B + len(base) - 1) // len(base)):
        w = base if c == 0 else torch.roll(base, shifts=7919 * c, dims=1) * (0.55 + 0.03 * (c % 13))
        if c:
            w = w + 0.02 * torch.randn(w.shape, device=dev, generator=gen)
        parts.append(w.clamp_(-1.0, 1.0) if c else w)
    wave_f = torch.cat(parts)[:C * B].contiguous()
    if os.environ.get("MT_BENCH_TILE_CHUNKS") == "1":       # rounds 1-2: four distinct chunks repeated through the forward (A/B of the data effect)
        wave_f = torch.cat([base[:4]] * (C * B // 4))[:C * B].contiguous()
    wave = wave_f[:B]
    del parts
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
    net.fuse_input_projection = NS == 2 and C == 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    mel = [torch.empty(BF, 1, N_MELS, T, device=dev) for _ in range(NS)]
    cmax = [torch.empty(BF, device=dev) for _ in range(NS)]

    nst = 3 + 3 * LAYERS

    def plan_forwards(ns):
        """The K timed steps as forwards [(stream, batches)]: whole rounds of ns forwards of C batches, and the batches left
        over (fewer than a round) dealt EVENLY over the streams as smaller forwards that run side by side -- a finite job
        must not end on one full-size forward running alone (K = 20 at 4 x 4: one round of 4 x 4 and a last round of 4 x 1;
        a forward of one batch takes a third of the time of a forward of four: its recurrences interleave nothing)."""
        rounds, rem = divmod(K, C * ns)
        out = [(s_, C) for _ in range(rounds) for s_ in range(ns)]
        tail = [rem // ns + (1 if s_ < rem % ns else 0) for s_ in range(ns)]
        return out + [(s_, nb_) for s_, nb_ in enumerate(tail) if nb_]

    def make_events(n):
        ev_m = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(n)]
        ev_n = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(n)]
        for row in ev_m + ev_n:          # create the underlying hipEvent_t handles before the timed region
            for e in row:
                e.record()
        return ev_m, ev_n

    def forward(j, i=None, nb=C, s=None):
        """One whole pass (mel + forward) over nb batches of B chunks, issued on stream s (default j % NS)."""
        s = j % NS if s is None else s
        n = nb * B
        with torch.cuda.stream(streams[s]), torch.no_grad():
            if i is not None:
                ev_mel[i][0].record()
            fe(wave_f[:n], clamp=False, out=mel[s][:n], chunk_max=cmax[s][:n])   # unclamped dB + per-chunk max; conv1 clamps on load
            if i is not None:
                ev_mel[i][1].record()
            return net(mel[s][:n], chunk_max_power=cmax[s][:n], events=None if i is None else ev_net[i])

    log(f"rank {rank}/{world}: setup done, B={B} x {C} per forward, T={T}, {NS} streams; warmup {W}, steps {K}")
    from music_transcription_amd._lib import MtError
    while True:
        try:
            sched = plan_forwards(NS)
            for nb_ in sorted({nb_ for _, nb_ in sched}):        # every forward shape of the timed region (workspaces), then the warm-up
                forward(0, nb=nb_)
            for j in range(max((W + C - 1) // C, NS if W else 0)):
                forward(j)
            torch.cuda.synchronize()
            break
        except MtError as e:
            # the library refuses a persistent launch that could not be resident next to the ones in flight (csrc/residency.hip):
            # run the timed region with one forward less in flight rather than not at all
            torch.cuda.synchronize()
            if NS == 1:
                raise
            NS -= 1
            net._ws.clear()
            log(f"{NS + 1} forwards in flight refused ({str(e)[:160]}): continuing with {NS}")
    sched = plan_forwards(NS)
    assert sum(nb_ for _, nb_ in sched) == K
    NF = len(sched)
    ev_mel, ev_net = make_events(NF)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (s_, nb_) in enumerate(sched):
        logits = forward(i, i, nb=nb_, s=s_)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for nb_ in sorted({nb_ for _, nb_ in sched}):
        net.raise_on_handoff_timeout(nb_ * B, T)
    log(f"timed region: {elapsed:.3f} s for {K} steps")
    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- un-overlapped passes: ONE forward in flight, so that per-kernel times are kernel efficiencies, not contention:
    #      (a) the forward shape of the timed region (C batches of 32 per forward: the kernels the headline runs),
    #      (b) one batch of 32 per forward
    one_ms = {}
    if rank == 0 and not args.no_sections:
        K1 = 8
        fused_hl = bool(net.fuse_input_projection)
        net.fuse_input_projection = False
        try:
            for nb in sorted({C, 1}, reverse=True):
                n = nb * B
                ev1m = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(K1)]
                ev1n = [[torch.cuda.Event(enable_timing=True) for _ in range(nst + 1)] for _ in range(K1)]
                with torch.cuda.stream(streams[0]), torch.no_grad():
                    for row in ev1m + ev1n:
                        for e in row:
                            e.record()
                    for i in range(K1 + 1):
                        ii = max(i - 1, 0)                   # (first iteration = warm-up of this shape, overwritten)
                        ev1m[ii][0].record()
                        fe(wave_f[:n], clamp=False, out=mel[0][:n], chunk_max=cmax[0][:n])
                        ev1m[ii][1].record()
                        net(mel[0][:n], chunk_max_power=cmax[0][:n], events=ev1n[ii])
                torch.cuda.synchronize()
                net.raise_on_handoff_timeout(n, T)
                row = [float(np.mean([ev1m[i][0].elapsed_time(ev1m[i][1]) for i in range(K1)]))]
                for s_ in range(nst):
                    row.append(float(np.mean([ev1n[i][s_].elapsed_time(ev1n[i][s_ + 1]) for i in range(K1)])))
                one_ms[nb] = row
        except Exception as e:                          # the headline line must not depend on these passes
            torch.cuda.synchronize()
            log(f"un-overlapped passes failed ({type(e).__name__}: {str(e)[:200]}): stages fall back to the timed region's")
            one_ms = {}
        net.fuse_input_projection = fused_hl

    if rank == 0:
        # ---- per-kernel times from the events recorded inside the timed region (several batches in flight: kernels of
        #      different steps overlap, so these are NOT kernel efficiencies -- those come from the 1-stream pass below)
        sizes = [nb_ for _, nb_ in sched]
        nb_o = max(set(sizes), key=sizes.count)             # the commonest forward size of the timed region
        idx_o = [i for i, nb_ in enumerate(sizes) if nb_ == nb_o]
        table = stage_table(nb_o * B, T, N_MELS, HIDDEN, LAYERS, fused=bool(net.fuse_input_projection))
        ms = [float(np.mean([ev_mel[i][0].elapsed_time(ev_mel[i][1]) for i in idx_o]))]
        for s in range(nst):
            ms.append(float(np.mean([ev_net[i][s].elapsed_time(ev_net[i][s + 1]) for i in idx_o])))
        stages_overlapped = _stage_rows(table, ms)
        dom_key_o, roofline_overlapped = _roofline_from_stages(stages_overlapped)
        stages_one_batch, roofline_one_batch = None, None
        if one_ms:
            stages = _stage_rows(stage_table(BF, T, N_MELS, HIDDEN, LAYERS, fused=False), one_ms[C])
            if 1 in one_ms and C > 1:
                stages_one_batch = _stage_rows(stage_table(B, T, N_MELS, HIDDEN, LAYERS, fused=False), one_ms[1])
                _, roofline_one_batch = _roofline_from_stages(stages_one_batch)
                roofline_one_batch["measured_in"] = f"un-overlapped pass of this run: one forward over ONE batch of {B} chunks in flight"
        else:
            stages = stages_overlapped
        dom_key, roofline = _roofline_from_stages(stages)
        roofline["measured_in"] = (f"un-overlapped pass of this run: ONE forward of the timed region's shape ({C} batches of {B} chunks, "
                                   f"the same kernels) in flight") if one_ms else "timed region (forwards overlap)"
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure comes from the committed
        # rocprofv3 --pmc passes of this same command (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950), and the line says so
        traffic, tsrc = None, None
        for fname in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", fname)))["kernels"]
                hit = [v for k, v in prof.items() if ("lstm_rec_kernel" in k if dom_key == "lstm_rec" else dom_key.split("_l")[0] in k.replace("::", "_"))]
                if hit and B == 32 and json.load(open(os.path.join(ROOT, "profiles", fname))).get("batches_per_forward", 1) == C:
                    traffic = round(sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit))
                    tsrc = f"profiles/{fname} (committed rocprofv3 --pmc passes of this command; not re-measured in this run)"
                    break
            except Exception:
                continue
        roofline["traffic"], roofline["traffic_source"] = traffic, tsrc

        # ---- CPU baseline: the oracle (port of the reference path) on this node's host cores
        cpu = None
        cores = host_cores()
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline_small(model, wave, logits[:B], cores)
            except Exception as e:                      # (reported, never fatal for the line)
                cpu = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        sections = {}
        if world == 1 and not args.no_sections:
            del mel, cmax
            net._ws.clear()
            torch.cuda.empty_cache()
            try:
                sections["configs1_other_schedules"] = section_coscheduled(mta, dev, net, fe, wave)
            except Exception as e:
                import traceback
                sections["configs1_other_schedules"] = {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}
            # (the training section first: the HIP runtime deals a process's streams onto its hardware queues in creation order,
            #  and after the streams of the Large section the training step's side stream ends up sharing a queue with the
            #  stream it is meant to overlap -- 630 instead of 700 chunks/s)
            for name, fn in (("configs3_train_b16", section_train), ("train_large_b16", section_train_large), ("configs2_large_b16", section_large),
                             ("configs4_corpus", section_corpus)):
                t1 = time.perf_counter()
                try:
                    sections[name] = fn(mta, dev, cores, not args.no_cpu_baseline)
                    sections.update(sections[name].pop("_extra", {}))
                except Exception as e:                      # a section must not cost the headline line
                    import traceback
                    sections[name] = {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}
                log(f"section {name}: {time.perf_counter() - t1:.1f} s")
                torch.cuda.empty_cache()

        # the LITERAL configs[1] schedule, first-class: one batch of 32 chunks per forward (3 or 4 forwards in flight, the faster of the two; and one alone)
        cands = [(ns_, (sections.get("configs1_other_schedules") or {}).get(f"b32_streams_{ns_}")) for ns_ in (3, 4)]
        good = [(ns_, c_) for ns_, c_ in cands if isinstance(c_, dict) and "value" in c_]
        lit_ns, lit = max(good, key=lambda nc: nc[1]["value"]) if good else cands[0]
        if isinstance(lit, dict) and "value" in lit:
            sections["configs1_literal_b32"] = {
                "workload": "CNNRNNModel inference, ONE batch of 32 chunks per forward (BASELINE.json configs[1] as written), mel + forward",
                "value": lit["value"], "unit": "chunks/s", "ms_per_step": lit["ms_per_forward"], "streams": lit_ns,
                "by_streams": {str(ns_): c_["value"] for ns_, c_ in good},
                "one_in_flight": round(1e3 * B / sum(one_ms[1]), 1) if 1 in one_ms else None}
        elif isinstance(lit, dict):
            sections["configs1_literal_b32"] = lit
        # every kernel's fraction of its roofline in one compact object (survives a truncated log): the headline forward's stages,
        # the Large forward's and the training step's timed launches
        roofline["all"] = {s_["kernel"]: s_["frac"] for s_ in stages if s_["work_per_launch"] > 0}
        for sec_name, key in (("configs2_large_b16", "stages_one_stream"), ("configs3_train_b16", "stages_timed")):
            sec_ = sections.get(sec_name)
            if isinstance(sec_, dict) and key in sec_:
                roofline["all_" + sec_name.split("_")[1]] = {s_["kernel"]: s_["frac"] for s_ in sec_[key] if s_["work_per_launch"] > 0}
        detail = {"metric": "30 s audio chunks/sec (mel+CNNRNN forward)", "value": round(world * B * K / elapsed, 2),
                  "unit": "chunks/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 3),
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": "f16 MFMA operands with f32 accumulate (conv2, input projections, recurrence, fc), f32 LSTM state / gates, "
                           "f32 FFT and conv1",
                  "data": "synthetic",
                  "config": {"workload": "CNNRNNModel inference, batch=32x30 s synthetic 16 kHz audio, mel+CNN-RNN HIP path "
                                         "(BASELINE.json configs[1])", "batch_per_gpu": B, "n_samples": N_SAMPLES,
                             "n_mels": N_MELS, "hidden": HIDDEN, "layers": LAYERS, "frames": T, "parallelism": f"dp{world} (independent chunks)",
                             "coscheduled_batches_per_forward": C, "streams_per_gpu": NS, "batches_in_flight": C * NS,
                             "forwards_in_timed_region": {f"{nb_}x{B}": sizes.count(nb_) for nb_ in sorted(set(sizes))},
                             "scheduling": f"a step = one batch of {B} chunks; the {K} timed steps are issued as forwards over {C} batches (the recurrence interleaves "
                                           f"their batch groups in one persistent launch), dealt round-robin over {NS} streams: "
                                           f"{NS} forwards in flight; the batches beyond whole rounds of {NS} x {C} run as a last round of smaller forwards side by side.  One batch of {B} per forward, the literal configs[1] schedule: configs1_literal_b32",
                             "distinct_chunks_per_forward": True,
                             "fused_input_projection": bool(net.fuse_input_projection)},
                  "roofline": roofline, "cpu_baseline": cpu, "stages": stages,
                  "roofline_one_batch": roofline_one_batch, "stages_one_batch": stages_one_batch,
                  "roofline_overlapped": roofline_overlapped, "stages_overlapped": stages_overlapped, **sections}
        emit(detail)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
