#!/usr/bin/env python3
"""BASELINE.json configs[4]: offline transcription of a whole corpus (MAESTRO-test sized: ~177 recordings, ~20 h),
recordings sharded over the GPUs of one node, end-to-end wall-clock + framewise F1.

    python scripts/transcribe_corpus.py [--wav-dir DIR | --synthetic 177 --hours 20] [--model CKPT.pth]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/transcribe_corpus.py ...

One process per GPU; recordings are assigned longest-first (LPT) so ranks finish together; there is NO data-path
collective: every rank runs chunk -> mel -> model -> threshold on its own recordings; one small all-reduce gathers the
per-recording F1 values at the end.  With --wav-dir, `<name>.wav` is transcribed and, if `<name>.roll.npy`
((88, T_total) {0,1}) exists, scored against it.  Without data (no MAESTRO in this image) --synthetic draws
recording durations to the requested total, synthesises noise + decaying tones on the GPU and scores against random
rolls: that exercises the whole path and gives throughput, the F1 is then meaningless.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per stream in flight (HIP's default of 4 makes streams share)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wav-dir")
    ap.add_argument("--synthetic", type=int, default=177)
    ap.add_argument("--hours", type=float, default=20.0)
    ap.add_argument("--model")
    ap.add_argument("--model-type", default="cnn_rnn_large")
    ap.add_argument("--n-mels", type=int, default=320)
    ap.add_argument("--hidden-size", type=int, default=512)
    ap.add_argument("--num-layers", type=int, default=3)
    ap.add_argument("--threshold", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=128,
                    help="chunks per forward (the recurrence interleaves up to four batch groups of 32 in one persistent launch)")
    ap.add_argument("--streams", type=int, default=4, help="forwards in flight (at most 3 for cnn_rnn_large: two recurrence launches each)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out-dir", help="directory: write <name>.mid per recording (notes extracted on the device)")
    ap.add_argument("--dump-rolls", help="directory: write <name>.roll.bits.npy (np.packbits of the (88, T_total) roll) per recording")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("MT_BENCH_BACKEND", "nccl")
    dev_index = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import music_transcription_amd as mta
    from music_transcription_amd import transcribe as tr
    from music_transcription_amd.parallel import lpt_assign, gather_values
    SR, CH = 16000, 480000

    model = mta.TranscriptionModel(args.model_type, n_mels=args.n_mels, hidden_size=args.hidden_size, num_layers=args.num_layers, device=dev)
    if args.model:
        model.load_state_dict(torch.load(args.model, map_location=dev))
    else:                                   # no checkpoint in this image: seeded random weights, same on every rank
        torch.manual_seed(1234)
        for p in model.parameters():
            if p.dim() > 1:
                torch.nn.init.uniform_(p, -0.05, 0.05)
    model.eval()

    if args.wav_dir:
        names = sorted(f[:-4] for f in os.listdir(args.wav_dir) if f.endswith(".wav"))
        durations = [os.path.getsize(os.path.join(args.wav_dir, n + ".wav")) for n in names]      # bytes ~ duration
    else:
        rng = np.random.default_rng(args.seed)
        raw = rng.gamma(2.5, 1.0, size=args.synthetic)
        durations = list(raw / raw.sum() * args.hours * 3600.0)                                    # seconds, MAESTRO-like spread
        names = [f"synthetic_{i:03d}" for i in range(args.synthetic)]
    mine = lpt_assign(durations, world)[rank]

    from music_transcription_amd import corpus
    NS = max(1, min(args.streams, 3 if args.model_type == "cnn_rnn_large" else 6))
    synth_chunks = None if args.wav_dir else {i: corpus.synth_recording(i, durations[i], dev, args.seed) for i in mine}   # resident, not timed

    def chunks_of(i):
        if args.wav_dir:                    # file read + H2D + GPU resample + split are part of the end-to-end time
            return tr.split_into_chunks_device(tr.load_audio_device(os.path.join(args.wav_dir, names[i] + ".wav"), SR, dev))[0]
        return synth_chunks[i]

    def reference_roll_of(i, T_total):
        ref_path = os.path.join(args.wav_dir, names[i] + ".roll.npy") if args.wav_dir else None
        if ref_path and os.path.exists(ref_path):
            return torch.from_numpy(np.load(ref_path)).float().to(dev)
        return (torch.rand(88, T_total, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) < 0.04).float()

    def midi_path_of(i):
        if not args.out_dir:
            return None
        os.makedirs(args.out_dir, exist_ok=True)
        return os.path.join(args.out_dir, names[i] + ".mid")

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    res = corpus.transcribe_shard(model, mine, chunks_of, n_mels=args.n_mels, device=dev, batch=args.batch, streams=NS,
                                  threshold=args.threshold, want_notes=True, reference_roll_of=reference_roll_of, midi_path_of=midi_path_of)
    if args.dump_rolls:                     # (debug / tests: the rolls are rebuilt from the notes -- they never left the GPU as rolls)
        os.makedirs(args.dump_rolls, exist_ok=True)
        fs = SR / 512
        for i in mine:
            T_total = res["chunks_per_recording"][i] * 938
            notes = res["notes"].get(i, [])
            roll = np.zeros((88, T_total), dtype=np.uint8)
            for pch, a_, b_ in notes:
                roll[pch - 21, int(round(a_ * fs)):int(round(b_ * fs))] = 1
            np.save(os.path.join(args.dump_rolls, names[i] + ".roll.bits.npy"), np.packbits(roll, axis=1))
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    f1s = [res["f1"].get(i, 0.0) for i in mine]
    n_chunks = res["chunks"]
    n_notes = gather_values([rank], [float(res["n_notes"])], world)
    allf1 = gather_values(mine, f1s, len(names))
    tot_chunks = gather_values([rank], [float(n_chunks)], world)
    if rank == 0:
        print(json.dumps({"workload": "offline corpus transcription (BASELINE.json configs[4])", "recordings": len(names),
                          "audio_hours": round(sum(durations) / 3600.0, 2) if not args.wav_dir else None, "n_gpus": world,
                          "chunks": int(sum(tot_chunks)), "wall_s": round(wall, 3), "chunks_per_s": round(sum(tot_chunks) / wall, 1),
                          "notes": int(sum(n_notes)), "mean_f1": float(np.mean(allf1)), "per_recording_f1": [float(v) for v in allf1], "model": args.model_type, "data": "wav" if args.wav_dir else "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
