"""Does the Large training step's time depend on WHICH streams of torch's pool its two side streams are?  One process; before every trial one
more dummy stream is taken from the pool and the step's side streams are re-created.  Measured (profiles/r03_train_large_stream_mapping.txt):
49.4 - 49.8 ms for most mappings, 51 - 52 for some, 55.5 - 56 (the single-stream time) for about one in eight -- and pairwise concurrency
probes (a tiny kernel beside eight 256-MB fills, or beside one 3-ms matrix product, every pair of {calling stream, side A, side B} in both
directions) report "runs beside" for ALL of them, the slow ones included; running the recurrences on high-priority streams does not help
(51 - 56 ms).  Round 4: the mechanism is the caching allocator, not the queues -- its pools are per stream, a fresh side stream starts with an
empty one, and a device allocation inside the timed steps costs up to 80 ms (0.5 - 1 GB blocks).  This version prints the device allocations
of the timed steps and the event-to-event step times beside the wall clock."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import music_transcription_amd as mta
from music_transcription_amd import train_step_large as TL
from bench import seeded_model, N_MELS

dev = torch.device("cuda", 0)
B, T = 16, 937
g = torch.Generator().manual_seed(1234)
model = seeded_model(mta, "cnn_rnn_large", str(dev), dropout=0.2)
opt = mta.make_optimizer(model, lr=1e-4)
lengths = torch.full((B,), T)
mel = (torch.rand(B, 1, N_MELS, T, generator=g) * 60.0 - 70.0).to(dev)
roll = (torch.rand(B, 88, T, generator=g) < 0.04).float().to(dev)
model.train()


def step():
    opt.zero_grad()
    loss = model.compute_loss(model(mel), roll, lengths)
    loss.backward()
    opt.step()


dummies = []
for trial in range(10):
    TL._SIDE2.clear()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    marks = []
    t0 = time.perf_counter()
    for _ in range(5):
        step()
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / 5
    ev = [marks[i].elapsed_time(marks[i + 1]) for i in range(4)]
    sa, sb = TL._SIDE2[TL.device_key(dev)]
    print(f"dummy streams created before the side streams: {len(dummies)}: wall {wall:.2f} ms per step, event to event {min(ev):.2f} .. {max(ev):.2f} ms, "
          f"device allocations in the timed steps {torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - a0} "
          f"(side streams {sa.cuda_stream:#x} {sb.cuda_stream:#x})", flush=True)
    dummies.append(torch.cuda.Stream(device=dev))
