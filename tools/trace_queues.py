"""From a rocprofv3 --kernel-trace CSV: which hardware queue every kernel of the CNNRNNModelLarge training step ran on, how much the queues overlap,
and a timeline of one step's long kernels -- to see WHY a step that takes 46 ms in a fresh process takes 55 - 66 ms behind other work of the
same process.  Usage: python tools/trace_queues.py <dir with *_kernel_trace.csv> [steps=4]"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mt::", "")[:48],
                     r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Thread_Id", "?"))))
rows.sort()
large = [i for i, r in enumerate(rows) if "bn_act_bwd_kernel" in r[2]]          # only the Large training step runs this kernel
if not large:
    sys.exit("no CNNRNNModelLarge training step in this trace")
lo, hi = rows[large[0]][0], rows[large[-1]][1]
ends = [e for s, e, n, q, st in rows if "adam_clip_kernel" in n and lo <= s <= hi + 5_000_000]
ends = ends[-(nsteps + 1):]
t0, t1 = ends[0], ends[-1]
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
n = len(ends) - 1
print(f"{n} steps of the Large training step, {(t1 - t0) / n * 1e-6:.2f} ms per step, {len(win) / n:.0f} kernels per step")
byq = defaultdict(lambda: [0.0, 0, defaultdict(float)])
for s, e, nm, q, st in win:
    byq[q][0] += e - s; byq[q][1] += 1; byq[q][2][nm] += e - s
for q, (busy, cnt, names) in sorted(byq.items(), key=lambda kv: -kv[1][0]):
    top = ", ".join(f"{k} {v / n * 1e-6:.2f}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:5])
    print(f"  queue {q}: busy {busy / n * 1e-6:6.2f} ms per step, {cnt / n:5.0f} kernels per step: {top}")
# concurrency: time with kernels of k different queues running
ev = []
for i, (s, e, nm, q, st) in enumerate(win):
    ev.append((s, 1, q)); ev.append((e, 0, q))
ev.sort()
act = defaultdict(int); last = t0; hist = defaultdict(float)
for t, kind, q in ev:
    k = sum(1 for v in act.values() if v > 0)
    hist[k] += t - last; last = t
    act[q] += 1 if kind else -1
print("  time per step with kernels of k queues running: " + ", ".join(f"k={k}: {v / n * 1e-6:.2f} ms" for k, v in sorted(hist.items())))
# one step's long kernels in start order
s0, s1 = ends[-2], ends[-1]
print(f"  last step ({(s1 - s0) * 1e-6:.2f} ms): kernels >= 0.25 ms in start order (start ms, duration ms, queue, name)")
for s, e, nm, q, st in win:
    if s >= s0 and e - s >= 250_000:
        print(f"    {(s - s0) * 1e-6:7.2f} {(e - s) * 1e-6:6.2f}  q{q}  {nm}")
