"""From a rocprofv3 --kernel-trace CSV: for the LAST `steps` fraction of the trace, the time each kernel class runs ALONE (nothing else on the
GPU) and the idle time -- what a multi-stream step's wall clock is made of.  Usage: python tools/trace_exclusive.py <dir> <n_steps_in_trace>"""
import csv, glob, os, re, sys
from collections import defaultdict

root, nsteps = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:60]))
rows.sort()
# the optimizer's kernel ends a step
ends = [e for s, e, n in rows if "adam_clip_kernel" in n]
ends = ends[-(nsteps):]
t0, t1 = ends[0], ends[-1]
win = [(s, e, n) for s, e, n in rows if s >= t0 and e <= t1]
ev = []
for i, (s, e, n) in enumerate(win):
    ev.append((s, 1, i)); ev.append((e, 0, i))
ev.sort()
active = set()
last = t0
excl = defaultdict(float); idle = 0.0; multi = 0.0
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        if len(active) == 0: idle += dt
        elif len(active) == 1: excl[win[next(iter(active))][2]] += dt
        else: multi += dt
    last = t
    if kind == 1: active.add(i)
    else: active.discard(i)
n = len(ends) - 1
print(f"{n} steps, {(t1 - t0) / n * 1e-6:.2f} ms per step: idle {idle / n * 1e-6:.2f} ms, >= 2 kernels {multi / n * 1e-6:.2f} ms, one kernel alone {sum(excl.values()) / n * 1e-6:.2f} ms")
for k, v in sorted(excl.items(), key=lambda kv: -kv[1])[:30]:
    print(f"  {v / n * 1e-6:7.3f} ms alone  {k}")
