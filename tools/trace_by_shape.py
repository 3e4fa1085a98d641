"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): calls, total ms, mean us.  Usage:
python tools/trace_by_shape.py <dir with *_kernel_trace.csv> [calls divisor (steps)]"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        agg[key][0] += 1
        agg[key][1] += d
tot = sum(v[1] for v in agg.values())
print(f"total {tot / div:.2f} ms per step over {len(agg)} (kernel, grid) classes")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    wg = int(k[4]) or 1
    print(f"{v[1] / div:8.3f} ms  {v[0] / div:6.1f} calls  {1e3 * v[1] / v[0]:9.1f} us  grid=({int(k[1]) // wg},{k[2]},{k[3]})x{k[4]}  {k[0]}")
