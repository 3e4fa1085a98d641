"""Per-kernel sums of the counters of a rocprofv3 --pmc run (counter_collection.csv).  python tools/pmc_generic_summary.py <dir> [name filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        if flt and flt not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    n = max(calls[(k, c)] for c in cs)
    print(k, f"launches {n}:", {c: round(v / n, 1) for c, v in cs.items()})
