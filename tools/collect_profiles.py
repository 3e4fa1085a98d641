"""Copies the evidence that tools/refresh_profiles.sh and tools/refresh_pmc.sh left under gpurun_out/ into profiles/<round>_*."""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, pmc, dst = os.path.join(ROOT, "gpurun_out", "refresh"), os.path.join(ROOT, "gpurun_out", "pmc"), os.path.join(ROOT, "profiles")
names = {"default": "bench_default", "driver": "bench_driver_cmd", "1stream": "bench_1stream", "1forward": "bench_1forward", "train": "train_b16",
         "train_large": "train_large_b16", "large": "large_b16"}
for k, n in names.items():
    line = os.path.join(src, f"{k}_line.json")
    if os.path.exists(line):
        txt = [l for l in open(line).read().strip().splitlines() if l.startswith("{")]
        open(os.path.join(dst, f"{R}_{n}_line.json"), "w").write(txt[-1] + "\n")
    detail = os.path.join(src, f"{k}_detail.json")          # the full record behind the compact line (stage tables, schedules)
    if os.path.exists(detail):
        shutil.copy(detail, os.path.join(dst, f"{R}_{n}_detail.json"))
    stats = glob.glob(os.path.join(src, f"p_{k}", "*", "*_kernel_stats.csv"))
    if stats:                                       # (one file per process: of the latest run's files the bench itself is the largest)
        newest = max(os.path.getmtime(f) for f in stats)
        latest = [f for f in stats if newest - os.path.getmtime(f) < 300]
        shutil.copy(max(latest, key=os.path.getsize), os.path.join(dst, f"{R}_{n}_kernel_stats.csv"))
for f in ("train_large_exclusive_time.txt", "train_large_by_shape.txt", "train_exclusive_time.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{R}_{f}"))
for f in ("pmc_traffic.json", "pmc_mfma_util.json", "pmc_traffic_b32.json", "pmc_mfma_util_b32.json", "pmc_traffic_large.json", "pmc_mfma_util_large.json",
          "pmc_traffic_train.json", "pmc_mfma_util_train.json", "pmc_traffic_train_large.json", "pmc_mfma_util_train_large.json"):
    if os.path.exists(os.path.join(pmc, f)):
        shutil.copy(os.path.join(pmc, f), os.path.join(dst, f"{R}_{f}"))
print("\n".join(sorted(os.listdir(dst))))
