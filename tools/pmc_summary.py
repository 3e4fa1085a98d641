#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_pmc_traffic.json.

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> "<command string>" [batches per forward of that command]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so streaming reads are doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0]
                if name.startswith("void "):
                    name = name[5:]
                acc[name].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch_dir, write_dir, out, cmd = sys.argv[1:5]
    bpf = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fe) | set(wr)):
        if not name.startswith("mt::"):
            continue
        f = sum(fe[name]) / max(len(fe[name]), 1)
        w = sum(wr[name]) / max(len(wr[name]), 1)
        kernels[name] = {
            "launches": len(fe[name]) or len(wr[name]),
            "FETCH_SIZE_KB_mean": f,
            "WRITE_SIZE_KB_mean": w,
            "hbm_bytes_per_launch_corrected": 2.0 * f * 1024.0 + w * 1024.0,
        }
    json.dump({"command": cmd, "batches_per_forward": bpf,
               "note": "per-launch means over all launches of the kernel name; corrected = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE halves streaming reads)",
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
