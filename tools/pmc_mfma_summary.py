#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc MfmaUtil pass into profiles/rNN_pmc_mfma_util.json.

usage: pmc_mfma_summary.py <pmc_dir> <out.json> "<command string>"
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out, cmd = sys.argv[1:4]
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != "MfmaUtil":
                    continue
                name = row["Kernel_Name"].split("(")[0]
                if name.startswith("void "):
                    name = name[5:]
                if name.startswith("mt::"):
                    acc[name].append(float(row["Counter_Value"]))
    kernels = {k: {"launches": len(v), "mfma_util_pct_mean": round(sum(v) / len(v), 2), "mfma_util_pct_max": round(max(v), 2)}
               for k, v in sorted(acc.items())}
    json.dump({"command": cmd,
               "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100, per dispatch; mean and max over the launches of each kernel",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(k, v)


if __name__ == "__main__":
    main()
