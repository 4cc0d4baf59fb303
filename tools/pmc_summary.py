#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel per launch.

    python tools/pmc_summary.py gpurun_out/prof_r01_fetch gpurun_out/prof_r01_write ...
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"ilqr::(\w+)<([^>]*?)>", name)
    if m:
        return f"{m.group(1)}<{m.group(2).replace('ilqr::', '').replace('CarModel<double>', 'Car').replace('CarModel<float>', 'Car')}>"
    if "rocclr_copyBuffer" in name:
        return "copyBuffer"
    m = re.search(r"at::native::(\w+)", name)
    return "torch:" + (m.group(1) if m else name[:40])


def summarize(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


if __name__ == "__main__":
    acc = summarize(sys.argv[1:])
    for k in sorted(acc):
        parts = [f"{c}: n={len(v)} mean={sum(v) / len(v):.6g} max={max(v):.6g}" for c, v in sorted(acc[k].items())]
        print(k)
        for p in parts:
            print("    " + p)
