#!/usr/bin/env python3
"""What each solver kernel of a bench.py step is bound by, from a rocprofv3 SQ counter pass
(SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY, SQ_INSTS_VALU, SQ_WAVES in one run):

    python tools/pmc_limits.py <sq_dir> [<flops_dir>] [out.json]

Per kernel (all template variants together, launches that left at once dropped): the fraction of
its wave cycles in which a wave issued a vector-ALU instruction (`valu_busy`), the fraction in
which it waited for an outstanding instruction (`wait_inst`), and, with the second directory
(SQ_INSTS_VALU_{ADD,MUL,FMA}_F64), the share of its vector instructions that are fp64 arithmetic.
`limiter` is "fp64 vector issue" when the waves issue vector instructions in 60 % or more of their
cycles (with one wave per SIMD nothing else can fill the rest), else "HBM bandwidth".  Stamped with
the hash of the kernel source, like tools/pmc_traffic.py."""
import collections
import csv
import glob
import json
import os
import sys

from pmc_traffic import NAMES, source_stamp


def totals(directory):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> value
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            key = next((k for sub, k in NAMES if sub in r["Kernel_Name"]), None)
            if key:
                acc[(key, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


def main():
    args = [a for a in sys.argv[1:] if not a.endswith(".json")]
    out = next((a for a in sys.argv[1:] if a.endswith(".json")), None)
    sq = totals(args[0])
    fl = totals(args[1]) if len(args) > 1 else {}
    res = {}
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for (key, _), c in sq.items():
        if c.get("SQ_INSTS_VALU", 0.0) < 1e5:   # a launch that left at once
            continue
        for name, v in c.items():
            by[key][name] += v
        by[key]["launches"] += 1
    fby = collections.defaultdict(lambda: collections.defaultdict(float))
    for (key, _), c in fl.items():
        if c.get("SQ_INSTS_VALU", 0.0) < 1e5:
            continue
        for name, v in c.items():
            fby[key][name] += v
    for key, c in by.items():
        cyc = c.get("SQ_WAVE_CYCLES", 0.0)
        if cyc <= 0:
            continue
        busy = c.get("SQ_ACTIVE_INST_VALU", 0.0) / cyc
        res[key] = {"launches": int(c["launches"]), "valu_busy": busy, "wait_inst": c.get("SQ_WAIT_INST_ANY", 0.0) / cyc,
                    "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0.0) / max(1.0, c.get("SQ_WAVES", 0.0)),
                    "limiter": "fp64 vector issue" if busy >= 0.6 else "HBM bandwidth"}
        f = fby.get(key)
        if f and f.get("SQ_INSTS_VALU", 0.0) > 0:
            res[key]["fp64_arith_share_of_valu"] = (f.get("SQ_INSTS_VALU_ADD_F64", 0.0) + f.get("SQ_INSTS_VALU_MUL_F64", 0.0)
                                                    + f.get("SQ_INSTS_VALU_FMA_F64", 0.0)) / f["SQ_INSTS_VALU"]
    res["_source"] = {"kernels_hpp": source_stamp()}
    txt = json.dumps(res, indent=1)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
