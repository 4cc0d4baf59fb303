#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
collected in separate runs, as the TCC block cannot hold both).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]

Units and corrections (gfx950): both counters are in KiB; FETCH_SIZE counts 64 B per 128 B
request, so a coalesced streaming read reports exactly half its bytes and is doubled here
(checked in the same run on cost_kernel, a pure streaming read of X, U and Xref with the access
shape of all solver kernels: 8 B per lane, 512 B per wave instruction); WRITE_SIZE is exact for
these stores (checked on relayout_kernel).  Reported per launch, as the mean over the launches
of one bench.py step that did work (all template variants of a kernel together), under the names
bench.py uses.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_stamp():
    """git's blob hash of the kernel source the counters were taken from: bench.py quotes a profile
    only while the file it was taken from is the one in the tree."""
    data = open(os.path.join(ROOT, "ilqr_amd", "csrc", "kernels.hpp"), "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()

NAMES = (("repack_retire_kernel", "retire"), ("backward_kernel", "backward"), ("backward_gen_kernel", "backward"), ("states_kernel", "states"), ("repack_move_kernel", "repack"), ("evaluate_listed_kernel", "evaluate_rest"), ("evaluate_kernel", "evaluate_first"), ("evaluate_ring_kernel", "evaluate_first"),
         ("commit_kernel", "commit"), ("cost_kernel", "cost"), ("relayout_kernel", "relayout"))


def per_launch(directory, counter, scale):
    acc = collections.defaultdict(float)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            key = next((k for sub, k in NAMES if sub in r["Kernel_Name"]), None)
            if key:
                acc[(key, r["Dispatch_Id"])] += float(r["Counter_Value"]) * scale
    out = collections.defaultdict(list)
    for (key, _), v in acc.items():
        out[key].append(v)
    return out


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rd = per_launch(fetch_dir, "FETCH_SIZE", 2.0 * 1024.0)
    wr = per_launch(write_dir, "WRITE_SIZE", 1024.0)
    res = {}
    for key in rd:
        # launches that leave at once (an idle line-search mapping, an empty list) are not counted
        keep = [i for i, v in enumerate(rd[key]) if v > 1e6] or list(range(len(rd[key])))
        r = [rd[key][i] for i in keep]
        w = [wr[key][i] for i in keep] if key in wr and len(wr[key]) == len(rd[key]) else wr.get(key, [0.0])
        res[key] = {"launches": len(r), "read_bytes_per_launch": sum(r) / len(r), "write_bytes_per_launch": sum(w) / len(w),
                    "bytes_per_launch": sum(r) / len(r) + sum(w) / len(w), "max_bytes_per_launch": max(r) + max(w)}
    res["_source"] = {"kernels_hpp": source_stamp()}
    txt = json.dumps(res, indent=1)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
