#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
collected in separate runs, as the TCC block cannot hold both).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]

Units and corrections (gfx950): both counters are in KiB; FETCH_SIZE counts 64 B per 128 B
request, so a coalesced streaming read reports exactly half its bytes and is doubled here
(checked in the same run on cost_kernel, a pure streaming read of X, U and Xref with the access
shape of all solver kernels: 8 B per lane, 512 B per wave instruction); WRITE_SIZE is exact for
these stores (checked on relayout_kernel).  Reported per launch for the launches in which the
whole batch is active (the maximum over launches).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize  # noqa: E402


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    acc = summarize([fetch_dir, write_dir])
    res = {}
    for kernel, ctr in acc.items():
        if "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
            continue
        key = kernel.split("<")[0].replace("_kernel", "")
        rd = 2.0 * max(ctr["FETCH_SIZE"]) * 1024.0
        wr = max(ctr["WRITE_SIZE"]) * 1024.0
        entry = {"kernel": kernel, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                 "bytes_per_launch": rd + wr, "launches": len(ctr["FETCH_SIZE"]),
                 "mean_bytes_per_launch": 2048.0 * sum(ctr["FETCH_SIZE"]) / len(ctr["FETCH_SIZE"])
                 + 1024.0 * sum(ctr["WRITE_SIZE"]) / len(ctr["WRITE_SIZE"])}
        if key not in res or entry["bytes_per_launch"] > res[key]["bytes_per_launch"]:
            res[key] = entry
    txt = json.dumps(res, indent=1)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
