#!/bin/bash
# usage: tools/prof_stats.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# per-kernel statistics of one bench.py run, written to gpurun_out/prof_<tag>/ and printed
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o p -- python3 $root/bench.py --no-cpu-baseline "$@" > $root/gpurun_out/prof_$tag.json 2> $root/gpurun_out/prof_$tag.err || exit 1
python3 - "$root/gpurun_out/prof_$tag/p_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3:
        print(f'{r["Name"][:100]:100s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} pct={r["Percentage"]}')
PY
