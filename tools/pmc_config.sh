#!/bin/bash
# usage: tools/pmc_config.sh <config> <tag> "<counters>"   (GPU box, from the repo root)
# one rocprofv3 counter pass over one solve of a tools/bench_configs.py configuration
cfg=$1; tag=$2; ctr=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/pmc_${cfg}_$tag -o p -- python3 $root/tools/bench_configs.py --only $cfg --reps 1 > $root/gpurun_out/pmc_${cfg}_$tag.json 2> $root/gpurun_out/pmc_${cfg}_$tag.err
