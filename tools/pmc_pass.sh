#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" <bench args...>   (GPU box, from the repo root)
# one rocprofv3 counter pass over a short bench.py run; csv under gpurun_out/pmc_<tag>/
tag=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/pmc_$tag -o p -- python3 $root/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $root/gpurun_out/pmc_$tag.json 2> $root/gpurun_out/pmc_$tag.err
