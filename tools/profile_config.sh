#!/bin/bash
# usage: tools/profile_config.sh <config>   (GPU box, from the repo root)
# rocprofv3 per-kernel statistics of one tools/bench_configs.py configuration -> gpurun_out/prof_<config>/
cfg=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$cfg -o p -- python3 $root/tools/bench_configs.py --only $cfg > $root/gpurun_out/prof_$cfg.json 2> $root/gpurun_out/prof_$cfg.err
