#!/bin/bash
# Run on the GPU box from the repo root: the GPU test suite, then the default bench line.
# Every step has its own time limit and the script stops at the first failure.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > $out/gputest.log 2>&1
rc=$?
tail -15 $out/gputest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > $out/bench_check.json 2> $out/bench_check.err || exit 1
cat $out/bench_check.json
