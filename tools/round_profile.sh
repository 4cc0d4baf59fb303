#!/bin/bash
# Run on the GPU box from the repo root: the bench line, the rocprofv3 per-kernel statistics of the
# same command, and the counter passes (HBM traffic, wave statistics).  Results under gpurun_out/round/.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/round
mkdir -p $out
timeout -k 10 300 python3 $root/bench.py > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err || exit 1
cp $out/stats/p_kernel_stats.csv $out/kernel_stats.csv
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "flops SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES" "occ OccupancyPercent" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES"; do
    set -- $pass; tag=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$tag -o p -- python3 $root/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $out/pmc_$tag.err || exit 1
done
cd $root
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/hbm_traffic.json > /dev/null || exit 1
python3 tools/pmc_summary.py $out/pmc_sq $out/pmc_flops $out/pmc_occ > $out/pmc_summary.txt || exit 1
python3 tools/pmc_limits.py $out/pmc_sq $out/pmc_flops $out/kernel_limits.json > /dev/null || exit 1
cat $out/bench.json
