cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_shard_sizes.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t1.log 2>&1; rc=$?; tail -5 gpurun_out/t1.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 > gpurun_out/b_auto.json 2>gpurun_out/b_auto.err && \
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --mapping 2,3 > gpurun_out/b_23.json 2>gpurun_out/b_23.err && \
ILQR_NO_LDS_RING=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --mapping 2,3 > gpurun_out/b_23_noring.json 2>gpurun_out/b_23_noring.err
python3 - <<'PY'
import json
for n in ('b_auto','b_23','b_23_noring'):
    b=json.load(open(f'gpurun_out/{n}.json'))
    print(n, round(b['value']), round(b['ms_per_step'],3), {k:round(v['avg_ms'],4) for k,v in b['kernels'].items()})
PY
