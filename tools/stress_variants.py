"""Randomised check that no scheduling choice changes a bit: random horizons, batch sizes, iteration limits and
combinations of mapping, sparse states, compaction, streams, single-launch solve, sine/cosine cache and time-parallel
commit, each against the plain solve of the same problems.  (GPU; development aid, the fixed cases are in tests/.)

    python tools/stress_variants.py [trials]
"""
import sys, os, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, conftest as cf
from ilqr_amd import problems
from ilqr_amd.ilqr import iterative_LQR
rng = random.Random(12345)
car = cf.demo_car_system()
keys = ("states", "inputs", "min_cost", "iterations", "converge", "status")
bad = 0
total_it = 0
conv = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    N = rng.choice([2, 3, 5, 8, 9, 10, 16, 17, 25, 33, 40, 64, 65, 90])
    B = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, 129, 200, 300])
    p = problems.car_batch(B, N, first=rng.randrange(1000))
    if B > 3 and rng.random() < 0.5:
        p["init_states"][1] = p["target_states"][1]; p["init_inputs"][1] = 0.0
    if B > 5 and rng.random() < 0.3:
        p["init_states"][4, :, 3] += 3.0e6; p["target_states"][4, :, 3] += 3.0e6
    def run(**kw):
        o = iterative_LQR(car, p["target_states"], car.dt, history=True, **kw)
        o.inputs, o.states = p["init_inputs"].copy(), p["init_states"].copy()
        o.maxIter = rng_iter
        o()
        return o
    rng_iter = rng.choice([1, 2, 5, 12, 30])
    ref = run(recompute_states=-1, repack_every=-1, segment_commit=-1, trig_cache=-1, fused_solve=-1)
    L, A = rng.choice([(16, 1), (8, 2), (4, 4), (2, 3), (1, 6), (2, 4), (1, 8), (4, 2), (0, 0)])
    kw = dict(lanes_per_problem=L, alphas_per_lane=A, recompute_states=rng.choice([0, 1, 2]), repack_every=rng.choice([-1, 1, 2, 3, 5]),
              n_streams=rng.choice([1, 1, 2, 3]), fused_solve=rng.choice([-1, 0, 1]), trig_cache=rng.choice([-1, 0, 1]),
              segment_commit=rng.choice([-1, 0, 1]))
    try:
        o = run(**kw)
    except Exception as e:
        print("trial", trial, "N", N, "B", B, kw, "raised", repr(e)[:120]); continue
    ok = all(np.array_equal(getattr(o, k), getattr(ref, k), equal_nan=True) for k in keys) and np.array_equal(o.K, ref.K, equal_nan=True) \
        and np.array_equal(o.k, ref.k, equal_nan=True) and np.array_equal(o.cost_history, ref.cost_history, equal_nan=True)
    total_it += int(ref.iterations.sum()); conv += int(ref.converge.sum())
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, "N", N, "B", B, "maxIter", rng_iter, kw, [k for k in keys if not np.array_equal(getattr(o, k), getattr(ref, k), equal_nan=True)])
print("done, mismatches:", bad, "iterations compared:", total_it, "converged problems:", conv)
