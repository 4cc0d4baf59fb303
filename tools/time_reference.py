#!/usr/bin/env python3
"""Time the REAL reference (``/root/reference/ilqr.py`` + ``systems.py``, imported by path) and the C
oracle on the same seeded Car N=200 problems of the benchmark workload, one process per core.
Runs in the build container only (the reference does not exist on the GPU box); its output
``profiles/r02/cpu_reference.json`` is what ``bench.py`` quotes as the calibration between the
C port it can time on the GPU box's host cores and the numpy reference.

    python tools/time_reference.py [--problems 64] [--procs 8]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _reference_worker(args):
    first, count, N = args
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REF)
    import ilqr as ref_ilqr          # the reference
    import systems as ref_systems    # the reference
    from ilqr_amd import problems
    car = ref_systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    car.set_control_limit(*problems.CAR_LIMITS)
    p = problems.car_batch(count, N, first=first)
    t0 = time.perf_counter()
    costs, iters = [], []
    for b in range(count):
        opt = ref_ilqr.iterative_LQR(car, p["target_states"][b], car.dt)
        opt.inputs = p["init_inputs"][b].copy()
        opt.states = p["init_states"][b].copy()
        calls = [0]
        bp = opt.backward_pass

        def counted():
            calls[0] += 1
            bp()

        opt.backward_pass = counted
        opt()
        costs.append(float(opt.min_cost))
        iters.append(calls[0])
    return time.perf_counter() - t0, costs, iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=64)
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "cpu_reference.json"))
    a = ap.parse_args()
    N, P = a.horizon, a.procs
    per = (a.problems + P - 1) // P
    total = per * P
    jobs = [(w * per, per, N) for w in range(P)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(P) as pool:
        res = pool.map(_reference_worker, jobs)
    wall_ref = time.perf_counter() - t0
    core_s_ref = sum(r[0] for r in res)
    ref_costs = np.concatenate([r[1] for r in res])
    ref_iters = np.concatenate([r[2] for r in res])

    # the C oracle on the same problems: single thread (per-core figure) and all cores
    sys.path.insert(0, ROOT)
    from ilqr_amd import problems, systems
    from oracle import oracle
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    car.set_control_limit(*problems.CAR_LIMITS)
    p = problems.car_batch(total, N)
    d = oracle.make_desc(car, N)
    oracle.lib()
    reps = 8                              # the port is fast: repeat to get a stable time
    t0 = time.perf_counter()
    for _ in range(reps):
        r1 = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy(), nthreads=1, history=False)
    t_port_1 = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy(), nthreads=P, history=False)
    t_port_P = (time.perf_counter() - t0) / reps
    rel = float(np.max(np.abs(r1["min_cost"] - ref_costs) / np.abs(ref_costs)))
    out = {
        "what": "reference ilqr.py (numpy) and the C oracle on the same seeded Car problems (ilqr_amd.problems.car_batch, first 0)",
        "where": "build container", "cpu": _cpu_name(), "procs": P, "problems": total, "horizon": N,
        "reference": {"solves_per_s_per_core": total / core_s_ref, "solves_per_s_all_procs": total / wall_ref,
                      "core_seconds": core_s_ref, "wall_seconds": wall_ref, "mean_iterations": float(ref_iters.mean())},
        "c_port": {"solves_per_s_per_core": total / t_port_1, "solves_per_s_all_procs": total / t_port_P,
                   "mean_iterations": float(r1["iters"].mean())},
        "port_over_reference_per_core": (total / t_port_1) / (total / core_s_ref),
        "max_rel_cost_difference": rel,
    }
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
