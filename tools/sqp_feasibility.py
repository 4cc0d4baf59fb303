#!/usr/bin/env python3
"""Why sequential-QP problems end with STATUS_QP_FAILED: are the linearised rows of the flagged QPs
really without a common point?  CPU only.  Replays the SQP iterations of the first problems of
``problems.sqp_batch`` with the numpy statement of the kernel's interior-point method
(oracle/sqp_oracle.py, same stopping rule: a collapsing step length = failure) and, for every QP,
asks an LP solver (scipy HiGHS) for the smallest uniform relaxation t of all inequality rows that
makes them feasible over the linearised dynamics:  t* > 0  <=>  the QP has no feasible point.

    python tools/sqp_feasibility.py [--problems 160] [--procs 8]      -> profiles/r02/sqp_feasibility.json
"""
import argparse
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    first, count, N = args
    from ilqr_amd import problems, systems
    from ilqr_amd.constraints import BubbleConstraint
    from oracle import sqp_oracle as so
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.SQP_Q.copy(), problems.SQP_R.copy())
    car.Q_f = car.Q * N / 100
    car.set_control_limit(*problems.SQP_LIMITS)
    umin, umax = np.array(car.control_lower_limit, float), np.array(car.control_upper_limit, float)
    p = problems.sqp_batch(count, N, first=first)
    out = []
    for b in range(count):
        con = BubbleConstraint(N)
        con.setup(p["centers"][b], p["radius"][b], p["vel_bounds"])
        target = p["target_states"][b]
        inputs = p["init_inputs"][b].copy()
        states = target.copy()
        x0 = target[0].copy()
        min_cost = np.inf
        rec = dict(problem=first + b, radius=float(p["radius"][b][0]), qps=[])
        for it in range(20):
            A, B, c, G, lo, hi = so.qp_data(car, con, target, states, inputs)
            info = []
            solved, _, its = so.solve_qp_riccati(car, target, x0, A, B, c, G, lo, hi, umin, umax, inputs, info=info)
            failed = (not np.isfinite(info[-1][2])) or info[-1][1] < 1e-8 or its >= 100
            t_star = so.min_relaxation(car, x0, A, B, c, G, lo, hi, umin, umax)
            rec["qps"].append(dict(sqp_iteration=it, ip_iterations=its, failed=bool(failed), min_relaxation=t_star))
            if failed:
                break          # the kernel flags the problem, takes a zero step and ends it
            d_u = solved - inputs
            alpha, accepted = 1.0, False
            prev_s, prev_u = states.copy(), inputs.copy()
            while True:
                inputs = prev_u + alpha * d_u
                states = so.sim(car, x0, inputs)
                cst = so.cost(car, target, states, inputs)
                if cst < min_cost:
                    min_cost, accepted = cst, True
                    break
                states, inputs = prev_s, prev_u
                if alpha < 1e-2 or abs(1 - cst / min_cost) < 1e-6:
                    break
                alpha /= 2.0
            if not accepted:
                break
        out.append(rec)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=160)
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--horizon", type=int, default=80)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "sqp_feasibility.json"))
    a = ap.parse_args()
    per = (a.problems + a.procs - 1) // a.procs
    with mp.get_context("spawn").Pool(a.procs) as pool:
        res = pool.map(worker, [(w * per, per, a.horizon) for w in range(a.procs)])
    recs = [r for part in res for r in part]
    tol = 1e-7
    qps = [q for r in recs for q in r["qps"]]
    failed = [q for q in qps if q["failed"]]
    summary = {
        "what": "first problems of ilqr_amd.problems.sqp_batch (N = %d), numpy statement of the kernel's QP method; "
                "min_relaxation = LP optimum t* (HiGHS): > 0 means the linearised rows have no common point" % a.horizon,
        "problems": len(recs), "qps": len(qps),
        "problems_ending_in_a_failed_qp": sum(1 for r in recs if r["qps"][-1]["failed"]),
        "failed_qps": len(failed),
        "failed_qps_proved_infeasible": sum(1 for q in failed if q["min_relaxation"] > tol),
        "failed_qps_that_were_feasible": [q for q in failed if not (q["min_relaxation"] > tol)],
        "feasible_qps": sum(1 for q in qps if q["min_relaxation"] <= tol),
        "feasible_qps_solved": sum(1 for q in qps if q["min_relaxation"] <= tol and not q["failed"]),
        "smallest_violation_among_failed": min([q["min_relaxation"] for q in failed], default=None),
        "failed_by_sqp_iteration": {str(k): sum(1 for q in failed if q["sqp_iteration"] == k) for k in sorted({q["sqp_iteration"] for q in failed})},
        "failure_rate_by_bubble_radius": {},
    }
    edges = [0.5, 0.6, 0.7, 0.8, 0.9]
    for lo_, hi_ in zip(edges[:-1], edges[1:]):
        sel = [r for r in recs if lo_ <= r["radius"] < hi_]
        if sel:
            summary["failure_rate_by_bubble_radius"][f"{lo_:.1f}-{hi_:.1f}"] = [sum(1 for r in sel if r["qps"][-1]["failed"]), len(sel)]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
