#!/usr/bin/env python3
"""Whole-solve timings of the secondary workloads (the ones bench.py does not quote): one JSON
line per configuration, inputs resident in HBM, time = ilqr_solve on the device, median of reps.

    python tools/bench_configs.py [--reps 3] [--only car4096,car8192,car16384,car32768,cilqr16384,car65536_f32,ltv_f64,ltv_f32,ltv_f64_2s,ltv_f32_2s,cilqr_ltv_f64,cilqr_ltv_f32,sqp16384,sqp65536]

car8192 / car16384 / car32768 are the per-GPU shards of the 65536-problem headline batch on 8 / 4 / 2 GPUs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd import problems, systems  # noqa: E402
from ilqr_amd.cilqr import iterative_LQR as CILQR  # noqa: E402
from ilqr_amd.constraints import BubbleConstraint  # noqa: E402
from ilqr_amd.ilqr import iterative_LQR  # noqa: E402


def car_system(qf_scale=None):
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    car.set_control_limit(*problems.CAR_LIMITS)
    if qf_scale is not None:
        car.Q_f = car.Q * qf_scale
    return car


def build(name):
    if name in ("car4096", "car8192", "car16384", "car32768", "car65536_f32"):
        B = {"car4096": 4096, "car8192": 8192, "car16384": 16384, "car32768": 32768, "car65536_f32": 65536}[name]
        p = problems.car_batch(B, 200)
        opt = iterative_LQR(car_system(), p["target_states"], 0.2, dtype=np.float32 if name.endswith("f32") else np.float64)
        opt.inputs, opt.states = p["init_inputs"], p["init_states"]
        return opt, f"Car n=4 m=2, B={B}, N=200, maxIter=30, {'fp32' if name.endswith('f32') else 'fp64'}"
    if name == "cilqr16384":
        B = 16384
        p = problems.car_batch(B, 200, noise_xy=15.0)
        centers, radius, vb = problems.bubbles_for(p["target_states"], radius=6.0, vel_bounds=(-1.0, 14.0))
        con = BubbleConstraint(200)
        con.setup(centers, radius, vb)
        opt = CILQR(car_system(2.0), p["target_states"], 0.2)
        opt.inputs, opt.states = p["init_inputs"], p["init_states"]
        opt.set_barrier(con, mu=50.0, delta=0.5)
        return opt, "cilqr: Car, box QP + relaxed log barrier (bubbles, speed box), B=16384, N=200, maxIter=100, 8 step sizes, fp64"
    if name in ("ltv_f64", "ltv_f32", "ltv_f64_2s", "ltv_f32_2s"):
        B, N = 32768, 500
        A, Bm, Q, R = problems.ltv_system(N=N)
        s = systems.LTVSystem(A, Bm, dt=0.05)
        s.set_cost(Q, R)
        s.set_control_limit([-2.0] * 4, [2.0] * 4)
        p = problems.ltv_batch(B, N)
        # "_2s": two halves of the batch on two streams (ilqr_desc.n_streams): the commit, bound by bandwidth, overlaps
        # a little with the other half's backward pass and line search, bound by the matrix and vector units
        # (fp64 74.8 -> 72.6 ms; nothing in fp32)
        ns = 2 if name.endswith("_2s") else 1
        opt = iterative_LQR(s, p["target_states"], 0.05, dtype=np.float32 if "f32" in name else np.float64, n_streams=ns)
        opt.inputs, opt.states = p["init_inputs"], p["init_states"]
        return opt, f"LTV n=12 m=4, B={B}, N={N}, maxIter=30, {'fp32' if 'f32' in name else 'fp64'}, n_streams={ns}"
    if name in ("cilqr_ltv_f64", "cilqr_ltv_f32"):
        # cilqr.py:87-139 (box QP in the backward pass, 8 step sizes, maxIter 100) on the table model
        B, N = 32768, 500
        A, Bm, Q, R = problems.ltv_system(N=N)
        s = systems.LTVSystem(A, Bm, dt=0.05)
        s.set_cost(Q, R)
        s.set_control_limit([-2.0] * 4, [2.0] * 4)
        p = problems.ltv_batch(B, N)
        opt = CILQR(s, p["target_states"], 0.05, dtype=np.float32 if name.endswith("f32") else np.float64)
        opt.inputs, opt.states = p["init_inputs"], p["init_states"]
        return opt, f"cilqr: LTV n=12 m=4, box QP, B={B}, N={N}, maxIter=100, 8 step sizes, {'fp32' if name.endswith('f32') else 'fp64'}"
    if name in ("sqp16384", "sqp65536"):
        from ilqr_amd.sqp import sequential_QP_optimizer
        B, N = int(name[3:]), 80
        p = problems.sqp_batch(B, N)
        car = systems.Car()
        car.set_dt(0.2)
        car.set_cost(problems.SQP_Q.copy(), problems.SQP_R.copy())
        car.Q_f = car.Q * N / 100
        car.set_control_limit(*problems.SQP_LIMITS)
        con = BubbleConstraint(N)
        con.setup(p["centers"], p["radius"], p["vel_bounds"])
        opt = sequential_QP_optimizer(car, con, p["target_states"], 0.2)
        opt.set_init_inputs(p["init_inputs"])
        return opt, f"sqp: Car, bubbles + speed box + input box, B={B}, N={N}, maxIter=20, fp64"
    raise SystemExit(f"unknown configuration {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="car4096,car8192,car16384,car32768,cilqr16384,car65536_f32,ltv_f64,ltv_f32,sqp16384")
    a = ap.parse_args()
    import torch
    for name in a.only.split(","):
        opt, what = build(name)
        dev = opt.upload()
        X0, U0 = dev.X.clone(), dev.U.clone()
        times = []
        for _ in range(a.reps + 1):
            dev.X.copy_(X0); dev.U.copy_(U0); dev.converged.zero_(); dev.iters.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opt.solve_device()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t = float(np.median(times[1:]))
        iters = dev.iters.cpu().numpy()[:opt._B]
        conv = dev.converged.cpu().numpy()[:opt._B]
        print(json.dumps({"config": name, "workload": what, "ms_per_solve": 1e3 * t, "solves_per_s": opt._B / t,
                          "mean_iterations": float(iters.mean()), "converged_fraction": float(conv.mean()),
                          "forward_mapping": dev.forward_mapping()}), flush=True)
        del opt, dev, X0, U0
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
