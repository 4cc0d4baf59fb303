#!/usr/bin/env python3
"""Whole-solve time of Car batches below the bandwidth-bound regime under the scheduling knobs that
change no result (forward mapping, streams, recomputing the previous states): one JSON line per
combination, time = ilqr_solve on the device, median of reps.

    python tools/sweep_small.py --batches 4096,8192,16384 [--reps 3]
"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd import problems, systems  # noqa: E402
from ilqr_amd.ilqr import iterative_LQR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="4096,8192,16384")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--mappings", default="0x0,16x1,8x2,4x4,8x1,4x2")
    ap.add_argument("--streams", default="1,2,4")
    ap.add_argument("--recompute", default="0,-1")
    ap.add_argument("--fused", default="-1", help="fused_solve values, e.g. -1,1")
    ap.add_argument("--trig", type=int, default=0, help="trig_cache (0 = the library's choice)")
    ap.add_argument("--seg", type=int, default=0, help="segment_commit (0 = the library's choice)")
    ap.add_argument("--repack", type=int, default=-1, help="repack_every (-1 = never, 0 = the library's choice)")
    a = ap.parse_args()
    import torch
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    car.set_control_limit(*problems.CAR_LIMITS)
    for B in (int(v) for v in a.batches.split(",")):
        p = problems.car_batch(B, 200)
        for mp, ns, rc, fs in itertools.product(a.mappings.split(","), a.streams.split(","), a.recompute.split(","), a.fused.split(",")):
            L, A = (int(v) for v in mp.split("x"))
            opt = iterative_LQR(car, p["target_states"], 0.2, lanes_per_problem=L, alphas_per_lane=A, n_streams=int(ns),
                                recompute_states=int(rc), repack_every=a.repack, fused_solve=int(fs), trig_cache=a.trig, segment_commit=a.seg)
            opt.inputs, opt.states = p["init_inputs"], p["init_states"]
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
            print(json.dumps({"B": B, "mapping": dev.forward_mapping(), "streams": int(ns), "recompute_states": int(rc), "fused": int(fs),
                              "ms_per_solve": round(1e3 * float(np.median(times[1:])), 3)}), flush=True)
            del opt, dev, X0, U0
    torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
