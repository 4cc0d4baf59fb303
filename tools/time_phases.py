#!/usr/bin/env python3
"""Time one launch of each phase (backward pass, line search, commit) of a tools/bench_configs.py
configuration on its initial data, all problems active:  python tools/time_phases.py [config]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as bc
name = sys.argv[1] if len(sys.argv) > 1 else "ltv_f64"
opt, what = bc.build(name)
dev = opt.upload()
dev.initial_cost()
for ph in ("backward", "evaluate", "commit"):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if ph == "backward": dev.backward()
        elif ph == "evaluate": dev.forward_evaluate()
        else: dev.forward_commit(0)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        if ph == "commit": break
    print(name, ph, "ms", " ".join(f"{1e3*t:.2f}" for t in ts), flush=True)
