"""Development aid: distribution of executed iterations per problem for two secondary workloads."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench_configs as bc
for name in ("cilqr16384", "car16384"):
    opt, desc = bc.build(name)
    opt()
    it = np.asarray(opt.iterations)
    print(name, "mean", it.mean(), "max", it.max(), "pcts", np.percentile(it, [50, 90, 99, 99.9]).tolist(), "hist/10", np.bincount(it // 10).tolist(), flush=True)
