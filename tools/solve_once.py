"""Development aid: one or more whole-batch ilqr_solve calls with chosen options (for rocprofv3)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as cf  # noqa: E402
from ilqr_amd import problems  # noqa: E402
from ilqr_amd.ilqr import iterative_LQR  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--horizon", type=int, default=200)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--repack", type=int, default=-1)
ap.add_argument("--mapping", default="2,8")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch  # noqa: E402
L, A = (int(v) for v in a.mapping.split(","))
p = problems.car_batch(a.batch, a.horizon)
car = cf.demo_car_system()
opt = iterative_LQR(car, p["target_states"], car.dt, lanes_per_problem=L, alphas_per_lane=A, n_streams=a.streams,
                    repack_every=a.repack)
opt.inputs, opt.states = p["init_inputs"], p["init_states"]
dev = opt.upload()
X0, U0 = dev.X.clone(), dev.U.clone()
for r in range(a.reps):
    dev.X.copy_(X0); dev.U.copy_(U0); dev.converged.zero_(); dev.iters.zero_()
    torch.cuda.synchronize()
    t0 = time.time()
    opt.solve_device()
    torch.cuda.synchronize()
    print(f"rep {r}: {1e3 * (time.time() - t0):.2f} ms", flush=True)
