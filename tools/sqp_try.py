"""Development aid: the batched sequential-QP solver against the numpy oracle on a few problems."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd import problems, systems
from ilqr_amd.constraints import BubbleConstraint
from ilqr_amd.sqp import sequential_QP_optimizer
from oracle import sqp_oracle as so
import torch

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 80
p = problems.sqp_batch(B, N)
car = systems.Car(); car.set_dt(0.2); car.set_cost(problems.SQP_Q.copy(), problems.SQP_R.copy()); car.Q_f = car.Q * N / 100
car.set_control_limit(*problems.SQP_LIMITS)
con = BubbleConstraint(N); con.setup(p["centers"], p["radius"], p["vel_bounds"])
opt = sequential_QP_optimizer(car, con, p["target_states"], 0.2, history=True)
opt.set_init_inputs(p["init_inputs"].copy())
t = time.time(); opt(); torch.cuda.synchronize(); print("first call %.3f s" % (time.time() - t))
opt.set_init_inputs(p["init_inputs"].copy()); opt.upload(); torch.cuda.synchronize(); t = time.time(); opt.solve_device(); torch.cuda.synchronize(); print("solve %.4f s" % (time.time() - t)); opt.download()
print("iters", opt.iterations[:8], "status", np.bincount(opt.status), "qp iters", opt.qp_iterations[:8], "cost", opt.min_cost[:4])
for b in range(min(B, 4)):
    c1 = BubbleConstraint(N); c1.setup(p["centers"][b], p["radius"][b], p["vel_bounds"])
    r = so.sqp_solve(car, c1, p["target_states"][b], p["init_inputs"][b], qp="riccati")
    n = min(r["iters"], int(opt.iterations[b]))
    print(b, "oracle iters", r["iters"], "gpu", opt.iterations[b], "rel cost diff", np.abs(opt.cost_history[b, :n] - r["hist_cost"][:n]) / r["hist_cost"][:n],
          "alpha", opt.alpha_history[b, :n], r["hist_alpha"][:n], "states diff", np.abs(opt.states[b] - r["states"]).max())
