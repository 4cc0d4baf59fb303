import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, conftest as cf
from ilqr_amd import problems
from ilqr_amd.ilqr import iterative_LQR
p = problems.car_batch(65536, 200)
car = cf.demo_car_system()
opt = iterative_LQR(car, p["target_states"], car.dt)
opt.inputs, opt.states = p["init_inputs"], p["init_states"]
opt()
it = opt.iterations
act = [(it > i).sum() for i in range(30)]
print("active at iteration i:", act)
print("sum", sum(act), "mean", sum(act)/65536)
