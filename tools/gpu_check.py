"""Development aid: runs the GPU path against the oracle on a few fixtures and prints the
deviations (used while bringing kernels up; the real checks live in tests/)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as cf  # noqa: E402
from oracle import oracle  # noqa: E402
from ilqr_amd.ilqr import iterative_LQR  # noqa: E402
from ilqr_amd import problems  # noqa: E402


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def main():
    for name in cf.golden_names():
        g = cf.load_golden(name)
        s = cf.system_from_golden(g)
        N = g["target_states"].shape[0]
        opt = iterative_LQR(s, np.array(g["target_states"]), s.dt, history=True)
        opt.inputs = np.array(g["init_inputs"])
        opt.states = np.array(g["init_states"])
        # phase-by-phase against the oracle on the first iteration
        dev = opt.upload()
        dev.initial_cost()
        J0 = float(dev.J.cpu()[0])
        d = oracle.make_desc(s, N)
        tg = np.ascontiguousarray(g["target_states"])
        st = np.array(g["init_states"]); ui = np.array(g["init_inputs"])
        c0 = oracle.cost(d, tg, st, ui)
        dev.backward()
        K, k = dev.download_K()[0], dev.download_k()[0][:, :, 0]
        Ko, ko, _ = oracle.backward(d, tg, st, ui)
        print(f"{name}: cost0 rel {abs(J0 - c0) / c0:.2e}  K rel {rel(K, Ko):.2e}  k rel {rel(k, ko):.2e}")
        dev.forward(0)
        acc, mc, cv, tries = oracle.forward(d, tg, st, ui, Ko, ko, c0)
        print(f"   fwd: alpha gpu {int(dev.alpha_idx.cpu()[0])} oracle {acc}  cost rel {abs(float(dev.J.cpu()[0]) - mc) / mc:.2e}"
              f"  X rel {rel(dev.download_states()[0], st):.2e} U rel {rel(dev.download_inputs()[0], ui):.2e}")
        # whole solve against the golden
        opt = iterative_LQR(s, np.array(g["target_states"]), s.dt, history=True)
        opt.inputs = np.array(g["init_inputs"])
        opt.states = np.array(g["init_states"])
        opt()
        it = int(g["iters"])
        n = min(it, opt.iterations)
        hc = opt.cost_history[:n]
        print(f"   solve: iters gpu {opt.iterations} ref {it} conv {opt.converge}/{bool(g['converge'])} "
              f"cost rel max {np.max(np.abs(hc - g['hist_cost'][:n]) / np.abs(g['hist_cost'][:n])):.2e} "
              f"alpha mismatch at {list(np.nonzero(opt.alpha_history[:n] != g['hist_alpha'][:n])[0])} status {opt.status}")
    # batch
    g = cf.load_golden("car_N200_batch32")
    s = cf.system_from_golden(g)
    opt = iterative_LQR(s, np.array(g["target_states"]), s.dt, history=True)
    opt.inputs = np.array(g["init_inputs"]); opt.states = np.array(g["init_states"])
    opt()
    print("batch32 iters gpu", list(opt.iterations), "\n        ref", list(g["iters"]))
    print("batch32 final cost rel", np.max(np.abs(opt.min_cost - g["min_cost"]) / g["min_cost"]))
    # timing
    import torch
    for B in (65536,):
        t0 = time.time(); p = problems.car_batch(B, 200); t1 = time.time()
        car = cf.demo_car_system()
        ref = None
        for L, A, ns, rep in ((2, 8, 1, -1), (2, 8, 1, 2), (2, 8, 1, 3), (2, 8, 1, 4), (2, 8, 1, 6), (2, 8, 2, -1), (2, 8, 2, 3), (2, 8, 2, 4), (2, 8, 2, 6)):
            if True:
                opt = iterative_LQR(car, p["target_states"], car.dt, lanes_per_problem=L, alphas_per_lane=A, n_streams=ns, repack_every=rep)
                opt.inputs = p["init_inputs"].copy(); opt.states = p["init_states"].copy()
                dev = opt.upload(); torch.cuda.synchronize()
                X0, U0 = dev.X.clone(), dev.U.clone()
                best = 1e9
                for rep in range(3):
                    dev.X.copy_(X0); dev.U.copy_(U0); dev.converged.zero_(); dev.iters.zero_(); torch.cuda.synchronize()
                    t2 = time.time(); opt.solve_device(); torch.cuda.synchronize(); t3 = time.time()
                    best = min(best, t3 - t2)
                J = dev.J.cpu().numpy()
                if ref is None:
                    ref = J
                print(f"B={B} L={L} A={A} streams={ns} repack={rep}: solve {1e3 * best:.1f} ms -> {B / best:.0f} solves/s, same bits as first: {np.array_equal(J, ref)}")


if __name__ == "__main__":
    main()
