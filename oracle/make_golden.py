"""Generate the committed golden fixtures under ``tests/golden/`` by RUNNING THE
REFERENCE ITSELF (``/root/reference/ilqr.py`` + ``systems.py``, imported by
path; numpy only).  Runs in the build container only: the reference does not
exist on the GPU box, the fixtures (inputs + expected outputs, plain arrays)
are what travels.

    python oracle/make_golden.py            # writes tests/golden/*.npz

Each single-problem fixture stores the inputs handed to the reference solver
and, per executed iteration, ``min_cost``, the number of rollouts its serial
line search made, the accepted step index (-1: none) and snapshots of
``states`` / ``inputs``; plus the gains of the first backward pass.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import ilqr as ref_ilqr  # noqa: E402  (the reference)
import systems as ref_systems  # noqa: E402  (the reference)

from ilqr_amd import problems  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run_reference(system, targets, init_inputs, x0=None, max_iter=None, snapshots=True,
                  init_states=None):
    """Drive the reference solver exactly as ``ilqr.py:107-114`` does, recording
    after every iteration."""
    opt = ref_ilqr.iterative_LQR(system, targets, system.dt)
    opt.inputs = np.array(init_inputs, dtype=float, copy=True)
    if x0 is not None:
        opt.states[0, :] = x0
    if init_states is not None:
        opt.states = np.array(init_states, dtype=float, copy=True)
    if max_iter is not None:
        opt.maxIter = max_iter
    n_cost_calls = [0]
    orig_cost = opt.cost

    def counting_cost():
        n_cost_calls[0] += 1
        return orig_cost()

    opt.cost = counting_cost
    rec = dict(hist_cost=[], hist_tries=[], hist_alpha=[], hist_states=[], hist_inputs=[])
    opt.min_cost = opt.cost()
    rec["cost0"] = opt.min_cost
    iters = 0
    for it in range(opt.maxIter):
        if opt.converge:
            break
        opt.backward_pass()
        if it == 0:
            rec["K_first"] = opt.K.copy()
            rec["k_first"] = opt.k[:, :, 0].copy()
        before = opt.min_cost
        n_cost_calls[0] = 0
        opt.forward_pass()
        tries = n_cost_calls[0]
        accepted = opt.min_cost < before
        rec["hist_cost"].append(opt.min_cost)
        rec["hist_tries"].append(tries)
        rec["hist_alpha"].append(tries - 1 if accepted else -1)
        if snapshots:
            rec["hist_states"].append(opt.states.copy())
            rec["hist_inputs"].append(opt.inputs.copy())
        iters += 1
    rec.update(iters=iters, converge=bool(opt.converge), min_cost=opt.min_cost,
               states=opt.states.copy(), inputs=opt.inputs.copy())
    for key in ("hist_cost", "hist_tries", "hist_alpha", "hist_states", "hist_inputs"):
        rec[key] = np.array(rec[key])
    return rec


def sys_arrays(system):
    out = dict(model=type(system).__name__, dt=system.dt, Q=np.array(system.Q, dtype=float),
               R=np.array(system.R, dtype=float), Q_f=np.array(system.Q_f, dtype=float),
               control_limited=bool(system.control_limited))
    if system.control_limited:
        out["lo"] = np.array(system.control_lower_limit, dtype=float)
        out["hi"] = np.array(system.control_upper_limit, dtype=float)
    return out


def make_car(limited=True):
    car = ref_systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    if limited:
        car.set_control_limit(*problems.CAR_LIMITS)
    return car


def save(name, system, targets, init_inputs, x0, rec, init_states=None):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    x0 = np.zeros(system.state_size) if x0 is None else x0
    if init_states is None:  # the reference default: zeros, x0 at step 0
        init_states = np.zeros_like(targets)
        init_states[0] = x0
    np.savez_compressed(path, target_states=targets, init_inputs=init_inputs, x0=x0,
                        init_states=init_states, **sys_arrays(system), **rec)
    print(f"{name}: iters={rec['iters']} converge={rec['converge']} "
          f"cost0={rec['cost0']:.6e} min_cost={rec['min_cost']:.9e} "
          f"tries={list(rec['hist_tries'])}  -> {os.path.getsize(path) / 1024:.0f} KiB")


def make_bicycle():
    from ilqr_amd.user_models import KinematicBicycle
    s = KinematicBicycle()
    s.set_cost(problems.BICYCLE_Q.copy(), problems.BICYCLE_R.copy())
    s.set_control_limit(*problems.BICYCLE_LIMITS)
    return s


def make_pendulum():
    from ilqr_amd.user_models import Pendulum
    s = Pendulum()
    s.set_cost(problems.PENDULUM_Q.copy(), problems.PENDULUM_R.copy())
    s.set_control_limit(*problems.PENDULUM_LIMITS)
    return s


def user_models():
    """Models outside the reference's systems.py: the reference SOLVER (ilqr.py) runs on their
    numpy methods through its System duck type; the fixtures pin the run-time compiled device
    dynamics of the same classes."""
    for name, system, gen, picks in (("bicycle_N150", make_bicycle(), problems.bicycle_batch, (0, 2, 5)),
                                     ("pendulum_N120", make_pendulum(), problems.pendulum_batch, (0, 1))):
        p = gen(max(picks) + 1)
        for b in picks:
            rec = run_reference(system, p["target_states"][b], p["init_inputs"][b], p["x0"][b])
            save(f"{name}_seed{b}", system, p["target_states"][b], p["init_inputs"][b], p["x0"][b], rec)


def main():
    # Car, the demo construction (ilqr_demo.py:9-49), N=200 x 3 seeds and the demo's N=120
    for N, seed in ((200, 0), (200, 1), (200, 2), (120, 0)):
        tg, ui, _ = problems.demo_car(N, seed)
        car = make_car()
        save(f"car_N{N}_seed{seed}", car, tg, ui, None, run_reference(car, tg, ui))
    # CarAcceleration demo (ilqr_demo.py:89-135)
    tg, ui = problems.demo_caracc(100, 0)
    s = ref_systems.CarAcceleration()
    s.set_dt(0.2)
    s.set_cost(problems.CARACC_Q.copy(), problems.CARACC_R.copy())
    s.set_control_limit(*problems.CARACC_LIMITS)
    save("caracc_N100_seed0", s, tg, ui, None, run_reference(s, tg, ui))
    # DubinsCar demo (ilqr_demo.py:170-192), inputs start at zero
    tg, ui = problems.demo_dubins(200, 0)
    s = ref_systems.DubinsCar()
    s.set_dt(0.2)
    s.set_cost(problems.DUBINS_Q.copy(), problems.DUBINS_R.copy())
    s.set_control_limit(*problems.DUBINS_LIMITS)
    save("dubins_N200_seed0", s, tg, ui, None, run_reference(s, tg, ui))
    # non-zero initial state: batched generator, problem 7, once with only states[0] set
    # (everything else zero, the reference default) and once with the generator's initial guess
    p = problems.car_batch(1, 200, first=7)
    car = make_car()
    save("car_N200_x0", car, p["target_states"][0], p["init_inputs"][0], p["x0"][0],
         run_reference(car, p["target_states"][0], p["init_inputs"][0], p["x0"][0]))
    save("car_N200_x0guess", car, p["target_states"][0], p["init_inputs"][0], p["x0"][0],
         run_reference(car, p["target_states"][0], p["init_inputs"][0],
                       init_states=p["init_states"][0]), init_states=p["init_states"][0])
    tg, ui, _ = problems.demo_car(200, 3)
    car = make_car(limited=False)
    save("car_N200_nolimit", car, tg, ui, None, run_reference(car, tg, ui))
    user_models()
    # 32 distinct problems of the batched generator: final results + cost history only
    B = 32
    p = problems.car_batch(B, 200)
    car = make_car()
    recs = [run_reference(car, p["target_states"][b], p["init_inputs"][b], snapshots=False,
                          init_states=p["init_states"][b]) for b in range(B)]
    mi = 30
    hc = np.full((B, mi), np.nan)
    ha = np.full((B, mi), -2, dtype=np.int32)
    ht = np.zeros((B, mi), dtype=np.int32)
    for b, r in enumerate(recs):
        hc[b, :r["iters"]] = r["hist_cost"]
        ha[b, :r["iters"]] = r["hist_alpha"]
        ht[b, :r["iters"]] = r["hist_tries"]
    path = os.path.join(OUT, "car_N200_batch32.npz")
    np.savez_compressed(
        path, target_states=p["target_states"], init_inputs=p["init_inputs"], x0=p["x0"],
        init_states=p["init_states"],
        **sys_arrays(car), cost0=np.array([r["cost0"] for r in recs]),
        min_cost=np.array([r["min_cost"] for r in recs]),
        converge=np.array([r["converge"] for r in recs]),
        iters=np.array([r["iters"] for r in recs], dtype=np.int32),
        states=np.array([r["states"] for r in recs]), inputs=np.array([r["inputs"] for r in recs]),
        hist_cost=hc, hist_alpha=ha, hist_tries=ht)
    print("car_N200_batch32: iters", [r["iters"] for r in recs],
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
