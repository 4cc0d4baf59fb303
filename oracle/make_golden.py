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
from oracle import demo_problems  # noqa: E402

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
    return drive(opt, snapshots)


def drive(opt, snapshots=True):
    """The loop of ``ilqr.py:107-114`` on an existing solver object, recording after every iteration."""
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


class StepCountingLTV(ref_systems.System):
    """x+ = A_t x + B_t u expressed through the reference's time-less callbacks.  The reference
    solver calls ``model_f`` once per time step in increasing order in every rollout
    (``ilqr.py:55-63``, no early exit) and ``compute_df_du`` then ``compute_df_dx`` once per time
    step in decreasing order in every backward pass (``ilqr.py:86-90``), so three call counters
    recover the time index.  This is how the REFERENCE solver produces a fixture for the
    table-driven LTV model, which its own ``systems.py`` cannot express."""

    def __init__(self, A, B, dt):
        super().__init__(A.shape[1], B.shape[2])
        self.A, self.B, self.T = A, B, A.shape[0]
        self.dt = dt
        self.nf = self.nu = self.nx = 0

    def model_f(self, x, u):
        t = self.nf % self.T
        self.nf += 1
        return np.dot(self.A[t], x) + np.dot(self.B[t], u)

    def compute_df_du(self, x, u):
        t = self.T - 1 - self.nu % self.T
        self.nu += 1
        return self.B[t]

    def compute_df_dx(self, x, u):
        t = self.T - 1 - self.nx % self.T
        self.nx += 1
        return self.A[t]


def ltv_fixtures():
    """n = 12, m = 4 table-driven model (``problems.ltv_system`` / ``ltv_batch``) solved by the
    reference ``ilqr.py``.  The tables are not stored (they follow from the seeded generator); their
    sums are, so a test notices if the generator changes."""
    for N, picks, snapshots in ((120, (0, 3), True), (500, (1,), False)):
        A, Bm, Q, R = problems.ltv_system(N=N)
        p = problems.ltv_batch(max(picks) + 1, N)
        for b in picks:
            s = StepCountingLTV(A, Bm, 0.05)
            s.set_cost(Q.copy(), R.copy())
            s.set_control_limit([-3.0] * 4, [3.0] * 4)
            rec = run_reference(s, p["target_states"][b], p["init_inputs"][b], p["x0"][b], snapshots=snapshots)
            assert s.nf % s.T == 0 and s.nu % s.T == 0 and s.nx == s.nu
            if not snapshots:
                rec.pop("hist_states"), rec.pop("hist_inputs")
            rec["table_sums"] = np.array([A.sum(), Bm.sum()])
            save(f"ltv12x4_N{N}_seed{b}", s, p["target_states"][b], p["init_inputs"][b], p["x0"][b], rec)


def warmstart_fixture():
    """What calling the reference solver object AGAIN does (``ilqr.py:107-111``: ``min_cost`` is
    recomputed, ``converge`` is never reset, the iteration continues from the attributes), and the
    receding-horizon use of it (``ilqr_demo.py:49`` assigns ``.inputs``): after a first call with a
    small ``maxIter``, (1) a plain second call, and (2) a call after the host shifted the previous
    solution by ``shift`` steps, installed a new initial state and new targets and cleared
    ``converge``."""
    N, shift = 200, 3
    p = problems.car_batch(1, N, first=11)
    tg, ui, st = p["target_states"][0], p["init_inputs"][0], p["init_states"][0]
    car = make_car()
    first = run_reference(car, tg, ui, init_states=st, max_iter=6, snapshots=False)
    # (1) the same object called again: replay the first call on a fresh object, then record the second
    opt = ref_ilqr.iterative_LQR(car, tg, car.dt)
    opt.inputs, opt.states, opt.maxIter = ui.copy(), st.copy(), 6
    opt()
    assert np.array_equal(opt.states, first["states"]) and opt.min_cost == first["min_cost"]
    again = drive(opt, snapshots=False)
    # (2) shifted warm start on a third object
    xs = np.concatenate([first["states"][shift:], np.repeat(first["states"][-1:], shift, axis=0)])
    us = np.concatenate([first["inputs"][shift:], np.repeat(first["inputs"][-1:], shift, axis=0)])
    x0_new = first["states"][shift] + np.array([0.05, -0.04, 0.03, 0.01])
    xs[0] = x0_new
    tg2 = np.concatenate([tg[shift:], np.repeat(tg[-1:], shift, axis=0)]) + 0.02
    opt2 = ref_ilqr.iterative_LQR(car, tg2, car.dt)
    opt2.inputs, opt2.states, opt2.maxIter = us.copy(), xs.copy(), 8
    shifted = drive(opt2, snapshots=False)
    out = dict(target_states=tg, init_inputs=ui, init_states=st, x0=p["x0"][0], shift=shift, x0_new=x0_new,
               target_states_shifted=tg2, first_max_iter=6, again_max_iter=6, shifted_max_iter=8, **sys_arrays(car))
    for tag, r in (("first", first), ("again", again), ("shifted", shifted)):
        for key in ("cost0", "hist_cost", "hist_alpha", "hist_tries", "iters", "converge", "min_cost", "states", "inputs"):
            out[f"{tag}_{key}"] = r[key]
    path = os.path.join(OUT, "car_N200_warmstart.npz")
    np.savez_compressed(path, **out)
    print("car_N200_warmstart:", {t: (int(out[f"{t}_iters"]), float(out[f"{t}_min_cost"])) for t in ("first", "again", "shifted")},
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def constraints_fixtures():
    """Outputs of the reference's ``constraints.py`` (``BubbleConstraint``, :10-48) on seeded
    states, centres and radii: the value and Jacobian every barrier / SQP row here is built from.
    ``inside_*`` is a trajectory strictly inside its bubbles and speed bounds (for the identity
    barrier cost = -mu * sum of logs of the reference's own h)."""
    import constraints as ref_constraints  # noqa: E402  (the reference)
    for seed in (0, 1):
        rng = np.random.Generator(np.random.PCG64(100 + seed))
        N = 16
        states = rng.normal(size=(N, 4)) * np.array([5.0, 5.0, 2.0, 1.0])
        centers = states[:, :2] + rng.normal(size=(N, 2)) * 1.5
        radius = rng.uniform(0.5, 3.0, size=N)
        vb = np.array([-0.5, 2.5])
        con = ref_constraints.BubbleConstraint(N)
        con.setup(centers, radius, vb)
        h = np.array([con.get_h(states[i], centers[i], radius[i]) for i in range(N)])
        dh = np.array([con.dh_dx(states[i], centers[i]) for i in range(N)])
        uieq = np.array([con.get_uieq(states[i], centers[i], radius[i]) for i in range(N)])
        Cu = con.get_linear_constraint(states)
        xmin, xmax = con.get_bounds(states)
        # a trajectory strictly inside: offsets of at most 0.6 r from the centres, speeds inside the box
        M = 60
        tg = np.zeros((M, 4))
        tg[:, 0] = np.cumsum(rng.uniform(0.1, 0.4, size=M))
        tg[:, 1] = np.cumsum(rng.uniform(-0.2, 0.3, size=M))
        tg[:, 2] = rng.uniform(0.5, 1.5, size=M)
        tg[:, 3] = rng.uniform(-0.5, 0.5, size=M)
        rad_in = rng.uniform(1.0, 4.0, size=M)
        ang, frac = rng.uniform(0, 2 * np.pi, size=M), rng.uniform(0.0, 0.6, size=M)
        st_in = tg.copy()
        st_in[:, 0] += frac * rad_in * np.cos(ang)
        st_in[:, 1] += frac * rad_in * np.sin(ang)
        st_in[:, 2] = rng.uniform(0.2, 1.8, size=M)
        u_in = rng.normal(size=(M - 1, 2)) * 0.1
        con_in = ref_constraints.BubbleConstraint(M)
        con_in.setup(tg[:, :2].copy(), rad_in, np.array([0.0, 2.0]))
        h_in = np.array([con_in.get_h(st_in[i], tg[i, :2], rad_in[i]) for i in range(M)])
        dh_in = np.array([con_in.dh_dx(st_in[i], tg[i, :2]) for i in range(M)])
        assert np.all(h_in < 0)
        path = os.path.join(OUT, f"constraints_seed{seed}.npz")
        np.savez_compressed(path, states=states, centers=centers, radius=radius, vel_bounds=vb, h=h, dh_dx=dh,
                            uieq=uieq, Cu=Cu, xmin=xmin, xmax=xmax, inside_targets=tg, inside_states=st_in,
                            inside_inputs=u_in, inside_radius=rad_in, inside_vel_bounds=np.array([0.0, 2.0]),
                            inside_h=h_in, inside_dh_dx=dh_in)
        print(f"constraints_seed{seed}: h in [{h.min():.3f}, {h.max():.3f}], inside h in [{h_in.min():.3f}, {h_in.max():.3f}]"
              f" -> {os.path.getsize(path) / 1024:.0f} KiB")


EXTRA = {"ltv": ltv_fixtures, "warmstart": warmstart_fixture, "constraints": constraints_fixtures}


def main():
    if len(sys.argv) > 1:     # only the named groups (the others are left as committed)
        for name in sys.argv[1:]:
            EXTRA[name]()
        return
    for fn in EXTRA.values():
        fn()
    # Car, the demo construction (ilqr_demo.py:9-49), N=200 x 3 seeds and the demo's N=120
    for N, seed in ((200, 0), (200, 1), (200, 2), (120, 0)):
        tg, ui, _ = demo_problems.demo_car(N, seed)
        car = make_car()
        save(f"car_N{N}_seed{seed}", car, tg, ui, None, run_reference(car, tg, ui))
    # CarAcceleration demo (ilqr_demo.py:89-135)
    tg, ui = demo_problems.demo_caracc(100, 0)
    s = ref_systems.CarAcceleration()
    s.set_dt(0.2)
    s.set_cost(problems.CARACC_Q.copy(), problems.CARACC_R.copy())
    s.set_control_limit(*problems.CARACC_LIMITS)
    save("caracc_N100_seed0", s, tg, ui, None, run_reference(s, tg, ui))
    # DubinsCar demo (ilqr_demo.py:170-192), inputs start at zero
    tg, ui = demo_problems.demo_dubins(200, 0)
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
    tg, ui, _ = demo_problems.demo_car(200, 3)
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
