"""GPU parity tests against the fixtures produced by running the reference's own files on cases its
``systems.py`` / demos do not cover (oracle/make_golden.py): the constraint functions of
``constraints.py`` under the barrier, the n = 12, m = 4 table-driven model solved by the reference
``ilqr.py`` through a step-counting System, and a second / warm-started call of the solver object.
Everything goes through the C-ABI library.  Tolerances as in test_gpu_parity.py (fp64: cost 1e-9
relative per iteration, gains 1e-11 where sums run in the matrix unit's order)."""
import numpy as np
import pytest

from conftest import demo_car_system, golden_names, load_golden, noise_floor_mask, system_from_golden

pytestmark = pytest.mark.gpu


def nrm(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def ilqr_cls():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqr_amd.ilqr import iterative_LQR
    return iterative_LQR


# --------------------------------------------------------------------------- constraints.py
@pytest.mark.parametrize("name", golden_names(prefix="constraints_"))
def test_barrier_cost_is_built_on_the_reference_h(oracle, name):
    """J(barrier) - J(plain) = -mu * sum_{i >= 1} [log(-h_i) + log(v_i - vlo) + log(vhi - v_i)] on a
    trajectory strictly inside its bubbles, with h_i = the reference's ``get_h`` (fixture); and one
    backward pass with the barrier's gradient / Hessian against the oracle (itself pinned to the
    reference's h and dh/dx by tests/test_pinned_cpu.py)."""
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    from ilqr_amd.constraints import BubbleConstraint
    g = load_golden(name)
    car = demo_car_system()
    st, tg, ui, rad = (np.ascontiguousarray(g[k]) for k in ("inside_states", "inside_targets", "inside_inputs", "inside_radius"))
    N = st.shape[0]
    vb = g["inside_vel_bounds"]
    mu = 7.0
    B = 3                      # the same problem three times, so that lanes 0..2 of a tile are checked
    rep = lambda a: np.stack([a] * B)
    plain = CILQR(car, rep(tg), car.dt)
    plain.states, plain.inputs = rep(st), rep(ui)
    Jp = plain.cost()
    con = BubbleConstraint(N)
    con.setup(rep(np.ascontiguousarray(tg[:, :2])), rep(rad), vb)
    opt = CILQR(car, rep(tg), car.dt)
    opt.states, opt.inputs = rep(st), rep(ui)
    opt.set_barrier(con, mu=mu, delta=1e-9)
    Jb = opt.cost()
    logs = np.log(-g["inside_h"][1:]) + np.log(st[1:, 2] - vb[0]) + np.log(vb[1] - st[1:, 2])
    want = -mu * logs.sum()
    assert np.all(np.abs((Jb - Jp) - want) <= 1e-10 * abs(want))
    # gradient and Hessian inside the fused backward pass
    opt.backward_pass()
    d = oracle.make_desc(car, N, max_iter=100, alpha_floor=1e-2, solver_control_limited=True, barrier=True, mu=mu,
                         delta=1e-9, vel_bounds=tuple(vb))
    Ko, ko, rc = oracle.backward(d, tg, st, ui, centers=np.ascontiguousarray(tg[:, :2]), radius=rad)
    assert rc == 0
    for b in range(B):
        assert nrm(opt.K[b], Ko) <= 1e-11 and nrm(opt.k[b][:, :, 0], ko) <= 1e-11


# --------------------------------------------------------------------------- LTV n = 12, m = 4
@pytest.mark.parametrize("name", golden_names(prefix="ltv12x4_"))
@pytest.mark.parametrize("generic", [False, True])
def test_ltv12x4_matches_reference_fixture(ilqr_cls, name, generic, monkeypatch):
    """The matrix-unit kernels (and, with ILQR_LTV_GENERIC, the lane-per-problem kernels) of the
    table-driven model against the REFERENCE solver.  The fixture problem sits in the middle of a
    batch of other problems, so that its wave shares a workgroup with neighbours."""
    from ilqr_amd import problems
    g = load_golden(name)
    if generic and g["target_states"].shape[0] > 200:
        pytest.skip("the lane-per-problem kernels are the cross-check at the small size only")
    if generic:
        monkeypatch.setenv("ILQR_LTV_GENERIC", "1")
    s = system_from_golden(g)
    N = g["target_states"].shape[0]
    B, at = 37, 21
    p = problems.ltv_batch(B, N, first=100)
    p["target_states"][at] = g["target_states"]
    p["init_states"][at] = g["init_states"]
    p["init_inputs"][at] = g["init_inputs"]
    opt = ilqr_cls(s, p["target_states"], s.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    dev.initial_cost()
    assert abs(float(dev.J.cpu()[at]) - float(g["cost0"])) <= 1e-12 * float(g["cost0"])
    dev.backward()
    assert nrm(dev.download_K()[at], g["K_first"]) <= 1e-11
    assert nrm(dev.download_k()[at][:, :, 0], g["k_first"]) <= 1e-11
    opt()
    it = int(g["iters"])
    solid = noise_floor_mask(g["hist_cost"][:it], float(g["cost0"]))
    last_solid = int(np.nonzero(solid)[0].max()) + 1
    n = min(int(opt.iterations[at]), it)
    assert n >= last_solid
    rel = np.abs(opt.cost_history[at, :n] - g["hist_cost"][:n]) / np.abs(g["hist_cost"][:n])
    assert rel.max() <= 1e-9
    assert np.array_equal(opt.alpha_history[at, :n][solid[:n]], g["hist_alpha"][:n][solid[:n]])
    assert abs(opt.min_cost[at] - float(g["min_cost"])) <= 1e-9 * float(g["min_cost"])
    assert nrm(opt.states[at], g["states"]) <= 1e-7 and nrm(opt.inputs[at], g["inputs"]) <= 1e-6   # reported at termination
    assert opt.status[at] == 0
    if "hist_states" in g.files:   # trajectories iteration by iteration while the decrease is real
        replay = ilqr_cls(s, p["target_states"], s.dt)
        replay.inputs, replay.states = p["init_inputs"].copy(), p["init_states"].copy()
        dev = replay.upload()
        dev.initial_cost()
        for k in range(last_solid):
            dev.backward()
            dev.forward(k)
            assert nrm(dev.download_states()[at], g["hist_states"][k]) <= 1e-9
            assert nrm(dev.download_inputs()[at], g["hist_inputs"][k]) <= 1e-8


def test_ltv12x4_fp32_against_reference_fixture(ilqr_cls):
    """fp32 mode, stated separately: first-iteration cost to 1e-4, final cost to 2e-3 of the reference."""
    g = load_golden("ltv12x4_N500_seed1")
    s = system_from_golden(g)
    opt = ilqr_cls(s, np.array(g["target_states"])[None], s.dt, history=True, dtype=np.float32)
    opt.inputs, opt.states = np.array(g["init_inputs"])[None], np.array(g["init_states"])[None]
    opt()
    assert abs(opt.cost_history[0, 0] - g["hist_cost"][0]) <= 1e-4 * g["hist_cost"][0]
    assert abs(opt.min_cost[0] - float(g["min_cost"])) <= 2e-3 * float(g["min_cost"])


# --------------------------------------------------------------------------- second call / warm start
def _hist_close(h, want, tol=1e-9):
    return (np.abs(h - want) / np.abs(want)).max() <= tol


def test_second_call_continues_like_the_reference(ilqr_cls):
    """ilqr.py:107-111: calling the object again recomputes min_cost from the attributes and keeps
    iterating from them; maxIter may be changed between the calls."""
    g = load_golden("car_N200_warmstart")
    s = system_from_golden(g)
    opt = ilqr_cls(s, np.array(g["target_states"]), s.dt, history=True)
    opt.inputs, opt.states = np.array(g["init_inputs"]), np.array(g["init_states"])
    opt.maxIter = int(g["first_max_iter"])
    opt()
    n = int(g["first_iters"])
    assert opt.iterations == n and opt.converge == bool(g["first_converge"])
    assert _hist_close(opt.cost_history[:n], g["first_hist_cost"])
    assert np.array_equal(opt.alpha_history[:n], g["first_hist_alpha"])
    assert nrm(opt.states, g["first_states"]) <= 1e-9 and nrm(opt.inputs, g["first_inputs"]) <= 1e-8
    opt.maxIter = int(g["again_max_iter"])
    opt()
    n = int(g["again_iters"])
    assert opt.iterations == n
    assert _hist_close(opt.cost_history[:n], g["again_hist_cost"])
    assert np.array_equal(opt.alpha_history[:n], g["again_hist_alpha"])
    assert nrm(opt.states, g["again_states"]) <= 1e-9 and nrm(opt.inputs, g["again_inputs"]) <= 1e-8
    assert abs(opt.min_cost - float(g["again_min_cost"])) <= 1e-9 * float(g["again_min_cost"])


def test_receding_horizon_driver_matches_reference_warm_start(ilqr_cls):
    """``RecedingHorizon`` / ``ilqr_shift`` against the reference: its first call, then on the host
    the previous solution shifted by three steps, a new initial state, new targets, ``converge``
    cleared, and a second call with another ``maxIter`` (fixture).  The driver does the same on the
    device.  The fixture problem is lane 5 of a batch of 9."""
    from ilqr_amd import problems
    from ilqr_amd.mpc import RecedingHorizon
    g = load_golden("car_N200_warmstart")
    s = system_from_golden(g)
    N, B, at = 200, 9, 5
    p = problems.car_batch(B, N, first=40)
    for key, src in (("target_states", "target_states"), ("init_inputs", "init_inputs"), ("init_states", "init_states")):
        p[key][at] = g[src]
    opt = ilqr_cls(s, p["target_states"], s.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    opt.maxIter = int(g["first_max_iter"])
    mpc = RecedingHorizon(opt)
    mpc.step(first=True)
    first = mpc.solution()
    assert nrm(first.states[at], g["first_states"]) <= 1e-9
    sh = int(g["shift"])
    tg2 = np.concatenate([p["target_states"][:, sh:], np.repeat(p["target_states"][:, -1:], sh, axis=1)], axis=1)
    tg2[at] = g["target_states_shifted"]
    x0 = first.states[:, sh].copy()
    x0[at] = g["x0_new"]
    opt.maxIter = int(g["shifted_max_iter"])
    mpc.prefetch_targets(tg2)
    u0 = mpc.step(shift=sh, x0=x0)
    sol = mpc.solution()
    n = int(g["shifted_iters"])
    assert sol.iterations[at] == n
    assert _hist_close(sol.cost_history[at, :n], g["shifted_hist_cost"])
    assert np.array_equal(sol.alpha_history[at, :n], g["shifted_hist_alpha"])
    assert nrm(sol.states[at], g["shifted_states"]) <= 1e-9 and nrm(sol.inputs[at], g["shifted_inputs"]) <= 1e-8
    assert np.array_equal(u0[at], sol.inputs[at, 0])
