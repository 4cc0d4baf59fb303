"""GPU tests of the batched sequential-QP solver (ilqr_amd/sqp.py, csrc/sqp.hpp) against the numpy
oracle (oracle/sqp_oracle.py).  PARITY UNPINNED towards the reference (its QP solver, the OSQP
package, is absent); towards the oracle: per-iteration cost 1e-9, equal accepted steps wherever
the accepted decrease is above round-off, solved QP inputs 1e-8 against the same method in numpy
and 1e-4 against the dense interior-point solver."""
import ctypes as C

import numpy as np
import pytest

from conftest import noise_floor_mask
from test_sqp_cpu import CASES, sqp_car
from oracle import demo_problems

pytestmark = pytest.mark.gpu


def _solver(car, centers, radius, vb, targets, inputs, **kw):
    from ilqr_amd.constraints import BubbleConstraint
    from ilqr_amd.sqp import sequential_QP_optimizer
    con = BubbleConstraint(targets.shape[-2])
    con.setup(centers, radius, vb)
    opt = sequential_QP_optimizer(car, con, targets, 0.2, history=True, **kw)
    opt.set_init_inputs(np.array(inputs))
    return opt, con


def _check_against_oracle(opt_cost, opt_alpha, opt_iters, r, tol=1e-9):
    n = min(int(opt_iters), r["iters"])
    assert n >= r["iters"] - 3
    rel = np.abs(opt_cost[:n] - r["hist_cost"][:n]) / np.abs(r["hist_cost"][:n])
    assert rel.max() <= tol, rel
    solid = noise_floor_mask(r["hist_cost"][:n], np.inf, rel=1e-10)
    assert np.array_equal(opt_alpha[:n][solid], r["hist_alpha"][:n][solid])
    return n


def test_single_problem_like_the_demo(oracle):
    """sqp_demo.py:12-58 with a seeded random stream, one problem, the reference's call sequence."""
    from oracle import sqp_oracle as so
    from ilqr_amd import problems
    tg, ui, c, r, vb = demo_problems.demo_sqp(80, 0)
    car = sqp_car(80)
    opt, con = _solver(car, c, r, vb, tg, ui)
    out = opt()
    assert out is opt.states and out.shape == (80, 4) and isinstance(opt.min_cost, float)
    ref = so.sqp_solve(car, con, tg, ui, qp="riccati")
    _check_against_oracle(opt.cost_history, opt.alpha_history, opt.iterations, ref)
    assert opt.converge == ref["converge"] and opt.status == 0
    assert np.abs(opt.states - ref["states"]).max() <= 1e-7 and np.abs(opt.inputs - ref["inputs"]).max() <= 1e-7
    assert np.allclose(so.sim(car, tg[0], opt.inputs), opt.states, rtol=0, atol=1e-12)


@pytest.mark.parametrize("seed,rad,vb,lim", CASES)
def test_first_qp_matches_both_oracle_solvers(seed, rad, vb, lim):
    """One ilqr_sqp_qp call through the C ABI: the solved inputs against the numpy statement of the
    same method and against the dense interior-point solver."""
    import torch
    from oracle import sqp_oracle as so
    from ilqr_amd import _lib, problems
    N = 60
    tg, ui, c, r, _ = demo_problems.demo_sqp(N, seed, radius=rad)
    car = sqp_car(N, lim)
    for fixed in (False, True):
        opt, con = _solver(car, c, r, vb, tg, ui)
        if fixed:
            opt.init_input_fixed = True
            opt.u0 = ui[0] + 0.01
        dev = opt.upload()
        o = opt._opts()
        lib = dev.lib
        _lib.check(lib.ilqr_sqp_begin(dev.handle, dev.stream()), "begin")
        _lib.check(lib.ilqr_sqp_qp(dev.handle, C.byref(o), opt._work.data_ptr(), opt._u0.data_ptr(), opt._qp_iters.data_ptr(),
                                   dev.stream()), "qp")
        uq = torch.zeros(64 * (N - 1) * 2, dtype=torch.float64, device="cuda")
        ptr = lib.ilqr_sqp_solved_inputs(dev.handle, opt._work.data_ptr())
        out = torch.empty((N - 1) * 2, dtype=torch.float64, device="cuda")
        _lib.check(lib.ilqr_from_tiled(ptr, out.data_ptr(), 1, (N - 1) * 2, 8, dev.stream()), "from_tiled")
        u_gpu = out.cpu().numpy().reshape(N - 1, 2)
        del uq
        A, B, cc, G, lo, hi = so.qp_data(car, con, tg, tg.copy(), ui)
        umin, umax = np.array(lim[0], float), np.array(lim[1], float)
        u0 = opt.u0 if fixed else None
        u_r, _, its = so.solve_qp_riccati(car, tg, tg[0], A, B, cc, G, lo, hi, umin, umax, ui, u0=u0)
        u_d, _ = so.solve_qp(car, tg, tg[0], A, B, cc, G, lo, hi, umin, umax, u0=u0)
        assert int(opt._qp_iters.cpu()[0]) == its
        assert np.abs(u_gpu - u_r).max() <= 1e-8
        assert np.abs(u_gpu - u_d).max() <= 1e-4
        assert int(dev.status.cpu()[0]) == 0


def test_batch_matches_oracle_and_does_not_depend_on_sharding():
    from oracle import sqp_oracle as so
    from ilqr_amd import problems
    from ilqr_amd.constraints import BubbleConstraint
    B, N = 100, 80
    p = problems.sqp_batch(B, N)
    car = sqp_car(N)
    opt, _ = _solver(car, p["centers"], p["radius"], p["vel_bounds"], p["target_states"], p["init_inputs"])
    opt()
    assert opt.states.shape == (B, N, 4) and opt.min_cost.shape == (B,)
    ok = opt.status == 0
    assert ok.mean() >= 0.95
    for b in np.flatnonzero(ok)[:10]:
        con = BubbleConstraint(N)
        con.setup(p["centers"][b], p["radius"][b], p["vel_bounds"])
        ref = so.sqp_solve(car, con, p["target_states"][b], p["init_inputs"][b], qp="riccati")
        _check_against_oracle(opt.cost_history[b], opt.alpha_history[b], opt.iterations[b], ref)
    # properties that hold for every problem, whatever its size: the returned states are the
    # simulation of the returned inputs (sqp.py:158), accepted costs only ever fall (sqp.py:160-162),
    # min_cost is the cost of the returned trajectory (sqp.py:44-54)
    for b in range(B):
        assert np.abs(so.sim(car, p["target_states"][b, 0], opt.inputs[b]) - opt.states[b]).max() <= 1e-11
        h = opt.cost_history[b, :int(opt.iterations[b])]
        assert np.all(np.diff(h) <= 0)
        c = so.cost(car, p["target_states"][b], opt.states[b], opt.inputs[b])
        assert abs(c - opt.min_cost[b]) <= 1e-11 * abs(c)
    # problems whose linearised rows cannot be met are flagged, stop, and keep a valid trajectory
    # (the reference hands such a QP to OSQP, which reports it infeasible and returns no solution:
    # the search of sqp.py:156-171 then rejects every step and sets `converge` as well)
    bad = ~ok
    assert np.all(opt.status[bad] == 8) and np.all(opt.converge[bad]) and np.all(np.isfinite(opt.states))
    # ... and only those: for every flagged problem an LP solver certifies that the rows of the QP at
    # the returned trajectory (its linearisation point: the step was zero) have no common point
    umin, umax = np.array(car.control_lower_limit, float), np.array(car.control_upper_limit, float)
    for b in np.flatnonzero(bad):
        con = BubbleConstraint(N)
        con.setup(p["centers"][b], p["radius"][b], p["vel_bounds"])
        A, Bm, c, G, lo, hi = so.qp_data(car, con, p["target_states"][b], opt.states[b], opt.inputs[b])
        assert so.min_relaxation(car, p["target_states"][b, 0], A, Bm, c, G, lo, hi, umin, umax) > 1e-6
    # shard invariance: two halves (ragged against the 64-problem tiles) give the same bits
    parts = []
    for sl in (slice(0, 37), slice(37, B)):
        o, _ = _solver(car, p["centers"][sl], p["radius"][sl], p["vel_bounds"], p["target_states"][sl], p["init_inputs"][sl])
        o()
        parts.append(o)
    assert np.array_equal(np.concatenate([o.states for o in parts]), opt.states)
    assert np.array_equal(np.concatenate([o.min_cost for o in parts]), opt.min_cost)
    assert np.array_equal(np.concatenate([o.iterations for o in parts]), opt.iterations)


def test_inputs_without_limits_and_iteration_budget():
    """control_limited = False: the QP has no input rows (sqp.py:38-39); maxIter is honoured."""
    from oracle import sqp_oracle as so
    from ilqr_amd import problems
    tg, ui, c, r, vb = demo_problems.demo_sqp(60, 4)
    car = sqp_car(60, limits=None)
    opt, con = _solver(car, c, r, vb, tg, ui)
    opt.maxIter = 2
    opt()
    ref = so.sqp_solve(car, con, tg, ui, max_iter=2, qp="riccati")
    assert opt.iterations == 2 == ref["iters"] and not opt.converge
    assert np.abs(opt.cost_history[:2] - ref["hist_cost"]).max() <= 1e-9 * ref["hist_cost"].max()


def test_numpy_only_model_in_the_sequential_qp():
    """sqp.py works on any System (it calls model_f / compute_df_* back); here a numpy-only car is
    recorded and compiled, and the sequential-QP kernels are instantiated for it."""
    from oracle import sqp_oracle as so
    from ilqr_amd import problems
    from ilqr_amd.user_models import NumpyCar
    tg, ui, c, r, vb = demo_problems.demo_sqp(70, 6)
    car = NumpyCar()
    car.set_cost(problems.SQP_Q.copy(), problems.SQP_R.copy())
    car.Q_f = car.Q * 70 / 100
    car.set_control_limit(*problems.SQP_LIMITS)
    opt, con = _solver(car, c, r, vb, tg, ui)
    opt()
    ref = so.sqp_solve(car, con, tg, ui, qp="riccati")
    _check_against_oracle(opt.cost_history, opt.alpha_history, opt.iterations, ref)
    assert opt.status == 0 and np.abs(opt.states - ref["states"]).max() <= 1e-7


@pytest.mark.parametrize("seed,rad,vb,lim", CASES[1:])
@pytest.mark.parametrize("fixed", [False, True])
def test_whole_solves_with_active_rows(seed, rad, vb, lim, fixed):
    """Whole solves in which speed rows, input bounds and bubble rows are active in the QPs, with
    and without the fixed first input (sqp.py:36, :111-112)."""
    from oracle import sqp_oracle as so
    from ilqr_amd import problems
    N = 60
    tg, ui, c, r, _ = demo_problems.demo_sqp(N, seed, radius=rad)
    car = sqp_car(N, lim)
    opt, con = _solver(car, c, r, vb, tg, ui)
    opt.init_input_fixed = fixed
    opt.maxIter = 6
    opt()
    ref = so.sqp_solve(car, con, tg, ui, max_iter=6, init_input_fixed=fixed, qp="riccati")
    if opt.status == 0:
        n = _check_against_oracle(opt.cost_history, opt.alpha_history, opt.iterations, ref, tol=1e-8)
        assert n >= 1
        if fixed and opt.alpha_history[0] == 0:
            assert np.allclose(opt.inputs[0], ui[0], rtol=0, atol=1e-12) or opt.iterations > 1
    else:
        assert opt.status == 8 and opt.converge


def test_assigned_x0_umin_umax_are_honoured():
    """sqp.py:37-42 are attributes a caller may assign: ``x0`` (every QP's first state and every
    rollout's start, :120, :158) and ``umin`` / ``umax`` (the QP's input rows, :130-131)."""
    from oracle import sqp_oracle as so
    from ilqr_amd import problems
    N = 60
    tg, ui, c, r, vb = demo_problems.demo_sqp(N, 4)
    car = sqp_car(N)
    x0 = tg[0] + np.array([0.02, -0.03, 0.1, 0.02])
    umin, umax = np.array([-0.02, -np.inf]), np.array([0.015, 0.1])      # three of the four rows get active
    opt, con = _solver(car, c, r, vb, tg, ui)
    opt.x0, opt.umin, opt.umax = x0, umin, umax
    opt()
    ref = so.sqp_solve(car, con, tg, ui, qp="riccati", x0=x0, umin=umin, umax=umax)
    _check_against_oracle(opt.cost_history, opt.alpha_history, opt.iterations, ref)
    assert opt.status == 0 and np.array_equal(opt.states[0], x0)
    assert np.abs(opt.states - ref["states"]).max() <= 1e-7 and np.abs(opt.inputs - ref["inputs"]).max() <= 1e-7
    assert opt.inputs[:, 0].min() >= -0.02 - 1e-6 and opt.inputs[:, 0].max() <= 0.015 + 1e-6 and opt.inputs[:, 1].max() <= 0.1 + 1e-6
    assert opt.inputs[:, 0].min() <= -0.02 + 1e-6 and opt.inputs[:, 1].max() >= 0.1 - 1e-6
    # and they change the problem: the defaults give another answer
    plain, _ = _solver(car, c, r, vb, tg, ui)
    plain()
    assert abs(plain.min_cost - opt.min_cost) > 1e-6 * plain.min_cost


def test_cost_and_sim_methods():
    """``cost()`` (sqp.py:44-54) and ``sim()`` (sqp.py:56-61) of the class, each one launch over the batch."""
    from oracle import sqp_oracle as so
    from ilqr_amd import problems
    B, N = 70, 50
    p = problems.sqp_batch(B, N)
    car = sqp_car(N)
    opt, _ = _solver(car, p["centers"], p["radius"], p["vel_bounds"], p["target_states"], p["init_inputs"])
    rng = np.random.default_rng(3)
    x0 = p["target_states"][:, 0] + rng.normal(size=(B, 4)) * 0.1
    st = opt.sim(x0, p["init_inputs"])
    assert st.shape == (B, N, 4)
    for b in (0, 33, 69):
        assert np.abs(st[b] - so.sim(car, x0[b], p["init_inputs"][b])).max() <= 1e-12
    opt.states = st
    J = opt.cost()
    for b in (0, 33, 69):
        want = so.cost(car, p["target_states"][b], st[b], p["init_inputs"][b])
        assert abs(J[b] - want) <= 1e-12 * want
