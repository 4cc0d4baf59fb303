"""CPU tests around the sequential-QP path: the numpy oracle's two independent QP solvers
against each other and against the optimality conditions, and the host-side refusals of the
C ABI.  The reference's own QP solver (the OSQP package) is absent, so nothing here or in the
GPU tests is pinned to reference output: PARITY UNPINNED for this path."""
import ctypes as C

import numpy as np
import pytest

from ilqr_amd import _lib, problems, systems
from ilqr_amd.constraints import BubbleConstraint
from oracle import sqp_oracle as so
from oracle import demo_problems


def sqp_car(N=80, limits=problems.SQP_LIMITS):
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.SQP_Q.copy(), problems.SQP_R.copy())
    car.Q_f = car.Q * N / 100            # sqp_demo.py:26
    if limits is not None:
        car.set_control_limit(*limits)
    return car


CASES = [  # seed, bubble radius, speed bounds, input limits: nothing active / speed rows active / inputs and bubbles active
    (0, 0.6, (0.0, 2.0), problems.SQP_LIMITS),
    (1, 0.6, (0.0, 1.01), problems.SQP_LIMITS),
    (2, 0.5, (0.0, 1.05), ([-0.05, -0.06], [0.05, 0.06])),
]


@pytest.mark.parametrize("seed,rad,vb,lim", CASES)
def test_qp_solvers_agree_and_satisfy_optimality(seed, rad, vb, lim):
    N = 60
    tg, ui, c, r, _ = demo_problems.demo_sqp(N, seed, radius=rad)
    car = sqp_car(N, lim)
    con = BubbleConstraint(N)
    con.setup(c, r, vb)
    umin, umax = np.array(lim[0], float), np.array(lim[1], float)
    A, B, cc, G, lo, hi = so.qp_data(car, con, tg, tg.copy(), ui)
    u_d, x_d = so.solve_qp(car, tg, tg[0], A, B, cc, G, lo, hi, umin, umax)
    u_r, x_r, its = so.solve_qp_riccati(car, tg, tg[0], A, B, cc, G, lo, hi, umin, umax, ui)
    assert its < 100
    assert np.abs(u_r - u_d).max() <= 1e-4
    # feasibility of the Riccati solution and no descent along feasible directions: for the convex
    # QP, u is optimal iff grad . (v - u) >= 0 for every feasible v; test with the other solver's
    # point and with random feasible perturbations towards the interior
    z = np.einsum("nij,nj->ni", G, x_r)
    assert (z - hi)[1:].max() <= 1e-8 and (lo - z)[1:].max() <= 1e-8
    assert (u_r - umax).max() <= 1e-8 and (umin - u_r).max() <= 1e-8

    def objective(u):
        x = np.zeros((N, 4))
        x[0] = tg[0]
        for i in range(N - 1):
            x[i + 1] = A[i] @ x[i] + B[i] @ u[i] + cc[i]
        d = x - tg
        return 0.5 * (np.einsum("ni,ij,nj->", d[:-1], car.Q, d[:-1]) + d[-1] @ car.Q_f @ d[-1]
                      + np.einsum("ni,ij,nj->", u, car.R, u))

    f_r, f_d = objective(u_r), objective(u_d)
    assert f_r <= f_d + 1e-7 * abs(f_d)
    # fixed first input (sqp.py:111-112)
    u0 = ui[0] + 0.01
    u_d0, _ = so.solve_qp(car, tg, tg[0], A, B, cc, G, lo, hi, umin, umax, u0=u0)
    u_r0, _, _ = so.solve_qp_riccati(car, tg, tg[0], A, B, cc, G, lo, hi, umin, umax, ui, u0=u0)
    assert np.array_equal(u_r0[0], u0) and np.abs(u_r0 - u_d0).max() <= 1e-4


def test_oracle_sqp_runs_the_loop_of_the_reference():
    N = 50
    tg, ui, c, r, vb = demo_problems.demo_sqp(N, 3)
    car = sqp_car(N)
    con = BubbleConstraint(N)
    con.setup(c, r, vb)
    a = so.sqp_solve(car, con, tg, ui, qp="dense")
    b = so.sqp_solve(car, con, tg, ui, qp="riccati")
    assert a["converge"] and b["converge"] and a["iters"] <= 20
    n = min(a["iters"], b["iters"])
    assert np.allclose(a["hist_cost"][:n], b["hist_cost"][:n], rtol=1e-6)
    assert np.all(np.diff(a["hist_cost"]) <= 0)           # accepted steps only ever lower the cost (sqp.py:160-162)
    assert a["hist_alpha"][0] == 0                         # min_cost starts at inf: the first full step is taken
    assert a["hist_alpha"][-1] == -1                       # the last search gave up (sqp.py:168-170)
    assert np.allclose(so.sim(car, tg[0], a["inputs"]), a["states"])


def _handle(constraint_mode, dtype=_lib.F64, model=0):
    d = _lib.Desc()
    d.B, d.N, d.n, d.m, d.model, d.dtype = 64, 20, 4, 2, model, dtype
    d.dt, d.alpha_floor, d.constraint_mode = 0.2, 1e-2, constraint_mode
    d.barrier_mu, d.barrier_delta = 0.0, 1.0
    for i in range(4):
        d.Q[i * 4 + i] = d.Qf[i * 4 + i] = 1.0
    d.R[0] = d.R[3] = 1.0
    h = C.c_void_p()
    assert _lib.lib().ilqr_create(C.byref(d), C.byref(h)) == 0
    b = _lib.Buffers()
    for name in ("X", "U", "Xref", "K", "k", "J", "converged", "iters", "alpha_idx", "work", "centers", "radius"):
        setattr(b, name, 4096)            # never dereferenced on the host
    assert _lib.lib().ilqr_bind(h, C.byref(b)) == 0
    return h


def test_sqp_entry_points_refuse_what_they_cannot_do():
    lib = _lib.lib()
    o = _lib.SqpOpts()
    o.max_iter, o.eps, o.cost_eps = 20, 1e-2, 1e-6
    h = _handle(_lib.CONSTR_BARRIER)
    assert lib.ilqr_sqp_workspace_bytes(h) > 0
    assert lib.ilqr_sqp_qp(h, C.byref(o), None, None, None, None) == -1             # no workspace
    o.eps = 0.0
    assert lib.ilqr_sqp_qp(h, C.byref(o), 4096, None, None, None) == -1             # eps must be positive
    o.eps, o.init_input_fixed = 1e-2, 1
    assert lib.ilqr_sqp_qp(h, C.byref(o), 4096, None, None, None) == -3             # fixed first input not given
    o.init_input_fixed, o.qp_sigma = 0, 1.5
    assert lib.ilqr_sqp_qp(h, C.byref(o), 4096, None, None, None) == -1
    assert lib.ilqr_sqp_search(h, C.byref(_lib.SqpOpts(max_iter=1, eps=1e-2)), 4096, -1, None) == -1
    lib.ilqr_destroy(h)
    h = _handle(_lib.CONSTR_NONE)                                                    # centers / radius not demanded
    assert lib.ilqr_sqp_begin(h, None) == -2
    lib.ilqr_destroy(h)
    h = _handle(_lib.CONSTR_BARRIER, dtype=_lib.F32)                                 # fp64 only
    assert lib.ilqr_sqp_begin(h, None) == -2
    lib.ilqr_destroy(h)


def test_python_class_surface_and_refusal_without_gpu():
    """Constructor, attributes and defaults of sqp.py:17-42 / :63-65; no CPU fallback."""
    import torch
    from ilqr_amd.sqp import sequential_QP_optimizer
    tg, ui, c, r, vb = demo_problems.demo_sqp(30, 0)
    car = sqp_car(30)
    con = BubbleConstraint(30)
    con.setup(c, r, vb)
    o = sequential_QP_optimizer(car, con, tg, 0.2)
    assert (o.horizon, o.n_states, o.m_inputs, o.maxIter, o.eps, o.cost_eps, o.init_input_fixed, o.converge) == \
        (30, 4, 2, 20, 1e-2, 1e-6, False, False)
    assert o.states.shape == (30, 4) and o.inputs.shape == (29, 2) and o.Qf is car.Q_f
    assert np.array_equal(o.x0, tg[0]) and list(o.umin) == [-1.0, -0.3] and list(o.umax) == [1.0, 0.3]
    o.set_init_inputs(ui)
    assert o.inputs is ui and np.array_equal(o.u0, ui[0])
    free = sequential_QP_optimizer(sqp_car(30, limits=None), con, tg, 0.2)
    assert np.all(np.isinf(free.umin)) and np.all(np.isinf(free.umax))           # sqp.py:38-39
    ob = sequential_QP_optimizer(car, con, np.stack([tg, tg, tg]), 0.2)
    assert ob.states.shape == (3, 30, 4) and ob.x0.shape == (3, 4)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.IlqrError):
            o()


def test_sqp_problem_generator_is_shard_invariant():
    """Problem b depends on its global index only, so ranks that build their own index range
    (ilqr_amd.dist.shard_range) hold exactly the problems of the whole batch."""
    whole = problems.sqp_batch(10, 40)
    part = problems.sqp_batch(4, 40, first=3)
    for key in ("target_states", "init_inputs", "centers", "radius"):
        assert np.array_equal(whole[key][3:7], part[key])
    bike = problems.bicycle_batch(6, 50)
    assert np.array_equal(bike["target_states"][2:5], problems.bicycle_batch(3, 50, first=2)["target_states"])
