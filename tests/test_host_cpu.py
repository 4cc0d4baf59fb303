"""CPU tests of the host side: the C-ABI library loads and exports what the header
declares, descriptor validation, the model-setup mirror of systems.py, the problem
generators and the constraint object.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, demo_car_system
from ilqr_amd import _lib, problems, systems
from oracle import demo_problems
from ilqr_amd.constraints import BubbleConstraint


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ilqr_amd.h")).read()
    declared = set(re.findall(r"\b(ilqr_[a-z_0-9]+)\s*\(", header))
    assert {"ilqr_create", "ilqr_solve", "ilqr_backward", "ilqr_forward", "ilqr_to_tiled"} <= declared
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert set(_lib.EXPORTED_SYMBOLS) == declared


def _desc(**kw):
    car = demo_car_system()
    d = _lib.Desc()
    d.B, d.N, d.n, d.m = 128, 50, 4, 2
    d.model, d.dtype = 0, _lib.F64
    d.control_limited = 1
    d.dt, d.alpha_floor = 0.2, 1e-4
    d.barrier_delta = 0.1
    for i, v in enumerate(np.asarray(car.Q).ravel()):
        d.Q[i] = v
        d.Qf[i] = 100 * v
    for i, v in enumerate(np.asarray(car.R).ravel()):
        d.R[i] = v
    for j in range(2):
        d.lo[j], d.hi[j] = car.control_lower_limit[j], car.control_upper_limit[j]
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def _create(d):
    h = C.c_void_p()
    rc = _lib.lib().ilqr_create(C.byref(d), C.byref(h))
    return rc, h


def test_create_validates_descriptor():
    lib = _lib.lib()
    rc, h = _create(_desc())
    assert rc == 0 and h.value
    assert lib.ilqr_num_alphas(h) == 15          # 2^0 .. 2^-14 for the 1e-4 floor of ilqr.py:71
    assert lib.ilqr_workspace_bytes(h) >= 128 * 8
    lib.ilqr_destroy(h)
    rc, h = _create(_desc(alpha_floor=1e-2))
    assert rc == 0 and lib.ilqr_num_alphas(h) == 8   # 2^0 .. 2^-7 for cilqr.py:41
    lib.ilqr_destroy(h)
    assert _create(_desc(B=0))[0] == -1
    assert _create(_desc(N=1))[0] == -1
    assert _create(_desc(n=5))[0] == -1              # Car is 4 x 2
    assert _create(_desc(model=7))[0] == -2
    assert _create(_desc(dtype=3))[0] == -1
    assert _create(_desc(alpha_floor=0.0))[0] == -1
    assert _create(_desc(lanes_per_problem=3, alphas_per_lane=1))[0] == -2
    assert _create(_desc(constraint_mode=2, barrier_delta=0.0))[0] == -1
    assert _create(_desc(model=1, n=5, constraint_mode=2))[0] == -2   # barrier needs the Car layout
    bad = _desc()
    bad.lo[0] = 5.0
    assert _create(bad)[0] == -1
    # unbound handle refuses to launch
    rc, h = _create(_desc())
    assert lib.ilqr_backward(h, None) == -3
    assert lib.ilqr_solve(h, 3, None) == -3
    lib.ilqr_destroy(h)
    assert b"bound" in lib.ilqr_strerror(-3)


def test_auto_choices_and_workspace_of_a_handle():
    """Host-only decisions of ilqr_create: the forward mapping picked for 0 = auto, the size
    threshold of active-set compaction (visible through the workspace the handle asks for) and
    the refusal of the phase-level entry points on an unbound handle."""
    lib = _lib.lib()

    def mapping(h):
        L, A = C.c_int32(0), C.c_int32(0)
        assert lib.ilqr_forward_mapping(h, C.byref(L), C.byref(A)) == 0
        return L.value, A.value

    rc, h = _create(_desc(B=65536, N=200))
    assert rc == 0 and mapping(h) == (1, 6)          # six candidates in the first stage, the rest in the second
    big = lib.ilqr_workspace_bytes(h)
    lib.ilqr_destroy(h)
    rc, h = _create(_desc(B=65536, N=200, repack_every=-1))
    small = lib.ilqr_workspace_bytes(h)
    assert rc == 0 and small < big                   # two working sets of X, U, Xref plus working gains
    assert big - small >= 2 * 65536 * (2 * 200 * 4 + 199 * 2) * 8
    lib.ilqr_destroy(h)
    rc, h = _create(_desc(B=4096, N=200))
    assert rc == 0 and mapping(h) == (16, 1)         # small batch: all step sizes in one stage, a lane each
    # below the compaction threshold: no working sets, but the sine/cosine pairs of the committed
    # trajectory and the candidates' states at the start of every 32-step segment
    auto = lib.ilqr_workspace_bytes(h)
    assert 4096 * (199 * 2 + 7 * 16 * 4) * 8 <= auto < 4096 * (199 * 2 + 7 * 16 * 4) * 8 + 4096 * 200 * 8
    for fn in (lib.ilqr_repack, lib.ilqr_unpack, lib.ilqr_initial_cost, lib.ilqr_forward_evaluate):
        assert fn(h, None) == -3
    assert lib.ilqr_forward_evaluate_stage(h, 1, None) == -3
    assert lib.ilqr_shift(h, 1, None, None) == -3
    lib.ilqr_destroy(h)
    rc, h = _create(_desc(B=4096, N=200, trig_cache=-1, segment_commit=-1))
    assert rc == 0 and lib.ilqr_workspace_bytes(h) < 4096 * 200 * 4 * 8   # neither of them: per-problem scalars only
    lib.ilqr_destroy(h)
    for B, want in ((8192, (8, 2)), (12288, (4, 4)), (16384, (4, 4)), (24576, (2, 3)), (32768, (2, 3)), (40960, (1, 6))):
        rc, h = _create(_desc(B=B, N=200))
        assert rc == 0 and mapping(h) == want
        lib.ilqr_destroy(h)
    rc, h = _create(_desc(alpha_floor=1e-2, B=65536))
    assert rc == 0 and mapping(h) == (1, 8)          # 8 step sizes (cilqr.py:41): a single stage
    lib.ilqr_destroy(h)
    # 8193 .. 16384 problems: ilqr_solve runs in one launch, built on the time-parallel commit, whose
    # candidate checkpoints the workspace then holds; switched off, neither
    ck = 16384 * 7 * 16 * 4 * 8
    rc, h = _create(_desc(B=16384, N=200))
    with_ck = lib.ilqr_workspace_bytes(h)
    lib.ilqr_destroy(h)
    rc2, h = _create(_desc(B=16384, N=200, fused_solve=-1))
    without = lib.ilqr_workspace_bytes(h)
    lib.ilqr_destroy(h)
    assert rc == 0 and rc2 == 0 and with_ck - without == ck
    rc, h = _create(_desc(B=16384, N=200, fused_solve=1, segment_commit=-1))   # asked for, but without what it is built on
    assert rc == 0 and lib.ilqr_workspace_bytes(h) == without
    lib.ilqr_destroy(h)
    # sparse states need no workspace of their own; the calls around them exist and refuse unbound handles
    rc, h = _create(_desc(B=65536, N=200, recompute_states=2, repack_every=-1))
    assert rc == 0 and lib.ilqr_workspace_bytes(h) == small and lib.ilqr_states(h, None) == -3
    lib.ilqr_destroy(h)


def test_systems_interface_matches_reference_surface():
    car = systems.Car()
    assert (car.state_size, car.control_size, car.dt, car.control_limited) == (4, 2, 0.2, False)
    car.set_dt(0.1)
    Q, R = np.diag([1.0, 2.0, 3.0, 4.0]), np.eye(2)
    car.set_cost(Q, R)
    assert np.array_equal(car.Q_f, 100 * Q)       # systems.py:16
    car.set_final_cost(2 * Q)
    assert np.array_equal(car.Q_f, 2 * Q)
    car.set_control_limit([-1, -2], [1, 2])
    assert car.control_limited and car.control_lower_limit == [-1, -2]
    assert systems.CarAcceleration().state_size == 5 and systems.DubinsCar().state_size == 3


@pytest.mark.parametrize("cls", [systems.Car, systems.CarAcceleration, systems.DubinsCar])
def test_model_jacobians_by_finite_differences(cls):
    s = cls()
    rng = np.random.default_rng(3)
    x, u = rng.normal(size=s.state_size), rng.normal(size=s.control_size)
    fx, fu = s.compute_df_dx(x, u), s.compute_df_du(x, u)
    eps = 1e-6
    for i in range(s.state_size):
        e = np.zeros(s.state_size)
        e[i] = eps
        assert np.allclose((s.model_f(x + e, u) - s.model_f(x - e, u)) / (2 * eps), fx[:, i], atol=1e-8)
    for i in range(s.control_size):
        e = np.zeros(s.control_size)
        e[i] = eps
        assert np.allclose((s.model_f(x, u + e) - s.model_f(x, u - e)) / (2 * eps), fu[:, i], atol=1e-8)
    # batched evaluation broadcasts
    X, U = rng.normal(size=(5, s.state_size)), rng.normal(size=(5, s.control_size))
    assert np.allclose(s.model_f(X, U)[2], s.model_f(X[2], U[2]))


def test_problem_generator_is_shard_invariant():
    a = problems.car_batch(8, 60)
    b = problems.car_batch(3, 60, first=4)
    for key in ("target_states", "init_inputs", "x0", "init_states"):
        assert np.array_equal(a[key][4:7], b[key])
    assert a["init_states"].shape == (8, 60, 4)
    assert np.array_equal(a["init_states"][:, 0], a["x0"])
    # demo construction is deterministic per seed and matches the stored fixture inputs
    tg, ui, _ = demo_problems.demo_car(200, 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "car_N200_seed0.npz"))
    assert np.array_equal(tg, g["target_states"]) and np.array_equal(ui, g["init_inputs"])


def test_bubble_constraint_functions():
    c = BubbleConstraint(5)
    centers = np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0], [3.0, 3.0], [4.0, 4.0]])
    c.setup(centers, np.full(5, 2.0), [0.0, 3.0])
    x = np.array([1.5, 0.5, 1.0, 0.3])
    assert c.get_h(x, centers[1], 2.0) == pytest.approx(0.25 + 0.25 - 4.0)   # constraints.py:21
    g = c.dh_dx(x, centers[1])
    assert g.shape == (2, 4) and np.allclose(g[0], [1.0, -1.0, 0, 0]) and np.allclose(g[1], [0, 0, 1, 0])
    states = np.tile(x, (5, 1))
    Cu = c.get_linear_constraint(states)
    assert Cu.shape == (10, 20) and np.allclose(Cu[2:4, 4:8], g)
    lo, hi = c.get_bounds(states)
    assert lo[0] == -np.inf and lo[1] == 0.0 and hi[1] == 3.0
    assert hi[2] == pytest.approx(g[0, 0] * x[0] + g[0, 1] * x[1] - c.get_h(x, centers[1], 2.0))


def test_solver_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ilqr_amd.ilqr import iterative_LQR
    tg, ui, _ = demo_problems.demo_car(20, 0)
    opt = iterative_LQR(demo_car_system(), tg, 0.2)
    opt.inputs = ui
    with pytest.raises(_lib.IlqrError):
        opt()


def test_solver_attribute_surface():
    from ilqr_amd.cilqr import iterative_LQR as CILQR
    from ilqr_amd.ilqr import iterative_LQR
    tg, ui, _ = demo_problems.demo_car(20, 0)
    car = demo_car_system()
    o = iterative_LQR(car, tg, 0.2)
    assert (o.horizon, o.n_states, o.m_inputs, o.maxIter, o.LM_parameter, o.converge, o.min_cost) == \
        (20, 4, 2, 30, 0.0, False, 0.0)
    assert o.states.shape == (20, 4) and o.inputs.shape == (19, 2) and o.Q is car.Q and o.Qf is car.Q_f
    ob = iterative_LQR(car, np.stack([tg, tg]), 0.2)
    assert ob.states.shape == (2, 20, 4) and ob.converge.shape == (2,)
    c = CILQR(car, tg, 0.2)
    assert (c.maxIter, c.control_limited, c.input_limit_lambda, c.alpha_terminal) == (100, True, 1e-4, 1e-2)
    with pytest.raises(ValueError):
        iterative_LQR(car, np.zeros((20, 3)), 0.2)
