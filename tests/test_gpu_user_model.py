"""GPU tests of the run-time compiled dynamics (systems.DeviceSystem, ilqr_amd/jit.py).

The fixture-based tests of test_gpu_parity.py already solve the KinematicBicycle and Pendulum
fixtures (produced by the reference solver running on the same classes' numpy methods).  Here:
the device snippets against the host statement point by point, batched solves against the
oracle (which calls the numpy methods back, as the reference does), and the other variants."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bicycle(dense=False):
    from ilqr_amd import problems
    from ilqr_amd.user_models import KinematicBicycle
    s = KinematicBicycle()
    Q = problems.BICYCLE_Q.copy()
    if dense:
        Q[0, 1] = Q[1, 0] = 2.0
    s.set_cost(Q, problems.BICYCLE_R.copy())
    s.set_control_limit(*problems.BICYCLE_LIMITS)
    return s


def _pendulum():
    from ilqr_amd import problems
    from ilqr_amd.user_models import Pendulum
    s = Pendulum()
    s.set_cost(problems.PENDULUM_Q.copy(), problems.PENDULUM_R.copy())
    s.set_control_limit(*problems.PENDULUM_LIMITS)
    return s


@pytest.mark.parametrize("make", [_bicycle, _pendulum])
def test_device_snippets_match_host_statement(make):
    s = make()
    rng = np.random.default_rng(11)
    x = rng.normal(size=(300, s.state_size)) * 2.0
    u = rng.normal(size=(300, s.control_size))
    ef, ex, eu = s.check_device_model(x, u)
    assert ef <= 1e-13 * 10 and ex <= 1e-13 * 10 and eu <= 1e-13 * 10, (ef, ex, eu)
    # fp32: away from the poles of tan, where absolute differences mean something
    ef, ex, eu = s.check_device_model(np.clip(x, -1.0, 1.0), u, dtype=np.float32)
    assert max(ef, ex, eu) <= 1e-5, (ef, ex, eu)


@pytest.mark.parametrize("name", ["Car", "CarAcceleration", "DubinsCar"])
def test_builtin_device_models_match_host_statement(name):
    """The same probe on the built-in models: device dynamics and Jacobians against the numpy
    methods of ilqr_amd/systems.py (which restate systems.py:33-57, :64-90, :97-121)."""
    from ilqr_amd import systems
    s = getattr(systems, name)()
    rng = np.random.default_rng(5)
    x = rng.normal(size=(200, s.state_size)) * 3.0
    u = rng.normal(size=(200, s.control_size))
    xn, fx, fu = systems.probe_model(s.device_model_id, s.dt, x, u)
    assert np.abs(xn - s.model_f(x, u)).max() <= 1e-14 * 10
    assert np.abs(fx - s.compute_df_dx(x, u)).max() <= 1e-14 * 10
    assert np.abs(fu - s.compute_df_du(x, u)).max() <= 1e-14 * 10


def test_probe_ltv_tables():
    from ilqr_amd import systems
    rng = np.random.default_rng(2)
    A = np.eye(12) + 0.05 * rng.normal(size=(6, 12, 12))
    Bm = 0.1 * rng.normal(size=(6, 12, 4))
    s = systems.LTVSystem(A, Bm)
    x, u = rng.normal(size=(70, 12)), rng.normal(size=(70, 4))
    xn, fx, fu = systems.probe_model(s.device_model_id, s.dt, x, u, t=4, tables=(A, Bm))
    assert np.abs(xn - (x @ A[4].T + u @ Bm[4].T)).max() <= 1e-13
    assert np.array_equal(fx[3], A[4]) and np.array_equal(fu[69], Bm[4])


def _well_conditioned(oracle, d, p, r, rel_perturbation=1e-13, bound=1e-11):
    """Problems whose cost history the oracle itself reproduces to `bound` when the initial inputs
    are perturbed by `rel_perturbation`: only there is a 1e-9 comparison of two correct
    implementations meaningful (an iLQR iteration can amplify last-bit differences by orders of
    magnitude per iteration when steps are rejected or the problem is badly scaled)."""
    rng = np.random.default_rng(0)
    ui = p["init_inputs"] * (1.0 + rel_perturbation * rng.choice([-1.0, 1.0], size=p["init_inputs"].shape))
    r2 = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), ui, nthreads=1)
    ok = np.zeros(len(ui), dtype=bool)
    for b in range(len(ui)):
        n = min(int(r["iters"][b]), int(r2["iters"][b]))
        rel = np.abs(r["hist_cost"][b, :n] - r2["hist_cost"][b, :n]) / np.abs(r["hist_cost"][b, :n])
        ok[b] = abs(int(r["iters"][b]) - int(r2["iters"][b])) <= 1 and rel.max() <= bound
    return ok


@pytest.mark.parametrize("make,gen,B", [(_bicycle, "bicycle_batch", 48), (_pendulum, "pendulum_batch", 40)])
def test_batched_solve_matches_oracle(oracle, make, gen, B):
    from ilqr_amd import problems
    from ilqr_amd.ilqr import iterative_LQR
    s = make()
    p = getattr(problems, gen)(B)
    N = p["target_states"].shape[1]
    opt = iterative_LQR(s, p["target_states"], s.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    opt()
    d = oracle.make_desc(s, N)
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy(), nthreads=1)
    stable = _well_conditioned(oracle, d, p, r)
    assert stable.mean() >= 0.5, f"only {stable.sum()} of {B} problems are well conditioned"
    assert np.all(opt.status == 0)
    for b in range(B):
        n = min(int(opt.iterations[b]), int(r["iters"][b]))
        rel = np.abs(opt.cost_history[b, :n] - r["hist_cost"][b, :n]) / np.abs(r["hist_cost"][b, :n])
        assert rel[0] <= 1e-11, (b, rel[0])          # one iteration: no room for amplification
        if stable[b]:
            assert n >= int(r["iters"][b]) - 3
            assert rel.max() <= 1e-9, (b, rel.max())


def test_dense_weights_and_fp32_variants(oracle):
    """Dense cost weights select another kernel instantiation of the same snippets; fp32 is the
    separate reduced-precision mode (checked loosely, first iterations only)."""
    from ilqr_amd import problems
    from ilqr_amd.ilqr import iterative_LQR
    s = _bicycle(dense=True)
    p = problems.bicycle_batch(20)
    opt = iterative_LQR(s, p["target_states"], s.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    opt()
    d = oracle.make_desc(s, 150)
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy(), nthreads=1)
    stable = _well_conditioned(oracle, d, p, r)
    assert stable.sum() >= 10
    for b in range(20):
        n = min(int(opt.iterations[b]), int(r["iters"][b]))
        rel = np.abs(opt.cost_history[b, :n] - r["hist_cost"][b, :n]) / np.abs(r["hist_cost"][b, :n])
        assert rel[0] <= 1e-11
        assert rel.max() <= 1e-9 or not stable[b]
    s32 = _bicycle()
    o32 = iterative_LQR(s32, p["target_states"], s32.dt, history=True, dtype=np.float32)
    o32.inputs, o32.states = p["init_inputs"].copy(), p["init_states"].copy()
    o32()
    d = oracle.make_desc(s32, 150)
    r = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy(), nthreads=1)
    rel0 = np.abs(o32.cost_history[:, 0] - r["hist_cost"][:, 0]) / np.abs(r["hist_cost"][:, 0])
    assert rel0.max() <= 1e-3
    assert np.all(o32.min_cost <= r["cost0"])


def test_compacted_solve_is_bit_identical():
    """Active-set compaction and the lane mappings work the same for a run-time compiled model."""
    from ilqr_amd import problems
    from ilqr_amd.ilqr import iterative_LQR
    s = _bicycle()
    p = problems.bicycle_batch(300)
    outs = []
    for kw in (dict(lanes_per_problem=16, alphas_per_lane=1), dict(lanes_per_problem=1, alphas_per_lane=6, repack_every=3),
               dict(lanes_per_problem=2, alphas_per_lane=4, recompute_states=-1)):
        o = iterative_LQR(s, p["target_states"], s.dt, history=True, **kw)
        o.inputs, o.states = p["init_inputs"].copy(), p["init_states"].copy()
        o()
        outs.append(o)
    for o in outs[1:]:
        assert np.array_equal(o.min_cost, outs[0].min_cost)
        assert np.array_equal(o.states, outs[0].states) and np.array_equal(o.inputs, outs[0].inputs)
        assert np.array_equal(o.alpha_history, outs[0].alpha_history)


def test_numpy_only_model_is_recorded_and_solved(oracle):
    """A System with numpy methods only (no device code, written like the reference's models) is
    recorded, compiled and solved; checked against the reference-generated bicycle fixtures and
    against the hand-written device snippets of the same dynamics."""
    from conftest import load_golden, noise_floor_mask
    from ilqr_amd import problems, systems
    from ilqr_amd.ilqr import iterative_LQR
    from ilqr_amd.user_models import NumpyBicycle
    s = NumpyBicycle()
    s.set_cost(problems.BICYCLE_Q.copy(), problems.BICYCLE_R.copy())
    s.set_control_limit(*problems.BICYCLE_LIMITS)
    for name in ("bicycle_N150_seed0", "bicycle_N150_seed2", "bicycle_N150_seed5"):
        g = load_golden(name)
        opt = iterative_LQR(s, np.array(g["target_states"]), s.dt, history=True)
        opt.inputs, opt.states = np.array(g["init_inputs"]), np.array(g["init_states"])
        opt()
        n = min(int(opt.iterations), int(g["iters"]))
        assert n >= int(g["iters"]) - 3
        rel = np.abs(opt.cost_history[:n] - g["hist_cost"][:n]) / np.abs(g["hist_cost"][:n])
        assert rel.max() <= 1e-9, (name, rel.max())
        solid = noise_floor_mask(g["hist_cost"][:n], float(g["cost0"]))
        assert np.array_equal(opt.alpha_history[:n][solid], g["hist_alpha"][:n][solid])
    # the recorded dynamics evaluate like the numpy methods (no fused operations: to the last
    # bit of sine and cosine)
    from ilqr_amd import trace
    rng = np.random.default_rng(4)
    x, u = rng.normal(size=(100, 5)), rng.normal(size=(100, 2))
    xn, fx, fu = systems.probe_model(trace.traced_model(s), s.dt, x, u)
    for i in range(100):
        assert np.abs(xn[i] - s.model_f(x[i], u[i])).max() <= 1e-14 * 10
        assert np.abs(fx[i] - s.compute_df_dx(x[i], u[i])).max() <= 1e-14 * 10
        assert np.array_equal(fu[i], s.compute_df_du(x[i], u[i]))


def test_time_varying_model_with_shared_tables():
    """Snippets may read two tables shared by the batch (par_a / par_b) at the time index t."""
    from ilqr_amd import jit, systems
    from test_user_model_cpu import TABLE_FU, TABLE_FX, TABLE_JAC, TABLE_STEP
    mid = jit.load_model(jit.build_model(jit.model_source(2, 1, TABLE_STEP, TABLE_JAC, TABLE_FX, TABLE_FU)))
    rng = np.random.default_rng(9)
    a, b = rng.normal(size=20), rng.normal(size=20)
    x, u = rng.normal(size=(130, 2)), rng.normal(size=(130, 1))
    for t in (0, 7, 19):
        xn, fx, fu = systems.probe_model(mid, 0.1, x, u, t=t, tables=(a, b))
        assert np.array_equal(xn[:, 0], x[:, 0] + a[t] * x[:, 1] * 0.1)
        assert np.array_equal(xn[:, 1], x[:, 1] + b[t] * u[:, 0] * 0.1)
        assert np.array_equal(fx[5], [[1.0, a[t] * 0.1], [0.0, 1.0]]) and np.array_equal(fu[5], [[0.0], [b[t] * 0.1]])
