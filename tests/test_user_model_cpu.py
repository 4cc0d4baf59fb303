"""CPU tests of the run-time compiled dynamics (ilqr_amd/jit.py, systems.DeviceSystem):
source generation, cross-compilation for gfx950, registration through the C ABI and the
host-side refusals.  No kernel is launched here; the GPU tests solve with these models."""
import ctypes as C
import os

import numpy as np
import pytest

from ilqr_amd import _lib, jit
from ilqr_amd.user_models import KinematicBicycle, Pendulum


def _desc(model, n, m, B=64, N=10, dtype=_lib.F64, constraint_mode=0):
    d = _lib.Desc()
    d.B, d.N, d.n, d.m, d.model, d.dtype = B, N, n, m, model, dtype
    d.dt, d.alpha_floor = 0.1, 1e-4
    d.constraint_mode = constraint_mode
    d.barrier_mu, d.barrier_delta = 1.0, 0.1
    for i in range(n):
        d.Q[i * n + i] = d.Qf[i * n + i] = 1.0
    for i in range(m):
        d.R[i * m + i] = 1.0
    return d


def _create(d):
    h = C.c_void_p()
    rc = _lib.lib().ilqr_create(C.byref(d), C.byref(h))
    if rc == 0:
        _lib.lib().ilqr_destroy(h)
    return rc


def test_model_source_states_patterns_and_variants():
    s = KinematicBicycle()
    src = jit.model_source(5, 2, s.device_step, s.device_jac, s.fx_pattern, s.fu_pattern,
                           variants=(("f64", True), ("f32", False)))
    assert "NX = 5, NU = 2" in src
    assert "Launch<double, UserModel, true>" in src and "Launch<float, UserModel, false>" in src
    assert "Launch<double, UserModel, false>" not in src
    # first row of fx_pattern [2, 0, 1, 1, 0]: non-zero mask and constant-one mask
    assert "tab[NX * NX] = {true, false, true, true, false," in src
    assert "tab[NX * NX] = {true, false, false, false, false," in src
    assert "SqpLaunch" not in src.split("user_sqp")[1]          # no (px, py, v) layout declared
    src_b = jit.model_source(5, 2, s.device_step, s.device_jac, has_barrier=True)
    assert "SqpLaunch<double, UserModel>::table()" in src_b.split("user_sqp")[1]
    with pytest.raises(ValueError):
        jit.model_source(5, 2, "", "", fx_pattern=[[1, 0]])
    with pytest.raises(ValueError):
        jit.model_source(13, 2, "", "")
    with pytest.raises(ValueError):
        jit.model_source(2, 1, "", "", fu_pattern=[[2], [0]])
    with pytest.raises(ValueError):
        jit.model_source(2, 1, "", "", has_barrier=True)


def test_build_load_and_create():
    lib = _lib.lib()
    s = KinematicBicycle()
    mid = s.device_model(np.float64, True)
    assert mid >= _lib.MODEL_USER_BASE
    assert s.device_model(np.float64, True) == mid          # cached per instance
    assert KinematicBicycle().device_model(np.float64, True) == mid   # same library -> same id
    other = Pendulum().device_model(np.float64, True)
    assert other != mid
    n, m = C.c_int32(), C.c_int32()
    assert lib.ilqr_model_dims(mid, C.byref(n), C.byref(m)) == 0 and (n.value, m.value) == (5, 2)
    assert lib.ilqr_model_dims(other, C.byref(n), C.byref(m)) == 0 and (n.value, m.value) == (2, 1)
    assert lib.ilqr_model_dims(0, C.byref(n), C.byref(m)) == 0 and (n.value, m.value) == (4, 2)
    assert lib.ilqr_model_dims(999, C.byref(n), C.byref(m)) == -2
    assert _create(_desc(mid, 5, 2)) == 0
    assert _create(_desc(mid, 4, 2)) == -1                   # sizes must be the model's
    assert _create(_desc(mid, 5, 2, dtype=_lib.F32)) == -2   # that variant was not built
    assert _create(_desc(mid, 5, 2, constraint_mode=_lib.CONSTR_BARRIER)) == -2   # no (px, py, v) layout declared
    assert _create(_desc(_lib.MODEL_USER_BASE + 200, 5, 2)) == -2  # never handed out
    d = _desc(mid, 5, 2)
    d.Q[1] = 0.5                                              # dense weights: not built either
    assert _create(d) == -2


def test_load_refuses_foreign_and_missing_libraries():
    lib = _lib.lib()
    mid = C.c_int32(7)
    here = os.path.dirname(os.path.abspath(__file__))
    oracle_so = os.path.join(os.path.dirname(here), "oracle", "libilqr_oracle.so")
    if os.path.exists(oracle_so):
        assert lib.ilqr_load_model(oracle_so.encode(), C.byref(mid)) == -1 and mid.value == -1
    assert lib.ilqr_load_model(b"/nonexistent/model.so", C.byref(mid)) == -1
    assert lib.ilqr_load_model(None, C.byref(mid)) == -1


def test_compile_error_is_reported():
    with pytest.raises(_lib.IlqrError, match="compiling the device dynamics failed"):
        jit.build_model(jit.model_source(2, 1, "xn[0] = undefined_symbol;", ""))


@pytest.mark.parametrize("cls", [KinematicBicycle, Pendulum])
def test_host_statement_is_consistent(cls):
    """The numpy methods (the model's definition, what the reference solver would call):
    Jacobians against central differences of model_f."""
    s = cls()
    rng = np.random.default_rng(3)
    n, m = s.state_size, s.control_size
    for _ in range(20):
        x, u = rng.normal(size=n), rng.normal(size=m)
        fx, fu = s.compute_df_dx(x, u), s.compute_df_du(x, u)
        eps = 1e-6
        for j in range(n):
            e = np.zeros(n)
            e[j] = eps
            assert np.allclose((s.model_f(x + e, u) - s.model_f(x - e, u)) / (2 * eps), fx[:, j], atol=1e-7)
        for j in range(m):
            e = np.zeros(m)
            e[j] = eps
            assert np.allclose((s.model_f(x, u + e) - s.model_f(x, u - e)) / (2 * eps), fu[:, j], atol=1e-7)
        # the declared patterns cover every non-zero entry
        if s.fx_pattern is not None:
            P = np.asarray(s.fx_pattern)
            assert np.all(fx[P == 0] == 0) and np.all(fx[P == 2] == 1)
            assert np.all(fu[np.asarray(s.fu_pattern) == 0] == 0)


TABLE_STEP = "xn[0] = x[0] + par_a[t] * x[1] * dt; xn[1] = x[1] + par_b[t] * u[0] * dt;"
TABLE_JAC = "fx[0][1] = par_a[t] * dt; fu[1][0] = par_b[t] * dt;"
TABLE_FX = [[2, 1], [0, 2]]
TABLE_FU = [[0], [1]]


def test_model_that_reads_tables_cannot_be_bound_without_them():
    """A snippet that names par_a / par_b reads the shared tables: binding such a handle without
    them must fail on the host instead of letting a kernel read through a null pointer."""
    lib = _lib.lib()
    src = jit.model_source(2, 1, TABLE_STEP, TABLE_JAC, TABLE_FX, TABLE_FU)
    assert "needs_tables = 1" in src
    assert "needs_tables = 0" in jit.model_source(2, 1, Pendulum.device_step, Pendulum.device_jac)
    mid = jit.load_model(jit.build_model(src))
    h = C.c_void_p()
    assert lib.ilqr_create(C.byref(_desc(mid, 2, 1)), C.byref(h)) == 0
    b = _lib.Buffers()
    for name in ("X", "U", "Xref", "K", "k", "J", "converged", "iters", "alpha_idx", "work"):
        setattr(b, name, 4096)   # never dereferenced: ilqr_bind only records the pointers
    assert lib.ilqr_bind(h, C.byref(b)) == -3
    b.ltv_A = b.ltv_B = 4096
    assert lib.ilqr_bind(h, C.byref(b)) == 0
    lib.ilqr_destroy(h)
    # the probe refuses as well, before anything is launched
    assert lib.ilqr_model_probe(mid, _lib.F64, 0.1, 0, None, None, 4096, 4096, 4096, 4096, 4096, 1, None) == -3
