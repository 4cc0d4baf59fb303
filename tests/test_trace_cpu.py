"""CPU tests of ilqr_amd/trace.py: recording a System's numpy methods into device snippets."""
import ctypes as C

import numpy as np
import pytest

from ilqr_amd import _lib, trace
from ilqr_amd.systems import System
from ilqr_amd.user_models import KinematicBicycle, NumpyBicycle


def test_record_reproduces_the_methods_and_finds_the_sparsity():
    s = NumpyBicycle()
    t = trace.trace_system(s)
    hand = KinematicBicycle()
    assert np.array_equal(t["fx_pattern"], np.asarray(hand.fx_pattern))
    assert np.array_equal(t["fu_pattern"], np.asarray(hand.fu_pattern))
    # one sincos per distinct angle, no fused multiply-add, constants as literals of the element type
    assert t["step"].count("sincos_<FAST>") == 2 and "fma_" not in t["step"]
    assert "T(0.1)" in t["step"] and "xn[4] =" in t["step"]
    assert "fx[2][4] =" in t["jac"] and "fx[0][0]" not in t["jac"] and "fu[0][0]" not in t["jac"]
    # the record is checked against the methods inside trace_system; the same model stated the
    # vectorised way agrees with it
    rng = np.random.default_rng(0)
    x, u = rng.normal(size=5), rng.normal(size=2)
    assert np.allclose(s.model_f(x, u), hand.model_f(x, u), rtol=1e-15, atol=1e-15)
    assert np.allclose(s.compute_df_dx(x, u), hand.compute_df_dx(x, u), rtol=1e-15, atol=1e-15)


class _Exotic(System):
    """every supported function once"""

    def __init__(self):
        super().__init__(3, 2)
        self.dt = 0.01

    def model_f(self, x, u):
        a = np.exp(-x[0] ** 2) + np.sqrt(x[1] * x[1] + 1.0) - np.tanh(u[0])
        b = np.arctan2(x[1], 2.0 + np.cos(x[2])) * np.log(3.0 + np.sin(x[0])) / (1.5 + abs(u[1]))
        c = x[2] ** 3 - np.arctan(u[0]) + np.sinh(0.1 * x[0]) * np.cosh(0.1 * x[1]) + 2.0 ** (0.5 * x[2])
        return np.array([x[0] + a * self.dt, x[1] + b * self.dt, x[2] + c * self.dt])

    def compute_df_dx(self, x, u):
        return np.array([[1.0, 0.0, 0.0], [x[1] * self.dt, 1.0, 0.0], [0.0, 0.0, 1.0 + 0.0 * x[0]]])

    def compute_df_du(self, x, u):
        return np.zeros((3, 2)) + np.array([[u[0], 0.0], [0.0, 0.0], [0.0, 1.0]]) * self.dt


def test_function_table_and_constant_folding():
    t = trace.trace_system(_Exotic())
    for fn in ("exp(", "sqrt(", "tanh(", "atan2(", "log(", "fabs_(", "atan(", "sinh(", "cosh(", "pow("):
        assert fn in t["step"], fn
    assert t["fx_pattern"].tolist() == [[2, 0, 0], [1, 2, 0], [0, 0, 1]]   # 1 + 0*x stays an expression
    assert t["fu_pattern"].tolist() == [[1, 0], [0, 0], [0, 1]]
    assert "fu[2][1] = T(0.01);" in t["jac"]


class _Branchy(System):
    def __init__(self):
        super().__init__(2, 1)
        self.dt = 0.1

    def model_f(self, x, u):
        return np.array([x[0] + (x[1] if x[1] > 0 else 0.0) * self.dt, x[1] + u[0] * self.dt])

    def compute_df_dx(self, x, u):
        return np.eye(2)

    def compute_df_du(self, x, u):
        return np.array([[0.0], [self.dt]])


class _Coercing(_Branchy):
    def model_f(self, x, u):
        x = np.asarray(x, dtype=float)
        return x + np.array([x[1], u[0]]) * self.dt


class _Stateful(_Branchy):
    calls = 0

    def model_f(self, x, u):
        type(self).calls += 1
        return np.array([x[0] + x[1] * self.dt * type(self).calls, x[1] + u[0] * self.dt])


class _WrongShape(_Branchy):
    def model_f(self, x, u):
        return np.array([x[0] + x[1] * self.dt, x[1] + u[0] * self.dt])

    def compute_df_du(self, x, u):
        return np.array([[0.0, 0.0], [self.dt, 0.0]])


class _FloatCall(_Branchy):
    def model_f(self, x, u):
        return np.array([x[0] + float(x[1]) * self.dt, x[1] + u[0] * self.dt])


class _EqualityGuard(_Branchy):
    """a guard on equality: random check points would never hit the equal branch"""

    def model_f(self, x, u):
        v = x[1] if x[1] == 0.0 else x[1] * 1.0
        return np.array([x[0] + v * self.dt, x[1] + u[0] * self.dt])


class _NotEqualGuard(_Branchy):
    def model_f(self, x, u):
        g = 1.0 if x[0] != 0 else 0.0
        return np.array([x[0] + g * x[1] * self.dt, x[1] + u[0] * self.dt])


class _KeywordConversion(System):
    """np.array(object=...) and np.asarray(a=...) by keyword"""

    def __init__(self):
        super().__init__(2, 1)
        self.dt = 0.1

    def model_f(self, x, u):
        xs = np.asarray(a=x, dtype=float)
        return np.array(object=[xs[0] + xs[1] * self.dt, xs[1] + u[0] * self.dt], dtype=float)

    def compute_df_dx(self, x, u):
        return np.array(object=[[1.0, self.dt], [0.0, 1.0]])

    def compute_df_du(self, x, u):
        return np.array([[0.0], [self.dt]])


class _FillIn(System):
    """arrays allocated with numpy and filled by item assignment"""

    def __init__(self):
        super().__init__(3, 1)
        self.dt = 0.1

    def model_f(self, x, u):
        xn = np.zeros(3)
        xn[0] = x[0] + x[1] * self.dt
        xn[1] = x[1] + np.sin(x[2]) * self.dt
        xn[2] = x[2] + u[0] * self.dt
        return xn

    def compute_df_dx(self, x, u):
        J = np.eye(3)
        J[0, 1] = self.dt
        J[1, 2] = np.cos(x[2]) * self.dt
        return J

    def compute_df_du(self, x, u):
        J = np.zeros((3, 1))
        J[2, 0] = self.dt
        return J


def test_allocation_and_coercion_idioms_are_recorded():
    """np.zeros / np.eye + item assignment and np.asarray(x, dtype=float) do not stop the recorder;
    numpy itself is untouched afterwards.  The vectorised statements of the shipped models record to
    the sparsity patterns their hand-written device code declares."""
    from ilqr_amd import systems
    t = trace.trace_system(_FillIn())
    assert t["fx_pattern"].tolist() == [[2, 1, 0], [0, 2, 1], [0, 0, 2]] and t["fu_pattern"].tolist() == [[0], [0], [1]]
    assert "fx[1][2] =" in t["jac"] and t["step"].count("sincos_<FAST>") == 1
    t = trace.trace_system(_Coercing())
    assert "xn[1] =" in t["step"]
    hand = KinematicBicycle()
    t = trace.trace_system(hand)                 # np.asarray(x, dtype=float), x[..., i], np.stack, J[..., i, j] = ...
    assert np.array_equal(t["fx_pattern"], hand.fx_pattern) and np.array_equal(t["fu_pattern"], hand.fu_pattern)
    car = trace.trace_system(systems.Car())
    assert car["fx_pattern"].tolist() == [[2, 0, 1, 1], [0, 2, 1, 1], [0, 0, 2, 0], [0, 0, 0, 2]]   # CarModel::fx_nz / fx_one
    assert car["fu_pattern"].tolist() == [[0, 0], [0, 0], [1, 0], [0, 1]]
    dub = trace.trace_system(systems.DubinsCar())
    assert dub["fu_pattern"].tolist() == [[1, 0], [1, 0], [1, 1]]                                      # DubinsModel::fu_nz
    assert np.zeros(2).dtype == np.float64 and np.eye(2).dtype == np.float64
    assert np.asarray([1, 2], dtype=float).dtype == np.float64


def test_keyword_conversions_are_recorded():
    t = trace.trace_system(_KeywordConversion())
    assert "xn[0] =" in t["step"] and t["fx_pattern"].tolist() == [[2, 1], [0, 2]]
    assert np.array(object=[1, 2], dtype=float).dtype == np.float64


@pytest.mark.parametrize("cls,needle", [(_Branchy, "compares or converts"), (_FloatCall, "compares or converts"),
                                        (_EqualityGuard, "compares or converts"), (_NotEqualGuard, "compares or converts"),
                                        (_Stateful, "does not compute the same values"),
                                        (_WrongShape, "returned 4 values, expected 2")])
def test_untraceable_models_are_refused_with_a_reason(cls, needle):
    with pytest.raises(trace.TraceError, match=needle):
        trace.trace_system(cls())


def test_traced_model_compiles_and_registers():
    mid = trace.traced_model(NumpyBicycle(), np.float64, True)
    assert mid >= _lib.MODEL_USER_BASE
    n, m = C.c_int32(), C.c_int32()
    assert _lib.lib().ilqr_model_dims(mid, C.byref(n), C.byref(m)) == 0 and (n.value, m.value) == (5, 2)
    assert trace.traced_model(NumpyBicycle(), np.float64, True) == mid
