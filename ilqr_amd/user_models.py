"""Models beyond the three of the reference's ``systems.py``, written the way a
user of the library writes one: the numpy methods of the ``System`` duck type
(``systems.py:4-24``; the reference's own solver runs on them unchanged, which is
how their fixtures under ``tests/golden/`` were produced) plus the device
snippets of ``systems.DeviceSystem``.
"""
import numpy as np

from .systems import DeviceSystem, System


class KinematicBicycle(DeviceSystem):
    """Rear-axle kinematic bicycle with steering-angle state.

    State ``(px, py, theta, v, delta)``, control ``(acc, steering rate)``, explicit Euler like
    the reference's models (``systems.py:33-40``)::

        px+ = px + dt v cos(theta)        v+     = v + dt acc
        py+ = py + dt v sin(theta)        delta+ = delta + dt rate
        theta+ = theta + dt v tan(delta) / wheelbase
    """

    wheelbase = 2.5

    device_step = """
        T s, c, sd, cd;
        sincos_<FAST>(x[2], s, c, bad);
        sincos_<FAST>(x[4], sd, cd, bad);
        xn[0] = fma_(x[3] * c, dt, x[0]);
        xn[1] = fma_(x[3] * s, dt, x[1]);
        xn[2] = fma_(x[3] * (sd / cd) / T(2.5), dt, x[2]);
        xn[3] = fma_(u[0], dt, x[3]);
        xn[4] = fma_(u[1], dt, x[4]);
    """
    device_jac = """
        T s, c, sd, cd;
        sincos_<FAST>(x[2], s, c, bad);
        sincos_<FAST>(x[4], sd, cd, bad);
        fx[0][2] = (-s * x[3]) * dt;
        fx[0][3] = c * dt;
        fx[1][2] = (c * x[3]) * dt;
        fx[1][3] = s * dt;
        fx[2][3] = ((sd / cd) / T(2.5)) * dt;
        fx[2][4] = (x[3] / (T(2.5) * cd * cd)) * dt;
        fu[3][0] = dt;
        fu[4][1] = dt;
    """
    fx_pattern = [[2, 0, 1, 1, 0],
                  [0, 2, 1, 1, 0],
                  [0, 0, 2, 1, 1],
                  [0, 0, 0, 2, 0],
                  [0, 0, 0, 0, 2]]
    fu_pattern = [[0, 0], [0, 0], [0, 0], [1, 0], [0, 1]]

    def __init__(self):
        super().__init__(5, 2)
        self.dt = 0.1

    def model_f(self, x, u):
        x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
        th, v, de = x[..., 2], x[..., 3], x[..., 4]
        rate = [v * np.cos(th), v * np.sin(th), v * np.tan(de) / self.wheelbase, u[..., 0], u[..., 1]]
        return x + np.stack(rate, axis=-1) * self.dt

    def compute_df_dx(self, x, u):
        x = np.asarray(x, dtype=float)
        th, v, de = x[..., 2], x[..., 3], x[..., 4]
        J = np.zeros(x.shape[:-1] + (5, 5)) + np.eye(5)
        J[..., 0, 2] = -np.sin(th) * v * self.dt
        J[..., 0, 3] = np.cos(th) * self.dt
        J[..., 1, 2] = np.cos(th) * v * self.dt
        J[..., 1, 3] = np.sin(th) * self.dt
        J[..., 2, 3] = np.tan(de) / self.wheelbase * self.dt
        J[..., 2, 4] = v / (self.wheelbase * np.cos(de) ** 2) * self.dt
        return J

    def compute_df_du(self, x, u):
        x = np.asarray(x, dtype=float)
        J = np.zeros(x.shape[:-1] + (5, 2))
        J[..., 3, 0] = self.dt
        J[..., 4, 1] = self.dt
        return J


class Pendulum(DeviceSystem):
    """Torque-driven pendulum, state ``(angle, rate)``, control ``(torque,)``; dense Jacobian
    patterns are left at their default (every entry treated as non-zero)."""

    gravity_over_length = 9.81

    device_step = """
        T s, c;
        sincos_<FAST>(x[0], s, c, bad);
        xn[0] = fma_(x[1], dt, x[0]);
        xn[1] = fma_(u[0] - T(9.81) * s, dt, x[1]);
    """
    device_jac = """
        T s, c;
        sincos_<FAST>(x[0], s, c, bad);
        fx[0][0] = T(1);
        fx[0][1] = dt;
        fx[1][0] = (T(-9.81) * c) * dt;
        fx[1][1] = T(1);
        fu[1][0] = dt;
    """

    def __init__(self):
        super().__init__(2, 1)
        self.dt = 0.05

    def model_f(self, x, u):
        x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
        rate = [x[..., 1], u[..., 0] - self.gravity_over_length * np.sin(x[..., 0])]
        return x + np.stack(rate, axis=-1) * self.dt

    def compute_df_dx(self, x, u):
        x = np.asarray(x, dtype=float)
        J = np.zeros(x.shape[:-1] + (2, 2)) + np.eye(2)
        J[..., 0, 1] = self.dt
        J[..., 1, 0] = -self.gravity_over_length * np.cos(x[..., 0]) * self.dt
        return J

    def compute_df_du(self, x, u):
        x = np.asarray(x, dtype=float)
        J = np.zeros(x.shape[:-1] + (2, 1))
        J[..., 1, 0] = self.dt
        return J


class NumpyBicycle(System):
    """The same bicycle with numpy methods ONLY, in the idiom of the reference's own models
    (``systems.py:33-57``: scalars picked out of ``x`` and ``u``, results assembled with
    ``np.array``).  It has no device code at all: the solver records these methods once
    (``ilqr_amd/trace.py``) and compiles the record."""

    def __init__(self):
        super().__init__(5, 2)
        self.dt = 0.1
        self.wheelbase = 2.5

    def model_f(self, x, u):
        theta, v, delta = x[2], x[3], x[4]
        rate = np.array([[v * np.cos(theta)], [v * np.sin(theta)], [v * np.tan(delta) / self.wheelbase],
                         [u[0]], [u[1]]])
        return np.reshape(np.reshape(x, (-1, 1)) + rate * self.dt, (-1,))

    def compute_df_dx(self, x, u):
        theta, v, delta = x[2], x[3], x[4]
        L, dt = self.wheelbase, self.dt
        return np.array([[1.0, 0.0, -np.sin(theta) * v * dt, np.cos(theta) * dt, 0.0],
                         [0.0, 1.0, np.cos(theta) * v * dt, np.sin(theta) * dt, 0.0],
                         [0.0, 0.0, 1.0, np.tan(delta) / L * dt, v / (L * np.cos(delta) ** 2) * dt],
                         [0.0, 0.0, 0.0, 1.0, 0.0],
                         [0.0, 0.0, 0.0, 0.0, 1.0]])

    def compute_df_du(self, x, u):
        return np.array([[0.0, 0.0], [0.0, 0.0], [0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]) * self.dt


class NumpyCar(System):
    """The car of ``systems.py:27-57`` (state ``(px, py, v, theta)``, control ``(acc, theta rate)``)
    stated with numpy methods only, for the recorder; its state starts with ``(px, py, v)``, so the
    rows of ``constraints.py`` apply and the sequential-QP solver accepts it."""

    device_has_barrier = True

    def __init__(self):
        super().__init__(4, 2)
        self.dt = 0.2

    def model_f(self, x, u):
        v, theta = x[2], x[3]
        rate = np.array([v * np.cos(theta), v * np.sin(theta), u[0], u[1]])
        return x + rate * self.dt

    def compute_df_dx(self, x, u):
        v, theta, dt = x[2], x[3], self.dt
        return np.array([[1.0, 0.0, np.cos(theta) * dt, -np.sin(theta) * v * dt],
                         [0.0, 1.0, np.sin(theta) * dt, np.cos(theta) * v * dt],
                         [0.0, 0.0, 1.0, 0.0],
                         [0.0, 0.0, 0.0, 1.0]])

    def compute_df_du(self, x, u):
        return np.array([[0.0, 0.0], [0.0, 0.0], [self.dt, 0.0], [0.0, self.dt]])
