"""Dynamics / cost model setup with the interface of the reference ``systems.py``.

Same class names, constructor signatures, setter names and attribute names as
``systems.py:4-24`` (``System``), ``:27-57`` (``Car``), ``:59-90``
(``CarAcceleration``) and ``:92-121`` (``DubinsCar``), so model-setup code written
for the reference runs unchanged.  Each shipped class additionally carries a
``device_model_id`` that selects the matching hand-written HIP dynamics in the
solver kernels; the numpy ``model_f`` / ``compute_df_*`` methods are the
host-side statement of the same model (they broadcast over leading batch axes)
and are never called by the GPU solve.

``LTVSystem`` is an extension with no reference counterpart: a table-driven
time-varying linear model ``x+ = A_t x + B_t u`` (``model_f(x, u)`` of the
reference has no time argument, so such a system cannot be expressed there).
"""
import numpy as np

MODEL_CAR, MODEL_CARACC, MODEL_DUBINS, MODEL_LTV = 0, 1, 2, 3


class System:
    """Base class (``systems.py:4-24``)."""

    device_model_id = None

    def __init__(self, state_size, control_size):
        self.state_size = state_size
        self.control_size = control_size
        self.control_limited = False

    def set_dt(self, dt):
        self.dt = dt

    def set_cost(self, Q, R):
        # the terminal weight defaults to 100 x the stage weight (systems.py:16)
        self.Q = Q
        self.R = R
        self.Q_f = Q * 100

    def set_final_cost(self, Q_f):
        self.Q_f = Q_f

    def set_control_limit(self, lower_limit, upper_limit):
        self.control_limited = True
        self.control_lower_limit = lower_limit
        self.control_upper_limit = upper_limit


def _euler(x, rate, dt):
    return np.asarray(x, dtype=float) + np.stack(rate, axis=-1) * dt


class Car(System):
    """State (px, py, v, theta), control (acc, theta_rate); ``systems.py:27-57``."""

    device_model_id = MODEL_CAR

    def __init__(self):
        super().__init__(4, 2)
        self.dt = 0.2
        self.control_bound = np.array([0.5, 5])  # present in the reference, unused there too

    def model_f(self, x, u):
        x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
        v, th = x[..., 2], x[..., 3]
        return _euler(x, [v * np.cos(th), v * np.sin(th), u[..., 0], u[..., 1]], self.dt)

    def compute_df_dx(self, x, u):
        x = np.asarray(x, dtype=float)
        v, th = x[..., 2], x[..., 3]
        J = np.zeros(x.shape[:-1] + (4, 4)) + np.eye(4)
        J[..., 0, 2] = np.cos(th) * self.dt
        J[..., 0, 3] = -np.sin(th) * v * self.dt
        J[..., 1, 2] = np.sin(th) * self.dt
        J[..., 1, 3] = np.cos(th) * v * self.dt
        return J

    def compute_df_du(self, x, u):
        x = np.asarray(x, dtype=float)
        J = np.zeros(x.shape[:-1] + (4, 2))
        J[..., 2, 0] = self.dt
        J[..., 3, 1] = self.dt
        return J


class CarAcceleration(System):
    """State (px, py, v, acc, theta), control (jerk, theta_rate); ``systems.py:59-90``."""

    device_model_id = MODEL_CARACC

    def __init__(self):
        super().__init__(5, 2)
        self.dt = 0.2

    def model_f(self, x, u):
        x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
        v, acc, th = x[..., 2], x[..., 3], x[..., 4]
        return _euler(x, [v * np.cos(th), v * np.sin(th), acc, u[..., 0], u[..., 1]], self.dt)

    def compute_df_dx(self, x, u):
        x = np.asarray(x, dtype=float)
        v, th = x[..., 2], x[..., 4]
        J = np.zeros(x.shape[:-1] + (5, 5)) + np.eye(5)
        J[..., 0, 2] = np.cos(th) * self.dt
        J[..., 0, 4] = -np.sin(th) * v * self.dt
        J[..., 1, 2] = np.sin(th) * self.dt
        J[..., 1, 4] = np.cos(th) * v * self.dt
        J[..., 2, 3] = self.dt
        return J

    def compute_df_du(self, x, u):
        x = np.asarray(x, dtype=float)
        J = np.zeros(x.shape[:-1] + (5, 2))
        J[..., 3, 0] = self.dt
        J[..., 4, 1] = self.dt
        return J


class DubinsCar(System):
    """State (px, py, theta), control (v, curvature); ``systems.py:92-121``.
    The control Jacobian depends on the state and the control."""

    device_model_id = MODEL_DUBINS

    def __init__(self):
        super().__init__(3, 2)
        self.dt = 0.2

    def model_f(self, x, u):
        x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
        th, v, curv = x[..., 2], u[..., 0], u[..., 1]
        return _euler(x, [v * np.cos(th), v * np.sin(th), v * curv], self.dt)

    def compute_df_dx(self, x, u):
        x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
        th, v = x[..., 2], u[..., 0]
        J = np.zeros(x.shape[:-1] + (3, 3)) + np.eye(3)
        J[..., 0, 2] = -np.sin(th) * v * self.dt
        J[..., 1, 2] = np.cos(th) * v * self.dt
        return J

    def compute_df_du(self, x, u):
        x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
        th, v, curv = x[..., 2], u[..., 0], u[..., 1]
        J = np.zeros(x.shape[:-1] + (3, 2))
        J[..., 0, 0] = np.cos(th) * self.dt
        J[..., 1, 0] = np.sin(th) * self.dt
        J[..., 2, 0] = curv * self.dt
        J[..., 2, 1] = v * self.dt
        return J


class LTVSystem(System):
    """Table-driven linear time-varying model ``x+ = A[t] x + B[t] u`` shared by
    the whole batch (``A (N-1, n, n)``, ``B (N-1, n, m)``).  No reference
    counterpart; used for the synthetic quadrotor-sized configuration."""

    device_model_id = MODEL_LTV

    def __init__(self, A, B, dt=0.1):
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        if A.ndim != 3 or B.ndim != 3 or A.shape[0] != B.shape[0] or A.shape[1] != A.shape[2] \
                or B.shape[1] != A.shape[1]:
            raise ValueError("LTVSystem: A must be (T, n, n) and B (T, n, m)")
        super().__init__(A.shape[1], B.shape[2])
        self.A, self.B = A, B
        self.dt = dt

    def model_f(self, x, u, t=0):
        return self.A[t] @ np.asarray(x, dtype=float) + self.B[t] @ np.asarray(u, dtype=float)

    def compute_df_dx(self, x, u, t=0):
        return self.A[t].copy()

    def compute_df_du(self, x, u, t=0):
        return self.B[t].copy()
