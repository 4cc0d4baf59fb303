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

``DeviceSystem`` is the base class for any other model: a subclass keeps the
reference's numpy methods and adds the same dynamics as a device-code snippet,
which is compiled on first use (``ilqr_amd/jit.py``).
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

    # The kernels built into the library serve n = 12, m = 4 with diagonal weights (matrix unit).
    # Any other size, or dense weights, gets the same table model compiled at run time for the
    # lane-per-problem kernels, once per (n, m, element type, diagonal or dense weights).
    _STEP = ("const T *A_ = par_a + (size_t)t * NX * NX, *B_ = par_b + (size_t)t * NX * NU;\n"
             "_Pragma(\"unroll\") for (int i = 0; i < NX; ++i) {\n"
             "    T s = A_[i * NX] * x[0];\n"
             "    _Pragma(\"unroll\") for (int j = 1; j < NX; ++j) s = fma_(A_[i * NX + j], x[j], s);\n"
             "    _Pragma(\"unroll\") for (int j = 0; j < NU; ++j) s = fma_(B_[i * NU + j], u[j], s);\n"
             "    xn[i] = s;\n"
             "}")
    _JAC = ("const T *A_ = par_a + (size_t)t * NX * NX, *B_ = par_b + (size_t)t * NX * NU;\n"
            "_Pragma(\"unroll\") for (int i = 0; i < NX; ++i) {\n"
            "    _Pragma(\"unroll\") for (int j = 0; j < NX; ++j) fx[i][j] = A_[i * NX + j];\n"
            "    _Pragma(\"unroll\") for (int j = 0; j < NU; ++j) fu[i][j] = B_[i * NU + j];\n"
            "}")
    _compiled = {}

    @property
    def device_tables(self):
        return (self.A, self.B)

    def builtin_kernels(self, diag):
        """True if the library's own n = 12, m = 4 kernels serve this system."""
        return self.state_size == 12 and self.control_size == 4 and bool(diag)

    def device_model(self, dtype=np.float64, diag=True, verbose=False):
        """Model id of the run-time compiled table model of this size."""
        from . import jit
        key = (self.state_size, self.control_size, "f64" if np.dtype(dtype) == np.float64 else "f32", bool(diag))
        if key not in LTVSystem._compiled:
            src = jit.model_source(self.state_size, self.control_size, self._STEP, self._JAC, None, None, False,
                                   variants=(key[2:],))
            LTVSystem._compiled[key] = jit.load_model(jit.build_model(src, verbose=verbose))
        return LTVSystem._compiled[key]

    def model_f(self, x, u, t=0):
        return self.A[t] @ np.asarray(x, dtype=float) + self.B[t] @ np.asarray(u, dtype=float)

    def compute_df_dx(self, x, u, t=0):
        return self.A[t].copy()

    def compute_df_du(self, x, u, t=0):
        return self.B[t].copy()


class DeviceSystem(System):
    """A user-defined model for the GPU solve.

    The reference solver works with any ``System`` subclass because it calls
    ``model_f`` / ``compute_df_dx`` / ``compute_df_du`` in Python at every time step
    (``ilqr.py:62``, ``:87-88``).  The kernels cannot, so a subclass states the model twice:
    the numpy methods exactly as for the reference (they stay the host-side definition and
    are what ``check_device_model`` compares against), and two device snippets::

        class Pendulum(DeviceSystem):
            device_step = "T s, c; sincos_<FAST>(x[0], s, c, bad);" \
                          "xn[0] = fma_(x[1], dt, x[0]); xn[1] = fma_(u[0] - T(9.81) * s, dt, x[1]);"
            device_jac = "T s, c; sincos_<FAST>(x[0], s, c, bad);" \
                         "fx[0][1] = dt; fx[1][0] = (T(-9.81) * c) * dt; fu[1][0] = dt;"
            fx_pattern = [[2, 1], [1, 2]]      # 0 = always zero, 1 = non-zero, 2 = always one
            fu_pattern = [[0], [1]]

    ``ilqr_amd/jit.py`` documents the names a snippet sees.  Without patterns every Jacobian
    entry is treated as non-zero (correct, slower); with them the Riccati code skips the
    structural zeros at compile time, as it does for the built-in models.  ``device_tables``
    may hold two arrays shared by the whole batch that the snippets read as ``par_a`` /
    ``par_b`` (per-problem parameters are better carried as extra state components with
    identity dynamics and zero cost weight).
    """

    device_step = None
    device_jac = None
    fx_pattern = None
    fu_pattern = None
    device_has_barrier = False   # state starts with (px, py, v): constraints.py:20-26 apply
    device_tables = None

    def device_model(self, dtype=np.float64, diag=True, verbose=False):
        """Compile (once) and register the device dynamics; returns the model id."""
        from . import jit
        if self.device_step is None or self.device_jac is None:
            raise ValueError(f"{type(self).__name__}: device_step and device_jac must be set")
        variant = ("f64" if np.dtype(dtype) == np.float64 else "f32", bool(diag))
        cache = self.__dict__.setdefault("_device_models", {})
        key = (variant, self.device_step, self.device_jac)
        if key not in cache:
            src = jit.model_source(self.state_size, self.control_size, self.device_step, self.device_jac,
                                   self.fx_pattern, self.fu_pattern, self.device_has_barrier, variants=(variant,))
            cache[key] = jit.load_model(jit.build_model(src, verbose=verbose))
        return cache[key]

    def check_device_model(self, x, u, dtype=np.float64, device=None, t=0):
        """Evaluate the device snippets at the points ``x (P, n)``, ``u (P, m)`` and return the
        largest absolute differences to ``model_f``, ``compute_df_dx`` and ``compute_df_du``
        (called point by point, as the reference solver calls them)."""
        model = self.device_model(dtype, True)
        xn, fx, fu = probe_model(model, self.dt, x, u, dtype=dtype, device=device, t=t, tables=self.device_tables)
        x, u = np.atleast_2d(np.asarray(x, dtype=float)), np.atleast_2d(np.asarray(u, dtype=float))
        ef = max(np.abs(xn[i] - np.asarray(self.model_f(x[i], u[i]), dtype=float)).max() for i in range(len(x)))
        ex = max(np.abs(fx[i] - np.asarray(self.compute_df_dx(x[i], u[i]), dtype=float)).max() for i in range(len(x)))
        eu = max(np.abs(fu[i] - np.asarray(self.compute_df_du(x[i], u[i]), dtype=float)).max() for i in range(len(x)))
        return ef, ex, eu


def probe_model(model_id, dt, x, u, dtype=np.float64, device=None, t=0, tables=None):
    """Device dynamics of ``model_id`` at the points ``x (P, n)``, ``u (P, m)``:
    returns ``(x_next (P, n), fx (P, n, n), fu (P, n, m))`` as float64 numpy arrays
    (``ilqr_model_probe`` of the C ABI)."""
    import ctypes as C
    import torch
    from . import _lib
    if not torch.cuda.is_available():
        raise _lib.IlqrError("no GPU visible: the device dynamics can only be evaluated on the HIP device")
    lib = _lib.lib()
    npd = np.dtype(dtype)
    td = torch.float64 if npd == np.float64 else torch.float32
    dev = torch.device(device if device is not None else "cuda")
    n, m = C.c_int32(0), C.c_int32(0)
    _lib.check(lib.ilqr_model_dims(int(model_id), C.byref(n), C.byref(m)), "ilqr_model_dims")
    n, m = n.value, m.value
    x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=npd)))
    u = np.ascontiguousarray(np.atleast_2d(np.asarray(u, dtype=npd)))
    P = x.shape[0]
    if x.shape != (P, n) or u.shape != (P, m):
        raise ValueError(f"x must be (P, {n}) and u (P, {m})")
    with torch.cuda.device(dev):
        xd, ud = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
        xn = torch.zeros(P * n, dtype=td, device=dev)
        fx = torch.zeros(P * n * n, dtype=td, device=dev)
        fu = torch.zeros(P * n * m, dtype=td, device=dev)
        ta = tb = None
        if tables is not None:
            ta = torch.from_numpy(np.ascontiguousarray(tables[0], dtype=npd)).to(dev)
            tb = torch.from_numpy(np.ascontiguousarray(tables[1], dtype=npd)).to(dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.ilqr_model_probe(int(model_id), _lib.F64 if npd == np.float64 else _lib.F32, float(dt), int(t),
                                        ta.data_ptr() if ta is not None else None,
                                        tb.data_ptr() if tb is not None else None,
                                        xd.data_ptr(), ud.data_ptr(), xn.data_ptr(), fx.data_ptr(), fu.data_ptr(),
                                        P, stream), "ilqr_model_probe")
        out = (xn.cpu().numpy().astype(np.float64).reshape(P, n),
               fx.cpu().numpy().astype(np.float64).reshape(P, n, n),
               fu.cpu().numpy().astype(np.float64).reshape(P, n, m))
    return out
