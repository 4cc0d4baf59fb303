"""ctypes front end of the CPU oracle (``oracle/ilqr_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, by ``__graft_entry__.smoke``
and by the ``cpu_baseline`` leg of ``bench.py`` as the checker / CPU baseline.
The shipped package ``ilqr_amd`` never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libilqr_oracle.so")
MAXN, MAXM = 12, 4
MODEL_IDS = {"Car": 0, "CarAcceleration": 1, "DubinsCar": 2, "LTVSystem": 3}


class Desc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("N", C.c_int32), ("model", C.c_int32),
        ("sys_control_limited", C.c_int32), ("solver_control_limited", C.c_int32),
        ("barrier", C.c_int32), ("max_iter", C.c_int32),
        ("dt", C.c_double), ("lm", C.c_double), ("alpha_floor", C.c_double),
        ("input_limit_lambda", C.c_double), ("mu", C.c_double), ("delta", C.c_double),
        ("Q", C.c_double * (MAXN * MAXN)), ("R", C.c_double * (MAXM * MAXM)),
        ("Qf", C.c_double * (MAXN * MAXN)),
        ("lo", C.c_double * MAXM), ("hi", C.c_double * MAXM),
        ("vel_bounds", C.c_double * 2),
        ("ltv_A", C.c_void_p), ("ltv_B", C.c_void_p),
        ("user_f", C.c_void_p), ("user_jac", C.c_void_p),
    ]


MODEL_CALLBACK = 100
_USER_F = C.CFUNCTYPE(None, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
_USER_JAC = C.CFUNCTYPE(None, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                        C.POINTER(C.c_double))


def _callbacks(system):
    """The System's own numpy methods as C callbacks: any object with the duck type of
    ``systems.py:4-24`` can then be solved by the oracle, as by the reference (single thread
    only: the callbacks run Python)."""
    n, m = int(system.state_size), int(system.control_size)

    def f(t, x, u, xn):
        xa, ua = np.ctypeslib.as_array(x, (n,)), np.ctypeslib.as_array(u, (m,))
        np.ctypeslib.as_array(xn, (n,))[:] = system.model_f(xa.copy(), ua.copy())

    def jac(t, x, u, fx, fu):
        xa, ua = np.ctypeslib.as_array(x, (n,)).copy(), np.ctypeslib.as_array(u, (m,)).copy()
        np.ctypeslib.as_array(fx, (n * n,))[:] = np.asarray(system.compute_df_dx(xa, ua), dtype=float).ravel()
        np.ctypeslib.as_array(fu, (n * m,))[:] = np.asarray(system.compute_df_du(xa, ua), dtype=float).ravel()

    return _USER_F(f), _USER_JAC(jac)


def build(force=False):
    """Compile the oracle with gcc if the shared object is missing or stale."""
    src = os.path.join(_HERE, "ilqr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libilqr_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_cost.restype = C.c_double
        _lib.orc_barrier.restype = C.c_double
        assert _lib.ilqr_oracle_desc_size() == C.sizeof(Desc)
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_desc(system, N, max_iter=30, alpha_floor=1e-4, lm=0.0, solver_control_limited=False,
              input_limit_lambda=1e-4, barrier=False, mu=0.0, delta=0.1, vel_bounds=(0.0, 0.0),
              keep=None):
    """Fill the oracle descriptor from a ``systems.py``-style System object
    (duck type of ``systems.py:4-24``)."""
    d = Desc()
    n, m = int(system.state_size), int(system.control_size)
    d.n, d.m, d.N = n, m, int(N)
    d.model = MODEL_IDS.get(type(system).__name__, MODEL_CALLBACK)
    if d.model == MODEL_CALLBACK:
        cf, cj = _callbacks(system)
        d._keep_cb = (cf, cj)  # the descriptor keeps the callback objects alive
        d.user_f, d.user_jac = C.cast(cf, C.c_void_p), C.cast(cj, C.c_void_p)
    d.sys_control_limited = int(bool(system.control_limited))
    d.solver_control_limited = int(bool(solver_control_limited))
    d.barrier = int(bool(barrier))
    d.max_iter = int(max_iter)
    d.dt = float(system.dt)
    d.lm = float(lm)
    d.alpha_floor = float(alpha_floor)
    d.input_limit_lambda = float(input_limit_lambda)
    d.mu, d.delta = float(mu), float(delta)
    Q = np.asarray(system.Q, dtype=np.float64).reshape(n, n)
    R = np.asarray(system.R, dtype=np.float64).reshape(m, m)
    Qf = np.asarray(system.Q_f, dtype=np.float64).reshape(n, n)
    for i, v in enumerate(Q.ravel()):
        d.Q[i] = v
    for i, v in enumerate(R.ravel()):
        d.R[i] = v
    for i, v in enumerate(Qf.ravel()):
        d.Qf[i] = v
    if system.control_limited or solver_control_limited:
        for j in range(m):
            d.lo[j] = float(system.control_lower_limit[j])
            d.hi[j] = float(system.control_upper_limit[j])
    d.vel_bounds[0], d.vel_bounds[1] = float(vel_bounds[0]), float(vel_bounds[1])
    if d.model == 3:
        A = np.ascontiguousarray(system.A, dtype=np.float64)
        Bm = np.ascontiguousarray(system.B, dtype=np.float64)
        if keep is not None:
            keep.extend([A, Bm])
        d._keep = (A, Bm)
        d.ltv_A, d.ltv_B = A.ctypes.data, Bm.ctypes.data
    return d


def cost(d, target, states, inputs, centers=None, radius=None):
    return lib().orc_cost(C.byref(d), _dp(target), _dp(states), _dp(inputs), _dp(centers),
                          _dp(radius))


def backward(d, target, states, inputs, centers=None, radius=None):
    K = np.zeros((d.N - 1, d.m, d.n))
    k = np.zeros((d.N - 1, d.m))
    rc = lib().orc_backward(C.byref(d), _dp(target), _dp(states), _dp(inputs), _dp(centers),
                            _dp(radius), _dp(K), _dp(k))
    return K, k, rc


def forward(d, target, states, inputs, K, k, min_cost, centers=None, radius=None):
    """In-place on states/inputs.  Returns (accepted_alpha_index, min_cost, converge, tries)."""
    mc = C.c_double(min_cost)
    cv = C.c_int32(0)
    tr = C.c_int32(0)
    acc = lib().orc_forward(C.byref(d), _dp(target), _dp(states), _dp(inputs), _dp(centers),
                            _dp(radius), _dp(K), _dp(k), C.byref(mc), C.byref(cv), C.byref(tr))
    return acc, mc.value, bool(cv.value), tr.value


def solve(d, target, states, inputs, centers=None, radius=None, converge=False, history=True):
    """One problem, in place on states/inputs.  Returns a dict with the final
    scalars and (if ``history``) per-iteration records."""
    N, n, m, mi = d.N, d.n, d.m, d.max_iter
    mc, cv, c0 = C.c_double(0.0), C.c_int32(int(converge)), C.c_double(0.0)
    out = {}
    hc = np.full(mi, np.nan)
    ha = np.full(mi, -2, dtype=np.int32)
    ht = np.zeros(mi, dtype=np.int32)
    hs = np.zeros((mi, N, n)) if history else None
    hu = np.zeros((mi, N - 1, m)) if history else None
    Kf, kf = np.zeros((N - 1, m, n)), np.zeros((N - 1, m))
    Kl, kl = np.zeros((N - 1, m, n)), np.zeros((N - 1, m))
    it = lib().ilqr_oracle_solve(C.byref(d), _dp(target), _dp(states), _dp(inputs), _dp(centers),
                                 _dp(radius), C.byref(mc), C.byref(cv), C.byref(c0), _dp(hc),
                                 _dp(ha), _dp(ht), _dp(hs), _dp(hu), _dp(Kf), _dp(kf), _dp(Kl),
                                 _dp(kl))
    out.update(iters=it, min_cost=mc.value, converge=bool(cv.value), cost0=c0.value,
               hist_cost=hc[:max(it, 0)], hist_alpha=ha[:max(it, 0)], hist_tries=ht[:max(it, 0)],
               K_first=Kf, k_first=kf, K_last=Kl, k_last=kl)
    if history:
        out.update(hist_states=hs[:max(it, 0)], hist_inputs=hu[:max(it, 0)])
    return out


def solve_batch(d, target, states, inputs, centers=None, radius=None, nthreads=1, history=True):
    """B problems (leading batch axis), in place on states/inputs."""
    B, mi = target.shape[0], d.max_iter
    if d.model == MODEL_CALLBACK:
        nthreads = 1  # the callbacks run Python: one thread
    mc = np.zeros(B)
    cv = np.zeros(B, dtype=np.int32)
    it = np.zeros(B, dtype=np.int32)
    c0 = np.zeros(B)
    hc = np.full((B, mi), np.nan) if history else None
    ha = np.full((B, mi), -2, dtype=np.int32) if history else None
    ht = np.zeros((B, mi), dtype=np.int32) if history else None
    lib().ilqr_oracle_solve_batch(C.byref(d), C.c_int32(B), _dp(target), _dp(states), _dp(inputs),
                                  _dp(centers), _dp(radius), _dp(mc), _dp(cv), _dp(it), _dp(c0),
                                  _dp(hc), _dp(ha), _dp(ht), C.c_int32(nthreads))
    return dict(min_cost=mc, converge=cv.astype(bool), iters=it, cost0=c0, hist_cost=hc,
                hist_alpha=ha, hist_tries=ht)


def barrier(d, x, c, r):
    g = np.zeros(d.n)
    H = np.zeros((d.n, d.n))
    v = lib().orc_barrier(C.byref(d), _dp(np.ascontiguousarray(x, dtype=np.float64)),
                          _dp(np.ascontiguousarray(c, dtype=np.float64)), C.c_double(r), _dp(g),
                          _dp(H))
    return v, g, H
