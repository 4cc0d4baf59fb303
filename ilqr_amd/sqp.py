"""Sequential-QP trajectory optimiser with the interface of the reference ``sqp.py``, batched on
the GPU.

``sequential_QP_optimizer(sys, constraint, target_states, dt)`` has the constructor, attributes
(``maxIter``, ``eps``, ``cost_eps``, ``init_input_fixed``, ``states``, ``inputs``, ``min_cost``,
``converge``) and methods (``set_init_inputs``, ``__call__``) of ``sqp.py:7-66`` /
``:136-171``; ``target_states`` may carry a leading batch axis ``(B, N, 4)``,
with the constraint's ``centers (B, N, 2)`` and ``radius (B, N)`` batched alike.

What differs from the reference, on purpose: the QP of every iteration (``sqp.py:77-134``) is
solved on the device by an interior-point iteration to 1e-10 instead of by the OSQP package at its
default 1e-3, so iterates agree with the reference only to the accuracy OSQP delivers; the
constraint rows of step 0, which only test the given initial state, are left out; a QP without a
solution sets ``STATUS_QP_FAILED`` in ``status`` and ends that problem's iteration.  fp64 and the
state layouts that start with ``(px, py, v)`` only (``constraints.py:12`` is written for them).
``cost()`` and ``sim()`` (``sqp.py:44-61``) run on the device as one launch over the batch; ``plot``
is not provided.  The attributes ``x0``, ``umin`` and ``umax`` are honoured as in ``sqp.py:120``,
``:130-131``, ``:158``: the rollouts start from ``x0`` and the QP bounds the inputs by ``umin`` /
``umax`` (infinite entries = no row).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._solver import DeviceProblem, _torch, resolve_model


class sequential_QP_optimizer:
    def __init__(self, sys, constraint, target_states, dt, device=None, history=False, qp_tol=0.0, qp_max_iter=0):
        self.target_states = target_states
        ts = np.asarray(target_states)
        if ts.ndim not in (2, 3):
            raise ValueError("target_states must be (N, n) or (B, N, n)")
        self._batched = ts.ndim == 3
        self._B = ts.shape[0] if self._batched else 1
        self.horizon = ts.shape[-2]
        self.dt = dt
        self.converge = False
        self.system = sys
        self.constraint = constraint
        self.n_states = sys.state_size
        self.m_inputs = sys.control_size
        self.Q, self.R, self.Qf = sys.Q, sys.R, sys.Q_f
        self.maxIter = 20                     # sqp.py:29
        self.min_cost = 0.0
        self.eps = 1e-2                       # sqp.py:31
        self.cost_eps = 1e-6                  # sqp.py:32
        lead = (self._B,) if self._batched else ()
        self.states = np.zeros(lead + (self.horizon, self.n_states))
        self.inputs = np.zeros(lead + (self.horizon - 1, self.m_inputs))
        self.u0 = self.inputs[..., 0, :]
        self.init_input_fixed = False
        self.x0 = ts[..., 0, :]
        self.umin = -np.ones(self.m_inputs) * np.inf
        self.umax = np.ones(self.m_inputs) * np.inf
        if sys.control_limited:               # sqp.py:40-42
            self.umin, self.umax = sys.control_lower_limit, sys.control_upper_limit
        self.iterations = self.status = self.qp_iterations = None
        self.cost_history = self.alpha_history = None
        self._device, self._history = device, bool(history)
        self._qp_tol, self._qp_max_iter = float(qp_tol), int(qp_max_iter)
        self._dev = None

    # -- host-side helpers with the reference's names ---------------------------------------
    def set_init_inputs(self, init_inputs):   # sqp.py:63-65
        self.inputs = init_inputs
        self.u0 = np.array(np.asarray(init_inputs)[..., 0, :], dtype=float)

    def _lead(self, a, tail):
        a = np.asarray(a, dtype=float)
        return a if a.ndim == tail + 1 else a[None]

    def _opts(self):
        o = _lib.SqpOpts()
        o.max_iter, o.init_input_fixed = int(self.maxIter), int(bool(self.init_input_fixed))
        o.qp_max_iter, o.eps, o.cost_eps = self._qp_max_iter, float(self.eps), float(self.cost_eps)
        o.qp_tol, o.qp_sigma = self._qp_tol, 0.0
        o.x0 = self._x0.data_ptr() if self._dev is not None else None
        return o

    def upload(self):
        """Device buffers and the start of sqp.py:137-140: states = target_states, inputs as set."""
        torch = _torch()
        c = self.constraint
        sy = self.system
        umin = np.asarray(self.umin, dtype=float).reshape(self.m_inputs)
        umax = np.asarray(self.umax, dtype=float).reshape(self.m_inputs)
        bounded = bool(np.isfinite(umin).any() or np.isfinite(umax).any())
        model_id = resolve_model(sy, np.float64, _lib.CONSTR_BARRIER)
        key = (self._B, self.horizon, int(self.maxIter), float(sy.dt), np.asarray(sy.Q).tobytes(), np.asarray(sy.R).tobytes(),
               np.asarray(sy.Q_f).tobytes(), umin.tobytes(), umax.tobytes(), float(self.eps), model_id,
               tuple(float(v) for v in c.vel_bounds))
        if self._dev is not None and self._dev_key != key:   # the model setup changed since the last call
            self._dev.close()
            self._dev = None
        if self._dev is None:
            self._dev_key = key
            # the solver's own umin / umax bound the QP (sqp.py:130-131); the rollouts of the search
            # are not clamped (sqp.py:56-61)
            self._dev = DeviceProblem(self.system, self._B, self.horizon, np.float64, self._device, max_iter=self.maxIter,
                                      lm_parameter=0.0, alpha_floor=self.eps, constraint_mode=_lib.CONSTR_BARRIER,
                                      barrier_mu=0.0, barrier_delta=1.0, vel_bounds=tuple(c.vel_bounds),
                                      history=self._history, repack_every=-1, model_id=model_id,
                                      limits=(umin, umax, bounded))
            dev = self._dev
            with torch.cuda.device(dev.device):
                self._work = torch.zeros(int(dev.lib.ilqr_sqp_workspace_bytes(dev.handle)), dtype=torch.uint8, device=dev.device)
                self._u0 = torch.zeros(_lib.TILE * ((self._B + _lib.TILE - 1) // _lib.TILE) * self.m_inputs,
                                       dtype=torch.float64, device=dev.device)
                self._x0 = torch.zeros(_lib.TILE * ((self._B + _lib.TILE - 1) // _lib.TILE) * self.n_states,
                                       dtype=torch.float64, device=dev.device)
                self._qp_iters = torch.zeros(self._B, dtype=torch.int32, device=dev.device)
        dev = self._dev
        tg = self._lead(self.target_states, 2)
        centers = np.array(np.broadcast_to(self._lead(c.centers, 2), (self._B, self.horizon, 2)))
        radius = np.array(np.broadcast_to(self._lead(np.asarray(c.radius, dtype=float), 1), (self._B, self.horizon)))
        # sqp.py:138: states = target_states (the first linearisation point); every QP and every
        # rollout starts from x0 (sqp.py:120, :158)
        dev.upload(tg, tg, self._lead(self.inputs, 2), centers=centers, radius=radius)
        dev._upload(np.array(np.broadcast_to(self._lead(self.x0, 1), (self._B, self.n_states))), self._x0, self.n_states)
        dev._upload(np.array(np.broadcast_to(self._lead(self.u0, 1), (self._B, self.m_inputs))), self._u0, self.m_inputs)
        dev.status.zero_()
        return dev

    def solve_device(self):
        dev = self._dev
        o = self._opts()
        _lib.check(dev.lib.ilqr_sqp_solve(dev.handle, C.byref(o), self._work.data_ptr(), self._u0.data_ptr(),
                                          self._qp_iters.data_ptr(), dev.stream()), "ilqr_sqp_solve")

    def download(self):
        dev = self._dev
        st, ui = dev.download_states(), dev.download_inputs()
        J = dev.J.cpu().numpy()
        cv = dev.converged.cpu().numpy().astype(bool)
        self.iterations, self.status = dev.iters.cpu().numpy(), dev.status.cpu().numpy()
        self.qp_iterations = self._qp_iters.cpu().numpy()
        if self._history:
            self.cost_history = dev.hist_cost.cpu().numpy().T.copy()
            self.alpha_history = dev.hist_alpha.cpu().numpy().T.copy()
        if self._batched:
            self.states, self.inputs, self.min_cost, self.converge = st, ui, J, cv
        else:
            self.states, self.inputs, self.min_cost, self.converge = st[0], ui[0], float(J[0]), bool(cv[0])
            self.iterations, self.status, self.qp_iterations = int(self.iterations[0]), int(self.status[0]), int(self.qp_iterations[0])
            if self._history:
                self.cost_history, self.alpha_history = self.cost_history[0], self.alpha_history[0]

    def cost(self):
        """``sqp.py:44-54`` for the current ``states`` / ``inputs``: scalar, or ``(B,)``; one launch of
        the cost kernel over the batch."""
        dev = self.upload()
        dev._upload(self._lead(self.states, 2), dev.X, self.horizon * self.n_states)
        J = dev.plain_cost()     # the handle's barrier weight is zero: the plain tracking cost
        return J if self._batched else float(J[0])

    def sim(self, x0, inputs):
        """``sqp.py:56-61``: the states the inputs lead to from ``x0``, rolled out on the device (the
        line-search commit with a zero step and zero gains is exactly that rollout).  ``x0 (n,)`` or
        ``(B, n)``, ``inputs (N-1, m)`` or ``(B, N-1, m)``."""
        dev = self.upload()
        st = np.zeros((self._B, self.horizon, self.n_states))
        st[:, 0, :] = np.broadcast_to(self._lead(x0, 1), (self._B, self.n_states))
        out = dev.rollout(st, np.broadcast_to(self._lead(inputs, 2), (self._B, self.horizon - 1, self.m_inputs)))
        return out if self._batched else out[0]

    def __call__(self):                       # sqp.py:136-171
        self.upload()
        self.solve_device()
        self.download()
        return self.states
