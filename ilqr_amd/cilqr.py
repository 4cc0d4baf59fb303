"""Batched constrained iterative LQR on MI355X behind the interface of the
reference ``cilqr.py`` (class ``iterative_LQR``, ``cilqr.py:14-150``).

Differences from ``ilqr.iterative_LQR`` exactly as in the reference:
``control_limited`` (solver-side flag, default True, ``cilqr.py:36``) enables the
per-time-step box QP on the control step in the backward pass
(``cilqr.py:112-134``); ``maxIter`` 100 (``:37``); ``input_limit_lambda`` 1e-4
(``:40``); ``alpha_terminal`` 1e-2 (``:41``), i.e. step sizes 2^0 .. 2^-7.

The reference solves that QP with a fresh OSQP instance per step; here it is
solved exactly on the device (the result is consumed only when every bound
multiplier exceeds ``input_limit_lambda``, which happens only at a corner of
the box, so the corners' KKT conditions decide).

Extension without a reference counterpart: ``set_barrier(constraint, mu,
delta)`` adds a relaxed log-barrier on the keep-in disc and the speed box of a
``constraints.BubbleConstraint`` (``constraints.py:20-26``) to the cost and to
its quadratisation inside the fused backward kernel.
"""
import numpy as np

from . import _lib
from ._solver import BatchedSolverBase


class iterative_LQR(BatchedSolverBase):
    _default_max_iter = 100

    def __init__(self, sys, target_states, dt, **kw):
        super().__init__(sys, target_states, dt, **kw)
        self.control_limited = True
        self.input_limit_lambda = 1e-4
        self.alpha_terminal = 1e-2
        self._barrier = None

    def set_barrier(self, constraint, mu=1.0, delta=0.1):
        """Enable the relaxed log barrier -mu*[log(-h) + log(v - v_lo) + log(v_hi - v)]
        for a ``BubbleConstraint`` whose ``setup(centers, radius, vel_bounds)`` has been
        called with per-problem ``centers (B, N, 2)`` / ``radius (B, N)`` (or unbatched)."""
        self._barrier = (constraint, float(mu), float(delta))

    @property
    def _constraint_mode(self):
        mode = _lib.CONSTR_BOXQP if self.control_limited else _lib.CONSTR_NONE
        if self._barrier is not None:
            mode |= _lib.CONSTR_BARRIER
        return mode

    def _alpha_floor(self):
        return float(self.alpha_terminal)

    def _extra_desc(self):
        d = dict(input_limit_lambda=float(self.input_limit_lambda), solver_limits=bool(self.control_limited))
        if self._barrier is not None:
            c, mu, delta = self._barrier
            d.update(barrier_mu=mu, barrier_delta=delta,
                     vel_bounds=(float(c.vel_bounds[0]), float(c.vel_bounds[1])))
        return d

    def _extra_upload(self):
        if self._barrier is None:
            return {}
        c = self._barrier[0]
        centers = np.asarray(c.centers, dtype=np.float64)
        radius = np.asarray(c.radius, dtype=np.float64)
        if centers.ndim == 2:
            centers, radius = centers[None], radius[None]
        return dict(centers=centers, radius=radius)
