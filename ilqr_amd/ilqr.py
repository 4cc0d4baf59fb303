"""Batched iterative LQR with quadratic cost on MI355X, behind the interface of
the reference ``ilqr.py``.

    from ilqr_amd.systems import Car
    from ilqr_amd.ilqr import iterative_LQR

    optimizer = iterative_LQR(car_system, noisy_targets, dt)   # ilqr.py:17
    optimizer.inputs = init_inputs                             # ilqr_demo.py:49
    optimizer()                                                # ilqr.py:107
    optimizer.states, optimizer.inputs, optimizer.min_cost, optimizer.converge

``noisy_targets`` may be one problem ``(N, n)`` -- then every attribute has the
reference's shape -- or a batch ``(B, N, n)``; then ``states (B, N, n)``,
``inputs (B, N-1, m)``, ``min_cost (B,)``, ``converge (B,)`` bool, ``k (B, N-1,
m, 1)``, ``K (B, N-1, m, n)``.  Cost: x'Qx + u'Ru without a 1/2 factor
(``ilqr.py:36-46``).  The whole solve runs in HIP kernels; see ``_solver.py``.
"""
from ._solver import BatchedSolverBase


class iterative_LQR(BatchedSolverBase):
    """``ilqr.py:6-114``: ``maxIter`` 30, ``LM_parameter`` 0, line-search floor 1e-4
    (step sizes 2^0 .. 2^-14, all rolled out in parallel on the device)."""

    _default_max_iter = 30
