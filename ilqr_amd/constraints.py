"""Constraint objects with the interface of the reference ``constraints.py``.

``BubbleConstraint`` (``constraints.py:10-48``) is a keep-IN region: the car must
stay inside the disc ``h(x) = (px-cx)^2 + (py-cy)^2 - r^2 <= 0`` around a
per-time-step centre, and its speed inside ``vel_bounds``.  ``get_h`` / ``dh_dx``
are the functions the device barrier term is built from; the linearised-bound
assembly (``get_linear_constraint`` / ``get_bounds``) exists for SQP-style
callers and is plain host numpy.
"""
import numpy as np


class Constraint:
    def __init__(self, state_size, constraint_size):
        self.state_size = state_size
        self.constraint_size = constraint_size
        self.horizon = 0


class BubbleConstraint(Constraint):
    def __init__(self, horizon):
        super().__init__(4, 2)
        self.horizon = horizon

    def setup(self, centers, radius, vel_bounds):
        self.centers = centers
        self.radius = radius
        self.vel_bounds = vel_bounds

    def get_h(self, x, center, r):
        dx, dy = x[0] - center[0], x[1] - center[1]
        return dx ** 2 + dy ** 2 - r ** 2

    def dh_dx(self, x, center):
        out = np.zeros((2, 4))
        out[0, 0] = 2 * (x[0] - center[0])
        out[0, 1] = 2 * (x[1] - center[1])
        out[1, 2] = 1.0
        return out

    def get_uieq(self, state, center, r):
        g = self.dh_dx(state, center)
        return g[0, 0] * state[0] + g[0, 1] * state[1] - self.get_h(state, center, r)

    def get_linear_constraint(self, states):
        n, c = self.state_size, self.constraint_size
        Cu = np.zeros((self.horizon * c, self.horizon * n))
        for i in range(self.horizon):
            Cu[i * c:(i + 1) * c, i * n:(i + 1) * n] = self.dh_dx(states[i, :], self.centers[i, :])
        return Cu

    def get_bounds(self, states):
        xmin = np.empty(2 * self.horizon)
        xmax = np.empty(2 * self.horizon)
        for i in range(self.horizon):
            xmin[2 * i], xmin[2 * i + 1] = -np.inf, self.vel_bounds[0]
            xmax[2 * i] = self.get_uieq(states[i, :], self.centers[i, :], self.radius[i])
            xmax[2 * i + 1] = self.vel_bounds[1]
        return xmin, xmax
