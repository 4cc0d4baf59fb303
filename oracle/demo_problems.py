"""Seeded restatements of the problem constructions of the reference demos (test infrastructure:
the golden fixtures and the parity tests are built on them; the product package does not import this).

``demo_car`` / ``demo_caracc`` / ``demo_dubins``: ``ilqr_demo.py:9-49`` Car, ``:89-135``
CarAcceleration, ``:170-192`` DubinsCar and ``cilqr.py:154-196``, with the module-level ``random``
stream of the demos replaced by an explicitly seeded one (same call order), so that they are
reproducible.  ``demo_sqp``: ``sqp_demo.py:12-53``.
"""
import random as _pyrandom

import numpy as np

from ilqr_amd.problems import sqp_arc_problem as demo_sqp  # noqa: F401  (sqp_demo.py:12-53, seeded)


def demo_car(horizon=120, seed=0, noise_xy=3.0, noise_th=0.5):
    """Problem of ``ilqr_demo.py:10-45`` (noise_xy=3) or ``cilqr.py:155-192``
    (noise_xy=15) with ``random.seed(seed)``.  Returns (noisy_targets,
    init_inputs, ref_vel)."""
    rnd = _pyrandom.Random(seed)
    target = np.zeros((horizon, 4))
    noisy = np.zeros((horizon, 4))
    ref_vel = np.zeros(horizon)
    dt, curv, a, v_max = 0.2, 0.1, 1.5, 11
    for i in range(40, horizon):
        if ref_vel[i - 1] > v_max:
            a = 0
        ref_vel[i] = ref_vel[i - 1] + a * dt
    for i in range(1, horizon):
        target[i, 0] = target[i - 1, 0] + np.cos(target[i - 1, 3]) * dt * ref_vel[i - 1]
        target[i, 1] = target[i - 1, 1] + np.sin(target[i - 1, 3]) * dt * ref_vel[i - 1]
        target[i, 2] = ref_vel[i]
        target[i, 3] = target[i - 1, 3] + curv * dt
        noisy[i, 0] = target[i, 0] + rnd.uniform(0, noise_xy)
        noisy[i, 1] = target[i, 1] + rnd.uniform(0, noise_xy)
        noisy[i, 2] = ref_vel[i]
        noisy[i, 3] = target[i, 3] + rnd.uniform(0, noise_th)
    init_inputs = np.zeros((horizon - 1, 2))
    for i in range(1, horizon):
        init_inputs[i - 1, 0] = (noisy[i, 2] - noisy[i - 1, 2]) / dt
        init_inputs[i - 1, 1] = (noisy[i, 3] - noisy[i - 1, 3]) / dt
    return noisy, init_inputs, ref_vel


def demo_caracc(horizon=100, seed=0):
    """Problem of ``ilqr_demo.py:89-131`` with ``random.seed(seed)``."""
    rnd = _pyrandom.Random(seed)
    target = np.zeros((horizon, 5))
    noisy = np.zeros((horizon, 5))
    ref_vel = np.zeros(horizon)
    ref_acc = np.zeros(horizon)
    dt, curv, a, v_max = 0.2, 0.1, 1.5, 11
    for i in range(40, horizon):
        if ref_vel[i - 1] > v_max:
            a = 0
        ref_acc[i] = a
        ref_vel[i] = ref_vel[i - 1] + a * dt
    for i in range(1, horizon):
        target[i, 0] = target[i - 1, 0] + np.cos(target[i - 1, 4]) * dt * ref_vel[i - 1]
        target[i, 1] = target[i - 1, 1] + np.sin(target[i - 1, 4]) * dt * ref_vel[i - 1]
        target[i, 2] = ref_vel[i]
        target[i, 3] = ref_acc[i]
        target[i, 4] = target[i - 1, 4] + curv * dt
        noisy[i, 0] = target[i, 0] + rnd.uniform(0, 1)
        noisy[i, 1] = target[i, 1] + rnd.uniform(0, 1)
        noisy[i, 2] = target[i, 2]
        noisy[i, 3] = target[i, 3]
        noisy[i, 4] = target[i, 4] + rnd.uniform(0, 0.1)
    init_inputs = np.zeros((horizon - 1, 2))
    for i in range(1, horizon):
        init_inputs[i - 1, 0] = (noisy[i, 3] - noisy[i - 1, 3]) / dt
        init_inputs[i - 1, 1] = (noisy[i, 4] - noisy[i - 1, 4]) / dt
    return noisy, init_inputs


def demo_dubins(horizon=200, seed=0):
    """Problem of ``ilqr_demo.py:170-185`` with ``random.seed(seed)``; the demo
    leaves the inputs at zero."""
    rnd = _pyrandom.Random(seed)
    target = np.zeros((horizon, 3))
    noisy = np.zeros((horizon, 3))
    dt, v, curv = 0.2, 1.0, 0.1
    for i in range(1, horizon):
        target[i, 0] = target[i - 1, 0] + np.cos(target[i - 1, 2]) * v * dt
        target[i, 1] = target[i - 1, 1] + np.sin(target[i - 1, 2]) * v * dt
        target[i, 2] = target[i - 1, 2] + v * curv * dt
        noisy[i, 0] = target[i, 0] + rnd.uniform(0, 1)
        noisy[i, 1] = target[i, 1] + rnd.uniform(0, 1)
        noisy[i, 2] = target[i, 2] + rnd.uniform(0, 1)
    return noisy, np.zeros((horizon - 1, 2))
