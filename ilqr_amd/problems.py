"""Seeded problem constructions for the batched iLQR path.

``car_batch`` / ``caracc_batch`` / ``dubins_batch`` / ``ltv_batch`` / ``sqp_batch`` ...: the batched
synthetic generators.  Problem ``b`` draws everything from
``numpy.random.Generator(PCG64(seed0 + b))`` so its data do not depend on the
batch size or on how the batch is sharded over GPUs.  (The seeded restatements of the reference
demos' own problem constructions, which the golden fixtures are made from, are test
infrastructure: ``oracle/demo_problems.py``.)

Everything here is host-side numpy in the reference's array layout
(``target_states (B, N, n)``, ``inputs (B, N-1, m)``, ``x0 (B, n)``).
"""
import random as _pyrandom

import numpy as np

CAR_Q = np.diag([50.0, 50.0, 1000.0, 0.0])
CAR_R = np.diag([3000.0, 1000.0])
CAR_LIMITS = ([-1.5, -0.3], [1.5, 0.3])
CARACC_Q = np.diag([50.0, 50.0, 1000.0, 1000.0, 0.0])
CARACC_R = np.diag([3000.0, 1000.0])
CARACC_LIMITS = ([-6.0, -0.2], [6.0, 0.2])
DUBINS_Q = 100.0 * np.diag([1.0, 1.0, 1.0])
DUBINS_R = np.diag([10.0, 100.0])
DUBINS_LIMITS = ([0.0, -0.2], [2.0, 0.2])


def _one_car(b, seed0, N, dt, noise_xy, noise_th, ramp_start):
    rng = np.random.Generator(np.random.PCG64(seed0 + b))
    curv = rng.uniform(-0.15, 0.15)
    a = rng.uniform(0.5, 2.0)
    v_max = rng.uniform(6.0, 12.0)
    x0 = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(0, 2),
                   rng.uniform(-np.pi, np.pi)])
    noise = rng.uniform(0.0, 1.0, size=(N - 1, 3))
    ref_vel = np.zeros(N)
    acc = a
    for i in range(ramp_start, N):
        if ref_vel[i - 1] > v_max:
            acc = 0.0
        ref_vel[i] = ref_vel[i - 1] + acc * dt
    # heading is an arithmetic progression; positions are its prefix sums
    theta = x0[3] + curv * dt * np.arange(N)
    target = np.zeros((N, 4))
    target[0, 0], target[0, 1] = x0[0], x0[1]
    target[1:, 0] = x0[0] + np.cumsum(np.cos(theta[:-1]) * dt * ref_vel[:-1])
    target[1:, 1] = x0[1] + np.cumsum(np.sin(theta[:-1]) * dt * ref_vel[:-1])
    target[:, 2] = ref_vel
    target[:, 3] = theta
    noisy = target.copy()
    noisy[1:, 0] += noise_xy * noise[:, 0]
    noisy[1:, 1] += noise_xy * noise[:, 1]
    noisy[1:, 3] += noise_th * noise[:, 2]
    init_inputs = np.empty((N - 1, 2))
    init_inputs[:, 0] = (noisy[1:, 2] - noisy[:-1, 2]) / dt
    init_inputs[:, 1] = (noisy[1:, 3] - noisy[:-1, 3]) / dt
    return noisy, init_inputs, x0


def car_batch(B, N=200, seed0=1234, first=0, dt=0.2, noise_xy=3.0, noise_th=0.5,
              ramp_start=40):
    """Batched synthetic Car problems ``first .. first+B-1``.

    Per problem: curvature U(-0.15,0.15), acceleration U(0.5,2), speed cap
    U(6,12), initial state (U(-5,5), U(-5,5), U(0,2), U(-pi,pi)); the
    reference arc of ``ilqr_demo.py:25-35`` starts at the initial pose; noise
    and finite-difference initial inputs as ``ilqr_demo.py:36-45``.

    ``init_states`` is the reference's all-zero initial state guess
    (``ilqr.py:31-32``) expressed in the problem's own frame: the initial pose
    (px0, py0, theta0) at every step with zero speed, and ``x0`` itself at
    step 0.  It is NOT a rollout of ``init_inputs`` (the reference never
    rolls its initial guess out either).

    Returns dict(target_states (B,N,4), init_inputs (B,N-1,2), x0 (B,4),
    init_states (B,N,4)).
    """
    tg = np.empty((B, N, 4))
    ui = np.empty((B, N - 1, 2))
    x0 = np.empty((B, 4))
    for j in range(B):
        tg[j], ui[j], x0[j] = _one_car(first + j, seed0, N, dt, noise_xy, noise_th,
                                       ramp_start)
    st = np.repeat(x0[:, None, :], N, axis=1)
    st[:, 1:, 2] = 0.0
    return {"target_states": tg, "init_inputs": ui, "x0": x0, "init_states": st}


def caracc_batch(B, N=100, seed0=4321, first=0, dt=0.2):
    """Batched CarAcceleration problems in the style of ``ilqr_demo.py:89-131``
    with per-problem curvature/acceleration/speed cap and initial state."""
    tg = np.zeros((B, N, 5))
    ui = np.zeros((B, N - 1, 2))
    x0 = np.zeros((B, 5))
    for j in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + first + j))
        curv = rng.uniform(-0.15, 0.15)
        a = rng.uniform(0.5, 2.0)
        v_max = rng.uniform(6.0, 12.0)
        x0[j] = [rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(0, 2), 0.0,
                 rng.uniform(-np.pi, np.pi)]
        noise = rng.uniform(0.0, 1.0, size=(N - 1, 3))
        ref_vel = np.zeros(N)
        ref_acc = np.zeros(N)
        acc = a
        for i in range(40, N):
            if ref_vel[i - 1] > v_max:
                acc = 0.0
            ref_acc[i] = acc
            ref_vel[i] = ref_vel[i - 1] + acc * dt
        theta = x0[j, 4] + curv * dt * np.arange(N)
        t = tg[j]
        t[0, 0], t[0, 1] = x0[j, 0], x0[j, 1]
        t[1:, 0] = x0[j, 0] + np.cumsum(np.cos(theta[:-1]) * dt * ref_vel[:-1])
        t[1:, 1] = x0[j, 1] + np.cumsum(np.sin(theta[:-1]) * dt * ref_vel[:-1])
        t[:, 2] = ref_vel
        t[:, 3] = ref_acc
        t[:, 4] = theta
        t[1:, 0] += noise[:, 0]
        t[1:, 1] += noise[:, 1]
        t[1:, 4] += 0.1 * noise[:, 2]
        ui[j, :, 0] = (t[1:, 3] - t[:-1, 3]) / dt
        ui[j, :, 1] = (t[1:, 4] - t[:-1, 4]) / dt
    st = np.repeat(x0[:, None, :], N, axis=1)
    st[:, 1:, 2] = 0.0
    return {"target_states": tg, "init_inputs": ui, "x0": x0, "init_states": st}


def dubins_batch(B, N=200, seed0=9876, first=0, dt=0.2):
    """Batched DubinsCar problems in the style of ``ilqr_demo.py:170-185``."""
    tg = np.zeros((B, N, 3))
    ui = np.zeros((B, N - 1, 2))
    x0 = np.zeros((B, 3))
    for j in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + first + j))
        curv = rng.uniform(-0.15, 0.15)
        v = rng.uniform(0.5, 1.5)
        x0[j] = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-np.pi, np.pi)]
        noise = rng.uniform(0.0, 1.0, size=(N - 1, 3))
        theta = x0[j, 2] + v * curv * dt * np.arange(N)
        t = tg[j]
        t[0, 0], t[0, 1] = x0[j, 0], x0[j, 1]
        t[1:, 0] = x0[j, 0] + np.cumsum(np.cos(theta[:-1]) * v * dt)
        t[1:, 1] = x0[j, 1] + np.cumsum(np.sin(theta[:-1]) * v * dt)
        t[:, 2] = theta
        t[1:] += noise
    st = np.repeat(x0[:, None, :], N, axis=1)
    return {"target_states": tg, "init_inputs": ui, "x0": x0, "init_states": st}


def bubbles_for(target_states, radius=4.0, vel_bounds=(-1.0, 14.0)):
    """Keep-in discs centred on the target positions, as ``sqp_demo.py:40-53``
    populates ``BubbleConstraint.setup(centers, radius, vel_bounds)``.

    Returns (centers (B,N,2), radius (B,N), vel_bounds (2,))."""
    centers = np.ascontiguousarray(target_states[..., :2])
    rad = np.full(target_states.shape[:-1], float(radius))
    return centers, rad, np.asarray(vel_bounds, dtype=float)


def ltv_system(N=500, n=12, m=4, seed=2024, dt=0.05):
    """Synthetic quadrotor-sized linear time-varying system: A_t = I + dt*M_t with a slowly
    rotating, mildly damped M_t, and B_t = dt*G_t.  Returns (A (N-1,n,n), B (N-1,n,m), Q, R).
    No reference counterpart (the reference's model_f has no time argument)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    M0 = rng.normal(size=(n, n)) * 0.4
    M0 = 0.5 * (M0 - M0.T) - 0.15 * np.eye(n)          # skew part + damping: stable
    M1 = rng.normal(size=(n, n)) * 0.1
    G0 = rng.normal(size=(n, m))
    G1 = rng.normal(size=(n, m)) * 0.3
    A = np.empty((N - 1, n, n))
    Bm = np.empty((N - 1, n, m))
    for t in range(N - 1):
        ph = 2 * np.pi * t / (N - 1)
        A[t] = np.eye(n) + dt * (M0 + np.sin(ph) * M1)
        Bm[t] = dt * (G0 + np.cos(ph) * G1)
    Q = np.diag(rng.uniform(1.0, 20.0, size=n))
    R = np.diag(rng.uniform(0.5, 5.0, size=m))
    return A, Bm, Q, R


def ltv_batch(B, N=500, n=12, m=4, seed0=777, first=0):
    """Per-problem smooth random targets, initial states and (zero) initial inputs for the LTV
    system; problem b depends only on seed0 + b."""
    tg = np.empty((B, N, n))
    x0 = np.empty((B, n))
    tt = np.linspace(0.0, 1.0, N)[:, None]
    for j in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + first + j))
        amp = rng.uniform(-2.0, 2.0, size=(3, n))
        tg[j] = amp[0] * np.sin(2 * np.pi * tt) + amp[1] * np.cos(np.pi * tt) + amp[2] * tt
        x0[j] = rng.uniform(-1.0, 1.0, size=n)
    st = np.zeros((B, N, n))
    st[:, 0] = x0
    return {"target_states": tg, "init_inputs": np.zeros((B, N - 1, m)), "x0": x0, "init_states": st}


# ---- problems for the models of ilqr_amd/user_models.py (no reference demo exists for them;
# the constructions follow the pattern of ilqr_demo.py:10-45: a smooth reference motion, noise
# on the targets, finite differences of the noisy targets as the initial inputs) ----------
# every state component carries weight: with an unweighted steering angle the Riccati recursion of
# this model amplifies last-bit differences by many orders of magnitude per iteration, which makes
# a problem useless as a parity fixture
BICYCLE_Q = np.diag([10.0, 10.0, 100.0, 100.0, 100.0])
BICYCLE_R = np.diag([100.0, 100.0])
BICYCLE_LIMITS = ([-3.0, -0.5], [3.0, 0.5])
PENDULUM_Q = np.diag([100.0, 1.0])
PENDULUM_R = np.diag([0.1])
PENDULUM_LIMITS = ([-20.0], [20.0])


def bicycle_batch(B, N=150, seed0=4321, first=0, dt=0.1, wheelbase=2.5):
    """``B`` kinematic-bicycle tracking problems; problem ``b`` depends on ``seed0 + first + b`` only."""
    tg = np.zeros((B, N, 5))
    ui = np.zeros((B, N - 1, 2))
    x0 = np.zeros((B, 5))
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + first + b))
        v = rng.uniform(3.0, 8.0) + rng.uniform(0.2, 1.0) * dt * np.arange(N)
        delta = rng.uniform(-0.2, 0.2) + 0.1 * np.sin(rng.uniform(0.5, 1.5) * dt * np.arange(N))
        theta = rng.uniform(-np.pi, np.pi) + np.concatenate([[0.0], np.cumsum(v[:-1] * np.tan(delta[:-1]) / wheelbase * dt)])
        px = rng.uniform(-5, 5) + np.concatenate([[0.0], np.cumsum(v[:-1] * np.cos(theta[:-1]) * dt)])
        py = rng.uniform(-5, 5) + np.concatenate([[0.0], np.cumsum(v[:-1] * np.sin(theta[:-1]) * dt)])
        ref = np.stack([px, py, theta, v, delta], axis=-1)
        noisy = ref.copy()
        noisy[1:, 0:2] += rng.uniform(0.0, 1.0, size=(N - 1, 2))
        noisy[1:, 2] += rng.uniform(0.0, 0.1, size=N - 1)
        tg[b] = noisy
        x0[b] = ref[0]
        ui[b, :, 0] = np.diff(noisy[:, 3]) / dt
        ui[b, :, 1] = np.diff(noisy[:, 4]) / dt
    init_states = np.zeros_like(tg)
    init_states[:, 0] = x0
    return dict(target_states=tg, init_inputs=ui, x0=x0, init_states=init_states)


def pendulum_batch(B, N=120, seed0=987, first=0):
    """``B`` pendulum swing problems: reach the upright position from a random hanging state."""
    tg = np.zeros((B, N, 2))
    x0 = np.zeros((B, 2))
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + first + b))
        x0[b] = [rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)]
        ramp = np.minimum(1.0, np.arange(N) / (0.6 * N))
        tg[b, :, 0] = x0[b, 0] + (np.pi - x0[b, 0]) * ramp
        tg[b, 1:, 1] = np.diff(tg[b, :, 0]) / 0.05
    ui = np.zeros((B, N - 1, 1))
    init_states = np.zeros_like(tg)
    init_states[:, 0] = x0
    return dict(target_states=tg, init_inputs=ui, x0=x0, init_states=init_states)


# ---- problems of the sequential-QP demo (sqp_demo.py:12-53) ---------------------------------
SQP_Q = np.diag([50.0, 50.0, 10.0, 1.0])
SQP_R = np.diag([300.0, 1000.0])
SQP_LIMITS = ([-1.0, -0.3], [1.0, 0.3])


def sqp_arc_problem(horizon=80, seed=0, noisy=0.5, curv=0.1, radius=0.6):
    """One problem shaped like ``sqp_demo.py:12-53`` (``random_example``), seeded: unit-speed arc,
    noisy targets, bubbles of the given radius centred on the noisy targets, speed bounds [0, 2].
    Returns (noisy_targets, init_inputs, centers, radius, vel_bounds)."""
    rnd = _pyrandom.Random(seed)
    target = np.zeros((horizon, 4))
    noisy_t = np.zeros((horizon, 4))
    noisy_t[0, 2] = target[0, 2] = 1
    centers = np.zeros((horizon, 2))
    ref_vel = np.ones(horizon)
    dt = 0.2
    for i in range(1, horizon):
        target[i, 0] = target[i - 1, 0] + np.cos(target[i - 1, 3]) * dt * ref_vel[i - 1]
        target[i, 1] = target[i - 1, 1] + np.sin(target[i - 1, 3]) * dt * ref_vel[i - 1]
        target[i, 2] = ref_vel[i]
        target[i, 3] = target[i - 1, 3] + curv * dt
        noisy_t[i, 0] = target[i, 0] + rnd.uniform(0, noisy)
        noisy_t[i, 1] = target[i, 1] + rnd.uniform(0, noisy)
        noisy_t[i, 2] = ref_vel[i]
        noisy_t[i, 3] = target[i, 3] + rnd.uniform(0, 0.1)
        centers[i, 0], centers[i, 1] = noisy_t[i, 0], noisy_t[i, 1]
    init_inputs = np.zeros((horizon - 1, 2))
    for i in range(1, horizon):
        init_inputs[i - 1, 0] = (noisy_t[i, 2] - noisy_t[i - 1, 2]) / dt
        init_inputs[i - 1, 1] = (noisy_t[i, 3] - noisy_t[i - 1, 3]) / dt
    return noisy_t, init_inputs, centers, np.full(horizon, float(radius)), (0.0, 2.0)


def sqp_batch(B, N=80, seed0=2468, first=0):
    """``B`` problems of the ``sqp_demo.py`` kind; problem ``b`` depends on ``seed0 + first + b`` only
    (own curvature, noise level and bubble radius)."""
    tg = np.zeros((B, N, 4))
    ui = np.zeros((B, N - 1, 2))
    centers = np.zeros((B, N, 2))
    radius = np.zeros((B, N))
    for b in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + first + b))
        t, u, c, r, vb = sqp_arc_problem(N, seed=int(rng.integers(1 << 30)), noisy=rng.uniform(0.2, 0.6),
                                  curv=rng.uniform(-0.2, 0.2), radius=rng.uniform(0.5, 0.9))
        tg[b], ui[b], centers[b], radius[b] = t, u, c, r
    return dict(target_states=tg, init_inputs=ui, centers=centers, radius=radius, vel_bounds=(0.0, 2.0))
