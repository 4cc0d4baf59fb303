"""CPU restatement of the reference's sequential-QP solver (``sqp.py``), numpy, one problem at
a time.

TEST INFRASTRUCTURE ONLY: imported by ``tests/`` as the checker of the batched GPU solver in
``ilqr_amd/sqp.py``; the shipped package never imports it.

PARITY UNPINNED.  The reference solves each QP subproblem with the third-party package
``osqp`` (``sqp.py:146-148``; un-vendored, version not pinned, absent from this image) at its
default tolerance of 1e-3, so its iterates cannot be reproduced here and no fixture of the
reference's output exists.  What IS restated line by line is everything around that call:

    sqp.py:44-54    cost                      -> cost()
    sqp.py:56-61    sim                       -> sim()
    sqp.py:77-85    compute_P_q               -> qp_data()  (stage weights, linear term)
    sqp.py:87-134   compute_A_l_u             -> qp_data()  (linearised dynamics, x0 / u0 rows,
                                                 constraints.py:32-48 rows and bounds, input bounds)
    sqp.py:136-171  __call__                  -> sqp_solve()

and the QP itself is solved to 1e-10 in two independent ways:

    solve_qp()          dynamics eliminated (states as an affine function of the inputs), the dense
                        inequality-constrained QP in the inputs solved by a Mehrotra
                        predictor-corrector interior-point iteration with dense factorisations;
    solve_qp_riccati()  the statement of the GPU kernel's method in numpy: a primal-dual
                        path-following iteration in (inputs, slacks, multipliers) whose Newton
                        step is an LQR problem solved by a Riccati recursion, two sweeps over
                        the horizon per iteration.
"""
import numpy as np


def cost(system, target, states, inputs):
    """sqp.py:44-54 (same as ilqr.py:36-46: no factor one half)."""
    d = states - target
    c = 0.0
    for i in range(states.shape[0] - 1):
        c += d[i] @ system.Q @ d[i] + inputs[i] @ system.R @ inputs[i]
    return c + d[-1] @ system.Q_f @ d[-1]


def sim(system, x0, inputs):
    """sqp.py:56-61."""
    N = inputs.shape[0] + 1
    states = np.zeros((N, system.state_size))
    states[0] = x0
    for i in range(N - 1):
        states[i + 1] = system.model_f(states[i], inputs[i])
    return states


def qp_data(system, constraint, target, states, inputs):
    """The pieces of the QP of sqp.py:77-134 in per-step form: x+ = A_i x + B_i u + c_i
    (rows ``A_i x_i - x_{i+1} + B_i u_i = d_i`` of sqp.py:123-127 with c_i = -d_i), the rows
    ``G_i x_i <= hi_i`` / ``lo_i <= G_i x_i`` of constraints.py:32-48."""
    N, n, m = states.shape[0], system.state_size, system.control_size
    A = np.zeros((N - 1, n, n))
    B = np.zeros((N - 1, n, m))
    c = np.zeros((N - 1, n))
    for i in range(N - 1):
        A[i] = system.compute_df_dx(states[i], inputs[i])
        B[i] = system.compute_df_du(states[i], inputs[i])
        c[i] = states[i + 1] - A[i] @ states[i] - B[i] @ inputs[i]
    G = np.zeros((N, 2, n))
    lo = np.zeros((N, 2))
    hi = np.zeros((N, 2))
    for i in range(N):
        G[i] = constraint.dh_dx(states[i], constraint.centers[i])
        lo[i] = [-np.inf, constraint.vel_bounds[0]]
        hi[i] = [constraint.get_uieq(states[i], constraint.centers[i], constraint.radius[i]), constraint.vel_bounds[1]]
    return A, B, c, G, lo, hi


def solve_qp(system, target, x0, A, B, c, G, lo, hi, umin, umax, u0=None, skip_first_state_rows=True):
    """Minimise sum 1/2 (x_i - t_i)' Q (x_i - t_i) + 1/2 u_i' R u_i (terminal weight Q_f) subject
    to the linearised dynamics from x0, lo_i <= G_i x_i <= hi_i and umin <= u_i <= umax
    (sqp.py:77-134 with P = blkdiag(Q.., Q_f, R..), q = -Q t).  Returns (inputs, states).
    The state rows of step 0 only test the given x0 and are left out."""
    N, n, m = target.shape[0], system.state_size, system.control_size
    T = N - 1
    # x_i = Phi_i u + phi_i
    Phi = np.zeros((N, n, T * m))
    phi = np.zeros((N, n))
    phi[0] = x0
    for i in range(T):
        Phi[i + 1] = A[i] @ Phi[i]
        Phi[i + 1][:, i * m:(i + 1) * m] += B[i]
        phi[i + 1] = A[i] @ phi[i] + c[i]
    H = np.zeros((T * m, T * m))
    g = np.zeros(T * m)
    for i in range(N):
        W = system.Q_f if i == N - 1 else system.Q
        H += Phi[i].T @ W @ Phi[i]
        g += Phi[i].T @ W @ (phi[i] - target[i])
    for i in range(T):
        H[i * m:(i + 1) * m, i * m:(i + 1) * m] += system.R
    rows, rhs = [], []
    for i in range(1 if skip_first_state_rows else 0, N):
        for r in range(2):
            if np.isfinite(hi[i, r]):
                rows.append(G[i, r] @ Phi[i])
                rhs.append(hi[i, r] - G[i, r] @ phi[i])
            if np.isfinite(lo[i, r]):
                rows.append(-(G[i, r] @ Phi[i]))
                rhs.append(-(lo[i, r] - G[i, r] @ phi[i]))
    E = np.eye(T * m)
    for i in range(T):
        for j in range(m):
            k = i * m + j
            lo_k, hi_k = (u0[j], u0[j]) if (u0 is not None and i == 0) else (umin[j], umax[j])
            if u0 is not None and i == 0:
                continue  # handled by elimination below
            if np.isfinite(hi_k):
                rows.append(E[k])
                rhs.append(hi_k)
            if np.isfinite(lo_k):
                rows.append(-E[k])
                rhs.append(-lo_k)
    Gm, h = np.array(rows), np.array(rhs)
    if u0 is not None:   # sqp.py:111-112, :121-122: the first input is fixed
        free = np.arange(m, T * m)
        g = g[free] + H[np.ix_(free, np.arange(m))] @ u0
        h = h - Gm[:, :m] @ u0
        H, Gm = H[np.ix_(free, free)], Gm[:, free]
    u = _interior_point(H, g, Gm, h)
    if u0 is not None:
        u = np.concatenate([u0, u])
    xs = np.array([Phi[i] @ u + phi[i] for i in range(N)])
    return u.reshape(T, m), xs


def _interior_point(H, g, G, h, tol=1e-10, max_iter=80):
    """min 1/2 u'Hu + g'u s.t. Gu <= h: primal-dual path following (Mehrotra predictor-corrector)."""
    nv, nc = H.shape[0], G.shape[0]
    u = np.zeros(nv)
    s = np.maximum(h - G @ u, 1.0)
    lam = np.ones(nc)
    for _ in range(max_iter):
        rd = H @ u + g + G.T @ lam
        rp = G @ u + s - h
        mu = s @ lam / nc
        if max(np.abs(rd).max(), np.abs(rp).max()) <= tol * (1 + np.abs(g).max()) and mu <= tol:
            break
        D = lam / s
        M = H + G.T @ (D[:, None] * G)
        L = np.linalg.cholesky(M + 1e-14 * np.eye(nv) * np.trace(M) / nv)

        def solve(rc):
            # G du + ds = -rp ; S dlam + Lam ds = -rc ; H du + G' dlam = -rd
            rhs = -rd - G.T @ ((-rc + lam * rp) / s)
            du = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            ds = -rp - G @ du
            dl = (-rc - lam * ds) / s
            return du, ds, dl

        def step_len(v, dv):
            neg = dv < 0
            return min(1.0, (-v[neg] / dv[neg]).min()) if neg.any() else 1.0

        du, ds, dl = solve(s * lam)
        a = min(step_len(s, ds), step_len(lam, dl))
        mu_aff = (s + a * ds) @ (lam + a * dl) / nc
        sig = (mu_aff / mu) ** 3
        du, ds, dl = solve(s * lam + ds * dl - sig * mu)
        a = 0.995 * min(step_len(s, ds), step_len(lam, dl))
        u, s, lam = u + a * du, s + a * ds, lam + a * dl
    return u


def sqp_solve(system, constraint, target, init_inputs, max_iter=20, eps=1e-2, cost_eps=1e-6, init_input_fixed=False,
              history=True, qp="dense", x0=None, umin=None, umax=None):
    """sqp.py:136-171 for one problem.  Returns the final states/inputs and per-iteration records.
    ``x0`` / ``umin`` / ``umax``: the attributes a caller may assign after construction (sqp.py:37-42)."""
    target = np.asarray(target, dtype=float)
    inputs = np.array(init_inputs, dtype=float)
    m = system.control_size
    x0 = target[0].copy() if x0 is None else np.array(x0, dtype=float)   # sqp.py:37
    u0 = inputs[0].copy()                                   # sqp.py:35, :65
    lim_lo, lim_hi = np.full(m, -np.inf), np.full(m, np.inf)
    if system.control_limited:                              # sqp.py:40-42
        lim_lo = np.asarray(system.control_lower_limit, dtype=float)
        lim_hi = np.asarray(system.control_upper_limit, dtype=float)
    umin = lim_lo if umin is None else np.asarray(umin, dtype=float)
    umax = lim_hi if umax is None else np.asarray(umax, dtype=float)
    states = target.copy()                                  # sqp.py:138
    min_cost, converge = np.inf, False
    rec = dict(hist_cost=[], hist_alpha=[], hist_qp_inputs=[], hist_states=[], hist_inputs=[])
    iters = 0
    for _ in range(max_iter):
        if converge:
            break
        A, B, c, G, lo, hi = qp_data(system, constraint, target, states, inputs)
        if qp == "dense":
            solved, _ = solve_qp(system, target, x0, A, B, c, G, lo, hi, umin, umax, u0 if init_input_fixed else None)
        else:
            solved, _, _ = solve_qp_riccati(system, target, x0, A, B, c, G, lo, hi, umin, umax, inputs,
                                            u0 if init_input_fixed else None)
        prev_states, prev_inputs = states.copy(), inputs.copy()
        d_u = solved - inputs
        alpha, idx, accepted = 1.0, 0, -1
        while True:
            inputs = prev_inputs + alpha * d_u
            states = sim(system, x0, inputs)
            cst = cost(system, target, states, inputs)
            if cst < min_cost:
                min_cost, accepted = cst, idx
                break
            states, inputs = prev_states.copy(), prev_inputs.copy()
            if alpha < eps or abs(1 - cst / min_cost) < cost_eps:
                converge = True
                break
            alpha /= 2.0
            idx += 1
        iters += 1
        rec["hist_cost"].append(min_cost)
        rec["hist_alpha"].append(accepted)
        if history:
            rec["hist_qp_inputs"].append(solved)
            rec["hist_states"].append(states.copy())
            rec["hist_inputs"].append(inputs.copy())
    out = dict(states=states, inputs=inputs, min_cost=min_cost, converge=converge, iters=iters)
    out.update({k: np.array(v) for k, v in rec.items()})
    return out


def solve_qp_riccati(system, target, x0, A, B, c, G, lo, hi, umin, umax, u_start, u0=None, sigma=0.1, tol=1e-10,
                     max_iter=100, info=None):
    """Same QP as solve_qp(), by the method of the GPU kernel (sqp_qp_kernel): the rows are
    r_j' z <= h_j with slacks s_j > 0 and multipliers lam_j > 0; per iteration
      backward sweep: Riccati recursion of  min 1/2 dz' (H + R' diag(lam/s) R) dz + g~' dz  over the
                      linearised dynamics dx+ = A dx + B du, dx_0 = 0, with
                      g~ = H z + g + R' (lam + (lam rp - rc) / s),  rp = R z + s - h,  rc = s lam - sigma mu;
      forward sweep 1: dz by rollout, ds = -rp - R dz, dlam = (-rc - lam ds) / s, largest steps that keep
                      s and lam positive;
      then the step is applied elementwise; mu and |rp| of the new point follow from sums taken
                      in the forward sweep.
    Rows of a step, in this order: bubble upper, speed lower, speed upper (steps 1..N-1), then per
    input lower, upper (steps 0..N-2).  Infinite bounds give rows that are skipped."""
    N, n, m = target.shape[0], system.state_size, system.control_size
    T = N - 1
    Q, R, Qf = np.asarray(system.Q, float), np.asarray(system.R, float), np.asarray(system.Q_f, float)
    ulo = np.tile(np.asarray(umin, float), (T, 1))
    uhi = np.tile(np.asarray(umax, float), (T, 1))
    if u0 is not None:
        ulo[0] = uhi[0] = u0
    NR = 3 + 2 * m

    def rows_of(i, x, u):
        """(value, bound, gradient wrt x, gradient wrt u) of every row of step i as 'value <= bound'."""
        out = []
        if i >= 1:
            out.append((G[i, 0] @ x, hi[i, 0], G[i, 0], None))
            out.append((-(G[i, 1] @ x), -lo[i, 1], -G[i, 1], None))
            out.append((G[i, 1] @ x, hi[i, 1], G[i, 1], None))
        else:
            out += [(0.0, np.inf, None, None)] * 3
        for j in range(m):
            e = np.zeros(m)
            e[j] = 1.0
            if i < T:
                out.append((-u[j], -ulo[i, j], None, -e))
                out.append((u[j], uhi[i, j], None, e))
            else:
                out += [(0.0, np.inf, None, None)] * 2
        return out

    def rollout(u):
        xs = np.zeros((N, n))
        xs[0] = x0
        for i in range(T):
            xs[i + 1] = A[i] @ xs[i] + B[i] @ u[i] + c[i]
        return xs

    u = np.array(u_start, float)
    if u0 is not None:
        u[0] = u0
    xs = rollout(u)
    s = np.ones((N, NR))
    lam = np.ones((N, NR))
    live = np.zeros((N, NR), dtype=bool)
    for i in range(N):
        for j, (val, bnd, gx, gu) in enumerate(rows_of(i, xs[i], u[i] if i < T else None)):
            live[i, j] = np.isfinite(bnd) and not (u0 is not None and i == 0 and j >= 3)
            if live[i, j]:
                s[i, j] = max(bnd - val, 1.0)
    nlive = live.sum()
    mu = (s * lam)[live].sum() / nlive
    K = np.zeros((T, m, n))
    kf = np.zeros((T, m))
    for it in range(max_iter):
        # backward sweep
        P = np.zeros((n, n))
        p = np.zeros(n)
        for i in range(N - 1, -1, -1):
            W = Qf if i == T else Q
            Hx = W.copy()
            gx_ = W @ (xs[i] - target[i])
            Hu = R.copy()
            gu_ = R @ u[i] if i < T else None
            for j, (val, bnd, gx, gu) in enumerate(rows_of(i, xs[i], u[i] if i < T else None)):
                if not live[i, j]:
                    continue
                rp = val + s[i, j] - bnd
                rc = s[i, j] * lam[i, j] - sigma * mu
                coef = lam[i, j] + (lam[i, j] * rp - rc) / s[i, j]
                d = lam[i, j] / s[i, j]
                if gx is not None:
                    Hx += d * np.outer(gx, gx)
                    gx_ = gx_ + coef * gx
                else:
                    Hu += d * np.outer(gu, gu)
                    gu_ = gu_ + coef * gu
            if i == T:
                P, p = Hx, gx_
                continue
            if u0 is not None and i == 0:
                K[i], kf[i] = 0.0, 0.0            # the first input is fixed: du_0 = 0
                P, p = Hx + A[i].T @ P @ A[i], gx_ + A[i].T @ p
                continue
            Quu = Hu + B[i].T @ P @ B[i]
            Qux = B[i].T @ P @ A[i]
            Qu = gu_ + B[i].T @ p
            Qxx = Hx + A[i].T @ P @ A[i]
            Qx = gx_ + A[i].T @ p
            K[i] = -np.linalg.solve(Quu, Qux)
            kf[i] = -np.linalg.solve(Quu, Qu)
            P = Qxx + K[i].T @ Qux
            p = Qx + K[i].T @ Qu
        # forward sweep 1: direction and step length
        dx = np.zeros(n)
        du = np.zeros((T, m))
        a_max = d_max = 1.0
        for i in range(N):
            if i < T:
                du[i] = K[i] @ dx + kf[i]
            for j, (val, bnd, gx, gu) in enumerate(rows_of(i, xs[i], u[i] if i < T else None)):
                if not live[i, j]:
                    continue
                rp = val + s[i, j] - bnd
                rc = s[i, j] * lam[i, j] - sigma * mu
                ds = -rp - (gx @ dx if gx is not None else gu @ du[i])
                dl = (-rc - lam[i, j] * ds) / s[i, j]
                if ds < 0:
                    a_max = min(a_max, -s[i, j] / ds)
                if dl < 0:
                    d_max = min(d_max, -lam[i, j] / dl)
            if i < T:
                dx = A[i] @ dx + B[i] @ du[i]
        # separate step lengths: `a` for inputs, states and slacks, `ad` for the multipliers
        a = min(1.0, 0.995 * a_max)
        ad = min(1.0, 0.995 * d_max)
        # the step is applied elementwise (the kernel does it at the start of its next backward sweep);
        # complementarity and row residual of the new point follow from four sums
        g0 = g1 = g2 = g3 = 0.0
        rp_abs = 0.0
        dx = np.zeros(n)
        dxs = np.zeros((N, n))
        dss = np.zeros((N, NR))
        dls = np.zeros((N, NR))
        for i in range(N):
            dxs[i] = dx
            for j, (val, bnd, gx, gu) in enumerate(rows_of(i, xs[i], u[i] if i < T else None)):
                if not live[i, j]:
                    continue
                rp = val + s[i, j] - bnd
                rc = s[i, j] * lam[i, j] - sigma * mu
                ds = -rp - (gx @ dx if gx is not None else gu @ du[i])
                dl = (-rc - lam[i, j] * ds) / s[i, j]
                dss[i, j], dls[i, j] = ds, dl
                g0 += s[i, j] * lam[i, j]
                g1 += ds * lam[i, j]
                g2 += s[i, j] * dl
                g3 += ds * dl
                rp_abs = max(rp_abs, abs(rp))
            if i < T:
                dx = A[i] @ dx + B[i] @ du[i]
        s = s + a * dss
        lam = lam + ad * dls
        xs = xs + a * dxs
        u = u + a * du
        gap = g0 + a * g1 + ad * g2 + (a * ad) * g3
        rp_max = rp_abs * (1.0 - a)
        mu = gap / nlive
        step = np.abs(du).max() * a
        if info is not None:
            info.append((it, a, mu, rp_max, step))
        if not np.isfinite(mu) or min(a, ad) < 1e-8:   # no solution: the step length collapses when the rows are infeasible
            break
        if mu <= tol and rp_max <= tol and np.abs(du).max() <= 1e-8:
            break
    return u, xs, it + 1


def min_relaxation(system, x0, A, B, c, G, lo, hi, umin, umax):
    """Feasibility certificate for the QP of sqp.py:87-134, independent of any QP method: smallest t >= 0 such that lo - t <= G x <= hi + t (steps 1..N-1) and umin - t <= u <= umax + t
    has a solution with x_{i+1} = A_i x_i + B_i u_i + c_i, x_0 given."""
    from scipy.optimize import linprog
    T, n, m = A.shape[0], A.shape[1], B.shape[2]
    N = T + 1
    Phi = np.zeros((N, n, T * m))
    phi = np.zeros((N, n))
    phi[0] = x0
    for i in range(T):
        Phi[i + 1] = A[i] @ Phi[i]
        Phi[i + 1][:, i * m:(i + 1) * m] += B[i]
        phi[i + 1] = A[i] @ phi[i] + c[i]
    rows, rhs = [], []
    for i in range(1, N):
        for j in range(2):
            g = G[i, j] @ Phi[i]
            off = G[i, j] @ phi[i]
            if np.isfinite(hi[i, j]):
                rows.append(np.concatenate([g, [-1.0]]))
                rhs.append(hi[i, j] - off)
            if np.isfinite(lo[i, j]):
                rows.append(np.concatenate([-g, [-1.0]]))
                rhs.append(off - lo[i, j])
    for i in range(T):
        for j in range(m):
            e = np.zeros(T * m + 1)
            e[i * m + j], e[-1] = 1.0, -1.0
            if np.isfinite(umax[j]):
                rows.append(e.copy())
                rhs.append(umax[j])
            e[i * m + j] = -1.0
            if np.isfinite(umin[j]):
                rows.append(e.copy())
                rhs.append(-umin[j])
    cost = np.zeros(T * m + 1)
    cost[-1] = 1.0
    res = linprog(cost, A_ub=np.array(rows), b_ub=np.array(rhs), bounds=[(None, None)] * (T * m) + [(0, None)], method="highs")
    return float(res.x[-1]) if res.status == 0 else float("nan")
