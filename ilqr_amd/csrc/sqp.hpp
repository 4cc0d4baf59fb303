// Batched sequential-QP solver (sqp.py of the reference) for models whose state starts with
// (px, py, v), the layout constraints.py:12 is written for.  One lane per problem, tiled
// structure-of-arrays buffers as everywhere else in this library.
//
//   sqp_init_kernel    sqp.py:139-140   min_cost = inf, converge = False
//   sqp_qp_kernel      sqp.py:145-148   the QP of compute_P_q / compute_A_l_u (sqp.py:77-134) is
//                      never assembled: its blocks are recomputed from the linearisation point
//                      in every sweep.  The reference hands the sparse QP to OSQP; here it is
//                      solved by a primal-dual path-following iteration whose Newton step is an
//                      LQR problem (all inequality rows act on one time step), solved exactly by
//                      a Riccati recursion: two sweeps over the horizon per iteration.
//                      Jacobian entries a model declares structurally zero or one are skipped at
//                      compile time (adding an exact zero changes no bit).
//   sqp_search_kernel  sqp.py:149-171   the step-halving search on the simulated cost, all step
//                      sizes of a problem on neighbouring lanes, and the accepted step written
//                      in place.
#pragma once
#include "kernels.hpp"

namespace ilqr {

constexpr int ILQR_STATUS_QP_FAILED_ = 8;

struct SqpArgs {
    void *S, *LAM;     // slacks and multipliers, rows = N * NR
    void *DU;          // Newton direction of the inputs, rows = (N-1) * NU
    void *DS, *DL, *DX; // ... of the slacks and multipliers (rows N * NR each) and of the states (rows N * NX)
    void *UQ, *XQ;     // QP iterate: inputs (rows (N-1)*NU) and states (rows N*NX); UQ ends as the solved inputs
    const void *U0;    // fixed first input (rows NU) or null (sqp.py:111-112)
    const void *X0;    // initial state x0 (rows NX; sqp.py:37, :120, :158) or null = first row of X
    int32_t *qp_iters; // per problem: iterations of the last QP solve (may be null)
    double eps, cost_eps, qp_tol, qp_sigma;
    int32_t qp_max_iter, n_alphas, fixed;
};

template <typename T, class M> struct SqpRows {
    static constexpr int NX = M::NX, NU = M::NU, NR = 3 + 2 * NU;
    // row j of time step i as  grad . (x or u) <= bound ; live = the row exists
    T gx[2];           // bubble row: gradient wrt (px, py) at the linearisation point (constraints.py:24-25)
    T bnd[NR];
    bool live[NR];
    template <class P>
    __device__ __forceinline__ void setup(const P &p, int i, int T1, const T (&xb)[NX], T cx, T cy, T rad, bool fixed0) {
        const T ex = xb[0] - cx, ey = xb[1] - cy;
        gx[0] = T(2) * ex;
        gx[1] = T(2) * ey;
        const T h = fma_(ex, ex, fma_(ey, ey, -(rad * rad)));       // constraints.py:20-22
        bnd[0] = fma_(gx[0], xb[0], fma_(gx[1], xb[1], -h));        // constraints.py:28-30
        bnd[1] = -p.vlo;
        bnd[2] = p.vhi;
        const bool st = i >= 1; // the rows of step 0 only test the given initial state
        live[0] = st;
        live[1] = st && (-bnd[1] > -inf_<T>());
        live[2] = st && (bnd[2] < inf_<T>());
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            bnd[3 + 2 * j] = -p.qlo[j];
            bnd[4 + 2 * j] = p.qhi[j];
            const bool in = i < T1 && !(fixed0 && i == 0);
            live[3 + 2 * j] = in && (p.qlo[j] > -inf_<T>());
            live[4 + 2 * j] = in && (p.qhi[j] < inf_<T>());
        }
    }
    __device__ __forceinline__ T value(int j, const T (&x)[NX], const T (&u)[NU]) const {
        if (j == 0) return fma_(gx[0], x[0], gx[1] * x[1]);
        if (j == 1) return -x[2];
        if (j == 2) return x[2];
        const int c = (j - 3) >> 1;
        return ((j - 3) & 1) ? u[c] : -u[c];
    }
    // directional derivative of row j along (dx, du)
    __device__ __forceinline__ T along(int j, const T (&dx)[NX], const T (&du)[NU]) const { return value(j, dx, du); }
};

// linearisation x+ = A x + B u + c of the model at (xb, ub) with c = xb_next - A xb - B ub (sqp.py:91-94, :124-126)
template <typename T, class M, class P>
__device__ __forceinline__ void sqp_linearise(const P &p, int i, const T (&xb)[M::NX], const T (&ub)[M::NU], const T (&xbn)[M::NX],
                                              T (&A)[M::NX][M::NX], T (&B)[M::NX][M::NU], T (&c)[M::NX]) {
    constexpr int NX = M::NX, NU = M::NU;
#pragma unroll
    for (int a = 0; a < NX; ++a) {
#pragma unroll
        for (int b = 0; b < NX; ++b) A[a][b] = M::fx_one(a, b) ? T(1) : T(0);
#pragma unroll
        for (int b = 0; b < NU; ++b) B[a][b] = T(0);
    }
    bool bad = false;
    M::template jac<true>(p, i, xb, ub, A, B, bad);
    if (bad) M::template jac<false>(p, i, xb, ub, A, B, bad); // an angle outside the fast routine's range
#pragma unroll
    for (int a = 0; a < NX; ++a) {
        T s = xbn[a];
#pragma unroll
        for (int b = 0; b < NX; ++b)
            if (M::fx_nz(a, b)) s = M::fx_one(a, b) ? s - xb[b] : fma_(-(A[a][b]), xb[b], s);
#pragma unroll
        for (int b = 0; b < NU; ++b)
            if (M::fu_nz(a, b)) s = fma_(-(B[a][b]), ub[b], s);
        c[a] = s;
    }
}

template <typename T> __global__ __launch_bounds__(64) void sqp_init_kernel(T *J, int32_t *converged, int32_t *iters, int32_t *alpha_idx, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    J[b] = inf_<T>();
    converged[b] = 0;
    iters[b] = 0;
    alpha_idx[b] = -1;
}

template <typename T, class M>
__global__ __launch_bounds__(64) void sqp_qp_kernel(Params<T, M::NX, M::NU> p, SqpArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NR = 3 + 2 * NU;
    const int b = blockIdx.x * 64 + threadIdx.x;
    const bool valid = b < p.B;
    const bool active = valid && p.converged[valid ? b : 0] == 0;
    if (__ballot(active) == 0ull) return;
    const int N = p.N, T1 = N - 1, tile = blockIdx.x, ln = threadIdx.x;
    const T *Xb = tile_ptr(p.X, tile, N * NX, ln);          // linearisation point (self.states)
    const T *Ub = tile_ptr(p.U, tile, T1 * NU, ln);         // (self.inputs)
    const T *Rt = tile_ptr(p.Xref, tile, N * NX, ln);
    const T *Cp = tile_ptr(p.centers, tile, N * 2, ln);
    const T *Dp = tile_ptr(p.radius, tile, N, ln);
    T *Kp = tile_ptr(p.K, tile, T1 * NU * NX, ln);
    T *kp = tile_ptr(p.k, tile, T1 * NU, ln);
    T *Sp = tile_ptr((T *)a.S, tile, N * NR, ln);
    T *Lp = tile_ptr((T *)a.LAM, tile, N * NR, ln);
    T *DUp = tile_ptr((T *)a.DU, tile, T1 * NU, ln);
    T *DSp = tile_ptr((T *)a.DS, tile, N * NR, ln);
    T *DLp = tile_ptr((T *)a.DL, tile, N * NR, ln);
    T *DXp = tile_ptr((T *)a.DX, tile, N * NX, ln);
    T *UQp = tile_ptr((T *)a.UQ, tile, T1 * NU, ln);
    T *XQp = tile_ptr((T *)a.XQ, tile, N * NX, ln);
    const T *U0p = a.U0 ? tile_ptr((const T *)a.U0, tile, NU, ln) : nullptr;
    const T *X0p = a.X0 ? tile_ptr((const T *)a.X0, tile, NX, ln) : nullptr;
    const bool fixed0 = a.fixed != 0 && U0p != nullptr;
    const T sigma = (T)a.qp_sigma, tol = (T)a.qp_tol;
    using Rows = SqpRows<T, M>;

    auto load_x = [&](const T *base, int i, T (&x)[NX]) {
#pragma unroll
        for (int q = 0; q < NX; ++q) x[q] = base[(size_t)(i * NX + q) * TILE];
    };
    auto load_u = [&](const T *base, int i, T (&u)[NU]) {
#pragma unroll
        for (int q = 0; q < NU; ++q) u[q] = base[(size_t)(i * NU + q) * TILE];
    };

    // Everything a sweep reads of one time step.  Each sweep fetches the record of the step after the
    // one it works on before it starts working (no store of a sweep touches another step's rows), so
    // the memory latency of a step hides behind the arithmetic of its predecessor; with one wave per
    // SIMD at these batch sizes nothing else would hide it.
    struct Step {
        T xb[NX], ub[NU], xq[NX], uq[NU], xr[NX], cx, cy, rad, s[NR], lam[NR], K[NU][NX], k[NU];
    };
    // what: 0 = backward sweep, 1 = forward sweep (needs the gains)
    auto fetch = [&](int i, int what, Step &st) {
        load_x(Xb, i, st.xb);
        load_x(XQp, i, st.xq);
        if (what == 0) load_x(Rt, i, st.xr);
        st.cx = Cp[(size_t)(i * 2) * TILE];
        st.cy = Cp[(size_t)(i * 2 + 1) * TILE];
        st.rad = Dp[(size_t)i * TILE];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            st.s[j] = Sp[(size_t)(i * NR + j) * TILE];
            st.lam[j] = Lp[(size_t)(i * NR + j) * TILE];
        }
#pragma unroll
        for (int q = 0; q < NU; ++q) st.ub[q] = st.uq[q] = st.k[q] = T(0);
        if (i < T1) {
            load_u(Ub, i, st.ub);
            load_u(UQp, i, st.uq);
            if (what == 1) {
                load_u(kp, i, st.k);
#pragma unroll
                for (int r = 0; r < NU; ++r)
#pragma unroll
                    for (int q = 0; q < NX; ++q) st.K[r][q] = Kp[(size_t)((i * NU + r) * NX + q) * TILE];
            }
        }
    };
    // the step the last forward sweep found, for one time step.  Not fetched ahead (that would
    // cost its registers for a whole step): a sweep issues these loads at the top of a step and does
    // work that does not need them first
    struct Delta {
        T ds[NR], dl[NR], dx[NX], du[NU];
    };
    auto fetch_delta = [&](int i, Delta &dv) {
        load_x(DXp, i, dv.dx);
#pragma unroll
        for (int q = 0; q < NU; ++q) dv.du[q] = T(0);
        if (i < T1) load_u(DUp, i, dv.du);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            dv.ds[j] = DSp[(size_t)(i * NR + j) * TILE];
            dv.dl[j] = DLp[(size_t)(i * NR + j) * TILE];
        }
    };

    // ---- start: inputs = current inputs, states by the linearised dynamics, slacks from the rows
    T mu = T(0);
    int nlive = 0;
    {
        T x[NX], xb[NX], xbn[NX];
        load_x(Xb, 0, xb);
#pragma unroll
        for (int q = 0; q < NX; ++q) x[q] = X0p ? X0p[(size_t)q * TILE] : xb[q]; // x_0 = x0 (sqp.py:37, :120), linearised around states[0]
        for (int i = 0; i < N; ++i) {
            T ub[NU], u[NU];
#pragma unroll
            for (int q = 0; q < NU; ++q) ub[q] = u[q] = T(0);
            if (i < T1) {
                load_u(Ub, i, ub);
#pragma unroll
                for (int q = 0; q < NU; ++q) u[q] = (fixed0 && i == 0) ? U0p[(size_t)q * TILE] : ub[q];
            }
            Rows rows;
            rows.setup(p, i, T1, xb, Cp[(size_t)(i * 2) * TILE], Cp[(size_t)(i * 2 + 1) * TILE], Dp[(size_t)i * TILE], fixed0);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                T s = T(1);
                if (rows.live[j]) {
                    const T gap = rows.bnd[j] - rows.value(j, x, u);
                    s = gap > T(1) ? gap : T(1);
                    mu += s;
                    ++nlive;
                }
                if (active) {
                    Sp[(size_t)(i * NR + j) * TILE] = s;
                    Lp[(size_t)(i * NR + j) * TILE] = T(1);
                }
            }
            if (active) {
#pragma unroll
                for (int q = 0; q < NX; ++q) XQp[(size_t)(i * NX + q) * TILE] = x[q];
            }
            if (i < T1) {
                if (active) {
#pragma unroll
                    for (int q = 0; q < NU; ++q) UQp[(size_t)(i * NU + q) * TILE] = u[q];
                }
                load_x(Xb, i + 1, xbn);
                T A[NX][NX], Bm[NX][NU], c[NX], xn[NX];
                sqp_linearise<T, M>(p, i, xb, ub, xbn, A, Bm, c);
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    T s = c[r];
#pragma unroll
                    for (int q = 0; q < NX; ++q)
                        if (M::fx_nz(r, q)) s = M::fx_one(r, q) ? s + x[q] : fma_(A[r][q], x[q], s);
#pragma unroll
                    for (int q = 0; q < NU; ++q)
                        if (M::fu_nz(r, q)) s = fma_(Bm[r][q], u[q], s);
                    xn[r] = s;
                }
#pragma unroll
                for (int q = 0; q < NX; ++q) {
                    x[q] = xn[q];
                    xb[q] = xbn[q];
                }
            }
        }
        mu = nlive > 0 ? mu / (T)nlive : T(0);
    }

    // Per iteration two sweeps: the backward sweep first applies the step found by the previous
    // forward sweep (an elementwise update, so it can run in the backward order) and then does the
    // Riccati recursion of the next Newton step on the updated point; the forward sweep finds the
    // direction and the step lengths and, from four sums, the complementarity the point will have
    // once the step is applied.  A last sweep applies the step of the final iteration.
    bool done = !active, failed = false, pending = false;
    T alpha = T(0), alphad = T(0);
    int it = 0;
    auto apply = [&](int i, const Step &st, const Delta &dv, T (&sv)[NR], T (&lv)[NR], T (&xq)[NX], T (&uq)[NU]) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            sv[j] = fma_(alpha, dv.ds[j], st.s[j]);
            lv[j] = fma_(alphad, dv.dl[j], st.lam[j]);
            Sp[(size_t)(i * NR + j) * TILE] = sv[j];
            Lp[(size_t)(i * NR + j) * TILE] = lv[j];
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            xq[q] = fma_(alpha, dv.dx[q], st.xq[q]);
            XQp[(size_t)(i * NX + q) * TILE] = xq[q];
        }
        if (i < T1) {
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                uq[q] = fma_(alpha, dv.du[q], st.uq[q]);
                UQp[(size_t)(i * NU + q) * TILE] = uq[q];
            }
        }
    };
    for (; it < a.qp_max_iter; ++it) {
        if (__ballot(!done) == 0ull) break;
        // ---- backward sweep: pending step applied, then the Riccati recursion of the Newton step
        {
            const bool any_pending = __ballot(pending) != 0ull;
            T Pm[NX][NX], pv[NX], xbn[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                pv[r] = T(0);
                xbn[r] = T(0);
#pragma unroll
                for (int q = 0; q < NX; ++q) Pm[r][q] = T(0);
            }
            bool bad = false;
            Step nxt;
            fetch(T1, 0, nxt);
            for (int i = T1; i >= 0; --i) {
                const Step st = nxt;
                if (i > 0) fetch(i - 1, 0, nxt);
                Delta dv;
                if (any_pending) fetch_delta(i, dv);
                const T(&xb)[NX] = st.xb;
                const T(&ub)[NU] = st.ub;
                const T(&xr)[NX] = st.xr;
                T xq[NX], uq[NU], sv[NR], lv[NR];
#pragma unroll
                for (int q = 0; q < NX; ++q) xq[q] = st.xq[q];
#pragma unroll
                for (int q = 0; q < NU; ++q) uq[q] = st.uq[q];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    sv[j] = st.s[j];
                    lv[j] = st.lam[j];
                }
                // the linearisation needs nothing of the pending step: its arithmetic covers the latency of
                // the loads just issued
                T A[NX][NX], Bm[NX][NU], c[NX];
                if (i < T1) sqp_linearise<T, M>(p, i, xb, ub, xbn, A, Bm, c);
                if (pending) apply(i, st, dv, sv, lv, xq, uq);
                Rows rows;
                rows.setup(p, i, T1, xb, st.cx, st.cy, st.rad, fixed0);
                const T *W = i == T1 ? p.Qf : p.Q;
                T Hx[NX][NX], gxv[NX], Hu[NU][NU], guv[NU];
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    T s = T(0);
#pragma unroll
                    for (int q = 0; q < NX; ++q) {
                        Hx[r][q] = W[r * NX + q];
                        s = fma_(W[r * NX + q], (xq[q] - xr[q]), s);
                    }
                    gxv[r] = s;
                }
#pragma unroll
                for (int r = 0; r < NU; ++r) {
                    T s = T(0);
#pragma unroll
                    for (int q = 0; q < NU; ++q) {
                        Hu[r][q] = p.R[r * NU + q];
                        s = fma_(p.R[r * NU + q], uq[q], s);
                    }
                    guv[r] = s;
                }
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    if (!rows.live[j]) continue;
                    const T s = sv[j], lam = lv[j];
                    const T rp = rows.value(j, xq, uq) + s - rows.bnd[j];
                    const T rc = s * lam - sigma * mu;
                    const T inv_s = fast_rcp(s); // (one reciprocal per row instead of two divisions: 6 instructions against 24)
                    const T coef = lam + (lam * rp - rc) * inv_s;
                    const T d = lam * inv_s;
                    if (j == 0) {
                        Hx[0][0] = fma_(d * rows.gx[0], rows.gx[0], Hx[0][0]);
                        Hx[0][1] = fma_(d * rows.gx[0], rows.gx[1], Hx[0][1]);
                        Hx[1][0] = fma_(d * rows.gx[1], rows.gx[0], Hx[1][0]);
                        Hx[1][1] = fma_(d * rows.gx[1], rows.gx[1], Hx[1][1]);
                        gxv[0] = fma_(coef, rows.gx[0], gxv[0]);
                        gxv[1] = fma_(coef, rows.gx[1], gxv[1]);
                    } else if (j < 3) {
                        Hx[2][2] += d;
                        gxv[2] += j == 1 ? -coef : coef;
                    } else {
                        const int cu = (j - 3) >> 1;
                        Hu[cu][cu] += d;
                        guv[cu] += ((j - 3) & 1) ? coef : -coef;
                    }
                }
                if (i == T1) {
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        pv[r] = gxv[r];
#pragma unroll
                        for (int q = 0; q < NX; ++q) Pm[r][q] = Hx[r][q];
                    }
                } else {
                    // PA = P A, PB = P B
                    T PA[NX][NX], PB[NX][NU];
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
#pragma unroll
                        for (int q = 0; q < NX; ++q) {
                            T s = T(0);
#pragma unroll
                            for (int w = 0; w < NX; ++w)
                                if (M::fx_nz(w, q)) s = M::fx_one(w, q) ? s + Pm[r][w] : fma_(Pm[r][w], A[w][q], s);
                            PA[r][q] = s;
                        }
#pragma unroll
                        for (int q = 0; q < NU; ++q) {
                            T s = T(0);
#pragma unroll
                            for (int w = 0; w < NX; ++w)
                                if (M::fu_nz(w, q)) s = fma_(Pm[r][w], Bm[w][q], s);
                            PB[r][q] = s;
                        }
                    }
                    T Qxx[NX][NX], Qx[NX], Qux[NU][NX], Quu[NU][NU], Qu[NU];
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        T s = gxv[r];
#pragma unroll
                        for (int w = 0; w < NX; ++w)
                            if (M::fx_nz(w, r)) s = M::fx_one(w, r) ? s + pv[w] : fma_(A[w][r], pv[w], s);
                        Qx[r] = s;
#pragma unroll
                        for (int q = 0; q < NX; ++q) {
                            T t = Hx[r][q];
#pragma unroll
                            for (int w = 0; w < NX; ++w)
                                if (M::fx_nz(w, r)) t = M::fx_one(w, r) ? t + PA[w][q] : fma_(A[w][r], PA[w][q], t);
                            Qxx[r][q] = t;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < NU; ++r) {
                        T s = guv[r];
#pragma unroll
                        for (int w = 0; w < NX; ++w)
                            if (M::fu_nz(w, r)) s = fma_(Bm[w][r], pv[w], s);
                        Qu[r] = s;
#pragma unroll
                        for (int q = 0; q < NX; ++q) {
                            T t = T(0);
#pragma unroll
                            for (int w = 0; w < NX; ++w)
                                if (M::fu_nz(w, r)) t = fma_(Bm[w][r], PA[w][q], t);
                            Qux[r][q] = t;
                        }
#pragma unroll
                        for (int q = 0; q < NU; ++q) {
                            T t = Hu[r][q];
#pragma unroll
                            for (int w = 0; w < NX; ++w)
                                if (M::fu_nz(w, r)) t = fma_(Bm[w][r], PB[w][q], t);
                            Quu[r][q] = t;
                        }
                    }
                    T rhs[NU][NX + 1]; // columns 0..NX-1: -Qux, column NX: -Qu
#pragma unroll
                    for (int r = 0; r < NU; ++r) {
#pragma unroll
                        for (int q = 0; q < NX; ++q) rhs[r][q] = -Qux[r][q];
                        rhs[r][NX] = -Qu[r];
                    }
                    solve_quu<T, NU, NX + 1, true>(Quu, rhs, bad);
                    if (fixed0 && i == 0) { // the first input does not move: du_0 = 0
#pragma unroll
                        for (int r = 0; r < NU; ++r)
#pragma unroll
                            for (int q = 0; q <= NX; ++q) rhs[r][q] = T(0);
                    }
                    if (!done) {
#pragma unroll
                        for (int r = 0; r < NU; ++r) {
                            kp[(size_t)(i * NU + r) * TILE] = rhs[r][NX];
#pragma unroll
                            for (int q = 0; q < NX; ++q) Kp[(size_t)((i * NU + r) * NX + q) * TILE] = rhs[r][q];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        T s = Qx[r];
#pragma unroll
                        for (int w = 0; w < NU; ++w) s = fma_(rhs[w][r], Qu[w], s);
                        pv[r] = s;
#pragma unroll
                        for (int q = 0; q < NX; ++q) {
                            T t = Qxx[r][q];
#pragma unroll
                            for (int w = 0; w < NU; ++w) t = fma_(rhs[w][r], Qux[w][q], t);
                            Pm[r][q] = t;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NX; ++q) xbn[q] = xb[q];
            }
            failed = failed || bad;
            pending = false;
        }
        // ---- forward sweep: direction, step lengths, and what the step will leave behind
        // the largest steps that keep slacks and multipliers positive, min over the rows of s / -ds and
        // lam / -dl, each kept as a fraction (numerator, denominator) and compared by cross-multiplication:
        // one division per sweep instead of two per row and time step
        T an = T(1), ad = T(1), dn = T(1), dd = T(1), rpabs = T(0), dumax = T(0);
        T g0 = T(0), g1 = T(0), g2 = T(0), g3 = T(0); // sums of s lam, ds lam, s dl, ds dl over the live rows
        {
            T dx[NX];
            Step nxt;
            fetch(0, 1, nxt);
#pragma unroll
            for (int q = 0; q < NX; ++q) dx[q] = T(0);
            for (int i = 0; i < N; ++i) {
                const Step st = nxt;
                if (i < T1) fetch(i + 1, 1, nxt);
                const T(&xb)[NX] = st.xb;
                const T(&ub)[NU] = st.ub;
                const T(&xq)[NX] = st.xq;
                const T(&uq)[NU] = st.uq;
                T du[NU];
#pragma unroll
                for (int q = 0; q < NU; ++q) du[q] = T(0);
                if (i < T1) {
#pragma unroll
                    for (int r = 0; r < NU; ++r) {
                        T s = st.k[r];
#pragma unroll
                        for (int q = 0; q < NX; ++q) s = fma_(st.K[r][q], dx[q], s);
                        du[r] = s;
                        // (the direction arrays are scratch between this sweep and the next one's `apply`,
                        // which a finished lane never runs: every lane stores, no branch around the stores)
                        DUp[(size_t)(i * NU + r) * TILE] = s;
                        const T m = fabs_(s);
                        dumax = m > dumax ? m : dumax;
                    }
                }
#pragma unroll
                for (int q = 0; q < NX; ++q) DXp[(size_t)(i * NX + q) * TILE] = dx[q];
                Rows rows;
                rows.setup(p, i, T1, xb, st.cx, st.cy, st.rad, fixed0);
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    T ds = T(0), dl = T(0);
                    if (rows.live[j]) {
                        const T s = st.s[j], lam = st.lam[j];
                        const T rp = rows.value(j, xq, uq) + s - rows.bnd[j];
                        const T rc = s * lam - sigma * mu;
                        ds = -rp - rows.along(j, dx, du);
                        dl = (-rc - lam * ds) * fast_rcp(s);
                        {   // s / -ds < an / ad  <=>  s ad < an (-ds)   (all four positive)
                            const bool lower = ds < T(0) && s * ad < an * -ds;
                            an = lower ? s : an;
                            ad = lower ? -ds : ad;
                        }
                        {
                            const bool lower = dl < T(0) && lam * dd < dn * -dl;
                            dn = lower ? lam : dn;
                            dd = lower ? -dl : dd;
                        }
                        g0 = fma_(s, lam, g0);
                        g1 = fma_(ds, lam, g1);
                        g2 = fma_(s, dl, g2);
                        g3 = fma_(ds, dl, g3);
                        const T r1 = fabs_(rp);
                        rpabs = r1 > rpabs ? r1 : rpabs;
                    }
                    DSp[(size_t)(i * NR + j) * TILE] = ds;
                    DLp[(size_t)(i * NR + j) * TILE] = dl;
                }
                if (i < T1) {
                    T A[NX][NX], Bm[NX][NU], c[NX];
                    sqp_linearise<T, M>(p, i, xb, ub, nxt.xb, A, Bm, c);
                    T dxn[NX];
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        T s = T(0);
#pragma unroll
                        for (int q = 0; q < NX; ++q)
                            if (M::fx_nz(r, q)) s = M::fx_one(r, q) ? s + dx[q] : fma_(A[r][q], dx[q], s);
#pragma unroll
                        for (int q = 0; q < NU; ++q)
                            if (M::fu_nz(r, q)) s = fma_(Bm[r][q], du[q], s);
                        dxn[r] = s;
                    }
#pragma unroll
                    for (int q = 0; q < NX; ++q) dx[q] = dxn[q];
                }
            }
        }
        if (!done) {
            // separate step lengths: alpha for inputs, states and slacks, alphad for the multipliers
            const T amax = an / ad, dmax = dn / dd;
            const T am = T(0.995) * amax, dm = T(0.995) * dmax;
            alpha = am < T(1) ? am : T(1);
            alphad = dm < T(1) ? dm : T(1);
            pending = true;
            // complementarity and row residual of the point once the step is applied
            const T gap = g0 + alpha * g1 + alphad * g2 + (alpha * alphad) * g3;
            const T rpmax = rpabs * (T(1) - alpha);
            mu = nlive > 0 ? gap / (T)nlive : T(0);
            const bool finite = fabs_(mu) < inf_<T>() && fabs_(dumax) < inf_<T>();
            // a step length that collapses means the rows cannot all be met (the multipliers of an
            // infeasible set of rows grow without bound while the slacks stall)
            if (!finite || alpha < T(1e-8) || alphad < T(1e-8)) {
                failed = true;
                done = true;
            } else if (mu <= tol && rpmax <= tol && dumax <= T(1e-8)) {
                done = true;
            }
            if (done && a.qp_iters) a.qp_iters[b] = it + 1;
        }
    }
    // ---- the step of the last iteration
    if (__ballot(pending) != 0ull) {
        for (int i = 0; i < N; ++i) {
            Step st;
            Delta dv;
            fetch(i, 0, st);
            fetch_delta(i, dv);
            T xq[NX], uq[NU], sv[NR], lv[NR];
            if (pending) apply(i, st, dv, sv, lv, xq, uq);
        }
    }
    if (active && (!done || failed)) {
        // no solution (infeasible rows, or the iteration limit): the step is zero, which the search
        // below turns into `converge`; the reference would receive OSQP's failure output instead
        if (p.status) p.status[b] |= ILQR_STATUS_QP_FAILED_;
        if (!done && a.qp_iters) a.qp_iters[b] = a.qp_max_iter;
        for (int i = 0; i < T1; ++i)
#pragma unroll
            for (int q = 0; q < NU; ++q) UQp[(size_t)(i * NU + q) * TILE] = Ub[(size_t)(i * NU + q) * TILE];
    }
}

// Step-halving search of sqp.py:153-171 with LS step sizes of a problem on LS neighbouring lanes:
// inputs = prev + alpha d_u, states by simulation from x0, cost as in sqp.py:44-54.  The serial
// loop accepts the first alpha whose cost is below min_cost and gives up (converge) at the first
// rejected alpha that is below eps or whose cost is within cost_eps of min_cost; evaluated in
// parallel that is: the lowest accepting index wins if no lower index gives up.  Then every lane of
// the problem rolls the winning step size out once more (same instruction sequence, same bits)
// and its first lane stores states and inputs in place.
template <typename T, class M, int LS>
__global__ __launch_bounds__(64) void sqp_search_kernel(Params<T, M::NX, M::NU> p, SqpArgs a, int it) {
    constexpr int NX = M::NX, NU = M::NU, PPW = 64 / LS;
    const int lane = threadIdx.x, pl = lane / LS, al = lane % LS;
    const int g = blockIdx.x * PPW + pl;
    const bool valid = g < p.B;
    const int b = valid ? g : p.B - 1;
    const bool active = valid && p.converged[b] == 0;
    if (__ballot(active) == 0ull) return;
    const int N = p.N, T1 = N - 1, tile = b >> 6, ln = b & 63;
    T *Xt = tile_ptr(p.X, tile, N * NX, ln);
    T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    const T *Rt = tile_ptr(p.Xref, tile, N * NX, ln);
    const T *UQp = tile_ptr((const T *)a.UQ, tile, T1 * NU, ln);
    const T *X0p = a.X0 ? tile_ptr((const T *)a.X0, tile, NX, ln) : nullptr;
    const T Jcur = p.J[b];
    const T eps = (T)a.eps, cost_eps = (T)a.cost_eps;

    auto rollout = [&](T alpha, bool store) -> T {
        T x[NX], cost = T(0);
#pragma unroll
        for (int q = 0; q < NX; ++q) x[q] = X0p ? X0p[(size_t)q * TILE] : Xt[(size_t)q * TILE]; // sim(self.x0, ...), sqp.py:158
        if (store && X0p) {
#pragma unroll
            for (int q = 0; q < NX; ++q) Xt[(size_t)q * TILE] = x[q];
        }
        bool bad = false;
        for (int i = 0; i < T1; ++i) {
            T u[NU], e[NX], xn[NX];
#pragma unroll
            for (int q = 0; q < NU; ++q) {
                const T up = Ut[(size_t)(i * NU + q) * TILE];
                u[q] = up + alpha * (UQp[(size_t)(i * NU + q) * TILE] - up);     // sqp.py:153, :157
                e[q] = T(0);
            }
#pragma unroll
            for (int q = 0; q < NX; ++q) e[q] = x[q] - Rt[(size_t)(i * NX + q) * TILE];
            cost += quad_<T, NX, false>(p.Q, e) + quad_<T, NU, false>(p.R, u);  // sqp.py:47-51
            M::template step<false>(p, i, x, u, xn, bad);                        // sqp.py:59-60
            if (store) {
#pragma unroll
                for (int q = 0; q < NU; ++q) Ut[(size_t)(i * NU + q) * TILE] = u[q];
#pragma unroll
                for (int q = 0; q < NX; ++q) Xt[(size_t)((i + 1) * NX + q) * TILE] = xn[q];
            }
#pragma unroll
            for (int q = 0; q < NX; ++q) x[q] = xn[q];
        }
        T e[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) e[q] = x[q] - Rt[(size_t)(T1 * NX + q) * TILE];
        return cost + quad_<T, NX, false>(p.Qf, e);                              // sqp.py:52-53
    };

    int win = 0x7fffffff, stop = 0x7fffffff;
    T win_cost = T(0);
    for (int base = 0; base < a.n_alphas; base += LS) {
        const int ai = base + al;
        const T alpha = pow2_neg<T>(ai);
        const T c = rollout(alpha, false);
        int acc = 0x7fffffff, giveup = 0x7fffffff;
        T acc_cost = T(0);
        if (ai < a.n_alphas) {
            if (c < Jcur) {
                acc = ai;
                acc_cost = c;
            } else if (alpha < eps || fabs_(T(1) - c / Jcur) < cost_eps) {         // sqp.py:168
                giveup = ai;
            }
        }
#pragma unroll
        for (int off = 1; off < LS; off <<= 1) {
            const int oa = __shfl_xor(acc, off), og = __shfl_xor(giveup, off);
            const T oc = __shfl_xor(acc_cost, off);
            if (oa < acc) {
                acc = oa;
                acc_cost = oc;
            }
            giveup = og < giveup ? og : giveup;
        }
        if (win == 0x7fffffff && stop == 0x7fffffff) {
            if (acc < giveup) {
                win = acc;
                win_cost = acc_cost;
            } else if (giveup != 0x7fffffff) {
                stop = giveup;
            }
        }
        if (__ballot(active && win == 0x7fffffff && stop == 0x7fffffff) == 0ull) break;
    }
    const bool found = active && win != 0x7fffffff;
    if (active && al == 0) {
        p.iters[b] += 1;
        p.alpha_idx[b] = found ? win : -1;
        if (found) {
            p.J[b] = win_cost;
            if (p.status && !(fabs_(win_cost) < inf_<T>())) p.status[b] |= ILQR_STATUS_NONFINITE_;
        } else {
            p.converged[b] = 1;                                                  // sqp.py:169
        }
        if (it < p.hist_cap) {
            if (p.hist_cost) p.hist_cost[size_t(it) * p.hist_stride + b] = found ? win_cost : Jcur;
            if (p.hist_alpha) p.hist_alpha[size_t(it) * p.hist_stride + b] = found ? win : -1;
        }
    }
    if (__ballot(found) == 0ull) return;
    if (found) (void)rollout(pow2_neg<T>(win), al == 0);
}

// launchers --------------------------------------------------------------------------------------
struct SqpOps {
    hipError_t (*init)(const HostCtx &, hipStream_t);
    hipError_t (*qp)(const HostCtx &, const SqpArgs &, hipStream_t);
    hipError_t (*search)(const HostCtx &, const SqpArgs &, int it, hipStream_t);
};

template <typename T, template <typename> class MT> struct SqpLaunch {
    using M = MT<T>;
    static constexpr int NX = M::NX, NU = M::NU;
    static Params<T, NX, NU> params(const HostCtx &c) {
        auto p = make_params<T, NX, NU>(c);
        // the QP bounds the inputs by umin / umax = the system's limits, or nothing (sqp.py:38-42)
        for (int i = 0; i < NU; ++i) {
            p.qlo[i] = c.d.control_limited ? (T)c.d.lo[i] : -std::numeric_limits<T>::infinity();
            p.qhi[i] = c.d.control_limited ? (T)c.d.hi[i] : std::numeric_limits<T>::infinity();
        }
        return p;
    }
    static hipError_t init(const HostCtx &c, hipStream_t s) {
        hipLaunchKernelGGL(sqp_init_kernel<T>, dim3((c.d.B + 63) / 64), dim3(64), 0, s, (T *)c.b.J, c.b.converged, c.b.iters, c.b.alpha_idx, c.d.B);
        return hipGetLastError();
    }
    static hipError_t qp(const HostCtx &c, const SqpArgs &a, hipStream_t s) {
        hipLaunchKernelGGL((sqp_qp_kernel<T, M>), dim3((c.d.B + 63) / 64), dim3(64), 0, s, params(c), a);
        return hipGetLastError();
    }
    static hipError_t search(const HostCtx &c, const SqpArgs &a, int it, hipStream_t s) {
        constexpr int LS = 8;
        hipLaunchKernelGGL((sqp_search_kernel<T, M, LS>), dim3((c.d.B + 64 / LS - 1) / (64 / LS)), dim3(64), 0, s, params(c), a, it);
        return hipGetLastError();
    }
    static const SqpOps *table() {
        static const SqpOps ops = {&init, &qp, &search};
        return &ops;
    }
};

// the sequential-QP kernels of the Car model, fp64 (ops_sqp.hip); nullptr otherwise
const SqpOps *sqp_ops(int model, int dtype);

} // namespace ilqr
