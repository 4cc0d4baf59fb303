/*
 * CPU restatement of the reference iLQR / constrained-iLQR solve path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the timed
 * "CPU port" baseline; nothing in the shipped GPU path links, loads or calls
 * it.  It restates, per problem and in plain scalar fp64 C, the algorithm of
 *
 *   ilqr.py:36-46    cost            -> orc_cost()
 *   ilqr.py:76-105   backward_pass   -> orc_backward()
 *   ilqr.py:48-74    forward_pass    -> orc_forward()
 *   ilqr.py:107-114  __call__        -> ilqr_oracle_solve()
 *   systems.py:33-57   Car             -> model 0
 *   systems.py:64-90   CarAcceleration -> model 1
 *   systems.py:97-121  DubinsCar       -> model 2
 *   any other System duck type (systems.py:4-24) -> model 100, dynamics called back
 *   cilqr.py:87-139  backward_pass with the per-step box QP -> orc_backward(),
 *                    solver_control_limited != 0
 *   constraints.py:20-26  BubbleConstraint.get_h / dh_dx -> barrier terms
 *
 * Pinning: the ilqr.py/systems.py part is checked against fixtures generated
 * by running the reference itself (tests/golden/, oracle/make_golden.py).
 * The box-QP branch follows cilqr.py as text with an exact corner-KKT solve in
 * place of the un-vendored third-party OSQP (version not pinned by the
 * reference): PARITY UNPINNED for that branch.  The relaxed log-barrier mode
 * has no reference implementation at all (only the constraint functions come
 * from constraints.py): PARITY UNPINNED, checked by finite differences.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see Makefile).
 * -ffp-contract=off keeps the operation sequence of the numpy reference
 * (no fused multiply-adds appear that numpy would not perform).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXN 12
#define ORC_MAXM 4

enum { ORC_CAR = 0, ORC_CARACC = 1, ORC_DUBINS = 2, ORC_LTV = 3, ORC_CALLBACK = 100 };

/* model 100: the dynamics are the caller's functions, the way the reference calls
 * system.model_f (ilqr.py:62) and system.compute_df_dx / compute_df_du (ilqr.py:87-88) on
 * whatever System object it was given */
typedef void (*orc_user_f)(int32_t t, const double *x, const double *u, double *xn);
typedef void (*orc_user_jac)(int32_t t, const double *x, const double *u, double *fx, double *fu);

typedef struct ilqr_oracle_desc {
    int32_t n, m, N, model;
    int32_t sys_control_limited;    /* system.control_limited: clamp in the rollout (ilqr.py:58-61) */
    int32_t solver_control_limited; /* cilqr.py:36: per-step box QP in the backward pass */
    int32_t barrier;                /* relaxed log-barrier on the bubble/velocity constraints */
    int32_t max_iter;               /* ilqr.py:28 (30) / cilqr.py:37 (100) */
    double dt;                      /* system.dt (systems.py:30) */
    double lm;                      /* LM_parameter, added to Vxx (ilqr.py:95) */
    double alpha_floor;             /* 1e-4 (ilqr.py:71) / alpha_terminal (cilqr.py:82) */
    double input_limit_lambda;      /* cilqr.py:40 */
    double mu, delta;               /* barrier weight and relaxation width */
    double Q[ORC_MAXN * ORC_MAXN], R[ORC_MAXM * ORC_MAXM], Qf[ORC_MAXN * ORC_MAXN];
    double lo[ORC_MAXM], hi[ORC_MAXM];
    double vel_bounds[2];
    const double *ltv_A;            /* model 3: [N-1][n][n], shared by the batch */
    const double *ltv_B;            /* model 3: [N-1][n][m] */
    orc_user_f user_f;              /* model 100 */
    orc_user_jac user_jac;          /* model 100 */
} ilqr_oracle_desc;

/* ---- dynamics ---------------------------------------------------------- */

static void model_f(const ilqr_oracle_desc *d, int t, const double *x, const double *u, double *xn)
{
    const double dt = d->dt;
    switch (d->model) {
    case ORC_CAR: { /* systems.py:33-40 */
        double th = x[3], v = x[2];
        xn[0] = x[0] + (v * cos(th)) * dt;
        xn[1] = x[1] + (v * sin(th)) * dt;
        xn[2] = x[2] + u[0] * dt;
        xn[3] = x[3] + u[1] * dt;
        break;
    }
    case ORC_CARACC: { /* systems.py:64-72 */
        double th = x[4], v = x[2];
        xn[0] = x[0] + (v * cos(th)) * dt;
        xn[1] = x[1] + (v * sin(th)) * dt;
        xn[2] = x[2] + x[3] * dt;
        xn[3] = x[3] + u[0] * dt;
        xn[4] = x[4] + u[1] * dt;
        break;
    }
    case ORC_DUBINS: { /* systems.py:97-103 */
        double th = x[2], v = u[0];
        xn[0] = x[0] + (v * cos(th)) * dt;
        xn[1] = x[1] + (v * sin(th)) * dt;
        xn[2] = x[2] + (v * u[1]) * dt;
        break;
    }
    case ORC_CALLBACK: d->user_f(t, x, u, xn); break;
    default: { /* table-driven LTV: x+ = A_t x + B_t u (no reference counterpart) */
        const int n = d->n, m = d->m;
        const double *A = d->ltv_A + (size_t)t * n * n, *Bm = d->ltv_B + (size_t)t * n * m;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += A[i * n + j] * x[j];
            for (int j = 0; j < m; ++j) s += Bm[i * m + j] * u[j];
            xn[i] = s;
        }
    }
    }
}

/* fx is n x n row-major, fu is n x m row-major */
static void model_jac(const ilqr_oracle_desc *d, int t, const double *x, const double *u,
                      double *fx, double *fu)
{
    const int n = d->n, m = d->m;
    const double dt = d->dt;
    memset(fx, 0, sizeof(double) * n * n);
    memset(fu, 0, sizeof(double) * n * m);
    switch (d->model) {
    case ORC_CAR: { /* systems.py:42-57 */
        double th = x[3], v = x[2];
        for (int i = 0; i < 4; ++i) fx[i * 4 + i] = 1.0;
        fx[0 * 4 + 2] = cos(th) * dt;
        fx[0 * 4 + 3] = -sin(th) * v * dt;
        fx[1 * 4 + 2] = sin(th) * dt;
        fx[1 * 4 + 3] = cos(th) * v * dt;
        fu[2 * 2 + 0] = 1.0 * dt;
        fu[3 * 2 + 1] = 1.0 * dt;
        break;
    }
    case ORC_CARACC: { /* systems.py:74-90 */
        double th = x[4], v = x[2];
        for (int i = 0; i < 5; ++i) fx[i * 5 + i] = 1.0;
        fx[0 * 5 + 2] = cos(th) * dt;
        fx[0 * 5 + 4] = -sin(th) * v * dt;
        fx[1 * 5 + 2] = sin(th) * dt;
        fx[1 * 5 + 4] = cos(th) * v * dt;
        fx[2 * 5 + 3] = dt;
        fu[3 * 2 + 0] = 1.0 * dt;
        fu[4 * 2 + 1] = 1.0 * dt;
        break;
    }
    case ORC_DUBINS: { /* systems.py:105-121 */
        double th = x[2], v = u[0], curv = u[1];
        for (int i = 0; i < 3; ++i) fx[i * 3 + i] = 1.0;
        fx[0 * 3 + 2] = -sin(th) * v * dt;
        fx[1 * 3 + 2] = cos(th) * v * dt;
        fu[0 * 2 + 0] = cos(th) * dt;
        fu[1 * 2 + 0] = sin(th) * dt;
        fu[2 * 2 + 0] = curv * dt;
        fu[2 * 2 + 1] = v * dt;
        break;
    }
    case ORC_CALLBACK: d->user_jac(t, x, u, fx, fu); break;
    default:
        memcpy(fx, d->ltv_A + (size_t)t * n * n, sizeof(double) * n * n);
        memcpy(fu, d->ltv_B + (size_t)t * n * m, sizeof(double) * n * m);
    }
}

/* ---- relaxed log barrier (no reference counterpart) -------------------- */
/* beta(z) = -log z for z > delta, quadratic extension below; returns value and
 * first/second derivative.  The constraint functions are constraints.py:20-26:
 * h = (px-cx)^2 + (py-cy)^2 - r^2 <= 0 and vel_bounds[0] <= v <= vel_bounds[1]. */
static double relaxed_barrier(double z, double delta, double *d1, double *d2)
{
    if (z > delta) {
        *d1 = -1.0 / z;
        *d2 = 1.0 / (z * z);
        return -log(z);
    }
    double w = (z - 2.0 * delta) / delta;
    *d1 = w / delta;
    *d2 = 1.0 / (delta * delta);
    return 0.5 * (w * w - 1.0) - log(delta);
}

/* value of the barrier at state x (uses components 0,1,2 like constraints.py:12) */
static double barrier_terms(const ilqr_oracle_desc *d, const double *x, const double *c, double r,
                            double *grad /*n or NULL*/, double *hess /*n*n or NULL*/)
{
    const int n = d->n;
    double d1, d2;
    double ex = x[0] - c[0], ey = x[1] - c[1];
    double h = ex * ex + ey * ey - r * r;          /* constraints.py:21 */
    double val = relaxed_barrier(-h, d->delta, &d1, &d2);
    if (grad) {
        memset(grad, 0, sizeof(double) * n);
        memset(hess, 0, sizeof(double) * n * n);
        /* z = -h, dz/dx = -dh/dx = -[2ex, 2ey, 0, 0] (constraints.py:25 row 0) */
        double gx = -2.0 * ex, gy = -2.0 * ey;
        grad[0] += d->mu * d1 * gx;
        grad[1] += d->mu * d1 * gy;
        hess[0 * n + 0] += d->mu * (d2 * gx * gx + d1 * -2.0);
        hess[0 * n + 1] += d->mu * (d2 * gx * gy);
        hess[1 * n + 0] += d->mu * (d2 * gy * gx);
        hess[1 * n + 1] += d->mu * (d2 * gy * gy + d1 * -2.0);
    }
    double e1, e2;
    double v1 = relaxed_barrier(x[2] - d->vel_bounds[0], d->delta, &e1, &e2);
    if (grad) {
        grad[2] += d->mu * e1;
        hess[2 * n + 2] += d->mu * e2;
    }
    double v2 = relaxed_barrier(d->vel_bounds[1] - x[2], d->delta, &e1, &e2);
    if (grad) {
        grad[2] += -d->mu * e1;
        hess[2 * n + 2] += d->mu * e2;
    }
    return d->mu * ((val + v1) + v2);
}

/* ---- cost (ilqr.py:36-46) ---------------------------------------------- */

static double quad_form(const double *M, const double *v, int n)
{
    /* np.dot(np.dot(v.T, M), v): row vector times matrix, then dot */
    double tmp[ORC_MAXN];
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += v[i] * M[i * n + j];
        tmp[j] = s;
    }
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += tmp[j] * v[j];
    return s;
}

double orc_cost(const ilqr_oracle_desc *d, const double *target, const double *states,
                const double *inputs, const double *centers, const double *radius)
{
    const int n = d->n, m = d->m, N = d->N;
    double cost = 0.0, e[ORC_MAXN];
    for (int i = 0; i < N - 1; ++i) {
        for (int j = 0; j < n; ++j) e[j] = states[i * n + j] - target[i * n + j];
        double stage = quad_form(d->Q, e, n) + quad_form(d->R, inputs + i * m, m);
        if (d->barrier && i > 0)
            stage += barrier_terms(d, states + i * n, centers + i * 2, radius[i], NULL, NULL);
        cost += stage;
    }
    for (int j = 0; j < n; ++j) e[j] = states[(N - 1) * n + j] - target[(N - 1) * n + j];
    double term = quad_form(d->Qf, e, n);
    if (d->barrier)
        term += barrier_terms(d, states + (N - 1) * n, centers + (N - 1) * 2, radius[N - 1], NULL, NULL);
    cost += term;
    return cost;
}

/* ---- small dense helpers (row-major) ----------------------------------- */

/* C(p x r) = A^T(p x q) * B(q x r) where A is stored q x p */
static void matTmul(const double *A, const double *B, double *C, int q, int p, int r)
{
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < r; ++j) {
            double s = 0.0;
            for (int k = 0; k < q; ++k) s += A[k * p + i] * B[k * r + j];
            C[i * r + j] = s;
        }
}
/* C(p x r) = A(p x q) * B(q x r) */
static void matmul(const double *A, const double *B, double *C, int p, int q, int r)
{
    for (int i = 0; i < p; ++i)
        for (int j = 0; j < r; ++j) {
            double s = 0.0;
            for (int k = 0; k < q; ++k) s += A[i * q + k] * B[k * r + j];
            C[i * r + j] = s;
        }
}

/* inverse by LU with partial pivoting, solving against the identity, which is
 * what numpy.linalg.inv (LAPACK gesv) does (ilqr.py:99).  Returns 0, or -1 on
 * an exactly zero pivot (numpy raises LinAlgError there). */
static int mat_inv(const double *A, double *Ainv, int m)
{
    double LU[ORC_MAXM * ORC_MAXM];
    int piv[ORC_MAXM];
    memcpy(LU, A, sizeof(double) * m * m);
    for (int c = 0; c < m; ++c) {
        int p = c;
        double best = fabs(LU[c * m + c]);
        for (int r = c + 1; r < m; ++r)
            if (fabs(LU[r * m + c]) > best) { best = fabs(LU[r * m + c]); p = r; }
        piv[c] = p;
        if (best == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < m; ++j) { double t = LU[c * m + j]; LU[c * m + j] = LU[p * m + j]; LU[p * m + j] = t; }
        double inv = 1.0 / LU[c * m + c];
        for (int r = c + 1; r < m; ++r) {
            LU[r * m + c] *= inv;
            for (int j = c + 1; j < m; ++j) LU[r * m + j] -= LU[r * m + c] * LU[c * m + j];
        }
    }
    for (int col = 0; col < m; ++col) {
        double b[ORC_MAXM];
        for (int i = 0; i < m; ++i) b[i] = (i == col) ? 1.0 : 0.0;
        for (int c = 0; c < m; ++c)
            if (piv[c] != c) { double t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }
        for (int i = 1; i < m; ++i)
            for (int j = 0; j < i; ++j) b[i] -= LU[i * m + j] * b[j];
        for (int i = m - 1; i >= 0; --i) {
            for (int j = i + 1; j < m; ++j) b[i] -= LU[i * m + j] * b[j];
            b[i] /= LU[i * m + i];
        }
        for (int i = 0; i < m; ++i) Ainv[i * m + col] = b[i];
    }
    return 0;
}

/* Exact solution test for the box QP of cilqr.py:112-124,
 *     min 1/2 du' P du + q' du,  l <= du <= u,   P = Quu, q = 2 Qu,
 * restricted to what cilqr.py:128-134 consumes: the result replaces the gains
 * only if EVERY multiplier has |y_j| > lambda, i.e. the minimiser sits in a
 * corner of the box with all bounds strictly active.  A corner c is the
 * minimiser iff y = -(P c + q) has y_j <= 0 on lower and y_j >= 0 on upper
 * bounds (OSQP's sign convention for l <= A x <= u).  Returns 1 and the corner
 * in du if such a corner exists with all |y_j| > lambda, else 0. */
static int boxqp_all_active(const double *P, const double *q, const double *l, const double *u,
                            int m, double lambda, double *du)
{
    for (int mask = 0; mask < (1 << m); ++mask) {
        double c[ORC_MAXM];
        for (int j = 0; j < m; ++j) c[j] = ((mask >> j) & 1) ? u[j] : l[j];
        int ok = 1;
        for (int i = 0; i < m && ok; ++i) {
            double g = q[i];
            for (int j = 0; j < m; ++j) g += P[i * m + j] * c[j];
            double y = -g;
            if ((mask >> i) & 1) { if (!(y > lambda)) ok = 0; }
            else                 { if (!(y < -lambda)) ok = 0; }
        }
        if (ok) { memcpy(du, c, sizeof(double) * m); return 1; }
    }
    return 0;
}

/* ---- backward pass (ilqr.py:76-105, cilqr.py:87-139) -------------------- */
/* K is [N-1][m][n], k is [N-1][m].  Returns 0, or -1 if a Quu was singular. */
int orc_backward(const ilqr_oracle_desc *d, const double *target, const double *states,
                 const double *inputs, const double *centers, const double *radius,
                 double *K, double *k)
{
    const int n = d->n, m = d->m, N = d->N;
    double Vx[ORC_MAXN], Vxx[ORC_MAXN * ORC_MAXN], e[ORC_MAXN];
    double fx[ORC_MAXN * ORC_MAXN], fu[ORC_MAXN * ORC_MAXM];
    double lx[ORC_MAXN], lu[ORC_MAXM], lxx[ORC_MAXN * ORC_MAXN];
    double Qx[ORC_MAXN], Qu[ORC_MAXM], Va[ORC_MAXN * ORC_MAXN];
    double T1[ORC_MAXN * ORC_MAXN], T2[ORC_MAXM * ORC_MAXN];
    double Qxx[ORC_MAXN * ORC_MAXN], Quu[ORC_MAXM * ORC_MAXM], Qux[ORC_MAXM * ORC_MAXN];
    double Qinv[ORC_MAXM * ORC_MAXM], kk[ORC_MAXM], KK[ORC_MAXM * ORC_MAXN];
    double bg[ORC_MAXN], bh[ORC_MAXN * ORC_MAXN];
    int rc = 0;

    memset(k, 0, sizeof(double) * (N - 1) * m);
    memset(K, 0, sizeof(double) * (N - 1) * m * n);
    for (int j = 0; j < n; ++j) e[j] = states[(N - 1) * n + j] - target[(N - 1) * n + j];
    for (int i = 0; i < n; ++i) { /* ilqr.py:79-82 */
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += d->Qf[i * n + j] * e[j];
        Vx[i] = 2.0 * s;
        for (int j = 0; j < n; ++j) Vxx[i * n + j] = 2.0 * d->Qf[i * n + j];
    }
    if (d->barrier) {
        barrier_terms(d, states + (N - 1) * n, centers + (N - 1) * 2, radius[N - 1], bg, bh);
        for (int i = 0; i < n; ++i) Vx[i] += bg[i];
        for (int i = 0; i < n * n; ++i) Vxx[i] += bh[i];
    }
    for (int t = N - 2; t >= 0; --t) {
        const double *u = inputs + t * m, *x = states + t * n;
        model_jac(d, t, x, u, fx, fu);
        for (int j = 0; j < n; ++j) e[j] = x[j] - target[t * n + j];
        for (int i = 0; i < n; ++i) { /* ilqr.py:91 */
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += d->Q[i * n + j] * e[j];
            lx[i] = 2.0 * s;
            for (int j = 0; j < n; ++j) lxx[i * n + j] = 2.0 * d->Q[i * n + j];
        }
        for (int i = 0; i < m; ++i) { /* ilqr.py:92 */
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += d->R[i * m + j] * u[j];
            lu[i] = 2.0 * s;
        }
        if (d->barrier && t > 0) {
            barrier_terms(d, x, centers + t * 2, radius[t], bg, bh);
            for (int i = 0; i < n; ++i) lx[i] += bg[i];
            for (int i = 0; i < n * n; ++i) lxx[i] += bh[i];
        }
        matTmul(fx, Vx, Qx, n, n, 1); /* ilqr.py:93-94 */
        for (int i = 0; i < n; ++i) Qx[i] = lx[i] + Qx[i];
        matTmul(fu, Vx, Qu, n, m, 1);
        for (int i = 0; i < m; ++i) Qu[i] = lu[i] + Qu[i];
        memcpy(Va, Vxx, sizeof(double) * n * n); /* ilqr.py:95 */
        for (int i = 0; i < n; ++i) Va[i * n + i] = Vxx[i * n + i] + d->lm * 1.0;
        matTmul(fx, Va, T1, n, n, n); /* ilqr.py:96 */
        matmul(T1, fx, Qxx, n, n, n);
        for (int i = 0; i < n * n; ++i) Qxx[i] = lxx[i] + Qxx[i];
        matTmul(fu, Va, T2, n, m, n); /* ilqr.py:97-98 */
        matmul(T2, fu, Quu, m, n, m);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) Quu[i * m + j] = 2.0 * d->R[i * m + j] + Quu[i * m + j];
        matmul(T2, fx, Qux, m, n, n);
        if (mat_inv(Quu, Qinv, m) != 0) { rc = -1; break; } /* ilqr.py:99 */
        matmul(Qinv, Qu, kk, m, m, 1); /* ilqr.py:100-101 */
        matmul(Qinv, Qux, KK, m, m, n);
        for (int i = 0; i < m; ++i) kk[i] = -kk[i];
        for (int i = 0; i < m * n; ++i) KK[i] = -KK[i];
        if (d->solver_control_limited) { /* cilqr.py:112-134 */
            double q2[ORC_MAXM], l[ORC_MAXM], ub[ORC_MAXM], du[ORC_MAXM];
            for (int i = 0; i < m; ++i) {
                q2[i] = 2.0 * Qu[i];
                l[i] = d->lo[i] - u[i];
                ub[i] = d->hi[i] - u[i];
            }
            if (boxqp_all_active(Quu, q2, l, ub, m, d->input_limit_lambda, du)) {
                for (int i = 0; i < m; ++i) kk[i] = du[i];
                for (int i = 0; i < m * n; ++i) KK[i] = 0.0;
            }
        }
        memcpy(k + t * m, kk, sizeof(double) * m);
        memcpy(K + t * m * n, KK, sizeof(double) * m * n);
        matTmul(KK, Qu, T1, m, n, 1); /* ilqr.py:104 */
        for (int i = 0; i < n; ++i) Vx[i] = Qx[i] + T1[i];
        matTmul(KK, Qux, T1, m, n, n); /* ilqr.py:105 */
        for (int i = 0; i < n * n; ++i) Vxx[i] = Qxx[i] + T1[i];
    }
    return rc;
}

/* ---- forward pass (ilqr.py:48-74) --------------------------------------- */
/* Returns the index of the accepted alpha = 2^-index, or -1 if none was
 * accepted (then states/inputs are restored and *converge is set). */
int orc_forward(const ilqr_oracle_desc *d, const double *target, double *states, double *inputs,
                const double *centers, const double *radius, const double *K, const double *k,
                double *min_cost, int32_t *converge, int32_t *tries_out)
{
    const int n = d->n, m = d->m, N = d->N;
    double *ps = (double *)malloc(sizeof(double) * N * n);
    double *pu = (double *)malloc(sizeof(double) * (N - 1) * m);
    memcpy(ps, states, sizeof(double) * N * n);
    memcpy(pu, inputs, sizeof(double) * (N - 1) * m);
    double alpha = 1.0;
    int cnt = 50, idx = 0, accepted = -1, tries = 0;
    while (cnt >= 0) {
        cnt -= 1;
        ++tries;
        for (int i = 0; i < N - 1; ++i) {
            double dx[ORC_MAXN], fb[ORC_MAXM];
            for (int j = 0; j < n; ++j) dx[j] = states[i * n + j] - ps[i * n + j];
            matmul(K + i * m * n, dx, fb, m, n, 1);
            for (int j = 0; j < m; ++j) /* ilqr.py:56-57 */
                inputs[i * m + j] = (inputs[i * m + j] + alpha * k[i * m + j]) + fb[j];
            if (d->sys_control_limited)
                for (int j = 0; j < m; ++j) { /* ilqr.py:58-61: min(max(u, lo), hi) */
                    double v = inputs[i * m + j];
                    v = (d->lo[j] > v) ? d->lo[j] : v;      /* python max(v, lo): lo if lo > v */
                    v = (d->hi[j] < v) ? d->hi[j] : v;      /* python min(v, hi): hi if hi < v */
                    inputs[i * m + j] = v;
                }
            model_f(d, i, states + i * n, inputs + i * m, states + (i + 1) * n);
        }
        double c = orc_cost(d, target, states, inputs, centers, radius);
        if (c < *min_cost) {
            *min_cost = c;
            accepted = idx;
            break;
        }
        memcpy(states, ps, sizeof(double) * N * n);
        memcpy(inputs, pu, sizeof(double) * (N - 1) * m);
        if (alpha < d->alpha_floor) {
            *converge = 1;
            break;
        }
        alpha /= 2.0;
        ++idx;
    }
    if (tries_out) *tries_out = tries;
    free(ps);
    free(pu);
    return accepted;
}

/* ---- __call__ (ilqr.py:107-114) ------------------------------------------ */
/* One problem.  All history pointers may be NULL.
 *   hist_cost[it]   min_cost after iteration it
 *   hist_alpha[it]  accepted alpha index, -1 if none (problem converged)
 *   hist_tries[it]  number of rollouts the serial line search performed
 *   hist_states / hist_inputs [it][N][n] / [it][N-1][m] snapshots after it
 *   K_first/k_first gains of the first backward pass
 * Returns the number of iterations executed, or -1 on a singular Quu. */
int ilqr_oracle_solve(const ilqr_oracle_desc *d, const double *target, double *states,
                      double *inputs, const double *centers, const double *radius,
                      double *min_cost, int32_t *converge, double *cost0, double *hist_cost,
                      int32_t *hist_alpha, int32_t *hist_tries, double *hist_states,
                      double *hist_inputs, double *K_first, double *k_first, double *K_last,
                      double *k_last)
{
    const int n = d->n, m = d->m, N = d->N;
    double *K = (double *)malloc(sizeof(double) * (N - 1) * m * n);
    double *k = (double *)malloc(sizeof(double) * (N - 1) * m);
    int iters = 0;
    *min_cost = orc_cost(d, target, states, inputs, centers, radius);
    if (cost0) *cost0 = *min_cost;
    for (int it = 0; it < d->max_iter; ++it) {
        if (*converge) break;
        if (orc_backward(d, target, states, inputs, centers, radius, K, k) != 0) { iters = -1; break; }
        if (it == 0 && K_first) {
            memcpy(K_first, K, sizeof(double) * (N - 1) * m * n);
            memcpy(k_first, k, sizeof(double) * (N - 1) * m);
        }
        int32_t tries = 0;
        int acc = orc_forward(d, target, states, inputs, centers, radius, K, k, min_cost, converge, &tries);
        if (hist_cost) hist_cost[it] = *min_cost;
        if (hist_alpha) hist_alpha[it] = acc;
        if (hist_tries) hist_tries[it] = tries;
        if (hist_states) memcpy(hist_states + (size_t)it * N * n, states, sizeof(double) * N * n);
        if (hist_inputs) memcpy(hist_inputs + (size_t)it * (N - 1) * m, inputs, sizeof(double) * (N - 1) * m);
        ++iters;
    }
    if (K_last && iters > 0) {
        memcpy(K_last, K, sizeof(double) * (N - 1) * m * n);
        memcpy(k_last, k, sizeof(double) * (N - 1) * m);
    }
    free(K);
    free(k);
    return iters;
}

/* Batch driver: B independent problems in the reference's array layout
 * (target [B][N][n], states [B][N][n], inputs [B][N-1][m]), OpenMP over
 * problems.  hist_* are [B][max_iter] (NULL allowed).  iters_out[b] as above. */
void ilqr_oracle_solve_batch(const ilqr_oracle_desc *d, int32_t B, const double *target,
                             double *states, double *inputs, const double *centers,
                             const double *radius, double *min_cost, int32_t *converge,
                             int32_t *iters_out, double *cost0, double *hist_cost,
                             int32_t *hist_alpha, int32_t *hist_tries, int32_t nthreads)
{
    const size_t n = d->n, m = d->m, N = d->N, mi = d->max_iter;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
    for (int32_t b = 0; b < B; ++b) {
        iters_out[b] = ilqr_oracle_solve(
            d, target + b * N * n, states + b * N * n, inputs + b * (N - 1) * m,
            centers ? centers + b * N * 2 : NULL, radius ? radius + b * N : NULL, min_cost + b,
            converge + b, cost0 ? cost0 + b : NULL, hist_cost ? hist_cost + b * mi : NULL,
            hist_alpha ? hist_alpha + b * mi : NULL, hist_tries ? hist_tries + b * mi : NULL, NULL,
            NULL, NULL, NULL, NULL, NULL);
    }
}

/* barrier value/gradient/Hessian at one state, exported for finite-difference tests */
double orc_barrier(const ilqr_oracle_desc *d, const double *x, const double *c, double r,
                   double *grad, double *hess)
{
    return barrier_terms(d, x, c, r, grad, hess);
}

int32_t ilqr_oracle_desc_size(void) { return (int32_t)sizeof(ilqr_oracle_desc); }

/* exported for the brute-force test of the box-QP decision */
int boxqp_all_active_export(const double *P, const double *q, const double *l, const double *u,
                            int m, double lambda, double *du)
{
    return boxqp_all_active(P, q, l, u, m, lambda, du);
}
