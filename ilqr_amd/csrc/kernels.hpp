// Batched iLQR kernels for gfx950 (MI355X): one launch per solver phase, each
// covering the whole horizon for every problem of the batch.
//
//   cost_kernel      ilqr.py:36-46   initial cost, one lane per problem
//   backward_kernel  ilqr.py:76-105  fused linearise + quadratise + Riccati sweep
//                    (cilqr.py:112-134 box-QP branch, barrier terms) one lane per problem
//   evaluate_kernel  ilqr.py:53-74   every line-search step size rolled out in parallel:
//                    L lanes of a wave share one problem, each lane carries A step sizes
//   commit_kernel    ilqr.py:56-63   the winning step size rolled out again, in place
//
// All trajectory arrays are structure-of-arrays with the batch index innermost,
// so lane b of a wave touches address base + b: every global access of the
// lane-per-problem kernels is a full-width coalesced row.  The per-problem
// matrices (n <= 12, m <= 4) live in registers; the time loop is sequential
// and software-prefetches the next steps' rows while the current step computes.
//
// Arithmetic is written with explicit fma() and the file is compiled with
// -ffp-contract=off: the evaluate and the commit kernel then execute the same
// rounding sequence, so the committed trajectory has exactly the cost the
// line search accepted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "fast_trig.hpp"

namespace ilqr {

// per-problem status bits (same values as ILQR_STATUS_* of the C header)
constexpr int ILQR_STATUS_NONPD_ = 1, ILQR_STATUS_SINGULAR_ = 2, ILQR_STATUS_NONFINITE_ = 4;

// ---------------------------------------------------------------- scalars
template <typename T> __device__ __forceinline__ T fma_(T a, T b, T c);
template <> __device__ __forceinline__ double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// fp64 sine and cosine of one angle: FAST = the branch-free routine of fast_trig.hpp (valid for
// |x| < 2^20, reports anything else in `bad`), else the library routine.  A kernel that sees `bad`
// anywhere in a wave redoes its work with FAST = false.
template <bool FAST> __device__ __forceinline__ void sincos_(double x, double &s, double &c, bool &bad) {
    if (!FAST) {
        sincos(x, &s, &c);
        return;
    }
    const bool ok = fast_sincos(x, s, c); // (evaluated whatever `bad` holds: no branch)
    bad = bad || !ok;
}
template <bool FAST> __device__ __forceinline__ void sincos_(float x, float &s, float &c, bool &) { sincosf(x, &s, &c); }
__device__ __forceinline__ double log_(double x) { return log(x); }
__device__ __forceinline__ float log_(float x) { return logf(x); }
__device__ __forceinline__ double fabs_(double x) { return fabs(x); }
__device__ __forceinline__ float fabs_(float x) { return fabsf(x); }

template <typename T> __device__ __forceinline__ T pow2_neg(int a);
template <> __device__ __forceinline__ double pow2_neg<double>(int a) { return __builtin_ldexp(1.0, -a); }
template <> __device__ __forceinline__ float pow2_neg<float>(int a) { return __builtin_ldexpf(1.0f, -a); }

template <typename T> __device__ __forceinline__ T inf_();
template <> __device__ __forceinline__ double inf_<double>() { return __builtin_huge_val(); }
template <> __device__ __forceinline__ float inf_<float>() { return __builtin_huge_valf(); }

// ---------------------------------------------------------------- layout
// Trajectory arrays are tiled: [tile = b / 64][row][lane = b % 64], one tile per
// 64 consecutive problems, rows = time step x component.  A wave therefore
// streams ONE contiguous region per array over the whole horizon, addressed by a
// single base pointer plus small constant offsets, and every access is a full
// 512-byte (fp64) row.
constexpr int TILE = 64;
#ifndef LTV_SYNC_EVERY
#define LTV_SYNC_EVERY 4
#endif
#ifndef EVAL_MIN_WAVES
#define EVAL_MIN_WAVES 1
#endif

// Time-parallel commit (small batches): the line search keeps the state of every candidate at the
// start of every segment of COMMIT_SEG time steps (Params::CK), so the accepted candidate can be
// rolled out again segment by segment, all segments at once (commit_segments_kernel).  Measured at
// N = 200: 8 / 16 / 32 steps per segment -> 7.6 / 7.4 / 7.2 ms per solve of 4096 problems and
// 10.7 / 9.95 / 9.7 ms of 8192 (the checkpoints are write traffic of the line search; the commit
// itself runs at memory speed from 8 steps per segment on).
#ifndef ILQR_COMMIT_SEG
#define ILQR_COMMIT_SEG 32
#endif
constexpr int COMMIT_SEG = ILQR_COMMIT_SEG;
// Sparse states: a commit may keep only every XCK-th row of X (see Params::lazy)
constexpr int XCK = 8;
constexpr int SLOW_TRIG_REGEN = 8; // bit of Params::slow_trig: rows 0, XCK, 2 XCK .. of X and U regenerate the trajectory
constexpr int CK_CANDIDATES = 16; // candidate slots per segment start
template <typename T> __device__ __forceinline__ T *tile_ptr(T *base, int tile, int rows, int lane) {
    return base + (size_t)tile * (size_t)rows * TILE + lane;
}
// The gain arrays K and k of the lane-per-problem models (n <= 8) interleave V consecutive rows per
// problem: [tile][row / V][64][V], i.e. V values of one problem side by side (V = 4 or 2 where the
// row count per time step allows, K: the n entries of a gain row, k: the m entries of a step).  A
// lane then moves 16 or 32 contiguous bytes where it moved V separate 8-byte pieces, and whoever
// reads the gains of a single problem (the second line-search stage, the retirement of a working
// set) touches a V-th of the 128-byte lines.  V = 1 is the plain tiled layout; the n = 12 kernels,
// whose rows travel as whole lines of 16 problems, keep it.
constexpr int gain_vec_of(int per_step, int nx) { return nx > 8 ? 1 : per_step % 4 == 0 ? 4 : per_step % 2 == 0 ? 2 : 1; }
template <class M> struct GainLayout {
    static constexpr int VK = gain_vec_of(M::NX, M::NX); // K: rows (t, i, j), j innermost
    static constexpr int Vk = gain_vec_of(M::NU, M::NX); // k: rows (t, i)
};
// The targets Xref interleave the same way (the n components of a time step side by side), except
// when the constraint rows of constraints.py are bound (barrier mode, sequential QP), whose kernels
// read the plain layout.
template <class M, bool BARRIER> struct TargetLayout {
    static constexpr int V = BARRIER ? 1 : gain_vec_of(M::NX, M::NX);
};
template <class M, bool BARRIER> __device__ __forceinline__ constexpr size_t target_x(int t, int i) {
    constexpr int V = TargetLayout<M, BARRIER>::V;
    return (size_t)(t * (M::NX / V) + i / V) * (size_t)(TILE * V) + (size_t)(i % V);
}
// offsets from a problem's pointer (tile_ptr_v): K[t][i][j] and k[t][i] with ti = t * NU + i (written so
// that the group index needs no division of a run-time value), and the flat row r of an array with V
// interleaved rows
template <class M> __device__ __forceinline__ constexpr size_t gain_K(int ti, int j) {
    constexpr int V = GainLayout<M>::VK;
    return (size_t)(ti * (M::NX / V) + j / V) * (size_t)(TILE * V) + (size_t)(j % V);
}
template <class M> __device__ __forceinline__ constexpr size_t gain_k(int t, int i) {
    constexpr int V = GainLayout<M>::Vk;
    return (size_t)(t * (M::NU / V) + i / V) * (size_t)(TILE * V) + (size_t)(i % V);
}
template <int V> __device__ __forceinline__ constexpr size_t vrow(int r) { return (size_t)(r / V) * (size_t)(TILE * V) + (size_t)(r % V); }
template <int V, typename T> __device__ __forceinline__ T *tile_ptr_v(T *base, int tile, int rows, int lane) {
    return base + (size_t)tile * (size_t)rows * TILE + (size_t)lane * V;
}

// ---------------------------------------------------------------- parameters
template <class P> __device__ __forceinline__ int slot_count(const P &p) { return p.nslots ? *p.nslots : p.B; }
template <class P> __device__ __forceinline__ int orig_of(const P &p, int slot) { return p.slot2orig ? p.slot2orig[slot] : slot; }

// Passed by value as the kernel argument.  Q2/R2/Qf2 are 2Q, 2R, 2Qf (exact
// scalings) used by the quadratisation (ilqr.py:79-92).
template <typename T, int NX, int NU> struct Params {
    T *X, *U;
    const T *Xref;
    T *K, *k;
    T *J, *Jtry;
    int32_t *converged, *iters, *alpha_idx, *status;
    int32_t *slow_trig; // per problem: which sine/cosine its line search and commits used (bits: see commit_kernel)
    // Active-set compaction (ilqr_solve only): the tiled trajectory arrays then hold `*nslots`
    // problems in slot order and slot2orig[slot] is the problem's index in the per-problem arrays
    // (J, converged, iters, alpha_idx, status, Jtry, slow_trig, history).  Both null otherwise.
    const int32_t *slot2orig;
    const int32_t *nslots;
    // line-search stage two: slots that found no winner in stage one (the host zeroes the count
    // before stage one, the stage-one kernel appends)
    int32_t *list;
    int32_t *list_count;
    const int32_t *carried; // per problem: was ever copied into a working set (select_converged only)
    // backward pass over the caller's buffers at the end of a compacted solve (select_converged): the
    // gains of the problems that never converged rest in the last working set (gK, gk, tiled by slot;
    // gslot[problem] = slot) and are copied into the rows this pass stores anyway.  Null = off.
    const T *gK, *gk;
    const int32_t *gslot;
    int32_t *redo; // per problem: the backward pass must be repeated with the pivoted-LU solve (n = 12 model)
    T *trash; // 64 values per tile: where lanes that must not change their trajectory store (commit, REROLL)
    // Sine/cosine cache (null = off): tiled, rows = (N-1) * 2 (sine, cosine of the model's shared angle
    // at the stored state of every time step).  Written by every commit for the problems it rolls
    // out; valid for the problems that are still iterating once a step has been committed
    // (the condition under which the forward kernels also roll the previous states out again), and
    // then read by the backward pass and the line search instead of evaluating the functions again:
    // a third of all instructions of a Car iteration.  Costs 16 bytes of traffic per step and pass,
    // so it pays where the passes are bound by their instruction stream (small batches).
    T *SC;
    // Candidate checkpoints (null = off): [tile][segment][candidate][NX][64], the state of every
    // line-search candidate at time step COMMIT_SEG * segment, segment = 0 .. ceil((N-1)/COMMIT_SEG) - 1.
    // Written by the (single-stage) line search, read by commit_segments_kernel.
    T *CK;
    const T *centers, *radius;
    const T *ltvA, *ltvB;
    T *hist_cost;
    int32_t *hist_alpha;
    int32_t hist_cap;
    int32_t hist_stride; // problems per history row (the whole batch, also when this launch covers a part)
    int32_t B, N;
    int32_t n_alphas;
    int32_t slots_lo, slots_hi; // evaluate_kernel runs only if slots_lo < number of slots <= slots_hi
    int32_t select_converged; // backward pass only: work on the converged problems instead of the active ones
    // Sparse states (0 = off).  Between a commit and the next backward pass the states are only ever
    // needed in reverse order, and every state is the model step of its predecessor under the stored
    // input, by the very instruction sequence of the commit.  With `lazy` a commit that rolls a problem
    // out with the fast sine/cosine marks it (SLOW_TRIG_REGEN in slow_trig) and, in its sparse variant,
    // stores only the state rows 0, XCK, 2 XCK ..; the backward pass rolls each segment of XCK steps
    // out again from its first row before it walks it backwards (the sine/cosine pair of that step is
    // the one the Jacobians need, so the arithmetic is what it was), and states_kernel fills in the
    // rows in between when somebody wants to read X.  Same bits, 5600 + 5600 of the 60 600 bytes the
    // two passes move per Car problem (N = 200) less.
    int32_t lazy;
    T dt, lm, lambda, mu, delta, vlo, vhi;
    T Q[NX * NX], R[NU * NU], Qf[NX * NX];
    T Q2[NX * NX], R2[NU * NU], Qf2[NX * NX];
    T lo[NU], hi[NU];   // rollout clamp; -inf/+inf when the system has no control limits
    T qlo[NU], qhi[NU]; // box of the per-step QP of the constrained solver
};

// ---------------------------------------------------------------- models
// Each model states the discrete dynamics and its Jacobians with the operation
// order of the reference (x + rate*dt) and declares which Jacobian entries can
// be non-zero, so the generic Riccati code skips structural zeros at compile time.

// A model whose step() and jac() need the sine and cosine of ONE angle of (x, u) may say so
// (NT = 1, trig_arg, step_sc, jac_sc): the kernels can then keep the pair of every time step of the
// committed trajectory in memory and hand it back instead of evaluating it again (see Params::SC).
template <typename T> struct CarModel { // systems.py:27-57
    static constexpr int NX = 4, NU = 2;
    static constexpr int NT = 1;
    static constexpr bool HAS_BARRIER = true; // state layout of constraints.py:12 (px, py, v, .)
    static constexpr bool fx_nz(int i, int j) { return i == j || (i < 2 && j >= 2); }
    static constexpr bool fx_one(int i, int j) { return i == j; }
    static constexpr bool fu_nz(int i, int j) { return (i == 2 && j == 0) || (i == 3 && j == 1); }
    static __device__ __forceinline__ T trig_arg(const T (&x)[NX], const T (&)[NU]) { return x[3]; }
    template <class P>
    static __device__ __forceinline__ void step_sc(const P &p, int, const T (&x)[NX], const T (&u)[NU], T (&xn)[NX], T s, T c) {
        xn[0] = fma_(x[2] * c, p.dt, x[0]);
        xn[1] = fma_(x[2] * s, p.dt, x[1]);
        xn[2] = fma_(u[0], p.dt, x[2]);
        xn[3] = fma_(u[1], p.dt, x[3]);
    }
    template <class P>
    static __device__ __forceinline__ void jac_sc(const P &p, int, const T (&x)[NX], const T (&)[NU], T (&fx)[NX][NX], T (&fu)[NX][NU], T s, T c) {
        fx[0][2] = c * p.dt;
        fx[0][3] = (-s * x[2]) * p.dt;
        fx[1][2] = s * p.dt;
        fx[1][3] = (c * x[2]) * p.dt;
        fu[2][0] = p.dt;
        fu[3][1] = p.dt;
    }
    template <bool FAST, class P>
    static __device__ __forceinline__ void step(const P &p, int t, const T (&x)[NX], const T (&u)[NU], T (&xn)[NX], bool &bad) {
        T s, c;
        sincos_<FAST>(x[3], s, c, bad);
        step_sc(p, t, x, u, xn, s, c);
    }
    template <bool FAST, class P>
    static __device__ __forceinline__ void jac(const P &p, int t, const T (&x)[NX], const T (&u)[NU], T (&fx)[NX][NX], T (&fu)[NX][NU], bool &bad) {
        T s, c;
        sincos_<FAST>(x[3], s, c, bad);
        jac_sc(p, t, x, u, fx, fu, s, c);
    }
};

template <typename T> struct CarAccModel { // systems.py:59-90
    static constexpr int NX = 5, NU = 2;
    static constexpr int NT = 1;
    static constexpr bool HAS_BARRIER = false;
    static constexpr bool fx_nz(int i, int j) { return i == j || (i < 2 && (j == 2 || j == 4)) || (i == 2 && j == 3); }
    static constexpr bool fx_one(int i, int j) { return i == j; }
    static constexpr bool fu_nz(int i, int j) { return (i == 3 && j == 0) || (i == 4 && j == 1); }
    static __device__ __forceinline__ T trig_arg(const T (&x)[NX], const T (&)[NU]) { return x[4]; }
    template <class P>
    static __device__ __forceinline__ void step_sc(const P &p, int, const T (&x)[NX], const T (&u)[NU], T (&xn)[NX], T s, T c) {
        xn[0] = fma_(x[2] * c, p.dt, x[0]);
        xn[1] = fma_(x[2] * s, p.dt, x[1]);
        xn[2] = fma_(x[3], p.dt, x[2]);
        xn[3] = fma_(u[0], p.dt, x[3]);
        xn[4] = fma_(u[1], p.dt, x[4]);
    }
    template <class P>
    static __device__ __forceinline__ void jac_sc(const P &p, int, const T (&x)[NX], const T (&)[NU], T (&fx)[NX][NX], T (&fu)[NX][NU], T s, T c) {
        fx[0][2] = c * p.dt;
        fx[0][4] = (-s * x[2]) * p.dt;
        fx[1][2] = s * p.dt;
        fx[1][4] = (c * x[2]) * p.dt;
        fx[2][3] = p.dt;
        fu[3][0] = p.dt;
        fu[4][1] = p.dt;
    }
    template <bool FAST, class P>
    static __device__ __forceinline__ void step(const P &p, int t, const T (&x)[NX], const T (&u)[NU], T (&xn)[NX], bool &bad) {
        T s, c;
        sincos_<FAST>(x[4], s, c, bad);
        step_sc(p, t, x, u, xn, s, c);
    }
    template <bool FAST, class P>
    static __device__ __forceinline__ void jac(const P &p, int t, const T (&x)[NX], const T (&u)[NU], T (&fx)[NX][NX], T (&fu)[NX][NU], bool &bad) {
        T s, c;
        sincos_<FAST>(x[4], s, c, bad);
        jac_sc(p, t, x, u, fx, fu, s, c);
    }
};

template <typename T> struct DubinsModel { // systems.py:92-121
    static constexpr int NX = 3, NU = 2;
    static constexpr int NT = 1;
    static constexpr bool HAS_BARRIER = false;
    static constexpr bool fx_nz(int i, int j) { return i == j || (i < 2 && j == 2); }
    static constexpr bool fx_one(int i, int j) { return i == j; }
    static constexpr bool fu_nz(int i, int j) { return j == 0 || (i == 2 && j == 1); }
    static __device__ __forceinline__ T trig_arg(const T (&x)[NX], const T (&)[NU]) { return x[2]; }
    template <class P>
    static __device__ __forceinline__ void step_sc(const P &p, int, const T (&x)[NX], const T (&u)[NU], T (&xn)[NX], T s, T c) {
        xn[0] = fma_(u[0] * c, p.dt, x[0]);
        xn[1] = fma_(u[0] * s, p.dt, x[1]);
        xn[2] = fma_(u[0] * u[1], p.dt, x[2]);
    }
    template <class P>
    static __device__ __forceinline__ void jac_sc(const P &p, int, const T (&)[NX], const T (&u)[NU], T (&fx)[NX][NX], T (&fu)[NX][NU], T s, T c) {
        fx[0][2] = (-s * u[0]) * p.dt;
        fx[1][2] = (c * u[0]) * p.dt;
        fu[0][0] = c * p.dt;
        fu[1][0] = s * p.dt;
        fu[2][0] = u[1] * p.dt;
        fu[2][1] = u[0] * p.dt;
    }
    template <bool FAST, class P>
    static __device__ __forceinline__ void step(const P &p, int t, const T (&x)[NX], const T (&u)[NU], T (&xn)[NX], bool &bad) {
        T s, c;
        sincos_<FAST>(x[2], s, c, bad);
        step_sc(p, t, x, u, xn, s, c);
    }
    template <bool FAST, class P>
    static __device__ __forceinline__ void jac(const P &p, int t, const T (&x)[NX], const T (&u)[NU], T (&fx)[NX][NX], T (&fu)[NX][NU], bool &bad) {
        T s, c;
        sincos_<FAST>(x[2], s, c, bad);
        jac_sc(p, t, x, u, fx, fu, s, c);
    }
};

// Table-driven linear time-varying model x+ = A_t x + B_t u (no reference counterpart: the
// reference's model_f(x, u) has no time argument).  A_t [T][NX][NX] and B_t [T][NX][NU] are
// shared by the whole batch, so every lane of a wave reads the same address: the loads are
// scalar and the matrix entries arrive as SGPR operands.
template <typename T, int NX_, int NU_> struct LtvModelT {
    static constexpr int NX = NX_, NU = NU_;
    static constexpr bool HAS_BARRIER = false;
    static constexpr bool fx_nz(int, int) { return true; }
    static constexpr bool fx_one(int, int) { return false; }
    static constexpr bool fu_nz(int, int) { return true; }
    template <bool FAST, class P>
    static __device__ __forceinline__ void step(const P &p, int t, const T (&x)[NX], const T (&u)[NU], T (&xn)[NX], bool &) {
        const T *A = p.ltvA + (size_t)t * NX * NX;
        const T *Bm = p.ltvB + (size_t)t * NX * NU;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            T s = A[i * NX] * x[0];
#pragma unroll
            for (int j = 1; j < NX; ++j) s = fma_(A[i * NX + j], x[j], s);
#pragma unroll
            for (int j = 0; j < NU; ++j) s = fma_(Bm[i * NU + j], u[j], s);
            xn[i] = s;
        }
    }
    template <bool FAST, class P>
    static __device__ __forceinline__ void jac(const P &p, int t, const T (&)[NX], const T (&)[NU], T (&fx)[NX][NX], T (&fu)[NX][NU], bool &) {
        const T *A = p.ltvA + (size_t)t * NX * NX;
        const T *Bm = p.ltvB + (size_t)t * NX * NU;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int j = 0; j < NX; ++j) fx[i][j] = A[i * NX + j];
#pragma unroll
            for (int j = 0; j < NU; ++j) fu[i][j] = Bm[i * NU + j];
        }
    }
};
template <typename T> using Ltv12x4Model = LtvModelT<T, 12, 4>;

// number of sine/cosine pairs a model shares between step() and jac(): its NT, or 0
template <class M, class = void> struct trig_pairs { static constexpr int value = 0; };
template <class M> struct trig_pairs<M, std::void_t<decltype(M::NT)>> { static constexpr int value = M::NT; };

// One model step / Jacobian.  For a model with a shared pair (s, c): GIVEN = the caller supplies the
// pair (read from Params::SC), else it is evaluated here and handed back.  Other models ignore it.
template <bool FAST, bool GIVEN, class M, typename T, class P>
__device__ __forceinline__ void model_step(const P &p, int t, const T (&x)[M::NX], const T (&u)[M::NU], T (&xn)[M::NX], T &s, T &c, bool &bad) {
    if constexpr (trig_pairs<M>::value == 1) {
        if (!GIVEN) sincos_<FAST>(M::trig_arg(x, u), s, c, bad);
        M::step_sc(p, t, x, u, xn, s, c);
    } else {
        M::template step<FAST>(p, t, x, u, xn, bad);
    }
}
template <bool FAST, bool GIVEN, class M, typename T, class P>
__device__ __forceinline__ void model_jac(const P &p, int t, const T (&x)[M::NX], const T (&u)[M::NU], T (&fx)[M::NX][M::NX],
                                          T (&fu)[M::NX][M::NU], T &s, T &c, bool &bad) {
    if constexpr (trig_pairs<M>::value == 1) {
        if (!GIVEN) sincos_<FAST>(M::trig_arg(x, u), s, c, bad);
        M::jac_sc(p, t, x, u, fx, fu, s, c);
    } else {
        M::template jac<FAST>(p, t, x, u, fx, fu, bad);
    }
}

// ---------------------------------------------------------------- cost pieces
// e' W e with W dense row-major, as (e' W) e (ilqr.py:42), or diagonal.
template <typename T, int D, bool DIAG>
__device__ __forceinline__ T quad_(const T *W, const T (&e)[D]) {
    T s = T(0);
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            T w = W[i * D + i] * e[i];
            s = (i == 0) ? w * e[i] : fma_(w, e[i], s);
        }
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            T r = e[0] * W[0 * D + j];
#pragma unroll
            for (int i = 1; i < D; ++i) r = fma_(e[i], W[i * D + j], r);
            s = (j == 0) ? r * e[j] : fma_(r, e[j], s);
        }
    }
    return s;
}

// relaxed log barrier: -log z above delta, C2 quadratic extension below
template <typename T>
__device__ __forceinline__ T relaxed_barrier(T z, T delta, T &d1, T &d2) {
    if (z > delta) {
        T r = T(1) / z;
        d1 = -r;
        d2 = r * r;
        return -log_(z);
    }
    T id = T(1) / delta;
    T w = (z - T(2) * delta) * id;
    d1 = w * id;
    d2 = id * id;
    return fma_(T(0.5), fma_(w, w, T(-1)), -log_(delta));
}

// barrier value at a state (components 0,1 position, 2 speed: constraints.py:20-26);
// optionally accumulates gradient and Hessian into lx / lxx
template <typename T, int NX, bool WITH_DERIV, class P>
__device__ __forceinline__ T barrier_terms(const P &p, const T (&x)[NX], T cx, T cy, T r, T (&lx)[NX], T (&lxx)[NX][NX]) {
    T d1, d2, e1, e2, f1, f2;
    T ex = x[0] - cx, ey = x[1] - cy;
    T h = fma_(ex, ex, fma_(ey, ey, -(r * r)));
    T val = relaxed_barrier(-h, p.delta, d1, d2);
    T v1 = relaxed_barrier(x[2] - p.vlo, p.delta, e1, e2);
    T v2 = relaxed_barrier(p.vhi - x[2], p.delta, f1, f2);
    if (WITH_DERIV) {
        T gx = T(-2) * ex, gy = T(-2) * ey;
        lx[0] = fma_(p.mu * d1, gx, lx[0]);
        lx[1] = fma_(p.mu * d1, gy, lx[1]);
        lxx[0][0] += p.mu * fma_(d2 * gx, gx, T(-2) * d1);
        lxx[0][1] += p.mu * (d2 * gx * gy);
        lxx[1][0] += p.mu * (d2 * gy * gx);
        lxx[1][1] += p.mu * fma_(d2 * gy, gy, T(-2) * d1);
        lx[2] += p.mu * (e1 - f1);
        lxx[2][2] += p.mu * (e2 + f2);
    }
    return p.mu * ((val + v1) + v2);
}

// ---------------------------------------------------------------- cost kernel
template <typename T, class M, bool DIAG, bool BARRIER>
__global__ __launch_bounds__(64) void cost_kernel(Params<T, M::NX, M::NU> p) {
    constexpr int NX = M::NX, NU = M::NU;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= p.B) return;
    constexpr bool BARRIER_ = BARRIER;
    const int tile = b >> 6, ln = b & 63;
    const T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const T *Rt = tile_ptr_v<TargetLayout<M, BARRIER>::V>(p.Xref, tile, p.N * NX, ln);
    const T *Ut = tile_ptr(p.U, tile, (p.N - 1) * NU, ln);
    const T *Cp = BARRIER_ ? tile_ptr(p.centers, tile, p.N * 2, ln) : nullptr;
    const T *Dp = BARRIER_ ? tile_ptr(p.radius, tile, p.N, ln) : nullptr;
    T cost = T(0);
    T dummy_lx[NX], dummy_lxx[NX][NX];
    for (int t = 0; t < p.N - 1; ++t) {
        T x[NX], e[NX], u[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            x[i] = Xt[(t * NX + i) * TILE];
            e[i] = x[i] - Rt[target_x<M, BARRIER>(t, i)];
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = Ut[(t * NU + i) * TILE];
        T stage = quad_<T, NX, DIAG>(p.Q, e) + quad_<T, NU, DIAG>(p.R, u);
        if (BARRIER && t > 0)
            stage += barrier_terms<T, NX, false>(p, x, Cp[(t * 2 + 0) * TILE],
                                                 Cp[(t * 2 + 1) * TILE],
                                                 Dp[t * TILE], dummy_lx, dummy_lxx);
        cost += stage;
    }
    {
        const int t = p.N - 1;
        T x[NX], e[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            x[i] = Xt[(t * NX + i) * TILE];
            e[i] = x[i] - Rt[target_x<M, BARRIER>(t, i)];
        }
        T term = quad_<T, NX, DIAG>(p.Qf, e);
        if (BARRIER)
            term += barrier_terms<T, NX, false>(p, x, Cp[(t * 2 + 0) * TILE],
                                                Cp[(t * 2 + 1) * TILE],
                                                Dp[t * TILE], dummy_lx, dummy_lxx);
        cost += term;
    }
    p.J[b] = cost;
}

// ---------------------------------------------------------------- Quu solve
// Solves Quu * [k | K] = -[Qu | Qux] for m <= 4 by the square-root-free
// Cholesky factorisation (L D L') of the lower triangle.  If a pivot is not
// positive the matrix is not positive definite: fall back to LU with partial
// pivoting on the full matrix, which is what numpy.linalg.inv does
// (ilqr.py:99), and flag the problem.
template <typename T, int NU, int NR, bool ASSUME_PD>
__device__ __forceinline__ int solve_quu(const T (&A)[NU][NU], T (&Bm)[NU][NR], bool &bad) {
    // Bm holds the right-hand sides on entry (already negated) and the solution on exit
    T L[NU][NU], d[NU], dinv[NU];
    bool pd = true;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        T dj = A[j][j];
#pragma unroll
        for (int q = 0; q < j; ++q) dj = fma_(-(L[j][q] * L[j][q]), d[q], dj);
        pd = pd && (dj > T(0));
        d[j] = dj;
        dinv[j] = T(1) / dj;
#pragma unroll
        for (int i = j + 1; i < NU; ++i) {
            T s = A[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) s = fma_(-(L[i][q] * L[j][q]), d[q], s);
            L[i][j] = s * dinv[j];
        }
    }
    // ASSUME_PD: no branch here (the time step stays one basic block); a pivot that is not
    // positive is reported through `bad` and the caller redoes its sweep with ASSUME_PD = false
    if (ASSUME_PD) bad = bad || !pd;
    if (ASSUME_PD || __builtin_expect(pd, 1)) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            T y[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) { // forward: L y = b
                T s = Bm[i][c];
#pragma unroll
                for (int q = 0; q < i; ++q) s = fma_(-L[i][q], y[q], s);
                y[i] = s;
            }
#pragma unroll
            for (int i = NU - 1; i >= 0; --i) { // backward: L' x = D^-1 y
                T s = y[i] * dinv[i];
#pragma unroll
                for (int q = i + 1; q < NU; ++q) s = fma_(-L[q][i], Bm[q][c], s);
                Bm[i][c] = s;
            }
        }
        return 0;
    }
    // pivoted LU (rare path; divergent lanes only)
    T LU[NU][NU];
    int piv[NU];
    int rc = ILQR_STATUS_NONPD_;
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < NU; ++j) LU[i][j] = A[i][j];
    for (int c = 0; c < NU; ++c) {
        int pr = c;
        T best = fabs_(LU[c][c]);
        for (int r = c + 1; r < NU; ++r)
            if (fabs_(LU[r][c]) > best) { best = fabs_(LU[r][c]); pr = r; }
        piv[c] = pr;
        if (best == T(0)) rc |= ILQR_STATUS_SINGULAR_;
        if (pr != c) {
            for (int j = 0; j < NU; ++j) { T t = LU[c][j]; LU[c][j] = LU[pr][j]; LU[pr][j] = t; }
            for (int j = 0; j < NR; ++j) { T t = Bm[c][j]; Bm[c][j] = Bm[pr][j]; Bm[pr][j] = t; }
        }
        T inv = T(1) / LU[c][c];
        for (int r = c + 1; r < NU; ++r) {
            LU[r][c] *= inv;
            for (int j = c + 1; j < NU; ++j) LU[r][j] = fma_(-LU[r][c], LU[c][j], LU[r][j]);
            for (int j = 0; j < NR; ++j) Bm[r][j] = fma_(-LU[r][c], Bm[c][j], Bm[r][j]);
        }
    }
    (void)piv;
    for (int j = 0; j < NR; ++j)
        for (int i = NU - 1; i >= 0; --i) {
            T s = Bm[i][j];
            for (int q = i + 1; q < NU; ++q) s = fma_(-LU[i][q], Bm[q][j], s);
            Bm[i][j] = s / LU[i][i];
        }
    return rc;
}

// ---------------------------------------------------------------- backward
// One Riccati step: consumes (x, u, xref) of step t and the value function of
// step t+1, produces the gains of step t and the value function of step t.
// TC: the sine/cosine pair of the step (sc_s, sc_c) comes from the cache instead of being evaluated.
template <typename T, class M, bool DIAG, int CONSTR, bool FAST, bool TC = false, class P>
__device__ __forceinline__ int riccati_step(const P &p, int t, const T (&x)[M::NX], const T (&u)[M::NU], const T (&xr)[M::NX],
                                            T bcx, T bcy, T brad, T (&Vx)[M::NX], T (&Vxx)[M::NX][M::NX],
                                            T (&Kt)[M::NU][M::NX], T (&kt)[M::NU], bool &bad, T sc_s = T(0), T sc_c = T(0)) {
    constexpr int NX = M::NX, NU = M::NU;
    T e[NX], lx[NX], lu[NU], lxx[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) e[i] = x[i] - xr[i];
    // l_x = 2Q(x - xref), l_u = 2Ru, l_xx = 2Q, l_uu = 2R (ilqr.py:83-84, :91-92)
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        if (DIAG) {
            lx[i] = p.Q2[i * NX + i] * e[i];
        } else {
            T s = p.Q2[i * NX] * e[0];
#pragma unroll
            for (int j = 1; j < NX; ++j) s = fma_(p.Q2[i * NX + j], e[j], s);
            lx[i] = s;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) lxx[i][j] = (DIAG && i != j) ? T(0) : p.Q2[i * NX + j];
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        if (DIAG) {
            lu[i] = p.R2[i * NU + i] * u[i];
        } else {
            T s = p.R2[i * NU] * u[0];
#pragma unroll
            for (int j = 1; j < NU; ++j) s = fma_(p.R2[i * NU + j], u[j], s);
            lu[i] = s;
        }
    }
    if ((CONSTR & 2) && t > 0) barrier_terms<T, NX, true>(p, x, bcx, bcy, brad, lx, lxx);

    T fx[NX][NX], fu[NX][NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j = 0; j < NX; ++j) fx[i][j] = M::fx_one(i, j) ? T(1) : T(0);
#pragma unroll
        for (int j = 0; j < NU; ++j) fu[i][j] = T(0);
    }
    model_jac<FAST, TC, M>(p, t, x, u, fx, fu, sc_s, sc_c, bad);

    // Qx = l_x + fx' Vx ; Qu = l_u + fu' Vx (ilqr.py:93-94)
    T Qx[NX], Qu[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        T s = lx[i];
#pragma unroll
        for (int q = 0; q < NX; ++q)
            if (M::fx_nz(q, i)) s = M::fx_one(q, i) ? s + Vx[q] : fma_(fx[q][i], Vx[q], s);
        Qx[i] = s;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        T s = lu[i];
#pragma unroll
        for (int q = 0; q < NX; ++q)
            if (M::fu_nz(q, i)) s = fma_(fu[q][i], Vx[q], s);
        Qu[i] = s;
    }
    // Vxx + lm I (ilqr.py:95)
#pragma unroll
    for (int i = 0; i < NX; ++i) Vxx[i][i] += p.lm;
    // Mx = fx' Vxx ; Qxx = l_xx + Mx fx (ilqr.py:96)
    T Mx[NX][NX], Qxx[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T s = T(0);
            bool have = false;
#pragma unroll
            for (int q = 0; q < NX; ++q)
                if (M::fx_nz(q, i)) {
                    s = !have ? (M::fx_one(q, i) ? Vxx[q][j] : fx[q][i] * Vxx[q][j])
                              : (M::fx_one(q, i) ? s + Vxx[q][j] : fma_(fx[q][i], Vxx[q][j], s));
                    have = true;
                }
            Mx[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T s = lxx[i][j];
#pragma unroll
            for (int q = 0; q < NX; ++q)
                if (M::fx_nz(q, j)) s = M::fx_one(q, j) ? s + Mx[i][q] : fma_(Mx[i][q], fx[q][j], s);
            Qxx[i][j] = s;
        }
    // Mu = fu' Vxx ; Quu = l_uu + Mu fu ; Qux = Mu fx (ilqr.py:97-98)
    T Mu[NU][NX], Quu[NU][NU], Qux[NU][NX];
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T s = T(0);
            bool have = false;
#pragma unroll
            for (int q = 0; q < NX; ++q)
                if (M::fu_nz(q, i)) {
                    s = !have ? fu[q][i] * Vxx[q][j] : fma_(fu[q][i], Vxx[q][j], s);
                    have = true;
                }
            Mu[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            T s = (DIAG && i != j) ? T(0) : p.R2[i * NU + j];
#pragma unroll
            for (int q = 0; q < NX; ++q)
                if (M::fu_nz(q, j)) s = fma_(Mu[i][q], fu[q][j], s);
            Quu[i][j] = s;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T s = T(0);
            bool have = false;
#pragma unroll
            for (int q = 0; q < NX; ++q)
                if (M::fx_nz(q, j)) {
                    s = !have ? (M::fx_one(q, j) ? Mu[i][q] : Mu[i][q] * fx[q][j])
                              : (M::fx_one(q, j) ? s + Mu[i][q] : fma_(Mu[i][q], fx[q][j], s));
                    have = true;
                }
            Qux[i][j] = s;
        }
    }
    // k = -Quu^-1 Qu ; K = -Quu^-1 Qux (ilqr.py:99-101)
    T rhs[NU][NX + 1];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        rhs[i][0] = -Qu[i];
#pragma unroll
        for (int j = 0; j < NX; ++j) rhs[i][j + 1] = -Qux[i][j];
    }
    int st = solve_quu<T, NU, NX + 1, FAST>(Quu, rhs, bad);
    if (CONSTR & 1) {
        // cilqr.py:112-134.  min 1/2 du'Quu du + (2 Qu)'du, lo-u <= du <= hi-u.  The result is
        // used only when every multiplier exceeds lambda in magnitude, i.e. the minimiser is a
        // corner of the box with all bounds strictly active; a corner c is the minimiser iff
        // y = -(Quu c + 2Qu) is < 0 on lower and > 0 on upper bounds.
        bool found = false;
        T du[NU];
#pragma unroll
        for (int mask = 0; mask < (1 << NU); ++mask) {
            T c[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) c[j] = ((mask >> j) & 1) ? (p.qhi[j] - u[j]) : (p.qlo[j] - u[j]);
            bool ok = true;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                T g = T(2) * Qu[i];
#pragma unroll
                for (int j = 0; j < NU; ++j) g = fma_(Quu[i][j], c[j], g);
                T y = -g;
                ok = ok && (((mask >> i) & 1) ? (y > p.lambda) : (y < -p.lambda));
            }
            if (ok && !found) {
                found = true;
#pragma unroll
                for (int j = 0; j < NU; ++j) du[j] = c[j];
            }
        }
        if (found) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                rhs[i][0] = du[i];
#pragma unroll
                for (int j = 0; j < NX; ++j) rhs[i][j + 1] = T(0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        kt[i] = rhs[i][0];
#pragma unroll
        for (int j = 0; j < NX; ++j) Kt[i][j] = rhs[i][j + 1];
    }
    // Vx = Qx + K' Qu ; Vxx = Qxx + K' Qux (ilqr.py:104-105; not symmetrised)
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        T s = Qx[i];
#pragma unroll
        for (int q = 0; q < NU; ++q) s = fma_(Kt[q][i], Qu[q], s);
        Vx[i] = s;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            T w = Qxx[i][j];
#pragma unroll
            for (int q = 0; q < NU; ++q) w = fma_(Kt[q][i], Qux[q][j], w);
            Vxx[i][j] = w;
        }
    }
    return st;
}

template <typename T, int NX, int NU, bool G = false> struct StepRow { // the rows of one time step the backward pass reads
    T x[NX], u[NU], xr[NX], cx, cy, rad, s, c;
    T gK[G ? NU : 1][G ? NX : 1], gk[G ? NU : 1]; // G: gains to be passed on instead of the computed ones (see Params::gK)
};

// One reverse sweep over the horizon for the problem of this lane; returns status bits and
// sets `bad` if the fast trigonometry met an argument outside its range.
// `store`: whether this lane writes its gains (only looked at when !FAST: the first, fast sweep
// stores from every lane, see below; the redo sweep must leave the gains of the other lanes alone)
// TC: sine/cosine pairs are read from SCp (Params::SC) and `store` also guards the fast sweep (the
// lanes of converged problems may hold no valid pairs, so they must not rewrite their gains).
// COND: also the fast sweep stores only for the lanes with `store`.
// GATHER: the lanes with `take` store the gains found at Kg, kg (a problem's rows in another tiling) instead
// of the ones they compute; the other lanes' Kg, kg point at any valid rows.
template <typename T, class M, bool DIAG, int CONSTR, int PF, bool FAST, bool TC = false, bool COND = false, bool GATHER = false, class P>
__device__ __forceinline__ int backward_sweep(const P &p, const T *Xt, const T *Rt, const T *Ut, const T *Cp, const T *Dp,
                                              T *Kp, T *kp, bool &bad, bool store = true, const T *SCp = nullptr,
                                              const T *Kg = nullptr, const T *kg = nullptr, bool take = false) {
    constexpr int NX = M::NX, NU = M::NU;
    const int T1 = p.N - 1;
    auto load_row = [&](int t, StepRow<T, NX, NU, GATHER> &r) {
        if constexpr (GATHER) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                r.gk[i] = kg[gain_k<M>(t, i)];
#pragma unroll
                for (int j = 0; j < NX; ++j) r.gK[i][j] = Kg[gain_K<M>(t * NU + i, j)];
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            r.x[i] = Xt[(t * NX + i) * TILE];
            r.xr[i] = Rt[target_x<M, (CONSTR & 2) != 0>(t, i)];
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) r.u[i] = Ut[(t * NU + i) * TILE];
        if (CONSTR & 2) {
            r.cx = Cp[(t * 2 + 0) * TILE];
            r.cy = Cp[(t * 2 + 1) * TILE];
            r.rad = Dp[t * TILE];
        } else {
            r.cx = r.cy = r.rad = T(0);
        }
        if (TC) {
            r.s = SCp[(t * 2 + 0) * TILE];
            r.c = SCp[(t * 2 + 1) * TILE];
        } else {
            r.s = r.c = T(0);
        }
    };

    // terminal condition (ilqr.py:79-82)
    T Vx[NX], Vxx[NX][NX];
    {
        T x[NX], e[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            x[i] = Xt[(T1 * NX + i) * TILE];
            e[i] = x[i] - Rt[target_x<M, (CONSTR & 2) != 0>(T1, i)];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (DIAG) {
                Vx[i] = p.Qf2[i * NX + i] * e[i];
            } else {
                T s = p.Qf2[i * NX] * e[0];
#pragma unroll
                for (int j = 1; j < NX; ++j) s = fma_(p.Qf2[i * NX + j], e[j], s);
                Vx[i] = s;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) Vxx[i][j] = (DIAG && i != j) ? T(0) : p.Qf2[i * NX + j];
        }
        if (CONSTR & 2)
            barrier_terms<T, NX, true>(p, x, Cp[(T1 * 2 + 0) * TILE], Cp[(T1 * 2 + 1) * TILE], Dp[T1 * TILE], Vx, Vxx);
    }

    StepRow<T, NX, NU, GATHER> ring[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (T1 - 1 - s >= 0) load_row(T1 - 1 - s, ring[s]);
    int status = 0;
    for (int tb = T1 - 1; tb >= 0; tb -= PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int t = tb - s;
            if (t < 0) break;
            StepRow<T, NX, NU, GATHER> cur = ring[s];
            if (t - PF >= 0) load_row(t - PF, ring[s]);
            T Kt[NU][NX], kt[NU];
            status |= riccati_step<T, M, DIAG, CONSTR, FAST, TC>(p, t, cur.x, cur.u, cur.xr, cur.cx, cur.cy, cur.rad, Vx, Vxx, Kt, kt, bad,
                                                                 cur.s, cur.c);
            // every lane stores, also those of converged problems (their trajectory has not
            // changed, so they rewrite the same bits) and the padding lanes of the last tile: a
            // branch around the stores would make the number of outstanding memory operations
            // path-dependent and force full drains of the memory queue in the loop
            if ((FAST && !TC && !COND) || store) {
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    if constexpr (GATHER) {
                        kp[gain_k<M>(t, i)] = take ? cur.gk[i] : kt[i];
#pragma unroll
                        for (int j = 0; j < NX; ++j) Kp[gain_K<M>(t * NU + i, j)] = take ? cur.gK[i][j] : Kt[i][j];
                    } else {
                        kp[gain_k<M>(t, i)] = kt[i];
#pragma unroll
                        for (int j = 0; j < NX; ++j) Kp[gain_K<M>(t * NU + i, j)] = Kt[i][j];
                    }
                }
            }
        }
    }
    return status;
}

// The backward pass of the problem in slot b0 (one lane per problem).
template <typename T, class M, bool DIAG, int CONSTR, int PF, bool TC = false, bool GATHER = false, class P>
__device__ __forceinline__ void backward_body(const P &p, int b0) {
    constexpr int NX = M::NX, NU = M::NU;
    const int nsl = slot_count(p);
    if (nsl <= 0) return;
    const bool valid = b0 < nsl;
    const int o = orig_of(p, valid ? b0 : nsl - 1);
    // normally the problems still iterating; at the end of a solve with compaction, the converged ones
    // whose gains were computed inside a working set
    const bool active = valid && (p.select_converged ? (p.converged[o] != 0 && p.carried[o] != 0) : p.converged[o] == 0);
    // (GATHER: the problems that never converged: every one of them is in the last working set)
    const bool take = GATHER && valid && p.converged[o] == 0;
    if (__ballot(active || take) == 0ull) return;
    const int T1 = p.N - 1;
    constexpr bool BARRIER_ = (CONSTR & 2) != 0;
    const int tile = b0 >> 6, ln = b0 & 63; // buffers are padded to whole tiles
    const T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const T *Rt = tile_ptr_v<TargetLayout<M, (CONSTR & 2) != 0>::V>(p.Xref, tile, p.N * NX, ln);
    const T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    const T *Cp = BARRIER_ ? tile_ptr(p.centers, tile, p.N * 2, ln) : nullptr;
    const T *Dp = BARRIER_ ? tile_ptr(p.radius, tile, p.N, ln) : nullptr;
    T *kp = tile_ptr_v<GainLayout<M>::Vk>(p.k, tile, T1 * NU, ln);
    T *Kp = tile_ptr_v<GainLayout<M>::VK>(p.K, tile, T1 * NU * NX, ln);
    const T *SCp = TC ? tile_ptr(p.SC, tile, T1 * 2, ln) : nullptr;
    bool bad = false;
    if constexpr (GATHER) {
        const int gs = take ? p.gslot[o] : 0;
        const T *kg = tile_ptr_v<GainLayout<M>::Vk>(p.gk, gs >> 6, T1 * NU, gs & 63);
        const T *Kg = tile_ptr_v<GainLayout<M>::VK>(p.gK, gs >> 6, T1 * NU * NX, gs & 63);
        int status = backward_sweep<T, M, DIAG, CONSTR, PF, true, false, false, true>(p, Xt, Rt, Ut, Cp, Dp, Kp, kp, bad, true, nullptr, Kg, kg, take);
        // (the lanes that pass gains on computed theirs for nothing: whatever their sweep met does not count)
        const bool redo = bad && valid && !take;
        if (__ballot(redo) != 0ull) {
            bool bad2 = false;
            const int status2 = backward_sweep<T, M, DIAG, CONSTR, 1, false, false>(p, Xt, Rt, Ut, Cp, Dp, Kp, kp, bad2, redo, nullptr);
            status = redo ? status2 : status;
        }
        if (active && status != 0 && p.status) p.status[o] |= status;
        return;
    }
    int status = backward_sweep<T, M, DIAG, CONSTR, PF, true, TC>(p, Xt, Rt, Ut, Cp, Dp, Kp, kp, bad, active, SCp);
    // rare: an angle outside the fast trigonometric range or a Quu that is not positive definite
    // somewhere in the wave: the lanes that raised the flag redo their sweep with the library
    // sine/cosine and the pivoted-LU fallback and replace their gains; the other lanes keep what
    // the first sweep stored, so a problem's gains do not depend on which problems share its wave.
    // (Lanes of converged problems count too: the first sweep has just overwritten their gains.)
    const bool redo = bad && valid && (!TC || active);
    if (__ballot(redo) != 0ull) {
        bool bad2 = false;
        const int status2 = backward_sweep<T, M, DIAG, CONSTR, 1, false, TC>(p, Xt, Rt, Ut, Cp, Dp, Kp, kp, bad2, redo, SCp);
        status = redo ? status2 : status;
    }
    if (active && status != 0 && p.status) p.status[o] |= status;
}

template <typename T, class M, bool DIAG, int CONSTR, int PF, bool TC = false, bool GATHER = false>
__global__ __launch_bounds__(64) void backward_kernel(Params<T, M::NX, M::NU> p) {
    backward_body<T, M, DIAG, CONSTR, PF, TC, GATHER>(p, blockIdx.x * 64 + threadIdx.x);
}

// The reverse sweep for a problem of whose states only the rows 0, XCK, 2 XCK .. are stored
// (Params::lazy): the horizon in segments of XCK steps, last segment first; each is rolled out
// forward from its stored first state under the stored inputs (the commit's instruction sequence, so
// these are the states the commit had in registers), keeping the states and, for a model with a
// shared sine/cosine pair, the pairs, and is then walked backwards by riccati_step.  The rows of the
// segment that comes next are under way meanwhile.  PRED: only the lanes with `store` write their
// gains (else every lane does, see backward_sweep).
template <typename T, class M, bool DIAG, int CONSTR, bool FAST, bool PRED, class P>
__device__ __forceinline__ int backward_sweep_gen(const P &p, const T *Xt, const T *Rt, const T *Ut, T *Kp, T *kp, bool &bad, bool store) {
    constexpr int NX = M::NX, NU = M::NU, S = XCK;
    constexpr bool SHARED = trig_pairs<M>::value == 1;
    static_assert((CONSTR & 2) == 0, "the regenerating sweep is built without the barrier terms");
    const int T1 = p.N - 1;
    const int nseg = (T1 + S - 1) / S;
    struct Head { // the rows of a segment: its first state, its inputs, its targets
        T x0[NX], u[S][NU], xr[S][NX];
    };
    auto load_head = [&](int seg, Head &h) {
        const int t0 = seg * S;
#pragma unroll
        for (int i = 0; i < NX; ++i) h.x0[i] = Xt[(t0 * NX + i) * TILE];
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const int t = t0 + q < T1 ? t0 + q : T1 - 1; // (past the horizon the last row again: no branch around a load)
#pragma unroll
            for (int i = 0; i < NU; ++i) h.u[q][i] = Ut[(t * NU + i) * TILE];
#pragma unroll
            for (int i = 0; i < NX; ++i) h.xr[q][i] = Rt[target_x<M, (CONSTR & 2) != 0>(t, i)];
        }
    };
    T Vx[NX], Vxx[NX][NX];
    Head cur, nxt;
    load_head(nseg - 1, cur);
    int status = 0;
    // one segment; GUARD: it may be shorter than S (only the last one is), the steps past its end are skipped
    auto segment = [&](int seg, auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        const int t0 = seg * S;
        const int len = GUARD ? (T1 - t0 < S ? T1 - t0 : S) : S;
        if (seg > 0) load_head(seg - 1, nxt); // a whole segment ahead: a segment takes a few microseconds, a loaded row about as long
        T x[S + 1][NX], ss[S], cc[S];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[0][i] = cur.x0[i];
#pragma unroll
        for (int q = 0; q < S; ++q) { // (past the end of a short segment: harmless arithmetic on the last row)
            ss[q] = cc[q] = T(0);
            // always the fast routine: the mark says the commit used it (all its angles were in range).
            // The fallback sweep (!FAST) regenerates the same states and lets riccati_step evaluate the
            // Jacobians' pair with the library routine, as the reading sweep does.
            bool b2 = false;
            model_step<true, false, M>(p, t0 + q, x[q], cur.u[q], x[q + 1], ss[q], cc[q], b2);
            if (FAST && (!GUARD || q < len)) bad = bad || b2;
        }
        if (GUARD) {
            // terminal condition (ilqr.py:79-82) at the state after the last step
            T xT[NX], e[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xT[i] = x[S][i];
#pragma unroll
            for (int q = S - 1; q >= 1; --q)
                if (q == len) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) xT[i] = x[q][i];
                }
#pragma unroll
            for (int i = 0; i < NX; ++i) e[i] = xT[i] - Rt[target_x<M, (CONSTR & 2) != 0>(T1, i)];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if (DIAG) {
                    Vx[i] = p.Qf2[i * NX + i] * e[i];
                } else {
                    T s = p.Qf2[i * NX] * e[0];
#pragma unroll
                    for (int j = 1; j < NX; ++j) s = fma_(p.Qf2[i * NX + j], e[j], s);
                    Vx[i] = s;
                }
#pragma unroll
                for (int j = 0; j < NX; ++j) Vxx[i][j] = (DIAG && i != j) ? T(0) : p.Qf2[i * NX + j];
            }
        }
#pragma unroll
        for (int q = S - 1; q >= 0; --q) {
            if (GUARD && q >= len) continue;
            const int t = t0 + q;
            T Kt[NU][NX], kt[NU];
            status |= riccati_step<T, M, DIAG, CONSTR, FAST, (SHARED && FAST)>(p, t, x[q], cur.u[q], cur.xr[q], T(0), T(0), T(0), Vx, Vxx, Kt, kt, bad,
                                                                               ss[q], cc[q]);
            if (!PRED || store) {
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    kp[gain_k<M>(t, i)] = kt[i];
#pragma unroll
                    for (int j = 0; j < NX; ++j) Kp[gain_K<M>(t * NU + i, j)] = Kt[i][j];
                }
            }
        }
        cur = nxt;
    };
    segment(nseg - 1, std::true_type());
    for (int seg = nseg - 2; seg >= 0; --seg) segment(seg, std::false_type());
    return status;
}

// The backward pass of the problem in slot b0 under Params::lazy: problems with the mark are
// regenerated (all of them, normally: one sweep); the others hold every state row in X, which need
// not be a rollout, and take the reading sweep in a second, predicated pass of the wave.
template <typename T, class M, bool DIAG, int CONSTR, int PF, class P>
__device__ __forceinline__ void backward_body_gen(const P &p, int b0) {
    constexpr int NX = M::NX, NU = M::NU;
    const int nsl = slot_count(p);
    if (nsl <= 0) return;
    const bool valid = b0 < nsl;
    const int o = orig_of(p, valid ? b0 : nsl - 1);
    const bool active = valid && p.converged[o] == 0;
    if (__ballot(active) == 0ull) return;
    const int T1 = p.N - 1;
    const int tile = b0 >> 6, ln = b0 & 63; // buffers are padded to whole tiles
    const T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const T *Rt = tile_ptr_v<TargetLayout<M, (CONSTR & 2) != 0>::V>(p.Xref, tile, p.N * NX, ln);
    const T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    T *kp = tile_ptr_v<GainLayout<M>::Vk>(p.k, tile, T1 * NU, ln);
    T *Kp = tile_ptr_v<GainLayout<M>::VK>(p.K, tile, T1 * NU * NX, ln);
    // (the padding lanes of the last tile go with the regenerating sweep: whatever they compute stays in the padding)
    const bool regen = !valid || (p.slow_trig[o] & SLOW_TRIG_REGEN) != 0;
    const bool whole = valid && !regen;
    bool bad = false;
    int status = 0;
    // every lane stores, as in backward_sweep: a converged problem with the mark regenerates the
    // trajectory of its last backward pass and rewrites the same bits; the lanes without the mark
    // write gains of no meaning here and their own in the second pass
    if (__ballot(regen) != 0ull) status = backward_sweep_gen<T, M, DIAG, CONSTR, true, false>(p, Xt, Rt, Ut, Kp, kp, bad, true);
    if (__ballot(whole) != 0ull) {
        bool bad1 = false;
        const int status1 = backward_sweep<T, M, DIAG, CONSTR, PF, true, false, true>(p, Xt, Rt, Ut, nullptr, nullptr, Kp, kp, bad1, whole);
        status = whole ? status1 : status;
        bad = whole ? bad1 : bad;
    }
    // the fallbacks act per problem, as in backward_body
    const bool redo = bad && valid;
    if (__ballot(redo) != 0ull) {
        int status2 = 0;
        bool bad2 = false;
        if (__ballot(redo && regen) != 0ull)
            status2 = backward_sweep_gen<T, M, DIAG, CONSTR, false, true>(p, Xt, Rt, Ut, Kp, kp, bad2, redo && regen);
        if (__ballot(redo && whole) != 0ull) {
            const int status3 = backward_sweep<T, M, DIAG, CONSTR, 1, false>(p, Xt, Rt, Ut, nullptr, nullptr, Kp, kp, bad2, redo && whole);
            status2 = whole ? status3 : status2;
        }
        status = redo ? status2 : status;
    }
    if (active && status != 0 && p.status) p.status[o] |= status;
}

template <typename T, class M, bool DIAG, int CONSTR, int PF>
__global__ __launch_bounds__(64) void backward_gen_kernel(Params<T, M::NX, M::NU> p) {
    backward_body_gen<T, M, DIAG, CONSTR, PF>(p, blockIdx.x * 64 + threadIdx.x);
}

// Fills in the state rows a sparse commit left out: for every problem with the mark, X[t+1] =
// f(X[t], U[t]) from X[0] by the commit's instruction sequence (one lane per problem; the mark stays:
// what it says remains true).
template <typename T, class M> __global__ __launch_bounds__(64) void states_kernel(Params<T, M::NX, M::NU> p) {
    constexpr int NX = M::NX, NU = M::NU, PF = 4;
    const int b0 = blockIdx.x * 64 + threadIdx.x;
    const int nsl = slot_count(p);
    if (nsl <= 0) return;
    const bool valid = b0 < nsl;
    const int o = orig_of(p, valid ? b0 : nsl - 1);
    const bool fill = valid && (p.slow_trig[o] & SLOW_TRIG_REGEN) != 0;
    if (__ballot(fill) == 0ull) return;
    const int T1 = p.N - 1;
    const int tile = b0 >> 6, ln = b0 & 63;
    T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    T x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = Xt[i * TILE];
    T ring[PF][NU];
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int i = 0; i < NU; ++i) ring[s][i] = Ut[((s < T1 ? s : T1 - 1) * NU + i) * TILE];
    bool bad = false;
    for (int tb = 0; tb < T1; tb += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int t = tb + s;
            if (t >= T1) break;
            T u[NU], xn[NX], sn = T(0), cn = T(0);
#pragma unroll
            for (int i = 0; i < NU; ++i) u[i] = ring[s][i];
            const int tn = t + PF < T1 ? t + PF : T1 - 1;
#pragma unroll
            for (int i = 0; i < NU; ++i) ring[s][i] = Ut[(tn * NU + i) * TILE];
            model_step<true, false, M>(p, t, x, u, xn, sn, cn, bad);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if (fill) {
#pragma unroll
                for (int i = 0; i < NX; ++i) Xt[((t + 1) * NX + i) * TILE] = xn[i];
            }
        }
    }
}

// ---------------------------------------------------------------- backward pass, n = 12, m = 4, fp64
// One lane per problem does not work for the 12-state model: Vxx, the products with the
// Jacobians and the Q-terms are a few hundred live fp64 values per lane.  Here one wave owns one
// problem and the matrix products run on the fp64 matrix unit (v_mfma_f64_16x16x4): with
// F = [A_t | B_t] (12 x 16), all second-order Q-terms are one 16 x 16 matrix G = F' (Vxx + lm I) F
// = [[Qxx - lxx, Qxu], [Qux, Quu - luu]] (ilqr.py:95-98).
//
// Register layouts of a 16 x 16 tile M over the wave, lane = 16 g + c:
//   D-layout: register r of lane (c, g) = M[g + 4 r][c]   (what the instruction returns)
//   T-layout: register r of lane (c, g) = M[c][g + 4 r]   (the D-layout of M')
// The instruction's A operand of its k-th call is the T-layout register k of the left factor,
// its B operand the D-layout register k of the right factor, so a result feeds the next
// product directly, and swapping the two operands returns the transposed layout:
//   D(M N) = mfma(A = T(M), B = D(N)),   T(M N) = mfma(A = D(N), B = T(M)).
// Rows 12..15 of F are zero (the fourth call of each product is skipped).  Vxx lives in
// T-layout; W = Vhat F comes out in D-layout; G is computed twice, in T-layout (its x-x block
// plus K'Qux, one more call, is the next Vxx, already in T-layout) and in D-layout (whose last
// register holds Qux[g][c] and Quu[g][c-12] exactly where the rank-4 update and the gains need
// them).  The 4 x 4 solve is done redundantly by every lane for its own column of K.
// Sums run in the matrix unit's order, so results agree with the lane-per-problem kernel to
// rounding, not bit for bit.
typedef double v4f64_ __attribute__((ext_vector_type(4)));

// One reverse sweep for the problem of this wave.  ASSUME_PD: branch-free LDL' solve of Quu that
// only raises `bad` when a pivot is not positive; the kernel then repeats the sweep with the
// pivoted-LU fallback (numpy.linalg.inv semantics, status bits).
// Reciprocal by two Newton steps on the hardware estimate (the matrix-unit kernels agree with the
// scalar ones to rounding only, so the correctly rounded division is not needed here): 5
// instructions instead of the 10 of an IEEE division, four times per time step.
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float fast_rcp(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}

// Branch-free L D L' solve of the 4 x 4 system Quu [k | K_c] = rhs for the matrix-unit kernels
// (same elimination order as solve_quu, reciprocals by fast_rcp); `bad` if a pivot is not positive.
template <typename T> __device__ __forceinline__ void solve_quu4_fast(const T (&A)[4][4], T (&Bm)[4][2], bool &bad) {
    T L[4][4], d[4], dinv[4];
    bool pd = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        T dj = A[j][j];
#pragma unroll
        for (int q = 0; q < j; ++q) dj = fma_(-(L[j][q] * L[j][q]), d[q], dj);
        pd = pd && (dj > T(0));
        d[j] = dj;
        dinv[j] = fast_rcp(dj);
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            T s = A[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) s = fma_(-(L[i][q] * L[j][q]), d[q], s);
            L[i][j] = s * dinv[j];
        }
    }
    bad = bad || !pd;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        T y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T s = Bm[i][c];
#pragma unroll
            for (int q = 0; q < i; ++q) s = fma_(-L[i][q], y[q], s);
            y[i] = s;
        }
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            T s = y[i] * dinv[i];
#pragma unroll
            for (int q = i + 1; q < 4; ++q) s = fma_(-L[q][i], Bm[q][c], s);
            Bm[i][c] = s;
        }
    }
}

// cilqr.py:112-134 for m = 4 with everything in one lane (the fallback sweeps): if a corner of the box
// lo - u <= du <= hi - u satisfies the stationarity conditions of min 1/2 du'Quu du + (2 Qu)'du with
// every multiplier beyond lambda, k = that corner and K = 0 (see riccati_step)
template <typename T, class P>
__device__ __forceinline__ void boxqp_corner4(const P &p, const T (&Quu)[4][4], const T (&Qu)[4], const T (&u)[4], T (&rhs)[4][2]) {
    bool found = false;
    T du[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int mask = 0; mask < 16; ++mask) {
        T c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = ((mask >> j) & 1) ? (p.qhi[j] - u[j]) : (p.qlo[j] - u[j]);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T g = T(2) * Qu[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) g = fma_(Quu[i][j], c[j], g);
            const T y = -g;
            ok = ok && (((mask >> i) & 1) ? (y > p.lambda) : (y < -p.lambda));
        }
        if (ok && !found) {
            found = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) du[j] = c[j];
        }
    }
    if (found) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rhs[i][0] = du[i];
            rhs[i][1] = T(0);
        }
    }
}

template <bool ASSUME_PD, bool BOXQP = false>
__device__ __forceinline__ int ltv12x4_backward_sweep(const Params<double, 12, 4> &p, double *sh, int slot, int lane, bool &bad) {
    constexpr int NX = 12, NU = 4;
    const int c = lane & 15, g = lane >> 4;
    const int T1 = p.N - 1;
    const int tile = slot >> 6, ln = slot & 63; // one problem per wave: uniform
    const double *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const double *Rt = tile_ptr(p.Xref, tile, p.N * NX, ln);
    const double *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    double *kp = tile_ptr(p.k, tile, T1 * NU, ln);
    double *Kp = tile_ptr(p.K, tile, T1 * NU * NX, ln);
    const bool isx = c < NX;
    const int cu = isx ? 0 : c - NX;
    // Per-lane constants, fixed before the time loop: the weight of this lane's component of
    // [x; u], the diagonal terms it adds to its registers of the tiles, and for every row it reads
    // or writes a base pointer and a stride per time step (state lanes and input lanes read
    // different arrays: with a pointer and a stride per lane no load or store sits behind a branch).
    const double lw = isx ? p.Q2[c * NX + c] : p.R2[cu * NU + cu];
    double lwd[4], lmd[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) lwd[r] = (g + 4 * r == c) ? lw : 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) lmd[r] = (g + 4 * r == c && isx) ? p.lm : 0.0;
    const double *xuP = isx ? Xt + (size_t)c * TILE : Ut + (size_t)cu * TILE;
    const double *xrP = Rt + (size_t)(isx ? c : 0) * TILE; // input lanes read a valid row and use zero
    const unsigned xuS = isx ? NX * TILE : NU * TILE;
    const double *FP = isx ? p.ltvA + g * NX + c : p.ltvB + g * NU + cu; // F[g + 4 kk][c]: + kk * FK per call
    const unsigned FS = isx ? NX * NX : NX * NU, FK = isx ? 4 * NX : 4 * NU;
    // gains: state lanes store K[g][c], the input lanes of group 0 store k[cu], the rest nothing
    double *outP = isx ? Kp + (size_t)(g * NX + c) * TILE : kp + (size_t)cu * TILE;
    const unsigned outS = isx ? NU * NX * TILE : NU * TILE;
    const bool outOn = isx || g == 0;
    // terminal condition (ilqr.py:79-82): Vx = 2 Qf (x - xref), Vxx = 2 Qf
    double vx = isx ? p.Qf2[c * NX + c] * (Xt[(size_t)(T1 * NX + c) * TILE] - Rt[(size_t)(T1 * NX + c) * TILE]) : 0.0;
    double Vt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) Vt[r] = (isx && g + 4 * r == c) ? p.Qf2[c * NX + c] : 0.0;
    int status = 0;
    // the rows of the next PFL steps are in flight (a register ring; the loop is unrolled by PFL)
    constexpr int PFL = 4;
    double rxu[PFL], rxr[PFL], rF[PFL][3];
    auto load_step = [&](int t, int s) {
        rxu[s] = xuP[(size_t)t * xuS];
        rxr[s] = xrP[(size_t)t * (NX * TILE)];
        const double *f = FP + (size_t)t * FS;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) rF[s][kk] = f[kk * FK];
    };
#pragma unroll
    for (int s = 0; s < PFL; ++s) load_step(T1 - 1 - s > 0 ? T1 - 1 - s : 0, s);
    for (int tb = T1 - 1; tb >= 0; tb -= PFL)
#pragma unroll
    for (int s = 0; s < PFL; ++s) {
        const int t = tb - s;
        if (t < 0) break;
        const double xu_c = rxu[s], xr_c = isx ? rxr[s] : 0.0, F0 = rF[s][0], F1 = rF[s][1], F2 = rF[s][2];
        load_step(t - PFL > 0 ? t - PFL : 0, s);
        // first-order terms: [Qx; Qu] = [lx; lu] + F' Vx (ilqr.py:93-94)
        double part = F0 * __shfl(vx, g) + F1 * __shfl(vx, g + 4) + F2 * __shfl(vx, g + 8);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const double qv = fma_(lw, xu_c - xr_c, part);
        // W = (Vxx + lm I) F.  Lanes c >= 12 hold rows 12..15 of the left factor, which only produce
        // rows of the result that no later product reads: they need no masking anywhere below.
        v4f64_ W = {0.0, 0.0, 0.0, 0.0};
        W = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[0] + lmd[0], F0, W, 0, 0, 0);
        W = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[1] + lmd[1], F1, W, 0, 0, 0);
        W = __builtin_amdgcn_mfma_f64_16x16x4f64(Vt[2] + lmd[2], F2, W, 0, 0, 0);
        // G = F' W in both layouts
        v4f64_ TG = {0.0, 0.0, 0.0, 0.0}, DG = {0.0, 0.0, 0.0, 0.0};
        TG = __builtin_amdgcn_mfma_f64_16x16x4f64(W[0], F0, TG, 0, 0, 0);
        TG = __builtin_amdgcn_mfma_f64_16x16x4f64(W[1], F1, TG, 0, 0, 0);
        TG = __builtin_amdgcn_mfma_f64_16x16x4f64(W[2], F2, TG, 0, 0, 0);
        DG = __builtin_amdgcn_mfma_f64_16x16x4f64(F0, W[0], DG, 0, 0, 0);
        DG = __builtin_amdgcn_mfma_f64_16x16x4f64(F1, W[1], DG, 0, 0, 0);
        DG = __builtin_amdgcn_mfma_f64_16x16x4f64(F2, W[2], DG, 0, 0, 0);
        // + lxx = 2Q, luu = 2R on the diagonal (ilqr.py:91-92)
#pragma unroll
        for (int r = 0; r < 4; ++r) TG[r] += lwd[r];
        const double qg = DG[3] + lwd[3]; // Qux[g][c] (c < 12), Quu[g][c - 12] (c >= 12)
        __syncthreads();                  // the previous step has read sh
        sh[g * 16 + c] = qg;
        if (g == 0) sh[64 + c] = qv;
        __syncthreads();
        double Quu[NU][NU], rhs[NU][2], Qu[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
#pragma unroll
            for (int j = 0; j < NU; ++j) Quu[i][j] = sh[i * 16 + NX + j];
            Qu[i] = sh[64 + NX + i];
            rhs[i][0] = -Qu[i];
            rhs[i][1] = -sh[i * 16 + c]; // -Qux[i][c]
        }
        // k = -Quu^-1 Qu, K[:, c] = -Quu^-1 Qux[:, c]
        if (ASSUME_PD)
            solve_quu4_fast<double>(Quu, rhs, bad);
        else
            status |= solve_quu<double, NU, 2, false>(Quu, rhs, bad);
        if (BOXQP) {
            double uj[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) uj[j] = __shfl(xu_c, NX + j);
            boxqp_corner4<double>(p, Quu, Qu, uj, rhs);
        }
        const double kg = g == 0 ? rhs[0][1] : g == 1 ? rhs[1][1] : g == 2 ? rhs[2][1] : rhs[3][1]; // K[g][c]
        const double kc = cu == 0 ? rhs[0][0] : cu == 1 ? rhs[1][0] : cu == 2 ? rhs[2][0] : rhs[3][0]; // k[cu]
        if (outOn) outP[(size_t)t * outS] = isx ? kg : kc;
        // Vx = Qx + K' Qu ; Vxx = Qxx + K' Qux (ilqr.py:104-105)
        double nvx = qv;
#pragma unroll
        for (int q = 0; q < NU; ++q) nvx = fma_(rhs[q][1], Qu[q], nvx);
        vx = nvx; // (lanes c >= 12 are never asked for theirs)
        TG = __builtin_amdgcn_mfma_f64_16x16x4f64(qg, kg, TG, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) Vt[r] = TG[r];
    }
    return status;
}



// The same sweep in fp32 (v_mfma_f32_16x16x4).  That instruction returns register r of lane
// (c, g) = M[4 g + r][c] (the fp64 one: M[g + 4 r][c]), so the contraction index of call k in lane
// group g is 4 g + k, rows 12..15 of a tile live in lane group 3 (all four calls of a product are
// needed; group 3 supplies zeros), and the last lane group of the D-layout G holds whole columns:
// lane (c, 3) has Qux[0..3][c] (c < 12) or Quu[0..3][c - 12], so it solves for column c of K on its
// own.  The rank-4 update K'Qux contracts over lane group 0 and takes four calls.
typedef float v4f32_ __attribute__((ext_vector_type(4)));

template <bool ASSUME_PD, bool BOXQP = false>
__device__ __forceinline__ int ltv12x4_backward_sweep(const Params<float, 12, 4> &p, float *sh, int slot, int lane, bool &bad) {
    constexpr int NX = 12, NU = 4;
    const int c = lane & 15, g = lane >> 4;
    const int T1 = p.N - 1;
    const int tile = slot >> 6, ln = slot & 63;
    const float *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const float *Rt = tile_ptr(p.Xref, tile, p.N * NX, ln);
    const float *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    float *kp = tile_ptr(p.k, tile, T1 * NU, ln);
    float *Kp = tile_ptr(p.K, tile, T1 * NU * NX, ln);
    const bool isx = c < NX, g3 = g == 3;
    const int cu = isx ? 0 : c - NX;
    const float lw = isx ? p.Q2[c * NX + c] : p.R2[cu * NU + cu];
    bool dpos[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dpos[r] = (4 * g + r == c);
    auto load_xu = [&](int t) { return isx ? Xt[(size_t)(t * NX + c) * TILE] : Ut[(size_t)(t * NU + cu) * TILE]; };
    auto load_xr = [&](int t) { return isx ? Rt[(size_t)(t * NX + c) * TILE] : 0.0f; };
    auto load_F = [&](int t, int kk) { // F[4 g + kk][c] of step t; rows 12..15 are zero
        const int k = g3 ? 0 : 4 * g + kk;
        const float v = isx ? p.ltvA[(size_t)t * NX * NX + k * NX + c] : p.ltvB[(size_t)t * NX * NU + k * NU + cu];
        return g3 ? 0.0f : v;
    };
    float vx = isx ? p.Qf2[c * NX + c] * (Xt[(size_t)(T1 * NX + c) * TILE] - Rt[(size_t)(T1 * NX + c) * TILE]) : 0.0f;
    float Vt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) Vt[r] = (isx && dpos[r]) ? p.Qf2[c * NX + c] : 0.0f;
    int status = 0;
    constexpr int PFL = 4;
    float rxu[PFL], rxr[PFL], rF[PFL][4];
#pragma unroll
    for (int s = 0; s < PFL; ++s) {
        const int ts = T1 - 1 - s > 0 ? T1 - 1 - s : 0;
        rxu[s] = load_xu(ts);
        rxr[s] = load_xr(ts);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) rF[s][kk] = load_F(ts, kk);
    }
    for (int tb = T1 - 1; tb >= 0; tb -= PFL)
#pragma unroll
    for (int s = 0; s < PFL; ++s) {
        const int t = tb - s;
        if (t < 0) break;
        const float xu_c = rxu[s], xr_c = rxr[s];
        float F[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) F[kk] = rF[s][kk];
        const int tn = t - PFL > 0 ? t - PFL : 0;
        rxu[s] = load_xu(tn);
        rxr[s] = load_xr(tn);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) rF[s][kk] = load_F(tn, kk);
        // [Qx; Qu] = [lx; lu] + F' Vx
        float part = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) part = fma_(F[kk], __shfl(vx, (4 * g + kk) & 15), part); // group 3: F is zero there
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float qv = fma_(lw, xu_c - xr_c, part);
        v4f32_ W = {0.f, 0.f, 0.f, 0.f}, TG = {0.f, 0.f, 0.f, 0.f}, DG = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) W = __builtin_amdgcn_mfma_f32_16x16x4f32(Vt[kk] + (dpos[kk] && isx ? p.lm : 0.0f), F[kk], W, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) TG = __builtin_amdgcn_mfma_f32_16x16x4f32(g3 ? 0.0f : W[kk], F[kk], TG, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) DG = __builtin_amdgcn_mfma_f32_16x16x4f32(F[kk], g3 ? 0.0f : W[kk], DG, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            TG[r] += dpos[r] ? lw : 0.0f;
            DG[r] += dpos[r] ? lw : 0.0f;
        }
        __syncthreads();
        if (g3 && !isx) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sh[r * 4 + cu] = DG[r]; // Quu[r][cu]
            sh[16 + cu] = qv;                                   // Qu[cu]
        }
        __syncthreads();
        float Quu[NU][NU], rhs[NU][2], Qu[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
#pragma unroll
            for (int j = 0; j < NU; ++j) Quu[i][j] = sh[i * 4 + j];
            Qu[i] = sh[16 + i];
            rhs[i][0] = -Qu[i];
            rhs[i][1] = -DG[i]; // lane (c, 3), c < 12: -Qux[i][c]
        }
        status |= solve_quu<float, NU, 2, ASSUME_PD>(Quu, rhs, bad);
        if (BOXQP) {
            float uj[NU];
#pragma unroll
            for (int j = 0; j < NU; ++j) uj[j] = __shfl(xu_c, NX + j);
            boxqp_corner4<float>(p, Quu, Qu, uj, rhs);
        }
        if (g3 && isx) {
#pragma unroll
            for (int q = 0; q < NU; ++q) Kp[(size_t)((t * NU + q) * NX + c) * TILE] = rhs[q][1];
        }
        if (g3 && !isx) kp[(size_t)(t * NU + cu) * TILE] = cu == 0 ? rhs[0][0] : cu == 1 ? rhs[1][0] : cu == 2 ? rhs[2][0] : rhs[3][0];
        float nvx = qv;
#pragma unroll
        for (int q = 0; q < NU; ++q) nvx = fma_(rhs[q][1], Qu[q], nvx);
        vx = isx ? __shfl(nvx, 48 + c) : 0.0f;
        // Vxx = Qxx + K' Qux: operands on lane group 0, fetched from lane group 3
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float qx = __shfl(DG[kk], 48 + c), kx = __shfl(rhs[kk][1], 48 + c);
            const bool on = g == 0 && isx;
            TG = __builtin_amdgcn_mfma_f32_16x16x4f32(on ? qx : 0.0f, on ? kx : 0.0f, TG, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Vt[r] = (isx && !g3) ? TG[r] : 0.0f;
    }
    return status;
}

// ---------------------------------------------------------------- LDS staging of the n = 12 kernels
// (the layout is described at ltv12x4_evaluate_mfma_kernel)
constexpr int LTV_RING = 4;
// time steps between two meetings of a workgroup's waves (the ring moves in groups of this many
// steps): with a meeting every step the waves of a SIMD all want the matrix unit, then all the
// vector unit; with a few steps between meetings they drift apart and the units overlap
#ifndef LTV_BACKWARD_GROUP
#define LTV_BACKWARD_GROUP 2
#endif
#ifndef LTV_EVALUATE_RING
#define LTV_EVALUATE_RING 4
#endif
#ifndef LTV_EVALUATE_GROUP
#define LTV_EVALUATE_GROUP 1
#endif
__device__ __forceinline__ void glds16(const void *src, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// Geometry of a staged row: the values of the workgroup's 16 problems are 16 / PPG granules of 16
// bytes (PPG problems each).  Granule GPR r + j of row r holds the problems PPG P .. PPG P + PPG - 1
// with P = (j - rot(r)) mod GPR, rot(r) = r GPR / 16.
template <typename T> struct LtvStage {
    static constexpr int PPG = 16 / (int)sizeof(T), GPR = 16 / PPG;
    static __device__ __forceinline__ int rot(int r) { return (r * GPR) >> 4; }
    // which problems granule q (in loading order) of a run of rows holds: first problem of the granule
    static __device__ __forceinline__ int first_problem(int q) { return PPG * (((q % GPR) - rot(q / GPR)) & (GPR - 1)); }
    // byte offset inside a step's LDS image of the value of problem j (0..15 of the workgroup) in row r
    static __device__ __forceinline__ unsigned off(int r, int j) {
        return (unsigned)((r * GPR + (((j / PPG) + rot(r)) & (GPR - 1))) * 16 + (j % PPG) * (int)sizeof(T));
    }
};
// The 16 x 16 x 4 matrix instruction of either precision.  Lane (c, g) supplies left[c][k] and
// right[k][c] with k = g; register r of its result holds row g + 4 r (fp64) or 4 g + r (fp32) of
// column c.  The kernels below are written for the fp64 association "register r of lane group g
// is row g + 4 r"; in fp32 they get it by giving lane c the role of index tau(c) = c / 4 + 4 (c % 4)
// wherever a lane stands for a row or a column of a matrix (hardware row 4 g + r then is index
// tau(4 g + r) = g + 4 r; tau is its own inverse).
template <typename T> struct Mfma4;
//
// `run4`: a product whose left factor has only four rows, M (4 x 4 per call) times a 4 x 16 right
// factor, result row g of column c in lane (c, g).  Lane (c, g) supplies M[c % 4][k] and right[k][c],
// k = g (every group of four lanes repeats M).  In fp64 this is the instruction that multiplies four
// independent 4 x 4 x 4 blocks (block b = columns 4 b .. 4 b + 3; measured on gfx950: its operands and
// result sit in exactly these lanes), a quarter of the 16 x 16 x 4 instruction's time: on this chip
// an fp64 matrix instruction and the vector instructions of the other waves of the SIMD do not
// overlap (tools/ubench_mfma.hip), so its 48 cycles less are 48 cycles less.  fp32 has no such
// shape; there it is the 16 x 16 x 4 instruction, whose rows 4.. nobody reads.
template <> struct Mfma4<double> {
    using acc = v4f64_;
    using acc4 = double;
    static constexpr bool HAS4 = true; // run4 is an instruction of its own
    static __device__ __forceinline__ acc run(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ acc4 zero4() { return 0.0; }
    static __device__ __forceinline__ acc4 run4(double a, double b, acc4 c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ double first(acc4 c) { return c; }
    static __device__ __forceinline__ int tau(int c) { return c; }
};
template <> struct Mfma4<float> {
    using acc = v4f32_;
    using acc4 = v4f32_;
    static constexpr bool HAS4 = false;
    static __device__ __forceinline__ acc run(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ acc4 zero4() { return acc4{0.0f, 0.0f, 0.0f, 0.0f}; }
    static __device__ __forceinline__ acc4 run4(float a, float b, acc4 c) { return run(a, b, c); }
    static __device__ __forceinline__ float first(acc4 c) { return c[0]; }
    static __device__ __forceinline__ int tau(int c) { return (c >> 2) + 4 * (c & 3); }
};

// x + (x of the lane 16 / 32 away), without the LDS crossbar: the swap instructions of gfx950 exchange
// the odd rows (16 lanes) resp. the upper half of one register with the even rows resp. the lower half
// of another, so that two copies of x become "x of the even partner" and "x of the odd partner" in
// every lane.  The sum is the one `x + __shfl_xor(x, 16)` gives, bit for bit (addition commutes).
__device__ __forceinline__ double add_across16(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double add_across32(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ float add_across16(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float add_across32(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
// the sum over the four lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48), in every lane
template <typename T> __device__ __forceinline__ T add_lane_groups(T x) { return add_across32(add_across16(x)); }

// Backward sweep of the n = 12, m = 4 model for 16 neighbouring problems of a tile at once: 16
// waves, one problem each (the arithmetic of ltv12x4_backward_sweep with the branch-free solve),
// whose rows travel as whole lines.  The first waves fetch the 28 rows of a step (x, u, xref) and
// [A_t | B_t] straight into an LDS ring LTV_RING steps ahead; every wave puts its gains into an LDS
// image of the 52 rows of K_t and k_t, which the next waves write out one step later, 16 bytes per
// lane.  (One wave per workgroup, each lane loading and storing its own 8 bytes, moved 3x the bytes
// to HBM, as partial lines, and was held to a third of the speed.)  Problems of the 16 that are not
// selected keep their stored gains: the writing lanes skip their parts of the granules.  A problem
// whose Quu was not positive definite somewhere is flagged in p.redo for the fallback launch.
// BOXQP (cilqr.py:112-134): lane (ci, g) tests row g of the stationarity conditions of corner ci of the
// box, one ballot tells which corners pass, and the first one replaces the gains by k = corner, K = 0.
template <typename T, bool BOXQP = false> __global__ __launch_bounds__(1024) void ltv12x4_backward_staged_kernel(Params<T, 12, 4> p) {
    using St = LtvStage<T>;
    using MF = Mfma4<T>;
    using acc = typename MF::acc;
    constexpr int NX = 12, NU = 4, D = LTV_RING, S = LTV_BACKWARD_GROUP, ES = (int)sizeof(T), GPR = St::GPR, PPG = St::PPG;
    constexpr int IN_ROWS = 28, OUT_ROWS = 52, AGR = NX * NX * ES / 16, TBL = NX * (NX + NU) * ES / 16;
    constexpr int NLOAD = (IN_ROWS * GPR + TBL + 63) / 64, NSTORE = (OUT_ROWS * GPR + 63) / 64; // waves that load / that store
    constexpr int IN_BYTES = NLOAD * 1024, OUT_BYTES = NSTORE * 1024;
    constexpr int SHW = 40, SH_VX = 24; // words of a wave's own LDS: Quu (16), the step's inputs (4, BOXQP), .., Vx (16)
    constexpr int OFF_OUT = D * S * IN_BYTES, OFF_SH = OFF_OUT + 2 * S * OUT_BYTES, OFF_FLAG = OFF_SH + 16 * SHW * ES;
    static_assert(NLOAD + NSTORE <= 16 && D % 2 == 0, "wave roles, image parity");
    __shared__ __attribute__((aligned(16))) unsigned char lds[OFF_FLAG + 64];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int ci = MF::tau(c); // the row / column of the 16 x 16 tiles this lane stands for
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsl = slot_count(p);
    const int slot0 = blockIdx.x * 16, slot_raw = slot0 + w;
    const int slot = slot_raw < nsl ? slot_raw : (nsl > 0 ? nsl - 1 : 0);
    const int o = orig_of(p, slot);
    const bool act = slot_raw < nsl && !(p.select_converged ? (p.converged[o] == 0 || p.carried[o] == 0) : p.converged[o] != 0);
    if (lane == 0) ((int *)(lds + OFF_FLAG))[w] = act ? 1 : 0;
    __syncthreads();
    unsigned actmask = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) actmask |= (unsigned)((const int *)(lds + OFF_FLAG))[j] << j;
    if (actmask == 0) return; // uniform over the workgroup
    const int T1 = p.N - 1;
    const int tile = slot0 >> 6, ln0 = slot0 & 63;
    const T *Xt = tile_ptr(p.X, tile, p.N * NX, ln0);
    const T *Rt = tile_ptr(p.Xref, tile, p.N * NX, ln0);
    const T *Ut = tile_ptr(p.U, tile, T1 * NU, ln0);
    const bool isx = ci < NX;
    const int cu = isx ? 0 : ci - NX;
    // per-lane constants: weight of this lane's component of [x; u], the diagonal terms it adds to
    // its registers of the tiles
    const T lw = isx ? p.Q2[ci * NX + ci] : p.R2[cu * NU + cu];
    T lwd[4], lmd[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) lwd[r] = (g + 4 * r == ci) ? lw : T(0);
#pragma unroll
    for (int r = 0; r < 3; ++r) lmd[r] = (g + 4 * r == ci && isx) ? p.lm : T(0);
    // terminal condition (ilqr.py:79-82): Vx = 2 Qf (x - xref), Vxx = 2 Qf
    T vx = isx ? p.Qf2[ci * NX + ci] * (Xt[(size_t)(T1 * NX + ci) * TILE + w] - Rt[(size_t)(T1 * NX + ci) * TILE + w]) : T(0);
    T Vt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) Vt[r] = (isx && g + 4 * r == ci) ? p.Qf2[ci * NX + ci] : T(0);
    // loading lanes: source of this lane's granule at step 0 and its stride per step
    const unsigned char *src = nullptr;
    unsigned stride = 0;
    {
        const int q = threadIdx.x, row = q / GPR;
        if (q < IN_ROWS * GPR) {
            const T *b = row < 12 ? Xt + (size_t)row * TILE : row < 16 ? Ut + (size_t)(row - 12) * TILE : Rt + (size_t)(row - 16) * TILE;
            src = (const unsigned char *)(b + St::first_problem(q));
            stride = (unsigned)((row >= 12 && row < 16 ? NU : NX) * TILE * ES);
        } else {
            const int e = q - IN_ROWS * GPR < TBL ? q - IN_ROWS * GPR : TBL - 1; // (padding lanes repeat the last granule)
            src = (const unsigned char *)(e < AGR ? p.ltvA + PPG * e : p.ltvB + PPG * (e - AGR));
            stride = (unsigned)((e < AGR ? NX * NX : NX * NU) * ES);
        }
    }
    auto stage = [&](int t, int buf) {
        if (w < NLOAD) glds16(src + (size_t)t * stride, lds + buf * IN_BYTES + w * 1024);
    };
    // writing lanes: granule j of the output image goes to row j / GPR of K_t (48 rows) or k_t
    unsigned char *dst = nullptr;
    unsigned dstride = 0, keep = 0, oj = 0;
    if (w >= NLOAD && w < NLOAD + NSTORE) {
        const int j = (w - NLOAD) * 64 + lane, row = j / GPR;
        oj = (unsigned)j * 16;
        if (row < OUT_ROWS) {
            T *b = row < 48 ? tile_ptr(p.K, tile, T1 * NU * NX, ln0) + (size_t)row * TILE : tile_ptr(p.k, tile, T1 * NU, ln0) + (size_t)(row - 48) * TILE;
            const int fp = St::first_problem(j);
            dst = (unsigned char *)(b + fp);
            dstride = (unsigned)((row < 48 ? NU * NX : NU) * TILE * ES);
            keep = (actmask >> fp) & ((1u << PPG) - 1u); // which of the granule's problems are written
        }
    }
    auto write_out = [&](int t, int buf) {
        if (keep == 0) return;
        const uint4 v = *(const uint4 *)(lds + OFF_OUT + buf * OUT_BYTES + oj);
        unsigned char *d = dst + (size_t)t * dstride;
        if (keep == (1u << PPG) - 1u) {
            *(uint4 *)d = v;
        } else {
            T e[PPG];
            __builtin_memcpy(e, &v, 16);
#pragma unroll
            for (int q = 0; q < PPG; ++q)
                if ((keep >> q) & 1u) ((T *)d)[q] = e[q];
        }
    };
    // where this lane's operands sit in a step's input image, and its gain in the output image
    const unsigned oxu = St::off(ci, w), oxr = St::off(16 + (isx ? ci : 0), w);
    const unsigned oF = IN_ROWS * GPR * 16 + (isx ? (g * NX + ci) * ES : NX * NX * ES + (g * NU + cu) * ES), oFk = isx ? 4 * NX * ES : 4 * NU * ES;
    const unsigned ogain = St::off(isx ? g * NX + ci : 48 + g, w);
    const bool gain_on = isx || ci == NX;
    T *sh = (T *)(lds + OFF_SH) + w * SHW; // Quu of this wave's problem, row-major; then the inputs of the step
    const T qlo_g = p.qlo[g], qhi_g = p.qhi[g];
    // lane (i, j) = (ci, g), i < 4, computes entry (i, j) of the inverse = cofactor (j, i) / det: the
    // entry (j, i) of Quu and the nine entries of the rows j+1.. and the columns i+1.. (cyclically)
    int oq[10];
    oq[0] = g * 4 + (ci & 3);
#pragma unroll
    for (int r = 1; r < 4; ++r)
#pragma unroll
        for (int q = 1; q < 4; ++q) oq[(r - 1) * 3 + q] = ((g + r) & 3) * 4 + (((ci & 3) + q) & 3);
    const T csign = ((ci + g) & 1) ? T(-1) : T(1);
    bool good = true;
    const bool is12 = ci == NX;
    auto one_step = [&](const unsigned char *img, unsigned char *out) {
        const T xu_c = *(const T *)(img + oxu), xr_raw = *(const T *)(img + oxr);
        const T F0 = *(const T *)(img + oF), F1 = *(const T *)(img + oF + oFk), F2 = *(const T *)(img + oF + 2 * oFk); // F[g + 4 kk][ci]
        const T xr_c = isx ? xr_raw : T(0);
        // first-order terms: [Qx; Qu] = [lx; lu] + F' Vx (ilqr.py:93-94)
        // (Vx[i] travels through the wave's own LDS words, at (i % 4) 4 + i / 4, so that the three
        // components g, g + 4, g + 8 a lane multiplies are neighbours: a write and two reads instead of
        // six crossbar instructions)
        if (g == 0) sh[SH_VX + (ci & 3) * 4 + (ci >> 2)] = vx;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const T vx0 = sh[SH_VX + g * 4], vx1 = sh[SH_VX + g * 4 + 1], vx2 = sh[SH_VX + g * 4 + 2];
        // W = (Vxx + lm I) F, then G = F' W with register r of lane (ci, g) = G[ci][g + 4 r] (layouts:
        // see ltv12x4_backward_sweep).  Lanes that stand for an input hold rows 12..15 of the left
        // factor of W, which only produce rows that no later product reads; row 12 is put to use: with
        // Vx' as that row, row 12 of W is Vx' F = (F' Vx)', the first-order product, for nothing
        // (register 3 of the lanes g = 0; a matrix instruction costs the SIMD the same 64 cycles
        // whatever its padding rows hold, and a sum over the lane groups costs ten more instructions).
        acc W = {T(0), T(0), T(0), T(0)};
        W = MF::run(is12 ? vx0 : Vt[0] + lmd[0], F0, W);
        W = MF::run(is12 ? vx1 : Vt[1] + lmd[1], F1, W);
        W = MF::run(is12 ? vx2 : Vt[2] + lmd[2], F2, W);
        const T qv = fma_(lw, xu_c - xr_c, W[3]); // (lanes g = 0)
        acc TG = {T(0), T(0), T(0), T(0)};
        TG = MF::run(W[0], F0, TG);
        TG = MF::run(W[1], F1, TG);
        TG = MF::run(W[2], F2, TG);
        // + lxx = 2Q, luu = 2R on the diagonal (ilqr.py:91-92)
#pragma unroll
        for (int r = 0; r < 4; ++r) TG[r] += lwd[r];
        // Quu^-1 by cofactors, one entry per lane.  Input lane (12 + i, j) holds Quu[i][j] (register 3)
        // and leaves it in the wave's own LDS words (a wave's LDS accesses execute in order); lane
        // (i, j), i < 4, reads the matrix with rows and columns rotated so that entry (j, i) comes
        // first, takes the 3 x 3 determinant of the rest (the minor: a rotation of three rows is even)
        // and the four lanes j of an i add up a_ji C_ji to the determinant (expansion along column i).
        if (!isx) sh[cu * 4 + g] = TG[3];
        if (BOXQP && !isx && g == 0) sh[16 + cu] = xu_c;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T cof = T(0), a_ij = T(0);
        bool pd = true;
        { // (every lane: lane (c, g) computes entry (c % 4, g), which is what run4 wants from it)
            const T b11 = sh[oq[1]], b12 = sh[oq[2]], b13 = sh[oq[3]], b21 = sh[oq[4]], b22 = sh[oq[5]], b23 = sh[oq[6]], b31 = sh[oq[7]],
                    b32 = sh[oq[8]], b33 = sh[oq[9]];
            a_ij = sh[oq[0]];
            const T m1 = fma_(b22, b33, -(b23 * b32)), m2 = fma_(b21, b33, -(b23 * b31)), m3 = fma_(b21, b32, -(b22 * b31));
            const T minor = fma_(b13, m3, fma_(b11, m1, -(b12 * m2)));
            cof = csign * minor;
            // positive definite <=> the nested principal minors {3}, {2,3}, {1,2,3}, {0,1,2,3} are
            // positive: lane (0, 0) has the first three at hand
            pd = b33 > T(0) && m1 > T(0) && minor > T(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const T det = add_lane_groups(a_ij * cof);
        good = good && pd && det > T(0); // (lane 0's is the one that counts)
        const T inv_e = cof * fast_rcp(det); // (Quu^-1)[ci % 4][g]
        // [K | k] = -Quu^-1 [Qux | Qu] as one more product: this lane supplies entry (g, ci) of the
        // right factor, Qux[g][ci] (state lanes; taken from G[ci][12 + g], which this lane holds: G is
        // symmetric up to the rounding of Vxx) or Qu[g] (the lane of index 12), and gets row g of
        // column ci of the result in register 0.
        const T qug = __shfl(qv, MF::tau(NX + g)); // Qu[g]
        const T qg = TG[3];
        T kg = -MF::first(MF::run4(inv_e, ci == NX ? qug : qg, MF::zero4())); // K[g][ci] (ci < 12), k[g] (ci == 12)
        if (BOXQP) {
            T uj[NU], gg = T(2) * qug;
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                uj[j] = sh[16 + j];
                const T cj = ((ci >> j) & 1) ? p.qhi[j] - uj[j] : p.qlo[j] - uj[j];
                gg = fma_(sh[g * 4 + j], cj, gg);
            }
            const T y = -gg;
            const bool okr = ((ci >> g) & 1) ? (y > p.lambda) : (y < -p.lambda);
            const unsigned long long bm = __ballot(okr);
            const unsigned okc = (unsigned)(bm & (bm >> 16) & (bm >> 32) & (bm >> 48)) & 0xffffu; // bit c: corner tau(c) passes
            int corner = -1;
#pragma unroll
            for (int cm = 15; cm >= 0; --cm)
                if ((okc >> MF::tau(cm)) & 1u) corner = cm; // the lowest passing corner, as the serial test finds it
            if (corner >= 0) { // (uniform over the wave)
                const T ug = g == 0 ? uj[0] : g == 1 ? uj[1] : g == 2 ? uj[2] : uj[3];
                const T dug = ((corner >> g) & 1) ? qhi_g - ug : qlo_g - ug;
                kg = ci == NX ? dug : T(0);
            }
        }
        if (gain_on) *(T *)(out + ogain) = kg;
        // Vx = Qx + K' Qu ; Vxx = Qxx + K' Qux (ilqr.py:104-105): one product, with Qu' as row 12 of the
        // left factor Qux' (and the rows 12.. of the sum, which have served, cleared)
        TG[3] = T(0);
        TG = MF::run(is12 ? qug : qg, kg, TG);
        vx = qv + TG[3]; // (lanes g = 0 of the states; nobody asks the others for theirs)
#pragma unroll
        for (int r = 0; r < 3; ++r) Vt[r] = TG[r];
    };
    // Group m handles the steps t = T1 - 1 - (m S + u), u = 0 .. S - 1, from ring slot m mod D and
    // leaves their gains in the output images of parity m mod 2.  A direct load is ordered before a
    // read of its data only by the loading wave's counted wait followed by a barrier, so per group:
    // wait until this wave's pieces of the group have landed, meet (every piece has landed; every
    // wave is done with the previous group and its gains are in the images), refill the ring slot,
    // write out the images of the previous group, compute.
    const int NM = (T1 + S - 1) / S;
    auto stage_group = [&](int m, int slot) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int t = T1 - 1 - (m * S + u);
            stage(t > 0 ? t : 0, slot * S + u);
        }
    };
    auto write_group = [&](int m) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int t = T1 - 1 - (m * S + u);
            if (t >= 0) write_out(t, (m & 1) * S + u);
        }
    };
#pragma unroll
    for (int m = 0; m < D - 1; ++m) stage_group(m, m);
    for (int mb = 0; mb < NM; mb += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int m = mb + s;
            if (m >= NM) break; // uniform over the workgroup
            if (w < NLOAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * S) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage_group(m + D - 1, (s + D - 1) % D);
            if (m > 0) write_group(m - 1);
            if (act) {
#pragma unroll
                for (int u = 0; u < S; ++u) {
                    if (T1 - 1 - (m * S + u) < 0) break;
                    one_step(lds + (s * S + u) * IN_BYTES, lds + OFF_OUT + ((s & 1) * S + u) * OUT_BYTES);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    write_group(NM - 1);
    if (act && lane == 0) p.redo[o] = good ? 0 : 1;
}

// The fallback launch behind ltv12x4_backward_staged_kernel (normally idle): the problems it flagged
// in p.redo, one wave per workgroup each, with the pivoted-LU solve of the lane-per-problem kernel
// (ilqr.py:99 inverts Quu whatever its signs).  A kernel of its own, so that the fallback's
// registers do not cost the main kernel its occupancy.
template <typename T, bool BOXQP = false> __global__ __launch_bounds__(64) void ltv12x4_backward_redo_kernel(Params<T, 12, 4> p) {
    __shared__ T sh[80];
    const int lane = threadIdx.x;
    const int slot = blockIdx.x;
    if (slot >= slot_count(p)) return;
    const int o = orig_of(p, slot);
    if (p.select_converged ? (p.converged[o] == 0 || p.carried[o] == 0) : p.converged[o] != 0) return;
    if (p.redo[o] == 0) return;
    bool bad = false;
    const int status = ltv12x4_backward_sweep<false, BOXQP>(p, sh, slot, lane, bad);
    if (lane == 0 && status != 0 && p.status) p.status[o] |= status;
}

// ---------------------------------------------------------------- rollout step
// u_new = (u_prev + alpha k) + K (x_new - x_prev), clamped (ilqr.py:56-61).
// The reference's clamp is two comparisons (`if u < lo: u = lo`, `if u > hi: u = hi`), which leave a
// NaN as it is.  !FAST states exactly that, three instructions per bound and input on this chip
// (compare, two 32-bit selects).  FAST takes the maximum and the minimum instead, one instruction
// per bound, which agree with the comparisons for every ordered value but return the bound for a
// NaN: an unordered input is reported in `bad` (one compare for a pair of inputs) and the caller
// repeats its sweep with FAST = false, as for an angle outside the fast trigonometric range.
template <bool FAST, typename T> __device__ __forceinline__ T fmax_(T a, T b);
template <> __device__ __forceinline__ double fmax_<true, double>(double a, double b) { return __builtin_fmax(a, b); }
template <> __device__ __forceinline__ float fmax_<true, float>(float a, float b) { return __builtin_fmaxf(a, b); }
template <bool FAST, typename T> __device__ __forceinline__ T fmin_(T a, T b);
template <> __device__ __forceinline__ double fmin_<true, double>(double a, double b) { return __builtin_fmin(a, b); }
template <> __device__ __forceinline__ float fmin_<true, float>(float a, float b) { return __builtin_fminf(a, b); }
template <bool FAST, typename T, class M, class P>
__device__ __forceinline__ void control_update(const P &p, T alpha, const T (&x)[M::NX], const T (&xp)[M::NX], const T (&up)[M::NU],
                                               const T (&Kt)[M::NU][M::NX], const T (&kt)[M::NU], T (&un)[M::NU], bool &bad) {
    constexpr int NX = M::NX, NU = M::NU;
    T dx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) dx[i] = x[i] - xp[i];
    T raw[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        T fb = Kt[i][0] * dx[0];
#pragma unroll
        for (int j = 1; j < NX; ++j) fb = fma_(Kt[i][j], dx[j], fb);
        T v = fma_(alpha, kt[i], up[i]) + fb;
        raw[i] = v;
        // Without control limits the host passes lo = -inf, hi = +inf, which makes the clamp the
        // identity, so the time step has no branch.
        if (FAST) {
            v = fmin_<true, T>(fmax_<true, T>(v, p.lo[i]), p.hi[i]);
        } else {
            v = (p.lo[i] > v) ? p.lo[i] : v;
            v = (p.hi[i] < v) ? p.hi[i] : v;
        }
        un[i] = v;
    }
    if (FAST) {
#pragma unroll
        for (int i = 0; i < NU; i += 2) bad = bad || __builtin_isunordered(raw[i], raw[i + 1 < NU ? i + 1 : i]);
    }
}

// Row prefetch of the first line-search stage through LDS (one lane per problem): the rows of a time
// step (K, k, U, Xref: 8 KiB per wave for the Car in fp64) are contiguous per array, so the wave
// fetches them with direct-to-LDS loads of 1 KiB each, EVAL_RING - 1 steps ahead of their use,
// into a ring of its own, and every lane then reads its values of the step from there.  In registers
// the same rows cost 32 VGPRs per step in flight, and with the six candidates' states a second step
// in flight no longer fits: the compiler then issues the loads late in the step before, and the wave
// waits for memory in a sixth of its cycles.
#ifndef ILQR_EVAL_RING
#define ILQR_EVAL_RING 4
#endif
constexpr int EVAL_RING = ILQR_EVAL_RING;
// (L lanes per problem: a wave works on the 64 / L problems of a part of a tile, whose values of a row
// group are PPW * V contiguous elements; a load instruction gathers 64 granules of 16 bytes, one per
// lane, from wherever they are, and LDS receives them in lane order: the step's image in LDS is the
// wave's granules of K, k, U and Xref in that order, row group by row group.)
template <typename T, class M, int L> struct EvalRing {
    static constexpr int G = 16 / (int)sizeof(T);                   // elements per 16-byte granule
    static constexpr int PPW = TILE / L;                            // problems per wave
    static constexpr int VK = GainLayout<M>::VK, Vk = GainLayout<M>::Vk, VR = TargetLayout<M, false>::V;
    // elements per step of the wave's problems, and where each array starts in the image
    static constexpr int KE = M::NU * M::NX * PPW, kE = M::NU * PPW, UE = M::NU * PPW, RE = M::NX * PPW;
    static constexpr int oK = 0, ok = KE, oU = KE + kE, oR = KE + kE + UE, STEP = KE + kE + UE + RE;
    static constexpr int NLD = STEP / (TILE * G);                   // load instructions per step
    static constexpr int BYTES = EVAL_RING * STEP * (int)sizeof(T); // per wave
    // whole granules per row group and whole instructions per step; the rings of the workgroups of a CU
    // (four waves each, L of them) within 128 KiB of LDS
    static constexpr bool OK = M::NX <= 8 && (L == 1 || L == 2) && (PPW * VK) % G == 0 && (PPW * Vk) % G == 0 && PPW % G == 0 &&
                               (PPW * VR) % G == 0 && STEP % (TILE * G) == 0 && 4 * L * BYTES <= (EVAL_RING > 4 ? 163840 : 131072);
    // element offsets inside the image: K[i][j], k[i], U[i], Xref[i] of the wave's problem number pl
    static __device__ __forceinline__ constexpr int at_K(int i, int j, int pl) { return oK + ((i * M::NX + j) / VK) * (PPW * VK) + pl * VK + (i * M::NX + j) % VK; }
    static __device__ __forceinline__ constexpr int at_k(int i, int pl) { return ok + (i / Vk) * (PPW * Vk) + pl * Vk + i % Vk; }
    static __device__ __forceinline__ constexpr int at_U(int i, int pl) { return oU + i * PPW + pl; }
    static __device__ __forceinline__ constexpr int at_R(int i, int pl) { return oR + (i / VR) * (PPW * VR) + pl * VR + i % VR; }
    // Source of granule number `gi` of a step's image: which array (0 K, 1 k, 2 U, 3 Xref), and the
    // element offset inside that array's rows of the step for the tile (`first` = the wave's first lane)
    static __device__ __forceinline__ void source(int gi, int first, int &arr, int &off) {
        const int e = gi * G; // element of the image
        int base, V;
        if (e < ok) arr = 0, base = oK, V = VK;
        else if (e < oU) arr = 1, base = ok, V = Vk;
        else if (e < oR) arr = 2, base = oU, V = 1;
        else arr = 3, base = oR, V = VR;
        const int w = e - base, grp = w / (PPW * V), in = w % (PPW * V);
        off = grp * (TILE * V) + first * V + in;
    }
};
template <typename T> struct EvalTile { // what the LDS path needs beside the per-lane pointers
    const T *K = nullptr, *k = nullptr, *U = nullptr, *R = nullptr; // the wave's tile of each array
    unsigned char *ring = nullptr;                                   // the wave's own ring
    int lane = 0, first = 0;                                         // lane; first lane of the tile this wave works on
};

template <typename T, int NX, int NU> struct FwdRow { // rows of one step shared by all step sizes of a problem
    T K[NU][NX], k[NU], xp[NX], up[NU], xr[NX], cx, cy, rad, s, c;
};

// One time step of one line-search candidate: stage cost of (x, u_new) and the next state
// (ilqr.py:54-63 with the cost of ilqr.py:36-43 accumulated on the way).
template <typename T, class M, bool DIAG, bool BARRIER, bool FAST, class P>
__device__ __forceinline__ void candidate_step(const P &p, int t, T alpha, const FwdRow<T, M::NX, M::NU> &r, T (&x)[M::NX], T &cost,
                                               bool &bad) {
    constexpr int NX = M::NX, NU = M::NU;
    T e[NX], un[NU], xn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) e[i] = x[i] - r.xr[i];
    control_update<FAST, T, M>(p, alpha, x, r.xp, r.up, r.K, r.k, un, bad);
    T stage = quad_<T, NX, DIAG>(p.Q, e) + quad_<T, NU, DIAG>(p.R, un);
    if (BARRIER && t > 0) {
        T dlx[NX], dlxx[NX][NX];
        stage += barrier_terms<T, NX, false>(p, x, r.cx, r.cy, r.rad, dlx, dlxx);
    }
    cost += stage;
    M::template step<FAST>(p, t, x, un, xn, bad);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
}

// Terminal term of a candidate (ilqr.py:44-45) added to its accumulated stage costs.
template <typename T, class M, bool DIAG, bool BARRIER, class P>
__device__ __forceinline__ T candidate_total(const P &p, const T (&x)[M::NX], const T (&xr)[M::NX], T cx, T cy, T rad, T cost) {
    constexpr int NX = M::NX;
    T e[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) e[i] = x[i] - xr[i];
    T term = quad_<T, NX, DIAG>(p.Qf, e);
    if (BARRIER) {
        T dlx[NX], dlxx[NX][NX];
        term += barrier_terms<T, NX, false>(p, x, cx, cy, rad, dlx, dlxx);
    }
    return cost + term;
}

// ---------------------------------------------------------------- evaluate
// L lanes per problem x A step sizes per lane = L*A candidates per round;
// rounds continue (in increasing index order, so the largest improving step
// wins exactly as in the serial search) until every problem of the wave has a
// winner or the candidates are exhausted.
// REROLL: the trajectory in X is the rollout of the inputs in U by this very code (true once a
// line search has committed a step, see commit_kernel), so the previous states the feedback
// term needs are rolled out again in registers instead of being read: one more model step per
// time step, NX fewer values of traffic.
// TC (with REROLL): the previous trajectory's sine/cosine pairs come from SCp (Params::SC), which
// turns its re-rollout into a few multiply-adds.
// CKPT: the state of every candidate at the start of every segment of COMMIT_SEG time steps is
// stored to CKp (this problem's first checkpoint slot, see Params::CK) by the lanes with `ck_store`.
// LDSR: the rows come through the wave's LDS ring (EvalRing; L = 1, REROLL, no constraint rows, no cache).
template <typename T, class M, bool DIAG, bool BARRIER, int L, int A, bool REROLL, bool FAST, bool TC = false, bool CKPT = false,
          bool LDSR = false, class P>
__device__ __forceinline__ void evaluate_rounds(const P &p, const T *Xt, const T *Rt, const T *Ut, const T *Kp, const T *kp,
                                                const T *Cp, const T *Dp, int al, bool active, T Jcur, int base0, int end,
                                                int &best, T &best_cost, bool &bad, const T *SCp = nullptr, T *CKp = nullptr,
                                                bool ck_store = false, EvalTile<T> et = EvalTile<T>()) {
    constexpr int NX = M::NX, NU = M::NU;
    static_assert(!LDSR || (REROLL && !BARRIER && !TC && !CKPT && EvalRing<T, M, L>::OK), "LDS row ring: see EvalRing");
    const int T1 = p.N - 1;
    auto load_row = [&](int t, FwdRow<T, NX, NU> &r) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            r.k[i] = kp[gain_k<M>(t, i)];
            r.up[i] = Ut[(t * NU + i) * TILE];
#pragma unroll
            for (int j = 0; j < NX; ++j) r.K[i][j] = Kp[gain_K<M>(t * NU + i, j)];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (!REROLL) r.xp[i] = Xt[(t * NX + i) * TILE];
            r.xr[i] = Rt[target_x<M, BARRIER>(t, i)];
        }
        if (BARRIER) {
            r.cx = Cp[(t * 2 + 0) * TILE];
            r.cy = Cp[(t * 2 + 1) * TILE];
            r.rad = Dp[t * TILE];
        } else {
            r.cx = r.cy = r.rad = T(0);
        }
        if (REROLL && TC) {
            r.s = SCp[(t * 2 + 0) * TILE];
            r.c = SCp[(t * 2 + 1) * TILE];
        } else {
            r.s = r.c = T(0);
        }
    };
    best = 0x7fffffff;
    best_cost = T(0);
    for (int base = base0; base < end; base += L * A) {
        T x[A][NX], cost[A], alpha[A];
#pragma unroll
        for (int j = 0; j < A; ++j) {
            alpha[j] = pow2_neg<T>(base + al * A + j);
            cost[j] = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[j][i] = Xt[i * TILE];
        }
        T xq[NX]; // state of the previous trajectory (REROLL)
#pragma unroll
        for (int i = 0; i < NX; ++i) xq[i] = Xt[i * TILE];
        FwdRow<T, NX, NU> nxt;
        if (!LDSR) load_row(0, nxt);
        auto one_step = [&](int t, FwdRow<T, NX, NU> &r) {
            if (REROLL) {
#pragma unroll
                for (int i = 0; i < NX; ++i) r.xp[i] = xq[i];
            }
#pragma unroll
            for (int j = 0; j < A; ++j) candidate_step<T, M, DIAG, BARRIER, FAST>(p, t, alpha[j], r, x[j], cost[j], bad);
            if (REROLL) {
                T xqn[NX];
                model_step<FAST, TC, M>(p, t, xq, r.up, xqn, r.s, r.c, bad);
#pragma unroll
                for (int i = 0; i < NX; ++i) xq[i] = xqn[i];
            }
        };
        if constexpr (LDSR) {
            using ER = EvalRing<T, M, L>;
            T *const img0 = (T *)et.ring;
            const int pl = et.lane / L;
            // this lane's granule of each load instruction: where it comes from at step 0, and how far
            // that moves per step (an array's rows of a step are contiguous)
            const T *src[ER::NLD];
            unsigned stride[ER::NLD];
#pragma unroll
            for (int q = 0; q < ER::NLD; ++q) {
                int arr, off;
                ER::source(q * TILE + et.lane, et.first, arr, off);
                src[q] = (arr == 0 ? et.K : arr == 1 ? et.k : arr == 2 ? et.U : et.R) + off;
                stride[q] = (unsigned)((arr == 0 ? NU * NX : arr == 3 ? NX : NU) * TILE);
            }
            // the rows of step t (past the end: of the last step again, so that the count of loads in
            // flight stays what the waits below assume) into ring slot `slot`
            auto issue = [&](int t, int slot) {
                const size_t tt = (size_t)(t < T1 ? t : T1 - 1);
                T *dst = img0 + slot * ER::STEP;
#pragma unroll
                for (int q = 0; q < ER::NLD; ++q) glds16(src[q] + tt * stride[q], dst + q * (TILE * ER::G));
            };
            auto read_row = [&](int slot, FwdRow<T, NX, NU> &r) {
                const T *img = img0 + slot * ER::STEP;
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    r.k[i] = img[ER::at_k(i, pl)];
                    r.up[i] = img[ER::at_U(i, pl)];
#pragma unroll
                    for (int j = 0; j < NX; ++j) r.K[i][j] = img[ER::at_K(i, j, pl)];
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) r.xr[i] = img[ER::at_R(i, pl)];
                r.cx = r.cy = r.rad = r.s = r.c = T(0);
            };
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // (nothing of an earlier sweep still touches the ring)
#pragma unroll
            for (int q = 0; q < EVAL_RING - 1; ++q) issue(q, q);
            int slot = 0;
            for (int t = 0; t < T1; ++t) {
                // in flight: the steps t .. t + EVAL_RING - 2; loads complete in order
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((EVAL_RING - 2) * ER::NLD) : "memory");
                const int refill = slot == 0 ? EVAL_RING - 1 : slot - 1; // the slot step t - 1 was read from
                issue(t + EVAL_RING - 1, refill);
                FwdRow<T, NX, NU> r;
                read_row(slot, r);
                one_step(t, r);
                slot = slot + 1 == EVAL_RING ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (CKPT) {
            // The horizon in segments.  The first step of a segment also stores the candidates'
            // states (the segment's checkpoint), AFTER it has issued the loads of the next row: loads
            // and stores retire in order, so stores issued before them would make the next step wait
            // for their acknowledgement as well.
            int t = 0;
            for (int seg = 0; t < T1; ++seg) {
                {
                    FwdRow<T, NX, NU> r = nxt;
                    if (t + 1 < T1) load_row(t + 1, nxt);
                    if (ck_store) {
#pragma unroll
                        for (int j = 0; j < A; ++j) {
                            const int a = base + al * A + j;
                            if (a < CK_CANDIDATES) {
#pragma unroll
                                for (int i = 0; i < NX; ++i) CKp[(size_t)((seg * CK_CANDIDATES + a) * NX + i) * TILE] = x[j][i];
                            }
                        }
                    }
                    one_step(t, r);
                    ++t;
                }
                const int tend = seg * COMMIT_SEG + COMMIT_SEG < T1 ? seg * COMMIT_SEG + COMMIT_SEG : T1;
                for (; t < tend; ++t) {
                    FwdRow<T, NX, NU> r = nxt;
                    if (t + 1 < T1) load_row(t + 1, nxt);
                    one_step(t, r);
                }
            }
        } else {
            // two steps per trip, the rows alternating between two register sets: a single set that is
            // copied at the top of every step costs a move per value and step
            FwdRow<T, NX, NU> nx2;
            int t = 0;
            for (; t + 2 < T1; t += 2) {
                load_row(t + 1, nx2);
                one_step(t, nxt);
                load_row(t + 2, nxt);
                one_step(t + 1, nx2);
            }
            if (t + 1 < T1) {
                load_row(t + 1, nx2);
                one_step(t, nxt);
                one_step(t + 1, nx2);
            } else if (t < T1) {
                one_step(t, nxt);
            }
        }
        // terminal term (ilqr.py:44-45)
        T xr[NX], cx = T(0), cy = T(0), rad = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = Rt[target_x<M, BARRIER>(T1, i)];
        if (BARRIER) {
            cx = Cp[(T1 * 2 + 0) * TILE];
            cy = Cp[(T1 * 2 + 1) * TILE];
            rad = Dp[T1 * TILE];
        }
        int cand = 0x7fffffff;
        T cand_cost = T(0);
#pragma unroll
        for (int j = A - 1; j >= 0; --j) {
            T c = candidate_total<T, M, DIAG, BARRIER>(p, x[j], xr, cx, cy, rad, cost[j]);
            const int a = base + al * A + j;
            if (a < end && c < Jcur) { // strict test of ilqr.py:65; NaN never passes
                cand = a;
                cand_cost = c;
            }
        }
        // lowest candidate index among the L lanes of the problem
#pragma unroll
        for (int off = 1; off < L; off <<= 1) {
            int oc = __shfl_xor(cand, off);
            T ocost = __shfl_xor(cand_cost, off);
            if (oc < cand) {
                cand = oc;
                cand_cost = ocost;
            }
        }
        if (best == 0x7fffffff && cand != 0x7fffffff) {
            best = cand;
            best_cost = cand_cost;
        }
        if (__ballot(active && best == 0x7fffffff) == 0ull) break;
    }
}

// The line search runs in two stages.  Stage one (evaluate_kernel) covers every active problem
// and rolls out the candidates [0, end) - all of them if L*A >= n_alphas, else the first L*A,
// which settle all but a few per cent of the problems.  A problem that found no winner among
// them while candidates remain is appended to a device-side list; stage two
// (evaluate_listed_kernel) walks that list with the remaining candidates.  Candidates are still
// examined in increasing index order, so the result is that of the serial search.
// (body of evaluate_kernel: the work of wave number `wave` of the launch)
template <typename T, class M, bool DIAG, bool BARRIER, int L, int A, bool REROLL, bool TC = false, bool CKPT = false, bool LDSR = false, class P>
__device__ __forceinline__ void evaluate_body(const P &p, int end, int wave, int lane, unsigned char *ring = nullptr) {
    constexpr int NX = M::NX, NU = M::NU;
    constexpr int PPW = 64 / L; // problems per wave
    const int pl = lane / L, al = lane % L;
    const int g = wave * PPW + pl;
    const int nsl = slot_count(p);
    // with active-set compaction the host launches several mappings and the one whose range
    // holds the current number of slots does the work
    if (nsl <= p.slots_lo || nsl > p.slots_hi) return;
    if (wave * PPW >= nsl) return;
    const bool valid = g < nsl;
    const int b = valid ? g : nsl - 1; // slot
    const int o = orig_of(p, b);
    const bool active = valid && p.converged[o] == 0;
    if (__ballot(active) == 0ull) return;
    const int T1 = p.N - 1;
    const T Jcur = p.J[o];
    const int tile = b >> 6, ln = b & 63;
    const T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const T *Rt = tile_ptr_v<TargetLayout<M, BARRIER>::V>(p.Xref, tile, p.N * NX, ln);
    const T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    const T *Cp = BARRIER ? tile_ptr(p.centers, tile, p.N * 2, ln) : nullptr;
    const T *Dp = BARRIER ? tile_ptr(p.radius, tile, p.N, ln) : nullptr;
    const T *kp = tile_ptr_v<GainLayout<M>::Vk>(p.k, tile, T1 * NU, ln);
    const T *Kp = tile_ptr_v<GainLayout<M>::VK>(p.K, tile, T1 * NU * NX, ln);
    int best;
    T best_cost;
    bool bad = false;
    const T *SCp = (REROLL && TC) ? tile_ptr(p.SC, tile, T1 * 2, ln) : nullptr;
    const int nseg = (T1 + COMMIT_SEG - 1) / COMMIT_SEG;
    T *CKp = CKPT ? tile_ptr(p.CK, tile, nseg * CK_CANDIDATES * NX, ln) : nullptr;
    EvalTile<T> et;
    if constexpr (LDSR) { // (the wave's problems are PPW neighbouring lanes of one tile)
        const int wt = (wave * PPW) >> 6;
        et.K = p.K + (size_t)wt * (T1 * NU * NX) * TILE;
        et.k = p.k + (size_t)wt * (T1 * NU) * TILE;
        et.U = p.U + (size_t)wt * (T1 * NU) * TILE;
        et.R = p.Xref + (size_t)wt * (p.N * NX) * TILE;
        et.ring = ring;
        et.lane = lane;
        et.first = (wave * PPW) & 63;
    }
    evaluate_rounds<T, M, DIAG, BARRIER, L, A, REROLL, true, TC, CKPT, LDSR>(p, Xt, Rt, Ut, Kp, kp, Cp, Dp, al, active, Jcur, 0, end, best,
                                                                              best_cost, bad, SCp, CKp, true, et);
    // an angle outside the fast trigonometric range anywhere in the wave: redo with the library
    // routine; a problem one of whose lanes saw such an angle takes the redone result (and tells
    // the commit kernel to use the library routine too), the others keep theirs, so a problem's
    // result does not depend on which problems share its wave
    bool pbad = bad;
#pragma unroll
    for (int w = 1; w < L; w <<= 1) pbad |= __shfl_xor((int)pbad, w) != 0;
    if (__ballot(pbad && active) != 0ull) {
        int best2;
        T best_cost2;
        bool bad2 = false;
        // (the checkpoints of the problems that take the redone result are replaced, the others' stay)
        evaluate_rounds<T, M, DIAG, BARRIER, L, A, REROLL, false, TC, CKPT, LDSR>(p, Xt, Rt, Ut, Kp, kp, Cp, Dp, al, active, Jcur, 0, end, best2,
                                                                                   best_cost2, bad2, SCp, CKp, pbad, et);
        best = pbad ? best2 : best;
        best_cost = pbad ? best_cost2 : best_cost;
    }
    const bool found = best != 0x7fffffff;
    // stage two continues with the remaining candidates: one atomic per wave
    const bool more = active && al == 0 && !found && end < p.n_alphas;
    const unsigned long long mm = __ballot(more);
    if (mm != 0ull) {
        int at = 0;
        if (lane == 0) at = atomicAdd(p.list_count, __popcll(mm));
        at = __shfl(at, 0) + __popcll(mm & ((1ull << lane) - 1ull));
        if (more) {
            p.list[at] = b;
            p.alpha_idx[o] = -2; // pending: stage two (or the commit that runs beside it) decides
        }
    }
    if (active && al == 0 && !more) {
        p.alpha_idx[o] = found ? best : -1;
        p.Jtry[o] = best_cost;
        p.slow_trig[o] = (p.slow_trig[o] & (6 | SLOW_TRIG_REGEN)) | (pbad ? 1 : 0);
    }
}

template <typename T, class M, bool DIAG, bool BARRIER, int L, int A, bool REROLL, bool TC = false, bool CKPT = false, bool LDSR = false>
__global__ __launch_bounds__(256, EVAL_MIN_WAVES) void evaluate_kernel(Params<T, M::NX, M::NU> p, int end) {
    static_assert(!LDSR, "the ring variant is evaluate_ring_kernel");
    evaluate_body<T, M, DIAG, BARRIER, L, A, REROLL, TC, CKPT>(p, end, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), threadIdx.x & 63);
}

// Stage one with the rows through LDS (EvalRing): L workgroups of four waves per CU, i.e. L waves per
// SIMD, each within 512 / L registers.  With two lanes per problem (three step sizes each) the two
// waves of a SIMD fill each other's stalls (dependent instructions, the LDS reads at the top of a
// step), which a single wave per SIMD with six step sizes per lane cannot.
template <typename T, class M, bool DIAG, int L, int A>
__global__ __launch_bounds__(256, L) void evaluate_ring_kernel(Params<T, M::NX, M::NU> p, int end) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[4 * EvalRing<T, M, L>::BYTES];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    evaluate_body<T, M, DIAG, false, L, A, true, false, false, true>(p, end, blockIdx.x * 4 + w, threadIdx.x & 63,
                                                                      ring + w * EvalRing<T, M, L>::BYTES);
}

// ---------------------------------------------------------------- evaluate, stage two
// Few problems reach stage two (a few per cent of the batch, from anywhere in it), so few waves
// run, each a serial chain over the horizon: the kernel is bound by latency, not by bandwidth.
// One wave of L lanes per problem x 64/L problems therefore stages the rows of SB time steps at a
// time through LDS: all lanes of a problem share the loads of a block (each value is fetched once
// and every load of the block is in flight together), the block after it is fetched into
// registers while the current one is rolled out, and the rollout reads its rows from LDS.
template <class M, bool BARRIER, int L, bool REROLL> struct ListedCfg {
    static constexpr int NX = M::NX, NU = M::NU, PPW = 64 / L;
    static constexpr int ROW = NU * NX + 2 * NU + (REROLL ? 1 : 2) * NX + (BARRIER ? 3 : 0); // values per time step
    static constexpr int FIT = 2560 / (PPW * ROW);                              // about 20 KB of fp64 per wave
    static constexpr int SB = FIT >= 16 ? 16 : FIT >= 8 ? 8 : FIT >= 4 ? 4 : 2;
    // per-problem LDS image of a block: K, k, U, X, Xref, centers, radius
    static constexpr int nK = SB * NU * NX, nk = SB * NU, nX = SB * NX, nC = BARRIER ? SB * 2 : 0, nD = BARRIER ? SB : 0;
    static constexpr int oK = 0, ok = oK + nK, oU = ok + nk, oX = oU + nk, oR = oX + (REROLL ? 0 : nX), oC = oR + nX, oD = oC + nC;
    static constexpr int TOTAL = oD + nD;
    static constexpr int STRIDE = TOTAL | 1; // odd: the problems of a wave read from different banks
    static constexpr int LDS = PPW * STRIDE;
    static constexpr int cnt(int n) { return (n + L - 1) / L; } // values each lane fetches of an array of n
};

template <typename T, class M, bool BARRIER, int L, bool REROLL> struct ListedStage { // one block in registers
    using C = ListedCfg<M, BARRIER, L, REROLL>;
    T K[C::cnt(C::nK)], k[C::cnt(C::nk)], U[C::cnt(C::nk)], X[REROLL ? 1 : C::cnt(C::nX)], R[C::cnt(C::nX)];
    T Cc[BARRIER ? C::cnt(C::nC) : 1], D[BARRIER ? C::cnt(C::nD) : 1];
};

// The block pipeline of stage two: global rows -> registers (one block ahead) -> LDS -> FwdRow.
template <typename T, class M, bool BARRIER, int L, bool REROLL> struct ListedPipe {
    using C = ListedCfg<M, BARRIER, L, REROLL>;
    using Stage = ListedStage<T, M, BARRIER, L, REROLL>;
    static constexpr int NX = M::NX, NU = M::NU;
    const T *Xt, *Rt, *Ut, *Kp, *kp, *Cp, *Dp;
    T *lds;
    int al, N, T1;
    // rows [r0, r0 + n) of an array with `rows` rows per problem: lane al takes every L-th; past
    // either end the nearest valid row is fetched instead (no branch around a load)
    // (V: rows interleaved per problem in this array, see GainLayout)
    template <int V = 1, class D> __device__ __forceinline__ void fetch(const T *src, int r0, int n, int rows, D &dst) const {
#pragma unroll
        for (int i = 0; i < C::cnt(n); ++i) {
            int e = al + L * i;
            e = e < n ? e : n - 1;
            int r = r0 + e;
            r = r < rows ? r : rows - 1;
            dst[i] = src[vrow<V>(r)];
        }
    }
    template <class D> __device__ __forceinline__ void put(int off, int n, const D &src) const {
#pragma unroll
        for (int i = 0; i < C::cnt(n); ++i) {
            int e = al + L * i;
            e = e < n ? e : n - 1;
            lds[off + e] = src[i];
        }
    }
    __device__ __forceinline__ void fetch_block(int t0, Stage &st) const {
        fetch<GainLayout<M>::VK>(Kp, t0 * NU * NX, C::nK, T1 * NU * NX, st.K);
        fetch<GainLayout<M>::Vk>(kp, t0 * NU, C::nk, T1 * NU, st.k);
        fetch(Ut, t0 * NU, C::nk, T1 * NU, st.U);
        if (!REROLL) fetch(Xt, t0 * NX, C::nX, N * NX, st.X);
        fetch<TargetLayout<M, BARRIER>::V>(Rt, t0 * NX, C::nX, N * NX, st.R);
        if (BARRIER) {
            fetch(Cp, t0 * 2, C::nC, N * 2, st.Cc);
            fetch(Dp, t0, C::nD, N, st.D);
        }
    }
    __device__ __forceinline__ void put_block(const Stage &st) const {
        put(C::oK, C::nK, st.K);
        put(C::ok, C::nk, st.k);
        put(C::oU, C::nk, st.U);
        if (!REROLL) put(C::oX, C::nX, st.X);
        put(C::oR, C::nX, st.R);
        if (BARRIER) {
            put(C::oC, C::nC, st.Cc);
            put(C::oD, C::nD, st.D);
        }
    }
    __device__ __forceinline__ void read_row(int s, FwdRow<T, NX, NU> &r) const {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            r.k[i] = lds[C::ok + s * NU + i];
            r.up[i] = lds[C::oU + s * NU + i];
#pragma unroll
            for (int j = 0; j < NX; ++j) r.K[i][j] = lds[C::oK + (s * NU + i) * NX + j];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (!REROLL) r.xp[i] = lds[C::oX + s * NX + i];
            r.xr[i] = lds[C::oR + s * NX + i];
        }
        if (BARRIER) {
            r.cx = lds[C::oC + s * 2 + 0];
            r.cy = lds[C::oC + s * 2 + 1];
            r.rad = lds[C::oD + s];
        } else {
            r.cx = r.cy = r.rad = T(0);
        }
    }
};

template <typename T, class M, bool DIAG, bool BARRIER, int L, bool REROLL, bool FAST, class P>
__device__ __forceinline__ void listed_rounds(const P &p, const ListedPipe<T, M, BARRIER, L, REROLL> &pipe, bool active, T Jcur, int base0,
                                              int end, int &best, T &best_cost, bool &bad) {
    using C = ListedCfg<M, BARRIER, L, REROLL>;
    constexpr int NX = M::NX, NU = M::NU, SB = C::SB;
    const int T1 = p.N - 1, al = pipe.al;
    best = 0x7fffffff;
    best_cost = T(0);
    for (int base = base0; base < end; base += L) {
        const int a = base + al;
        const T alpha = pow2_neg<T>(a);
        T x[NX], cost = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = pipe.Xt[i * TILE];
        T xq[NX]; // state of the previous trajectory (REROLL)
#pragma unroll
        for (int i = 0; i < NX; ++i) xq[i] = x[i];
        typename ListedPipe<T, M, BARRIER, L, REROLL>::Stage st;
        pipe.fetch_block(0, st);
        for (int t0 = 0; t0 < T1; t0 += SB) {
            __syncthreads(); // the rollout of the previous block has read its rows
            pipe.put_block(st);
            __syncthreads();
            pipe.fetch_block(t0 + SB, st); // in flight during the rollout below
            const int ns = T1 - t0 < SB ? T1 - t0 : SB;
            FwdRow<T, NX, NU> rn;
            pipe.read_row(0, rn);
            for (int s = 0; s < ns; ++s) {
                FwdRow<T, NX, NU> r = rn;
                pipe.read_row(s + 1 < SB ? s + 1 : s, rn); // the next step's row leaves LDS during this step
                if (REROLL) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) r.xp[i] = xq[i];
                }
                candidate_step<T, M, DIAG, BARRIER, FAST>(p, t0 + s, alpha, r, x, cost, bad);
                if (REROLL) {
                    T xqn[NX];
                    M::template step<FAST>(p, t0 + s, xq, r.up, xqn, bad);
#pragma unroll
                    for (int i = 0; i < NX; ++i) xq[i] = xqn[i];
                }
            }
        }
        T xr[NX], cx = T(0), cy = T(0), rad = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = pipe.Rt[target_x<M, BARRIER>(T1, i)];
        if (BARRIER) {
            cx = pipe.Cp[(T1 * 2 + 0) * TILE];
            cy = pipe.Cp[(T1 * 2 + 1) * TILE];
            rad = pipe.Dp[T1 * TILE];
        }
        const T c = candidate_total<T, M, DIAG, BARRIER>(p, x, xr, cx, cy, rad, cost);
        int cand = 0x7fffffff;
        T cand_cost = T(0);
        if (a < end && c < Jcur) { // strict test of ilqr.py:65; NaN never passes
            cand = a;
            cand_cost = c;
        }
#pragma unroll
        for (int off = 1; off < L; off <<= 1) { // lowest candidate index among the L lanes of the problem
            int oc = __shfl_xor(cand, off);
            T ocost = __shfl_xor(cand_cost, off);
            if (oc < cand) {
                cand = oc;
                cand_cost = ocost;
            }
        }
        if (best == 0x7fffffff && cand != 0x7fffffff) {
            best = cand;
            best_cost = cand_cost;
        }
        if (__ballot(active && best == 0x7fffffff) == 0ull) break;
    }
}

// The accepted step of the listed problems rolled out once more and written in place (what
// commit_kernel does for the others), through the same block pipeline: every lane of a problem
// computes the same rollout and its first lane stores.  A block's rows are in LDS and the next
// block's in registers before any row of the block is overwritten.
template <typename T, class M, bool BARRIER, int L, bool REROLL, bool FAST, class P>
__device__ __forceinline__ void listed_commit(const P &p, const ListedPipe<T, M, BARRIER, L, REROLL> &pipe, T *Xw, T *Uw, T alpha,
                                              bool store) {
    using C = ListedCfg<M, BARRIER, L, REROLL>;
    constexpr int NX = M::NX, NU = M::NU, SB = C::SB;
    const int T1 = p.N - 1;
    T x[NX], xq[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xq[i] = x[i] = pipe.Xt[i * TILE];
    typename ListedPipe<T, M, BARRIER, L, REROLL>::Stage st;
    pipe.fetch_block(0, st);
    bool bad = false;
    for (int t0 = 0; t0 < T1; t0 += SB) {
        __syncthreads();
        pipe.put_block(st);
        __syncthreads();
        pipe.fetch_block(t0 + SB, st);
        const int ns = T1 - t0 < SB ? T1 - t0 : SB;
        for (int s = 0; s < ns; ++s) {
            const int t = t0 + s;
            FwdRow<T, NX, NU> r;
            pipe.read_row(s, r);
            if (REROLL) {
#pragma unroll
                for (int i = 0; i < NX; ++i) r.xp[i] = xq[i];
            }
            T un[NU], xn[NX];
            control_update<FAST, T, M>(p, alpha, x, r.xp, r.up, r.K, r.k, un, bad);
            M::template step<FAST>(p, t, x, un, xn, bad);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if (REROLL) {
                T xqn[NX];
                M::template step<FAST>(p, t, xq, r.up, xqn, bad);
#pragma unroll
                for (int i = 0; i < NX; ++i) xq[i] = xqn[i];
            }
            if (store) {
#pragma unroll
                for (int i = 0; i < NU; ++i) Uw[(size_t)(t * NU + i) * TILE] = un[i];
#pragma unroll
                for (int i = 0; i < NX; ++i) Xw[(size_t)((t + 1) * NX + i) * TILE] = xn[i];
            }
        }
    }
}

// one wave per workgroup; the grid covers the worst case (every problem listed) and the waves
// beyond the end of the list leave at once.  COMMIT: the kernel also commits the outcome of its
// problems (state, input, cost, flags and history exactly as commit_kernel would), so that it
// can run beside the commit of all other problems instead of before it.
template <typename T, class M, bool DIAG, bool BARRIER, int L, bool REROLL, bool COMMIT>
__global__ __launch_bounds__(64) void evaluate_listed_kernel(Params<T, M::NX, M::NU> p, int base0, int end, int it) {
    using C = ListedCfg<M, BARRIER, L, REROLL>;
    constexpr int NX = M::NX, NU = M::NU, PPW = C::PPW;
    __shared__ T rows[C::LDS];
    const int lane = threadIdx.x;
    const int pl = lane / L, al = lane % L;
    const int nsl = *p.list_count;
    if ((int)blockIdx.x * PPW >= nsl) return;
    const int g = blockIdx.x * PPW + pl;
    const bool valid = g < nsl;
    const int b = p.list[valid ? g : nsl - 1]; // slot
    const int o = orig_of(p, b);
    const bool active = valid; // listed problems are active by construction
    const int T1 = p.N - 1;
    const T Jcur = p.J[o];
    const int tile = b >> 6, ln = b & 63;
    ListedPipe<T, M, BARRIER, L, REROLL> pipe;
    pipe.Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    pipe.Rt = tile_ptr_v<TargetLayout<M, BARRIER>::V>(p.Xref, tile, p.N * NX, ln);
    pipe.Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    pipe.Cp = BARRIER ? tile_ptr(p.centers, tile, p.N * 2, ln) : nullptr;
    pipe.Dp = BARRIER ? tile_ptr(p.radius, tile, p.N, ln) : nullptr;
    pipe.kp = tile_ptr_v<GainLayout<M>::Vk>(p.k, tile, T1 * NU, ln);
    pipe.Kp = tile_ptr_v<GainLayout<M>::VK>(p.K, tile, T1 * NU * NX, ln);
    pipe.lds = rows + pl * C::STRIDE;
    pipe.al = al;
    pipe.N = p.N;
    pipe.T1 = T1;
    int best;
    T best_cost;
    bool bad = false;
    listed_rounds<T, M, DIAG, BARRIER, L, REROLL, true>(p, pipe, active, Jcur, base0, end, best, best_cost, bad);
    bool pbad = bad; // as in evaluate_kernel
#pragma unroll
    for (int w = 1; w < L; w <<= 1) pbad |= __shfl_xor((int)pbad, w) != 0;
    if (__ballot(pbad && active) != 0ull) {
        int best2;
        T best_cost2;
        bool bad2 = false;
        listed_rounds<T, M, DIAG, BARRIER, L, REROLL, false>(p, pipe, active, Jcur, base0, end, best2, best_cost2, bad2);
        best = pbad ? best2 : best;
        best_cost = pbad ? best_cost2 : best_cost;
    }
    const bool found = best != 0x7fffffff;
    if (!COMMIT) {
        if (active && al == 0) {
            p.alpha_idx[o] = found ? best : -1;
            p.Jtry[o] = best_cost;
            p.slow_trig[o] = (p.slow_trig[o] & (6 | SLOW_TRIG_REGEN)) | (pbad ? 1 : 0);
        }
        return;
    }
    // commit (mirrors commit_kernel for these problems)
    const bool roll = active && found;
    const bool slow = roll && pbad;
    if (active && al == 0) {
        const int idx = found ? best : -1;
        p.alpha_idx[o] = idx;
        p.Jtry[o] = best_cost;
        const int bits = (p.slow_trig[o] & 6) | (pbad ? 1 : 0);
        const int committed = roll ? (4 | (slow ? 2 : 0)) : 0;
        p.slow_trig[o] = REROLL ? (bits | committed) : ((bits & 1) | committed);
        p.iters[o] += 1;
        const T Jn = roll ? best_cost : Jcur;
        if (roll) {
            p.J[o] = Jn;
            if (p.status && !(fabs_(Jn) < inf_<T>())) p.status[o] |= ILQR_STATUS_NONFINITE_;
        } else {
            p.converged[o] = 1;
        }
        if (it < p.hist_cap) {
            if (p.hist_cost) p.hist_cost[size_t(it) * p.hist_stride + o] = Jn;
            if (p.hist_alpha) p.hist_alpha[size_t(it) * p.hist_stride + o] = idx;
        }
    }
    if (__ballot(roll) == 0ull) return;
    const T alpha = pow2_neg<T>(roll ? best : 0);
    T *Xw = tile_ptr(p.X, tile, p.N * NX, ln), *Uw = tile_ptr(p.U, tile, T1 * NU, ln);
    // per problem, as in commit_kernel: the library-routine sweep writes only the problems that need it
    if (__ballot(roll && !slow) != 0ull) listed_commit<T, M, BARRIER, L, REROLL, true>(p, pipe, Xw, Uw, alpha, roll && !slow && al == 0);
    if (__ballot(slow) != 0ull) listed_commit<T, M, BARRIER, L, REROLL, false>(p, pipe, Xw, Uw, alpha, slow && al == 0);
}

// ---------------------------------------------------------------- line search, n = 12, m = 4
// One wave per problem on the matrix unit, all 16 step sizes at once: the candidates are the
// columns of a 16 x 16 tile Z = [x; u] (rows 0..11 states, 12..15 inputs; layouts: see Mfma4 and
// ltv12x4_backward_sweep), so lane (c, g) carries candidate c and the components g, g+4, g+8 of
// its state and input g.  A time step is X+ = [A_t | B_t] Z (four calls) after the feedback
// K (X - x_prev) (three calls, result row g = input g in the lane that holds that input).  Each
// lane accumulates its share of the candidate's cost; the four lanes of a candidate are summed at
// the end.  Sums run in the matrix unit's order: costs and the accepted step agree with the
// lane-per-candidate kernel to rounding, not bit for bit.
//
// A workgroup is 16 waves = 16 neighbouring problems of a tile.  A problem's share of a row is 8
// (4) bytes of a line that the 16 problems share, and a wave that loads its own operands touches
// 16 to 64 lines per load instruction, which the texture-address unit works through one line at a
// time (measured: the kernel ran at the speed of that unit, not of the matrix unit or of HBM).  So
// the workgroup fetches the 80 rows of a time step (K, k, x, u, xref: one line each for its 16
// problems) and that step's [A_t | B_t] (shared by all problems) as whole lines straight into LDS
// (12 wave instructions of 1 KiB per step in fp64 instead of 240 scattered ones), LTV_RING steps
// ahead, and every wave takes its operands from there.
//
// LDS image of one step: 16-byte granules in the order the loading lanes hold them (the direct
// load writes lane i of a wave at base + 16 i), rotated within each row (LtvStage): a wave reads
// the same problem of 16 different rows in one instruction, and the rotation puts those 16 on 16
// different bank groups.
//
// PW problems per wave (workgroup = 16 / PW waves): the chains of a wave's problems are independent,
// so the instructions of one fill the waits of the other (a matrix instruction's result is ready 4
// to 16 passes after it issues; with one problem per wave and four waves per SIMD the SIMD sat idle
// a third of the time).  The problems of a wave are neighbours inside one 16-byte granule of every
// row, so one set of addresses serves them.
#ifndef LTV_EVALUATE_PW
#define LTV_EVALUATE_PW 2
#endif
template <typename T, int PW = LTV_EVALUATE_PW> __global__ __launch_bounds__(1024 / PW, 4) void ltv12x4_evaluate_mfma_kernel(Params<T, 12, 4> p) {
    using St = LtvStage<T>;
    using MF = Mfma4<T>;
    using acc = typename MF::acc;
    constexpr int NX = 12, NU = 4, D = LTV_EVALUATE_RING, S = LTV_EVALUATE_GROUP, ES = (int)sizeof(T), GPR = St::GPR, PPG = St::PPG;
    constexpr int ROWS = 80; // rows of a step: K 0..47, k 48..51, x 52..63, u 64..67, xref 68..79
    constexpr int AGR = NX * NX * ES / 16, TBL = NX * (NX + NU) * ES / 16;
    constexpr int NLOAD = (ROWS * GPR + TBL + 63) / 64, CHUNK = NLOAD * 1024;
    constexpr int NW = 16 / PW, LPW = (NLOAD + NW - 1) / NW; // waves; load instructions per step of a wave (the first waves: LPW, the others LPW - 1)
    static_assert(PPG % PW == 0 && 16 % PW == 0, "a wave's problems share a granule");
    __shared__ __attribute__((aligned(16))) unsigned char ring[D * S * CHUNK];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int ci = MF::tau(c);                                      // the row of a left factor this lane supplies
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave: problems PW w .. PW w + PW - 1 of the workgroup
    const int nsl = slot_count(p);
    const int slot0 = blockIdx.x * 16;
    int o[PW];
    bool act[PW], any = false;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int slot_raw = slot0 + PW * w + j;
        const int slot = slot_raw < nsl ? slot_raw : (nsl > 0 ? nsl - 1 : 0);
        o[j] = orig_of(p, slot);
        act[j] = slot_raw < nsl && p.converged[o[j]] == 0;
        any = any || act[j];
    }
    if (!__syncthreads_or(any)) return; // no wave of this workgroup has work
    const int T1 = p.N - 1;
    const int tile = slot0 >> 6, ln0 = slot0 & 63; // the workgroup's 16 problems: lanes ln0 .. ln0 + 15 of one tile
    const T *Xt = tile_ptr(p.X, tile, p.N * NX, ln0);
    const T *Rt = tile_ptr(p.Xref, tile, p.N * NX, ln0);
    const T alpha = pow2_neg<T>(c);
    T qw[3], qfw[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        qw[r] = p.Q[(g + 4 * r) * NX + g + 4 * r];
        qfw[r] = p.Qf[(g + 4 * r) * NX + g + 4 * r];
    }
    const T rw = p.R[g * NU + g], lo = p.lo[g], hi = p.hi[g];
    // candidates start from the stored first state
    T z[PW][4], cost[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
#pragma unroll
        for (int r = 0; r < 3; ++r) z[j][r] = Xt[(size_t)(g + 4 * r) * TILE + PW * w + j];
        z[j][3] = T(0);
        cost[j] = T(0);
    }
    // this lane's granules of every step (load instruction number w + i NW of the step): source
    // address at step 0 and stride per step
    const unsigned char *src[LPW];
    unsigned stride[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int q = (w + i * NW) * 64 + lane, row = q / GPR;
        if (q < ROWS * GPR) {
            const T *b;
            int r, rows;
            if (row < 48) b = tile_ptr(p.K, tile, T1 * NU * NX, ln0), r = row, rows = NU * NX;
            else if (row < 52) b = tile_ptr(p.k, tile, T1 * NU, ln0), r = row - 48, rows = NU;
            else if (row < 64) b = Xt, r = row - 52, rows = NX;
            else if (row < 68) b = tile_ptr(p.U, tile, T1 * NU, ln0), r = row - 64, rows = NU;
            else b = Rt, r = row - 68, rows = NX;
            src[i] = (const unsigned char *)(b + (size_t)r * TILE + St::first_problem(q));
            stride[i] = (unsigned)(rows * TILE * ES);
        } else {
            const int e = q - ROWS * GPR < TBL ? q - ROWS * GPR : TBL - 1; // (padding lanes repeat the last granule)
            src[i] = (const unsigned char *)(e < AGR ? p.ltvA + PPG * e : p.ltvB + PPG * (e - AGR));
            stride[i] = (unsigned)((e < AGR ? NX * NX : NX * NU) * ES);
        }
    }
    auto stage = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            if (w + i * NW < NLOAD) glds16(src[i] + (size_t)t * stride[i], ring + buf * CHUNK + (w + i * NW) * 1024);
    };
    const bool full_loader = w + (LPW - 1) * NW < NLOAD; // this wave issues LPW loads per step (else LPW - 1)
    // where this lane's operands sit in a step's image (first problem of the wave; the next ones
    // follow at the element size).  The matrix instruction's left operand of a lane is one row of
    // the left factor: rows >= 4 of K and rows >= 12 of [A_t | B_t] only produce result rows nobody
    // reads, so those lanes simply read a valid row.
    unsigned oxp[3], oxr[3], oK[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        oxp[q] = St::off(52 + g + 4 * q, PW * w);
        oxr[q] = St::off(68 + g + 4 * q, PW * w);
        oK[q] = St::off((ci & 3) * NX + g + 4 * q, PW * w);
    }
    const unsigned oup = St::off(64 + g, PW * w), okf = St::off(48 + g, PW * w);
    const int cc = ci < NX ? ci : 0;
    const unsigned oA = ROWS * GPR * 16 + (cc * NX + g) * ES, oB = ROWS * GPR * 16 + NX * NX * ES + (cc * NU + g) * ES;
    const unsigned oA4 = ROWS * GPR * 16 + ((c & 3) * NX + g) * ES, oB4 = ROWS * GPR * 16 + NX * NX * ES + ((c & 3) * NU + g) * ES;
    auto one_step = [&](const unsigned char *img) {
        auto at = [&](unsigned off) { return *(const T *)(img + off); };
        // the table operands are the same for the wave's problems
        T FA[MF::HAS4 ? 12 : 4];
        if constexpr (MF::HAS4) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) FA[r * 4 + kk] = at(oA4 + (4 * r * NX + 4 * kk) * ES);
                FA[r * 4 + 3] = at(oB4 + 4 * r * NU * ES);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) FA[q] = at(oA + 4 * ES * q);
            FA[3] = at(oB);
        }
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            T dx[3], e[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                dx[q] = z[j][q] - at(oxp[q] + j * ES);
                e[q] = z[j][q] - at(oxr[q] + j * ES);
            }
            typename MF::acc4 KD = MF::zero4();
#pragma unroll
            for (int q = 0; q < 3; ++q) KD = MF::run4(at(oK[q] + j * ES), dx[q], KD);
            // u_new = (u_prev + alpha k) + K (x_new - x_prev), clamped (ilqr.py:56-61)
            T v = fma_(alpha, at(okf + j * ES), at(oup + j * ES)) + MF::first(KD);
            v = (lo > v) ? lo : v;
            v = (hi < v) ? hi : v;
            z[j][3] = v;
            T stage_cost = (qw[0] * e[0]) * e[0];
            stage_cost = fma_(qw[1] * e[1], e[1], stage_cost);
            stage_cost = fma_(qw[2] * e[2], e[2], stage_cost);
            stage_cost = fma_(rw * v, v, stage_cost);
            cost[j] += stage_cost;
            if constexpr (MF::HAS4) {
                // X+ = [A_t | B_t] Z by four-row products: rows g, g + 4, g + 8 of the result are three
                // chains of four calls (12 x 16 cycles instead of 4 x 64: the 16-row form computes four rows
                // nobody reads).  Left operand of lane (c, g) for rows r, step kk: F[c % 4 + 4 r][g + 4 kk].
                typename MF::acc4 xn[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    xn[r] = MF::zero4();
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) xn[r] = MF::run4(FA[r * 4 + kk], z[j][kk], xn[r]);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) z[j][r] = MF::first(xn[r]);
            } else {
                acc Xn = {T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int q = 0; q < 4; ++q) Xn = MF::run(FA[q], z[j][q], Xn);
#pragma unroll
                for (int q = 0; q < 3; ++q) z[j][q] = Xn[q];
            }
        }
    };
    // The steps move through the ring in groups of S: group m (steps m S .. m S + S - 1) is computed
    // from ring slot m mod D while the groups m + 1 .. m + D - 1 are on their way.  A direct load is
    // ordered before a read of its data only by the loading wave's counted wait followed by a
    // barrier, so per group: wait until this wave's pieces of the group have landed, meet (every
    // piece has landed, and every wave is done with the previous group), refill the slot that group
    // used, compute.
    const int NM = (T1 + S - 1) / S;
    auto stage_group = [&](int m, int slot) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            const int t = m * S + u;
            stage(t < T1 ? t : T1 - 1, slot * S + u);
        }
    };
#pragma unroll
    for (int m = 0; m < D - 1; ++m) stage_group(m, m);
    for (int mb = 0; mb < NM; mb += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int m = mb + s;
            if (m >= NM) break; // uniform over the workgroup
            if (full_loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * S * LPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * S * (LPW - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
            stage_group(m + D - 1, (s + D - 1) % D);
            if (any) {
#pragma unroll
                for (int u = 0; u < S; ++u) {
                    if (m * S + u >= T1) break;
                    one_step(ring + (s * S + u) * CHUNK);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        T cj = cost[j];
        {   // terminal term (ilqr.py:44-45)
            T term = T(0);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const T e = z[j][q] - Rt[(size_t)(T1 * NX + g + 4 * q) * TILE + PW * w + j];
                term = fma_(qfw[q] * e, e, term);
            }
            cj += term;
        }
        const T Jcur = p.J[o[j]];
        cj = add_lane_groups(cj);
        // lowest step index whose cost improves strictly (ilqr.py:65; NaN never passes)
        const unsigned long long win = __ballot(g == 0 && c < p.n_alphas && cj < Jcur);
        const int best = win ? __ffsll((long long)win) - 1 : -1;
        const T best_cost = __shfl(cj, best < 0 ? 0 : best);
        if (act[j] && lane == 0) {
            p.alpha_idx[o[j]] = best;
            p.Jtry[o[j]] = best < 0 ? T(0) : best_cost;
            p.slow_trig[o[j]] = p.slow_trig[o[j]] & 6;
        }
    }
}

// ---------------------------------------------------------------- commit
// One lane per problem.  Rolls the accepted step size out once more and
// overwrites X and U in place (each lane reads a row before it overwrites it);
// problems without an accepted step keep their trajectory and become converged
// (ilqr.py:69-73).
// REROLL (see evaluate_rounds): the previous states are rolled out again from the stored inputs
// instead of being read.  A lane that must leave its trajectory unchanged (converged problem,
// no accepted step, padding) then has no old state values to write back.  If its stored states
// are known to be this code's rollout of its inputs (`keep_ok`), the recomputed states are the
// stored bits and are written back, which keeps every store a full line; otherwise its state
// stores go to a scratch slot `trash` (still no branch around the stores).
// `mute`: this lane belongs to the other sweep of the kernel (fast / library trigonometry): all
// its stores go to the scratch slot.
// TC: the sine/cosine pair of every step of the trajectory being written goes to SCt (Params::SC),
// and with REROLL the old trajectory's pairs are read from there first.
// PF: rows of the next PF time steps are in flight while a step is computed.
// LAZY (with REROLL, fast sweep only): of the states only the rows XCK, 2 XCK .. are stored.
template <typename T, class M, int PF, bool REROLL, bool FAST, bool TC = false, bool LAZY = false, class P>
__device__ __forceinline__ void commit_sweep(const P &p, T *Xt, T *Ut, const T *Kp, const T *kp, T *trash, T alpha, bool roll,
                                             bool keep_ok, bool pend, bool mute, T *SCt = nullptr) {
    constexpr int NX = M::NX, NU = M::NU;
    const int T1 = p.N - 1;
    struct Row {
        T K[NU][NX], k[NU], xp[NX], up[NU], s, c;
    };
    auto load_row = [&](int t, Row &r) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            r.k[i] = kp[gain_k<M>(t, i)];
            r.up[i] = Ut[(t * NU + i) * TILE];
#pragma unroll
            for (int j = 0; j < NX; ++j) r.K[i][j] = Kp[gain_K<M>(t * NU + i, j)];
        }
        if (!REROLL) {
#pragma unroll
            for (int i = 0; i < NX; ++i) r.xp[i] = Xt[(t * NX + i) * TILE];
        }
        if (REROLL && TC) {
            r.s = SCt[(t * 2 + 0) * TILE];
            r.c = SCt[(t * 2 + 1) * TILE];
        } else {
            r.s = r.c = T(0);
        }
    };
    T x[NX], xq[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xq[i] = x[i] = Xt[i * TILE];
    // every row is read before any store reaches it: the ring holds rows t .. t+PF-1 while step t
    // writes rows t (inputs, pairs) and t+1 (states, which nothing but the !REROLL ring reads, ahead of the store)
    Row ring[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (s < T1) load_row(s, ring[s]);
    T xlast[NX]; // previous states of the terminal step, needed to keep lanes without a winner unchanged
    if (!REROLL) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xlast[i] = Xt[(T1 * NX + i) * TILE];
    }
    bool bad = false, badq = false;
    static_assert(!LAZY || (REROLL && FAST && XCK % PF == 0), "sparse states: fast sweep over recomputed previous states");
    constexpr int UNR = LAZY ? XCK : PF; // steps per trip: which step stores a state row is then known at compile time
    for (int tb = 0; tb < T1; tb += UNR) {
#pragma unroll
        for (int sl = 0; sl < UNR; ++sl) {
            const int t = tb + sl;
            if (t >= T1) break;
            Row r = ring[sl % PF];
            if (t + PF < T1) load_row(t + PF, ring[sl % PF]);
            const Row &nxt = ring[(sl + 1) % PF]; // row t+1 (if there is one)
            if (REROLL) {
#pragma unroll
                for (int i = 0; i < NX; ++i) r.xp[i] = xq[i];
            }
            T un[NU], xn[NX], sn = T(0), cn = T(0);
            control_update<FAST, T, M>(p, alpha, x, r.xp, r.up, r.K, r.k, un, bad);
            model_step<FAST, false, M>(p, t, x, un, xn, sn, cn, bad);
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            if (REROLL) {
                T xqn[NX];
                model_step<FAST, TC, M>(p, t, xq, r.up, xqn, r.s, r.c, badq);
#pragma unroll
                for (int i = 0; i < NX; ++i) xq[i] = xqn[i];
            }
            // Every lane stores; a lane whose problem has no accepted step writes back the values it
            // read (no branch around the stores: see backward_sweep)
            // `pend`: the problem is being decided and committed by the second line-search stage,
            // which may run beside this kernel: nothing of it may be written here
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                T *dst = (pend || mute) ? trash : &Ut[(t * NU + i) * TILE];
                *dst = roll ? un[i] : r.up[i];
            }
            if (TC) {
                // without old pairs at hand (!REROLL) only the lanes that roll have something to store
                T *ds = (pend || mute || (!REROLL && !roll)) ? trash : &SCt[(t * 2 + 0) * TILE];
                T *dc = (pend || mute || (!REROLL && !roll)) ? trash : &SCt[(t * 2 + 1) * TILE];
                *ds = (roll || !REROLL) ? sn : r.s;
                *dc = (roll || !REROLL) ? cn : r.c;
            }
            // the recomputed old states are the stored ones only while the fast trigonometry was in
            // range (the library routine may differ from it in the last bit)
            const bool real = !mute && (roll || (!pend && FAST && keep_ok && !badq));
            if (LAZY && sl != XCK - 1) continue; // row t+1 is not one of the rows a sparse commit keeps
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if (REROLL) {
                    T *dst = real ? &Xt[((t + 1) * NX + i) * TILE] : trash;
                    *dst = roll ? xn[i] : xq[i];
                } else {
                    T *dst = (pend || mute) ? trash : &Xt[((t + 1) * NX + i) * TILE];
                    *dst = roll ? xn[i] : (t + 1 < T1 ? nxt.xp[i] : xlast[i]);
                }
            }
        }
    }
}

template <typename T, class M, int PF, bool REROLL, bool TC = false, bool LAZY = false>
__global__ __launch_bounds__(64) void commit_kernel(Params<T, M::NX, M::NU> p, int it) {
    constexpr int NX = M::NX, NU = M::NU;
    const int b0 = blockIdx.x * 64 + threadIdx.x;
    const int nsl = slot_count(p);
    if (nsl <= 0) return;
    const bool valid = b0 < nsl;
    const int b = orig_of(p, valid ? b0 : nsl - 1); // index into the per-problem arrays
    const bool active = valid && p.converged[b] == 0;
    if (__ballot(active) == 0ull) return;
    const int T1 = p.N - 1;
    const int tile = b0 >> 6, ln = b0 & 63; // buffers are padded to whole tiles
    T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    const T *kp = tile_ptr_v<GainLayout<M>::Vk>(p.k, tile, T1 * NU, ln);
    const T *Kp = tile_ptr_v<GainLayout<M>::VK>(p.K, tile, T1 * NU * NX, ln);
    T *trash = p.trash + b0;
    const int idx = p.alpha_idx[b];
    const bool pend = active && idx == -2; // listed for the second line-search stage, which also commits it
    const bool roll = active && idx >= 0;
    // slow_trig bits: 1 = this line search needed the library sine/cosine (set by the evaluate
    // kernels); 2 = some step of this problem was committed by the library-routine sweep;
    // 4 = a step of this problem has been committed since the caller last wrote its trajectory.
    // The first commit after that (always the variant that reads X) defines bits 2 and 4 for
    // every problem, so nothing depends on what the workspace held before.
    const int slow_bits = p.slow_trig[b];
    const bool slow = roll && (slow_bits & 1) != 0;
    // stored states = this code's fast-path rollout of the stored inputs
    const bool keep_ok = !valid || (slow_bits & 6) == 4;
    if (valid && !pend) {
        // (SLOW_TRIG_REGEN: this commit's trajectory is the fast-path rollout of its inputs; the sweep with
        // the library routine writes every state row, and a problem that is not rolled out keeps its mark)
        const int committed = roll ? (4 | (slow ? 2 : 0)) : 0, regen = (p.lazy && roll && !slow) ? SLOW_TRIG_REGEN : 0;
        if (!REROLL)
            p.slow_trig[b] = (slow_bits & 1) | committed | regen;
        else if (roll)
            p.slow_trig[b] = ((slow_bits | committed) & ~SLOW_TRIG_REGEN) | regen;
    }
    if (active && !pend) {
        p.iters[b] += 1;
        T Jn = roll ? p.Jtry[b] : p.J[b];
        if (roll) {
            p.J[b] = Jn;
            if (p.status && !(fabs_(Jn) < inf_<T>())) p.status[b] |= ILQR_STATUS_NONFINITE_;
        } else {
            p.converged[b] = 1;
        }
        if (it < p.hist_cap) {
            if (p.hist_cost) p.hist_cost[size_t(it) * p.hist_stride + b] = Jn;
            if (p.hist_alpha) p.hist_alpha[size_t(it) * p.hist_stride + b] = idx;
        }
    }
    if (__ballot(roll) == 0ull) return;
    const T alpha = pow2_neg<T>(roll ? idx : 0);
    // The trajectory is overwritten in place, so the choice between the fast trigonometry and the
    // library routine is made up front and per problem, from what the evaluate kernel saw on it:
    // the fast sweep writes nothing for a problem that needs the library routine, and the second
    // sweep (rare) writes only those, so a problem's result does not depend on its wave-mates.
    T *SCt = TC ? tile_ptr(p.SC, tile, T1 * 2, ln) : nullptr;
    commit_sweep<T, M, PF, REROLL, true, TC, LAZY>(p, Xt, Ut, Kp, kp, trash, alpha, roll && !slow, keep_ok, pend, slow, SCt);
    if (__ballot(slow) != 0ull)
        commit_sweep<T, M, 1, REROLL, false, TC>(p, Xt, Ut, Kp, kp, trash, alpha, slow, keep_ok, pend, !slow, SCt);
}

// ---------------------------------------------------------------- commit, n = 12, m = 4
// The accepted step size rolled out with 16 problems per wave: the problems are the columns of the
// tile Z = [x; u] (lane (c, g): problem c of the wave, state components g, g+4, g+8 and input g),
// so X+ = [A_t | B_t] Z is four matrix instructions for 16 problems (the tables are the same for
// all problems), and the 16 lanes of a lane group read and write 16 neighbouring problems of a row:
// whole lines.  The feedback K (x - x_prev) has a gain matrix per problem and stays on the vector
// unit: every lane multiplies its three components, the four lanes of a problem add up.
// A workgroup is the four waves of one tile.  (One lane per problem leaves 512 waves for 32768
// problems, half a wave per SIMD, each with a 12 x 12 product per step.)
// Lanes whose problem keeps its trajectory write back what they read: every store is a full line.
// `first`: the caller wrote the trajectories since the last commit (see commit_kernel, slow_trig).
template <typename T> __global__ __launch_bounds__(256) void ltv12x4_commit_mfma_kernel(Params<T, 12, 4> p, int it, int first) {
    using MF = Mfma4<T>;
    using acc = typename MF::acc;
    constexpr int NX = 12, NU = 4, PF = 2;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int w = threadIdx.x >> 6;
    const int nsl = slot_count(p);
    if (nsl <= 0) return;
    const int tile = blockIdx.x, ln = 16 * w + c;
    const int b0 = tile * 64 + ln;
    const bool valid = b0 < nsl;
    const int b = orig_of(p, valid ? b0 : nsl - 1);
    const bool active = valid && p.converged[b] == 0;
    if (__ballot(active) == 0ull) return;
    const int T1 = p.N - 1;
    const int idx = p.alpha_idx[b];
    const bool pend = active && idx == -2;
    const bool roll = active && idx >= 0;
    if (g == 0) { // per-problem outcome, as in commit_kernel
        const int slow_bits = p.slow_trig[b];
        if (valid && !pend) {
            const int committed = roll ? 4 : 0;
            if (first)
                p.slow_trig[b] = (slow_bits & 1) | committed;
            else if (roll)
                p.slow_trig[b] = slow_bits | committed;
        }
        if (active && !pend) {
            p.iters[b] += 1;
            const T Jn = roll ? p.Jtry[b] : p.J[b];
            if (roll) {
                p.J[b] = Jn;
                if (p.status && !(fabs_(Jn) < inf_<T>())) p.status[b] |= ILQR_STATUS_NONFINITE_;
            } else {
                p.converged[b] = 1;
            }
            if (it < p.hist_cap) {
                if (p.hist_cost) p.hist_cost[size_t(it) * p.hist_stride + b] = Jn;
                if (p.hist_alpha) p.hist_alpha[size_t(it) * p.hist_stride + b] = idx;
            }
        }
    }
    if (__ballot(roll) == 0ull) return;
    T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    const T *kp = tile_ptr(p.k, tile, T1 * NU, ln);
    const T *Kp = tile_ptr(p.K, tile, T1 * NU * NX, ln);
    const T alpha = pow2_neg<T>(roll ? idx : 0);
    const T lo = p.lo[g], hi = p.hi[g];
    // left operand of X+ = [A_t | B_t] Z: this lane supplies row ci of the tables (rows >= 12 only
    // produce result rows nobody reads: a valid row is read instead)
    const int ci = MF::tau(c), cc = ci < NX ? ci : 0;
    const T *Ap = p.ltvA + cc * NX + g, *Bp = p.ltvB + cc * NU + g;
    struct Row {
        T xp[3], K[NU][3], up, kf, F[4];
    };
    // (row t of the gains, inputs and tables, t <= T1 - 1; of the states row min(t, T1), so that the
    // row after the last step's is the final state)
    auto load_row = [&](int tt, Row &r) {
        const int t = tt < T1 ? tt : T1 - 1;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            r.xp[q] = Xt[(size_t)((tt < T1 ? tt : T1) * NX + g + 4 * q) * TILE];
            r.F[q] = Ap[(size_t)t * (NX * NX) + 4 * q];
#pragma unroll
            for (int i = 0; i < NU; ++i) r.K[i][q] = Kp[(size_t)((t * NU + i) * NX + g + 4 * q) * TILE];
        }
        r.F[3] = Bp[(size_t)t * (NX * NU)];
        r.up = Ut[(size_t)(t * NU + g) * TILE];
        r.kf = kp[(size_t)(t * NU + g) * TILE];
    };
    T x[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) x[q] = Xt[(size_t)(g + 4 * q) * TILE];
    // every row is read before any store reaches it: the ring holds rows t .. t+PF-1 while step t
    // writes the inputs of row t and the states of row t+1
    Row ring[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) load_row(s, ring[s]);
    for (int tb = 0; tb < T1; tb += PF) {
#pragma unroll
        for (int sl = 0; sl < PF; ++sl) {
            const int t = tb + sl;
            if (t >= T1) break;
            const Row r = ring[sl];
            load_row(t + PF, ring[sl]);
            const Row &nxt = ring[(sl + 1) % PF]; // row t+1 (its states are the final state after the last step)
            // u_new = (u_prev + alpha k) + K (x_new - x_prev), clamped (ilqr.py:56-61)
            T dx[3], fb[NU];
#pragma unroll
            for (int q = 0; q < 3; ++q) dx[q] = x[q] - r.xp[q];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                fb[i] = r.K[i][0] * dx[0];
                fb[i] = fma_(r.K[i][1], dx[1], fb[i]);
                fb[i] = fma_(r.K[i][2], dx[2], fb[i]);
                fb[i] += __shfl_xor(fb[i], 16);
                fb[i] += __shfl_xor(fb[i], 32);
            }
            T v = fma_(alpha, r.kf, r.up) + (g == 0 ? fb[0] : g == 1 ? fb[1] : g == 2 ? fb[2] : fb[3]);
            v = (lo > v) ? lo : v;
            v = (hi < v) ? hi : v;
            acc Xn = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int q = 0; q < 3; ++q) Xn = MF::run(r.F[q], x[q], Xn);
            Xn = MF::run(r.F[3], v, Xn);
            Ut[(size_t)(t * NU + g) * TILE] = roll ? v : r.up;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                x[q] = Xn[q];
                Xt[(size_t)((t + 1) * NX + g + 4 * q) * TILE] = roll ? Xn[q] : nxt.xp[q];
            }
        }
    }
}

// ---------------------------------------------------------------- commit, all segments at once
// The rollout of the accepted step size is a recurrence over the horizon, and with one lane per
// problem a small batch leaves most of the chip idle for its whole length.  With the candidates'
// states at the segment starts at hand (Params::CK, kept by the line search) every segment of
// COMMIT_SEG time steps can be rolled out on its own: wave (tile, segment) starts from the accepted
// candidate's checkpoint and executes the instruction sequence of commit_sweep for its steps, so
// the trajectory comes out bit for bit as from the sequential kernel, after COMMIT_SEG dependent
// steps instead of N-1.  Segment `seg` owns the time steps [t0, t1): it writes the inputs (and the
// sine/cosine pairs) of those steps and the states of the time steps t0 .. t1-1 (t0 from its
// checkpoint; the last segment also the terminal state), and reads the previous states of the
// same steps before it overwrites them: no row is shared between waves.
// The wave of segment 0 also does the per-problem bookkeeping of commit_kernel.
template <typename T, class M, bool FAST, bool TC, int RING = 4, class P>
__device__ __forceinline__ void commit_segment_sweep(const P &p, T *Xt, T *Ut, const T *Kp, const T *kp, T *SCt, T alpha, const T (&xs)[M::NX],
                                                     int t0, int t1, bool store, bool first) {
    constexpr int NX = M::NX, NU = M::NU, S = COMMIT_SEG;
    static_assert(S % RING == 0, "the segment length must be a multiple of the prefetch ring");
    const int T1 = p.N - 1;
    struct Row {
        T K[NU][NX], k[NU], xp[NX], up[NU];
    };
    // a wave lives for S steps only, so nothing but its own loads in flight hides the memory latency:
    // the rows of the next RING steps are always under way, and no load sits behind a branch (past
    // the end of the segment its last row is fetched again)
    auto load_row = [&](int tt, Row &r) {
        const int t = tt < t1 ? tt : t1 - 1;
        // one row pointer per array and compile-time offsets from it: the time index may differ from
        // lane to lane (solve_owned_kernel), and an index expression per value would then take an
        // address register pair per load
        // (a time step's rows are a whole number of interleaved groups: NU % Vk == 0, NX % VK == 0)
        const T *kr = kp + (size_t)t * (NU * TILE), *ur = Ut + (size_t)t * (NU * TILE);
        const T *Kr = Kp + (size_t)t * (NU * NX * TILE), *xr = Xt + (size_t)t * (NX * TILE);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            r.k[i] = kr[gain_k<M>(0, i)];
            r.up[i] = ur[i * TILE];
#pragma unroll
            for (int j = 0; j < NX; ++j) r.K[i][j] = Kr[gain_K<M>(i, j)];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) r.xp[i] = xr[i * TILE];
    };
    T x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xs[i];
    Row ring[RING];
#pragma unroll
    for (int q = 0; q < RING; ++q) load_row(t0 + q, ring[q]); // before any store: these states are the previous trajectory's
    if (store && !first) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Xt[(t0 * NX + i) * TILE] = x[i];
    }
    bool bad = false;
    for (int q0 = 0; q0 < S; q0 += RING)
#pragma unroll
    for (int qr = 0; qr < RING; ++qr) {
        const int t = t0 + q0 + qr;
        Row r = ring[qr];
        load_row(t + RING, ring[qr]); // row t+RING is read before step t+RING-1 stores to it
        T un[NU], xn[NX], sn = T(0), cn = T(0);
        control_update<FAST, T, M>(p, alpha, x, r.xp, r.up, r.K, r.k, un, bad);
        model_step<FAST, false, M>(p, t, x, un, xn, sn, cn, bad);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
        if (store && t < t1) {
            T *uw = Ut + (size_t)t * (NU * TILE);
#pragma unroll
            for (int i = 0; i < NU; ++i) uw[i * TILE] = un[i];
            if (TC) {
                T *sw = SCt + (size_t)t * (2 * TILE);
                sw[0] = sn;
                sw[TILE] = cn;
            }
            if (t + 1 < t1 || t + 1 == T1) { // the next segment writes its own first state
                T *xw = Xt + (size_t)(t + 1) * (NX * TILE);
#pragma unroll
                for (int i = 0; i < NX; ++i) xw[i * TILE] = xn[i];
            }
        }
    }
}

// (body of commit_segments_kernel: segment `seg` of the problem b0, one lane each)
// RING: rows in flight per lane (with a segment per lane instead of per wave the addresses are per lane
// too, and four rows no longer fit the register file beside them)
template <typename T, class M, bool TC, int RING = 4, class P>
__device__ __forceinline__ void commit_segments_body(const P &p, int it, int nseg, int b0, int seg) {
    constexpr int NX = M::NX, NU = M::NU;
    const int tile = b0 >> 6, ln = b0 & 63;
    const bool valid = b0 < p.B;
    const int b = valid ? b0 : p.B - 1;
    // (the wave of segment 0 may already have marked a problem without an accepted step as
    // converged while another wave reads the flag: such a problem has idx < 0 and is not rolled
    // out either way; the flag of a problem that is rolled out does not change)
    const int idx = p.alpha_idx[b];
    const bool roll = valid && idx >= 0 && p.converged[b] == 0;
    const int T1 = p.N - 1;
    const int slow_bits = p.slow_trig[b];
    const bool slow = roll && (slow_bits & 1) != 0;
    if (seg == 0) {
        // per-problem outcome, as in commit_kernel
        const bool act0 = valid && p.converged[b] == 0;
        if (act0) {
            p.slow_trig[b] = (slow_bits & 1) | (roll ? (4 | (slow ? 2 : 0)) : 0);
            p.iters[b] += 1;
            const T Jn = roll ? p.Jtry[b] : p.J[b];
            if (roll) {
                p.J[b] = Jn;
                if (p.status && !(fabs_(Jn) < inf_<T>())) p.status[b] |= ILQR_STATUS_NONFINITE_;
            } else {
                p.converged[b] = 1;
            }
            if (it < p.hist_cap) {
                if (p.hist_cost) p.hist_cost[size_t(it) * p.hist_stride + b] = Jn;
                if (p.hist_alpha) p.hist_alpha[size_t(it) * p.hist_stride + b] = idx;
            }
        }
    }
    if (__ballot(roll) == 0ull) return;
    T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    const T *kp = tile_ptr_v<GainLayout<M>::Vk>(p.k, tile, T1 * NU, ln);
    const T *Kp = tile_ptr_v<GainLayout<M>::VK>(p.K, tile, T1 * NU * NX, ln);
    T *SCt = TC ? tile_ptr(p.SC, tile, T1 * 2, ln) : nullptr;
    const T *CKp = tile_ptr(p.CK, tile, nseg * CK_CANDIDATES * NX, ln);
    const int t0 = seg * COMMIT_SEG, t1 = t0 + COMMIT_SEG < T1 ? t0 + COMMIT_SEG : T1;
    const int cand = roll ? idx : 0;
    const T alpha = pow2_neg<T>(cand);
    T xs[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
        xs[i] = seg == 0 ? Xt[i * TILE] : CKp[(size_t)((seg * CK_CANDIDATES + cand) * NX + i) * TILE];
    // per problem: the fast sweep writes nothing for a problem that needs the library sine/cosine
    // and the second sweep (rare) writes only those, as in commit_kernel
    if (__ballot(roll && !slow) != 0ull) commit_segment_sweep<T, M, true, TC, RING>(p, Xt, Ut, Kp, kp, SCt, alpha, xs, t0, t1, roll && !slow, seg == 0);
    if (__ballot(slow) != 0ull) commit_segment_sweep<T, M, false, TC, RING>(p, Xt, Ut, Kp, kp, SCt, alpha, xs, t0, t1, slow, seg == 0);
}

template <typename T, class M, bool TC>
__global__ __launch_bounds__(64) void commit_segments_kernel(Params<T, M::NX, M::NU> p, int it, int nseg) {
    commit_segments_body<T, M, TC>(p, it, nseg, (int)(blockIdx.x / nseg) * 64 + (int)threadIdx.x, (int)(blockIdx.x % nseg));
}

// ---------------------------------------------------------------- whole solve in one launch
// The loop of ilqr.py:107-114 for the 256 / L problems of a workgroup, which never have to wait
// for any other problem: backward pass (one lane per problem, first wave), line search (L lanes per
// problem, A step sizes per lane, all four waves; keeps the candidates' checkpoints) and
// time-parallel commit (one lane per problem and segment), separated by workgroup barriers instead
// of kernel boundaries.  The phases are the device functions the per-phase kernels are made of and
// exchange their data through the same arrays, so the results are the same bits.  With a launch per
// phase a small batch is bound by the serial sweep of each phase while most of the chip idles
// (the backward pass of 8192 problems occupies 128 of 1024 SIMDs); here the workgroups drift apart
// and the latency-bound sweeps of some overlap the arithmetic of the line search of others, and a
// workgroup whose problems have all converged leaves.
// `may_reroll`: from the second iteration on the previous states may be rolled out again instead of
// read (every problem still iterating has had a step committed by then).
template <typename T, class M, bool DIAG, int CONSTR, int L, int A, int PFB>
__global__ __launch_bounds__(256, 2) void solve_owned_kernel(Params<T, M::NX, M::NU> p, int max_iter, int nseg, int may_reroll) {
    constexpr int G = 256 / L;
    constexpr bool HAS_TRIG = trig_pairs<M>::value == 1;
    constexpr bool BARRIER = (CONSTR & 2) != 0;
    // the line search reads cached sine/cosine pairs only with two or more step sizes per lane (see eval_first)
    constexpr bool EVAL_TC = HAS_TRIG && A >= 2;
    __shared__ int live;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g0 = blockIdx.x * G;
    for (int it = 0; it < max_iter; ++it) {
        if (tid == 0) live = 0;
        __syncthreads();
        if (tid < G && g0 + tid < p.B && p.converged[g0 + tid] == 0) live = 1;
        __syncthreads();
        if (live == 0) break;
        const bool again = it > 0 && may_reroll != 0;
        const bool tc = HAS_TRIG && again && p.SC != nullptr;
        // (the indices every phase derives its addresses from pass through an opaque statement, so that
        // the compiler computes the addresses inside the phase instead of keeping the pointers of all
        // three phases in registers across the whole loop)
        int gb = g0 + tid, wv = blockIdx.x * 4 + w, ln = lane, tq = tid;
        asm volatile("" : "+v"(gb), "+v"(wv), "+v"(ln), "+v"(tq));
        if (tid < G) {
            if constexpr (HAS_TRIG) {
                if (tc)
                    backward_body<T, M, DIAG, CONSTR, PFB, true>(p, gb);
                else
                    backward_body<T, M, DIAG, CONSTR, PFB, false>(p, gb);
            } else {
                backward_body<T, M, DIAG, CONSTR, PFB, false>(p, gb);
            }
        }
        __syncthreads(); // gains
                if (again) {
            if constexpr (EVAL_TC) {
                if (tc)
                    evaluate_body<T, M, DIAG, BARRIER, L, A, true, true, true>(p, p.n_alphas, wv, ln);
                else
                    evaluate_body<T, M, DIAG, BARRIER, L, A, true, false, true>(p, p.n_alphas, wv, ln);
            } else {
                evaluate_body<T, M, DIAG, BARRIER, L, A, true, false, true>(p, p.n_alphas, wv, ln);
            }
        } else {
            evaluate_body<T, M, DIAG, BARRIER, L, A, false, false, true>(p, p.n_alphas, wv, ln);
        }
        __syncthreads(); // accepted step sizes, trial costs, checkpoints
        for (int q = tq; q < G * nseg; q += 256) {
            if (HAS_TRIG && p.SC != nullptr)
                commit_segments_body<T, M, HAS_TRIG, 2>(p, it, nseg, g0 + q % G, q / G);
            else
                commit_segments_body<T, M, false, 2>(p, it, nseg, g0 + q % G, q / G);
        }
        __syncthreads(); // trajectory, costs, flags
    }
}

// ---------------------------------------------------------------- active-set compaction
// Converged problems stop costing time only when a whole 64-problem tile is converged, which
// is rare because problems converge independently.  ilqr_solve therefore repacks every few
// iterations: the still-active problems of the current working set are copied, in order, into
// dense tiles of a second working set, and the problems that converged since the last repack
// are retired, i.e. their states and inputs are copied to their place in the caller's buffers.
// Three tiny kernels build the slot map; one streaming kernel moves the rows.  The gains K, k are
// dead between a commit and the next backward pass, so they are never moved: the backward pass
// writes them for the working set, and at the end of the solve the gains of the problems that
// never converged are retired while those of the converged problems are computed once more, in
// place, from their final trajectories (the trajectory has not changed since their last
// backward pass, so these are the same bits).

// one wave per tile of the current working set: number of active problems in the tile
static __global__ __launch_bounds__(64) void repack_count_kernel(const int32_t *slot2orig, const int32_t *nslots, int B, int first,
                                                          const int32_t *converged, int32_t *tile_count) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    const int nsl = nslots ? *nslots : B;
    const bool valid = s < nsl;
    const int o = valid ? (slot2orig ? slot2orig[s] : s + first) : 0;
    const bool act = valid && converged[o] == 0;
    const unsigned long long m = __ballot(act);
    if (threadIdx.x == 0) tile_count[blockIdx.x] = (int)__popcll(m);
}

// one workgroup: exclusive prefix sum of the tile counts, total -> *nslots_new
static __global__ __launch_bounds__(1024) void repack_scan_kernel(const int32_t *tile_count, int32_t *tile_offset, int ntiles,
                                                           int32_t *nslots_new) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (ntiles + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, ntiles);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += tile_count[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan of the partial sums
        int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        tile_offset[i] = run;
        run += tile_count[i];
    }
    if (tid == 1023) *nslots_new = part[1023];
}

template <typename T> struct RepackArray {
    const T *src;   // current working set, slot order
    const T *src_prev; // the working set before it (repack_retire_kernel, RepackArgs::gen_prev)
    T *dst_active;  // next working set, slot order (repack_move_kernel)
    T *dst_retire;  // caller's buffer, problem order (repack_retire_kernel)
    int rows;       // values per problem
    int chunk0;     // first chunk (REPACK_CHUNK rows each) of this array in the grid's y range
    int vshift;     // log2 of the rows interleaved per problem (GainLayout; 0: plain tiled layout)
    // repack_move_kernel, states only: > 0 = the state size; of a tile whose active problems are all marked
    // as regenerable (SLOW_TRIG_REGEN) only the rows of the time steps 0, XCK, 2 XCK .. move on (the
    // destination is then as sparse as a sparse commit leaves it)
    int sparse_nx;
};
// offset of row i of lane `lane` from the start of a tile, V = 1 << vshift rows interleaved
__device__ __forceinline__ size_t tile_elem(int i, int lane, int vshift) {
    return (((size_t)(i >> vshift) * TILE + (size_t)lane) << vshift) + (size_t)(i & ((1 << vshift) - 1));
}
constexpr int REPACK_CHUNK = 32;
// which problems of the current working set repack_retire_kernel copies back to the caller's buffers
constexpr int REPACK_CARRY = 0;       // none (the active ones are being carried over to the next working set, the converged ones stay where they are)
constexpr int REPACK_RETIRE_ALL = 1;  // every one (end of the solve)
constexpr int REPACK_RETIRE_LIVE = 2; // the ones that never converged (end of the solve: their gains)
template <typename T> struct RepackArgs {
    RepackArray<T> arr[7];
    int narr;
    const int32_t *slot2orig_old; // null: identity
    int32_t *slot2orig_new;
    const int32_t *nslots_old;    // null: B
    const int32_t *tile_offset;
    const int32_t *converged;     // per problem, global index
    int B;                        // problems of this part (slots before the first repack)
    int first;                    // global index of the part's first problem
    int mode;                     // REPACK_*
    // per problem: the generation (1, 2, ..) of the last working set it was copied into, 0 = never (its
    // trajectory and gains are in the caller's buffers); and its slot there
    int32_t *carried;
    int32_t *orig2slot;
    int gen;                      // move: generation of the working set being built; retire: of the current one
    // retire: the generation before the current one, or 0.  The problems that stayed behind in its working
    // set (they converged while it was the current one) leave together with this launch's: a working set
    // is only read out when the next but one is about to be built over it, or at the end of the solve
    int gen_prev;
    int first_repack;             // the current arrays are the caller's buffers: define `carried` for every problem
    const int32_t *slow_trig;     // per problem, global index (RepackArray::sparse_nx); may be null
};

// Source-major half of a repack: the active problems of the current working set go, in order, into
// dense tiles of the next one.  blockIdx.x: tile of the current working set; blockIdx.y: a chunk of
// REPACK_CHUNK rows of one of the arrays.  Every wave moves its chunk for the active problems of the
// tile with all its loads in flight together; a wave's stores are one contiguous run per row.
template <typename T> __global__ __launch_bounds__(64) void repack_move_kernel(RepackArgs<T> a) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int s = tile * 64 + lane;
    const int nsl = a.nslots_old ? *a.nslots_old : a.B;
    if (tile * 64 >= nsl) return;
    const bool valid = s < nsl;
    const int o = valid ? (a.slot2orig_old ? a.slot2orig_old[s] : s + a.first) : 0;
    const bool act = valid && a.converged[o] == 0;
    const unsigned long long m = __ballot(act);
    const int ns = act ? a.tile_offset[tile] + (int)__popcll(m & ((1ull << lane) - 1ull)) : 0;
    if (blockIdx.y == 0) {
        if (act) {
            a.slot2orig_new[ns] = o;
            a.orig2slot[o] = ns;
        }
        if (valid && (act || a.first_repack)) a.carried[o] = act ? a.gen : 0;
    }
    if (m == 0ull) return;
    int q = 0;
    while (q + 1 < a.narr && (int)blockIdx.y >= a.arr[q + 1].chunk0) ++q;
    const RepackArray<T> r = a.arr[q];
    const int i0 = ((int)blockIdx.y - r.chunk0) * REPACK_CHUNK;
    T *dst = act ? r.dst_active + (size_t)(ns >> 6) * r.rows * TILE : nullptr;
    const T *src = r.src + (size_t)tile * r.rows * TILE;
    // states of regenerable problems: only the stored rows (uniform over the wave)
    bool sparse = false;
    if (r.sparse_nx > 0 && a.slow_trig) sparse = __ballot(act && (a.slow_trig[o] & SLOW_TRIG_REGEN) == 0) == 0ull;
    T v[REPACK_CHUNK];
#pragma unroll
    for (int j = 0; j < REPACK_CHUNK; ++j) {
        const int i = i0 + j < r.rows ? i0 + j : r.rows - 1; // no branch around the loads but a uniform one
        if (!sparse || (i / r.sparse_nx) % XCK == 0) v[j] = src[tile_elem(i, lane, r.vshift)];
    }
    if (dst) {
#pragma unroll
        for (int j = 0; j < REPACK_CHUNK; ++j)
            if (i0 + j < r.rows && (!sparse || ((i0 + j) / r.sparse_nx) % XCK == 0)) dst[tile_elem(i0 + j, ns & 63, r.vshift)] = v[j];
    }
}

// Destination-major half: problems of the current working set go back to their place in the caller's
// buffers.  A problem's values there are 8 bytes of a line it shares with 15 neighbours that leave at
// other times (or never entered a working set), and lines written in pieces reach memory at a fraction
// of the streaming rate.  So a wave owns a tile of the CALLER's buffer x a chunk of rows: every lane
// reads its row value either from its problem's slot in the working set (problems that leave now; a
// tile's problems sit in the working set in order, within one or two of its tiles) or from the
// caller's buffer itself (everybody else), and whole lines are written back; the lines of a row in
// which nobody leaves are skipped.  blockIdx.x: tile of the part in the caller's buffers.
template <typename T> __global__ __launch_bounds__(64) void repack_retire_kernel(RepackArgs<T> a) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int ol = tile * 64 + lane;
    const bool valid = ol < a.B;
    const int o = a.first + (valid ? ol : a.B - 1);
    const bool live = a.converged[o] == 0;
    const int g = valid ? a.carried[o] : 0;
    const bool prev = a.gen_prev > 0 && g == a.gen_prev;
    const bool ret = prev || (g > 0 && g == a.gen && (a.mode == REPACK_RETIRE_ALL || (a.mode == REPACK_RETIRE_LIVE && live)));
    const unsigned long long m = __ballot(ret);
    if (m == 0ull) return;
    const bool line_on = ((m >> (lane & 48)) & 0xffffull) != 0ull; // somebody of this lane's line leaves
    const int slot = ret ? a.orig2slot[o] : 0;
    int q = 0;
    while (q + 1 < a.narr && (int)blockIdx.y >= a.arr[q + 1].chunk0) ++q;
    const RepackArray<T> r = a.arr[q];
    const int i0 = ((int)blockIdx.y - r.chunk0) * REPACK_CHUNK;
    T *dst = r.dst_retire + (size_t)((a.first >> 6) + tile) * r.rows * TILE;
    const T *src = ret ? (prev ? r.src_prev : r.src) + (size_t)(slot >> 6) * r.rows * TILE : dst;
    const int sl = ret ? (slot & 63) : lane;
    if (!line_on) return;
    T v[REPACK_CHUNK];
#pragma unroll
    for (int j = 0; j < REPACK_CHUNK; ++j) {
        const int i = i0 + j < r.rows ? i0 + j : r.rows - 1; // no branch around the loads
        v[j] = src[tile_elem(i, sl, r.vshift)];
    }
#pragma unroll
    for (int j = 0; j < REPACK_CHUNK; ++j)
        if (i0 + j < r.rows) dst[tile_elem(i0 + j, lane, r.vshift)] = v[j];
}

// ---------------------------------------------------------------- model probe
// The device dynamics evaluated at given points, one lane per point, reference layout in and
// out: x [count][NX], u [count][NU] -> xn [count][NX] (model_f, systems.py:33-40),
// fx [count][NX][NX] (compute_df_dx, systems.py:42-50), fu [count][NX][NU] (compute_df_du,
// systems.py:52-57), all at time step t.  Lets a caller check a device model against its host
// statement or against finite differences before solving with it.  Not on the solve path.
template <typename T, class M>
__global__ __launch_bounds__(64) void model_probe_kernel(Params<T, M::NX, M::NU> p, const T *x, const T *u, T *xn, T *fx, T *fu, int count,
                                                         int t) {
    constexpr int NX = M::NX, NU = M::NU;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= count) return;
    T xl[NX], ul[NU], xo[NX], fxl[NX][NX], ful[NX][NU];
    for (int i = 0; i < NX; ++i) xl[i] = x[(size_t)b * NX + i];
    for (int i = 0; i < NU; ++i) ul[i] = u[(size_t)b * NU + i];
    for (int i = 0; i < NX; ++i) {
        for (int j = 0; j < NX; ++j) fxl[i][j] = M::fx_one(i, j) ? T(1) : T(0);
        for (int j = 0; j < NU; ++j) ful[i][j] = T(0);
    }
    bool bad = false;
    M::template step<false>(p, t, xl, ul, xo, bad);
    M::template jac<false>(p, t, xl, ul, fxl, ful, bad);
    for (int i = 0; i < NX; ++i) {
        xn[(size_t)b * NX + i] = xo[i];
        // entries a model declares structurally zero are never read by the solver kernels
        for (int j = 0; j < NX; ++j) fx[((size_t)b * NX + i) * NX + j] = M::fx_nz(i, j) ? fxl[i][j] : T(0);
        for (int j = 0; j < NU; ++j) fu[((size_t)b * NX + i) * NU + j] = M::fu_nz(i, j) ? ful[i][j] : T(0);
    }
}

// ---------------------------------------------------------------- plain rollout
// sqp.py:56-61 (sim): X[t+1] = f(X[t], U[t]) from the stored X[0], one lane per problem, inputs taken
// as they are (no clamp, no feedback).  Library sine/cosine: this is a utility, not on the solve path.
template <typename T, class M> __global__ __launch_bounds__(64) void rollout_kernel(Params<T, M::NX, M::NU> p) {
    constexpr int NX = M::NX, NU = M::NU;
    const int tile = blockIdx.x, ln = threadIdx.x, T1 = p.N - 1; // buffers are padded to whole tiles
    T *Xt = tile_ptr(p.X, tile, p.N * NX, ln);
    const T *Ut = tile_ptr(p.U, tile, T1 * NU, ln);
    T x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = Xt[i * TILE];
    bool bad = false;
    for (int t = 0; t < T1; ++t) {
        T u[NU], xn[NX];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = Ut[(t * NU + i) * TILE];
        M::template step<false>(p, t, x, u, xn, bad);
#pragma unroll
        for (int i = 0; i < NX; ++i) Xt[((t + 1) * NX + i) * TILE] = x[i] = xn[i];
    }
}

// ---------------------------------------------------------------- receding-horizon warm start
// One lane per problem: the trajectory is moved `shift` steps towards the start (the last
// state/input is repeated at the end), optionally the first state is replaced by a new initial
// state given in the reference layout [B][n], and the problem is marked active again.  In place:
// a row is read before the rows behind it are written.
template <typename T>
__global__ __launch_bounds__(64) void shift_kernel(T *X, T *U, int32_t *converged, int32_t *iters, const T *x0, int B, int N, int n,
                                                   int m, int shift) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    const bool valid = b < B;
    const int tile = blockIdx.x, ln = threadIdx.x;
    T *Xt = tile_ptr(X, tile, N * n, ln);
    T *Ut = tile_ptr(U, tile, (N - 1) * m, ln);
    for (int t = 0; t < N; ++t) {
        const int ts = t + shift < N ? t + shift : N - 1;
        for (int i = 0; i < n; ++i) {
            T v = Xt[(size_t)(ts * n + i) * TILE];
            if (t == 0 && x0 && valid) v = x0[(size_t)b * n + i];
            Xt[(size_t)(t * n + i) * TILE] = v;
        }
    }
    for (int t = 0; t < N - 1; ++t) {
        const int ts = t + shift < N - 1 ? t + shift : N - 2;
        for (int i = 0; i < m; ++i) Ut[(size_t)(t * m + i) * TILE] = Ut[(size_t)(ts * m + i) * TILE];
    }
    if (valid) {
        converged[b] = 0;
        iters[b] = 0;
    }
}

// ---------------------------------------------------------------- layout conversion
// reference layout [B][rows] (one problem contiguous) <-> tiled device layout
// [ceil(B/64)][rows][64].  One workgroup moves a 64-problem x 64-row patch through LDS so
// both the global read and the global write are contiguous.
// vshift: log2 of the rows interleaved per problem on the tiled side (0: plain; see GainLayout)
template <typename E, bool TO_TILED>
__global__ __launch_bounds__(256) void relayout_kernel(const E *__restrict__ src, E *__restrict__ dst, int B, int rows, int vshift) {
    __shared__ E patch[64][65];
    const int tile = blockIdx.y, r0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t tbase = (size_t)tile * rows * TILE;
    if (TO_TILED) {
        for (int i = ty; i < 64; i += 4) { // problem i of the tile, row r0 + tx
            const int b = tile * 64 + i, r = r0 + tx;
            patch[i][tx] = (b < B && r < rows) ? src[(size_t)b * rows + r] : E(0);
        }
        __syncthreads();
        for (int i = ty; i < 64; i += 4) { // row r0 + i, lane tx
            const int r = r0 + i;
            if (r < rows) dst[tbase + tile_elem(r, tx, vshift)] = patch[tx][i];
        }
    } else {
        for (int i = ty; i < 64; i += 4) {
            const int r = r0 + i;
            if (r < rows) patch[i][tx] = src[tbase + tile_elem(r, tx, vshift)];
        }
        __syncthreads();
        for (int i = ty; i < 64; i += 4) {
            const int b = tile * 64 + i, r = r0 + tx;
            if (b < B && r < rows) dst[(size_t)b * rows + r] = patch[tx][i];
        }
    }
}

} // namespace ilqr
