// Host-side glue between the C ABI (api.hip) and the templated kernels:
// a type-erased launch table per (model, dtype, diagonal-cost) combination.
#pragma once
#include "../../include/ilqr_amd.h"
#include "kernels.hpp"

#include <cstdlib>
#include <limits>
#include <type_traits>

namespace ilqr {

struct HostCtx {
    ilqr_desc d;
    ilqr_buffers b;
    bool bound = false;
    bool diag = false;   // Q, R, Qf all diagonal
    int n_alphas = 0;
    int L = 16, A = 1;   // forward mapping
    // a launch may cover only the problems [first, first + count) of the bound batch
    // (first is a multiple of 64 so that the part starts on a tile boundary)
    int first = 0, count = 0;
    // the trajectory in X is the library's own rollout of U for every non-converged problem
    // (true between a commit and the next change of the buffers): the forward kernels may then
    // roll the previous states out again instead of reading them
    bool reroll = false;
    bool select_converged = false; // backward pass: recompute the gains of the converged problems
    // ... and, in the same pass, pass on the gains of the problems that never converged from the last
    // working set (tiled by slot; gather_slot[global problem index] = slot); null = off (Params::gK)
    const void *gather_K = nullptr, *gather_k = nullptr;
    const int32_t *gather_slot = nullptr;
    // the commits keep the sine/cosine pairs of the committed trajectory in `work` (Params::SC) and,
    // while `reroll` holds, the backward pass and the line search read them instead of evaluating
    bool trig = false;
    // time-parallel commit: the (single-stage) line search keeps candidate checkpoints in `work`
    // (Params::CK) and the commit rolls all segments out at once (commit_segments_kernel)
    bool seg = false;
    // sparse states (Params::lazy): commits mark the problems they roll out with the fast sine/cosine,
    // and once `reroll` holds they store only every XCK-th state row and the backward pass regenerates
    // the rest; ModelOps::states fills X in again
    bool lazy = false;
    int eval_stages = 3;           // line search: bit 0 = stage one, bit 1 = stage two (see evaluate_kernel)
    // stage two fused with the commit of its problems, beside the commit of the others: the stream
    // and events it runs on (null: stage two runs before the commit, on the caller's stream)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // working set of ilqr_solve's active-set compaction: when `slot2orig` is set, the tiled
    // arrays below replace the bound ones (they are already relative to this part and hold
    // *nslots problems in slot order) and the per-problem arrays are indexed through the map
    struct WorkSet {
        void *X = nullptr, *U = nullptr, *Xref = nullptr, *K = nullptr, *k = nullptr, *centers = nullptr, *radius = nullptr;
        const int32_t *slot2orig = nullptr;
        const int32_t *nslots = nullptr;
    } ws;
};

// Layout of the caller-provided `work` buffer (byte offsets).  The first seven regions always
// exist; the rest only with active-set compaction.
struct WorkLayout {
    size_t jtry = 0, slow = 0, counters = 0, list = 0, list_count = 0, trash = 0, redo = 0;
    size_t map[2] = {0, 0}, tile_count = 0, tile_offset = 0, nslots = 0, carried = 0, orig2slot = 0;
    size_t set[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}; // X, U, Xref, centers, radius of the two working sets
    size_t Kw = 0, kw = 0;
    size_t sc = 0; // sine/cosine cache (only with `trig`; never together with compaction)
    size_t ck = 0; // candidate checkpoints (only with `seg`; never together with compaction)
    size_t total = 0;
};

static inline int commit_segments(const ilqr_desc &d) { return (d.N - 1 + COMMIT_SEG - 1) / COMMIT_SEG; }

static inline WorkLayout work_layout(const ilqr_desc &d, bool compaction, bool trig = false, bool seg = false) {
    WorkLayout w;
    const size_t e = d.dtype == ILQR_F64 ? 8 : 4;
    const size_t tiles = ((size_t)d.B + TILE - 1) / TILE, Bp = tiles * TILE;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    w.jtry = take((size_t)d.B * e);
    w.slow = take((size_t)d.B * 4);
    w.counters = take(256);
    w.list = take(Bp * 4);          // line-search stage two: slots still without a winner
    w.list_count = take(tiles * 4); // their number, one counter per possible first tile of a part
    w.trash = take(Bp * e);         // one value per lane: sink of the stores of lanes that keep their trajectory
    w.redo = take((size_t)d.B * 4); // per problem: repeat the backward pass with the fallback solve
    if (compaction) {
        const size_t N = d.N, T1 = d.N - 1, n = d.n, m = d.m;
        const bool barrier = (d.constraint_mode & ILQR_CONSTR_BARRIER) != 0;
        w.map[0] = take(Bp * 4);
        w.map[1] = take(Bp * 4);
        w.tile_count = take(tiles * 4);
        w.tile_offset = take(tiles * 4);
        w.nslots = take(256);
        w.carried = take((size_t)d.B * 4);
        w.orig2slot = take((size_t)d.B * 4);
        for (int q = 0; q < 2; ++q) {
            w.set[q][0] = take(Bp * N * n * e);
            w.set[q][1] = take(Bp * T1 * m * e);
            w.set[q][2] = take(Bp * N * n * e);
            w.set[q][3] = take(barrier ? Bp * N * 2 * e : 0);
            w.set[q][4] = take(barrier ? Bp * N * e : 0);
        }
        w.Kw = take(Bp * T1 * m * n * e);
        w.kw = take(Bp * T1 * m * e);
    } else {
        // (the cache comes first: its offset does not depend on `seg`)
        if (trig) w.sc = take(Bp * (size_t)(d.N - 1) * 2 * e);
        if (seg) w.ck = take(Bp * (size_t)commit_segments(d) * CK_CANDIDATES * d.n * e);
    }
    w.total = off;
    return w;
}

typedef hipError_t (*launch_fn)(const HostCtx &, int it, hipStream_t);
// the dynamics at `count` given points (model_probe_kernel); element type = the table's
typedef hipError_t (*probe_fn)(double dt, const void *ltvA, const void *ltvB, const void *x, const void *u, void *xn, void *fx, void *fu,
                               int count, int t, hipStream_t);
struct ModelOps {
    launch_fn cost, backward, evaluate, commit;
    probe_fn probe;
    launch_fn rollout; // X[t+1] = f(X[t], U[t]) from X[0] (sqp.py:56-61)
    int trig_pairs;    // sine/cosine pairs the model shares between step and Jacobian (0: HostCtx::trig is ignored)
    // `it` iterations of backward pass + line search + commit in ONE launch (solve_owned_kernel), for a context
    // that runs the time-parallel commit (HostCtx::seg); hipErrorNotSupported if the combination is not built
    launch_fn solve;
    // fills in the state rows sparse commits left out (states_kernel); null if the model has no sparse variant
    launch_fn states;
    // the unconstrained backward pass honours HostCtx::gather_K (else the caller retires those gains itself)
    int gathers;
};

// What a separately compiled model library (ilqr_load_model) hands to this one.  `abi` must equal
// plugin_abi() of the loading library: both sides are compiled from the same headers, and the
// launch tables exchange HostCtx by reference.
struct SqpOps; // launch table of the sequential-QP kernels (sqp.hpp)
struct PluginInfo {
    int32_t abi;
    int32_t n, m;
    int32_t has_barrier; // the state starts with (px, py, v): the barrier terms of constraints.py:20-26 apply
    int32_t needs_tables; // the dynamics read the shared tables (ilqr_buffers.ltv_A / ltv_B): they must be bound
    const ModelOps *(*ops)(int dtype, bool diag);
    const SqpOps *(*sqp)(int dtype); // sequential-QP kernels for this model, or a function returning null
};
constexpr int32_t PLUGIN_ABI_VERSION = 8;
static inline int32_t plugin_abi() {
    return (int32_t)(PLUGIN_ABI_VERSION * 1000003u + sizeof(HostCtx) * 31u + sizeof(ilqr_desc) * 7u + sizeof(ilqr_buffers));
}

// one table per translation unit (ops_*.hip); nullptr if the combination is not built
const ModelOps *ops_car(int dtype, bool diag);
const ModelOps *ops_caracc(int dtype, bool diag);
const ModelOps *ops_dubins(int dtype, bool diag);
// diagonal cost weights only (keeps the build of the large model small); nullptr otherwise
const ModelOps *ops_ltv12x4_f64(bool diag);
const ModelOps *ops_ltv12x4_f32(bool diag);

template <typename T, int NX, int NU> static Params<T, NX, NU> make_params(const HostCtx &c) {
    Params<T, NX, NU> p;
    const ilqr_desc &d = c.d;
    const ilqr_buffers &b = c.b;
    const int N = d.N, T1 = d.N - 1;
    const bool packed = c.ws.slot2orig != nullptr;
    // with a working set the per-problem arrays are indexed by the global problem index held in
    // the slot map, so they are not offset by the part's first problem
    const size_t tile0 = (size_t)(c.first / TILE), f = packed ? 0 : (size_t)c.first;
    auto tiled = [&](const void *base, int rows) -> T * {
        return base ? (T *)base + tile0 * (size_t)rows * TILE : nullptr;
    };
    auto flat = [&](auto *base) { return base ? base + f : base; };
    const WorkLayout wl = work_layout(d, false);
    if (packed) {
        p.X = (T *)c.ws.X;
        p.U = (T *)c.ws.U;
        p.Xref = (const T *)c.ws.Xref;
        p.K = (T *)c.ws.K;
        p.k = (T *)c.ws.k;
        p.centers = (const T *)c.ws.centers;
        p.radius = (const T *)c.ws.radius;
    } else {
        p.X = tiled(b.X, N * NX);
        p.U = tiled(b.U, T1 * NU);
        p.Xref = tiled(b.Xref, N * NX);
        p.K = tiled(b.K, T1 * NU * NX);
        p.k = tiled(b.k, T1 * NU);
        p.centers = tiled(b.centers, N * 2);
        p.radius = tiled(b.radius, N);
    }
    p.slot2orig = c.ws.slot2orig;
    p.nslots = c.ws.nslots;
    p.list = (int32_t *)((char *)b.work + wl.list) + c.first;
    p.list_count = (int32_t *)((char *)b.work + wl.list_count) + tile0;
    p.trash = (T *)((char *)b.work + wl.trash) + c.first;
    p.SC = (c.trig && !packed) ? (T *)((char *)b.work + work_layout(d, false, true).sc) + tile0 * (size_t)(T1 * 2) * TILE : nullptr;
    p.CK = (c.seg && !packed) ? (T *)((char *)b.work + work_layout(d, false, c.trig, true).ck) +
                                    tile0 * (size_t)commit_segments(d) * CK_CANDIDATES * NX * TILE
                              : nullptr;
    p.redo = flat((int32_t *)((char *)b.work + wl.redo));
    p.J = flat((T *)b.J);
    p.Jtry = flat((T *)((char *)b.work + wl.jtry));
    p.slow_trig = flat((int32_t *)((char *)b.work + wl.slow));
    p.converged = flat(b.converged);
    p.iters = flat(b.iters);
    p.alpha_idx = flat(b.alpha_idx);
    p.status = flat(b.status);
    p.ltvA = (const T *)b.ltv_A;
    p.ltvB = (const T *)b.ltv_B;
    p.hist_cost = flat((T *)b.hist_cost);
    p.hist_alpha = flat(b.hist_alpha);
    p.hist_cap = (b.hist_cost || b.hist_alpha) ? b.hist_capacity : 0;
    p.hist_stride = d.B;
    p.B = c.count;
    p.N = d.N;
    p.n_alphas = c.n_alphas;
    p.select_converged = c.select_converged ? 1 : 0;
    p.lazy = c.lazy ? 1 : 0;
    // (indexed like the other per-problem arrays: by global problem index through the slot map, by
    // the index within the part otherwise)
    p.carried = c.select_converged ? flat((const int32_t *)((char *)b.work + work_layout(d, true).carried)) : nullptr;
    p.gK = (const T *)c.gather_K;
    p.gk = (const T *)c.gather_k;
    p.gslot = flat(c.gather_slot);
    p.slots_lo = -1;
    p.slots_hi = 0x7fffffff;
    p.dt = (T)d.dt;
    p.lm = (T)d.lm_parameter;
    p.lambda = (T)d.input_limit_lambda;
    p.mu = (T)d.barrier_mu;
    p.delta = (T)d.barrier_delta;
    p.vlo = (T)d.vel_bounds[0];
    p.vhi = (T)d.vel_bounds[1];
    for (int i = 0; i < NX * NX; ++i) {
        p.Q[i] = (T)d.Q[i];
        p.Qf[i] = (T)d.Qf[i];
        p.Q2[i] = (T)(2.0 * d.Q[i]);
        p.Qf2[i] = (T)(2.0 * d.Qf[i]);
    }
    for (int i = 0; i < NU * NU; ++i) {
        p.R[i] = (T)d.R[i];
        p.R2[i] = (T)(2.0 * d.R[i]);
    }
    for (int i = 0; i < NU; ++i) { // rollout clamp (ilqr.py:58-61); the identity without limits
        p.lo[i] = d.control_limited ? (T)d.lo[i] : -std::numeric_limits<T>::infinity();
        p.hi[i] = d.control_limited ? (T)d.hi[i] : std::numeric_limits<T>::infinity();
    }
    for (int i = 0; i < NU; ++i) { // box of the backward-pass QP (cilqr.py:116-117)
        p.qlo[i] = (T)d.lo[i];
        p.qhi[i] = (T)d.hi[i];
    }
    return p;
}

// prefetch depth (time steps in flight per lane) of the lane-per-problem kernels
#ifndef ILQR_PF_BACKWARD
#define ILQR_PF_BACKWARD 3
#endif
#ifndef ILQR_PF_COMMIT
#define ILQR_PF_COMMIT 2
#endif
#ifndef ILQR_PF_COMMIT_SPARSE
#define ILQR_PF_COMMIT_SPARSE 4
#endif
constexpr int PF_BACKWARD = ILQR_PF_BACKWARD;
constexpr int PF_COMMIT = ILQR_PF_COMMIT;
// a row of the 12-state model is 56 values: a second row in flight spills (1148 bytes of scratch per lane)
template <int NX> constexpr int pf_commit() { return NX > 8 ? 1 : PF_COMMIT; }

template <typename T, template <typename> class MT, bool DIAG> struct Launch {
    using M = MT<T>;
    static constexpr int NX = M::NX, NU = M::NU;

    static bool barrier_on(const HostCtx &c) { return (c.d.constraint_mode & ILQR_CONSTR_BARRIER) != 0; }

    static hipError_t cost(const HostCtx &c, int, hipStream_t s) {
        auto p = make_params<T, NX, NU>(c);
        const dim3 grid((c.count + 63) / 64), block(64);
        if constexpr (M::HAS_BARRIER) {
            if (barrier_on(c)) {
                hipLaunchKernelGGL((cost_kernel<T, M, DIAG, true>), grid, block, 0, s, p);
                return hipGetLastError();
            }
        }
        if (barrier_on(c)) return hipErrorInvalidValue;
        hipLaunchKernelGGL((cost_kernel<T, M, DIAG, false>), grid, block, 0, s, p);
        return hipGetLastError();
    }
    static constexpr bool HAS_TRIG = trig_pairs<M>::value == 1;
    // the cached pairs are valid exactly when the previous states may be rolled out again
    static bool use_trig(const HostCtx &c) { return HAS_TRIG && c.trig && c.reroll && c.ws.slot2orig == nullptr && !c.select_converged; }
    // the time-parallel commit needs every candidate in one round and four or more lanes per problem
    // (the mappings built with checkpoints), the plain buffers and the whole batch in one launch
    static bool use_seg(const HostCtx &c) {
        return M::NX <= 8 && c.seg && c.L >= 4 && c.L * c.A >= c.n_alphas && c.n_alphas <= CK_CANDIDATES && c.ws.slot2orig == nullptr &&
               c.side == nullptr;
    }
    // sparse states: the lane-per-problem kernels of the small models, unconstrained, without the
    // sine/cosine cache, the candidate checkpoints and a second stage that commits on its own
    static constexpr bool LAZY_BUILT = M::NX <= 8;
    static bool use_lazy(const HostCtx &c) {
        return LAZY_BUILT && c.lazy && (c.d.constraint_mode & 3) == 0 && !c.trig && !c.seg && c.side == nullptr && !c.select_converged &&
               c.d.recompute_states >= 0;
    }
    template <int CONSTR> static hipError_t backward_c(const HostCtx &c, hipStream_t s) {
        auto p = make_params<T, NX, NU>(c);
        if constexpr (LAZY_BUILT && CONSTR == 0) {
            if (use_lazy(c) && c.reroll) {
                hipLaunchKernelGGL((backward_gen_kernel<T, M, DIAG, 0, PF_BACKWARD>), dim3((c.count + 63) / 64), dim3(64), 0, s, p);
                return hipGetLastError();
            }
        }
        if constexpr (HAS_TRIG) {
            if (use_trig(c)) {
                hipLaunchKernelGGL((backward_kernel<T, M, DIAG, CONSTR, PF_BACKWARD, true>), dim3((c.count + 63) / 64), dim3(64), 0, s, p);
                return hipGetLastError();
            }
        }
        if constexpr (LAZY_BUILT && CONSTR == 0) {
            if (c.select_converged && c.gather_slot != nullptr) {
                hipLaunchKernelGGL((backward_kernel<T, M, DIAG, 0, PF_BACKWARD, false, true>), dim3((c.count + 63) / 64), dim3(64), 0, s, p);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((backward_kernel<T, M, DIAG, CONSTR, PF_BACKWARD>), dim3((c.count + 63) / 64), dim3(64), 0, s, p);
        return hipGetLastError();
    }
    static hipError_t backward(const HostCtx &c, int, hipStream_t s) {
        const int mode = c.d.constraint_mode & 3;
        if constexpr (std::is_same<M, LtvModelT<T, 12, 4>>::value && DIAG) {
            // one wave per problem on the matrix unit (kernels.hpp); the lane-per-problem
            // kernel remains for the barrier modes and as the cross-check (ILQR_LTV_GENERIC=1)
            static const bool generic = getenv("ILQR_LTV_GENERIC") != nullptr;
            if ((mode == 0 || mode == 1) && !generic) {
                auto p = make_params<T, NX, NU>(c);
                if (mode == 1) { // cilqr.py:112-134: the corner test of the box QP inside the same kernels
                    hipLaunchKernelGGL((ltv12x4_backward_staged_kernel<T, true>), dim3((c.count + 15) / 16), dim3(1024), 0, s, p);
                    hipLaunchKernelGGL((ltv12x4_backward_redo_kernel<T, true>), dim3(c.count), dim3(64), 0, s, p);
                } else {
                    hipLaunchKernelGGL((ltv12x4_backward_staged_kernel<T, false>), dim3((c.count + 15) / 16), dim3(1024), 0, s, p);
                    hipLaunchKernelGGL((ltv12x4_backward_redo_kernel<T, false>), dim3(c.count), dim3(64), 0, s, p);
                }
                return hipGetLastError();
            }
        }
        if constexpr (M::HAS_BARRIER) {
            if (mode == 2) return backward_c<2>(c, s);
            if (mode == 3) return backward_c<3>(c, s);
        }
        if (mode == 0) return backward_c<0>(c, s);
        if (mode == 1) return backward_c<1>(c, s);
        return hipErrorInvalidValue;
    }
    // Line search in two stages: the L*A first candidates for every active problem, then the
    // rest for the few problems that found no winner among them (kernels.hpp, evaluate_kernel)
    template <bool BARRIER, int L, int A> static hipError_t eval_la(const HostCtx &c, hipStream_t s) {
        return c.reroll ? eval_lar<BARRIER, L, A, true>(c, s) : eval_lar<BARRIER, L, A, false>(c, s);
    }
    template <bool BARRIER, int L, int A, bool REROLL>
    static hipError_t eval_first(const HostCtx &c, hipStream_t s, int first_end, int slots_lo, int slots_hi) {
        auto p = make_params<T, NX, NU>(c);
        p.slots_lo = slots_lo;
        p.slots_hi = slots_hi;
        const int ppw = 64 / L;                    // problems per wave
        const long most = slots_hi < c.count ? slots_hi : c.count;
        const long waves = (most + ppw - 1) / ppw;
        const int blocks = (int)((waves + 3) / 4); // 4 waves per workgroup
        // with cached sine/cosine pairs: built for the mappings small batches use (4 or more lanes per
        // problem).  Not with one step size per lane: there the evaluation of the pair is what covers
        // the latency of the next row's loads, and the kernel is faster with it (measured: 9.24
        // against 9.47 ms per solve of 4096 problems; with two or four step sizes per lane the cache wins)
        constexpr bool TC_BUILT = HAS_TRIG && REROLL && L >= 4 && A >= 2;
        const bool tc = TC_BUILT && use_trig(c);
        // with candidate checkpoints for the time-parallel commit: single-stage mappings of small batches
        if constexpr (M::NX <= 8 && L >= 4) {
            if (use_seg(c)) {
                if constexpr (TC_BUILT) {
                    if (tc) {
                        hipLaunchKernelGGL((evaluate_kernel<T, M, DIAG, BARRIER, L, A, REROLL, true, true>), dim3(blocks), dim3(256), 0, s, p, first_end);
                        return hipGetLastError();
                    }
                }
                hipLaunchKernelGGL((evaluate_kernel<T, M, DIAG, BARRIER, L, A, REROLL, false, true>), dim3(blocks), dim3(256), 0, s, p, first_end);
                return hipGetLastError();
            }
        }
        if constexpr (TC_BUILT) {
            if (tc) {
                hipLaunchKernelGGL((evaluate_kernel<T, M, DIAG, BARRIER, L, A, REROLL, true>), dim3(blocks), dim3(256), 0, s, p, first_end);
                return hipGetLastError();
            }
        }
        // one lane per problem: the rows travel through a ring in LDS (EvalRing), several steps ahead
        // (two lanes per problem share a SIMD with a second wave, i.e. 256 registers: enough for up to four
        // step sizes per lane)
        if constexpr ((L == 1 || (L == 2 && A <= 4)) && REROLL && !BARRIER && EvalRing<T, M, L>::OK) {
            static const bool no_ring = getenv("ILQR_NO_LDS_RING") != nullptr;
            if (!no_ring) {
                hipLaunchKernelGGL((evaluate_ring_kernel<T, M, DIAG, L, A>), dim3(blocks), dim3(256), 0, s, p, first_end);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((evaluate_kernel<T, M, DIAG, BARRIER, L, A, REROLL>), dim3(blocks), dim3(256), 0, s, p, first_end);
        return hipGetLastError();
    }
    template <bool BARRIER, int L, int A, bool REROLL> static hipError_t eval_lar(const HostCtx &c, hipStream_t s) {
        auto p = make_params<T, NX, NU>(c);
        const int first_end = c.n_alphas < L * A ? c.n_alphas : L * A;
        const bool two = first_end < c.n_alphas;
        if (c.eval_stages & 1) {
            hipError_t e = two ? hipMemsetAsync(p.list_count, 0, sizeof(int32_t), s) : hipSuccess;
            if (e == hipSuccess) e = eval_stage_one<BARRIER, L, A, REROLL>(c, s, first_end);
            if (e != hipSuccess) return e;
        }
        if (!two || !(c.eval_stages & 2) || c.side != nullptr) return hipSuccess;
        return launch_listed<BARRIER, REROLL, false>(c, s, first_end, 0);
    }
    // stage two over the listed problems; COMMIT: it also commits them (see evaluate_listed_kernel)
    template <bool BARRIER, bool REROLL, bool COMMIT>
    static hipError_t launch_listed(const HostCtx &c, hipStream_t s, int first_end, int it) {
        auto p = make_params<T, NX, NU>(c);
        // the length of the list is only known on the device: the grid covers the worst case and
        // the waves beyond the list leave at once
        // one round of stage two should cover all remaining candidates: 8 lanes per listed problem
        // if at most 8 remain, else 16
        if (M::NX > 8 || c.n_alphas - first_end > 8) {
            const int blocks = (int)(((long)c.count + 3) / 4); // one wave per workgroup, 4 problems per wave
            hipLaunchKernelGGL((evaluate_listed_kernel<T, M, DIAG, BARRIER, 16, REROLL, COMMIT>), dim3(blocks), dim3(64), 0, s, p, first_end,
                               c.n_alphas, it);
        } else if constexpr (M::NX <= 8) {
            const int blocks = (int)(((long)c.count + 7) / 8);
            hipLaunchKernelGGL((evaluate_listed_kernel<T, M, DIAG, BARRIER, 8, REROLL, COMMIT>), dim3(blocks), dim3(64), 0, s, p, first_end,
                               c.n_alphas, it);
        }
        return hipGetLastError();
    }
    template <bool BARRIER, int L, int A, bool REROLL> static hipError_t eval_stage_one(const HostCtx &c, hipStream_t s, int first_end) {
        hipError_t e;
        if constexpr (M::NX <= 8 && L == 1 && (A == 8 || A == 6)) {
            // Compacted working set: the number of slots shrinks during the solve and is only
            // known on the device.  With fewer than one tile per SIMD, one lane per problem leaves
            // SIMDs idle while each wave still walks its candidates in sequence, so mappings with
            // more lanes per problem (and the same number of candidates, which stage two relies
            // on) take over; all are launched and each looks at the count.
            if (c.ws.slot2orig != nullptr) {
                e = eval_first<BARRIER, 1, A, REROLL>(c, s, first_end, 40960, 0x7fffffff);
                if constexpr (A == 8) {
                    if (e == hipSuccess) e = eval_first<BARRIER, 2, 4, REROLL>(c, s, first_end, 20480, 40960);
                    if (e == hipSuccess) e = eval_first<BARRIER, 4, 2, REROLL>(c, s, first_end, -1, 20480);
                } else {
                    if (e == hipSuccess) e = eval_first<BARRIER, 2, 3, REROLL>(c, s, first_end, -1, 40960);
                }
            } else {
                e = eval_first<BARRIER, L, A, REROLL>(c, s, first_end, -1, 0x7fffffff);
            }
        } else {
            e = eval_first<BARRIER, L, A, REROLL>(c, s, first_end, -1, 0x7fffffff);
        }
        return e;
    }
    template <bool BARRIER> static hipError_t eval_b(const HostCtx &c, hipStream_t s) {
        const int key = c.L * 100 + c.A;
        if constexpr (M::NX > 8) {
            // large models: only the mappings with few step sizes per lane are built (more would
            // not fit the register file)
            switch (key) {
            case 1601: return eval_la<BARRIER, 16, 1>(c, s);
            case 802: return eval_la<BARRIER, 8, 2>(c, s);
            default: return hipErrorInvalidValue;
            }
        } else {
            switch (key) {
            case 1601: return eval_la<BARRIER, 16, 1>(c, s);
            case 802: return eval_la<BARRIER, 8, 2>(c, s);
            case 404: return eval_la<BARRIER, 4, 4>(c, s);
            case 208: return eval_la<BARRIER, 2, 8>(c, s);
            case 801: return eval_la<BARRIER, 8, 1>(c, s);
            case 402: return eval_la<BARRIER, 4, 2>(c, s);
            case 204: return eval_la<BARRIER, 2, 4>(c, s);
            case 108: return eval_la<BARRIER, 1, 8>(c, s);
            case 106: return eval_la<BARRIER, 1, 6>(c, s);
            case 203: return eval_la<BARRIER, 2, 3>(c, s);
            default: return hipErrorInvalidValue;
            }
        }
    }
    static hipError_t evaluate(const HostCtx &c, int, hipStream_t s) {
        if constexpr (std::is_same<M, LtvModelT<T, 12, 4>>::value && DIAG) {
            // all step sizes of a problem at once on the matrix unit (kernels.hpp); a single stage
            static const bool generic = getenv("ILQR_LTV_GENERIC") != nullptr;
            if (!barrier_on(c) && c.n_alphas <= 16 && !generic) {
                if (!(c.eval_stages & 1)) return hipSuccess;
                auto p = make_params<T, NX, NU>(c);
                hipLaunchKernelGGL((ltv12x4_evaluate_mfma_kernel<T, LTV_EVALUATE_PW>), dim3((c.count + 15) / 16), dim3(1024 / LTV_EVALUATE_PW), 0, s, p);
                return hipGetLastError();
            }
        }
        if constexpr (M::HAS_BARRIER) {
            if (barrier_on(c)) return eval_b<true>(c, s);
        }
        if (barrier_on(c)) return hipErrorInvalidValue;
        return eval_b<false>(c, s);
    }
    template <bool REROLL> static hipError_t commit_r(const HostCtx &c, int it, hipStream_t s) {
        auto p = make_params<T, NX, NU>(c);
        if (!use_lazy(c)) p.lazy = 0; // (no marks without the kernels that honour them)
        if constexpr (std::is_same<M, LtvModelT<T, 12, 4>>::value && DIAG) {
            // 16 problems per wave on the matrix unit (kernels.hpp), whenever the line search was the
            // single-stage matrix-unit kernel
            static const bool generic = getenv("ILQR_LTV_GENERIC") != nullptr;
            if (!barrier_on(c) && c.n_alphas <= 16 && !generic) {
                hipLaunchKernelGGL(ltv12x4_commit_mfma_kernel<T>, dim3((c.count + 63) / 64), dim3(256), 0, s, p, it, REROLL ? 0 : 1);
                return hipGetLastError();
            }
        }
        if constexpr (M::NX <= 8) {
            if (use_seg(c)) {
                const int nseg = commit_segments(c.d);
                const dim3 grid(((c.count + 63) / 64) * nseg);
                if (HAS_TRIG && c.trig)
                    hipLaunchKernelGGL((commit_segments_kernel<T, M, HAS_TRIG>), grid, dim3(64), 0, s, p, it, nseg);
                else
                    hipLaunchKernelGGL((commit_segments_kernel<T, M, false>), grid, dim3(64), 0, s, p, it, nseg);
                return hipGetLastError();
            }
        }
        if constexpr (LAZY_BUILT && REROLL) {
            if (use_lazy(c)) {
                // (four rows in flight: this variant reads five times what it writes, and only its own loads hide the latency)
                hipLaunchKernelGGL((commit_kernel<T, M, ILQR_PF_COMMIT_SPARSE, true, false, true>), dim3((c.count + 63) / 64), dim3(64), 0, s, p, it);
                return hipGetLastError();
            }
        }
        if constexpr (HAS_TRIG) {
            // every commit of a handle with the cache keeps it up to date, whether or not it can read it yet
            if (c.trig && c.ws.slot2orig == nullptr) {
                hipLaunchKernelGGL((commit_kernel<T, M, pf_commit<NX>(), REROLL, true>), dim3((c.count + 63) / 64), dim3(64), 0, s, p, it);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((commit_kernel<T, M, pf_commit<NX>(), REROLL>), dim3((c.count + 63) / 64), dim3(64), 0, s, p, it);
        return hipGetLastError();
    }
    // number of candidates of the first line-search stage; >= n_alphas: there is no second stage
    static int first_stage_candidates(const HostCtx &c) {
        if constexpr (std::is_same<M, LtvModelT<T, 12, 4>>::value && DIAG) {
            static const bool generic = getenv("ILQR_LTV_GENERIC") != nullptr;
            if (!barrier_on(c) && c.n_alphas <= 16 && !generic) return 16;
        }
        return c.L * c.A;
    }
    template <bool REROLL> static hipError_t commit_all(const HostCtx &c, int it, hipStream_t s) {
        const int first_end = first_stage_candidates(c);
        const bool beside = c.side != nullptr && first_end < c.n_alphas;
        hipError_t e = hipSuccess;
        if (beside) {
            // fork: the listed problems are decided and committed on the side stream while the
            // commit of all others runs here (the two never touch the same problem)
            e = hipEventRecord(c.ev_fork, s);
            if (e == hipSuccess) e = hipStreamWaitEvent(c.side, c.ev_fork, 0);
            if (e == hipSuccess) {
                if constexpr (M::HAS_BARRIER) {
                    e = barrier_on(c) ? launch_listed<true, REROLL, true>(c, c.side, first_end, it)
                                      : launch_listed<false, REROLL, true>(c, c.side, first_end, it);
                } else {
                    e = launch_listed<false, REROLL, true>(c, c.side, first_end, it);
                }
            }
            if (e == hipSuccess) e = hipEventRecord(c.ev_join, c.side);
            if (e != hipSuccess) return e;
        }
        e = commit_r<REROLL>(c, it, s);
        if (beside && e == hipSuccess) e = hipStreamWaitEvent(s, c.ev_join, 0);
        return e;
    }
    static hipError_t commit(const HostCtx &c, int it, hipStream_t s) {
        return c.reroll ? commit_all<true>(c, it, s) : commit_all<false>(c, it, s);
    }
    static hipError_t probe(double dt, const void *ltvA, const void *ltvB, const void *x, const void *u, void *xn, void *fx, void *fu,
                            int count, int t, hipStream_t s) {
        Params<T, NX, NU> p = {};
        p.dt = (T)dt;
        p.ltvA = (const T *)ltvA;
        p.ltvB = (const T *)ltvB;
        hipLaunchKernelGGL((model_probe_kernel<T, M>), dim3((count + 63) / 64), dim3(64), 0, s, p, (const T *)x, (const T *)u, (T *)xn,
                           (T *)fx, (T *)fu, count, t);
        return hipGetLastError();
    }
    static hipError_t rollout(const HostCtx &c, int, hipStream_t s) {
        auto p = make_params<T, NX, NU>(c);
        hipLaunchKernelGGL((rollout_kernel<T, M>), dim3((c.count + 63) / 64), dim3(64), 0, s, p);
        return hipGetLastError();
    }
    // The whole iteration loop in one launch (solve_owned_kernel): unconstrained problems, the
    // single-stage mappings with 16 candidates per round, time-parallel commit.
    template <int L, int A> static hipError_t solve_la(const HostCtx &c, int max_iter, hipStream_t s) {
        auto p = make_params<T, NX, NU>(c);
        constexpr int G = 256 / L;
        hipLaunchKernelGGL((solve_owned_kernel<T, M, DIAG, 0, L, A, 2>), dim3((c.count + G - 1) / G), dim3(256), 0, s, p, max_iter,
                           commit_segments(c.d), c.d.recompute_states >= 0 ? 1 : 0);
        return hipGetLastError();
    }
    static hipError_t solve(const HostCtx &c, int max_iter, hipStream_t s) {
        if constexpr (M::NX <= 8) {
            if (!use_seg(c) || (c.d.constraint_mode & 3) != 0 || c.reroll) return hipErrorNotSupported;
            switch (c.L * 100 + c.A) {
            case 1601: return solve_la<16, 1>(c, max_iter, s);
            case 802: return solve_la<8, 2>(c, max_iter, s);
            case 404: return solve_la<4, 4>(c, max_iter, s);
            default: return hipErrorNotSupported;
            }
        }
        return hipErrorNotSupported;
    }
    static hipError_t states(const HostCtx &c, int, hipStream_t s) {
        if constexpr (LAZY_BUILT) {
            auto p = make_params<T, NX, NU>(c);
            hipLaunchKernelGGL((states_kernel<T, M>), dim3((c.count + 63) / 64), dim3(64), 0, s, p);
            return hipGetLastError();
        }
        return hipErrorNotSupported;
    }
    static const ModelOps *table() {
        static const ModelOps ops = {&cost, &backward, &evaluate, &commit, &probe, &rollout, trig_pairs<M>::value, &solve,
                                     LAZY_BUILT ? &states : nullptr, LAZY_BUILT ? 1 : 0};
        return &ops;
    }
};

template <template <typename> class MT> static const ModelOps *ops_for(int dtype, bool diag) {
    if (dtype == ILQR_F64) return diag ? Launch<double, MT, true>::table() : Launch<double, MT, false>::table();
    if (dtype == ILQR_F32) return diag ? Launch<float, MT, true>::table() : Launch<float, MT, false>::table();
    return nullptr;
}
} // namespace ilqr
