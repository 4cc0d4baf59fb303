/*
 * ilqr_amd C ABI: batched iLQR / constrained iLQR on AMD MI355X (gfx950).
 *
 * The reference (zetian/ilqr) is pure Python and exposes no FFI; its boundary is
 * the object protocol of ilqr.py:17 (constructor), ilqr.py:107 (__call__) and
 * the attributes a caller assigns or reads (ilqr_demo.py:47-58).  The entry
 * points below are what a binding for that path needs: every function cites
 * the reference code it replaces.  All pointers are plain device pointers owned
 * by the caller; no torch types appear; nothing here allocates or frees device
 * memory, and nothing synchronises the stream (ilqr_active_count excepted).
 *
 * Device layout is structure-of-arrays with the batch index innermost, in tiles
 * of 64 problems (one gfx950 wavefront): an array with `rows` values per problem
 * is stored as [ceil(B/64)][rows][64]; element (b, r) lives at
 * (b/64)*rows*64 + r*64 + b%64.  Buffers are sized for ceil(B/64)*64 problems.
 *   X      rows = N*n        states            (ilqr.py:31  self.states)
 *   U      rows = (N-1)*m    inputs            (ilqr.py:33  self.inputs)
 *   Xref   rows = N*n        target_states     (ilqr.py:18)
 *   K      rows = (N-1)*m*n  feedback gains    (ilqr.py:78  self.K)
 *   k      rows = (N-1)*m    feed-forward      (ilqr.py:77  self.k)
 *          The two gain arrays and Xref interleave V consecutive rows of a problem (ilqr_gain_layout,
 *          ilqr_target_layout; V = 1, 2 or 4): [ceil(B/64)][rows/V][64][V], element (b, r) at
 *          ((b/64)*(rows/V) + r/V)*64*V + (b%64)*V + r%V.  Convert with ilqr_to_tiled_v /
 *          ilqr_from_tiled_v.  V = 1 is the plain layout above.
 *   centers rows = N*2, radius rows = N        (constraints.py:15-18, per problem)
 * Per-problem scalars are plain arrays of B elements:
 *   J      [B]            min_cost          (ilqr.py:29)
 *   converged, iters, alpha_idx, status [B] int32
 * ilqr_to_tiled / ilqr_from_tiled convert from / to the reference's layout
 * ([B][rows], one problem contiguous).
 * Element type is double (ILQR_F64) or float (ILQR_F32) as set in the descriptor.
 *
 * Return value: 0 on success, a negative ilqr_status otherwise; never throws.
 * Per-problem numerical events are reported in the device array `status`, not
 * in the return code.  A handle is not thread-safe; use one handle per stream.
 */
#ifndef ILQR_AMD_H
#define ILQR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILQR_MAX_N 12
#define ILQR_MAX_M 4

typedef enum ilqr_status {
    ILQR_OK = 0,
    ILQR_ERR_INVALID = -1,     /* bad descriptor field or NULL pointer */
    ILQR_ERR_UNSUPPORTED = -2, /* combination not built (model/dtype/shape) */
    ILQR_ERR_NOT_BOUND = -3,   /* a required buffer was not bound */
    ILQR_ERR_HIP = -4,         /* a HIP runtime call failed (see ilqr_last_hip_error) */
    ILQR_ERR_TOO_LARGE = -5,   /* ilqr_solve_static: more problems than its static scratch holds */
    ILQR_ERR_SPARSE_STATES = -6 /* recompute_states == 2: X holds only every 8th state row; call ilqr_states first */
} ilqr_status;

/* device dynamics: systems.py:27-57 / :59-90 / :92-121; LTV has no reference counterpart */
typedef enum ilqr_model {
    ILQR_MODEL_CAR = 0,
    ILQR_MODEL_CARACC = 1,
    ILQR_MODEL_DUBINS = 2,
    ILQR_MODEL_LTV = 3
} ilqr_model;
/* ids from here on are handed out by ilqr_load_model for dynamics compiled at run time */
#define ILQR_MODEL_USER_BASE 16

typedef enum ilqr_dtype { ILQR_F64 = 0, ILQR_F32 = 1 } ilqr_dtype;

/* bit flags for ilqr_desc.constraint_mode */
#define ILQR_CONSTR_NONE 0
#define ILQR_CONSTR_BOXQP 1   /* cilqr.py:112-134: per-step box QP on the control step */
#define ILQR_CONSTR_BARRIER 2 /* relaxed log barrier on constraints.py:20-26 (no reference solver uses it) */

/* per-problem bits in the device array `status` */
#define ILQR_STATUS_NONPD 1   /* a Quu was not positive definite; solved by pivoted LU like numpy.linalg.inv */
#define ILQR_STATUS_SINGULAR 2 /* a Quu was exactly singular (ilqr.py:99 would raise LinAlgError) */
#define ILQR_STATUS_NONFINITE 4 /* the accepted cost is not finite */
#define ILQR_STATUS_QP_FAILED 8 /* sequential QP: a QP subproblem had no solution (infeasible rows) or hit the iteration limit */

typedef struct ilqr_desc {
    int32_t B;               /* problems in this handle (this GPU's shard) */
    int32_t N;               /* horizon = target_states.shape[0] (ilqr.py:19) */
    int32_t n, m;            /* sys.state_size / sys.control_size (ilqr.py:23-24) */
    int32_t model;           /* ilqr_model */
    int32_t dtype;           /* ilqr_dtype */
    int32_t control_limited; /* system.control_limited: clamp inputs in the rollout (ilqr.py:58-61) */
    int32_t constraint_mode; /* ILQR_CONSTR_* flags */
    int32_t n_alphas;        /* line-search candidates 2^0..2^-(n_alphas-1); 0 = derive from alpha_floor */
    int32_t lanes_per_problem; /* forward mapping: lanes of a wave that share one problem (1,2,4,8,16); 0 = auto */
    int32_t alphas_per_lane;   /* step sizes each lane rolls out per round (1,2,4,8,16); 0 = auto */
    int32_t n_streams;         /* ilqr_solve splits the batch into this many tile ranges, each on a stream of
                                  its own, so phases of different parts overlap (1..8); 0 = auto */
    int32_t repack_every;      /* ilqr_solve compacts the still-active problems into dense tiles every this many
                                  iterations, from the second multiple on (needs the larger workspace, see
                                  ilqr_workspace_bytes); 0 = auto (8 for B >= 32768, else never); < 0 = never */
    int32_t recompute_states;  /* >= 0: once a step has been committed, the forward kernels roll the previous
                                  states out again from U instead of reading X (same bits, less traffic);
                                  < 0: always read X.  Sparse states on top of that: between a commit and the
                                  next backward pass only every 8th state row is stored and the backward pass
                                  regenerates the others segment by segment (same bits; 11 200 of the 60 600
                                  bytes the two passes move per Car problem less): 0 = inside ilqr_solve for
                                  B >= 32768 (X is complete when it returns), 1 = never, 2 = always, also
                                  between the phase calls (then read X only after ilqr_states).  Small models
                                  (n <= 8), unconstrained, without trig_cache / segment_commit / overlap_stages. */
    int32_t overlap_stages;    /* > 0: the second line-search stage also commits its problems and runs beside the
                                  commit of the others, on a stream of the handle's own (ilqr_forward_commit then
                                  launches it; ilqr_forward_evaluate_stage 1 does nothing); <= 0 (default): it
                                  runs before the commit, on the caller's stream.  Same results either way;
                                  slower on MI355X at B = 65536 (the two kernels fight over memory). */
    int32_t trig_cache;        /* > 0: the commits keep sine and cosine of the heading of every step of the committed
                                  trajectory in `work` (16 bytes per step and problem) and the backward pass and the
                                  line search read them instead of evaluating them again (same bits; a third fewer
                                  instructions per iteration, more traffic); 0 = auto (on up to 16384 problems, where
                                  the passes are bound by their instruction stream); < 0: off.  Built-in models without
                                  compaction only; ignored otherwise. */
    int32_t segment_commit;    /* > 0: the line search keeps the state of every candidate every 32 time steps in `work`
                                  and the commit rolls the accepted candidate out in all segments of 32 steps at once
                                  (same bits; the sequential commit leaves a small batch's chip idle for N-1 dependent
                                  steps); 0 = auto (on up to 12288 problems); < 0: off.  Needs a single-stage line
                                  search with 4 or more lanes per problem, n <= 8, no compaction, one stream. */
    int32_t fused_solve;       /* > 0: ilqr_solve runs all iterations in ONE launch in which a workgroup owns
                                  256 / lanes_per_problem problems from the first backward pass to the last commit
                                  (same bits as the launch-per-phase path; the latency-bound sweeps of some
                                  workgroups overlap the line searches of others, and a workgroup whose problems
                                  have all converged leaves); 0 = auto (12288 .. 16384 problems); < 0: off.  Needs what
                                  segment_commit needs, and unconstrained problems; forces segment_commit on when
                                  asked for explicitly.  The phase calls (ilqr_backward, ...) are not affected. */
    double dt;               /* system.dt (systems.py:30); the solver's own dt argument is unused (ilqr.py:20) */
    double lm_parameter;     /* ilqr.py:30, added to Vxx (ilqr.py:95) */
    double alpha_floor;      /* 1e-4 (ilqr.py:71) or alpha_terminal (cilqr.py:41) */
    double input_limit_lambda; /* cilqr.py:40 */
    double barrier_mu;       /* barrier weight */
    double barrier_delta;    /* barrier relaxation width (quadratic extension below it) */
    double vel_bounds[2];    /* constraints.py:18 */
    double Q[ILQR_MAX_N * ILQR_MAX_N];  /* row-major n x n (ilqr.py:25) */
    double R[ILQR_MAX_M * ILQR_MAX_M];  /* row-major m x m (ilqr.py:26) */
    double Qf[ILQR_MAX_N * ILQR_MAX_N]; /* row-major n x n (ilqr.py:27) */
    double lo[ILQR_MAX_M], hi[ILQR_MAX_M]; /* systems.py:21-24 */
} ilqr_desc;

/* Device buffers, caller-owned.  Optional ones may be NULL. */
typedef struct ilqr_buffers {
    void *X, *U;             /* in/out */
    const void *Xref;        /* in */
    void *K, *k;             /* out (rewritten by every backward pass, ilqr.py:77-78) */
    void *J;                 /* in/out: min_cost */
    int32_t *converged;      /* in/out: ilqr.py:21 self.converge, per problem */
    int32_t *iters;          /* in/out: backward+forward pairs executed */
    int32_t *alpha_idx;      /* out: accepted step index of the last forward pass, -1 = none */
    int32_t *status;         /* in/out: ILQR_STATUS_* bits, or NULL */
    void *work;              /* scratch of ilqr_workspace_bytes(); may be NULL if only ilqr_solve_static is used */
    const void *centers, *radius; /* barrier mode only */
    const void *ltv_A, *ltv_B;    /* ILQR_MODEL_LTV: [N-1][n][n], [N-1][n][m], shared by the batch; a model loaded at
                                     run time may read them as parameter tables of its own layout */
    void *hist_cost;         /* optional [hist_capacity][B]: min_cost after each iteration */
    int32_t *hist_alpha;     /* optional [hist_capacity][B]: accepted step index after each iteration */
    int32_t hist_capacity;
    int32_t reserved0;
} ilqr_buffers;

typedef struct ilqr_handle ilqr_handle;

/* replaces iterative_LQR.__init__ (ilqr.py:17-34, cilqr.py:25-45): captures sizes,
 * weights, limits and solver parameters.  Host-only; no device work. */
int ilqr_create(const ilqr_desc *desc, ilqr_handle **out);
void ilqr_destroy(ilqr_handle *h);

/* bytes of the `work` scratch buffer this handle needs */
size_t ilqr_workspace_bytes(const ilqr_handle *h);
/* number of line-search candidates the handle uses (15 for ilqr.py, 8 for cilqr.py defaults) */
int32_t ilqr_num_alphas(const ilqr_handle *h);
/* the forward mapping in use (the descriptor's, or the one chosen for 0 = auto): the first stage
 * of the line search rolls out lanes_per_problem * alphas_per_lane step sizes per problem */
int ilqr_forward_mapping(const ilqr_handle *h, int32_t *lanes_per_problem, int32_t *alphas_per_lane);
/* the effective sparse-states setting (the descriptor's recompute_states after the exclusions with
 * the sine/cosine cache, the time-parallel commit and the constraint modes): 0 = commits store every
 * state row, 1 = every 8th row inside ilqr_solve only, 2 = also between the phase calls (the caller
 * asks for X with ilqr_states).  No reference counterpart (ilqr.py keeps `states` whole). */
int32_t ilqr_sparse_states(const ilqr_handle *h);

/* attaches device buffers; replaces the attribute assignments of ilqr_demo.py:47-49 */
int ilqr_bind(ilqr_handle *h, const ilqr_buffers *bufs);

/* J[b] = cost of the current (X, U) for every problem: ilqr.py:108 / :36-46.
 * Call it whenever X or U were written by anything but this library: it also tells the
 * handle that X is no longer known to be the rollout of U. */
int ilqr_initial_cost(ilqr_handle *h, void *stream);
/* Brings X up to date for a handle with recompute_states == 2, whose commits between phase calls
 * store only every 8th state row (the backward pass and the line search regenerate what they need):
 * call it before reading or rewriting X, and before ilqr_initial_cost / ilqr_solve (which start from
 * a trajectory the caller may have written and return ILQR_ERR_SPARSE_STATES while X is sparse).  A no-op
 * otherwise; ilqr_solve leaves X complete, and ilqr_repack, ilqr_unpack and ilqr_shift, which move
 * rows of X inside the iteration, fill it in themselves.  No reference counterpart
 * (ilqr.py:62-63 stores every state as it goes). */
int ilqr_states(ilqr_handle *h, void *stream);
/* one backward pass for every non-converged problem: ilqr.py:76-105, cilqr.py:87-139 */
int ilqr_backward(ilqr_handle *h, void *stream);
/* one forward pass (line search) for every non-converged problem: ilqr.py:48-74.
 * `iteration` indexes the history buffers (ignored if none are bound).
 * ilqr_forward = ilqr_forward_evaluate (all step sizes rolled out in parallel,
 * winner selected) followed by ilqr_forward_commit (winner written in place). */
int ilqr_forward(ilqr_handle *h, int32_t iteration, void *stream);
int ilqr_forward_evaluate(ilqr_handle *h, void *stream);
int ilqr_forward_commit(ilqr_handle *h, int32_t iteration, void *stream);
/* ilqr_forward_evaluate in its two launches, for callers that time or schedule them apart:
 * stage 0 rolls out the first lanes_per_problem*alphas_per_lane step sizes for every problem
 * and lists the problems that found no improvement among them; stage 1 rolls out the remaining
 * step sizes for the listed problems (a no-op when stage 0 covered all of them).  Stage 1 is only
 * meaningful directly after stage 0 of the same line search (ilqr.py:53-74 walks the same step
 * sizes in one loop). */
int ilqr_forward_evaluate_stage(ilqr_handle *h, int32_t stage, void *stream);
/* Active-set compaction for callers that drive the phases themselves (ilqr_solve does this on its
 * own every repack_every iterations).  ilqr_repack copies the problems that are still iterating
 * into dense tiles of a working set inside `work` and retires the states and inputs of the others
 * to the bound buffers; the phase calls then work on that set.  ilqr_unpack brings everything
 * back (states, inputs and gains of every problem are in the bound buffers afterwards, bit for
 * bit what they would be without compaction) and must precede any access to the bound X, U, K, k
 * as well as ilqr_initial_cost, ilqr_shift and ilqr_solve.  Needs the workspace of a handle created
 * with compaction enabled (repack_every >= 0 and B >= 32768, or repack_every > 0).  No reference
 * counterpart: the reference solves one problem at a time. */
int ilqr_repack(ilqr_handle *h, void *stream);
int ilqr_unpack(ilqr_handle *h, void *stream);
/* iterative_LQR.__call__ (ilqr.py:107-114): initial cost, then max_iter x
 * (backward, forward); converged problems are skipped on the device. */
int ilqr_solve(ilqr_handle *h, int32_t max_iter, void *stream);
/* ilqr_solve for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same results
 * as ilqr_solve, bit for bit, but the kernels' scratch (what ilqr_buffers.work otherwise brings in:
 * trial costs, the list of the second line-search stage, counters) lives in an array inside the
 * library's device image, one copy per GPU, held until the process ends (1.8 MB), so the handle may
 * be bound with work = NULL.  The call allocates and frees nothing and runs no active-set
 * compaction.  Limits: at most ILQR_STATIC_MAX_PROBLEMS problems per handle (ILQR_ERR_TOO_LARGE
 * beyond that); n_streams <= 1 and overlap_stages <= 0 (ILQR_ERR_UNSUPPORTED otherwise).  The
 * first call on a GPU looks the array's address up; make it outside any graph capture.  All calls
 * on one GPU share that GPU's array: calls that overlap in time on one GPU (different streams or
 * threads) race and must be serialised by the caller.  Device = the one current at the call. */
#define ILQR_STATIC_MAX_PROBLEMS 65536
int ilqr_solve_static(ilqr_handle *h, int32_t max_iter, void *stream);
/* sqp.py:56-61 (sim): X[t+1] = f(X[t], U[t]) for t = 0..N-2 from the stored X[0], inputs taken as
 * they are (no clamp).  Afterwards X is no longer a line-search result: follow with
 * ilqr_initial_cost before any phase call. */
int ilqr_rollout(ilqr_handle *h, void *stream);
/* number of problems with converged == 0; synchronises `stream` */
int ilqr_active_count(ilqr_handle *h, void *stream, int32_t *out);

/* Receding-horizon warm start, the device-resident form of re-assigning `.inputs` (and
 * `.states[0]`) before calling the solver object again (ilqr_demo.py:47-49; ilqr.py:107-111
 * continues from whatever the attributes hold): X[t] <- X[min(t+shift, N-1)],
 * U[t] <- U[min(t+shift, N-2)] for every problem, then X[0] <- x0[b] if x0 (device pointer,
 * reference layout [B][n], same element type) is not NULL; converged and iters are reset to 0.
 * Follow it with ilqr_solve (or ilqr_initial_cost and the phase calls). */
int ilqr_shift(ilqr_handle *h, int32_t shift, const void *x0, void *stream);

/* Dynamics compiled at run time.  The reference accepts any object with the System duck type
 * (systems.py:4-24: model_f, compute_df_dx, compute_df_du) because it calls back into Python at
 * every time step (ilqr.py:62, :87-88); here the same three functions are a device-code snippet
 * compiled with hipcc against this library's kernel headers into a shared library of its own
 * (ilqr_amd/jit.py writes and builds it).  ilqr_load_model opens such a library, checks that it
 * was built from the same headers and returns the model id to put into ilqr_desc.model (the same
 * path returns the same id).  ILQR_ERR_UNSUPPORTED: built against other headers. */
int ilqr_load_model(const char *path, int32_t *model_id);
/* state and control size of a model id (built in or loaded) */
int ilqr_model_dims(int32_t model, int32_t *n, int32_t *m);
/* The device dynamics of a model at `count` points, reference layout, element type `dtype`:
 * x [count][n], u [count][m] -> x_next [count][n] (model_f, systems.py:33-40), fx [count][n][n]
 * (compute_df_dx, systems.py:42-50), fu [count][n][m] (compute_df_du, systems.py:52-57) at time
 * step t; entries a model declares structurally zero are returned as 0.  params_a / params_b are
 * the tables the solver would get as ltv_A / ltv_B (NULL if the model reads none).  All device
 * pointers.  For checking a device model against its host statement; not on the solve path. */
int ilqr_model_probe(int32_t model, int32_t dtype, double dt, int32_t t, const void *params_a, const void *params_b, const void *x,
                     const void *u, void *x_next, void *fx, void *fu, int32_t count, void *stream);

/* ---- sequential QP (sqp.py) -------------------------------------------------------------------
 * The reference's second solver: per iteration the dynamics are linearised around the current
 * trajectory, a QP over all states and inputs (sqp.py:77-134: tracking cost, linearised dynamics,
 * the linearised rows of constraints.py:32-48, input bounds) gives new inputs, and a step-halving
 * search on the simulated cost accepts a step towards them (sqp.py:136-171).  Batched here with one
 * lane per problem.  The reference hands the QP to the OSQP package; this library solves it with a
 * primal-dual interior-point iteration whose Newton step is a Riccati recursion (to qp_tol, far
 * below OSQP's default 1e-3).
 *
 * Needs a handle created for ILQR_MODEL_CAR (or for a model loaded with ilqr_load_model whose
 * library was built with the (px, py, v) state layout declared), ILQR_F64, with ILQR_CONSTR_BARRIER
 * set in constraint_mode (so that centers and radius are bound; barrier_mu / barrier_delta are not used)
 * and vel_bounds filled; control_limited selects whether the QP bounds the inputs (sqp.py:38-42).
 * Before the call: X = target_states (sqp.py:138; its first row is x0, sqp.py:37), U = initial
 * inputs.  After it: X, U, J (min_cost), converged, iters, alpha_idx, status, history as for
 * ilqr_solve.  K and k are used as scratch.  `work` is ilqr_sqp_workspace_bytes() of device memory;
 * u0 (tiled, rows = m) is the fixed first input when init_input_fixed != 0 (sqp.py:111-112), else
 * NULL; qp_iters (optional, [B]) receives the iteration count of each problem's last QP. */
typedef struct ilqr_sqp_opts {
    int32_t max_iter;          /* sqp.py:29 (20) */
    int32_t init_input_fixed;  /* sqp.py:36 */
    int32_t qp_max_iter;       /* interior-point iterations per QP; 0 = 200 */
    int32_t reserved;
    double eps;                /* sqp.py:31 (1e-2): smallest step size tried */
    double cost_eps;           /* sqp.py:32 (1e-6): relative cost change that ends the search */
    double qp_tol;             /* complementarity and row residual at which a QP counts as solved; 0 = 1e-10 */
    double qp_sigma;           /* centring parameter of the path-following iteration; 0 = 0.1 */
    const void *x0;            /* sqp.py:37, :120, :158: the initial state every QP and rollout starts from (device
                                  pointer, tiled, rows = n), or NULL = the first row of X */
} ilqr_sqp_opts;
size_t ilqr_sqp_workspace_bytes(const ilqr_handle *h);
int ilqr_sqp_solve(ilqr_handle *h, const ilqr_sqp_opts *opts, void *work, const void *u0, int32_t *qp_iters, void *stream);
/* the two phases of one iteration, for callers that time or check them apart: ilqr_sqp_begin
 * (min_cost = inf, converge = False), then per iteration ilqr_sqp_qp (the solved inputs are left in
 * `work`, see ilqr_sqp_solved_inputs) and ilqr_sqp_search */
int ilqr_sqp_begin(ilqr_handle *h, void *stream);
int ilqr_sqp_qp(ilqr_handle *h, const ilqr_sqp_opts *opts, void *work, const void *u0, int32_t *qp_iters, void *stream);
int ilqr_sqp_search(ilqr_handle *h, const ilqr_sqp_opts *opts, void *work, int32_t iteration, void *stream);
/* device pointer (inside work) of the inputs the last ilqr_sqp_qp solved for: tiled, rows = (N-1)*m */
void *ilqr_sqp_solved_inputs(const ilqr_handle *h, void *work);

/* layout helpers: reference layout [B][rows] (what numpy holds, e.g. states (B, N, n)
 * flattened) <-> tiled device layout [ceil(B/64)][rows][64]; elem_bytes 4 or 8; both
 * pointers are device pointers; padding lanes of the last tile are zero-filled */
int ilqr_to_tiled(const void *aos, void *tiled, int32_t B, int32_t rows, int32_t elem_bytes, void *stream);
int ilqr_from_tiled(const void *tiled, void *aos, int32_t B, int32_t rows, int32_t elem_bytes, void *stream);
/* the same for an array that interleaves `vec` (1, 2 or 4; rows % vec == 0) consecutive rows per problem
 * on the tiled side: the gain buffers K and k.  The reference holds the gains as plain arrays
 * (ilqr.py:77-78); these two calls and ilqr_gain_layout stand between that and the device layout. */
int ilqr_to_tiled_v(const void *aos, void *tiled, int32_t B, int32_t rows, int32_t elem_bytes, int32_t vec, void *stream);
int ilqr_from_tiled_v(const void *tiled, void *aos, int32_t B, int32_t rows, int32_t elem_bytes, int32_t vec, void *stream);
/* rows interleaved per problem in this handle's K and k buffers (a function of n and m only) */
int ilqr_gain_layout(const ilqr_handle *h, int32_t *vec_K, int32_t *vec_k);
/* ... and in its Xref buffer (target_states, ilqr.py:18): a function of n, and 1 whenever the handle
 * has the constraint rows bound (ILQR_CONSTR_BARRIER: barrier mode and the sequential-QP solver) */
int32_t ilqr_target_layout(const ilqr_handle *h);

const char *ilqr_strerror(int code);
/* hipError_t of the last failing HIP call on this thread (0 if none) */
int ilqr_last_hip_error(void);
/* ABI guards: sizeof(ilqr_desc) / sizeof(ilqr_buffers) as compiled into the library */
int32_t ilqr_desc_size(void);
int32_t ilqr_buffers_size(void);

#ifdef __cplusplus
}
#endif
#endif /* ILQR_AMD_H */
