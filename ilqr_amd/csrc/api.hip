// C ABI of the batched iLQR library (declared in include/ilqr_amd.h).
// Host-only logic: descriptor validation, dispatch to the kernel tables and
// the iteration driver.  Nothing here allocates device memory.
#include "host.hpp"
#include "sqp.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <string>

using namespace ilqr;

constexpr int MAX_PARTS = 8;

// State of one part of the batch (a tile range): its launch context and, once its active problems
// have been compacted, which working set it is in.
struct PartState {
    HostCtx ctx;      // launch context of this part (tile range [first, first + count))
    int index = 0;    // part number
    bool packed = false;
    bool stale = false; // a sparse commit has run since X was last filled in (HostCtx::lazy)
    int parity = 0;   // working set in use once packed
    int gen = 0;      // generation of that working set: 1 for the first repack's, 2 for the next ..
};

struct ilqr_handle {
    HostCtx ctx;
    const ModelOps *ops = nullptr;
    // ilqr_solve may split the batch into parts that run on streams of their own, so that one
    // part's arithmetic-bound line search overlaps another part's bandwidth-bound passes
    int n_parts = 1;
    int repack_every = 0;   // 0: no compaction
    bool trig = false;      // sine/cosine cache in `work` (HostCtx::trig)
    bool seg = false;       // candidate checkpoints in `work`, time-parallel commit (HostCtx::seg)
    bool fused = false;     // ilqr_solve runs its iterations in one launch (solve_owned_kernel)
    bool lazy_solve = false; // sparse states inside ilqr_solve (HostCtx::lazy)
    bool lazy_calls = false; // ... and between the phase calls: the caller asks for X with ilqr_states
    // the phase calls (ilqr_backward ...) and the single-stream ilqr_solve work on the whole batch as
    // one part whose compaction state lives here, between calls
    PartState whole;
    // stream and events of the second line-search stage when it runs beside the commit
    bool overlap = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool streams_ready = false;
    hipStream_t part_stream[MAX_PARTS] = {};
    hipEvent_t fork_event = nullptr, join_event[MAX_PARTS] = {};
};

static thread_local int g_last_hip_error = 0;

// Batch sizes for which ilqr_solve chooses the single-launch solve by itself.  Measured (Car, N = 200,
// ms per solve, launch per phase -> single launch): 4096 problems 7.35 -> 7.85, 8192: 9.81 -> 9.82,
// 16384: 14.43 -> 13.78, 32768: 21.2 -> 24.1; against the phase launches WITH the time-parallel commit,
// (4,4) mapping: 9216 problems 11.8 -> 12.4, 10240: 12.4 -> 12.6, 12288: 13.0 -> 12.9.  A workgroup's iteration is the same three serial sweeps
// either way; what the single launch saves is the gaps and tails of 90 launches, what it costs is
// shallower prefetching (its phases share one register budget).
constexpr int FUSED_AUTO_MIN = 12288, FUSED_AUTO_MAX = 16384;

// ---- scratch of ilqr_solve_static ---------------------------------------------------------------
// One array per GPU, in the library's own device image, so that the entry point never allocates.
// `scratch` is only a host-side stand-in for the symbol: the device address is looked up once per
// device (each GPU holds its own copy at its own address) and that pointer is what every launch,
// memset and copy gets.
constexpr size_t STATIC_SCRATCH_BYTES = 1843200; // work_layout() of ILQR_STATIC_MAX_PROBLEMS fp64 problems, rounded up
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void *g_scratch_ptr[MAX_DEVICES];
} // namespace

// Model libraries loaded at run time (ilqr_load_model).  Entries are never removed: handles keep
// pointers into the launch tables of the library they were created from.
namespace {
struct LoadedModel {
    std::string path;
    PluginInfo info;
};
constexpr int MAX_LOADED_MODELS = 256;
std::mutex g_models_mutex;
LoadedModel *g_models[MAX_LOADED_MODELS];
int g_n_models = 0;

const LoadedModel *loaded_model(int model_id) {
    std::lock_guard<std::mutex> lock(g_models_mutex);
    const int i = model_id - ILQR_MODEL_USER_BASE;
    return (i >= 0 && i < g_n_models) ? g_models[i] : nullptr;
}

// n, m and the launch table of a model id; false if the id is unknown
bool model_lookup(int model, int dtype, bool diag, int &n, int &m, bool &has_barrier, const ModelOps *&ops) {
    has_barrier = false;
    switch (model) {
    case ILQR_MODEL_CAR: n = 4; m = 2; has_barrier = true; ops = ops_car(dtype, diag); return true;
    case ILQR_MODEL_CARACC: n = 5; m = 2; ops = ops_caracc(dtype, diag); return true;
    case ILQR_MODEL_DUBINS: n = 3; m = 2; ops = ops_dubins(dtype, diag); return true;
    case ILQR_MODEL_LTV:
        // table-driven x+ = A_t x + B_t u; built for n = 12, m = 4 with diagonal cost weights
        n = 12; m = 4; ops = dtype == ILQR_F64 ? ops_ltv12x4_f64(diag) : ops_ltv12x4_f32(diag);
        return true;
    default: break;
    }
    const LoadedModel *lm = loaded_model(model);
    if (!lm) return false;
    n = lm->info.n;
    m = lm->info.m;
    has_barrier = lm->info.has_barrier != 0;
    ops = lm->info.ops(dtype, diag);
    return true;
}
} // namespace

static int hip_rc(hipError_t e) {
    if (e == hipSuccess) return ILQR_OK;
    g_last_hip_error = (int)e;
    return ILQR_ERR_HIP;
}

static bool is_diag(const double *M, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (i != j && M[i * n + j] != 0.0) return false;
    return true;
}

// Number of step sizes the serial search of ilqr.py:53-74 can visit: alpha_j = 2^-j is
// tried while every earlier alpha was >= the floor, i.e. 1 + #{j >= 0 : 2^-j >= floor}.
static int alphas_from_floor(double floor_) {
    if (!(floor_ > 0.0)) return -1;
    int cnt = 1;
    double a = 1.0;
    while (a >= floor_ && cnt < 64) {
        ++cnt;
        a *= 0.5;
    }
    return cnt;
}

template <bool TO_TILED> static int relayout(const void *src, void *dst, int B, int rows, int elem, int vec, void *stream) {
    if (!src || !dst || B <= 0 || rows <= 0) return ILQR_ERR_INVALID;
    if ((vec != 1 && vec != 2 && vec != 4) || rows % vec != 0) return ILQR_ERR_INVALID;
    const int vshift = vec == 4 ? 2 : vec == 2 ? 1 : 0;
    dim3 grid((rows + 63) / 64, (B + 63) / 64), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (elem == 8)
        hipLaunchKernelGGL((relayout_kernel<double, TO_TILED>), grid, block, 0, s, (const double *)src, (double *)dst, B, rows, vshift);
    else if (elem == 4)
        hipLaunchKernelGGL((relayout_kernel<float, TO_TILED>), grid, block, 0, s, (const float *)src, (float *)dst, B, rows, vshift);
    else
        return ILQR_ERR_INVALID;
    return hip_rc(hipGetLastError());
}

// ---- active-set compaction (see kernels.hpp) --------------------------------------------
// mode REPACK_CARRY: build the next working set; REPACK_RETIRE_ALL: states and inputs of every
// problem still in the working set go back to the caller's buffers; REPACK_RETIRE_LIVE: gains of
// the problems that never converged go back to the caller's buffers.
template <typename T>
static hipError_t repack_launch(ilqr_handle *h, PartState &ps, int mode, hipStream_t s) {
    const ilqr_desc &d = h->ctx.d;
    const ilqr_buffers &b = ps.ctx.b;
    const WorkLayout wl = work_layout(d, true);
    char *work = (char *)b.work;
    const int N = d.N, T1 = d.N - 1, n = d.n, m = d.m;
    const bool barrier = (d.constraint_mode & ILQR_CONSTR_BARRIER) != 0;
    const size_t tile0 = (size_t)(ps.ctx.first / TILE);
    const int ntiles = (ps.ctx.count + TILE - 1) / TILE;
    const int q = ps.packed ? 1 - ps.parity : 0;
    auto part_rel = [&](const void *base, int rows) -> T * { return base ? (T *)base + tile0 * (size_t)rows * TILE : nullptr; };
    auto set_ptr = [&](int set, int which, int rows) -> T * { return part_rel(work + wl.set[set][which], rows); };
    int32_t *map_new = (int32_t *)(work + wl.map[q]) + ps.ctx.first;
    int32_t *tile_count = (int32_t *)(work + wl.tile_count) + tile0;
    int32_t *tile_offset = (int32_t *)(work + wl.tile_offset) + tile0;
    int32_t *nslots_new = (int32_t *)(work + wl.nslots) + ps.index * 2 + q;
    const int32_t *map_old = ps.ctx.ws.slot2orig, *nslots_old = ps.ctx.ws.nslots;
    const bool carry = mode == REPACK_CARRY;

    if (carry) {
        hipLaunchKernelGGL(repack_count_kernel, dim3(ntiles), dim3(64), 0, s, map_old, nslots_old, ps.ctx.count, ps.ctx.first,
                           b.converged, tile_count);
        hipLaunchKernelGGL(repack_scan_kernel, dim3(1), dim3(1024), 0, s, (const int32_t *)tile_count, tile_offset, ntiles, nslots_new);
    }
    RepackArgs<T> a;
    a.slot2orig_old = map_old;
    a.slot2orig_new = map_new;
    a.nslots_old = nslots_old;
    a.tile_offset = tile_offset;
    a.converged = b.converged;
    a.B = ps.ctx.count;
    a.first = ps.ctx.first;
    a.mode = mode;
    a.carried = (int32_t *)(work + wl.carried);
    a.orig2slot = (int32_t *)(work + wl.orig2slot);
    const bool pk = ps.packed;
    a.first_repack = pk ? 0 : 1;
    a.slow_trig = (const int32_t *)(work + wl.slow);
    a.gen_prev = 0;
    int chunks = 0;
    auto add = [&](const void *src, T *dst_active, T *dst_retire, int rows, int vec = 1) {
        a.arr[a.narr].vshift = vec == 4 ? 2 : vec == 2 ? 1 : 0;
        a.arr[a.narr].sparse_nx = 0;
        a.arr[a.narr].src_prev = nullptr;
        a.arr[a.narr].src = (const T *)src;
        a.arr[a.narr].dst_active = dst_active;
        a.arr[a.narr].dst_retire = dst_retire;
        a.arr[a.narr].rows = rows;
        a.arr[a.narr].chunk0 = chunks;
        chunks += (rows + REPACK_CHUNK - 1) / REPACK_CHUNK;
        ++a.narr;
    };
    // current arrays: the caller's buffers until the first repack, a working set afterwards
    const void *Xc = pk ? ps.ctx.ws.X : part_rel(b.X, N * n), *Uc = pk ? ps.ctx.ws.U : part_rel(b.U, T1 * m);
    const void *Rc = pk ? ps.ctx.ws.Xref : part_rel(b.Xref, N * n);
    const void *Cc = pk ? ps.ctx.ws.centers : part_rel(b.centers, N * 2), *Dc = pk ? ps.ctx.ws.radius : part_rel(b.radius, N);
    // destination-major: problems leave a working set for their own place in the caller's buffers (before
    // the first repack they are there already).  States and inputs leave late: the problems that
    // converge while a working set is the current one stay in it, and it is read out when the next but
    // one generation is about to be built over it (a repack with ps.gen >= 2) or at the end of the
    // solve, together with the current one; the two repacks of a 30-iteration solve retire nothing.
    if (pk) {
        a.narr = 0;
        chunks = 0;
        a.gen = ps.gen;
        a.gen_prev = 0;
        const int qp = 1 - ps.parity; // the set the generation before the current one lives in
        if (mode == REPACK_RETIRE_LIVE) {
            add(ps.ctx.ws.K, nullptr, (T *)b.K, T1 * m * n, gain_vec_of(n, n));
            add(ps.ctx.ws.k, nullptr, (T *)b.k, T1 * m, gain_vec_of(m, n));
        } else {
            a.gen_prev = ps.gen >= 2 ? ps.gen - 1 : 0;
            add(Xc, nullptr, (T *)b.X, N * n);
            a.arr[a.narr - 1].src_prev = set_ptr(qp, 0, N * n);
            add(Uc, nullptr, (T *)b.U, T1 * m);
            a.arr[a.narr - 1].src_prev = set_ptr(qp, 1, T1 * m);
        }
        if (mode != REPACK_CARRY || a.gen_prev > 0)
            hipLaunchKernelGGL((repack_retire_kernel<T>), dim3(ntiles, chunks), dim3(64), 0, s, a);
    }
    if (carry) {
        // source-major: the active problems move on into the working set of the next generation
        a.narr = 0;
        chunks = 0;
        a.gen = ps.gen + 1;
        add(Xc, set_ptr(q, 0, N * n), nullptr, N * n);
        // (with sparse states the problems whose trajectory regenerates from its stored rows take only those along)
        if (ps.ctx.lazy) a.arr[a.narr - 1].sparse_nx = n;
        add(Uc, set_ptr(q, 1, T1 * m), nullptr, T1 * m);
        add(Rc, set_ptr(q, 2, N * n), nullptr, N * n, barrier ? 1 : gain_vec_of(n, n)); // (TargetLayout)
        if (barrier) {
            add(Cc, set_ptr(q, 3, N * 2), nullptr, N * 2);
            add(Dc, set_ptr(q, 4, N), nullptr, N);
        }
        hipLaunchKernelGGL((repack_move_kernel<T>), dim3(ntiles, chunks), dim3(64), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !carry) return e;
    ps.gen += 1;
    ps.ctx.ws.X = set_ptr(q, 0, N * n);
    ps.ctx.ws.U = set_ptr(q, 1, T1 * m);
    ps.ctx.ws.Xref = set_ptr(q, 2, N * n);
    ps.ctx.ws.centers = barrier ? set_ptr(q, 3, N * 2) : nullptr;
    ps.ctx.ws.radius = barrier ? set_ptr(q, 4, N) : nullptr;
    ps.ctx.ws.K = part_rel(work + wl.Kw, T1 * m * n);
    ps.ctx.ws.k = part_rel(work + wl.kw, T1 * m);
    ps.ctx.ws.slot2orig = map_new;
    ps.ctx.ws.nslots = nslots_new;
    ps.packed = true;
    ps.parity = q;
    if (ps.ctx.lazy) ps.stale = true; // (the new working set holds the stored rows only, as after a sparse commit)
    return hipSuccess;
}

// X of this part in full again, if sparse commits have run since it last was
static int fill_states(ilqr_handle *h, PartState &ps, hipStream_t s) {
    if (!ps.stale) return ILQR_OK;
    ps.stale = false;
    return h->ops->states ? hip_rc(h->ops->states(ps.ctx, 0, s)) : ILQR_ERR_UNSUPPORTED;
}

static int repack(ilqr_handle *h, PartState &ps, int mode, hipStream_t s) {
    // the rows are moved as they are: whoever retires or carries a trajectory needs all of it
    if (int rc = fill_states(h, ps, s); rc != ILQR_OK) return rc;
    return hip_rc(h->ctx.d.dtype == ILQR_F64 ? repack_launch<double>(h, ps, mode, s) : repack_launch<float>(h, ps, mode, s));
}

// End of a solve whose part was compacted: everything goes back to the caller's buffers.
static int unpack(ilqr_handle *h, PartState &ps, hipStream_t s) {
    if (int rc0 = fill_states(h, ps, s); rc0 != ILQR_OK) return rc0;
    if (!ps.packed) return ILQR_OK; // everything is in the caller's buffers already
    int rc = repack(h, ps, REPACK_RETIRE_ALL, s);
    if (rc != ILQR_OK) return rc;
    // gains of the converged problems: one backward pass over them in the caller's buffers.  Every lane
    // of a tile stores, so the pass also writes gains for their tile neighbours that never converged:
    // where the kernels can, it passes those on from the working set (HostCtx::gather_K), else a
    // retirement launch replaces them afterwards.
    HostCtx c = ps.ctx;
    const bool gather = h->ops->gathers != 0 && (h->ctx.d.constraint_mode & 3) == 0;
    if (gather) {
        c.gather_K = ps.ctx.ws.K;
        c.gather_k = ps.ctx.ws.k;
        c.gather_slot = (const int32_t *)((char *)ps.ctx.b.work + work_layout(h->ctx.d, true).orig2slot);
    }
    c.ws = HostCtx::WorkSet();
    c.select_converged = true;
    rc = hip_rc(h->ops->backward(c, 0, s));
    if (rc != ILQR_OK || gather) return rc;
    return repack(h, ps, REPACK_RETIRE_LIVE, s);
}

extern "C" {

int ilqr_create(const ilqr_desc *desc, ilqr_handle **out) {
    if (!desc || !out) return ILQR_ERR_INVALID;
    *out = nullptr;
    const ilqr_desc &d = *desc;
    if (d.B <= 0 || d.N < 2 || d.n <= 0 || d.m <= 0 || d.n > ILQR_MAX_N || d.m > ILQR_MAX_M) return ILQR_ERR_INVALID;
    if (d.dtype != ILQR_F64 && d.dtype != ILQR_F32) return ILQR_ERR_INVALID;
    if (d.constraint_mode & ~(ILQR_CONSTR_BOXQP | ILQR_CONSTR_BARRIER)) return ILQR_ERR_INVALID;
    if (!(d.dt == d.dt)) return ILQR_ERR_INVALID;
    int want_n = 0, want_m = 0;
    const ModelOps *ops = nullptr;
    bool has_barrier = false;
    bool diag = is_diag(d.Q, d.n) && is_diag(d.Qf, d.n) && is_diag(d.R, d.m);
    if (!model_lookup(d.model, d.dtype, diag, want_n, want_m, has_barrier, ops)) return ILQR_ERR_UNSUPPORTED;
    if (d.model == ILQR_MODEL_LTV && (d.n != 12 || d.m != 4)) return ILQR_ERR_UNSUPPORTED;
    if (d.n != want_n || d.m != want_m) return ILQR_ERR_INVALID;
    if (!ops) return ILQR_ERR_UNSUPPORTED;
    if ((d.constraint_mode & ILQR_CONSTR_BARRIER)) {
        // the constraint functions are written for the Car state layout (constraints.py:12)
        if (!has_barrier) return ILQR_ERR_UNSUPPORTED;
        if (!(d.barrier_mu >= 0.0) || !(d.barrier_delta > 0.0)) return ILQR_ERR_INVALID;
    }
    if ((d.constraint_mode & ILQR_CONSTR_BOXQP) || d.control_limited)
        for (int j = 0; j < d.m; ++j)
            if (!(d.lo[j] <= d.hi[j])) return ILQR_ERR_INVALID;
    int na = d.n_alphas > 0 ? d.n_alphas : alphas_from_floor(d.alpha_floor);
    if (na <= 0 || na > 64) return ILQR_ERR_INVALID;
    int L = d.lanes_per_problem, A = d.alphas_per_lane;
    if (L == 0 && A == 0) {
        // How the candidates of a round are spread over lanes is a pure speed choice.  Few lanes
        // per problem means more independent rollouts per lane (instruction-level parallelism,
        // shared row loads); many lanes per problem means more waves, which small batches need to
        // occupy the 1024 SIMDs.  Eight candidates per round settle all but a few per cent of the
        // problems; the rest finish in the second stage (evaluate_listed_kernel).  The large model
        // is only built with 16 candidates per round.
        if (d.n > 8) {
            L = 8;
            while (L < 16 && (long)d.B * L < 131072L) L *= 2;
            A = 16 / L;
        } else {
            L = 1;
            while (L < 8 && (long)d.B * L < 65536L) L *= 2;
            A = 8 / L;
            // More than eight step sizes (ilqr.py:71: 15): six candidates in the first stage are the better
            // split (the accepted step is below index 6 in about 98 % of the line searches that accept one),
            // on one lane per problem while that fills the chip, on two lanes with three each below (measured,
            // ms per solve of 24576 / 32768 / 49152 problems: (4,2) or (2,4) 21.4 / 20.9 / 27.9, (2,3) 17.9 /
            // 19.9 / 27.2, (1,6) 20.4 / 20.8 / 26.2); the same split the compacted working set uses
            if (na > 8) {
                L = d.B < 40960 ? 2 : 1;
                A = 6 / L;
            }
            // Eight step sizes or fewer (cilqr.py:41) up to 16384 problems: a lane per candidate.  These
            // solves run up to 100 iterations, most of them for a few stragglers, where only the length
            // of a lane's serial rollout counts (measured at 16384 problems with box QP and barrier:
            // 55.2 -> 49.7 ms per solve)
            if (na <= 8 && d.B <= 16384) {
                L = 8;
                A = 1;
            }
            // Up to 16384 problems every kernel is bound by the latency of one serial rollout, not by
            // throughput, and the second stage costs a whole rollout for a few problems: sixteen
            // candidates in a single stage are faster (measured at 4096 / 8192 / 16384 problems:
            // 13.4 -> 10.7, 13.6 -> 11.8, 17.3 -> 15.9 ms per solve; at 32768 two stages win)
            // (lanes per problem by batch size, measured: (8,2) / (4,4) at 8192 problems 9.8 / 11.6 ms, at 9216: 14.7 / 12.4)
            if (na > 8 && na <= 16 && d.B <= 16384) {
                L = d.B <= 4096 ? 16 : d.B <= 8192 ? 8 : 4;
                A = 16 / L;
            }
        }
    } else if (L == 0 || A == 0) {
        return ILQR_ERR_INVALID;
    }
    {
        const int key = L * 100 + A;
        const int ok[] = {1601, 802, 404, 208, 801, 402, 204, 108, 106, 203};
        bool found = false;
        for (int v : ok) found = found || (v == key);
        if (!found) return ILQR_ERR_UNSUPPORTED;
    }
    ilqr_handle *h = new (std::nothrow) ilqr_handle();
    if (!h) return ILQR_ERR_INVALID;
    h->ctx.d = d;
    std::memset(&h->ctx.b, 0, sizeof(h->ctx.b));
    h->ctx.diag = diag;
    h->ctx.n_alphas = na;
    h->ctx.L = L;
    h->ctx.A = A;
    h->ctx.first = 0;
    h->ctx.count = d.B;
    h->ops = ops;
    {
        int parts = d.n_streams;
        if (parts < 0 || parts > MAX_PARTS) {
            delete h;
            return ILQR_ERR_INVALID;
        }
        const int tiles = (d.B + TILE - 1) / TILE;
        if (parts == 0) parts = 1;
        if (parts > tiles) parts = tiles;
        h->n_parts = parts;
        // 0 = auto: large batches are bound by bandwidth, where skipping the converged problems
        // pays for the copies (38.7 -> 35.9 ms at B = 65536 Car problems); small ones are bound by
        // the latency of a time step, where it cannot
        h->repack_every = d.repack_every > 0 ? d.repack_every : (d.repack_every == 0 && d.B >= 32768 ? 8 : 0);
    }
    // Sine/cosine cache: 0 = auto.  Up to 16384 problems the passes are bound by their instruction
    // stream (one wave per SIMD or fewer), a third of which is the sine/cosine of the heading; above,
    // by bandwidth, where 16 more bytes per step and pass cost more than the instructions saved.
    // Not with compaction (the cache is not carried into the working sets) nor beside a second
    // line-search stage that commits on its own.
    // (sparse states asked for by value, recompute_states = 2, win over the automatic choice here and of
    // the time-parallel commit below: the three exclude one another)
    const bool want_sparse = d.recompute_states >= 2;
    h->trig = ops->trig_pairs == 1 && h->repack_every == 0 && d.recompute_states >= 0 && d.overlap_stages <= 0 &&
              (d.trig_cache > 0 || (d.trig_cache == 0 && d.B <= 16384 && !want_sparse));
    h->ctx.trig = h->trig;
    // Time-parallel commit: 0 = auto, for the same batch sizes (one lane per problem leaves most of the
    // chip idle during the sequential commit); needs a single-stage line search with four or more
    // lanes per problem (the launch tables check that), no compaction, one part.
    const bool seg_ok = d.n <= 8 && h->repack_every == 0 && d.overlap_stages <= 0 && h->n_parts == 1 && L >= 4 && L * A >= na && na <= 16;
    // One launch for the whole solve (solve_owned_kernel): 0 = auto.  It is built on the time-parallel
    // commit, for unconstrained problems and the mappings with 16 candidates per round.
    const bool fused_ok = seg_ok && ops->solve != nullptr && (d.constraint_mode & 3) == 0 && L * A == 16 && d.segment_commit >= 0;
    h->fused = fused_ok && (d.fused_solve > 0 || (d.fused_solve == 0 && !want_sparse && d.B >= FUSED_AUTO_MIN && d.B <= FUSED_AUTO_MAX));
    h->seg = seg_ok && (h->fused || d.segment_commit > 0 ||
                        (d.segment_commit == 0 && !want_sparse && d.B <= (d.constraint_mode != 0 ? 16384 : 12288)));
    // (unconstrained: 9216 problems 12.9 -> 11.8 ms, 12288: 13.7 -> 13.0, at 16384: 14.6 against 14.5 ms, no gain; the constrained solver's long tail of
    // iterations with few active problems gains at 16384 too: 58.0 -> 55.2 ms)
    h->ctx.seg = h->seg;
    // Sparse states (recompute_states: 0 = auto, 1 = never, 2 = everywhere).  They save traffic, which
    // is what bounds large batches; small ones are bound by the instruction stream of a sweep.
    {
        const bool ok = ops->states != nullptr && d.n <= 8 && (d.constraint_mode & 3) == 0 && !h->trig && !h->seg && d.overlap_stages <= 0;
        h->lazy_calls = ok && d.recompute_states >= 2;
        h->lazy_solve = ok && (d.recompute_states >= 2 || (d.recompute_states == 0 && d.B >= 32768));
    }
    // off unless asked for: beside the bandwidth-bound commit the latency-bound second stage runs
    // 2.5x slower and slows the commit down (measured: 34.8 -> 42.0 ms per solve at B = 65536)
    h->overlap = d.overlap_stages > 0 && h->n_parts == 1;
    *out = h;
    return ILQR_OK;
}

void ilqr_destroy(ilqr_handle *h) {
    if (!h) return;
    if (h->streams_ready) {
        for (int s = 0; s < h->n_parts; ++s) {
            if (h->part_stream[s]) (void)hipStreamDestroy(h->part_stream[s]);
            if (h->join_event[s]) (void)hipEventDestroy(h->join_event[s]);
        }
        if (h->fork_event) (void)hipEventDestroy(h->fork_event);
    }
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

// layout of `work`: see work_layout() in host.hpp
size_t ilqr_workspace_bytes(const ilqr_handle *h) { return h ? work_layout(h->ctx.d, h->repack_every > 0, h->trig, h->seg).total : 0; }

int32_t ilqr_num_alphas(const ilqr_handle *h) { return h ? h->ctx.n_alphas : 0; }

int ilqr_forward_mapping(const ilqr_handle *h, int32_t *lanes_per_problem, int32_t *alphas_per_lane) {
    if (!h || !lanes_per_problem || !alphas_per_lane) return ILQR_ERR_INVALID;
    *lanes_per_problem = h->ctx.L;
    *alphas_per_lane = h->ctx.A;
    return ILQR_OK;
}

int32_t ilqr_sparse_states(const ilqr_handle *h) { return h ? (h->lazy_calls ? 2 : h->lazy_solve ? 1 : 0) : 0; }

int ilqr_bind(ilqr_handle *h, const ilqr_buffers *b) {
    if (!h || !b) return ILQR_ERR_INVALID;
    // `work` may be NULL for a caller that only uses ilqr_solve_static; every other call checks it
    if (!b->X || !b->U || !b->Xref || !b->K || !b->k || !b->J || !b->converged || !b->iters || !b->alpha_idx)
        return ILQR_ERR_NOT_BOUND;
    if ((h->ctx.d.constraint_mode & ILQR_CONSTR_BARRIER) && (!b->centers || !b->radius)) return ILQR_ERR_NOT_BOUND;
    if (h->ctx.d.model == ILQR_MODEL_LTV && (!b->ltv_A || !b->ltv_B)) return ILQR_ERR_NOT_BOUND;
    if (const LoadedModel *lm = loaded_model(h->ctx.d.model); lm && lm->info.needs_tables && (!b->ltv_A || !b->ltv_B))
        return ILQR_ERR_NOT_BOUND; // its kernels would read through a null pointer
    if ((b->hist_cost || b->hist_alpha) && b->hist_capacity <= 0) return ILQR_ERR_INVALID;
    // X may hold only every 8th state row (sparse commits between phase calls) and the marks that say
    // how to fill it in live in `work`: binding the same X again keeps that state (with another
    // workspace the marks would be lost); binding another X starts afresh
    const bool sparse_now = h->ctx.bound && h->whole.stale && b->X == h->ctx.b.X;
    if (sparse_now && b->work != h->ctx.b.work) return ILQR_ERR_SPARSE_STATES;
    h->ctx.b = *b;
    h->ctx.bound = true;
    h->ctx.reroll = false;
    h->whole = PartState();
    h->whole.stale = sparse_now;
    return ILQR_OK;
}

#define REQUIRE_BOUND(h)                          \
    if (!(h)) return ILQR_ERR_INVALID;            \
    if (!(h)->ctx.bound) return ILQR_ERR_NOT_BOUND;
// calls whose kernels use the caller's scratch buffer
#define REQUIRE_WORK(h)                           \
    REQUIRE_BOUND(h);                             \
    if (!(h)->ctx.b.work) return ILQR_ERR_NOT_BOUND;

// the handle's own stream for the second line-search stage (created on first use, on the device
// that is current then)
static hipError_t ensure_side(ilqr_handle *h) {
    if (!h->overlap || h->side) return hipSuccess;
    hipError_t e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) {
        h->ctx.side = h->side;
        h->ctx.ev_fork = h->ev_fork;
        h->ctx.ev_join = h->ev_join;
    }
    return e;
}

// launch context of the phase calls: the handle's settings with the whole-batch compaction state
static HostCtx &phase_ctx(ilqr_handle *h) {
    h->whole.ctx.d = h->ctx.d;
    h->whole.ctx.b = h->ctx.b;
    h->whole.ctx.bound = h->ctx.bound;
    h->whole.ctx.diag = h->ctx.diag;
    h->whole.ctx.n_alphas = h->ctx.n_alphas;
    h->whole.ctx.L = h->ctx.L;
    h->whole.ctx.A = h->ctx.A;
    h->whole.ctx.first = 0;
    h->whole.ctx.count = h->ctx.d.B;
    h->whole.ctx.reroll = h->ctx.reroll;
    h->whole.ctx.select_converged = false;
    h->whole.ctx.trig = h->trig;
    h->whole.ctx.seg = h->seg;
    h->whole.ctx.lazy = h->lazy_calls;
    h->whole.ctx.eval_stages = 3;
    h->whole.ctx.side = h->ctx.side;
    h->whole.ctx.ev_fork = h->ctx.ev_fork;
    h->whole.ctx.ev_join = h->ctx.ev_join;
    return h->whole.ctx;
}

#define REQUIRE_UNPACKED(h) \
    if ((h)->whole.packed) return ILQR_ERR_INVALID; /* ilqr_unpack first: the caller's buffers are stale */

// every call that reads or moves X outside the iteration first has it filled in (sparse states)
#define FILL_STATES(h, stream)                                                       \
    if ((h)->whole.stale) {                                                          \
        phase_ctx(h);                                                                \
        if (int rcf = fill_states((h), (h)->whole, (hipStream_t)(stream)); rcf != ILQR_OK) return rcf; \
    }

// Calls that start from a trajectory the caller may have written: X filled in by the library now would
// overwrite it, X left sparse would be read as it is; the caller says which by calling ilqr_states first.
#define REQUIRE_STATES(h) \
    if ((h)->whole.stale) return ILQR_ERR_SPARSE_STATES;

int ilqr_states(ilqr_handle *h, void *stream) {
    REQUIRE_BOUND(h); // (sparse states between phase calls imply a bound workspace: those calls need one)
    FILL_STATES(h, stream);
    return ILQR_OK;
}

int ilqr_initial_cost(ilqr_handle *h, void *stream) {
    REQUIRE_WORK(h);
    REQUIRE_UNPACKED(h);
    REQUIRE_STATES(h);
    h->ctx.reroll = false; // the caller (re)wrote X and U: X need not be the rollout of U
    return hip_rc(h->ops->cost(h->ctx, 0, (hipStream_t)stream));
}

int ilqr_backward(ilqr_handle *h, void *stream) {
    REQUIRE_WORK(h);
    return hip_rc(h->ops->backward(phase_ctx(h), 0, (hipStream_t)stream));
}

int ilqr_forward_evaluate(ilqr_handle *h, void *stream) {
    REQUIRE_WORK(h);
    if (hipError_t e = ensure_side(h); e != hipSuccess) return hip_rc(e);
    return hip_rc(h->ops->evaluate(phase_ctx(h), 0, (hipStream_t)stream));
}

int ilqr_forward_evaluate_stage(ilqr_handle *h, int32_t stage, void *stream) {
    REQUIRE_WORK(h);
    if (stage != 0 && stage != 1) return ILQR_ERR_INVALID;
    if (hipError_t e = ensure_side(h); e != hipSuccess) return hip_rc(e);
    HostCtx c = phase_ctx(h);
    c.eval_stages = stage == 0 ? 1 : 2;
    return hip_rc(h->ops->evaluate(c, 0, (hipStream_t)stream));
}

int ilqr_forward_commit(ilqr_handle *h, int32_t iteration, void *stream) {
    REQUIRE_WORK(h);
    if (iteration < 0) return ILQR_ERR_INVALID;
    if (hipError_t e = ensure_side(h); e != hipSuccess) return hip_rc(e);
    const int rc = hip_rc(h->ops->commit(phase_ctx(h), iteration, (hipStream_t)stream));
    if (rc == ILQR_OK && h->lazy_calls && h->ctx.reroll) h->whole.stale = true; // (a sparse commit, if the launch table has one for this handle)
    // every problem that is still active afterwards has had a step committed by this library
    if (rc == ILQR_OK && h->ctx.d.recompute_states >= 0) h->ctx.reroll = true;
    return rc;
}

int ilqr_repack(ilqr_handle *h, void *stream) {
    REQUIRE_WORK(h);
    if (h->repack_every <= 0) return ILQR_ERR_INVALID; // the handle's workspace holds no working sets
    phase_ctx(h);
    return repack(h, h->whole, REPACK_CARRY, (hipStream_t)stream);
}

int ilqr_unpack(ilqr_handle *h, void *stream) {
    REQUIRE_WORK(h);
    phase_ctx(h);
    const int rc = unpack(h, h->whole, (hipStream_t)stream);
    if (rc == ILQR_OK) {
        h->whole.packed = false;
        h->whole.gen = 0;
        h->whole.ctx.ws = HostCtx::WorkSet();
    }
    return rc;
}

int ilqr_forward(ilqr_handle *h, int32_t iteration, void *stream) {
    int rc = ilqr_forward_evaluate(h, stream);
    if (rc != ILQR_OK) return rc;
    return ilqr_forward_commit(h, iteration, stream);
}

// The iteration driver behind ilqr_solve and ilqr_solve_static.  `work`: the scratch buffer the
// kernels use for this call (the bound one, or the library's own static array).
// `extras`: `work` has room for what the handle's workspace holds beyond the scratch proper
// (sine/cosine cache, candidate checkpoints).
static int solve_impl(ilqr_handle *h, int32_t max_iter, void *work, bool compaction, bool extras, void *stream) {
    if (hipError_t e0 = ensure_side(h); e0 != hipSuccess) return hip_rc(e0);
    hipStream_t user = (hipStream_t)stream;
    const int np = h->n_parts;
    hipError_t e = hipSuccess;
    if (np > 1) {
        // fork: every part waits for the caller's stream, runs its whole solve on a stream of its
        // own (parts are disjoint tile ranges, so they never touch the same memory), and the
        // caller's stream waits for all parts.  No host synchronisation.
        if (!h->streams_ready) {
            e = hipEventCreateWithFlags(&h->fork_event, hipEventDisableTiming);
            for (int s = 0; s < np && e == hipSuccess; ++s) {
                e = hipStreamCreateWithFlags(&h->part_stream[s], hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&h->join_event[s], hipEventDisableTiming);
            }
            if (e != hipSuccess) return hip_rc(e);
            h->streams_ready = true;
        }
        e = hipEventRecord(h->fork_event, user);
        if (e != hipSuccess) return hip_rc(e);
    }
    const int tiles = (h->ctx.d.B + TILE - 1) / TILE;
    int rc = ILQR_OK;
    PartState part[MAX_PARTS];
    hipStream_t st[MAX_PARTS];
    h->ctx.reroll = false; // ilqr_solve starts from the caller's trajectory
    int forked = 0; // part streams that wait on the fork event: each must be joined again, also after an error
    for (int s = 0; s < np && rc == ILQR_OK; ++s) {
        part[s].ctx = h->ctx;
        part[s].ctx.b.work = work;
        part[s].ctx.trig = extras && h->trig;
        part[s].ctx.seg = extras && h->seg;
        part[s].ctx.lazy = h->lazy_solve;
        part[s].index = s;
        const int t0 = (int)((long)tiles * s / np), t1 = (int)((long)tiles * (s + 1) / np);
        part[s].ctx.first = t0 * TILE;
        const int end = t1 * TILE < h->ctx.d.B ? t1 * TILE : h->ctx.d.B;
        part[s].ctx.count = end - part[s].ctx.first;
        st[s] = np > 1 ? h->part_stream[s] : user;
        if (np > 1) {
            rc = hip_rc(hipStreamWaitEvent(st[s], h->fork_event, 0));
            if (rc != ILQR_OK) break;
            forked = s + 1;
        }
        rc = hip_rc(h->ops->cost(part[s].ctx, 0, st[s]));
    }
    if (rc == ILQR_OK && np == 1 && h->fused && extras && part[0].ctx.seg) {
        // all iterations in one launch; nothing to unpack (no compaction with this path)
        return hip_rc(h->ops->solve(part[0].ctx, max_iter, st[0]));
    }
    // Launch order is iteration-major across parts.  The line search is arithmetic-bound, the
    // backward pass and the commit are bandwidth-bound; a token (an event chain) lets only one
    // part run its line search at a time, which pushes the parts into opposite phases.
    int prev = -1;
    const bool use_token = getenv("ILQR_NO_TOKEN") == nullptr;
    for (int it = 0; it < max_iter && rc == ILQR_OK; ++it)
        for (int s = 0; s < np && rc == ILQR_OK; ++s) {
            // from the second iteration on, every active problem's X is this library's rollout of U
            part[s].ctx.reroll = it > 0 && h->ctx.d.recompute_states >= 0;
            if (compaction && h->repack_every > 0 && it >= 2 * h->repack_every && it % h->repack_every == 0)
                rc = repack(h, part[s], REPACK_CARRY, st[s]);
            if (rc == ILQR_OK) rc = hip_rc(h->ops->backward(part[s].ctx, 0, st[s]));
            if (np > 1 && use_token) {
                if (rc == ILQR_OK && prev >= 0) rc = hip_rc(hipStreamWaitEvent(st[s], h->join_event[prev], 0));
                if (rc == ILQR_OK) rc = hip_rc(h->ops->evaluate(part[s].ctx, 0, st[s]));
                if (rc == ILQR_OK) rc = hip_rc(hipEventRecord(h->join_event[s], st[s]));
                prev = s;
            } else if (rc == ILQR_OK) {
                rc = hip_rc(h->ops->evaluate(part[s].ctx, 0, st[s]));
            }
            if (rc == ILQR_OK) rc = hip_rc(h->ops->commit(part[s].ctx, it, st[s]));
            if (part[s].ctx.lazy && part[s].ctx.reroll) part[s].stale = true;
        }
    for (int s = 0; s < np; ++s) {
        if (rc == ILQR_OK) rc = unpack(h, part[s], st[s]); // retire what is still in a working set
        else if (!part[s].packed) (void)fill_states(h, part[s], st[s]); // after an error: at least no sparse rows left behind
        if (np > 1 && s < forked) {
            e = hipEventRecord(h->join_event[s], st[s]);
            if (e == hipSuccess) e = hipStreamWaitEvent(user, h->join_event[s], 0);
            if (e != hipSuccess && rc == ILQR_OK) rc = hip_rc(e);
        }
    }
    return rc;
}

int ilqr_solve(ilqr_handle *h, int32_t max_iter, void *stream) {
    REQUIRE_BOUND(h);
    if (max_iter < 0) return ILQR_ERR_INVALID;
    REQUIRE_UNPACKED(h);
    if (!h->ctx.b.work) return ILQR_ERR_NOT_BOUND;
    REQUIRE_STATES(h);
    return solve_impl(h, max_iter, h->ctx.b.work, true, true, stream);
}

int ilqr_solve_static(ilqr_handle *h, int32_t max_iter, void *stream) {
    REQUIRE_BOUND(h);
    if (max_iter < 0) return ILQR_ERR_INVALID;
    REQUIRE_UNPACKED(h);
    REQUIRE_STATES(h);
    // streams and events of its own would have to be created: not in a call that must not allocate
    if (h->n_parts > 1 || h->overlap) return ILQR_ERR_UNSUPPORTED;
    const size_t need = work_layout(h->ctx.d, false).total;
    if (h->ctx.d.B > ILQR_STATIC_MAX_PROBLEMS || need > STATIC_SCRATCH_BYTES) return ILQR_ERR_TOO_LARGE;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_rc(e);
    if (dev < 0 || dev >= MAX_DEVICES) return ILQR_ERR_UNSUPPORTED;
    void *ptr = nullptr;
    {
        // set-up, once per device, with that device current: the address of this GPU's copy of the array
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            void *sym = nullptr;
            e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !sym) return hip_rc(e != hipSuccess ? e : hipErrorInvalidSymbol);
            g_scratch_ptr[dev] = sym;
        }
        ptr = g_scratch_ptr[dev];
    }
    // the array keeps what the last call left in it: start from zeros
    e = hipMemsetAsync(ptr, 0, need, (hipStream_t)stream);
    if (e != hipSuccess) return hip_rc(e);
    return solve_impl(h, max_iter, ptr, false, false, stream); // the static array holds neither cache nor checkpoints
}

int ilqr_rollout(ilqr_handle *h, void *stream) {
    REQUIRE_BOUND(h);
    REQUIRE_UNPACKED(h);
    h->ctx.reroll = false; // X was rewritten by something other than a commit
    h->whole.stale = false; // (every row of it)
    return hip_rc(h->ops->rollout(h->ctx, 0, (hipStream_t)stream));
}

namespace {
__global__ void count_active_kernel(const int32_t *conv, int B, int32_t *out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    bool a = b < B && conv[b] == 0;
    unsigned long long m = __ballot(a);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (int)__popcll(m));
}
} // namespace

int ilqr_active_count(ilqr_handle *h, void *stream, int32_t *out) {
    REQUIRE_WORK(h);
    if (!out) return ILQR_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int32_t *counter = (int32_t *)((char *)h->ctx.b.work + work_layout(h->ctx.d, false).counters);
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(int32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(count_active_kernel, dim3((h->ctx.d.B + 255) / 256), dim3(256), 0, s, h->ctx.b.converged, h->ctx.d.B, counter);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, counter, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return hip_rc(e);
}

int ilqr_shift(ilqr_handle *h, int32_t shift, const void *x0, void *stream) {
    REQUIRE_BOUND(h);
    const ilqr_desc &d = h->ctx.d;
    if (shift < 0 || shift >= d.N) return ILQR_ERR_INVALID;
    REQUIRE_UNPACKED(h);
    FILL_STATES(h, stream);
    const ilqr_buffers &b = h->ctx.b;
    const dim3 grid((d.B + TILE - 1) / TILE), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (d.dtype == ILQR_F64)
        hipLaunchKernelGGL(shift_kernel<double>, grid, block, 0, s, (double *)b.X, (double *)b.U, b.converged, b.iters,
                           (const double *)x0, d.B, d.N, d.n, d.m, shift);
    else
        hipLaunchKernelGGL(shift_kernel<float>, grid, block, 0, s, (float *)b.X, (float *)b.U, b.converged, b.iters,
                           (const float *)x0, d.B, d.N, d.n, d.m, shift);
    h->ctx.reroll = false; // the shifted trajectory is a new initial guess
    return hip_rc(hipGetLastError());
}

int ilqr_to_tiled(const void *aos, void *tiled, int32_t B, int32_t rows, int32_t elem_bytes, void *stream) {
    return relayout<true>(aos, tiled, B, rows, elem_bytes, 1, stream);
}

int ilqr_from_tiled(const void *tiled, void *aos, int32_t B, int32_t rows, int32_t elem_bytes, void *stream) {
    return relayout<false>(tiled, aos, B, rows, elem_bytes, 1, stream);
}

int ilqr_to_tiled_v(const void *aos, void *tiled, int32_t B, int32_t rows, int32_t elem_bytes, int32_t vec, void *stream) {
    return relayout<true>(aos, tiled, B, rows, elem_bytes, vec, stream);
}

int ilqr_from_tiled_v(const void *tiled, void *aos, int32_t B, int32_t rows, int32_t elem_bytes, int32_t vec, void *stream) {
    return relayout<false>(tiled, aos, B, rows, elem_bytes, vec, stream);
}

int32_t ilqr_target_layout(const ilqr_handle *h) {
    if (!h) return 0;
    return (h->ctx.d.constraint_mode & ILQR_CONSTR_BARRIER) ? 1 : gain_vec_of(h->ctx.d.n, h->ctx.d.n);
}

int ilqr_gain_layout(const ilqr_handle *h, int32_t *vec_K, int32_t *vec_k) {
    if (!h || !vec_K || !vec_k) return ILQR_ERR_INVALID;
    *vec_K = gain_vec_of(h->ctx.d.n, h->ctx.d.n);
    *vec_k = gain_vec_of(h->ctx.d.m, h->ctx.d.n);
    return ILQR_OK;
}

int ilqr_load_model(const char *path, int32_t *model_id) {
    if (!path || !model_id) return ILQR_ERR_INVALID;
    *model_id = -1;
    std::lock_guard<std::mutex> lock(g_models_mutex);
    for (int i = 0; i < g_n_models; ++i)
        if (g_models[i]->path == path) {
            *model_id = ILQR_MODEL_USER_BASE + i;
            return ILQR_OK;
        }
    if (g_n_models >= MAX_LOADED_MODELS) return ILQR_ERR_UNSUPPORTED;
    void *dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) return ILQR_ERR_INVALID;
    typedef int (*query_fn)(PluginInfo *);
    query_fn query = (query_fn)dlsym(dl, "ilqr_model_plugin");
    PluginInfo info;
    std::memset(&info, 0, sizeof(info));
    if (!query || query(&info) != 0 || info.abi != plugin_abi() || !info.ops || info.n <= 0 || info.n > ILQR_MAX_N || info.m <= 0 ||
        info.m > ILQR_MAX_M) {
        dlclose(dl);
        return query && info.abi != plugin_abi() ? ILQR_ERR_UNSUPPORTED : ILQR_ERR_INVALID;
    }
    LoadedModel *lm = new (std::nothrow) LoadedModel();
    if (!lm) {
        dlclose(dl);
        return ILQR_ERR_INVALID;
    }
    lm->path = path;
    lm->info = info;
    g_models[g_n_models] = lm;
    *model_id = ILQR_MODEL_USER_BASE + g_n_models;
    ++g_n_models;
    return ILQR_OK;
}

int ilqr_model_dims(int32_t model, int32_t *n, int32_t *m) {
    if (!n || !m) return ILQR_ERR_INVALID;
    int nn = 0, mm = 0;
    bool hb = false;
    const ModelOps *ops = nullptr;
    if (!model_lookup(model, ILQR_F64, true, nn, mm, hb, ops)) return ILQR_ERR_UNSUPPORTED;
    *n = nn;
    *m = mm;
    return ILQR_OK;
}

int ilqr_model_probe(int32_t model, int32_t dtype, double dt, int32_t t, const void *params_a, const void *params_b, const void *x,
                     const void *u, void *x_next, void *fx, void *fu, int32_t count, void *stream) {
    if (!x || !u || !x_next || !fx || !fu || count <= 0 || t < 0) return ILQR_ERR_INVALID;
    if (dtype != ILQR_F64 && dtype != ILQR_F32) return ILQR_ERR_INVALID;
    int n = 0, m = 0;
    bool hb = false;
    const ModelOps *ops = nullptr;
    if (!model_lookup(model, dtype, true, n, m, hb, ops)) return ILQR_ERR_UNSUPPORTED;
    if (!ops && !model_lookup(model, dtype, false, n, m, hb, ops)) return ILQR_ERR_UNSUPPORTED;
    if (!ops) return ILQR_ERR_UNSUPPORTED;
    if (model == ILQR_MODEL_LTV && (!params_a || !params_b)) return ILQR_ERR_NOT_BOUND;
    if (const LoadedModel *lm = loaded_model(model); lm && lm->info.needs_tables && (!params_a || !params_b)) return ILQR_ERR_NOT_BOUND;
    return hip_rc(ops->probe(dt, params_a, params_b, x, u, x_next, fx, fu, count, t, (hipStream_t)stream));
}

// ---- sequential QP (sqp.hpp) -----------------------------------------------------------------
namespace {
struct SqpLayout {
    size_t S, LAM, DU, DS, DL, DX, UQ, XQ, total;
};
SqpLayout sqp_layout(const ilqr_desc &d) {
    const size_t e = 8, Bp = ((size_t)d.B + TILE - 1) / TILE * TILE;
    const size_t N = d.N, T1 = d.N - 1, n = d.n, m = d.m, NR = 3 + 2 * m;
    SqpLayout l;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    l.S = take(Bp * N * NR * e);
    l.LAM = take(Bp * N * NR * e);
    l.DU = take(Bp * T1 * m * e);
    l.DS = take(Bp * N * NR * e);
    l.DL = take(Bp * N * NR * e);
    l.DX = take(Bp * N * n * e);
    l.UQ = take(Bp * T1 * m * e);
    l.XQ = take(Bp * N * n * e);
    l.total = off;
    return l;
}
const SqpOps *sqp_check(const ilqr_handle *h, int &rc) {
    rc = ILQR_OK;
    if (!h) {
        rc = ILQR_ERR_INVALID;
        return nullptr;
    }
    if (!h->ctx.bound || !h->ctx.b.work) {
        rc = ILQR_ERR_NOT_BOUND;
        return nullptr;
    }
    const SqpOps *ops = sqp_ops(h->ctx.d.model, h->ctx.d.dtype);
    if (!ops)
        if (const LoadedModel *lm = loaded_model(h->ctx.d.model); lm && lm->info.sqp) ops = lm->info.sqp(h->ctx.d.dtype);
    // the constraint rows need centers and radius, which a handle only demands in barrier mode
    if (!ops || !(h->ctx.d.constraint_mode & ILQR_CONSTR_BARRIER)) rc = ILQR_ERR_UNSUPPORTED;
    if (h->whole.packed) rc = ILQR_ERR_INVALID;
    return rc == ILQR_OK ? ops : nullptr;
}
int sqp_args(const ilqr_handle *h, const ilqr_sqp_opts *o, void *work, const void *u0, bool need_u0, int32_t *qp_iters, SqpArgs &a) {
    if (!o || !work) return ILQR_ERR_INVALID;
    if (o->max_iter < 0 || o->qp_max_iter < 0 || !(o->eps > 0.0) || !(o->cost_eps >= 0.0) || o->qp_tol < 0.0 || o->qp_sigma < 0.0 ||
        o->qp_sigma >= 1.0)
        return ILQR_ERR_INVALID;
    if (need_u0 && o->init_input_fixed && !u0) return ILQR_ERR_NOT_BOUND;
    const SqpLayout l = sqp_layout(h->ctx.d);
    char *w = (char *)work;
    a.S = w + l.S;
    a.LAM = w + l.LAM;
    a.DU = w + l.DU;
    a.DS = w + l.DS;
    a.DL = w + l.DL;
    a.DX = w + l.DX;
    a.UQ = w + l.UQ;
    a.XQ = w + l.XQ;
    a.U0 = o->init_input_fixed ? u0 : nullptr;
    a.X0 = o->x0;
    a.qp_iters = qp_iters;
    a.eps = o->eps;
    a.cost_eps = o->cost_eps;
    a.qp_tol = o->qp_tol > 0.0 ? o->qp_tol : 1e-10;
    a.qp_sigma = o->qp_sigma > 0.0 ? o->qp_sigma : 0.1;
    a.qp_max_iter = o->qp_max_iter > 0 ? o->qp_max_iter : 200;
    a.n_alphas = alphas_from_floor(o->eps); // the step sizes the loop of sqp.py:156-171 can visit
    a.fixed = o->init_input_fixed ? 1 : 0;
    return a.n_alphas > 0 ? ILQR_OK : ILQR_ERR_INVALID;
}
} // namespace

size_t ilqr_sqp_workspace_bytes(const ilqr_handle *h) { return h ? sqp_layout(h->ctx.d).total : 0; }

void *ilqr_sqp_solved_inputs(const ilqr_handle *h, void *work) { return (h && work) ? (char *)work + sqp_layout(h->ctx.d).UQ : nullptr; }

int ilqr_sqp_begin(ilqr_handle *h, void *stream) {
    int rc;
    const SqpOps *ops = sqp_check(h, rc);
    if (!ops) return rc;
    h->ctx.reroll = false;
    return hip_rc(ops->init(h->ctx, (hipStream_t)stream));
}

int ilqr_sqp_qp(ilqr_handle *h, const ilqr_sqp_opts *opts, void *work, const void *u0, int32_t *qp_iters, void *stream) {
    int rc;
    const SqpOps *ops = sqp_check(h, rc);
    if (!ops) return rc;
    SqpArgs a;
    rc = sqp_args(h, opts, work, u0, true, qp_iters, a);
    if (rc != ILQR_OK) return rc;
    return hip_rc(ops->qp(h->ctx, a, (hipStream_t)stream));
}

int ilqr_sqp_search(ilqr_handle *h, const ilqr_sqp_opts *opts, void *work, int32_t iteration, void *stream) {
    int rc;
    const SqpOps *ops = sqp_check(h, rc);
    if (!ops) return rc;
    if (iteration < 0) return ILQR_ERR_INVALID;
    SqpArgs a;
    rc = sqp_args(h, opts, work, nullptr, false, nullptr, a);
    if (rc != ILQR_OK) return rc;
    return hip_rc(ops->search(h->ctx, a, iteration, (hipStream_t)stream));
}

int ilqr_sqp_solve(ilqr_handle *h, const ilqr_sqp_opts *opts, void *work, const void *u0, int32_t *qp_iters, void *stream) {
    // every argument is checked before the first launch: an argument error must not touch the caller's buffers
    int rc;
    if (!sqp_check(h, rc)) return rc;
    SqpArgs a;
    rc = sqp_args(h, opts, work, u0, true, qp_iters, a);
    if (rc != ILQR_OK) return rc;
    rc = ilqr_sqp_begin(h, stream);
    for (int it = 0; rc == ILQR_OK && it < opts->max_iter; ++it) {
        rc = ilqr_sqp_qp(h, opts, work, u0, qp_iters, stream);
        if (rc == ILQR_OK) rc = ilqr_sqp_search(h, opts, work, it, stream);
    }
    return rc;
}

const char *ilqr_strerror(int code) {
    switch (code) {
    case ILQR_OK: return "ok";
    case ILQR_ERR_INVALID: return "invalid argument or descriptor field";
    case ILQR_ERR_UNSUPPORTED: return "model/dtype/shape combination not built into this library";
    case ILQR_ERR_NOT_BOUND: return "a required device buffer is not bound";
    case ILQR_ERR_HIP: return "HIP runtime error (see ilqr_last_hip_error)";
    case ILQR_ERR_TOO_LARGE: return "batch exceeds the static scratch of ilqr_solve_static (ILQR_STATIC_MAX_PROBLEMS problems)";
    case ILQR_ERR_SPARSE_STATES: return "X holds only every 8th state row (recompute_states = 2): call ilqr_states first";
    default: return "unknown error code";
    }
}

int ilqr_last_hip_error(void) { return g_last_hip_error; }
int32_t ilqr_desc_size(void) { return (int32_t)sizeof(ilqr_desc); }
int32_t ilqr_buffers_size(void) { return (int32_t)sizeof(ilqr_buffers); }

} // extern "C"
