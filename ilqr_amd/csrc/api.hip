// C ABI of the batched iLQR library (declared in include/ilqr_amd.h).
// Host-only logic: descriptor validation, dispatch to the kernel tables and
// the iteration driver.  Nothing here allocates device memory.
#include "host.hpp"

#include <cmath>
#include <cstring>
#include <new>

using namespace ilqr;

constexpr int MAX_PARTS = 8;

struct ilqr_handle {
    HostCtx ctx;
    const ModelOps *ops = nullptr;
    // ilqr_solve may split the batch into parts that run on streams of their own, so that one
    // part's arithmetic-bound line search overlaps another part's bandwidth-bound passes
    int n_parts = 1;
    bool streams_ready = false;
    hipStream_t part_stream[MAX_PARTS] = {};
    hipEvent_t fork_event = nullptr, join_event[MAX_PARTS] = {};
};

static thread_local int g_last_hip_error = 0;

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

template <bool TO_TILED> static int relayout(const void *src, void *dst, int B, int rows, int elem, void *stream) {
    if (!src || !dst || B <= 0 || rows <= 0) return ILQR_ERR_INVALID;
    dim3 grid((rows + 63) / 64, (B + 63) / 64), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (elem == 8)
        hipLaunchKernelGGL((relayout_kernel<double, TO_TILED>), grid, block, 0, s, (const double *)src, (double *)dst, B, rows);
    else if (elem == 4)
        hipLaunchKernelGGL((relayout_kernel<float, TO_TILED>), grid, block, 0, s, (const float *)src, (float *)dst, B, rows);
    else
        return ILQR_ERR_INVALID;
    return hip_rc(hipGetLastError());
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
    bool diag = is_diag(d.Q, d.n) && is_diag(d.Qf, d.n) && is_diag(d.R, d.m);
    switch (d.model) {
    case ILQR_MODEL_CAR: want_n = 4; want_m = 2; ops = ops_car(d.dtype, diag); break;
    case ILQR_MODEL_CARACC: want_n = 5; want_m = 2; ops = ops_caracc(d.dtype, diag); break;
    case ILQR_MODEL_DUBINS: want_n = 3; want_m = 2; ops = ops_dubins(d.dtype, diag); break;
    case ILQR_MODEL_LTV:
        // table-driven x+ = A_t x + B_t u; built for n = 12, m = 4 with diagonal cost weights
        if (d.n != 12 || d.m != 4) return ILQR_ERR_UNSUPPORTED;
        want_n = 12; want_m = 4; ops = d.dtype == ILQR_F64 ? ops_ltv12x4_f64(diag) : ops_ltv12x4_f32(diag);
        break;
    default: return ILQR_ERR_UNSUPPORTED;
    }
    if (d.n != want_n || d.m != want_m) return ILQR_ERR_INVALID;
    if (!ops) return ILQR_ERR_UNSUPPORTED;
    if ((d.constraint_mode & ILQR_CONSTR_BARRIER)) {
        // the constraint functions are written for the Car state layout (constraints.py:12)
        if (d.model != ILQR_MODEL_CAR) return ILQR_ERR_UNSUPPORTED;
        if (!(d.barrier_mu >= 0.0) || !(d.barrier_delta > 0.0)) return ILQR_ERR_INVALID;
    }
    if ((d.constraint_mode & ILQR_CONSTR_BOXQP) || d.control_limited)
        for (int j = 0; j < d.m; ++j)
            if (!(d.lo[j] <= d.hi[j])) return ILQR_ERR_INVALID;
    int na = d.n_alphas > 0 ? d.n_alphas : alphas_from_floor(d.alpha_floor);
    if (na <= 0 || na > 64) return ILQR_ERR_INVALID;
    int L = d.lanes_per_problem, A = d.alphas_per_lane;
    if (L == 0 && A == 0) {
        // All 16 candidates of a round are rolled out at once; how they are spread over lanes is a
        // pure speed choice.  Few lanes per problem means more independent rollouts per lane
        // (instruction-level parallelism, shared row loads); many lanes per problem means more
        // waves, which small batches need to occupy the 1024 SIMDs.  Aim at >= 2048 waves.
        L = d.n > 8 ? 8 : 2;
        while (L < 16 && (long)d.B * L < 131072L) L *= 2;
        A = 16 / L;
    } else if (L == 0 || A == 0) {
        return ILQR_ERR_INVALID;
    }
    {
        const int key = L * 100 + A;
        const int ok[] = {1601, 802, 404, 208, 116, 801, 401, 104, 204};
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
    }
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
    delete h;
}

static size_t jtry_bytes(const ilqr_handle *h) {
    const size_t e = h->ctx.d.dtype == ILQR_F64 ? 8 : 4;
    return ((size_t)h->ctx.d.B * e + 255) & ~(size_t)255;
}

static size_t flag_bytes(const ilqr_handle *h) { return ((size_t)h->ctx.d.B * 4 + 255) & ~(size_t)255; }

// layout of `work`: [ trial cost of the accepted step size, per problem | slow-trigonometry flag,
// per problem (int32) | 256 B of counters ]
size_t ilqr_workspace_bytes(const ilqr_handle *h) { return h ? jtry_bytes(h) + flag_bytes(h) + 256 : 0; }

int32_t ilqr_num_alphas(const ilqr_handle *h) { return h ? h->ctx.n_alphas : 0; }

int ilqr_bind(ilqr_handle *h, const ilqr_buffers *b) {
    if (!h || !b) return ILQR_ERR_INVALID;
    if (!b->X || !b->U || !b->Xref || !b->K || !b->k || !b->J || !b->converged || !b->iters || !b->alpha_idx || !b->work)
        return ILQR_ERR_NOT_BOUND;
    if ((h->ctx.d.constraint_mode & ILQR_CONSTR_BARRIER) && (!b->centers || !b->radius)) return ILQR_ERR_NOT_BOUND;
    if (h->ctx.d.model == ILQR_MODEL_LTV && (!b->ltv_A || !b->ltv_B)) return ILQR_ERR_NOT_BOUND;
    if ((b->hist_cost || b->hist_alpha) && b->hist_capacity <= 0) return ILQR_ERR_INVALID;
    h->ctx.b = *b;
    h->ctx.bound = true;
    return ILQR_OK;
}

#define REQUIRE_BOUND(h)                          \
    if (!(h)) return ILQR_ERR_INVALID;            \
    if (!(h)->ctx.bound) return ILQR_ERR_NOT_BOUND;

int ilqr_initial_cost(ilqr_handle *h, void *stream) {
    REQUIRE_BOUND(h);
    return hip_rc(h->ops->cost(h->ctx, 0, (hipStream_t)stream));
}

int ilqr_backward(ilqr_handle *h, void *stream) {
    REQUIRE_BOUND(h);
    return hip_rc(h->ops->backward(h->ctx, 0, (hipStream_t)stream));
}

int ilqr_forward_evaluate(ilqr_handle *h, void *stream) {
    REQUIRE_BOUND(h);
    return hip_rc(h->ops->evaluate(h->ctx, 0, (hipStream_t)stream));
}

int ilqr_forward_commit(ilqr_handle *h, int32_t iteration, void *stream) {
    REQUIRE_BOUND(h);
    if (iteration < 0) return ILQR_ERR_INVALID;
    return hip_rc(h->ops->commit(h->ctx, iteration, (hipStream_t)stream));
}

int ilqr_forward(ilqr_handle *h, int32_t iteration, void *stream) {
    int rc = ilqr_forward_evaluate(h, stream);
    if (rc != ILQR_OK) return rc;
    return ilqr_forward_commit(h, iteration, stream);
}

static int solve_part(ilqr_handle *h, const HostCtx &c, int max_iter, hipStream_t s) {
    int rc = hip_rc(h->ops->cost(c, 0, s));
    for (int it = 0; it < max_iter && rc == ILQR_OK; ++it) {
        rc = hip_rc(h->ops->backward(c, 0, s));
        if (rc == ILQR_OK) rc = hip_rc(h->ops->evaluate(c, 0, s));
        if (rc == ILQR_OK) rc = hip_rc(h->ops->commit(c, it, s));
    }
    return rc;
}

int ilqr_solve(ilqr_handle *h, int32_t max_iter, void *stream) {
    REQUIRE_BOUND(h);
    if (max_iter < 0) return ILQR_ERR_INVALID;
    hipStream_t user = (hipStream_t)stream;
    if (h->n_parts <= 1) return solve_part(h, h->ctx, max_iter, user);
    // fork: every part waits for the caller's stream, runs its whole solve on a stream of its
    // own (parts are disjoint tile ranges, so they never touch the same memory), and the
    // caller's stream waits for all parts.  No host synchronisation.
    if (!h->streams_ready) {
        hipError_t e = hipEventCreateWithFlags(&h->fork_event, hipEventDisableTiming);
        for (int s = 0; s < h->n_parts && e == hipSuccess; ++s) {
            e = hipStreamCreateWithFlags(&h->part_stream[s], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->join_event[s], hipEventDisableTiming);
        }
        if (e != hipSuccess) return hip_rc(e);
        h->streams_ready = true;
    }
    hipError_t e = hipEventRecord(h->fork_event, user);
    if (e != hipSuccess) return hip_rc(e);
    const int tiles = (h->ctx.d.B + TILE - 1) / TILE;
    int rc = ILQR_OK;
    // launch order: iteration-major across parts, so that the parts advance together and the
    // hardware queues always hold kernels of different phases
    HostCtx part[MAX_PARTS];
    for (int s = 0; s < h->n_parts; ++s) {
        part[s] = h->ctx;
        const int t0 = (int)((long)tiles * s / h->n_parts), t1 = (int)((long)tiles * (s + 1) / h->n_parts);
        part[s].first = t0 * TILE;
        const int end = t1 * TILE < h->ctx.d.B ? t1 * TILE : h->ctx.d.B;
        part[s].count = end - part[s].first;
        e = hipStreamWaitEvent(h->part_stream[s], h->fork_event, 0);
        if (e != hipSuccess) return hip_rc(e);
        rc = hip_rc(h->ops->cost(part[s], 0, h->part_stream[s]));
        if (rc != ILQR_OK) return rc;
    }
    // The line search is arithmetic-bound, the backward pass and the commit are bandwidth-bound.
    // A token (an event chain) lets only one part run its line search at a time, which pushes the
    // parts into opposite phases: while one computes, the others stream.
    int prev = -1;
    for (int it = 0; it < max_iter && rc == ILQR_OK; ++it)
        for (int s = 0; s < h->n_parts && rc == ILQR_OK; ++s) {
            rc = hip_rc(h->ops->backward(part[s], 0, h->part_stream[s]));
            if (rc == ILQR_OK && prev >= 0) rc = hip_rc(hipStreamWaitEvent(h->part_stream[s], h->join_event[prev], 0));
            if (rc == ILQR_OK) rc = hip_rc(h->ops->evaluate(part[s], 0, h->part_stream[s]));
            if (rc == ILQR_OK) rc = hip_rc(hipEventRecord(h->join_event[s], h->part_stream[s]));
            prev = s;
            if (rc == ILQR_OK) rc = hip_rc(h->ops->commit(part[s], it, h->part_stream[s]));
        }
    for (int s = 0; s < h->n_parts; ++s) {
        e = hipEventRecord(h->join_event[s], h->part_stream[s]);
        if (e == hipSuccess) e = hipStreamWaitEvent(user, h->join_event[s], 0);
        if (e != hipSuccess && rc == ILQR_OK) rc = hip_rc(e);
    }
    return rc;
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
    REQUIRE_BOUND(h);
    if (!out) return ILQR_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int32_t *counter = (int32_t *)((char *)h->ctx.b.work + jtry_bytes(h) + flag_bytes(h));
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(int32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(count_active_kernel, dim3((h->ctx.d.B + 255) / 256), dim3(256), 0, s, h->ctx.b.converged, h->ctx.d.B, counter);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, counter, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return hip_rc(e);
}

int ilqr_to_tiled(const void *aos, void *tiled, int32_t B, int32_t rows, int32_t elem_bytes, void *stream) {
    return relayout<true>(aos, tiled, B, rows, elem_bytes, stream);
}

int ilqr_from_tiled(const void *tiled, void *aos, int32_t B, int32_t rows, int32_t elem_bytes, void *stream) {
    return relayout<false>(tiled, aos, B, rows, elem_bytes, stream);
}

const char *ilqr_strerror(int code) {
    switch (code) {
    case ILQR_OK: return "ok";
    case ILQR_ERR_INVALID: return "invalid argument or descriptor field";
    case ILQR_ERR_UNSUPPORTED: return "model/dtype/shape combination not built into this library";
    case ILQR_ERR_NOT_BOUND: return "a required device buffer is not bound";
    case ILQR_ERR_HIP: return "HIP runtime error (see ilqr_last_hip_error)";
    default: return "unknown error code";
    }
}

int ilqr_last_hip_error(void) { return g_last_hip_error; }
int32_t ilqr_desc_size(void) { return (int32_t)sizeof(ilqr_desc); }
int32_t ilqr_buffers_size(void) { return (int32_t)sizeof(ilqr_buffers); }

} // extern "C"
