"""Host-side driver shared by ``ilqr_amd.ilqr`` and ``ilqr_amd.cilqr``.

Owns the device buffers (PyTorch-ROCm tensors, used only as carriers of device
memory and of the HIP stream), converts between the reference's array layout
and the tiled device layout, and calls the C-ABI library.  All numerical work
happens in the HIP kernels; nothing here computes a cost, a gain or a rollout.
"""
import ctypes as C

import numpy as np

from . import _lib
from .systems import System

_TORCH = None


def _torch():
    global _TORCH
    if _TORCH is None:
        import torch
        _TORCH = torch
    return _TORCH


def _np_dtype(dtype):
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise ValueError("dtype must be float64 or float32")
    return dt


def n_tiles(B):
    return (int(B) + _lib.TILE - 1) // _lib.TILE


_DYNAMICS = ("model_f", "compute_df_dx", "compute_df_du")


def _shipped_dynamics(system):
    """True if ``system`` computes the dynamics of the shipped class that declares its
    ``device_model_id``: neither the class nor the instance replaces one of the three methods the
    reference would call (``ilqr.py:62``, ``:87-88``).  ``class MyCar(Car)`` with its own ``model_f``
    must not be solved with the built-in Car kernels."""
    from . import systems
    for base in (systems.Car, systems.CarAcceleration, systems.DubinsCar, systems.LTVSystem):
        if isinstance(system, base):
            return all(getattr(type(system), name) is getattr(base, name) and name not in vars(system)
                       for name in _DYNAMICS)
    return False


def resolve_model(system, dtype, constraint_mode=_lib.CONSTR_NONE):
    """Model id of the device dynamics of ``system`` for ``ilqr_desc.model``: the built-in one of a
    shipped class, else dynamics compiled at run time for this element type and for diagonal or
    dense cost weights, whichever the system has: from the snippets of a ``systems.DeviceSystem``,
    or for any other System duck type (``systems.py:4-24``) from a record of its numpy methods.
    Recording is repeated on every call (milliseconds) so that a changed parameter the record
    holds as a literal yields another model; compiled libraries are cached by content."""
    np_dtype = _np_dtype(dtype)
    model_id = getattr(system, "device_model_id", None)
    diag = all(np.count_nonzero(W - np.diag(np.diagonal(W))) == 0
               for W in (np.asarray(system.Q, dtype=float), np.asarray(system.R, dtype=float),
                         np.asarray(system.Q_f, dtype=float)))
    if model_id is not None and _shipped_dynamics(system):
        from . import systems
        if isinstance(system, systems.LTVSystem) and not system.builtin_kernels(diag):
            return int(system.device_model(np_dtype, diag))   # a table model of another size, or dense weights
        return int(model_id)
    if getattr(system, "device_step", None) is not None and hasattr(system, "device_model"):
        return int(system.device_model(np_dtype, diag))
    from . import trace
    try:
        # a caller that attaches constraints.py rows vouches for the (px, py, v) state layout
        return int(trace.traced_model(system, np_dtype, diag,
                                      has_barrier=bool(constraint_mode & _lib.CONSTR_BARRIER)
                                      or bool(getattr(system, "device_has_barrier", False))))
    except trace.TraceError as e:
        raise _lib.IlqrError(f"{type(system).__name__} has no device dynamics and its numpy methods "
                             f"could not be recorded: {e}") from e


class DeviceProblem:
    """Device-resident state of one batched solve: the tiled trajectory/gain
    buffers, the per-problem scalars and the library handle."""

    def __init__(self, system, B, N, dtype, device, *, max_iter, lm_parameter, alpha_floor,
                 constraint_mode=_lib.CONSTR_NONE, input_limit_lambda=1e-4, barrier_mu=0.0,
                 barrier_delta=0.1, vel_bounds=(0.0, 0.0), lanes_per_problem=0, alphas_per_lane=0,
                 history=False, solver_limits=False, n_streams=0, repack_every=0, recompute_states=0,
                 overlap_stages=0, model_id=None, limits=None, trig_cache=0, segment_commit=0, fused_solve=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.IlqrError("no GPU visible: the batched iLQR solve runs on the HIP device only")
        self.lib = _lib.lib()
        self.device = torch.device(device if device is not None else "cuda")
        self.np_dtype = _np_dtype(dtype)
        if model_id is None:
            model_id = resolve_model(system, self.np_dtype, constraint_mode)
        self.t_dtype = torch.float64 if self.np_dtype == np.float64 else torch.float32
        self.B, self.N = int(B), int(N)
        self.n, self.m = int(system.state_size), int(system.control_size)
        self.max_iter = int(max_iter)
        n, m, N, B = self.n, self.m, self.N, self.B

        d = _lib.Desc()
        d.B, d.N, d.n, d.m = B, N, n, m
        d.model = int(model_id)
        d.dtype = _lib.F64 if self.np_dtype == np.float64 else _lib.F32
        # `limits` = (lower, upper, limited) replaces the system's control limits (a solver that
        # keeps bounds of its own, sqp.py:38-42)
        d.control_limited = int(bool(system.control_limited if limits is None else limits[2]))
        d.constraint_mode = int(constraint_mode)
        d.n_alphas = 0
        d.lanes_per_problem, d.alphas_per_lane = int(lanes_per_problem), int(alphas_per_lane)
        d.n_streams = int(n_streams)
        d.repack_every = int(repack_every)
        d.recompute_states = int(recompute_states)
        d.overlap_stages = int(overlap_stages)
        d.trig_cache = int(trig_cache)
        d.segment_commit = int(segment_commit)
        d.fused_solve = int(fused_solve)
        d.dt = float(system.dt)
        d.lm_parameter = float(lm_parameter)
        d.alpha_floor = float(alpha_floor)
        d.input_limit_lambda = float(input_limit_lambda)
        d.barrier_mu, d.barrier_delta = float(barrier_mu), float(barrier_delta)
        d.vel_bounds[0], d.vel_bounds[1] = float(vel_bounds[0]), float(vel_bounds[1])
        Q = np.asarray(system.Q, dtype=np.float64)
        R = np.asarray(system.R, dtype=np.float64)
        Qf = np.asarray(system.Q_f, dtype=np.float64)
        if Q.shape != (n, n) or Qf.shape != (n, n) or R.shape != (m, m):
            raise ValueError("Q / Q_f must be (n, n) and R (m, m)")
        for i, v in enumerate(Q.ravel()):
            d.Q[i] = v
        for i, v in enumerate(Qf.ravel()):
            d.Qf[i] = v
        for i, v in enumerate(R.ravel()):
            d.R[i] = v
        if limits is not None or system.control_limited or solver_limits:
            lo, hi = (system.control_lower_limit, system.control_upper_limit) if limits is None else limits[:2]
            for j in range(m):
                d.lo[j], d.hi[j] = float(lo[j]), float(hi[j])
        self.desc = d
        self.handle = C.c_void_p()
        _lib.check(self.lib.ilqr_create(C.byref(d), C.byref(self.handle)), "ilqr_create")
        self.n_alphas = int(self.lib.ilqr_num_alphas(self.handle))

        Bp = n_tiles(B) * _lib.TILE
        dev, td = self.device, self.t_dtype
        with torch.cuda.device(self.device):
            self.X = torch.zeros(Bp * N * n, dtype=td, device=dev)
            self.U = torch.zeros(Bp * (N - 1) * m, dtype=td, device=dev)
            self.Xref = torch.zeros(Bp * N * n, dtype=td, device=dev)
            self.K = torch.zeros(Bp * (N - 1) * m * n, dtype=td, device=dev)
            self.k = torch.zeros(Bp * (N - 1) * m, dtype=td, device=dev)
            self.J = torch.zeros(B, dtype=td, device=dev)
            self.converged = torch.zeros(B, dtype=torch.int32, device=dev)
            self.iters = torch.zeros(B, dtype=torch.int32, device=dev)
            self.alpha_idx = torch.full((B,), -1, dtype=torch.int32, device=dev)
            self.status = torch.zeros(B, dtype=torch.int32, device=dev)
            self.work = torch.zeros(int(self.lib.ilqr_workspace_bytes(self.handle)), dtype=torch.uint8,
                                    device=dev)
            self.centers = self.radius = None
            if constraint_mode & _lib.CONSTR_BARRIER:
                self.centers = torch.zeros(Bp * N * 2, dtype=td, device=dev)
                self.radius = torch.ones(Bp * N, dtype=td, device=dev)
            self.ltv_A = self.ltv_B = None
            if d.model != 3 and getattr(system, "device_tables", None) is not None:  # tables of a run-time compiled model
                ta, tb = system.device_tables
                if hasattr(system, "builtin_kernels") and (np.shape(ta) != (N - 1, n, n) or np.shape(tb) != (N - 1, n, m)):
                    raise ValueError("LTVSystem tables must be A (N-1, n, n) and B (N-1, n, m)")
                self.ltv_A = torch.from_numpy(np.ascontiguousarray(ta, dtype=self.np_dtype)).to(dev)
                self.ltv_B = torch.from_numpy(np.ascontiguousarray(tb, dtype=self.np_dtype)).to(dev)
            if d.model == 3:  # table-driven LTV: the tables are shared by the whole batch
                A = np.ascontiguousarray(system.A, dtype=self.np_dtype)
                Bm = np.ascontiguousarray(system.B, dtype=self.np_dtype)
                if A.shape != (N - 1, n, n) or Bm.shape != (N - 1, n, m):
                    raise ValueError("LTVSystem tables must be A (N-1, n, n) and B (N-1, n, m)")
                self.ltv_A = torch.from_numpy(A).to(dev)
                self.ltv_B = torch.from_numpy(Bm).to(dev)
            self.hist_cost = self.hist_alpha = None
            if history:
                self.hist_cost = torch.full((self.max_iter, B), float("nan"), dtype=td, device=dev)
                self.hist_alpha = torch.full((self.max_iter, B), -2, dtype=torch.int32, device=dev)
        self._bind()

    # -- plumbing ---------------------------------------------------------
    def _bind(self):
        b = _lib.Buffers()
        b.X, b.U, b.Xref = self.X.data_ptr(), self.U.data_ptr(), self.Xref.data_ptr()
        b.K, b.k, b.J = self.K.data_ptr(), self.k.data_ptr(), self.J.data_ptr()
        b.converged, b.iters = self.converged.data_ptr(), self.iters.data_ptr()
        b.alpha_idx, b.status = self.alpha_idx.data_ptr(), self.status.data_ptr()
        b.work = self.work.data_ptr()
        if self.centers is not None:
            b.centers, b.radius = self.centers.data_ptr(), self.radius.data_ptr()
        if self.ltv_A is not None:
            b.ltv_A, b.ltv_B = self.ltv_A.data_ptr(), self.ltv_B.data_ptr()
        if self.hist_cost is not None:
            b.hist_cost, b.hist_alpha = self.hist_cost.data_ptr(), self.hist_alpha.data_ptr()
            b.hist_capacity = int(self.hist_cost.shape[0])
        self.buffers = b
        _lib.check(self.lib.ilqr_bind(self.handle, C.byref(b)), "ilqr_bind")

    def reserve_history(self, max_iter):
        """Make the history buffers (if any) hold ``max_iter`` iterations; trajectories, gains and
        flags stay where they are."""
        if self.hist_cost is not None and max_iter > self.hist_cost.shape[0]:
            torch = _torch()
            self.fill_states()   # (binding again makes the library forget that X holds sparse rows)
            with torch.cuda.device(self.device):
                self.hist_cost = torch.full((max_iter, self.B), float("nan"), dtype=self.t_dtype, device=self.device)
                self.hist_alpha = torch.full((max_iter, self.B), -2, dtype=torch.int32, device=self.device)
            self.max_iter = max_iter
            self._bind()
        self.max_iter = max(self.max_iter, max_iter)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ilqr_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def target_layout(self):
        """Rows interleaved per problem in the device buffer Xref."""
        return int(self.lib.ilqr_target_layout(self.handle))

    def _upload(self, host, tiled, rows, vec=1):
        """host: array-like (B, rows...) in the reference layout -> tiled device buffer."""
        torch = _torch()
        if isinstance(host, torch.Tensor):
            src = host.to(device=self.device, dtype=self.t_dtype).reshape(self.B, rows).contiguous()
        else:
            a = np.ascontiguousarray(np.asarray(host, dtype=self.np_dtype).reshape(self.B, rows))
            if not a.flags.writeable:   # (a view of a read-only array: torch wants to be able to write)
                a = a.copy()
            src = torch.from_numpy(a).to(self.device)
        _lib.check(self.lib.ilqr_to_tiled_v(src.data_ptr(), tiled.data_ptr(), self.B, rows,
                                            self.np_dtype.itemsize, vec, self.stream()), "ilqr_to_tiled_v")
        # `src` must outlive the asynchronous kernel: both run on the current stream, and torch's
        # caching allocator only reuses the block for later work on that same stream.

    def _download(self, tiled, rows, shape, vec=1):
        torch = _torch()
        dst = torch.empty(self.B * rows, dtype=self.t_dtype, device=self.device)
        _lib.check(self.lib.ilqr_from_tiled_v(tiled.data_ptr(), dst.data_ptr(), self.B, rows,
                                              self.np_dtype.itemsize, vec, self.stream()), "ilqr_from_tiled_v")
        return dst.cpu().numpy().reshape(shape)

    def gain_layout(self):
        """Rows interleaved per problem in the device buffers K and k."""
        vK, vk = C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.ilqr_gain_layout(self.handle, C.byref(vK), C.byref(vk)), "ilqr_gain_layout")
        return int(vK.value), int(vk.value)

    # -- data movement ------------------------------------------------------
    def upload(self, target_states, states, inputs, converged=None, centers=None, radius=None):
        N, n, m = self.N, self.n, self.m
        self.fill_states()   # (the library must not hold sparse states of its own when X is rewritten)
        self._upload(target_states, self.Xref, N * n, vec=self.target_layout())
        self._upload(states, self.X, N * n)
        self._upload(inputs, self.U, (N - 1) * m)
        torch = _torch()
        if converged is not None:
            c = np.ascontiguousarray(np.asarray(converged).reshape(self.B).astype(np.int32))
            self.converged.copy_(torch.from_numpy(c))
        if self.centers is not None:
            if centers is None or radius is None:
                raise ValueError("barrier mode needs centers (B, N, 2) and radius (B, N)")
            self._upload(centers, self.centers, N * 2)
            self._upload(radius, self.radius, N)

    def download_states(self):
        self.fill_states()   # (a no-op unless sparse commits have run between phase calls)
        return self._download(self.X, self.N * self.n, (self.B, self.N, self.n))

    def download_inputs(self):
        return self._download(self.U, (self.N - 1) * self.m, (self.B, self.N - 1, self.m))

    def download_K(self):
        return self._download(self.K, (self.N - 1) * self.m * self.n, (self.B, self.N - 1, self.m, self.n),
                              vec=self.gain_layout()[0])

    def download_k(self):
        return self._download(self.k, (self.N - 1) * self.m, (self.B, self.N - 1, self.m, 1), vec=self.gain_layout()[1])

    # -- solver phases (asynchronous on the current stream) -----------------
    def initial_cost(self):
        _lib.check(self.lib.ilqr_initial_cost(self.handle, self.stream()), "ilqr_initial_cost")

    def backward(self):
        _lib.check(self.lib.ilqr_backward(self.handle, self.stream()), "ilqr_backward")

    def forward(self, iteration=0):
        _lib.check(self.lib.ilqr_forward(self.handle, int(iteration), self.stream()), "ilqr_forward")

    def forward_evaluate(self):
        _lib.check(self.lib.ilqr_forward_evaluate(self.handle, self.stream()), "ilqr_forward_evaluate")

    def repack(self):
        _lib.check(self.lib.ilqr_repack(self.handle, self.stream()), "ilqr_repack")

    def fill_states(self):
        """Bring X up to date (handles created with ``recompute_states=2`` keep only every 8th state
        row between the phase calls)."""
        _lib.check(self.lib.ilqr_states(self.handle, self.stream()), "ilqr_states")

    def unpack(self):
        _lib.check(self.lib.ilqr_unpack(self.handle, self.stream()), "ilqr_unpack")

    def forward_mapping(self):
        """(lanes per problem, step sizes per lane) of the first line-search stage."""
        L, A = C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.ilqr_forward_mapping(self.handle, C.byref(L), C.byref(A)), "ilqr_forward_mapping")
        return int(L.value), int(A.value)

    def sparse_states(self):
        """0: commits store every state row; 1: every 8th inside ``solve`` only; 2: also between phase calls."""
        return int(self.lib.ilqr_sparse_states(self.handle))

    def num_alphas(self):
        return int(self.lib.ilqr_num_alphas(self.handle))

    def forward_evaluate_stage(self, stage):
        _lib.check(self.lib.ilqr_forward_evaluate_stage(self.handle, int(stage), self.stream()),
                   "ilqr_forward_evaluate_stage")

    def forward_commit(self, iteration=0):
        _lib.check(self.lib.ilqr_forward_commit(self.handle, int(iteration), self.stream()),
                   "ilqr_forward_commit")

    def solve(self, max_iter=None):
        mi = self.max_iter if max_iter is None else int(max_iter)
        _lib.check(self.lib.ilqr_solve(self.handle, mi, self.stream()), "ilqr_solve")

    def solve_static(self, max_iter=None):
        """``ilqr_solve`` on the library's own static scratch (allocates nothing)."""
        mi = self.max_iter if max_iter is None else int(max_iter)
        _lib.check(self.lib.ilqr_solve_static(self.handle, mi, self.stream()), "ilqr_solve_static")

    def plain_cost(self):
        """Cost of the current device X, U for every problem (``ilqr.py:36-46`` / ``sqp.py:44-54``)."""
        self.initial_cost()
        return self.J.cpu().numpy().astype(np.float64)

    def rollout(self, states, inputs):
        """``sqp.py:56-61``: upload ``states`` (only step 0 matters) and ``inputs``, roll out on the
        device, return the states ``(B, N, n)``."""
        self._upload(states, self.X, self.N * self.n)
        self._upload(inputs, self.U, (self.N - 1) * self.m)
        _lib.check(self.lib.ilqr_rollout(self.handle, self.stream()), "ilqr_rollout")
        return self.download_states()

    def active_count(self):
        out = C.c_int32(0)
        _lib.check(self.lib.ilqr_active_count(self.handle, self.stream(), C.byref(out)), "ilqr_active_count")
        return out.value


class BatchedSolverBase:
    """Attribute surface of the reference solver object (``ilqr.py:17-34``) with a
    leading batch axis.  ``target_states`` may be ``(N, n)`` (one problem, exactly
    the reference call) or ``(B, N, n)``."""

    _default_max_iter = 30
    _constraint_mode = _lib.CONSTR_NONE

    def __init__(self, sys, target_states, dt, device=None, dtype=np.float64, history=False,
                 lanes_per_problem=0, alphas_per_lane=0, n_streams=0, repack_every=0, recompute_states=0,
                 overlap_stages=0, trig_cache=0, segment_commit=0, fused_solve=0):
        if not isinstance(sys, System) and not hasattr(sys, "state_size"):
            raise TypeError("sys must provide the systems.System interface")
        self.target_states = target_states
        ts = np.asarray(target_states)
        if ts.ndim == 2:
            self._batched = False
            self._B = 1
        elif ts.ndim == 3:
            self._batched = True
            self._B = ts.shape[0]
        else:
            raise ValueError("target_states must be (N, n) or (B, N, n)")
        self.horizon = ts.shape[-2]
        if ts.shape[-1] != sys.state_size:
            raise ValueError("target_states last axis must equal sys.state_size")
        self.dt = dt  # kept for interface parity; the dynamics use sys.dt (ilqr.py:20)
        self.system = sys
        self.n_states = sys.state_size
        self.m_inputs = sys.control_size
        self.Q = sys.Q
        self.R = sys.R
        self.Qf = sys.Q_f
        self.maxIter = self._default_max_iter
        self.LM_parameter = 0.0
        lead = (self._B,) if self._batched else ()
        self.min_cost = np.zeros(lead) if self._batched else 0.0
        self.converge = np.zeros(lead, dtype=bool) if self._batched else False
        self.states = np.zeros(lead + (self.horizon, self.n_states))
        self.inputs = np.zeros(lead + (self.horizon - 1, self.m_inputs))
        self.iterations = np.zeros(lead, dtype=np.int32) if self._batched else 0
        self.status = np.zeros(lead, dtype=np.int32) if self._batched else 0
        self._device = device
        self._dtype = _np_dtype(dtype)
        self._history = bool(history)
        self._mapping = (int(lanes_per_problem), int(alphas_per_lane))
        self._n_streams = int(n_streams)
        self._repack_every = int(repack_every)
        self._recompute_states = int(recompute_states)
        self._overlap_stages = int(overlap_stages)
        self._trig_cache = int(trig_cache)
        self._segment_commit = int(segment_commit)
        self._fused_solve = int(fused_solve)
        self._dev = None
        self._k_cache = self._K_cache = None
        self.cost_history = None
        self.alpha_history = None

    # -- hooks for the constrained variant ---------------------------------
    def _alpha_floor(self):
        return 1e-4  # ilqr.py:71

    def _extra_desc(self):
        return {}

    def _extra_upload(self):
        return {}

    # -- device life cycle ----------------------------------------------------
    def _key(self):
        s = self.system
        lim = (tuple(s.control_lower_limit), tuple(s.control_upper_limit)) if s.control_limited else None
        self._model_id = resolve_model(s, self._dtype, self._constraint_mode)
        return (self._B, self.horizon, self._model_id, float(self.LM_parameter), float(s.dt),
                np.asarray(self.Q).tobytes(), np.asarray(self.R).tobytes(), np.asarray(self.Qf).tobytes(),
                bool(s.control_limited), lim, self._alpha_floor(), repr(sorted(self._extra_desc().items())),
                id(getattr(s, "A", None)), id(getattr(s, "B", None)))

    def _ensure_device(self):
        key = self._key()
        if self._dev is None or self._dev_key != key:
            if self._dev is not None:
                self._dev.close()
            self._dev = DeviceProblem(
                self.system, self._B, self.horizon, self._dtype, self._device, max_iter=self.maxIter,
                lm_parameter=self.LM_parameter, alpha_floor=self._alpha_floor(),
                constraint_mode=self._constraint_mode, history=self._history,
                lanes_per_problem=self._mapping[0], alphas_per_lane=self._mapping[1],
                n_streams=self._n_streams, repack_every=self._repack_every,
                recompute_states=self._recompute_states, overlap_stages=self._overlap_stages,
                trig_cache=self._trig_cache, segment_commit=self._segment_commit, fused_solve=self._fused_solve, model_id=self._model_id,
                **self._extra_desc())
            self._dev_key = key
        # maxIter may change between calls (the iteration count is an argument of ilqr_solve); only
        # the history buffers are sized by it, and growing them keeps everything else on the device
        self._dev.reserve_history(int(self.maxIter))
        return self._dev

    def _lead(self, a, tail_ndim):
        a = np.asarray(a)
        return a if a.ndim == tail_ndim + 1 else a[None]

    def upload(self):
        """Copy targets, the current states/inputs guess and the converge flags to the device."""
        dev = self._ensure_device()
        dev.upload(self._lead(self.target_states, 2), self._lead(self.states, 2), self._lead(self.inputs, 2),
                   converged=np.atleast_1d(np.asarray(self.converge)), **self._extra_upload())
        dev.iters.zero_()
        dev.status.zero_()
        if dev.hist_cost is not None:   # (rows of an earlier call must not show through)
            dev.hist_cost.fill_(float("nan"))
            dev.hist_alpha.fill_(-2)
        self._k_cache = self._K_cache = None
        return dev

    def solve_device(self):
        """Run ``__call__`` on the device-resident data without any host transfer."""
        self._ensure_device().solve(self.maxIter)

    def download(self):
        dev = self._dev
        st, ui = dev.download_states(), dev.download_inputs()
        J = dev.J.cpu().numpy().astype(np.float64)
        cv = dev.converged.cpu().numpy().astype(bool)
        it = dev.iters.cpu().numpy()
        stt = dev.status.cpu().numpy()
        if self._history:
            mi = int(self.maxIter)   # (the buffers only ever grow)
            self.cost_history = dev.hist_cost[:mi].cpu().numpy().T.copy()     # (B, maxIter)
            self.alpha_history = dev.hist_alpha[:mi].cpu().numpy().T.copy()
        # the reference mutates the caller's arrays in place (ilqr.py:56, :62): write back into
        # the same objects when they are numpy arrays of the right shape
        if self._batched:
            self._assign("states", st)
            self._assign("inputs", ui)
            self.min_cost, self.converge, self.iterations, self.status = J, cv, it, stt
        else:
            self._assign("states", st[0])
            self._assign("inputs", ui[0])
            self.min_cost, self.converge = float(J[0]), bool(cv[0])
            self.iterations, self.status = int(it[0]), int(stt[0])
            if self._history:
                self.cost_history, self.alpha_history = self.cost_history[0], self.alpha_history[0]

    def _assign(self, name, value):
        cur = getattr(self, name)
        if isinstance(cur, np.ndarray) and cur.shape == value.shape and cur.flags.writeable:
            cur[...] = value
        else:
            setattr(self, name, value.astype(np.float64))

    # -- the reference entry point -----------------------------------------------
    def __call__(self):
        """``ilqr.py:107-114``: initial cost, then up to ``maxIter`` backward/forward
        iterations per problem; returns ``self.states``."""
        self.upload()
        self.solve_device()
        self.download()
        return self.states

    # -- the reference's phase methods, each one device launch over the batch --------------------
    def cost(self):
        """``ilqr.py:36-46`` for the current ``states`` / ``inputs``: scalar, or ``(B,)``."""
        dev = self.upload()
        dev.initial_cost()
        J = dev.J.cpu().numpy().astype(np.float64)
        return J if self._batched else float(J[0])

    def backward_pass(self):
        """``ilqr.py:76-105``: gains ``k`` / ``K`` of the current ``states`` / ``inputs``."""
        dev = self.upload()
        dev.converged.zero_()        # the reference's backward_pass does not look at `converge`
        dev.backward()

    def forward_pass(self):
        """``ilqr.py:48-74``: line search with the gains of the last ``backward_pass`` against
        ``min_cost``; updates ``states``, ``inputs``, ``min_cost``, ``converge`` (in the reference's
        call order: ``min_cost = cost()``, ``backward_pass()``, ``forward_pass()``)."""
        torch = _torch()
        if self._dev is None:
            raise _lib.IlqrError("forward_pass needs the gains of a backward_pass")
        dev = self._dev
        K, k = dev.K.clone(), dev.k.clone()          # upload() may rebuild nothing, but keep the order explicit
        dev = self.upload()
        dev.K.copy_(K)
        dev.k.copy_(k)
        dev.initial_cost()                            # also tells the handle that X was written by the caller
        mc = np.atleast_1d(np.asarray(self.min_cost, dtype=self._dtype))
        dev.J.copy_(torch.from_numpy(np.ascontiguousarray(mc)))
        dev.forward(0)
        self.download()

    # gains of the last backward pass (ilqr.py:77-78), fetched on demand
    @property
    def k(self):
        if self._k_cache is None and self._dev is not None:
            v = self._dev.download_k()
            self._k_cache = v if self._batched else v[0]
        return self._k_cache

    @property
    def K(self):
        if self._K_cache is None and self._dev is not None:
            v = self._dev.download_K()
            self._K_cache = v if self._batched else v[0]
        return self._K_cache
