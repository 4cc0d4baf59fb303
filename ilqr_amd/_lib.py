"""ctypes binding of the C-ABI library ``libilqr_amd.so`` (include/ilqr_amd.h).

There is no CPU fallback: if the shared library is missing, or a call returns
an error code, this module raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C ilqr_amd/csrc``.
"""
import ctypes as C
import os

MAX_N, MAX_M = 12, 4
MODEL_USER_BASE = 16
F64, F32 = 0, 1
CONSTR_NONE, CONSTR_BOXQP, CONSTR_BARRIER = 0, 1, 2
STATIC_MAX_PROBLEMS = 65536
STATUS_NONPD, STATUS_SINGULAR, STATUS_NONFINITE, STATUS_QP_FAILED = 1, 2, 4, 8
TILE = 64

_HERE = os.path.dirname(os.path.abspath(__file__))
# ILQR_AMD_LIB selects another build of the same library (used to A/B compiler options)
LIB_PATH = os.environ.get("ILQR_AMD_LIB") or os.path.join(_HERE, "libilqr_amd.so")


class IlqrError(RuntimeError):
    pass


class Desc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
        ("model", C.c_int32), ("dtype", C.c_int32), ("control_limited", C.c_int32),
        ("constraint_mode", C.c_int32), ("n_alphas", C.c_int32),
        ("lanes_per_problem", C.c_int32), ("alphas_per_lane", C.c_int32),
        ("n_streams", C.c_int32), ("repack_every", C.c_int32), ("recompute_states", C.c_int32),
        ("overlap_stages", C.c_int32), ("trig_cache", C.c_int32),
        ("segment_commit", C.c_int32), ("fused_solve", C.c_int32),
        ("dt", C.c_double), ("lm_parameter", C.c_double), ("alpha_floor", C.c_double),
        ("input_limit_lambda", C.c_double), ("barrier_mu", C.c_double),
        ("barrier_delta", C.c_double), ("vel_bounds", C.c_double * 2),
        ("Q", C.c_double * (MAX_N * MAX_N)), ("R", C.c_double * (MAX_M * MAX_M)),
        ("Qf", C.c_double * (MAX_N * MAX_N)),
        ("lo", C.c_double * MAX_M), ("hi", C.c_double * MAX_M),
    ]


class Buffers(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("U", C.c_void_p), ("Xref", C.c_void_p),
        ("K", C.c_void_p), ("k", C.c_void_p), ("J", C.c_void_p),
        ("converged", C.c_void_p), ("iters", C.c_void_p), ("alpha_idx", C.c_void_p),
        ("status", C.c_void_p), ("work", C.c_void_p),
        ("centers", C.c_void_p), ("radius", C.c_void_p),
        ("ltv_A", C.c_void_p), ("ltv_B", C.c_void_p),
        ("hist_cost", C.c_void_p), ("hist_alpha", C.c_void_p),
        ("hist_capacity", C.c_int32), ("reserved0", C.c_int32),
    ]


class SqpOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("init_input_fixed", C.c_int32), ("qp_max_iter", C.c_int32),
                ("reserved", C.c_int32), ("eps", C.c_double), ("cost_eps", C.c_double), ("qp_tol", C.c_double),
                ("qp_sigma", C.c_double), ("x0", C.c_void_p)]


_SIGNATURES = {
    "ilqr_create": (C.c_int, [C.POINTER(Desc), C.POINTER(C.c_void_p)]),
    "ilqr_destroy": (None, [C.c_void_p]),
    "ilqr_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "ilqr_num_alphas": (C.c_int32, [C.c_void_p]),
    "ilqr_bind": (C.c_int, [C.c_void_p, C.POINTER(Buffers)]),
    "ilqr_initial_cost": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_backward": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ilqr_forward_evaluate": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_forward_evaluate_stage": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ilqr_repack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_unpack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_shift": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ilqr_forward_mapping": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ilqr_sparse_states": (C.c_int32, [C.c_void_p]),
    "ilqr_forward_commit": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ilqr_solve": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ilqr_solve_static": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ilqr_rollout": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_states": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_active_count": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "ilqr_to_tiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ilqr_from_tiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ilqr_to_tiled_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ilqr_from_tiled_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ilqr_gain_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ilqr_target_layout": (C.c_int32, [C.c_void_p]),
    "ilqr_load_model": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "ilqr_model_dims": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ilqr_model_probe": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ilqr_sqp_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "ilqr_sqp_solved_inputs": (C.c_void_p, [C.c_void_p, C.c_void_p]),
    "ilqr_sqp_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ilqr_sqp_qp": (C.c_int, [C.c_void_p, C.POINTER(SqpOpts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ilqr_sqp_search": (C.c_int, [C.c_void_p, C.POINTER(SqpOpts), C.c_void_p, C.c_int32, C.c_void_p]),
    "ilqr_sqp_solve": (C.c_int, [C.c_void_p, C.POINTER(SqpOpts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ilqr_strerror": (C.c_char_p, [C.c_int]),
    "ilqr_last_hip_error": (C.c_int, []),
    "ilqr_desc_size": (C.c_int32, []),
    "ilqr_buffers_size": (C.c_int32, []),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load the shared library once; raise if it is absent or ABI-mismatched."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IlqrError(
                f"{LIB_PATH} not found: the HIP library is not built "
                "(run __graft_entry__.build() or make -C ilqr_amd/csrc); there is no CPU fallback")
        # PyTorch-ROCm ships its own copy of the HIP runtime; load it first so that this
        # library binds to the same runtime instance as the tensors that carry its buffers
        # (two runtimes in one process do not share devices, streams or allocations).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.ilqr_desc_size() != C.sizeof(Desc) or handle.ilqr_buffers_size() != C.sizeof(Buffers):
            raise IlqrError("ilqr_desc / ilqr_buffers layout differs between Python and the library")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        l = lib()
        msg = l.ilqr_strerror(rc).decode()
        if rc == -4:
            msg += f" [hipError {l.ilqr_last_hip_error()}]"
        raise IlqrError(f"{what} failed: {msg} ({rc})")
