"""ilqr_amd: batched iLQR / constrained iLQR on AMD MI355X behind the Python
interface of zetian/ilqr (``systems`` model setup, ``ilqr`` / ``cilqr`` / ``sqp`` solve
entry points, ``constraints``).  The solve runs in hand-written HIP kernels reached through a
C-ABI shared library; importing this package does not touch the GPU."""
from . import systems, problems  # noqa: F401

__all__ = ["systems", "problems", "ilqr", "cilqr", "sqp", "constraints", "user_models", "trace", "jit", "mpc", "dist"]
