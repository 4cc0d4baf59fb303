"""Receding-horizon (MPC) driver: repeated solves with shifted warm starts.

The reference warm-starts by assigning ``optimizer.inputs`` (``ilqr_demo.py:49``) and calling the
solver object again continues from whatever its attributes hold (``ilqr.py:107-111``).  Doing that
through the host arrays would move every trajectory over PCIe twice per control step; this driver
keeps all buffers on the device, shifts the previous solution there (``ilqr_shift``) and uploads
the next targets on a side stream while the current solve is still running.
"""
import ctypes as C

import numpy as np

from . import _lib


class RecedingHorizon:
    """``mpc = RecedingHorizon(opt)`` for a batched ``iterative_LQR`` whose ``inputs`` / ``states`` /
    ``target_states`` hold the first problem instance.  Then per control step::

        mpc.prefetch_targets(next_targets)     # optional, overlaps with the running solve
        u0 = mpc.step(shift=1, x0=measured)    # (B, m) inputs to apply now

    ``step`` without a prefetch keeps the current targets.  ``mpc.solution()`` downloads the full
    trajectories of the last solve."""

    def __init__(self, opt):
        import torch
        self.torch = torch
        self.opt = opt
        self.dev = opt.upload()
        d = self.dev
        self.B, self.N, self.n, self.m = d.B, d.N, d.n, d.m
        self._side = torch.cuda.Stream(device=d.device)
        self._staged = torch.empty(self.B * self.N * self.n, dtype=d.t_dtype, device=d.device)   # next targets, reference layout
        self._pinned = torch.empty(self.B * self.N * self.n, dtype=d.t_dtype).pin_memory()
        self._x0 = torch.empty(self.B * self.n, dtype=d.t_dtype, device=d.device)
        self._ready = None      # event: the staged targets have arrived
        self._consumed = None   # event: the staged targets have been converted into the solver's buffer
        self.steps = 0

    def prefetch_targets(self, target_states):
        """Start the host-to-device copy of the next ``target_states (B, N, n)`` on a side stream."""
        torch = self.torch
        a = np.ascontiguousarray(np.asarray(target_states, dtype=self.dev.np_dtype).reshape(-1))
        if a.size != self._pinned.numel():
            raise ValueError("target_states must have shape (B, N, n)")
        if self._ready is not None:
            self._ready.synchronize()           # the pinned buffer is free again
        self._pinned.copy_(torch.from_numpy(a))
        with torch.cuda.stream(self._side):
            if self._consumed is not None:
                self._side.wait_event(self._consumed)   # do not overwrite targets still being converted
            self._staged.copy_(self._pinned, non_blocking=True)
            self._ready = torch.cuda.Event()
            self._ready.record(self._side)

    def step(self, shift=1, x0=None, first=False):
        """Shift the previous solution by ``shift`` steps, install the prefetched targets and the
        new initial states ``x0 (B, n)`` if given, solve, and return the first inputs ``(B, m)``.
        ``first=True`` solves the instance uploaded at construction without shifting."""
        torch, d = self.torch, self.dev
        main = torch.cuda.current_stream(d.device)
        if self._ready is not None:
            main.wait_event(self._ready)
            _lib.check(d.lib.ilqr_to_tiled_v(self._staged.data_ptr(), d.Xref.data_ptr(), self.B, self.N * self.n,
                                             d.np_dtype.itemsize, d.target_layout(), d.stream()), "ilqr_to_tiled_v")
            self._consumed = torch.cuda.Event()
            self._consumed.record(main)
            self._ready = None
        if not first:
            x0p = None
            if x0 is not None:
                a = np.ascontiguousarray(np.asarray(x0, dtype=d.np_dtype).reshape(self.B, self.n))
                self._x0.copy_(torch.from_numpy(a).reshape(-1), non_blocking=False)
                x0p = C.c_void_p(self._x0.data_ptr())
            _lib.check(d.lib.ilqr_shift(d.handle, int(shift), x0p, d.stream()), "ilqr_shift")
        self.opt.solve_device()
        self.steps += 1
        return self.first_inputs()

    def first_inputs(self):
        """Inputs of time step 0 of every problem, ``(B, m)`` (a small strided read of the tiled buffer)."""
        d = self.dev
        tiles = (self.B + _lib.TILE - 1) // _lib.TILE
        u = d.U.view(tiles, (self.N - 1) * self.m, _lib.TILE)[:, : self.m, :]
        return u.permute(0, 2, 1).reshape(tiles * _lib.TILE, self.m)[: self.B].cpu().numpy()

    def solution(self):
        """Full result of the last solve, as the solver's attributes (``states``, ``inputs``, ``min_cost`` ...)."""
        self.opt.download()
        return self.opt
