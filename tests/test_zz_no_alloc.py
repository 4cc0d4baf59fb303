"""ilqr_solve_static: the variant of ilqr_solve that keeps the kernels' scratch in a static device
array of the library and allocates nothing.  Two calls in a row through the C ABI, the second with a
smaller input, each against ilqr_solve on the same inputs: every output must be equal bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import demo_car_system

pytestmark = pytest.mark.gpu


def _run(B, N, static, work_null=False, want_sparse=None, **kw):
    import torch
    from ilqr_amd import problems
    from ilqr_amd.ilqr import iterative_LQR
    car = demo_car_system()
    p = problems.car_batch(B, N, first=3)
    opt = iterative_LQR(car, p["target_states"], car.dt, history=True, repack_every=-1, **kw)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    if want_sparse is not None:
        assert dev.sparse_states() == want_sparse
    if work_null:       # a caller without a workspace of its own
        dev.buffers.work = None
        from ilqr_amd import _lib
        _lib.check(dev.lib.ilqr_bind(dev.handle, C.byref(dev.buffers)), "ilqr_bind")
    if static:
        # the call allocates nothing: the device's free memory is the same before and after
        # (the caching allocator of the carrier tensors is idle in between)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        dev.solve_static(opt.maxIter)
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        opt.free_delta = free0 - free1
    else:
        dev.solve(opt.maxIter)
    opt.download()
    return opt, opt.K.copy(), opt.k.copy()


def test_static_scratch_solve_equals_ilqr_solve_twice():
    import torch
    assert torch.cuda.is_available()
    # (the second call is the smaller one; the third keeps sparse states, whose bookkeeping lives in the
    # same scratch: asked for by value, they win over the sine/cosine cache and the time-parallel commit
    # a batch of this size would get by default, and the handle says so)
    cases = ((700, 90, dict(), None), (130, 50, dict(lanes_per_problem=1, alphas_per_lane=6), None),
             (333, 70, dict(lanes_per_problem=2, alphas_per_lane=3, recompute_states=2), 2))
    for i, (B, N, kw, sparse) in enumerate(cases):
        ref, Kr, kr = _run(B, N, static=False, want_sparse=sparse, **kw)
        got, Kg, kg = _run(B, N, static=True, work_null=True, want_sparse=sparse, **kw)
        assert (ref.iterations > 2).any() and ref.converge.any()
        for key in ("states", "inputs", "min_cost", "iterations", "converge", "status"):
            assert np.array_equal(getattr(got, key), getattr(ref, key)), (B, key)
        assert np.array_equal(got.alpha_history, ref.alpha_history)
        assert np.array_equal(got.cost_history, ref.cost_history, equal_nan=True)
        assert np.array_equal(Kg, Kr) and np.array_equal(kg, kr)
        if i > 0:   # (the very first call on a device looks the array's address up, which may load the code object)
            assert got.free_delta == 0, got.free_delta


def test_sparse_states_between_phase_calls_on_request():
    """recompute_states = 2 at a batch size that would get the sine/cosine cache by default: the
    phase calls leave X sparse (the calls that start from X refuse until ilqr_states has run)."""
    from ilqr_amd import problems
    from ilqr_amd.ilqr import iterative_LQR
    car = demo_car_system()
    p = problems.car_batch(333, 70, first=3)
    opt = iterative_LQR(car, p["target_states"], car.dt, repack_every=-1, recompute_states=2)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    assert dev.sparse_states() == 2
    dev.initial_cost()
    for it in range(3):
        dev.backward()
        dev.forward(it)
    assert dev.lib.ilqr_initial_cost(dev.handle, dev.stream()) == -6
    dev.fill_states()
    assert dev.lib.ilqr_initial_cost(dev.handle, dev.stream()) == 0


def test_static_scratch_rejects_what_it_cannot_hold():
    from ilqr_amd import _lib, problems
    from ilqr_amd.ilqr import iterative_LQR
    car = demo_car_system()
    B, N = _lib.STATIC_MAX_PROBLEMS + 64, 2
    tg = np.zeros((B, N, 4))
    opt = iterative_LQR(car, tg, car.dt, repack_every=-1)
    dev = opt.upload()
    assert dev.lib.ilqr_solve_static(dev.handle, 1, dev.stream()) == -5
    assert b"static scratch" in dev.lib.ilqr_strerror(-5)
    # more than one stream would have to be created: refused, nothing launched
    p = problems.car_batch(200, 20)
    o2 = iterative_LQR(car, p["target_states"], car.dt, n_streams=2)
    d2 = o2.upload()
    assert d2.lib.ilqr_solve_static(d2.handle, 1, d2.stream()) == -2
    # a handle without a workspace refuses the calls that need one
    d2.buffers.work = None
    _lib.check(d2.lib.ilqr_bind(d2.handle, C.byref(d2.buffers)), "ilqr_bind")
    assert d2.lib.ilqr_solve(d2.handle, 1, d2.stream()) == -3
    assert d2.lib.ilqr_backward(d2.handle, d2.stream()) == -3
