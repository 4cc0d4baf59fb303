"""The automatic settings at the per-GPU shard sizes of the headline batch (65536 Car problems,
N = 200, split over 16 / 8 / 4 / 2 GPUs, and the 12288 in between where the single-launch solve takes
over): what `ilqr_create` picks for each size with default keywords (forward mapping, sine/cosine
cache, time-parallel commit, single-launch solve, sparse states, active-set compaction) must give,
bit for bit, the results of ONE run of the largest size with every such option switched off
(problems do not depend on the batch they are solved in), and sampled problems must follow the CPU
oracle per iteration (ilqr.py:107-114)."""
import numpy as np
import pytest

from conftest import demo_car_system

pytestmark = pytest.mark.gpu

N = 200
SIZES = (4096, 8192, 12288, 16384, 32768)
# (lanes per problem, step sizes per lane) ilqr_create documents for 15 step sizes
MAPPING = {4096: (16, 1), 8192: (8, 2), 12288: (4, 4), 16384: (4, 4), 32768: (2, 3)}
# commits keep every 8th state row: inside ilqr_solve from 32768 problems on
SPARSE = {4096: 0, 8192: 0, 12288: 0, 16384: 0, 32768: 1}
KEYS = ("states", "inputs", "min_cost", "iterations", "converge", "status")


@pytest.fixture(scope="module")
def plain_run():
    """The largest size with every optimisation off: the statement the automatic paths are held to."""
    from ilqr_amd import problems
    from ilqr_amd.ilqr import iterative_LQR
    p = problems.car_batch(max(SIZES), N)
    car = demo_car_system()
    opt = iterative_LQR(car, p["target_states"], car.dt, history=True, recompute_states=-1, trig_cache=-1,
                        segment_commit=-1, fused_solve=-1, repack_every=-1)
    opt.inputs, opt.states = p["init_inputs"].copy(), p["init_states"].copy()
    dev = opt.upload()
    assert dev.sparse_states() == 0
    opt.solve_device()
    opt.download()
    assert np.all(opt.status == 0) and opt.converge.any() and (~opt.converge).any()
    return p, opt


@pytest.mark.parametrize("B", SIZES)
def test_automatic_paths_equal_the_plain_solve(B, plain_run, oracle):
    from ilqr_amd.ilqr import iterative_LQR
    p, ref = plain_run
    car = demo_car_system()
    opt = iterative_LQR(car, p["target_states"][:B], car.dt, history=True)
    opt.inputs, opt.states = p["init_inputs"][:B].copy(), p["init_states"][:B].copy()
    dev = opt.upload()
    assert dev.forward_mapping() == MAPPING[B]
    assert dev.sparse_states() == SPARSE[B]
    opt.solve_device()
    opt.download()
    for key in KEYS:
        assert np.array_equal(getattr(opt, key), getattr(ref, key)[:B]), (B, key)
    assert np.array_equal(opt.alpha_history, ref.alpha_history[:B])
    assert np.array_equal(opt.cost_history, ref.cost_history[:B], equal_nan=True)
    # 8 problems spread over the shard against the oracle, every iteration both executed
    d = oracle.make_desc(car, N)
    for b in np.linspace(0, B - 1, 8).astype(int):
        r = oracle.solve(d, np.ascontiguousarray(p["target_states"][b]), p["init_states"][b].copy(),
                         p["init_inputs"][b].copy(), history=False)
        n = min(int(opt.iterations[b]), r["iters"])
        assert n >= r["iters"] - 3
        rel = np.abs(opt.cost_history[b, :n] - r["hist_cost"][:n]) / np.abs(r["hist_cost"][:n])
        assert rel.max() <= 1e-9, (B, b, rel.max())
