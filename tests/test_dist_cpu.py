"""world_size-2 gloo test of the multi-GPU path on CPU: the batch is sharded by problem
index, every rank solves its own range with no communication, and the optional summary
gather reassembles the results in global order.  The per-shard solve is injected; here it
is the CPU oracle (the product passes the GPU solver)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from ilqr_amd import dist as idist


def test_shard_ranges_cover_batch():
    for B in (1, 7, 64, 65536, 1000):
        for world in (1, 2, 3, 8):
            spans = [idist.shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == B
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert idist.shard_range(65536, 3, 8) == (24576, 8192)


def _worker(rank, world, port, B, N, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from conftest import demo_car_system
    from ilqr_amd import dist as idist2, problems
    from oracle import oracle
    idist2.init_process_group(backend="gloo")
    car = demo_car_system()
    d = oracle.make_desc(car, N, max_iter=6)

    def solve_shard(p):
        st, ui = p["init_states"].copy(), p["init_inputs"].copy()
        r = oracle.solve_batch(d, p["target_states"], st, ui, history=False)
        return r["min_cost"], r["iters"], r["converge"]

    cost, iters, conv = idist2.solve_sharded(B, lambda f, c: problems.car_batch(c, N, first=f), solve_shard)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cost=cost, iters=iters, conv=conv)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_solve_matches_single_rank(tmp_path, oracle):
    from conftest import demo_car_system
    from ilqr_amd import problems
    B, N, world = 9, 40, 2      # uneven split: 5 + 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, B, N, str(tmp_path)), nprocs=world, join=True)
    p = problems.car_batch(B, N)
    d = oracle.make_desc(demo_car_system(), N, max_iter=6)
    ref = oracle.solve_batch(d, p["target_states"], p["init_states"].copy(), p["init_inputs"].copy(), history=False)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        # problems are independent, so sharding must not change a single bit
        assert np.array_equal(got["cost"], ref["min_cost"])
        assert np.array_equal(got["iters"], ref["iters"])
        assert np.array_equal(got["conv"], ref["converge"])


def test_bench_shards_are_the_strong_scaling_configuration():
    """bench.py --gpus N splits the 65536 problems of the headline configuration by problem index:
    32768 / 16384 / 8192 per GPU at N = 2 / 4 / 8, contiguous and complete; --weak keeps --batch per GPU."""
    import bench
    for world, per in ((1, 65536), (2, 32768), (4, 16384), (8, 8192)):
        shards = [bench.bench_shard(65536, r, world, weak=False) for r in range(world)]
        assert all(s[1] == per and s[2] == 65536 for s in shards)
        assert [s[0] for s in shards] == [r * per for r in range(world)]
    assert bench.bench_shard(4096, 3, 8, weak=True) == (3 * 4096, 4096, 8 * 4096)
    # ragged totals stay complete and contiguous
    shards = [bench.bench_shard(1000, r, 3, weak=False) for r in range(3)]
    assert sum(s[1] for s in shards) == 1000 and shards[0][0] == 0
    assert all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(2))
