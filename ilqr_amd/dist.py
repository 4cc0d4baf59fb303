"""Multi-GPU sharding of a batch of independent iLQR problems.

Problems never interact, so the batch is split by problem index into one
contiguous range per rank (one process per GPU, ``torch.distributed``; backend
``nccl`` is RCCL on ROCm, ``gloo`` on CPU-only hosts) and every rank solves its
range with its own device buffers.  There is no collective on the solve path.
The only communication is the optional epilogue ``gather_summary``: an
all-gather of the per-problem final cost, iteration count and converged flag
(8 + 4 + 1 bytes per problem), which is latency-bound over xGMI.
"""
import os

import numpy as np


def shard_range(B, rank, world):
    """Contiguous problem range [first, first+count) of ``rank``; ranges differ in
    size by at most one and cover 0..B-1 exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(B), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment if WORLD_SIZE > 1.
    Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_summary(min_cost, iterations, converge, B_total, device=None):
    """All-gather the per-problem summary of every rank's shard.  Inputs are this
    rank's arrays (numpy or torch); returns numpy arrays of length ``B_total`` in
    global problem order on every rank.  Shards may have unequal sizes."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return (np.asarray(_to_np(min_cost), dtype=np.float64), np.asarray(_to_np(iterations), dtype=np.int32),
                np.asarray(_to_np(converge), dtype=bool))
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    maxc = max(shard_range(B_total, r, world)[1] for r in range(world))
    first, count = shard_range(B_total, rank, world)
    # one padded fp64 payload per rank: [cost | iterations | converged]
    payload = torch.zeros(3, maxc, dtype=torch.float64, device=device)
    payload[0, :count] = torch.as_tensor(_to_np(min_cost), dtype=torch.float64)
    payload[1, :count] = torch.as_tensor(_to_np(iterations).astype(np.float64))
    payload[2, :count] = torch.as_tensor(_to_np(converge).astype(np.float64))
    out = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(out, payload)
    cost = np.empty(B_total)
    iters = np.empty(B_total, dtype=np.int32)
    conv = np.empty(B_total, dtype=bool)
    for r in range(world):
        f, c = shard_range(B_total, r, world)
        blk = out[r].cpu().numpy()
        cost[f:f + c], iters[f:f + c], conv[f:f + c] = blk[0, :c], blk[1, :c].astype(np.int32), blk[2, :c] > 0.5
    return cost, iters, conv


def _to_np(a):
    try:
        import torch
        if isinstance(a, torch.Tensor):
            return a.detach().cpu().numpy()
    except ImportError:
        pass
    return np.asarray(a)


def solve_sharded(B_total, make_problems, solve_shard, gather=True):
    """Generic driver: ``make_problems(first, count)`` builds this rank's problems,
    ``solve_shard(problems)`` returns (min_cost, iterations, converge) for them.
    Returns the gathered global summary (or the local one if ``gather`` is false)."""
    rank, world, _ = env_rank_world()
    first, count = shard_range(B_total, rank, world)
    res = solve_shard(make_problems(first, count))
    if gather:
        return gather_summary(res[0], res[1], res[2], B_total)
    return res
