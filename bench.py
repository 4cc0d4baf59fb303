#!/usr/bin/env python3
"""Benchmark of the batched iLQR hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one complete batched solve (``iterative_LQR.__call__``: initial cost, then up to
30 backward/forward iterations per problem until its ``converge`` flag) of B Car problems with
horizon N=200 in fp64, the configuration the headline metric is quoted on.  Inputs are
resident in HBM when the timed region starts: every step has its own copy of the initial guess (a
solve overwrites it with the solution) and binds it with a host-only call.  With N > 1 GPUs the batch of
65536 problems is split by problem index into contiguous shards, one per rank (strong scaling,
the configuration the metric names: 32768 / 16384 / 8192 problems per GPU at N = 2 / 4 / 8; no
communication on the solve path); ``--weak`` gives every rank its own ``--batch`` problems
instead.  The reported value is the whole-job rate.

Prints ONE JSON line on rank 0 with the metric, a ``roofline`` object for the dominant kernel
(the one with the largest total time; per-launch HIP-event timings taken inside the timed region, on
its first step, achieved = the bytes the kernel must move for the problems still iterating / its duration) and, at N=1, a ``cpu_baseline``
object: the C oracle (a port of the reference's per-problem loop) timed on the host cores on
a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def algorithmic_bytes(N, n, m, w=8, sparse=False):
    """Bytes each kernel must move per problem it works on (w bytes per value), the least the
    phase decomposition allows: backward reads X, U, Xref and writes K, k; the line search reads
    K, k, U, Xref and x0 once for all step sizes (the previous states follow from x0 and U);
    the commit reads K, k, U, x0 and writes the accepted X, U.  With sparse states only every 8th
    state row passes between the commit and the backward pass."""
    T = N - 1
    gains = T * m * n + T * m
    xrows = (T + 7) // 8 if sparse else N      # state rows written by the commit, read by the backward pass
    return {
        "backward": w * (xrows * n + N * n + T * m + gains),
        "evaluate_first": w * (gains + T * m + N * n + n),
        "evaluate_rest": w * (gains + T * m + N * n + n),
        "commit": w * (gains + T * m + n + xrows * n + T * m),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="problems of the whole job (per GPU with --weak)")
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--weak", action="store_true", help="every GPU solves --batch problems of its own instead of a shard")
    ap.add_argument("--strong", action="store_true", help="(default) split --batch over the GPUs by problem index")
    ap.add_argument("--mapping", default="auto", help="forward mapping L,A (lanes per problem, step sizes per lane)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--repack-every", type=int, default=None,
                    help="active-set compaction period; default = what ilqr_solve picks: 8 from 32768 problems per GPU on, else none")
    ap.add_argument("--no-repack", action="store_true")
    ap.add_argument("--segment-commit", type=int, default=0, help="time-parallel commit: 0 = the library's choice, -1 off, 1 on")
    ap.add_argument("--trig-cache", type=int, default=0, help="sine/cosine cache: 0 = the library's choice, -1 off, 1 on")
    ap.add_argument("--sparse-states", type=int, default=0,
                    help="commits keep every 8th state row only: 0 = what ilqr_solve picks (on from 32768 problems per GPU), -1 off, 1 on")
    ap.add_argument("--event-steps", type=int, default=1,
                    help="timed steps whose launches are bracketed by HIP events for the per-kernel table (0 = all)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a box with one GPU: every rank uses device 0 and the ranks meet over gloo (checks the "
                         "sharding, the barriers and the reductions; the time it prints means nothing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="upper bound on problems in the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ilqr_amd import dist as idist, problems, systems
    from ilqr_amd.ilqr import iterative_LQR

    rank, world, local = idist.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    if args.rehearse:
        local = 0
    torch.cuda.set_device(local)
    idist.init_process_group(("gloo" if args.rehearse else "nccl") if world > 1 else None)

    N = args.horizon
    first, B, B_total = bench_shard(args.batch, rank, world, args.weak)
    car = systems.Car()
    car.set_dt(0.2)
    car.set_cost(problems.CAR_Q.copy(), problems.CAR_R.copy())
    car.set_control_limit(*problems.CAR_LIMITS)
    p = problems.car_batch(B, N, first=first)

    kw = {}
    if args.mapping != "auto":
        L, A = (int(v) for v in args.mapping.split(","))
        kw = dict(lanes_per_problem=L, alphas_per_lane=A)
    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    repack_every = 0 if args.no_repack else (args.repack_every if args.repack_every is not None else (8 if B >= 32768 else 0))
    # the handle is created with the same period, so that it holds the workspace ilqr_repack needs
    # sparse states between the phase calls as inside ilqr_solve (recompute_states: 1 = never, 2 = always)
    sparse = args.sparse_states > 0 or (args.sparse_states == 0 and B >= 32768)
    opt = iterative_LQR(car, p["target_states"], car.dt, history=True, dtype=np_dtype,
                        repack_every=repack_every if repack_every > 0 else -1, segment_commit=args.segment_commit,
                        trig_cache=args.trig_cache, recompute_states=2 if sparse else 1, **kw)
    opt.inputs, opt.states = p["init_inputs"], p["init_states"]
    dev = opt.upload()
    X0, U0 = dev.X.clone(), dev.U.clone()
    max_iter = opt.maxIter
    # Every solve overwrites its initial guess (X, U) with the solution.  The guesses of all steps are
    # therefore resident before the timed region starts, one copy of X and U per step (0.63 GB each at
    # the headline size), and a step binds its own; beyond MAX_SETS steps the copies are restored by
    # device-to-device copies inside the timed region instead.
    MAX_SETS = 48
    n_sets = args.steps + args.warmup + 1
    sets = [(X0.clone(), U0.clone()) for _ in range(n_sets)] if n_sets <= MAX_SETS else None
    next_set = [0]

    def reset():
        if sets is not None:
            dev.X, dev.U = sets[next_set[0] % n_sets]
            next_set[0] += 1
            dev._bind()
        else:
            dev.X.copy_(X0)
            dev.U.copy_(U0)
        dev.converged.zero_()
        dev.iters.zero_()
        dev.hist_alpha.fill_(-2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def solve_step(events=None):
        """The launches of ilqr_solve, issued one by one so that events can bracket them."""
        reset()
        dev.initial_cost()
        for it in range(max_iter):
            if repack_every > 0 and it >= 2 * repack_every and it % repack_every == 0:
                dev.repack()
            if events is not None:
                e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                e[0].record()
                dev.backward()
                e[1].record()
                dev.forward_evaluate_stage(0)
                e[2].record()
                dev.forward_evaluate_stage(1)
                e[3].record()
                dev.forward_commit(it)
                e[4].record()
                events.append(e)
            else:
                dev.backward()
                dev.forward(it)
        if repack_every > 0:
            dev.unpack()
        dev.fill_states()     # every state row again (a no-op without sparse states, or after the unpack)

    for _ in range(args.warmup):
        solve_step()
    barrier()
    # HIP events bracket every launch of the first `event_steps` timed steps (an event record costs about
    # 5.6 us on the stream, 0.8 ms per step of 150 bracketed launches; the other steps issue the
    # same launches back to back, as ilqr_solve does)
    events = []
    event_steps = args.steps if args.event_steps <= 0 else min(args.steps, args.event_steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        solve_step(events if i < event_steps else None)
    barrier()
    dt_local = time.perf_counter() - t0
    tmax = torch.tensor([dt_local], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # cross-check, outside the timed region: the same solve issued by the single C call
    reset()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dev.solve(max_iter)
    torch.cuda.synchronize()
    ilqr_solve_ms = 1e3 * (time.perf_counter() - t1)

    # per-problem outcome of the last step (identical every step: same inputs)
    iters = dev.iters.cpu().numpy()
    conv = dev.converged.cpu().numpy()
    ha = dev.hist_alpha.cpu().numpy()                     # [max_iter][B]
    active = (ha != -2).sum(axis=1).astype(np.int64)       # problems that executed iteration it
    accepted = (ha >= 0).sum(axis=1).astype(np.int64)
    iter_sum = torch.tensor([float(iters.sum()), float(conv.sum())], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(iter_sum)

    # per-kernel timings from the events of the timed region (this rank)
    names = ("backward", "evaluate_first", "evaluate_rest", "commit")
    kt = {k: [] for k in names}
    for e in events:
        for i, k in enumerate(names):
            kt[k].append(e[i].elapsed_time(e[i + 1]))
    ab = algorithmic_bytes(N, 4, 2, 8 if args.dtype == "f64" else 4, sparse)
    ab0 = algorithmic_bytes(N, 4, 2, 8 if args.dtype == "f64" else 4, False)
    # bytes per problem in iteration `it`: the first backward pass and the first commit move every state row
    ab_it = {k: np.array([ab0[k] if it == 0 else ab[k] for it in range(max_iter)], dtype=np.float64) for k in ab}
    L_, A_ = dev.forward_mapping()
    n_alphas = dev.num_alphas()
    # problems the second line-search stage works on: no improvement among the first L*A step sizes
    listed = (((ha == -1) | (ha >= L_ * A_)).sum(axis=1).astype(np.int64) if L_ * A_ < n_alphas
              else np.zeros(max_iter, dtype=np.int64))
    units = {"backward": active, "evaluate_first": active, "evaluate_rest": listed, "commit": accepted}
    kernels = {}
    for name, ms in kt.items():
        ms = np.array(ms).reshape(event_steps, max_iter)
        live = units[name] > 0
        total_ms = ms[:, live].sum()
        total_bytes = float(event_steps) * float((units[name][live] * ab_it[name][live]).sum())
        kernels[name] = {
            "launches": int(live.sum()) * event_steps,
            "avg_ms": float(total_ms / max(1, live.sum() * event_steps)),
            "total_ms": float(total_ms),
            "achieved_GBs": float(total_bytes / (total_ms * 1e-3) / 1e9) if total_ms > 0 else 0.0,
            "bytes_per_problem": ab[name],
            "problems_per_launch": float(units[name][live].mean()) if live.any() else 0.0,
        }
    dominant = max(kernels, key=lambda k: kernels[k]["total_ms"])
    # counter profiles of this very command, committed under profiles/ (tools/round_profile.sh: separate
    # rocprofv3 --pmc passes); quoted only while they were taken from the kernel source in the tree
    traffic_all = stamped_profile("hbm_traffic.json")
    limits_all = stamped_profile("kernel_limits.json")
    traffic = (traffic_all or {}).get(dominant, {}).get("bytes_per_launch")
    limit = (limits_all or {}).get(dominant)

    out = None
    if rank == 0:
        solves = float(B_total) * args.steps
        out = {
            "metric": f"batched iLQR solves/sec (B={B_total}, N={N})",
            "value": solves / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (seeded Car tracking problems, ilqr_amd.problems.car_batch)",
            "config": {"workload": f"Car n=4 m=2, N={N}, B={B_total} problems"
                                   + (f" sharded by problem index over {world} GPUs ({B} per GPU)" if world > 1 and not args.weak
                                      else f" on each of {world} GPUs" if world > 1 else "")
                                   + ", maxIter=30, 15 parallel step sizes",
                       "batch_per_gpu": B, "batch_total": B_total, "horizon": N,
                       "forward_mapping": f"L={L_},A={A_} ({'two-stage' if L_ * A_ < n_alphas else 'one stage'})",
                       "active_set_compaction_every": repack_every,
                       "sparse_states": bool(sparse)},
            "mean_iterations_per_solve": float(iter_sum[0].item()) / B_total,
            "converged_fraction": float(iter_sum[1].item()) / B_total,
            "executed_iterations_per_step": float(iter_sum[0].item()),
            "roofline": {
                "kernel": dominant, "bound": "hbm",
                "achieved": kernels[dominant]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dominant]["achieved_GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": kernels[dominant]["avg_ms"],
                "bytes_per_problem_per_launch": ab[dominant],
                # what the counters say the kernel is bound by (share of its wave cycles that issue a vector
                # instruction; tools/pmc_limits.py), null without a profile of the current kernel source
                "limiter": limiter_of(limit, traffic, kernels[dominant]["avg_ms"]),
                "valu_busy": limit["valu_busy"] if limit else None,
                # measured HBM bytes per launch / this run's average launch duration: the rate the memory
                # system actually ran at (converged problems inside live tiles and whole-line fetches of
                # scattered rows make it higher than `achieved`)
                "traffic_GBs": (traffic / (kernels[dominant]["avg_ms"] * 1e-3) / 1e9) if traffic else None,
                "traffic_GBs_by_kernel": ({k: traffic_all[k]["bytes_per_launch"] / (v["avg_ms"] * 1e-3) / 1e9
                                           for k, v in kernels.items() if k in traffic_all and v["avg_ms"] > 0} if traffic_all else None),
                "hbm_frac_by_kernel": {k: v["achieved_GBs"] / HBM_PEAK_GBS for k, v in kernels.items()},
                "limiter_by_kernel": ({k: limits_all[k]["limiter"] for k in kernels if k in limits_all} if limits_all else None),
            },
            "kernels": kernels,
            "solve_achieved_GBs": float(args.steps * (active * (ab_it["backward"] + ab_it["evaluate_first"]) + listed * ab_it["evaluate_rest"]
                                                      + accepted * ab_it["commit"]).sum()
                                        / elapsed / 1e9) * world,   # rank 0's shard x ranks (shards are equal up to one tile)
            "event_steps": event_steps,       # timed steps whose launches carry the HIP events the `kernels` table is made of
            "ilqr_solve_ms": ilqr_solve_ms,     # one ilqr_solve() call on the same inputs (host clock, rank 0, untimed region)
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(car, p, N, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def limiter_of(limit, traffic, avg_ms):
    """What the dominant kernel is bound by: the counters' verdict on vector issue (tools/pmc_limits.py) and,
    beside it, the memory system when the measured traffic moves at 55 % of the HBM peak or more (the
    read/write mixes of these kernels top out at 4.9 to 6.1 TB/s on this chip)."""
    if not limit:
        return None
    rate = traffic / (avg_ms * 1e-3) / 1e9 if traffic and avg_ms > 0 else 0.0
    if limit["limiter"] != "HBM bandwidth" and rate >= 0.55 * HBM_PEAK_GBS:
        return limit["limiter"] + " and HBM bandwidth"
    return limit["limiter"]


def bench_shard(batch, rank, world, weak):
    """(first problem, problems of this rank, problems of the whole job)."""
    from ilqr_amd import dist as idist
    if weak:
        return rank * batch, batch, batch * world
    first, count = idist.shard_range(batch, rank, world)
    return first, count, batch


def kernel_source_stamp():
    """git's blob hash of ilqr_amd/csrc/kernels.hpp (what tools/pmc_traffic.py stamps its output with)."""
    import hashlib
    data = open(os.path.join(ROOT, "ilqr_amd", "csrc", "kernels.hpp"), "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def stamped_profile(name):
    """profiles/<name> if it was taken from the kernel source in the tree, else None."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            prof = json.load(f)
    except (OSError, ValueError):
        return None
    if prof.get("_source", {}).get("kernels_hpp") != kernel_source_stamp():
        return None
    return prof


def reference_calibration():
    """The measured ratio between the C port and the numpy reference (tools/time_reference.py, build
    container), or None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r02", "cpu_reference.json")
    try:
        with open(path) as f:
            c = json.load(f)
        return {"file": "profiles/r02/cpu_reference.json", "where": c["where"], "cpu": c["cpu"],
                "reference_solves_per_s_per_core": c["reference"]["solves_per_s_per_core"],
                "c_port_solves_per_s_per_core": c["c_port"]["solves_per_s_per_core"],
                "port_over_reference_per_core": c["port_over_reference_per_core"]}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(car, p, N, sample):
    """The C oracle (a scalar per-problem port of ilqr.py) on the first `sample` problems of
    the same workload, OpenMP over problems on the host cores this process may use."""
    from oracle import oracle
    oracle.lib()
    # every core this process may use, up to the worker count the box assigns to one GPU's job
    # (OMP_NUM_THREADS there; the affinity mask shows the whole machine, which is shared)
    avail = len(os.sched_getaffinity(0))
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = min(avail, share) if share > 0 else avail
    d = oracle.make_desc(car, N)
    tg = np.ascontiguousarray(p["target_states"][:64])
    # scale the sample so the run takes roughly 30 core-seconds of CPU work in total
    st, ui = p["init_states"][:64].copy(), p["init_inputs"][:64].copy()
    t0 = time.perf_counter()
    oracle.solve_batch(d, tg[:64], st, ui, nthreads=cores, history=False)
    probe = time.perf_counter() - t0
    per_problem_core_s = probe * cores / 64.0
    sample = int(max(64, min(sample, p["target_states"].shape[0], 30.0 / max(per_problem_core_s, 1e-9))))
    tg = np.ascontiguousarray(p["target_states"][:sample])
    st, ui = p["init_states"][:sample].copy(), p["init_inputs"][:sample].copy()
    t0 = time.perf_counter()
    r = oracle.solve_batch(d, tg[:sample], st, ui, nthreads=cores, history=False)
    el = time.perf_counter() - t0
    out = {"value": sample / el, "unit": "solves/s", "cores": cores, "cores_available": avail, "kind": "port",
           "sample": f"first {sample} problems of the same batch, C oracle (scalar fp64 port of ilqr.py), "
                     f"OpenMP over problems, {el:.1f} s wall, mean {float(r['iters'].mean()):.1f} iterations"}
    cal = reference_calibration()
    if cal is not None:
        # the reference (numpy ilqr.py) cannot travel to the GPU box: its speed relative to this port
        # was measured where both exist
        out["reference_calibration"] = cal
        out["reference_equivalent_value"] = out["value"] / cal["port_over_reference_per_core"]
    return out


if __name__ == "__main__":
    main()
