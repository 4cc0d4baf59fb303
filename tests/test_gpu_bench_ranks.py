"""The N > 1 path of bench.py on a box with one GPU: two ranks launched as the driver launches them
(torch.distributed.run, one process per rank), both on device 0, meeting over gloo (--rehearse).
Checks what cannot be checked without the launcher: the environment parsing, the shard arithmetic
inside the bench, the barriers and reductions around the timed region, and that rank 0 prints ONE
JSON line whose per-problem statistics equal the single-process run of the same batch (shards are
bit-identical to the whole batch, so the iteration counts must add up exactly)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(ranks, port):
    common = ["bench.py", "--gpus", str(ranks), "--steps", "1", "--warmup", "0", "--batch", "4160", "--horizon", "60", "--no-cpu-baseline"]
    if ranks == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + common + ["--rehearse"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_report_the_whole_job():
    one = _bench(1, 0)
    two = _bench(2, 29533)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["batch_total"] == 4160 and two["config"]["batch_per_gpu"] == 2080
    assert two["value"] > 0 and two["ms_per_step"] > 0
    assert two["executed_iterations_per_step"] == one["executed_iterations_per_step"]
    assert two["converged_fraction"] == one["converged_fraction"]
    assert "cpu_baseline" not in two and "roofline" in two
