"""bench.py under the driver's multi-GPU launch line (python -m torch.distributed.run ... bench.py --gpus N), exercised on a
one-GPU box: EXON_BENCH_SHARE_GPU=1 lets both ranks use cuda:0 and reduce over gloo (RCCL refuses two ranks on one
device).  Checks the contract fields and that the two shards add up to the single-rank result over the same rows."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, launcher=None):
    cmd = [sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env={**os.environ, **(env or {})}, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_on_one_gpu_add_up_to_the_single_rank_answer():
    rows = 20_000_000
    two = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", str(rows), "--no-cpu-baseline"],
                 env={"EXON_BENCH_SHARE_GPU": "1"},
                 launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29533"])
    one = _bench(["--steps", "3", "--warmup", "1", "--rows", str(2 * rows), "--no-cpu-baseline"])
    for d, n in ((two, 2), (one, 1)):
        assert d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
        assert d["unit"] == "Mrows/s" and d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["config"]["rows_total"] == 2 * rows
    # rank 0 holds rows [0, rows), rank 1 rows [rows, 2 rows): the reduced state equals one rank over [0, 2 rows)
    assert two["result"]["filter_rows"] == one["result"]["filter_rows"]
    for a, b in zip(two["result"]["avg_qual"], one["result"]["avg_qual"]):
        assert a == pytest.approx(b, rel=1e-12)
