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


LAUNCH = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]


def test_two_ranks_strong_scaling_split_the_table(tmp_path):
    """Default mode: --rows is the TOTAL, rank k owns [k N/2, (k+1) N/2); the merged state equals one rank over all rows."""
    rows = 40_000_000
    two = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", str(rows), "--no-cpu-baseline"],
                 env={"EXON_BENCH_SHARE_GPU": "1"}, launcher=LAUNCH + ["--master-port", "29533"])
    one = _bench(["--steps", "3", "--warmup", "1", "--rows", str(rows), "--no-cpu-baseline", "--no-extras"])
    for d, n in ((two, 2), (one, 1)):
        assert d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong" and d["higher_is_better"] is True
        assert d["unit"] == "Mrows/s" and d["value"] > 0 and d["roofline"]["bound"] == "hbm"
        assert d["config"]["rows_total"] == rows and d["config"]["rows_per_gpu"] == rows // n
    assert "all_gather" in two["config"]["reduce"]
    assert two["result"]["filter_rows"] == one["result"]["filter_rows"]
    for a, b in zip(two["result"]["avg_qual"], one["result"]["avg_qual"]):
        assert a == pytest.approx(b, rel=1e-12)


def test_two_ranks_weak_scaling_grow_the_table():
    rows = 20_000_000
    two = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", str(rows), "--scaling", "weak", "--no-cpu-baseline"],
                 env={"EXON_BENCH_SHARE_GPU": "1"}, launcher=LAUNCH + ["--master-port", "29534"])
    one = _bench(["--steps", "3", "--warmup", "1", "--rows", str(2 * rows), "--no-cpu-baseline", "--no-extras"])
    assert two["scaling"] == "weak" and two["config"]["rows_total"] == 2 * rows and two["config"]["rows_per_gpu"] == rows
    # rank 0 holds rows [0, rows), rank 1 rows [rows, 2 rows): the merged state equals one rank over [0, 2 rows)
    assert two["result"]["filter_rows"] == one["result"]["filter_rows"]
    for a, b in zip(two["result"]["avg_qual"], one["result"]["avg_qual"]):
        assert a == pytest.approx(b, rel=1e-12)


LAUNCH8 = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1"]


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_eight_ranks_under_the_drivers_launch_line(scaling):
    """The SCALE run's 8-rank line rehearsed on one GPU (EXON_BENCH_SHARE_GPU=1: eight ranks on cuda:0, gloo carries the merge):
    rank k owns rows [k N/8, (k+1) N/8); the merged answer equals one rank over the whole table, strong and weak."""
    rows = 64_000_000 if scaling == "strong" else 8_000_000
    total = rows if scaling == "strong" else 8 * rows
    eight = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--rows", str(rows), "--scaling", scaling, "--no-cpu-baseline"],
                   env={"EXON_BENCH_SHARE_GPU": "1"}, launcher=LAUNCH8 + ["--master-port", "29537" if scaling == "strong" else "29538"])
    one = _bench(["--steps", "2", "--warmup", "1", "--rows", str(total), "--no-cpu-baseline", "--no-extras"])
    assert eight["n_gpus"] == 8 and eight["scaling"] == scaling and eight["config"]["rows_total"] == total
    assert eight["config"]["rows_per_gpu"] == total // 8 and "all_gather" in eight["config"]["reduce"]
    assert eight["result"]["filter_rows"] == one["result"]["filter_rows"]
    for a, b in zip(eight["result"]["avg_qual"], one["result"]["avg_qual"]):
        assert a == pytest.approx(b, rel=1e-12)


def test_eight_ranks_histogram_workload():
    """config 5 at 8 ranks: 204.8 KB of state per rank, 8 x that through the gather + fold merge."""
    reads = 4_000_000
    eight = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--rows", str(reads), "--workload", "c5", "--no-cpu-baseline"],
                   env={"EXON_BENCH_SHARE_GPU": "1"}, launcher=LAUNCH8 + ["--master-port", "29539"])
    one = _bench(["--steps", "2", "--warmup", "1", "--rows", str(reads), "--workload", "c5", "--no-cpu-baseline"])
    assert eight["n_gpus"] == 8 and eight["result"]["counts"] == one["result"]["counts"] and sum(one["result"]["counts"]) == reads * 100


def test_histogram_workload_merges_a_large_state_across_ranks():
    """config 5 under the launcher: 204.8 KB of state per rank through the gather + fold merge."""
    reads = 2_000_000
    two = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", str(reads), "--workload", "c5", "--no-cpu-baseline"],
                 env={"EXON_BENCH_SHARE_GPU": "1"}, launcher=LAUNCH + ["--master-port", "29535"])
    one = _bench(["--steps", "2", "--warmup", "1", "--rows", str(reads), "--workload", "c5", "--no-cpu-baseline"])
    assert two["result"]["counts"] == one["result"]["counts"] and sum(one["result"]["counts"]) == reads * 100


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must come back with n_gpus == 2 (it starts the
    ranks itself) -- never with a 1-GPU line."""
    rows = 20_000_000
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", str(rows),
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env={**env, "EXON_BENCH_SHARE_GPU": "1"}, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    two = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["config"]["rows_per_gpu"] == rows // 2
    one = _bench(["--steps", "2", "--warmup", "1", "--rows", str(rows), "--no-cpu-baseline", "--no-extras"])
    assert two["result"]["filter_rows"] == one["result"]["filter_rows"]


def test_gpus_flag_beyond_the_node_fails_loudly():
    """Without the share-GPU test hook a one-GPU box must refuse --gpus 2 (exit code != 0, no JSON line)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EXON_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus_flag():
    """--gpus 1 under a 2-rank launcher is a mistake, not a 1-GPU run."""
    r = subprocess.run([sys.executable] + LAUNCH + ["--master-port", "29536", os.path.join(ROOT, "bench.py"), "--gpus", "1",
                                                    "--steps", "1", "--warmup", "0", "--rows", "1000000", "--no-cpu-baseline"],
                       capture_output=True, text=True, env={**os.environ, "EXON_BENCH_SHARE_GPU": "1"}, cwd=ROOT, timeout=300)
    assert r.returncode != 0
