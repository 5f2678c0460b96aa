"""The native collectives fail TOGETHER and never hang: exon_hip_stream_reconcile_keys and
exon_hip_stream_all_reduce with 2 and 8 ranks that share cuda:0.  RCCL refuses several ranks on one device, so the communicator is the
library's callback kind (exon_hip_comm_from_callbacks) over gloo: the vote / reconcile / merge logic under test is the same C++ the
RCCL kind runs, only the bytes travel through torch.distributed.  EXON_HIP_FAULT=site@rank makes one rank fail at a named site; every
rank must come back with the same status within seconds.  What AggregateExec(Final) over file groups does in the reference:
exon-core/src/datasources/exon_file_scan_config.rs:79-110."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_expect as OX
from test_key_reconcile import cat_vcf, k4_by_value, write_vcf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import datetime, json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import exon_amd
from exon_amd.distributed import CallbackComm
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("GLOO_TIMEOUT_S", "60"))))
rank, world = dist.get_rank(), dist.get_world_size()
ctx = exon_amd.Context(0)
paths = json.loads(sys.argv[2])
mode = sys.argv[3]
plan = ctx.plan_cmp_avg_by_group(">", 0.01, int(sys.argv[4]), columns=(4, 2, 3))
st = plan.open()
rows = 0
for p in paths[rank::world]:
    s = exon_amd.Scan(p, "vcf", info_field="AF", gpu_parse=True)
    rows += st.consume(s)
    s.close()
comm = CallbackComm(ctx)
assert comm.count() == (world, rank)
if mode == "finished_rank" and rank == 1:
    st.finish()                                  # this rank's stream is closed: it must still ENTER the collectives and say so
dist.barrier()
t0 = time.time()
out = {"rank": rank, "rows": rows}
try:
    if mode != "skip_reconcile":
        st.reconcile_keys(comm.h.value)
    st.all_reduce(comm.h.value)
    keys, agreed = st.keys()
    c, s_ = st.finish()
    out.update(ok=True, keys=keys, counts=c.tolist(), sums=s_.tolist())
except exon_amd.ExonHipError as e:
    out.update(ok=False, code=e.code, text=str(e))
out["seconds"] = time.time() - t0
with open(os.path.join(sys.argv[5], f"rank{rank}.json"), "w") as fh:  # (stdout of 8 ranks interleaves)
    json.dump(out, fh)
try:
    dist.barrier()
except Exception:
    pass
st.close(); plan.close(); comm.close(); ctx.close()
os._exit(0)  # (a rank stalled on purpose may have left gloo in a broken state: leave without its teardown)
'''


def run(tmp_path, paths, world, mode="ok", n_groups=64, fault=None, port=29700, env=None, timeout=300):
    script = tmp_path / "rank.py"
    script.write_text(_SCRIPT)
    e = dict(os.environ, EXON_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if fault:
        e["EXON_HIP_FAULT"] = fault
    e.update(env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(script), ROOT, json.dumps(paths), mode, str(n_groups), str(tmp_path)]
    for k in range(world):
        if os.path.exists(tmp_path / f"rank{k}.json"):
            os.remove(tmp_path / f"rank{k}.json")
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=timeout, env=e)
    have = [k for k in range(world) if os.path.exists(tmp_path / f"rank{k}.json")]
    assert len(have) == world, (r.stdout + r.stderr)[-3000:]
    ranks = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]
    return sorted(ranks, key=lambda x: x["rank"]), time.time() - t0


def files(tmp_path, world):
    pools = [["PASS", ".", "q10"], ["s50", "q10;s50", "q10", ".", "PASS"], ["q10;s50", "PASS"], ["lowGQ", "PASS"], ["."], ["s50"], ["PASS", "lowGQ;s50"], ["q10"]]
    paths = [str(tmp_path / f"f{i}.vcf") for i in range(world)]
    for i, p in enumerate(paths):
        write_vcf(p, 3000 + 500 * i, 10 + i, pools[i % len(pools)])
    return paths


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_native_reconcile_and_merge_over_n_ranks(tmp_path, oracle, world):
    """no fault: the native path end to end -- every rank ends with the same keys and the same state, equal to the oracle over the
    one concatenated table"""
    paths = files(tmp_path, world)
    one = str(tmp_path / "all.vcf")
    cat_vcf(one, paths)
    n, want = OX.k4_expected(oracle, one, "vcf", "AF")
    ranks, _ = run(tmp_path, paths, world, port=29700 + world)
    assert all(r["ok"] for r in ranks), ranks
    assert sum(r["rows"] for r in ranks) == n
    for r in ranks:
        assert r["keys"] == ranks[0]["keys"] and r["counts"] == ranks[0]["counts"] and r["sums"] == ranks[0]["sums"]
    assert k4_by_value(ranks[0]["keys"], np.array(ranks[0]["counts"]), np.array(ranks[0]["sums"]), 64) == want


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("fault,code", [("reconcile_enter@1", -5), ("reconcile_malloc@1", -2), ("reconcile_rekey@1", -2), ("allreduce_enter@1", -5),
                                        ("allreduce_malloc@1", -2)])
def test_a_fault_on_one_rank_fails_every_rank_together(tmp_path, world, fault, code):
    """one rank cannot go on (at each of the sites that used to be rank-local returns): EVERY rank returns that status, quickly, and the
    text names the rank"""
    paths = files(tmp_path, world)
    ranks, _ = run(tmp_path, paths, world, fault=fault, port=29720 + world)
    assert not any(r["ok"] for r in ranks), ranks
    assert {r["code"] for r in ranks} == {code}, ranks
    assert all(r["seconds"] < 5.0 for r in ranks), [r["seconds"] for r in ranks]
    assert all("rank 1" in r["text"] or "this rank, 1" in r["text"] for r in ranks), [r["text"] for r in ranks]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_real_refusals_fail_together_too(tmp_path, world):
    """(a) a rank whose stream has finished; (b) nobody reconciled: states keyed by rank-local dictionaries; (c) the union of the
    dictionaries exceeds n_groups -- every rank gets the same status, none hangs"""
    paths = files(tmp_path, world)
    ranks, _ = run(tmp_path, paths, world, mode="finished_rank", port=29740 + world)
    assert not any(r["ok"] for r in ranks) and {r["code"] for r in ranks} == {-5}, ranks
    ranks, _ = run(tmp_path, paths, world, mode="skip_reconcile", port=29750 + world)
    assert not any(r["ok"] for r in ranks) and {r["code"] for r in ranks} == {-5}, ranks
    assert any("reconcile" in r["text"] for r in ranks)
    small = [str(tmp_path / f"s{i}.vcf") for i in range(world)]
    for i, p in enumerate(small):
        write_vcf(p, 2000, 40 + i, ["lowGQ"] if i % 2 == 0 else ["hiDP"])    # the 5 common FILTER values + one of its own
    ranks, _ = run(tmp_path, small, world, n_groups=6, port=29760 + world)   # every file fits 6 keys, the union (7) does not
    assert not any(r["ok"] for r in ranks) and {r["code"] for r in ranks} == {-6}, ranks
    assert all(r["seconds"] < 5.0 for r in ranks)


@pytest.mark.gpu
def test_a_rank_that_never_enters_is_met_by_a_bound(tmp_path):
    """EXON_HIP_FAULT=stall@1: rank 1 sleeps instead of entering.  The others' transport gives up after its own bound (gloo: 3 s here;
    the RCCL kind: EXON_HIP_COLLECTIVE_TIMEOUT_S + ncclCommAbort), the call fails on every rank -- nobody waits for ever."""
    paths = files(tmp_path, 2)
    ranks, wall = run(tmp_path, paths, 2, fault="stall@1", port=29790, env={"GLOO_TIMEOUT_S": "3", "EXON_HIP_COLLECTIVE_TIMEOUT_S": "1"}, timeout=120)
    assert not any(r["ok"] for r in ranks), ranks
    assert all(r["seconds"] < 10.0 for r in ranks), [r["seconds"] for r in ranks]
