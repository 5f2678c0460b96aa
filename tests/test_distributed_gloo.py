"""world_size-2 gloo test of the N>1 path: shard -> per-rank packed partial state -> ONE all-gather + fixed-order fold
-> final result.  The per-rank partials come from the oracle here (no GPU in this container); on the GPU box the same
`merge_state` runs on the kernels' device-resident state with backend "nccl" (= RCCL) and the fold is the HIP kernel."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exon_amd.distributed import all_reduce_state, finalize_avg, merge_state, shard_rows
    from oracle import Oracle
    orc = Oracle()
    lo, hi = shard_rows(n_total, rank, world)
    af, av, q, qv, fid = orc.gen_c4(4, lo, hi)
    s, cn, cr, _ = orc.c4_cmp_avg_by_group(af, av, q, qv, fid, orc.c4_filters(), 0.01, ">", threads=2)
    # the packed state [counts i64 x 10][sums f64 x 5 bit-cast], exactly what exon_hip_plan_launch writes
    state = torch.from_numpy(np.concatenate([cn, cr, s.view(np.int64)]))
    merged = merge_state(state, 10)
    counts, sums = merged[:10].clone(), merged[10:].view(torch.float64).clone()
    # every rank must hold the same bits (fixed-order fold): compare with rank 0's copy
    probe = merged.clone()
    dist.broadcast(probe, 0)
    assert torch.equal(probe, merged)
    # and the two-collective form agrees on the integers
    c2_, s2_ = torch.from_numpy(np.concatenate([cn, cr])), torch.from_numpy(s.copy())
    all_reduce_state(c2_, s2_)
    assert torch.equal(c2_, counts) and torch.allclose(s2_, sums, rtol=1e-14, atol=0)
    f, mq, mv, ref, rv = orc.gen_c3(3, lo, hi)
    c3, _ = orc.c3_flag_mapq_group_count(f, mq, mv, ref, rv, orc.c3_refs(), 1284, 0, 30, threads=2)
    c3 = merge_state(torch.from_numpy(c3), 26)
    if rank == 0:
        avg, rows = finalize_avg(counts.numpy(), sums.numpy(), 5)
        np.savez(out, counts=counts.numpy(), sums=sums.numpy(), c3=c3.numpy(), avg=np.array(avg, float), rows=rows)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_sharded_partials_reduce_to_the_single_rank_answer(tmp_path, oracle, world):
    n_total = 400_003
    out = str(tmp_path / "r.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n_total, out), nprocs=world, join=True)
    r = np.load(out)
    af, av, q, qv, fid = oracle.gen_c4(4, 0, n_total)
    s, cn, cr, _ = oracle.c4_cmp_avg_by_group(af, av, q, qv, fid, oracle.c4_filters(), 0.01, ">")
    assert np.array_equal(r["counts"], np.concatenate([cn, cr]))          # counts bit-exact
    assert np.allclose(r["sums"], s, rtol=1e-12, atol=0)                  # f64 sums: association order only
    assert np.allclose(r["avg"], s / cn, rtol=1e-12) and np.array_equal(r["rows"], cr)
    f, mq, mv, ref, rv = oracle.gen_c3(3, 0, n_total)
    c3, _ = oracle.c3_flag_mapq_group_count(f, mq, mv, ref, rv, oracle.c3_refs(), 1284, 0, 30)
    assert np.array_equal(r["c3"], c3)
