"""Multi-GPU merge of partial aggregates: the AggregateExec(Final) step across ranks.

One process per GPU; every rank scans its own file splits (the reference deals whole files round-robin by
ascending size: exon-core/src/datasources/exon_file_scan_config.rs:79-110) and owns a partial state of
int64 counters (+ float64 sums).  The only exchange on the path is one all-reduce(sum) of that state --
RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.  The payload is tiny (config 4: 120 B,
config 5: 204.8 KB), so it is latency-bound: counts and sums travel as two collectives, counts stay
integers (bit-exact), sums are float64.
"""
import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def shard_rows(n_total, rank, world, align=8):
    """Contiguous row range [lo, hi) of `rank`; boundaries aligned so validity bitmaps split on bytes."""
    per = (n_total + world - 1) // world
    per = (per + align - 1) // align * align
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def shard_files(sizes, rank, world):
    """Original indexes of the files this rank scans (same rule as regroup_files_by_size)."""
    from .engine import regroup_files_by_size
    groups = regroup_files_by_size(list(sizes), world)
    return groups[rank] if rank < len(groups) else []


def all_reduce_state(counts, sums=None, group=None):
    """In-place sum of the partial state over all ranks (no-op without an initialised process group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return counts, sums
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    if sums is not None:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return counts, sums


def finalize_avg(counts, sums, n_groups):
    """(avg[g] or None, rows[g]) from the reduced K4 state: counts = [count(y)[G], count(*)[G]], sums[G]."""
    c = [int(x) for x in counts]
    s = [float(x) for x in sums]
    avg = [s[g] / c[g] if c[g] else None for g in range(n_groups)]
    return avg, c[n_groups:2 * n_groups]
