"""Multi-GPU merge of partial aggregates: the AggregateExec(Final) step across ranks.

One process per GPU; every rank scans its own file splits (the reference deals whole files round-robin by
ascending size: exon-core/src/datasources/exon_file_scan_config.rs:79-110) and owns ONE packed partial state
`[n_i64 x int64][n_f64 x float64]` (config 4: 15 words = 120 B, config 5: 204.8 KB).  The only exchange on the
path is ONE all-gather of that state followed by a fold in rank order 0..world-1 on every rank: counts stay
integers (bit-exact), the float64 sums come out bit-identical on all ranks and for every algorithm the collective
library may pick (an all-reduce's association order is the library's business).  Latency-bound, so it is issued

* natively -- `exon_hip_merge_states`: ncclAllGather on the kernels' own hipStream_t through a communicator made with
  `exon_hip_rccl_comm_init` (no cross-stream hops; what a Rust host would call), or
* through torch.distributed (`all_gather_into_tensor`; backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)
  followed by the same fold kernel (`exon_hip_fold_states`; a torch loop for CPU tensors, which only tests hold).
"""
import ctypes as C
import os

from ._lib import PLAN_CMP_AVG_BY_GROUP as L_PLAN_K4, PLAN_FLAG_MAPQ_GROUP_COUNT as L_PLAN_K3


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


def _initialised(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def fold_states(gathered, world, n_i64, out, ctx=None, stream=None):
    """out[v] = sum over ranks r = 0..world-1 (in that order) of gathered[r][v]; words [0, n_i64) are int64 counters,
    the rest float64 sums bit-cast into the int64 tensor.  Device tensors go through the HIP kernel of the C ABI."""
    import torch
    V = out.numel()
    if gathered.is_cuda:
        if ctx is None:
            raise RuntimeError("fold_states on device tensors needs the exon_amd Context (no torch fallback on the GPU)")
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        ctx._check(ctx.lib.exon_hip_fold_states(ctx.h, s, gathered.data_ptr(), world, n_i64, V - n_i64, out.data_ptr()))
        return out
    g = gathered.view(world, V)
    out[:n_i64] = g[:, :n_i64].sum(0)
    if V > n_i64:
        acc = torch.zeros(V - n_i64, dtype=torch.float64)
        for r in range(world):  # fixed order, like the kernel
            acc = acc + g[r, n_i64:].view(torch.float64)
        out[n_i64:] = acc.view(torch.int64)
    return out


def merge_state(state, n_i64, gathered=None, out=None, group=None, ctx=None):
    """ONE all-gather of the packed int64-typed `state` (float64 sums bit-cast) + the fixed-order fold.  Returns the
    merged state (`out`, or a new tensor); without an initialised process group of more than one rank: `state`."""
    import torch
    import torch.distributed as dist
    if not _initialised(group):
        if out is not None and out.data_ptr() != state.data_ptr():
            out.copy_(state)
            return out
        return state
    world = dist.get_world_size(group)
    V = state.numel()
    if gathered is None:
        gathered = torch.empty(world * V, dtype=state.dtype, device=state.device)
    if out is None:
        out = torch.empty_like(state)
    # gloo knows CPU tensors only for this collective: a device-resident state (the one-GPU launcher test) is staged
    if state.is_cuda and dist.get_backend(group) == "gloo":
        h = torch.empty(world * V, dtype=state.dtype)
        dist.all_gather_into_tensor(h, state.cpu(), group=group)
        gathered.copy_(h)
    else:
        dist.all_gather_into_tensor(gathered, state, group=group)
    return fold_states(gathered, world, n_i64, out, ctx=ctx)


# ---- group keys by VALUE across ranks ("dictionaries identical across shards, else union") -----------------
# A K3 / K4 state is indexed by dictionary id and ids are per file (FILTER lists: order of first appearance; references: each
# BAM's @SQ order).  Before the states of different ranks may be added index by index, the ranks agree on ONE dictionary -- the
# union in rank order -- and every rank permutes its state into it.  The reference gets the same result from
# AggregateExec(Final), which merges the partitions' partial states by key value.

def state_layout(kind, n_groups):
    """(G, planes_i64, tail_i64, planes_f64) of a plan's packed state, as exon_hip_stream_set_keys lays it out:
    [planes_i64 x G int64][tail_i64 int64 (K3: the NULL-reference group)][planes_f64 x G float64]."""
    from ._lib import PLAN_CMP_AVG_BY_GROUP, PLAN_FLAG_MAPQ_GROUP_COUNT
    if kind == PLAN_FLAG_MAPQ_GROUP_COUNT:
        return n_groups, 1, 1, 0
    if kind == PLAN_CMP_AVG_BY_GROUP:
        return n_groups, 2, 0, 1
    raise ValueError(f"plan kind {kind} has no group keys")


def permute_state(state, layout, mapping):
    """CPU statement of the device re-keying (launch_permute_add_state): a packed int64-typed host tensor keyed by local
    ids -> the same state under the ids `mapping[local]`.  Host tensors only -- tests and gloo launchers hold those; a
    device-resident state is permuted by exon_hip_stream_set_keys."""
    import torch
    if state.is_cuda:
        raise RuntimeError("permute_state is the host form; a device state is re-keyed by Stream.set_keys (no torch fallback on the GPU)")
    G, pi, tail, pf = layout
    out = torch.zeros_like(state)
    idx = torch.as_tensor(list(mapping), dtype=torch.int64)
    src = torch.arange(len(idx))
    for p in range(pi):
        out[p * G:(p + 1) * G].index_add_(0, idx, state[p * G + src])
    out[pi * G:pi * G + tail] = state[pi * G:pi * G + tail]
    fb = pi * G + tail
    for p in range(pf):
        o = torch.zeros(G, dtype=torch.float64)
        o.index_add_(0, idx, state[fb + p * G + src].view(torch.float64))
        out[fb + p * G:fb + (p + 1) * G] = o.view(torch.int64)
    return out


def reconcile_keys(keys=None, state=None, layout=None, stream=None, group=None):
    """Every rank of `group` ends up with the same dictionary (the union in rank order, first appearance first) and a state
    keyed by it.  Stream form (`stream=`): the names travel through torch.distributed, the device state is permuted by
    exon_hip_stream_set_keys.  Host form (`keys=, state=, layout=`): returns (union, permuted host tensor).  Collective."""
    import torch.distributed as dist
    from .engine import keys_union
    if stream is not None:
        keys, _ = stream.keys()
    keys = list(keys)
    if _initialised(group):
        world = dist.get_world_size(group)
        dicts = [None] * world
        dist.all_gather_object(dicts, keys, group=group)
        rank = dist.get_rank(group)
    else:
        dicts, rank = [keys], 0
    union, maps = keys_union(dicts)
    if stream is not None:
        stream.set_keys(union)
        return union
    return union, permute_state(state, layout, maps[rank])


def scan_files(ctx, paths, fmt, make_plan, rank=None, world=None, group=None, comm=None, region_contig=None, **scan_kw):
    """A sharded file scan end to end, one process per GPU: the files are dealt by `regroup_files_by_size`
    (exon-core/src/datasources/exon_file_scan_config.rs:79-110), this rank's files are consumed on the GPU into ONE stream
    (keys by value: every file is re-keyed into the stream's dictionary), the ranks agree on the union dictionary, and the
    states are merged -- natively over RCCL when `comm` (a NativeComm) is given, else through torch.distributed.
    `make_plan(ctx)` builds the same plan on every rank (its n_groups = the capacity for distinct keys).
    Returns {"keys", "counts", "sums", "rows", "files"} with counts / sums as host arrays, equal on every rank."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from .engine import Scan
    if rank is None or world is None:
        if _initialised(group):
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        else:
            rank, world = 0, 1
    sizes = [os.path.getsize(p) for p in paths]
    mine = shard_files(sizes, rank, world)
    plan = make_plan(ctx)
    stream = plan.open(rank)
    rows = 0
    try:
        if region_contig is not None:
            stream.set_region_contig(region_contig)
        for i in mine:
            scan = Scan(paths[i], fmt, gpu_parse=True, **scan_kw)
            try:
                rows += stream.consume(scan)
            finally:
                scan.close()
        keyed = plan.desc.kind in (L_PLAN_K3, L_PLAN_K4)
        if world > 1 and comm is not None:
            if keyed:
                stream.reconcile_keys(comm.h.value)
            stream.all_reduce(comm.h.value)
            counts, sums = stream.snapshot()
        else:
            if keyed and world > 1:
                reconcile_keys(stream=stream, group=group)
            counts, sums = stream.snapshot()
            if world > 1:
                state = torch.from_numpy(np.concatenate([counts, sums.view(np.int64)]))
                merged = merge_state(state, len(counts), group=group).numpy()
                counts, sums = merged[:len(counts)].copy(), merged[len(counts):].view(np.float64).copy()
        keys = stream.keys()[0] if keyed else []
        total_rows = rows
        if world > 1 and _initialised(group):
            dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else torch.device("cuda", ctx.device)
            t = torch.tensor([rows], dtype=torch.int64, device=dev)
            dist.all_reduce(t, group=group)
            total_rows = int(t[0])
        return {"keys": keys, "counts": counts, "sums": sums, "rows": total_rows, "files": [paths[i] for i in mine]}
    finally:
        stream.close()
        plan.close()


class CallbackComm:
    """exon_hip_comm_from_callbacks over a torch.distributed group (gloo works): the library's collectives -- votes, key
    reconciliation, merge -- with the bytes moved by the caller's all-gather of host buffers.  For hosts that own a transport
    already, and for rehearsals of the multi-rank logic where RCCL cannot form a communicator (several ranks on one GPU)."""

    def __init__(self, ctx, group=None):
        import torch
        import torch.distributed as dist
        from ._lib import ALLGATHER_FN
        self.ctx, self.h = ctx, None
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

        def all_gather(_user, send, recv, nbytes):
            try:
                mine = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
                out = torch.empty(nbytes * self.world, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, mine, group=group)
                C.memmove(recv, out.data_ptr(), nbytes * self.world)
                return 0
            except Exception:  # noqa: BLE001 -- a time-out / a dead peer: the library reports it
                return -1
        self._cb = ALLGATHER_FN(all_gather)  # (kept alive with the communicator)
        h = C.c_void_p()
        ctx._check(ctx.lib.exon_hip_comm_from_callbacks(self.world, self.rank, self._cb, None, C.byref(h)))
        self.h = h

    def count(self):
        w, r = C.c_int32(), C.c_int32()
        self.ctx._check(self.ctx.lib.exon_hip_rccl_comm_count(self.h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def close(self):
        if self.h:
            self.ctx.lib.exon_hip_rccl_comm_destroy(self.h)
            self.h = None


class NativeComm:
    """An ncclComm_t created through the C ABI (exon_hip_rccl_unique_id / _comm_init): rank 0 makes the unique id, the
    128 bytes travel through the already initialised torch.distributed group, every rank joins on its own GPU."""

    def __init__(self, ctx, group=None):
        import torch
        import torch.distributed as dist
        self.ctx = ctx
        self.h = None
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            ctx._check(ctx.lib.exon_hip_rccl_unique_id(uid))
        dev = torch.device("cuda", ctx.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0, group=group)
        uid = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        ctx._check(ctx.lib.exon_hip_rccl_comm_init(ctx.h, uid, world, rank, C.byref(h)))
        self.h, self.world, self.rank = h, world, rank

    def count(self):
        """(ncclCommCount, ncclCommUserRank) of the communicator itself -- not what the launcher believes."""
        w, r = C.c_int32(), C.c_int32()
        self.ctx._check(self.ctx.lib.exon_hip_rccl_comm_count(self.h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def merge(self, state, n_i64, gathered, out, stream=None):
        """exon_hip_merge_states on `stream` (default: torch's current stream): ncclAllGather + fold, no stream hops."""
        import torch
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.ctx._check(self.ctx.lib.exon_hip_merge_states(self.ctx.h, s, self.h, state.data_ptr(), n_i64,
                                                           state.numel() - n_i64, gathered.data_ptr(), out.data_ptr()))
        return out

    def close(self):
        if self.h:
            self.ctx.lib.exon_hip_rccl_comm_destroy(self.h)
            self.h = None


def all_reduce_state(counts, sums=None, group=None):
    """In-place sum of separately held counters / sums over all ranks (two all-reduces; kept for hosts that hold the
    state as two tensors -- the packed form above needs one collective)."""
    import torch.distributed as dist
    if not _initialised(group):
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
