"""Group keys travel by VALUE ("dictionaries identical across shards (checked; else union)").

K3 / K4 states are indexed by dictionary id and ids are per file: FILTER lists are numbered in order of first appearance in
a scan, references in each BAM / SAM header's own @SQ order.  The reference merges partitions by key value
(AggregateExec(Final) over the file groups of exon-core/src/datasources/exon_file_scan_config.rs:79-110).  These tests feed
files whose dictionaries DISAGREE -- FILTER lists appearing in opposite orders, @SQ lines in different orders, a reference
one file lacks -- through one stream, through two ranks, and compare with the oracle's answer over the single table that
holds all the rows.

CPU: exon_hip_keys_union, the host form of the re-keying, and a world-size-2 gloo run whose per-rank partials come from the
product's host decoders + numpy (no GPU in that container; on the GPU box the same reconcile + merge runs on device states).
-m gpu: the device path -- exon_hip_stream_consume_scan re-keying every further file, exon_hip_stream_set_keys,
exon_hip_stream_reconcile_keys over RCCL, the refusal of an unreconciled merge, two ranks sharing the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import exon_amd
from exon_amd import _lib as L
from exon_amd.distributed import permute_state, state_layout
from exon_amd.engine import keys_union

import oracle_expect as OX

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGZIP = os.path.join(ROOT, "tools", "bin", "bgzip")

VCF_HEAD = ('##fileformat=VCFv4.3\n##contig=<ID=1>\n##contig=<ID=2>\n##contig=<ID=7>\n'
            '##FILTER=<ID=q10,Description="x">\n##FILTER=<ID=s50,Description="x">\n'
            '##INFO=<ID=AF,Number=1,Type=Float,Description="x">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n')
FILTERS = ["PASS", ".", "q10", "q10;s50", "s50"]


def write_vcf(path, n, seed, first_filters, contig_order=("1", "2", "7")):
    """n rows; the first len(first_filters) rows carry those FILTER values in that order (so THEY decide the scan's ids),
    the rest are random; AF straddles 0.01, QUAL is in eighths (sums compare for equality), some NULLs of both."""
    rng = np.random.default_rng(seed)
    head = VCF_HEAD
    for c in ("1", "2", "7"):
        head = head.replace(f"##contig=<ID={c}>\n", "")
    head = head.replace("##FILTER=<ID=q10", "".join(f"##contig=<ID={c}>\n" for c in contig_order) + "##FILTER=<ID=q10", 1)
    lines = []
    for i in range(n):
        f = first_filters[i] if i < len(first_filters) else FILTERS[int(rng.integers(0, 5))]
        af = "." if rng.random() < 0.05 else ("%.4g" % (10 ** rng.uniform(-4, 0)))
        q = "." if rng.random() < 0.05 else str(int(rng.integers(0, 8000)) / 8)
        info = "." if rng.random() < 0.02 else f"AF={af};DP=3"
        chrom = contig_order[int(rng.integers(0, 3))]
        lines.append(f"{chrom}\t{i + 1}\t.\tA\tC\t{q}\t{f}\t{info}\n")
    with open(path, "w") as fh:
        fh.write(head + "".join(lines))
    return lines


def write_sam(path, n, seed, sq_order):
    """n alignment lines over the references of `sq_order` (that header order = the file's reference ids) + unmapped reads."""
    rng = np.random.default_rng(seed)
    with open(path, "w") as fh:
        fh.write("@HD\tVN:1.6\n" + "".join(f"@SQ\tSN:{r}\tLN:1000000\n" for r in sq_order))
        for i in range(n):
            unm = rng.random() < 0.05
            flag = 4 if unm else int(rng.choice([99, 147, 83, 163, 1123, 355]))
            ref = "*" if unm else sq_order[int(rng.integers(0, len(sq_order)))]
            if not unm and rng.random() < 0.03:   # no RNAME although the flag does not say unmapped: a row of the NULL-reference group
                ref, unm = "*", True
            pos = 0 if unm else int(rng.integers(1, 900000))
            mapq = int(rng.choice([0, 20, 30, 40, 60, 255]))
            cigar = "*" if unm else "50M"
            fh.write(f"r{i}\t{flag}\t{ref}\t{pos}\t{mapq}\t{cigar}\t*\t0\t0\t{'A' * 50}\t{'I' * 50}\n")


def cat_vcf(out, paths):
    """the single table: one header, every file's data lines"""
    with open(out, "w") as fh:
        fh.write(VCF_HEAD)
        for p in paths:
            fh.write("".join(ln for ln in open(p) if not ln.startswith("#")))


def k4_by_value(keys, counts, sums, G):
    return {keys[g]: (int(counts[g]), int(counts[G + g]), float(sums[g])) for g in range(len(keys)) if counts[G + g]}


def k3_expected_by_name(orc, paths, fmt):
    """{reference name or None: COUNT(*)} over all files -- per-file oracle counts added by NAME"""
    want = {}
    for p in paths:
        names, _ = OX.bam_columns(p, fmt)
        _, cnt = OX.k3_expected(orc, p, fmt)
        for g, c in enumerate(cnt):
            k = names[g] if g < len(names) else None
            if c:
                want[k] = want.get(k, 0) + int(c)
    return want


# ------------------------------------------------------------------------------------------------ CPU

def test_keys_union_is_rank_order_first_appearance_and_maps_every_local_id():
    union, maps = keys_union([["PASS", "", "q10"], ["q10", "PASS", "s50"], [], ["s50", "q10;s50"]])
    assert union == ["PASS", "", "q10", "s50", "q10;s50"]                # "" = the empty FILTER list is an ordinary key
    assert maps == [[0, 1, 2], [2, 0, 3], [], [3, 4]]
    assert keys_union([[]]) == ([], [[]])
    lib = exon_amd.load()
    n = (L.C.c_int32 * 1)(3)                                             # names that end before n_keys do: an error, not a read past
    no, nb = L.C.c_int32(), L.C.c_size_t()
    assert lib.exon_hip_keys_union(b"a\0b\0", 4, n, 1, None, 0, L.C.byref(no), L.C.byref(nb), None) == -1


def test_permute_state_host_form_moves_all_planes_and_keeps_the_null_group():
    G = 5
    lay = state_layout(L.PLAN_CMP_AVG_BY_GROUP, G)
    st = torch.zeros(3 * G, dtype=torch.int64)
    st[0:3] = torch.tensor([1, 2, 3])
    st[G:G + 3] = torch.tensor([10, 20, 30])
    st[2 * G:2 * G + 3] = torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64).view(torch.int64)
    out = permute_state(st, lay, [4, 0, 2])
    assert out[:G].tolist() == [2, 0, 3, 0, 1] and out[G:2 * G].tolist() == [20, 0, 30, 0, 10]
    assert out[2 * G:].view(torch.float64).tolist() == [2.5, 0, 3.5, 0, 1.5]
    lay3 = state_layout(L.PLAN_FLAG_MAPQ_GROUP_COUNT, 3)                  # count[3] + the NULL-reference group
    out = permute_state(torch.tensor([7, 8, 9, 100]), lay3, [2, 1, 0])
    assert out.tolist() == [9, 8, 7, 100]


def _host_k4_partial(path, G):
    """one rank's K4 partial state from the product's HOST decoder + numpy, indexed by the SCAN's dictionary ids"""
    s = exon_amd.Scan(path, "vcf", info_field="AF")
    counts, sums = np.zeros(2 * G, np.int64), np.zeros(G)
    for b in s:
        af = b.field(4).to_numpy(zero_copy_only=False).astype(np.float64)
        q = b.field(2).to_numpy(zero_copy_only=False).astype(np.float64)
        fid = np.asarray(b.field(3).indices)
        keep = ~np.isnan(af) & (af > 0.01)
        qv = keep & ~np.isnan(q)
        counts[:G] += np.bincount(fid[qv], minlength=G)
        counts[G:] += np.bincount(fid[keep], minlength=G)
        sums += np.bincount(fid[qv], weights=q[qv], minlength=G)
    keys = s.dictionary(3)
    s.close()
    return keys, torch.from_numpy(np.concatenate([counts, sums.view(np.int64)]))


def _host_k3_partial(path, G):
    s = exon_amd.Scan(path, "sam")
    counts = np.zeros(G + 1, np.int64)
    for b in s:
        flag = b.field(0).to_numpy(zero_copy_only=False)
        mq = b.field(1).to_numpy(zero_copy_only=False).astype(np.float64)
        ref = b.field(2)
        rid = np.asarray(ref.indices.fill_null(G))
        keep = ((flag & 1284) == 0) & ~np.isnan(mq) & (mq >= 30)
        counts += np.bincount(rid[keep], minlength=G + 1)
    keys = s.dictionary(2)
    s.close()
    return keys, torch.from_numpy(counts)


def _gloo_worker(rank, world, port, vcfs, sams, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["EXON_HIP_DECODE_THREADS"] = "1"   # sequential host decode: ids in order of first appearance, so the two files
    dist.init_process_group("gloo", rank=rank, world_size=world)  # provably number their FILTER lists in opposite orders
    from exon_amd.distributed import merge_state, reconcile_keys, shard_files
    G = 8
    mine = shard_files([os.path.getsize(p) for p in vcfs], rank, world)
    assert len(mine) == 1
    keys, state = _host_k4_partial(vcfs[mine[0]], G)
    local_keys = list(keys)
    union, state = reconcile_keys(keys=keys, state=state, layout=state_layout(L.PLAN_CMP_AVG_BY_GROUP, G))
    merged = merge_state(state, 2 * G)
    mine3 = shard_files([os.path.getsize(p) for p in sams], rank, world)
    keys3, st3 = _host_k3_partial(sams[mine3[0]], G)
    union3, st3 = reconcile_keys(keys=keys3, state=st3, layout=state_layout(L.PLAN_FLAG_MAPQ_GROUP_COUNT, G))
    merged3 = merge_state(st3, G + 1)
    both = [None, None]
    dist.all_gather_object(both, (local_keys, union, union3))
    assert both[0][1] == both[1][1] and both[0][2] == both[1][2]           # every rank computed the same unions
    if rank == 0:
        np.savez(out, k4=merged.numpy(), k3=merged3.numpy(), union=np.array(union), union3=np.array(union3),
                 local0=np.array(both[0][0]), local1=np.array(both[1][0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_whose_dictionaries_disagree_merge_by_value(tmp_path, oracle):
    """Two VCFs whose FILTER lists first appear in opposite orders, two SAMs whose @SQ lines are in different orders (and one
    reference only the second file has): per-rank partials by LOCAL id, reconciled, merged = the oracle over the single table."""
    a, b = str(tmp_path / "a.vcf"), str(tmp_path / "b.vcf")
    write_vcf(a, 3000, 1, ["PASS", ".", "q10", "q10;s50", "s50"])
    write_vcf(b, 3300, 2, ["s50", "q10;s50", "q10", ".", "PASS"])         # larger: regroup deals a -> rank 0, b -> rank 1
    sa, sb = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    write_sam(sa, 2000, 3, ["chr1", "chr2", "chr3"])
    write_sam(sb, 2200, 4, ["chrX", "chr3", "chr1", "chr2"])
    out = str(tmp_path / "r.npz")
    mp.spawn(_gloo_worker, args=(2, 29500 + (os.getpid() % 2000), [a, b], [sa, sb], out), nprocs=2, join=True)
    r = np.load(out)
    assert r["local0"].tolist() == ["PASS", "", "q10", "q10;s50", "s50"] and r["local1"].tolist() == ["s50", "q10;s50", "q10", "", "PASS"]
    G = 8
    union = r["union"].tolist()
    assert union == ["PASS", "", "q10", "q10;s50", "s50"]
    one = str(tmp_path / "all.vcf")
    cat_vcf(one, [a, b])
    n, want = OX.k4_expected(oracle, one, "vcf", "AF")
    assert n == 6300
    got = k4_by_value(union, r["k4"][:2 * G], r["k4"][2 * G:].view(np.float64), G)
    assert got == want                                                      # counts exact; QUAL in eighths: sums exact too
    # adding the two partials index by index -- what the merge did before -- is NOT the answer on these files
    os.environ["EXON_HIP_DECODE_THREADS"] = "1"
    ka, sa_ = _host_k4_partial(a, G)
    kb, sb_ = _host_k4_partial(b, G)
    del os.environ["EXON_HIP_DECODE_THREADS"]
    naive = k4_by_value(ka, (sa_ + sb_).numpy()[:2 * G], (sa_[2 * G:].view(torch.float64) + sb_[2 * G:].view(torch.float64)).numpy(), G)
    assert naive != want
    want3 = k3_expected_by_name(oracle, [sa, sb], "sam")
    union3 = r["union3"].tolist()
    assert union3 == ["chr1", "chr2", "chr3", "chrX"]
    got3 = {(union3[g] if g < len(union3) else None): int(c) for g, c in enumerate(r["k3"][:len(union3)]) if c}
    if r["k3"][G]:
        got3[None] = int(r["k3"][G])
    assert got3 == want3 and "chrX" in got3


# ------------------------------------------------------------------------------------------------ GPU

def _k4_plan(ctx, G=64):
    return ctx.plan_cmp_avg_by_group(">", 0.01, G, columns=(4, 2, 3))


@pytest.mark.gpu
def test_one_stream_consuming_files_with_opposite_filter_orders_is_keyed_by_value(ctx, tmp_path, oracle):
    """exon_hip_stream_consume_scan x 3 files on ONE stream (host decoders and GPU decoders): every further file is aggregated
    under its own ids and added in under the stream's; the result by key value = the oracle over the single table."""
    paths = [str(tmp_path / f"{c}.vcf") for c in "abc"]
    write_vcf(paths[0], 5000, 1, ["PASS", ".", "q10"])
    write_vcf(paths[1], 7000, 2, ["s50", "q10;s50", "q10", ".", "PASS"])
    write_vcf(paths[2], 6000, 3, ["q10;s50", "PASS"])
    one = str(tmp_path / "all.vcf")
    cat_vcf(one, paths)
    n, want = OX.k4_expected(oracle, one, "vcf", "AF")
    for gpu_parse in (False, True):
        plan = _k4_plan(ctx)
        st = plan.open()
        rows = 0
        for p in paths:
            s = exon_amd.Scan(p, "vcf", info_field="AF", gpu_parse=gpu_parse)
            rows += st.consume(s)
            if gpu_parse:
                assert s.decoded_on_gpu()[0]
            s.close()
        keys, agreed = st.keys()
        # (which id a value gets inside one scan is the decoder's business -- several decode threads / wavefronts intern
        # concurrently --; what is checked is that every file's ids were translated to the stream's by VALUE)
        assert rows == n and not agreed and sorted(keys) == sorted(["PASS", "", "q10", "q10;s50", "s50"])
        c, s_ = st.finish()
        assert k4_by_value(keys, c, s_, 64) == want
        st.close()
        plan.close()


@pytest.mark.gpu
def test_set_keys_permutes_the_device_state_and_rejects_a_dictionary_that_drops_a_key(ctx, tmp_path, oracle):
    p = str(tmp_path / "a.vcf")
    write_vcf(p, 4000, 5, ["PASS", ".", "q10", "q10;s50", "s50"])
    n, want = OX.k4_expected(oracle, p, "vcf", "AF")
    plan = _k4_plan(ctx, 8)
    st = plan.open()
    s = exon_amd.Scan(p, "vcf", info_field="AF", gpu_parse=True)
    st.consume(s)
    s.close()
    with pytest.raises(exon_amd.ExonHipError, match="lacks the key"):
        st.set_keys(["PASS", "q10"])
    with pytest.raises(exon_amd.ExonHipError, match="twice"):
        st.set_keys(["PASS", "", "q10", "q10;s50", "s50", "PASS"])
    with pytest.raises(exon_amd.ExonHipError, match="n_groups"):
        st.set_keys(["PASS", "", "q10", "q10;s50", "s50", "a", "b", "c", "d"])
    new = ["zz", "s50", "q10;s50", "q10", "", "PASS", "never"]
    st.set_keys(new)
    assert st.keys() == (new, True)
    c, s_ = st.snapshot()
    assert k4_by_value(new, c, s_, 8) == want and c[0] == 0 and c[6] == 0
    st.set_keys(new)                                                        # the same order again: nothing moves
    c2, s2 = st.finish()
    assert np.array_equal(c, c2) and np.array_equal(s_, s2)
    st.close()
    plan.close()


@pytest.mark.gpu
def test_bam_files_with_different_sq_orders_count_by_reference_name(ctx, tmp_path, oracle):
    """K3 GROUP BY reference over a SAM / BAM pair whose headers order (and own) different references; the NULL-reference
    group stays the last word of the state."""
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    sa, sb = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    write_sam(sa, 30000, 3, ["chr1", "chr2", "chr3"])
    write_sam(sb, 34000, 4, ["chrX", "chr3", "chr1", "chr2"])
    want = k3_expected_by_name(oracle, [sa, sb], "sam")
    G = 6
    for gpu_parse in (False, True):
        plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, G)
        st = plan.open()
        for p in (sa, sb):
            s = exon_amd.Scan(p, "sam", gpu_parse=gpu_parse)
            st.consume(s)
            s.close()
        keys, _ = st.keys()
        assert keys == ["chr1", "chr2", "chr3", "chrX"]
        c, _ = st.finish()
        got = {keys[g]: int(c[g]) for g in range(len(keys)) if c[g]}
        if c[G]:
            got[None] = int(c[G])
        assert got == want and want.get(None, 0) > 0
        st.close()
        plan.close()
    assert os.path.exists(gen)


@pytest.mark.gpu
def test_region_contig_by_name_over_files_with_different_contig_orders(ctx, tmp_path):
    """K2 over two VCFs whose headers list the contigs in different orders: exon_hip_stream_set_region_contig resolves '7' per
    file (a fixed region_chrom_id would count contig '1' of the second file)."""
    a, b = str(tmp_path / "a.vcf"), str(tmp_path / "b.vcf")
    write_vcf(a, 5000, 1, [], contig_order=("1", "2", "7"))
    write_vcf(b, 5000, 2, [], contig_order=("7", "1", "2"))
    want = sum(OX.region_count_expected(p, "vcf", "7", 100, 4000)[1] for p in (a, b))
    for gpu_parse in (False, True):
        plan = ctx.plan_region_count(0, 100, 4000)                           # the id in the plan is a placeholder
        st = plan.open()
        st.set_region_contig("7")
        for p in (a, b):
            s = exon_amd.Scan(p, "vcf", gpu_parse=gpu_parse)
            st.consume(s)
            s.close()
        c, _ = st.finish()
        assert int(c[0]) == want and want > 0
        st.close()
        plan.close()
    plan = _k4_plan(ctx)
    st = plan.open()
    with pytest.raises(exon_amd.ExonHipError, match="no region"):
        st.set_region_contig("7")
    st.close()
    plan.close()


@pytest.mark.gpu
def test_merge_of_a_rank_local_dictionary_is_refused_until_reconciled(ctx, tmp_path, oracle):
    """exon_hip_stream_all_reduce on a state keyed by this rank's own scans: EXON_HIP_ESTATE.  After
    exon_hip_stream_reconcile_keys on a (one-rank) RCCL communicator the same call goes through and the answer stands."""
    from exon_amd.distributed import NativeComm
    p = str(tmp_path / "a.vcf")
    write_vcf(p, 4000, 7, ["s50", "PASS"])
    n, want = OX.k4_expected(oracle, p, "vcf", "AF")
    plan = _k4_plan(ctx)
    st = plan.open()
    s = exon_amd.Scan(p, "vcf", info_field="AF", gpu_parse=True)
    st.consume(s)
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        comm = NativeComm(ctx)
        with pytest.raises(exon_amd.ExonHipError) as e:
            st.all_reduce(comm.h.value)                                       # refused in the vote (ABI 5): every rank would return this
        assert e.value.code == -5 and "reconcile" in str(e.value)
        st.reconcile_keys(comm.h.value)
        keys, agreed = st.keys()
        assert agreed and {"PASS", "s50"} <= set(keys)
        st.all_reduce(comm.h.value)
        c, s_ = st.finish()
        assert k4_by_value(keys, c, s_, 64) == want
        comm.close()
    finally:
        dist.destroy_process_group()
    st.close()
    plan.close()


_RANK_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import exon_amd
from exon_amd.distributed import scan_files
dist.init_process_group("gloo")
ctx = exon_amd.Context(0)
paths = json.loads(sys.argv[2])
kind = sys.argv[3]
if kind == "k4":
    r = scan_files(ctx, paths, "vcf", lambda c: c.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3)), info_field="AF")
elif kind == "k4small":  # capacity for 4 distinct FILTER lists only
    r = scan_files(ctx, paths, "vcf", lambda c: c.plan_cmp_avg_by_group(">", 0.01, 4, columns=(4, 2, 3)), info_field="AF")
else:
    r = scan_files(ctx, paths, "sam", lambda c: c.plan_flag_mapq_group_count(1284, 0, 30, 8))
both = [None] * dist.get_world_size()
dist.all_gather_object(both, (r["keys"], r["counts"].tolist(), r["sums"].tolist(), r["files"]))
assert all(b[:3] == both[0][:3] for b in both), "ranks disagree after the merge"
if dist.get_rank() == 0:
    print("RESULT " + json.dumps({"keys": r["keys"], "counts": r["counts"].tolist(), "sums": r["sums"].tolist(), "rows": r["rows"],
                                  "files": [b[3] for b in both]}))
dist.barrier()
ctx.close()
dist.destroy_process_group()
'''


def _ranks(tmp_path, paths, kind, port, world=2, expect_fail=False):
    import json
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT, json.dumps(paths), kind]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900)
    if expect_fail:
        return r
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


def _two_ranks(tmp_path, paths, kind, port):
    return _ranks(tmp_path, paths, kind, port, 2)


@pytest.mark.gpu
def test_two_ranks_scan_files_with_disagreeing_dictionaries(tmp_path, oracle):
    """distributed.scan_files under the driver's launch line, two ranks sharing the one GPU (gloo moves the names and the
    states; the device states are permuted by exon_hip_stream_set_keys): three BGZF VCFs dealt by regroup_files_by_size whose
    FILTER lists first appear in different orders = the oracle over the single table; same for SAMs with different @SQ orders."""
    paths = [str(tmp_path / f"{c}.vcf") for c in "abc"]
    write_vcf(paths[0], 20000, 1, ["PASS", ".", "q10"])
    write_vcf(paths[1], 26000, 2, ["s50", "q10;s50", "q10", ".", "PASS"])
    write_vcf(paths[2], 23000, 3, ["q10;s50", "PASS"])
    one = str(tmp_path / "all.vcf")
    cat_vcf(one, paths)
    n, want = OX.k4_expected(oracle, one, "vcf", "AF")
    gz = []
    for p in paths:
        subprocess.check_call([BGZIP, p, p + ".gz"])
        gz.append(p + ".gz")
    r = _two_ranks(tmp_path, gz, "k4", 29541)
    assert r["rows"] == n and sorted(len(f) for f in r["files"]) == [1, 2]
    assert k4_by_value(r["keys"], np.array(r["counts"]), np.array(r["sums"]), 64) == want
    sams = [str(tmp_path / f"{c}.sam") for c in "ab"]
    write_sam(sams[0], 20000, 3, ["chr1", "chr2", "chr3"])
    write_sam(sams[1], 24000, 4, ["chrX", "chr3", "chr1", "chr2"])
    want3 = k3_expected_by_name(oracle, sams, "sam")
    r = _two_ranks(tmp_path, sams, "k3", 29542)
    got3 = {r["keys"][g]: int(c) for g, c in enumerate(r["counts"][:len(r["keys"])]) if c}
    if r["counts"][8]:
        got3[None] = int(r["counts"][8])
    assert got3 == want3 and r["rows"] == 44000


@pytest.mark.gpu
def test_eight_ranks_three_files_five_ranks_with_empty_dictionaries(tmp_path, oracle):
    """The driver's 8-GPU launch line rehearsed on the one GPU (eight ranks share cuda:0, gloo carries names and states): three files
    over eight ranks = five ranks that scan nothing and bring an EMPTY dictionary and an all-zero state into the reconcile and the
    merge (regroup_files_by_size leaves their groups empty, exon_file_scan_config.rs:79-110).  Every rank ends with the oracle's
    answer over the concatenated table; same for two SAMs over eight ranks."""
    paths = [str(tmp_path / f"{c}.vcf") for c in "abc"]
    write_vcf(paths[0], 9000, 1, ["PASS", ".", "q10"])
    write_vcf(paths[1], 12000, 2, ["s50", "q10;s50", "q10", ".", "PASS"])
    write_vcf(paths[2], 10000, 3, ["q10;s50", "PASS"])
    one = str(tmp_path / "all.vcf")
    cat_vcf(one, paths)
    n, want = OX.k4_expected(oracle, one, "vcf", "AF")
    gz = []
    for p in paths:
        subprocess.check_call([BGZIP, p, p + ".gz"])
        gz.append(p + ".gz")
    r = _ranks(tmp_path, gz, "k4", 29551, world=8)
    assert r["rows"] == n and sorted(len(f) for f in r["files"]) == [0, 0, 0, 0, 0, 1, 1, 1]
    assert k4_by_value(r["keys"], np.array(r["counts"]), np.array(r["sums"]), 64) == want
    sams = [str(tmp_path / f"{c}.sam") for c in "ab"]
    write_sam(sams[0], 9000, 3, ["chr1", "chr2", "chr3"])
    write_sam(sams[1], 11000, 4, ["chrX", "chr3", "chr1", "chr2"])
    want3 = k3_expected_by_name(oracle, sams, "sam")
    r = _ranks(tmp_path, sams, "k3", 29552, world=8)
    got3 = {r["keys"][g]: int(c) for g, c in enumerate(r["counts"][:len(r["keys"])]) if c}
    if r["counts"][8]:
        got3[None] = int(r["counts"][8])
    assert got3 == want3 and r["rows"] == 20000


@pytest.mark.gpu
def test_a_union_beyond_the_plans_capacity_fails_on_every_rank(tmp_path):
    """Each rank's own dictionary fits the plan (3 and 3 distinct FILTER lists, capacity 4), the UNION (5) does not: the
    reconcile must fail loudly on every rank -- not merge a truncated dictionary -- and the launcher comes back non-zero."""
    paths = [str(tmp_path / f"{c}.vcf") for c in "ab"]
    for path, seed, filters in ((paths[0], 1, ["PASS", ".", "q10"]), (paths[1], 2, ["s50", "q10;s50", "PASS"])):
        rng = np.random.default_rng(seed)
        rows = [f"1\t{i + 1}\t.\tA\tC\t{int(rng.integers(0, 8000)) / 8}\t{filters[i] if i < 3 else filters[int(rng.integers(0, 3))]}\tAF={10 ** rng.uniform(-4, 0):.4g}\n"
                for i in range(6000)]
        with open(path, "w") as fh:
            fh.write(VCF_HEAD + "".join(rows))
    r = _ranks(tmp_path, paths, "k4small", 29553, world=2, expect_fail=True)
    assert r.returncode != 0 and "RESULT " not in r.stdout
    assert "n_groups" in (r.stdout + r.stderr) or "distinct" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2000:]


@pytest.mark.gpu
def test_a_scan_that_does_not_fit_leaves_the_stream_as_it_was(ctx, tmp_path, oracle):
    """exon_hip_stream_consume_scan of a file whose FILTER lists do not fit the plan's n_groups any more: EXON_HIP_ECAPACITY (-6),
    and the stream's dictionary and state are EXACTLY what they were before that file (the keys used to be appended
    before the failure) -- finishing the stream gives the first file's answer; the file then goes through a fresh stream (what the
    shim does: one more partial batch for AggregateExec(Final))."""
    a, b = str(tmp_path / "a.vcf"), str(tmp_path / "b.vcf")

    def write(path, seed, filters):  # every row draws from `filters` only: the file's dictionary is exactly that set
        rng = np.random.default_rng(seed)
        rows = []
        for i in range(5000):
            f = filters[i] if i < len(filters) else filters[int(rng.integers(0, len(filters)))]
            af = "." if rng.random() < 0.05 else ("%.4g" % (10 ** rng.uniform(-4, 0)))
            q = "." if rng.random() < 0.05 else str(int(rng.integers(0, 8000)) / 8)
            rows.append(f"1\t{i + 1}\t.\tA\tC\t{q}\t{f}\tAF={af}\n")
        with open(path, "w") as fh:
            fh.write(VCF_HEAD + "".join(rows))

    write(a, 1, ["PASS", ".", "q10"])                 # 3 keys
    write(b, 2, ["s50", "q10;s50", "PASS"])           # 3 keys, two of them new: the union (5) exceeds n_groups = 4
    na, want_a = OX.k4_expected(oracle, a, "vcf", "AF")
    nb, want_b = OX.k4_expected(oracle, b, "vcf", "AF")
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 4, columns=(4, 2, 3))
    st = plan.open()
    s = exon_amd.Scan(a, "vcf", info_field="AF", gpu_parse=True)
    assert st.consume(s) == na
    s.close()
    keys_before = st.keys()[0]
    s = exon_amd.Scan(b, "vcf", info_field="AF", gpu_parse=True)
    with pytest.raises(exon_amd.ExonHipError) as e:
        st.consume(s)
    s.close()
    assert e.value.code == -6 and "n_groups" in str(e.value)
    assert st.keys()[0] == keys_before
    c, sm = st.finish()
    assert k4_by_value(keys_before, c, sm, 4) == want_a
    st.close()
    st = plan.open()                                   # the file that did not fit, on a stream of its own
    s = exon_amd.Scan(b, "vcf", info_field="AF", gpu_parse=True)
    assert st.consume(s) == nb
    s.close()
    keys_b = st.keys()[0]
    c, sm = st.finish()
    assert k4_by_value(keys_b, c, sm, 4) == want_b
    st.close()
    plan.close()


@pytest.mark.gpu
def test_a_first_scan_that_does_not_fit_leaves_the_stream_empty(ctx, tmp_path, oracle):
    """the FIRST scan of a stream is not redirected into a scratch state (there is nothing to protect), so its rows
    land in the state before the dictionary is found too large.  EXON_HIP_ECAPACITY must still leave the stream as it was --
    here: EMPTY, no keys, zero counts -- and usable: a file that fits, consumed next, gives exactly its own answer."""
    a, b = str(tmp_path / "a.vcf"), str(tmp_path / "b.vcf")

    def write(path, seed, filters):
        rng = np.random.default_rng(seed)
        rows = []
        for i in range(4000):
            f = filters[i] if i < len(filters) else filters[int(rng.integers(0, len(filters)))]
            af = "." if rng.random() < 0.05 else ("%.4g" % (10 ** rng.uniform(-4, 0)))
            q = "." if rng.random() < 0.05 else str(int(rng.integers(0, 8000)) / 8)
            rows.append(f"1\t{i + 1}\t.\tA\tC\t{q}\t{f}\tAF={af}\n")
        with open(path, "w") as fh:
            fh.write(VCF_HEAD + "".join(rows))

    write(a, 3, ["PASS", ".", "q10", "s50", "q10;s50"])   # 5 keys: does not fit n_groups = 4
    write(b, 4, ["PASS", "q10", "."])
    nb, want_b = OX.k4_expected(oracle, b, "vcf", "AF")
    plan = ctx.plan_cmp_avg_by_group(">", 0.01, 4, columns=(4, 2, 3))
    st = plan.open()
    s = exon_amd.Scan(a, "vcf", info_field="AF", gpu_parse=True)
    with pytest.raises(exon_amd.ExonHipError) as e:
        st.consume(s)
    s.close()
    assert e.value.code == -6 and "n_groups" in str(e.value)
    assert st.keys()[0] == []
    s = exon_amd.Scan(b, "vcf", info_field="AF", gpu_parse=True)
    assert st.consume(s) == nb
    s.close()
    keys_b = st.keys()[0]
    c, sm = st.finish()
    assert k4_by_value(keys_b, c, sm, 4) == want_b   # nothing of file a is left in the state
    st.close()
    st = plan.open()                                 # and finishing right after the failure gives the empty answer
    s = exon_amd.Scan(a, "vcf", info_field="AF", gpu_parse=True)
    with pytest.raises(exon_amd.ExonHipError):
        st.consume(s)
    s.close()
    c, sm = st.finish()
    assert not np.any(c) and not np.any(sm)
    st.close()
    plan.close()
