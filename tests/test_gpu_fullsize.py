"""GPU tests at BASELINE.json's full sizes, through size-independent properties (the oracle cannot run 1e9 rows in
a test): partition identities, shard additivity (the multi-GPU merge rule), determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def k4(ctx, cols, n, thr, op, G=5):
    af, av, q, qv, fid = cols
    dc, ds = ctx.zeros(np.int64, 2 * G), ctx.zeros(np.float64, G)
    ctx.cmp_avg_by_group(af, av, q, qv, fid, n, thr, op, G, dc, ds)
    ctx.sync()
    return dc.to_host(), ds.to_host()


def test_config4_one_billion_rows_properties(ctx):
    n = 1_000_000_000
    cols = ctx.gen_c4(4, 0, n)
    gt_c, gt_s = k4(ctx, cols, n, 0.01, ">")
    le_c, le_s = k4(ctx, cols, n, 0.01, "<=")
    all_c, all_s = k4(ctx, cols, n, float("-inf"), ">=")
    # {x > t} and {x <= t} partition the rows with a valid x
    assert np.array_equal(gt_c + le_c, all_c)
    assert np.allclose(gt_s + le_s, all_s, rtol=1e-12, atol=0)
    valid_rows = int(all_c[5:].sum())
    assert 0.9895 * n < valid_rows < 0.9905 * n                      # 1 % NULL AF by construction
    assert abs(gt_c[5:].sum() / valid_rows - 0.4803) < 0.002         # selectivity of AF > 0.01 for this generator
    assert np.all(gt_c[:5] <= gt_c[5:])                              # COUNT(qual) <= COUNT(*)
    shares = gt_c[5:] / gt_c[5:].sum()
    assert np.allclose(shares, [0.85, 0.05, 0.06, 0.03, 0.01], atol=2e-4)
    avg = gt_s / gt_c[:5]
    assert np.all(np.abs(avg - 499.95) < 0.6)                        # qual ~ U{0.0 .. 999.9}
    # shard additivity: 8 file splits processed separately add up to the whole (what the RCCL merge relies on)
    tot_c, tot_s = np.zeros(10, np.int64), np.zeros(5)
    per = n // 8
    for r in range(8):
        shard = ctx.gen_c4(4, r * per, (r + 1) * per)
        c, s = k4(ctx, shard, per, 0.01, ">")
        tot_c += c
        tot_s += s
        del shard
    assert np.array_equal(tot_c, gt_c) and np.allclose(tot_s, gt_s, rtol=1e-12, atol=0)
    # run-to-run determinism at full size
    again_c, again_s = k4(ctx, cols, n, 0.01, ">")
    assert np.array_equal(again_c, gt_c) and again_s.tobytes() == gt_s.tobytes()


def test_config2_region_partition(ctx, oracle):
    n = 10_000_000  # config 2's stated size
    c, p = ctx.gen_c2(2, n)
    starts = np.zeros(25, np.int64)
    lens = oracle.c2_contig_lens()
    total = sum(lens)
    cum = 0
    for i, ln in enumerate(lens):
        cum += ln
        starts[i + 1] = n * cum // total
    starts[24] = n

    def count(cid, a, b):
        d = ctx.zeros(np.int64, 1)
        ctx.region_count(c, p, n, cid, a, b, d)
        ctx.sync()
        return int(d.to_host()[0])

    per_contig = [count(cid, 1, None) for cid in range(24)]
    assert per_contig == [int(starts[i + 1] - starts[i]) for i in range(24)] and sum(per_contig) == n
    assert count(6, 1, 49_999_999) + count(6, 50_000_000, 100_000_000) + count(6, 100_000_001, None) == per_contig[6]
    hc, hp = oracle.gen_c2(2, n)
    assert count(6, 50_000_000, 100_000_000) == oracle.c2_region_count(hc, hp, oracle.c2_contigs(), "7:50000000-100000000")[0]


def test_config3_one_hundred_million_rows_properties(ctx):
    n = 100_000_000
    f, mq, mv, ref, rv = ctx.gen_c3(3, 0, n)

    def run(mask, value, qmin):
        d = ctx.zeros(np.int64, 26)
        ctx.flag_mapq_group_count(f, mq, mv, ref, rv, n, mask, value, qmin, 25, d)
        ctx.sync()
        return d.to_host()

    every = run(0, 0, -1)                      # all rows with a non-NULL mapq
    assert 0.9795 * n < every.sum() < 0.9805 * n   # 2 % NULL mapq
    mapped, unmapped = run(4, 0, -1), run(4, 4, -1)
    assert np.array_equal(mapped + unmapped, every)
    assert unmapped[:25].sum() == 0 and mapped[25] == 0   # reference is NULL exactly when the read is unmapped
    q30 = run(1284, 0, 30)
    lo = run(1284, 0, -1) - q30                # same flag predicate, mapq < 30
    assert np.all(lo >= 0) and abs(q30.sum() / n - 0.89 * 0.78 / 0.98 * 0.98) < 0.01


def test_config5_histogram_rows_sum_to_reads(ctx):
    n, L = 16_000_000, 100
    off, data = ctx.gen_c5(5, 0, n, L)
    d = ctx.zeros(np.int64, L * 256)
    ctx.qual_pos_hist(off, data, n, L, d)
    ctx.qual_pos_hist(off, data, n, L, d)  # accumulate a second batch
    ctx.sync()
    h = d.to_host().reshape(L, 256)
    assert np.all(h.sum(axis=1) == 2 * n)
    assert h[:, :33].sum() == 0 and h[:, 75:].sum() == 0   # bytes are 33 + [0, 41]
    mean_q = (h * np.arange(256)).sum(axis=1) / (2 * n) - 33
    assert np.all(np.diff(mean_q) < 0.2) and mean_q[0] > mean_q[-1] + 8   # quality declines along the read


def test_config5_one_billion_reads_in_one_chunked_launch(ctx):
    """BASELINE config 5 at its full size (1e9 reads of 100 bytes = 50 Arrow batches, 104 GB resident) through
    exon_hip_plan_launch_chunks: every position's row sums to the number of reads, the byte range is the generator's, and the
    fused launch equals the per-batch launches bit for bit (checked on all 50 batches)."""
    import ctypes as C
    n, L, batch = 1_000_000_000, 100, 20_000_000
    if ctx.info()["hbm_bytes"] < 200 * 2**30:
        pytest.skip("needs a 288 GB part: 104 GB of resident input")
    data = ctx.empty(np.uint8, n * L + 64)
    nb = n // batch
    off = ctx.empty(np.int32, n + nb + 4)
    chunks = []
    for k in range(nb):
        d_off, d_bytes = off.ptr + 4 * (k * batch + k), data.ptr + k * batch * L
        ctx._check(ctx.lib.exon_hip_gen_c5(ctx.h, None, 5, k * batch, (k + 1) * batch, L, C.c_void_p(d_off), C.c_void_p(d_bytes)))
        chunks.append(([(d_bytes, None, d_off)], batch))
    plan = ctx.plan_qual_pos_hist(L)
    fused = ctx.to_device(np.full(L * 256, 7, np.int64))          # OVERWRITE must not care what was there
    plan.launch_chunks(chunks, fused, overwrite=True)
    each = ctx.zeros(np.int64, L * 256)
    for cols, m in chunks:
        plan.launch(cols, m, each, overwrite=False)
    ctx.sync()
    h = fused.to_host().reshape(L, 256)
    assert np.array_equal(h, each.to_host().reshape(L, 256))
    assert np.all(h.sum(axis=1) == n)
    assert h[:, :33].sum() == 0 and h[:, 75:].sum() == 0
    plan.close()
    data.free(); off.free()


def test_high_cardinality_group_by_beyond_one_tier3_launch_properties(ctx):
    """K4 with 50 000 keys over 3e8 rows: more rows than one launch of the partitioned tier 3 covers (2^28), so the table runs
    as two launches.  Property: folding the 50 000 keys onto the 5 FILTER ids they were derived from reproduces the 5-key
    kernel's counts exactly (and its sums to 1e-12) -- registers, LDS tier and the compacted / partitioned tier all feed it --
    and every key is seen."""
    n, G = 300_000_000, 50_000
    af, av, q, qv, fid = ctx.gen_c4(4, 0, n)
    base_c, base_s = k4(ctx, (af, av, q, qv, fid), n, 0.01, ">")
    keys = fid.to_host().astype(np.int32)
    keys += (5 * (np.arange(n, dtype=np.int64) % 10_000)).astype(np.int32)  # key = filter id + 5 * (row mod 10 000)
    d_keys = ctx.to_device(keys)
    del keys
    c, s = k4(ctx, (af, av, q, qv, d_keys), n, 0.01, ">", G=G)

    def fold(v):
        return v.reshape(10_000, 5).sum(axis=0)
    assert np.array_equal(fold(c[:G]), base_c[:5]) and np.array_equal(fold(c[G:]), base_c[5:])
    assert np.allclose(fold(s), base_s, rtol=1e-12, atol=0)
    assert (c[G:] > 0).all()


def test_more_than_two_to_the_32_rows_in_one_launch(ctx):
    """Maximum sizes: a single launch over more rows than a 32-bit index can count (4.6e9 rows: 43 GB of config-3 columns, then
    56 GB of config-4 columns -- a fraction of the 288 GB of HBM these kernels are laid out for).  Tile, row and bitmap indices
    must be 64-bit end to end, in the kernels and in the counter-based generators: the whole-table launch has to equal the sum
    over eight file splits generated and counted separately (each well below 2^32 rows), and the per-group totals must add up
    to the row count."""
    import gc
    n = (1 << 32) + 300_000_007
    cuts = [n * k // 8 for k in range(9)]
    # K3: flag / mapq / reference
    f, mq, mv, ref, rv = ctx.gen_c3(3, 0, n)

    def k3(cols, rows, mask, value, qmin):
        d = ctx.zeros(np.int64, 26)
        ctx.flag_mapq_group_count(*cols, rows, mask, value, qmin, 25, d)
        ctx.sync()
        return d.to_host()

    whole = k3((f, mq, mv, ref, rv), n, 1284, 0, 30)
    every = k3((f, mq, mv, ref, rv), n, 0, 0, -1)
    assert 0.9795 * n < every.sum() < 0.9805 * n            # 2 % NULL mapq, over ALL 4.6e9 rows
    del f, mq, mv, ref, rv
    gc.collect()
    parts = np.zeros(26, np.int64)
    for k in range(8):
        cols = ctx.gen_c3(3, cuts[k], cuts[k + 1])
        parts += k3(cols, cuts[k + 1] - cuts[k], 1284, 0, 30)
        del cols
    assert np.array_equal(parts, whole)
    gc.collect()
    # K4: AF / qual / FILTER id
    cols = ctx.gen_c4(4, 0, n)
    gt_c, gt_s = k4(ctx, cols, n, 0.01, ">")
    all_c, _ = k4(ctx, cols, n, float("-inf"), ">=")
    assert 0.9895 * n < int(all_c[5:].sum()) < 0.9905 * n    # 1 % NULL AF
    del cols
    gc.collect()
    tot_c, tot_s = np.zeros(10, np.int64), np.zeros(5)
    for k in range(8):
        shard = ctx.gen_c4(4, cuts[k], cuts[k + 1])
        c, s = k4(ctx, shard, cuts[k + 1] - cuts[k], 0.01, ">")
        tot_c += c
        tot_s += s
        del shard
    assert np.array_equal(tot_c, gt_c) and np.allclose(tot_s, gt_s, rtol=1e-12, atol=0)


def test_more_than_two_to_the_32_rows_point_and_range_hits(ctx):
    """K2 and K6 over 2^32 + 3e8 rows in one launch (55 GB / 93 GB of columns): per-contig counts add up to the row count and
    equal the generator's contig boundaries; the range-hit count of the whole table equals the sum over eight splits."""
    import gc
    n = (1 << 32) + 300_000_007
    c, p = ctx.gen_c2(2, n)

    def k2(cid, a, b):
        d = ctx.zeros(np.int64, 1)
        ctx.region_count(c, p, n, cid, a, b, d)
        ctx.sync()
        return int(d.to_host()[0])

    per_contig = [k2(cid, 1, None) for cid in range(24)]
    assert sum(per_contig) == n and min(per_contig) > 0
    assert k2(6, 1, 49_999_999) + k2(6, 50_000_000, 100_000_000) + k2(6, 100_000_001, None) == per_contig[6]
    del c, p
    gc.collect()
    cuts = [n * k // 8 for k in range(9)]

    def k6(cols, rows):
        ref, rv, st, en, pv = cols
        d = ctx.zeros(np.int64, 1)
        ctx.overlap_count(ref, rv, st, pv, en, pv, rows, 3, 50_000_000, 60_000_000, d)
        ctx.sync()
        return int(d.to_host()[0])

    cols = ctx.gen_c6(6, 0, n)
    whole = k6(cols, n)
    del cols
    gc.collect()
    parts = 0
    for k in range(8):
        cols = ctx.gen_c6(6, cuts[k], cuts[k + 1])
        parts += k6(cols, cuts[k + 1] - cuts[k])
        del cols
    assert whole == parts and whole > 1_000_000
