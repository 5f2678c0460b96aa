"""Randomised differential test of K2 / K3 / K4 / K6 against plain numpy statements of the queries (no oracle): sizes around the
tile borders, columns that start in the middle of their allocations (an Arrow slice: 64 rows in, so values stay 16-byte aligned
and bitmaps start on a byte), validity bitmaps absent / sparse / dense, group counts on every tier border, two launches that
accumulate.  The K5 twin is tests/test_gpu_k5_fuzz.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 8191, 8192, 8193, 16384, 16385, 40_000, 100_003, 262_144, 300_001]


def _bits(ok):
    return np.packbits(np.asarray(ok, np.uint8), bitorder="little")


class Col:
    """values (+ validity) uploaded with `lead` junk rows in front; .ptr / .vptr address row `lead`"""

    def __init__(self, ctx, values, ok, lead, rng):
        junk = rng.integers(0, 100, lead).astype(values.dtype)
        self.buf = ctx.to_device(np.concatenate([junk, values, np.zeros(64, values.dtype)]))
        self.ptr = self.buf.ptr + lead * values.dtype.itemsize
        self.vbuf = None
        self.vptr = None
        if ok is not None:
            self.vbuf = ctx.to_device(np.concatenate([_bits(rng.integers(0, 2, lead)), _bits(ok), np.zeros(64, np.uint8)]))
            self.vptr = self.vbuf.ptr + lead // 8


def _validity(rng, n):
    kind = rng.integers(0, 4)
    if kind == 0:
        return None
    p = (0.02, 0.5, 0.98)[kind - 1]
    return rng.random(n) < p


def _cases(seed, k):
    rng = np.random.default_rng(seed)
    for i in range(k):
        n = int(SIZES[i % len(SIZES)] if i < 2 * len(SIZES) else rng.integers(1, 400_000))
        yield rng, n, int(rng.choice([0, 64, 128, 1024]))


def test_k2_region_count(ctx):
    for rng, n, lead in _cases(101, 60):
        chrom = rng.integers(0, 6, n).astype(np.int32)
        pos = rng.integers(1, 1000, n).astype(np.int64)
        cok, pok = _validity(rng, n), _validity(rng, n)
        c, p = Col(ctx, chrom, cok, lead, rng), Col(ctx, pos, pok, lead, rng)
        cid, a = int(rng.integers(0, 6)), int(rng.integers(1, 600))
        b = None if rng.integers(0, 4) == 0 else int(a + rng.integers(0, 500))
        d = ctx.zeros(np.int64, 1)
        ctx.region_count(c.ptr, p.ptr, n, cid, a, b, d, chrom_valid=c.vptr, pos_valid=p.vptr)
        ctx.region_count(c.ptr, p.ptr, n, cid, a, b, d, chrom_valid=c.vptr, pos_valid=p.vptr)  # accumulates
        ctx.sync()
        hit = (chrom == cid) & (pos >= a) & (True if b is None else pos <= b)
        if cok is not None:
            hit &= cok
        if pok is not None:
            hit &= pok
        assert int(d.to_host()[0]) == 2 * int(hit.sum()), (n, lead, cid, a, b)


def test_k6_overlap_count(ctx):
    for rng, n, lead in _cases(103, 60):
        ref = rng.integers(0, 5, n).astype(np.int32)
        start = rng.integers(1, 5000, n).astype(np.int64)
        end = start + rng.integers(0, 300, n)
        rok, sok, eok = _validity(rng, n), _validity(rng, n), _validity(rng, n)
        r, s, e = Col(ctx, ref, rok, lead, rng), Col(ctx, start, sok, lead, rng), Col(ctx, end, eok, lead, rng)
        rid, a = int(rng.integers(0, 5)), int(rng.integers(1, 4000))
        b = None if rng.integers(0, 4) == 0 else int(a + rng.integers(0, 2000))
        d = ctx.zeros(np.int64, 1)
        ctx.overlap_count(r.ptr, r.vptr, s.ptr, s.vptr, e.ptr, e.vptr, n, rid, a, b, d)
        ctx.sync()
        hit = (ref == rid) & (end >= a) & (True if b is None else start <= b)
        for ok in (rok, sok, eok):
            if ok is not None:
                hit &= ok
        assert int(d.to_host()[0]) == int(hit.sum()), (n, lead, rid, a, b)


def test_k3_flag_mapq_group_count(ctx):
    for rng, n, lead in _cases(107, 60):
        R = int(rng.choice([1, 2, 25, 300, 4094, 4095]))
        flag = rng.choice(np.array([0, 4, 16, 99, 147, 256, 1024, 1028, 1284, 2048], np.int32), n)
        mapq = rng.integers(0, 61, n).astype(np.uint8)
        ref = rng.integers(0, R, n).astype(np.int32)
        if rng.integers(0, 3) == 0:
            ref = np.sort(ref)  # coordinate-sorted input: long runs of one reference
        mok, rok = _validity(rng, n), _validity(rng, n)
        f, m, r = Col(ctx, flag, None, lead, rng), Col(ctx, mapq, mok, lead, rng), Col(ctx, ref, rok, lead, rng)
        mask, value, qmin = int(rng.choice([0, 4, 1284, 1028])), 0, int(rng.choice([-1, 0, 30, 60]))
        d = ctx.zeros(np.int64, R + 1)
        ctx.flag_mapq_group_count(f.ptr, m.ptr, m.vptr, r.ptr, r.vptr, n, mask, value, qmin, R, d)
        ctx.sync()
        keep = ((flag & mask) == value) & (mapq.astype(np.int32) >= qmin)
        if mok is not None:
            keep &= mok
        gid = np.where(rok, ref, R) if rok is not None else ref  # NULL reference: the last group
        want = np.bincount(gid[keep], minlength=R + 1).astype(np.int64)
        assert np.array_equal(d.to_host(), want), (n, lead, R, mask, qmin)


@pytest.mark.parametrize("op", [">", ">=", "<", "<=", "==", "!="])
def test_k4_cmp_avg_by_group(ctx, op):
    fn = {">": np.greater, ">=": np.greater_equal, "<": np.less, "<=": np.less_equal, "==": np.equal, "!=": np.not_equal}[op]
    for rng, n, lead in _cases(109 + len(op) + ord(op[0]), 30):
        G = int(rng.choice([1, 5, 8, 9, 64, 4100, 4101, 5000]))
        x = rng.choice(np.array([0.0, 0.01, 0.25, 0.5, 1.0, -1.0], np.float32), n)
        y = (rng.integers(0, 8000, n) / 8.0).astype(np.float32)  # eighths: f64 sums are exact whatever the order
        gid = rng.integers(0, G, n).astype(np.int32)
        if rng.integers(0, 3) == 0:
            gid = np.sort(gid)
        xok, yok = _validity(rng, n), _validity(rng, n)
        cx, cy, cg = Col(ctx, x, xok, lead, rng), Col(ctx, y, yok, lead, rng), Col(ctx, gid, None, lead, rng)
        thr = float(rng.choice([0.0, 0.01, 0.25, 0.5]))
        dc, ds = ctx.zeros(np.int64, 2 * G), ctx.zeros(np.float64, G)
        ctx.cmp_avg_by_group(cx.ptr, cx.vptr, cy.ptr, cy.vptr, cg.ptr, n, thr, op, G, dc, ds)
        ctx.sync()
        keep = fn(x.astype(np.float64), thr)  # a Float32 column against a Float64 literal: DataFusion widens the COLUMN (0.01f < 0.01)
        if xok is not None:
            keep &= xok
        yv = keep if yok is None else keep & yok
        crow = np.bincount(gid[keep], minlength=G)
        cnn = np.bincount(gid[yv], minlength=G)
        sums = np.bincount(gid[yv], weights=y[yv].astype(np.float64), minlength=G)
        c = dc.to_host()
        assert np.array_equal(c[:G], cnn) and np.array_equal(c[G:], crow), (n, lead, G, op, thr)
        assert np.array_equal(ds.to_host(), sums), (n, lead, G, op, thr)
