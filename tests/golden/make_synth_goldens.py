#!/usr/bin/env python3
"""Builds tests/golden/synth/*.npz: ORACLE-INDEPENDENT golden vectors for the five fused query shapes (SURVEY 8c "golden
vectors to commit (ii)").  Nothing here imports oracle/ or exon_amd: inputs come from numpy's PCG64 generator (seed in the
file; a SHA-256 of every input array is stored so a test can tell "numpy generated something else" from "the kernel is
wrong"), expected values from pyarrow.compute (an independent Arrow implementation: filter, Kleene AND, group_by) and,
where pyarrow has no operator (IEEE totalOrder compares, per-position histograms), from numpy integer arithmetic.

    python tests/golden/make_synth_goldens.py         # rewrites the .npz files (they are committed)

tests/test_gpu_synth_goldens.py regenerates the inputs from the seeds, checks the hashes, runs the HIP kernels through the
C ABI and compares with the stored expectations -- the GPU box never needs the oracle for these."""
import hashlib
import os

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bitmap(valid):
    return np.concatenate([np.packbits(valid, bitorder="little"), np.zeros(64, np.uint8)])


def total_order_key(x64):
    b = np.asarray(x64, np.float64).view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    return np.where(neg, ~b, b | np.uint64(1 << 63))


# ---- input generators (shared with the test: it imports this module) ------------------------------------------------------
def gen_c2(seed, n):
    rng = np.random.default_rng(seed)
    chrom = np.sort(rng.integers(0, 24, n)).astype(np.int32)
    pos = rng.integers(1, 250_000_000, n).astype(np.int64)
    cvalid = rng.random(n) < 0.98
    pvalid = rng.random(n) < 0.97
    return dict(chrom=chrom, pos=pos, cvalid=cvalid, pvalid=pvalid)


def gen_c3(seed, n, n_refs):
    rng = np.random.default_rng(seed)
    flags = np.array([99, 147, 83, 163, 1123, 1171, 77, 141, 355, 65, 4, 1024], np.int32)
    flag = flags[rng.integers(0, len(flags), n)]
    mapq = rng.choice(np.array([0, 1, 15, 29, 30, 31, 59, 60, 254], np.uint8), n)
    mvalid = rng.random(n) < 0.97
    ref = rng.integers(0, n_refs, n).astype(np.int32)
    rvalid = (flag & 4) == 0
    fvalid = rng.random(n) < 0.99
    return dict(flag=flag, mapq=mapq, mvalid=mvalid, ref=ref, rvalid=rvalid, fvalid=fvalid)


def gen_c4(seed, n, n_groups, specials):
    rng = np.random.default_rng(seed)
    af = np.exp2(-rng.integers(1, 15, n) - rng.random(n)).astype(np.float32)
    if specials:  # NaN / +-0 / +-inf / denormals / the f32(0.01) trap
        sp = np.array([np.nan, -np.nan, 0.0, -0.0, np.inf, -np.inf, 0.01, 1e-45, -1e-45, 3.4e38], np.float32)
        idx = rng.random(n) < 0.05
        af[idx] = sp[rng.integers(0, len(sp), int(idx.sum()))]
    qual = (rng.integers(0, 80_000, n) / 8).astype(np.float32)  # eighths: every f64 partial sum is exact
    gid = np.minimum(rng.exponential(1.5, n).astype(np.int32), n_groups - 1)
    gid[rng.integers(0, n, 50)] = n_groups - 1
    avalid = rng.random(n) < 0.99
    qvalid = rng.random(n) < 0.97
    return dict(af=af, qual=qual, gid=gid, avalid=avalid, qvalid=qvalid)


def gen_c5(seed, n, lo, hi):
    """reads of lo..hi quality bytes (lo == hi: uniform length); bytes 33..126 with a few >= 128"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, n).astype(np.int64)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(lens)
    data = (33 + np.clip(np.round(rng.normal(30, 6, int(off[-1]))), 0, 93)).astype(np.uint8)
    hot = rng.integers(0, len(data), max(1, len(data) // 5000))
    data[hot] = rng.integers(128, 256, len(hot)).astype(np.uint8)
    return dict(off=off, data=data)


def gen_c6(seed, n):
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 25, n).astype(np.int32)
    start = rng.integers(1, 249_000_000, n).astype(np.int64)
    end = start + rng.integers(0, 20_000, n)
    rvalid = rng.random(n) < 0.98
    svalid = rvalid & (rng.random(n) < 0.995)
    evalid = svalid.copy()
    return dict(ref=ref, start=start, end=end, rvalid=rvalid, svalid=svalid, evalid=evalid)


# ---- expectations -----------------------------------------------------------------------------------------------------------
def expect_c2(d, region):
    cid, a, b = region
    t = pa.table({"chrom": pa.array(d["chrom"], mask=~d["cvalid"]), "pos": pa.array(d["pos"], mask=~d["pvalid"])})
    keep = pc.and_kleene(pc.and_kleene(pc.equal(t["chrom"], cid), pc.greater_equal(t["pos"], a)), pc.less_equal(t["pos"], b))
    return np.array([t.filter(keep).num_rows], np.int64)  # filter() keeps TRUE only, like FilterExec


def expect_c3(d, n_refs, mask, value, qmin):
    t = pa.table({"flag": pa.array(d["flag"], mask=~d["fvalid"]), "mapq": pa.array(d["mapq"], mask=~d["mvalid"]),
                  "ref": pa.array(d["ref"], mask=~d["rvalid"])})
    keep = pc.and_kleene(pc.equal(pc.bit_wise_and(t["flag"], mask), value), pc.greater_equal(pc.cast(t["mapq"], pa.int32()), qmin))
    g = t.filter(keep).group_by("ref").aggregate([([], "count_all")])
    out = np.zeros(n_refs + 1, np.int64)
    for r in g.to_pylist():
        out[n_refs if r["ref"] is None else r["ref"]] = r["count_all"]
    return out


def expect_c4(d, n_groups, op, thr, specials):
    G = n_groups
    if not specials:  # plain numbers: pyarrow's IEEE compare IS the answer
        t = pa.table({"af": pa.array(d["af"], mask=~d["avalid"]), "qual": pa.array(d["qual"], mask=~d["qvalid"]), "g": pa.array(d["gid"])})
        f = {">": pc.greater, ">=": pc.greater_equal, "<": pc.less, "<=": pc.less_equal, "=": pc.equal, "!=": pc.not_equal}[op]
        keep = f(pc.cast(t["af"], pa.float64()), pa.scalar(thr, pa.float64()))
        g = t.filter(keep).group_by("g").aggregate([("qual", "sum"), ("qual", "count"), ([], "count_all")])
        cn, cr, sm = np.zeros(G, np.int64), np.zeros(G, np.int64), np.zeros(G, np.float64)
        for r in g.to_pylist():
            cn[r["g"]], cr[r["g"]], sm[r["g"]] = r["qual_count"], r["count_all"], (r["qual_sum"] or 0.0)
        return cn, cr, sm
    # NaN / -0 rows: arrow-rs compares floats in IEEE totalOrder, pyarrow does not -> integer sort keys
    kx, kt = total_order_key(d["af"].astype(np.float64)), total_order_key(np.float64(thr))
    keep = {">": kx > kt, ">=": kx >= kt, "<": kx < kt, "<=": kx <= kt, "=": kx == kt, "!=": kx != kt}[op] & d["avalid"]
    cr = np.bincount(d["gid"][keep], minlength=G).astype(np.int64)
    m = keep & d["qvalid"]
    cn = np.bincount(d["gid"][m], minlength=G).astype(np.int64)
    sm = np.bincount(d["gid"][m], weights=d["qual"][m].astype(np.float64), minlength=G)
    return cn, cr, sm


def expect_c5(d, lmax):
    off, data = d["off"].astype(np.int64), d["data"]
    n = len(off) - 1
    lens = off[1:] - off[:-1]
    pos = np.arange(len(data), dtype=np.int64) - np.repeat(off[:-1], lens)
    assert n > 0 and pos.max() < lmax
    return np.bincount(pos * 256 + data, minlength=lmax * 256).astype(np.int64)


def expect_c6(d, region, strict):
    rid, a, b = region
    t = pa.table({"ref": pa.array(d["ref"], mask=~d["rvalid"]), "start": pa.array(d["start"], mask=~d["svalid"]),
                  "end": pa.array(d["end"], mask=~d["evalid"])})
    if strict:
        keep = pc.and_kleene(pc.and_kleene(pc.equal(t["ref"], rid), pc.greater(t["start"], a)), pc.less(t["end"], b))
    else:
        keep = pc.and_kleene(pc.and_kleene(pc.equal(t["ref"], rid), pc.less_equal(t["start"], b)), pc.greater_equal(t["end"], a))
    return np.array([t.filter(keep).num_rows], np.int64)


CASES = {
    "c2_region_count_1M": dict(kind="c2", seed=1002, n=1_000_000, region=(6, 50_000_000, 100_000_000)),
    "c2_region_count_10k": dict(kind="c2", seed=1003, n=10_007, region=(0, 1, 2**62)),
    "c3_flag_mapq_group_count_1M": dict(kind="c3", seed=1004, n=1_000_000, n_refs=25, mask=1284, value=0, qmin=30),
    "c3_flag_mapq_group_count_300_refs": dict(kind="c3", seed=1005, n=200_003, n_refs=300, mask=4, value=0, qmin=1),
    "c4_cmp_avg_by_group_1M": dict(kind="c4", seed=1006, n=1_000_000, n_groups=5, op=">", thr=0.01, specials=False),
    "c4_cmp_avg_by_group_le_40_groups": dict(kind="c4", seed=1007, n=300_001, n_groups=40, op="<=", thr=0.001, specials=False),
    "c4_cmp_avg_by_group_nan_zero_inf": dict(kind="c4", seed=1008, n=250_000, n_groups=3, op=">=", thr=-0.0, specials=True),
    "c4_cmp_avg_by_group_ne_nan": dict(kind="c4", seed=1009, n=100_000, n_groups=8, op="!=", thr=float("nan"), specials=True),
    "c5_qual_pos_hist_uniform_100": dict(kind="c5", seed=1010, n=200_000, lo=100, hi=100, lmax=100),
    "c5_qual_pos_hist_ragged": dict(kind="c5", seed=1011, n=150_000, lo=1, hi=151, lmax=160),
    "c6_overlap_count_1M": dict(kind="c6", seed=1012, n=1_000_000, region=(6, 50_000_000, 100_000_000), strict=False),
    "c6_within_count_300k": dict(kind="c6", seed=1013, n=300_000, region=(3, 10_000_000, 200_000_000), strict=True),
}


def build(name):
    """-> (inputs dict, expected dict) of a case"""
    c = CASES[name]
    if c["kind"] == "c2":
        d = gen_c2(c["seed"], c["n"])
        return d, dict(count=expect_c2(d, c["region"]))
    if c["kind"] == "c3":
        d = gen_c3(c["seed"], c["n"], c["n_refs"])
        return d, dict(counts=expect_c3(d, c["n_refs"], c["mask"], c["value"], c["qmin"]))
    if c["kind"] == "c4":
        d = gen_c4(c["seed"], c["n"], c["n_groups"], c["specials"])
        cn, cr, sm = expect_c4(d, c["n_groups"], c["op"], c["thr"], c["specials"])
        return d, dict(count_y=cn, count_rows=cr, sum_y=sm)
    if c["kind"] == "c5":
        d = gen_c5(c["seed"], c["n"], c["lo"], c["hi"])
        return d, dict(hist=expect_c5(d, c["lmax"]))
    d = gen_c6(c["seed"], c["n"])
    return d, dict(count=expect_c6(d, c["region"], c["strict"]))


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in CASES:
        d, exp = build(name)
        hashes = {"sha256_" + k: np.array(sha(v)) for k, v in d.items()}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **exp, **hashes)
        print(name, {k: (v.tolist() if v.size <= 8 else f"{v.size} values, sum {v.sum()}") for k, v in exp.items()})


if __name__ == "__main__":
    main()
