"""What the ORACLE says a file-to-aggregate query must return: oracle/decode.py turns the file into the reference's columns,
oracle/exon_oracle.c runs the reference's filter + aggregate on them.  The GPU decode tests compare the device path with
this directly (the product's own host decoders are only a second opinion there).  Test infrastructure."""
import numpy as np

from oracle import decode


def _bits(flags):
    return np.packbits(np.asarray(flags, bool), bitorder="little")


def vcf_columns(path, fmt="vcf", info_field=None):
    """chrom / pos / qual / filter (';'-joined, '' = []) / info.<field> as python lists, + the header contigs."""
    v = decode.decode_bcf(path) if fmt == "bcf" else decode.decode_vcf(path)
    out = {"chrom": v["chrom"], "pos": v["pos"], "qual": [None if q is None else float(np.float32(q)) for q in v["qual"]],
           "filter": [";".join(f) for f in v["filter"]], "contigs": v["contigs"]}
    if info_field:
        vals = []
        for i in v["info"]:
            x = None if i is None else i.get(info_field)
            vals.append(None if x is None or x is True or x == "." else float(np.float32(x)))
        out["info"] = vals
    return out


def k4_expected(orc, path, fmt, info_field, thr=0.01, op=">"):
    """{filter text: (COUNT(qual), COUNT(*), SUM(qual))} + row count: WHERE info.<field> <op> thr GROUP BY filter."""
    c = vcf_columns(path, fmt, info_field)
    n = len(c["chrom"])
    names = sorted(set(c["filter"]))
    idx = {f: i for i, f in enumerate(names)}
    af = np.array([0.0 if x is None else x for x in c["info"]], np.float32)
    q = np.array([0.0 if x is None else x for x in c["qual"]], np.float32)
    fid = np.array([idx[f] for f in c["filter"]], np.int32)
    av, qv = _bits([x is not None for x in c["info"]]), _bits([x is not None for x in c["qual"]])
    pad = np.zeros(64, np.uint8)
    s, cn, cr, _ = orc.c4_cmp_avg_by_group(af, np.concatenate([av, pad]), q, np.concatenate([qv, pad]), fid, names, thr, op)
    return n, {names[g]: (int(cn[g]), int(cr[g]), float(s[g])) for g in range(len(names)) if cr[g]}


def region_count_expected(path, fmt, chrom, start=1, end=None):
    c = vcf_columns(path, fmt)
    return len(c["chrom"]), sum(1 for ch, p in zip(c["chrom"], c["pos"]) if ch == chrom and p is not None and p >= start and (end is None or p <= end))


def bam_columns(path, fmt="bam"):
    refs, recs = decode.decode_sam(path) if fmt == "sam" else decode.decode_cram(path) if fmt == "cram" else decode.decode_bam(path)
    names = [r[0] for r in refs]
    return names, {"flag": [r["flag"] for r in recs], "mapq": [r["mapq"] for r in recs],
                   "ref": [None if r["ref_id"] is None else names[r["ref_id"]] for r in recs],
                   "start": [r["start"] for r in recs], "end": [r["end"] for r in recs]}


def k3_expected(orc, path, fmt="bam", mask=1284, value=0, qmin=30):
    """(rows, COUNT(*) per reference + the NULL group): WHERE flag & mask = value AND mapq >= qmin GROUP BY reference."""
    names, c = bam_columns(path, fmt)
    n = len(c["flag"])
    pad = np.zeros(64, np.uint8)
    flag = np.array(c["flag"], np.int32)
    mapq = np.array([255 if m is None else m for m in c["mapq"]], np.uint8)
    ref = np.array([-1 if r is None else names.index(r) for r in c["ref"]], np.int32)
    mv, rv = _bits([m is not None for m in c["mapq"]]), _bits([r is not None for r in c["ref"]])
    cnt, _ = orc.c3_flag_mapq_group_count(flag, np.concatenate([mapq, pad]), np.concatenate([mv, pad]), ref, np.concatenate([rv, pad]), names,
                                          mask, value, qmin)
    return n, np.asarray(cnt)


def k6_expected(path, fmt, ref, a, b):
    """SemiLazyRecord::intersects over the oracle's columns (python; b = None: open end)."""
    _, c = bam_columns(path, fmt)
    hit = 0
    for r, s, e in zip(c["ref"], c["start"], c["end"]):
        if r == ref and s is not None and e is not None and (b is None or s <= b) and e >= a:
            hit += 1
    return len(c["flag"]), hit
