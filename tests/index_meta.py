"""Test infrastructure: the per-reference record counts htslib stores in the pseudo-bin 37450 of a tabix / BAI index (what
`samtools idxstats` / `bcftools index --stats` print): n_mapped and n_unmapped per reference, written by a third party for the
reference's fixtures.  Index layout: SAM specification section 5.2 (BAI), tabix format description (TBI)."""
import gzip
import struct

META_BIN = 37450


def _walk(b, o, n_ref):
    out = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", b, o)
        o += 4
        meta = None
        for _ in range(n_bin):
            bin_, n_chunk = struct.unpack_from("<Ii", b, o)
            o += 8
            if bin_ == META_BIN:
                _beg, _end, n_mapped, n_unmapped = struct.unpack_from("<QQQQ", b, o)
                meta = (n_mapped, n_unmapped)
            o += 16 * n_chunk
        n_intv, = struct.unpack_from("<i", b, o)
        o += 4 + 8 * n_intv
        out.append(meta)
    return out, o


def tabix_counts(path):
    """{contig: records} of a .tbi"""
    b = gzip.open(path, "rb").read()
    assert b[:4] == b"TBI\x01"
    n_ref, _fmt, _cs, _cb, _ce, _meta, _skip, l_nm = struct.unpack_from("<8i", b, 4)
    names = [x.decode() for x in b[36:36 + l_nm].split(b"\0")[:-1]]
    meta, _ = _walk(b, 36 + l_nm, n_ref)
    return {n: m[0] for n, m in zip(names, meta) if m}


def bai_counts(path):
    """([(mapped, unmapped) or None per reference], reads without coordinates) of a .bai"""
    b = open(path, "rb").read()
    assert b[:4] == b"BAI\x01"
    n_ref, = struct.unpack_from("<i", b, 4)
    meta, o = _walk(b, 8, n_ref)
    n_no_coor = struct.unpack_from("<Q", b, o)[0] if o + 8 <= len(b) else None
    return meta, n_no_coor


def csi_counts(path):
    """[records or None per contig, in header order] of a .csi (CSI v1: bcftools / htslib's index for BCF)"""
    b = gzip.open(path, "rb").read()
    assert b[:4] == b"CSI\x01"
    _min_shift, depth, l_aux = struct.unpack_from("<3i", b, 4)
    o = 16 + l_aux
    n_ref, = struct.unpack_from("<i", b, o)
    o += 4
    meta_bin = ((1 << ((depth + 1) * 3)) - 1) // 7 + 1
    out = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", b, o)
        o += 4
        meta = None
        for _ in range(n_bin):
            bin_, _loffset, n_chunk = struct.unpack_from("<IQi", b, o)
            o += 16
            if bin_ == meta_bin:
                meta = struct.unpack_from("<QQQQ", b, o)[2]
            o += 16 * n_chunk
        out.append(meta)
    return out
