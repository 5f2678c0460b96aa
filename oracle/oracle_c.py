"""ctypes front end of oracle/liboracle.so (the plain-C restatement; TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build_oracle(force=False):
    src = [os.path.join(_HERE, f) for f in ("exon_oracle.c", "exon_oracle.h")]
    if force or not os.path.exists(_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB


class Timing(C.Structure):
    _fields_ = [("seconds_materialize", C.c_double), ("seconds_exec", C.c_double), ("threads", C.c_int)]


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _names(names):
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    return arr


class Oracle:
    """Thin, typed wrapper.  All arrays are numpy, bitmaps are Arrow LSB-first uint8."""

    CMP = {">": 0, ">=": 1, "<": 2, "<=": 3, "=": 4, "!=": 5}

    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        L.orc_rnd.restype = C.c_uint64
        L.orc_rnd.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_c2_region_count.restype = C.c_int64
        L.orc_c2_contig_name.restype = C.c_char_p
        L.orc_c2_contig_len.restype = C.c_int64
        L.orc_c3_ref_name.restype = C.c_char_p
        L.orc_c4_filter_name.restype = C.c_char_p

    # ---- dictionaries -------------------------------------------------------------------
    def c2_contigs(self):
        return [self.lib.orc_c2_contig_name(i).decode() for i in range(self.lib.orc_c2_num_contigs())]

    def c2_contig_lens(self):
        return [self.lib.orc_c2_contig_len(i) for i in range(self.lib.orc_c2_num_contigs())]

    def c3_refs(self):
        return [self.lib.orc_c3_ref_name(i).decode() for i in range(self.lib.orc_c3_num_refs())]

    def c4_filters(self):
        return [self.lib.orc_c4_filter_name(i).decode() for i in range(self.lib.orc_c4_num_filters())]

    # ---- generators ---------------------------------------------------------------------
    def rnd(self, seed, col, i):
        return self.lib.orc_rnd(seed, col, i)

    def gen_c6(self, seed, lo, hi):
        n = hi - lo
        nb = (n + 7) // 8
        ref, rv = np.empty(n, np.int32), np.empty(max(nb, 1), np.uint8)
        start, end, pv = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(max(nb, 1), np.uint8)
        self.lib.orc_gen_c6(C.c_uint64(seed), C.c_int64(lo), C.c_int64(hi), _p(ref, C.c_int32), _p(rv, C.c_uint8),
                            _p(start, C.c_int64), _p(end, C.c_int64), _p(pv, C.c_uint8))
        return ref, rv, start, end, pv

    def gen_c2(self, seed, n_total, lo=0, hi=None):
        hi = n_total if hi is None else hi
        n = hi - lo
        chrom = np.empty(n, np.int32)
        pos = np.empty(n, np.int64)
        self.lib.orc_gen_c2(C.c_uint64(seed), C.c_int64(n_total), C.c_int64(lo), C.c_int64(hi),
                            _p(chrom, C.c_int32), _p(pos, C.c_int64))
        return chrom, pos

    def gen_c3(self, seed, lo, hi):
        assert lo % 8 == 0
        n = hi - lo
        flag = np.empty(n, np.int32)
        mapq = np.empty(n, np.uint8)
        mv = np.zeros((n + 7) // 8, np.uint8)
        ref = np.empty(n, np.int32)
        rv = np.zeros((n + 7) // 8, np.uint8)
        self.lib.orc_gen_c3(C.c_uint64(seed), C.c_int64(lo), C.c_int64(hi), _p(flag, C.c_int32),
                            _p(mapq, C.c_uint8), _p(mv, C.c_uint8), _p(ref, C.c_int32), _p(rv, C.c_uint8))
        return flag, mapq, mv, ref, rv

    def gen_c4(self, seed, lo, hi):
        assert lo % 8 == 0
        n = hi - lo
        af = np.empty(n, np.float32)
        av = np.zeros((n + 7) // 8, np.uint8)
        q = np.empty(n, np.float32)
        qv = np.zeros((n + 7) // 8, np.uint8)
        fid = np.empty(n, np.int32)
        self.lib.orc_gen_c4(C.c_uint64(seed), C.c_int64(lo), C.c_int64(hi), _p(af, C.c_float),
                            _p(av, C.c_uint8), _p(q, C.c_float), _p(qv, C.c_uint8), _p(fid, C.c_int32))
        return af, av, q, qv, fid

    def gen_c5(self, seed, lo, hi, read_len):
        n = hi - lo
        off = np.empty(n + 1, np.int32)
        by = np.empty(n * read_len, np.uint8)
        self.lib.orc_gen_c5(C.c_uint64(seed), C.c_int64(lo), C.c_int64(hi), C.c_int32(read_len),
                            _p(off, C.c_int32), _p(by, C.c_uint8))
        return off, by

    # ---- grammar / UDFs -----------------------------------------------------------------
    def parse_region(self, s):
        name = C.create_string_buffer(256)
        a, b = C.c_int64(), C.c_int64()
        rc = self.lib.orc_parse_region(s.encode(), name, 256, C.byref(a), C.byref(b))
        if rc:
            raise ValueError(f"invalid region {s!r}")
        return name.value.decode(), a.value, (None if b.value == 2**63 - 1 else b.value)

    def region_match(self, chrom, pos, region):
        r = self.lib.orc_region_match(None if chrom is None else chrom.encode(), pos is not None,
                                      C.c_int64(pos or 0), region.encode())
        if r < 0:
            raise ValueError("region_match error")
        return bool(r)

    def interval_match(self, pos, interval):
        r = self.lib.orc_interval_match(pos is not None, C.c_int64(pos or 0), interval.encode())
        if r < 0:
            raise ValueError("interval_match error")
        return bool(r)

    def chrom_match(self, chrom, name):
        r = self.lib.orc_chrom_match(None if chrom is None else chrom.encode(), name.encode())
        if r < 0:
            raise ValueError("chrom_match error")
        return bool(r)

    def sam_flag(self, flag, bit):
        return bool(self.lib.orc_sam_flag(C.c_int32(flag), C.c_uint16(bit)))

    def quality_scores_to_list(self, s):
        out = (C.c_int32 * max(1, len(s.encode())))()
        n = self.lib.orc_quality_scores_to_list(s.encode(), out, len(out))
        return list(out[:n])

    def bam_intersects(self, ref_id, start, end, region_ref_id, rstart, rend):
        return bool(self.lib.orc_bam_intersects(
            ref_id is not None, C.c_int32(ref_id if ref_id is not None else -1),
            start is not None, C.c_int64(start or 0), end is not None, C.c_int64(end or 0),
            C.c_int32(region_ref_id), C.c_int64(rstart), C.c_int64(rend if rend is not None else 2**63 - 1)))

    def c6_overlap_count(self, ref_id, ref_valid, start, start_valid, end, end_valid, ref_names, region):
        """Interval hit, range form, over whole columns: numpy restatement of SemiLazyRecord::intersects
        (exon-bam/src/indexed_async_batch_stream.rs:66-87) on the reference layout -- the `reference` column is compared
        as a STRING with the region name (the reference materialises it from the header, array_builder.rs:118-127);
        a missing reference / start / end never matches.  Pinned on the fixture through orc_bam_intersects
        (tests/test_oracle_pins.py).  Validity arguments are Arrow LSB-first bitmaps or None."""
        name, a, b = self.parse_region(region)
        n = len(ref_id)

        def bits(bm):
            return np.ones(n, bool) if bm is None else np.unpackbits(np.asarray(bm, np.uint8), bitorder="little")[:n].astype(bool)

        rv, sv, ev = bits(ref_valid), bits(start_valid), bits(end_valid)
        names = np.array(list(ref_names) + [""], dtype=object)
        ref_str = names[np.where(rv, np.asarray(ref_id), len(ref_names))]
        hi = np.iinfo(np.int64).max if b is None else b
        hit = rv & sv & ev & (ref_str == name) & (np.asarray(start) <= hi) & (np.asarray(end) >= a)
        return int(hit.sum())

    @staticmethod
    def start_end_interval_from_expr(column, op, value):
        """StartEndIntervalPhysicalExpr::try_from(BinaryExpr) (exon-core/src/physical_plan/start_end_interval_physical_expr.rs:
        93-139): `start > v` -> (v, None); `end < v` -> (0, Some(v)); anything else is an error."""
        if op == ">":
            if column != "start":
                raise ValueError("Failed to parse interval: left name is not start")
            return int(value), None
        if op == "<":
            if column != "end":
                raise ValueError("Failed to parse interval: left name is not end")
            return 0, int(value)
        raise ValueError("Failed to parse interval: operator is not > or <")

    def c7_within_count(self, ref_id, ref_valid, start, start_valid, end, end_valid, ref_names, name, after=0, before=None):
        """COUNT(*) WHERE reference = name AND start > after AND "end" < before: the BED / GFF predicate, evaluated as its
        inner BinaryExprs are (same file :186-191), Kleene AND, keep TRUE.  Column-wise numpy (single-threaded)."""
        n = len(ref_id)

        def bits(bm):
            return np.ones(n, bool) if bm is None else np.unpackbits(np.asarray(bm, np.uint8), bitorder="little")[:n].astype(bool)
        if name not in ref_names:
            return 0
        keep = bits(ref_valid) & bits(start_valid) & bits(end_valid) & (np.asarray(ref_id) == list(ref_names).index(name))
        keep &= np.asarray(start, np.int64) > after
        if before is not None:
            keep &= np.asarray(end, np.int64) < before
        return int(keep.sum())

    def regroup_files_by_size(self, sizes, target):
        s = np.asarray(sizes, np.int64)
        g = np.zeros(len(s), np.int32)
        ng = self.lib.orc_regroup_files_by_size(_p(s, C.c_int64), len(s), target, _p(g, C.c_int))
        groups = [[] for _ in range(ng)]
        # within a group the reference pushes in ascending-size order
        order = sorted(range(len(s)), key=lambda i: (int(s[i]), i))
        for i in order:
            groups[g[i]].append(i)
        return groups

    # ---- plan restatements --------------------------------------------------------------
    def c2_region_count(self, chrom_id, pos, contigs, region, chrom_valid=None, pos_valid=None, threads=0):
        t = Timing()
        r = self.lib.orc_c2_region_count(_p(chrom_id, C.c_int32), _p(pos, C.c_int64),
                                         _p(chrom_valid, C.c_uint8), _p(pos_valid, C.c_uint8),
                                         C.c_int64(len(pos)), _names(contigs), len(contigs),
                                         region.encode(), threads, C.byref(t))
        if r < 0:
            raise RuntimeError(f"oracle c2 failed ({r})")
        return int(r), t

    def c3_flag_mapq_group_count(self, flag, mapq, mapq_valid, ref_id, ref_valid, refs, flag_mask,
                                 flag_value, mapq_min, threads=0):
        t = Timing()
        counts = np.zeros(len(refs) + 1, np.int64)
        r = self.lib.orc_c3_flag_mapq_group_count(
            _p(flag, C.c_int32), _p(mapq, C.c_uint8), _p(mapq_valid, C.c_uint8), _p(ref_id, C.c_int32),
            _p(ref_valid, C.c_uint8), C.c_int64(len(flag)), _names(refs), len(refs), flag_mask,
            flag_value, mapq_min, threads, _p(counts, C.c_int64), C.byref(t))
        if r < 0:
            raise RuntimeError(f"oracle c3 failed ({r})")
        return counts, t

    def c4_cmp_avg_by_group(self, af, af_valid, qual, qual_valid, filter_id, filters, thr, op=">", threads=0):
        t = Timing()
        G = len(filters)
        s = np.zeros(G, np.float64)
        cn = np.zeros(G, np.int64)
        cr = np.zeros(G, np.int64)
        r = self.lib.orc_c4_cmp_avg_by_group(
            _p(af, C.c_float), _p(af_valid, C.c_uint8), _p(qual, C.c_float), _p(qual_valid, C.c_uint8),
            _p(filter_id, C.c_int32), C.c_int64(len(af)), _names(filters), G, C.c_double(thr),
            self.CMP[op], threads, _p(s, C.c_double), _p(cn, C.c_int64), _p(cr, C.c_int64), C.byref(t))
        if r < 0:
            raise RuntimeError(f"oracle c4 failed ({r})")
        return s, cn, cr, t

    def c5_qual_pos_hist(self, offsets, data, lmax, threads=0):
        t = Timing()
        h = np.zeros((lmax, 256), np.int64)
        r = self.lib.orc_c5_qual_pos_hist(_p(offsets, C.c_int32), _p(data, C.c_uint8),
                                          C.c_int64(len(offsets) - 1), lmax, threads,
                                          _p(h, C.c_int64), C.byref(t))
        if r < 0:
            raise RuntimeError(f"oracle c5 failed ({r})")
        return h, t
