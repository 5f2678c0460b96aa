"""Python host side above the C ABI (test harness / bench driver; the product is the HIP library).

`Context` owns a device; `DeviceBuffer` is a typed HBM allocation; `Plan` / `Stream` mirror the
ExecutionPlan-shaped surface of include/exon_hip.h (plan_create / stream_open / push / finish).
Every compute call goes through libexon_hip.so -- there is no numpy or oracle fallback.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import CMP, Column, ExonHipError


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A typed allocation in HBM (hipMalloc through the C ABI)."""

    def __init__(self, ctx, dtype, n):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self.nbytes = self.n * self.dtype.itemsize
        p = C.c_void_p()
        ctx._check(ctx.lib.exon_hip_malloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    def copy_from(self, host, stream=None):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size <= self.n
        self.ctx._check(self.ctx.lib.exon_hip_memcpy_h2d(self.ctx.h, self.ptr, _np_ptr(host), host.nbytes, stream))
        self.ctx.sync(stream)
        return self

    def to_host(self, n=None, stream=None):
        n = self.n if n is None else n
        out = np.empty(n, self.dtype)
        self.ctx._check(self.ctx.lib.exon_hip_memcpy_d2h(self.ctx.h, _np_ptr(out), self.ptr, out.nbytes, stream))
        return out

    def zero(self, stream=None):
        self.ctx._check(self.ctx.lib.exon_hip_memset(self.ctx.h, self.ptr, 0, self.nbytes, stream))
        return self

    def free(self):
        if self.ptr:
            self.ctx.lib.exon_hip_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _col(values=None, validity=None, offsets=None, length=0):
    def p(x):
        if x is None:
            return None
        return x.ptr if isinstance(x, DeviceBuffer) else int(x)
    return Column(p(values), p(validity), p(offsets), int(length))


class Context:
    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.exon_hip_ctx_create(device, C.byref(h))
        if rc:
            raise ExonHipError(rc, self.lib.exon_hip_last_error(None).decode(errors="replace"))
        self.h = h
        self.device = device

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise ExonHipError(rc, self.lib.exon_hip_last_error(self.h).decode(errors="replace"))
        return rc

    def info(self):
        d = L.DeviceInfo()
        self._check(self.lib.exon_hip_ctx_info(self.h, C.byref(d)))
        return {"name": d.name.decode(), "gcn_arch": d.gcn_arch.decode(), "compute_units": d.compute_units,
                "wavefront_size": d.wavefront_size, "hbm_bytes": d.hbm_bytes, "clock_khz": d.clock_khz}

    def empty(self, dtype, n):
        return DeviceBuffer(self, dtype, n)

    def zeros(self, dtype, n):
        return DeviceBuffer(self, dtype, n).zero()

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceBuffer(self, host.dtype, host.size).copy_from(host)

    def sync(self, stream=None):
        self._check(self.lib.exon_hip_sync(self.h, stream))

    def read_probe(self, ptrs, bytes_each, reps=20, stream=None):
        """exon_hip_read_probe: (GB/s, ms per pass) of a bare streaming read over 1..4 device buffers in lock-step."""
        arr = (C.c_void_p * len(ptrs))(*[int(p) for p in ptrs])
        ms, nbytes = C.c_double(), C.c_int64()
        self._check(self.lib.exon_hip_read_probe(self.h, stream, arr, len(ptrs), int(bytes_each), int(reps), C.byref(ms), C.byref(nbytes)))
        return nbytes.value / (ms.value * 1e-3) / 1e9, ms.value

    def gzip_inflate(self, raw, slab_bytes=None, out_cap=None, scratch_bytes=0, return_stats=False):
        """A whole plain-gzip file through exon_hip_gzip_stream_*, `slab_bytes` of compressed input per call (the unused tail of
        a slab is passed again in front of the next bytes, as a file pipeline does): (bytes, stats)."""
        raw = bytes(raw)
        slab = int(slab_bytes or max(len(raw), 1))
        cap = int(out_cap or max(4 << 20, 40 * slab))
        h = C.c_void_p()
        self._check(self.lib.exon_hip_gzip_stream_create(self.h, slab, int(scratch_bytes), C.byref(h)))
        d_comp = DeviceBuffer(self, np.uint8, slab + 8192)
        d_out = DeviceBuffer(self, np.uint8, cap)
        out = []
        try:
            pos, ended = 0, False
            while not ended:
                n = min(slab, len(raw) - pos)
                final = pos + n == len(raw)
                buf = np.zeros(n + 4096, np.uint8)
                buf[:n] = np.frombuffer(raw, np.uint8, n, pos)
                d_comp.copy_from(buf)
                consumed, produced, end = C.c_int64(), C.c_int64(), C.c_int32()
                self._check(self.lib.exon_hip_gzip_stream_decode(h, None, d_comp.ptr, n, int(final), d_out.ptr, cap, C.byref(consumed),
                                                                 C.byref(produced), C.byref(end)))
                if produced.value:
                    out.append(d_out.to_host(produced.value).tobytes())
                ended = bool(end.value)
                if not ended and consumed.value == 0 and produced.value == 0:
                    raise ExonHipError(-1, "gzip_inflate: no progress")
                pos += consumed.value
                if final and not ended and consumed.value == 0:
                    raise ExonHipError(-1, "gzip_inflate: input ended without the stream's end")
            st = L.GzipStats()
            self._check(self.lib.exon_hip_gzip_stream_get_stats(h, C.byref(st)))
        finally:
            self.lib.exon_hip_gzip_stream_destroy(h)
            d_comp.free()
            d_out.free()
        data = b"".join(out)
        return (data, {n: getattr(st, n) for n, _ in L.GzipStats._fields_}) if return_stats else data

    def timer_start(self, stream=None):
        self._check(self.lib.exon_hip_timer_start(self.h, stream))

    def timer_stop_ms(self, stream=None):
        ms = C.c_float()
        self._check(self.lib.exon_hip_timer_stop_ms(self.h, stream, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            self.lib.exon_hip_ctx_destroy(self.h)
            self.h = None

    # -- operators (asynchronous; accumulate into device state) ---------------------------------
    def region_count(self, chrom_id, pos, n, region_chrom_id, start, end, d_count, chrom_valid=None,
                     pos_valid=None, stream=None):
        end = L.REGION_OPEN_END if end is None else end
        c0, c1 = _col(chrom_id, chrom_valid, None, n), _col(pos, pos_valid, None, n)
        self._check(self.lib.exon_hip_region_count(self.h, stream, C.byref(c0), C.byref(c1), n, region_chrom_id,
                                                   start, end, d_count.ptr))

    def flag_mapq_group_count(self, flag, mapq, mapq_valid, ref_id, ref_valid, n, flag_mask, flag_value, mapq_min,
                              n_refs, d_counts, flag_valid=None, stream=None):
        c0, c1, c2 = _col(flag, flag_valid, None, n), _col(mapq, mapq_valid, None, n), _col(ref_id, ref_valid, None, n)
        self._check(self.lib.exon_hip_flag_mapq_group_count(self.h, stream, C.byref(c0), C.byref(c1), C.byref(c2), n,
                                                            flag_mask, flag_value, mapq_min, n_refs, d_counts.ptr))

    def cmp_avg_by_group(self, x, x_valid, y, y_valid, group_id, n, threshold, op, n_groups, d_counts, d_sums,
                         stream=None):
        c0, c1, c2 = _col(x, x_valid, None, n), _col(y, y_valid, None, n), _col(group_id, None, None, n)
        self._check(self.lib.exon_hip_cmp_avg_by_group(self.h, stream, C.byref(c0), C.byref(c1), C.byref(c2), n,
                                                       float(threshold), CMP[op], n_groups, d_counts.ptr, d_sums.ptr))

    def overlap_count(self, ref_id, ref_valid, start, start_valid, end, end_valid, n, region_ref_id, region_start, region_end,
                      d_count, stream=None):
        """K6: rows whose [start, end] on reference `region_ref_id` overlaps [region_start, region_end] (all three valid)."""
        c0, c1, c2 = _col(ref_id, ref_valid, None, n), _col(start, start_valid, None, n), _col(end, end_valid, None, n)
        end_v = L.REGION_OPEN_END if region_end is None else region_end
        self._check(self.lib.exon_hip_overlap_count(self.h, stream, C.byref(c0), C.byref(c1), C.byref(c2), n, region_ref_id,
                                                    region_start, end_v, d_count.ptr))

    def within_count(self, ref_id, ref_valid, start, start_valid, end, end_valid, n, region_ref_id, after, before, d_count,
                     stream=None):
        """K6 strict form: rows with start > after AND end < before on reference `region_ref_id` (BED / GFF predicate)."""
        c0, c1, c2 = _col(ref_id, ref_valid, None, n), _col(start, start_valid, None, n), _col(end, end_valid, None, n)
        self._check(self.lib.exon_hip_within_count(self.h, stream, C.byref(c0), C.byref(c1), C.byref(c2), n, region_ref_id,
                                                   after, L.REGION_OPEN_END if before is None else before, d_count.ptr))

    def qual_pos_hist(self, offsets, data, n_reads, lmax, d_hist, stream=None):
        c0 = _col(data, None, offsets, n_reads)
        self._check(self.lib.exon_hip_qual_pos_hist(self.h, stream, C.byref(c0), n_reads, lmax, d_hist.ptr))

    def qual_pos_hist_chunks(self, chunks, lmax, d_hist, stream=None):
        """chunks = [(offsets, data, n_reads)]: one Utf8 batch each; fused launches of up to 64 chunks."""
        arr = (Column * max(1, len(chunks)))(*[_col(d, None, o, n) for (o, d, n) in chunks])
        ns = (C.c_int64 * max(1, len(chunks)))(*[n for _, _, n in chunks])
        self._check(self.lib.exon_hip_qual_pos_hist_chunks(self.h, stream, arr, len(chunks), ns, lmax, d_hist.ptr))

    def qual_pos_hist_views(self, d_text, starts, ends, n_reads, lmax, d_hist, stream=None):
        """K5 over [starts[r], ends[r]) views into `d_text` (device pointers / DeviceBuffers)."""
        ptr = lambda x: x.ptr if isinstance(x, DeviceBuffer) else x  # noqa: E731
        self._check(self.lib.exon_hip_qual_pos_hist_views(self.h, stream, ptr(d_text), ptr(starts), ptr(ends), n_reads, lmax,
                                                          d_hist.ptr))

    def bgzf_inflate(self, data, verify_crc=True, stream=None):
        """Test / tool helper: BGZF bytes -> inflated bytes through the GPU (exon_hip_bgzf_inflate).  Returns
        (inflated uint8 array, seconds spent in the device call)."""
        import time
        blocks, n, consumed, out_bytes = bgzf_scan(data)
        comp = np.frombuffer(data, np.uint8)[:consumed]
        d_comp = self.to_device(np.concatenate([comp, np.zeros(4096 + (-len(comp)) % 4, np.uint8)]))
        d_out = self.empty(np.uint8, out_bytes + 64)
        bad = C.c_int32(-1)
        t = time.perf_counter()
        self._check(self.lib.exon_hip_bgzf_inflate(self.h, stream, d_comp.ptr, blocks, n, d_out.ptr, 1 if verify_crc else 0,
                                                   C.byref(bad)))
        dt = time.perf_counter() - t
        return d_out.to_host()[:out_bytes], dt

    # -- synthetic inputs in HBM ------------------------------------------------------------------
    def gen_c2(self, seed, n_total, lo=0, hi=None, stream=None):
        hi = n_total if hi is None else hi
        n = hi - lo
        chrom, pos = self.empty(np.int32, n), self.empty(np.int64, n)
        self._check(self.lib.exon_hip_gen_c2(self.h, stream, seed, n_total, lo, hi, chrom.ptr, pos.ptr))
        return chrom, pos

    def gen_c6(self, seed, lo, hi, stream=None):
        n = hi - lo
        nb = (n + 7) // 8 + 64
        ref, rv = self.empty(np.int32, n), self.zeros(np.uint8, nb)
        start, end, pv = self.empty(np.int64, n), self.empty(np.int64, n), self.zeros(np.uint8, nb)
        self._check(self.lib.exon_hip_gen_c6(self.h, stream, seed, lo, hi, ref.ptr, rv.ptr, start.ptr, end.ptr, pv.ptr))
        return ref, rv, start, end, pv

    def gen_c3(self, seed, lo, hi, stream=None):
        n = hi - lo
        nb = (n + 7) // 8 + 64
        flag, mapq, mv = self.empty(np.int32, n), self.empty(np.uint8, n + 64), self.zeros(np.uint8, nb)
        ref, rv = self.empty(np.int32, n), self.zeros(np.uint8, nb)
        self._check(self.lib.exon_hip_gen_c3(self.h, stream, seed, lo, hi, flag.ptr, mapq.ptr, mv.ptr, ref.ptr, rv.ptr))
        return flag, mapq, mv, ref, rv

    def gen_c4(self, seed, lo, hi, stream=None):
        n = hi - lo
        nb = (n + 7) // 8 + 64
        af, av = self.empty(np.float32, n), self.zeros(np.uint8, nb)
        q, qv = self.empty(np.float32, n), self.zeros(np.uint8, nb)
        fid = self.empty(np.int32, n)
        self._check(self.lib.exon_hip_gen_c4(self.h, stream, seed, lo, hi, af.ptr, av.ptr, q.ptr, qv.ptr, fid.ptr))
        return af, av, q, qv, fid

    def gen_c5(self, seed, lo, hi, read_len, stream=None):
        n = hi - lo
        off, data = self.empty(np.int32, n + 1), self.empty(np.uint8, n * read_len + 64)
        self._check(self.lib.exon_hip_gen_c5(self.h, stream, seed, lo, hi, read_len, off.ptr, data.ptr))
        return off, data

    # -- plans ------------------------------------------------------------------------------------
    def plan_region_count(self, region_chrom_id, start=1, end=None, columns=(0, 1)):
        d = L.PlanDesc(kind=L.PLAN_REGION_COUNT, region_chrom_id=region_chrom_id, region_start=start,
                       region_end=L.REGION_OPEN_END if end is None else end)
        return Plan(self, d, columns)

    def plan_overlap_count(self, region_ref_id, start=1, end=None, columns=(2, 3, 4)):
        """COUNT(*) of BAM-layout rows overlapping a region (scan columns 2 reference, 3 start, 4 end)."""
        d = L.PlanDesc(kind=L.PLAN_OVERLAP_COUNT, region_chrom_id=region_ref_id, region_start=start,
                       region_end=L.REGION_OPEN_END if end is None else end)
        return Plan(self, d, columns)

    def plan_within_count(self, region_ref_id, after=0, before=None, columns=(2, 3, 4)):
        """COUNT(*) of rows with reference = id AND start > after AND end < before (StartEndIntervalPhysicalExpr)."""
        d = L.PlanDesc(kind=L.PLAN_WITHIN_COUNT, region_chrom_id=region_ref_id, region_start=after,
                       region_end=L.REGION_OPEN_END if before is None else before)
        return Plan(self, d, columns)

    def plan_flag_mapq_group_count(self, flag_mask, flag_value, mapq_min, n_refs, columns=(0, 1, 2)):
        d = L.PlanDesc(kind=L.PLAN_FLAG_MAPQ_GROUP_COUNT, n_groups=n_refs, flag_mask=flag_mask,
                       flag_value=flag_value, mapq_min=mapq_min)
        return Plan(self, d, columns)

    def plan_cmp_avg_by_group(self, op, threshold, n_groups, columns=(0, 1, 2), x_type="f32", y_type="f32"):
        """x_type / y_type "i32": the compared column / AVG's argument holds Int32 values (an INFO field of Type=Integer);
        a plan fed by Stream.consume(scan) takes both from the file's header instead."""
        d = L.PlanDesc(kind=L.PLAN_CMP_AVG_BY_GROUP, n_groups=n_groups, cmp_op=CMP[op], threshold=threshold,
                       x_type=1 if x_type == "i32" else 0, y_type=1 if y_type == "i32" else 0)
        return Plan(self, d, columns)

    def plan_qual_pos_hist(self, lmax, columns=(0,)):
        d = L.PlanDesc(kind=L.PLAN_QUAL_POS_HIST, lmax=lmax)
        return Plan(self, d, columns)


def bgzf_scan(data, out_base=0):
    """Walk the BGZF block headers of `data` (bytes / uint8 array) -> (ctypes BgzfBlock array, n, consumed, out_bytes).
    Host-only helper of the C ABI (exon_hip_bgzf_scan)."""
    lib = L.load()
    buf = np.frombuffer(data, np.uint8)
    n, consumed, out_bytes = C.c_int32(), C.c_size_t(), C.c_size_t()
    ptr = buf.ctypes.data if len(buf) else None
    rc = lib.exon_hip_bgzf_scan(ptr, len(buf), out_base, None, 1 << 30, C.byref(n), C.byref(consumed), C.byref(out_bytes))
    if rc:
        raise ExonHipError(rc, lib.exon_hip_last_error(None).decode(errors="replace"))
    blocks = (L.BgzfBlock * max(n.value, 1))()
    rc = lib.exon_hip_bgzf_scan(ptr, len(buf), out_base, blocks, n.value, C.byref(n), C.byref(consumed), C.byref(out_bytes))
    if rc:
        raise ExonHipError(rc, lib.exon_hip_last_error(None).decode(errors="replace"))
    return blocks, n.value, consumed.value, out_bytes.value


def parse_region(region):
    """`name[:start[-end]]` -> (name, start, end|None)   (host helper of the C ABI)."""
    lib = L.load()
    name = C.create_string_buffer(512)
    a, b = C.c_int64(), C.c_int64()
    rc = lib.exon_hip_parse_region(region.encode(), name, 512, C.byref(a), C.byref(b))
    if rc:
        raise ExonHipError(rc, lib.exon_hip_last_error(None).decode(errors="replace"))
    return name.value.decode(), a.value, (None if b.value == L.REGION_OPEN_END else b.value)


def index_query(index_path, region=None, ref_id=None, start=1, end=None, is_bai=False):
    """Region -> list of (start, end) BGZF virtual-position chunks from a .tbi / .bai index."""
    lib = L.load()
    name = None
    if region is not None:
        name, start, end = parse_region(region)
    cap = 4096
    st, en = (C.c_uint64 * cap)(), (C.c_uint64 * cap)()
    n = C.c_int32()
    rc = lib.exon_hip_index_query(str(index_path).encode(), 1 if is_bai else 0, name.encode() if name else None,
                                  -1 if ref_id is None else ref_id, start, L.REGION_OPEN_END if end is None else end,
                                  st, en, cap, C.byref(n))
    if rc:
        raise ExonHipError(rc, lib.exon_hip_last_error(None).decode(errors="replace"))
    return [(st[i], en[i]) for i in range(min(n.value, cap))]


def pack_names(names):
    """'\\0'-terminated names back to back (the wire form of a key dictionary in the C ABI)."""
    return b"".join((n if isinstance(n, bytes) else n.encode()) + b"\0" for n in names)


def unpack_names(packed, n):
    parts = packed.split(b"\0")[:n]
    return [p.decode(errors="replace") for p in parts]


def keys_union(dicts):
    """exon_hip_keys_union (host-only C): the union of the ranks' dictionaries in rank order, first appearance first.
    Returns (union names, [per-rank list: union id of every local id])."""
    lib = L.load()
    packed = b"".join(pack_names(d) for d in dicts)
    n_keys = (C.c_int32 * len(dicts))(*[len(d) for d in dicts])
    n_out, nb = C.c_int32(), C.c_size_t()
    maps = (C.c_int32 * max(1, sum(len(d) for d in dicts)))()
    rc = lib.exon_hip_keys_union(packed, len(packed), n_keys, len(dicts), None, 0, C.byref(n_out), C.byref(nb), maps)
    if rc:
        raise ExonHipError(rc, lib.exon_hip_last_error(None).decode(errors="replace"))
    buf = C.create_string_buffer(max(nb.value, 1))
    rc = lib.exon_hip_keys_union(packed, len(packed), n_keys, len(dicts), buf, nb.value, C.byref(n_out), C.byref(nb), maps)
    if rc:
        raise ExonHipError(rc, lib.exon_hip_last_error(None).decode(errors="replace"))
    out, o = [], 0
    for d in dicts:
        out.append(list(maps[o:o + len(d)]))
        o += len(d)
    return unpack_names(buf.raw[:nb.value], n_out.value), out


def regroup_files_by_size(sizes, target_groups):
    """Whole-file round-robin repartition; returns a list of groups of ORIGINAL file indexes."""
    lib = L.load()
    n = len(sizes)
    s = (C.c_int64 * max(n, 1))(*sizes)
    g = (C.c_int32 * max(n, 1))()
    ng = lib.exon_hip_regroup_files_by_size(s, n, target_groups, g)
    if ng < 0:
        raise ExonHipError(ng, lib.exon_hip_last_error(None).decode(errors="replace"))
    groups = [[] for _ in range(ng)]
    for i in sorted(range(n), key=lambda i: (sizes[i], i)):
        groups[g[i]].append(i)
    return groups


class Scan:
    """Native decoder + device-layout array builder over one file (FileOpener::open + read_batch analogue).
    CPU-only: usable without a GPU."""

    PROJECT = {"vcf": {"id": 1, "ref": 2, "alt": 4, "info": 8, "formats": 16}, "bam": {"name": 1, "cigar": 2, "sequence": 4, "quality_score": 8},
               "bcf": {"id": 1, "ref": 2, "alt": 4}, "sam": {"name": 1, "cigar": 2, "sequence": 4, "quality_score": 8}}

    def __init__(self, path, fmt, compression=None, batch_size=0, info_field=None, region=None, use_index=False,
                 gpu_parse=False, project=()):
        """project: names of the reference's columns beyond the fused kernels' operands (EXON_HIP_PROJECT_*): VCF "id", "ref", "alt",
        "info", "formats" (the last two as the reference's unparsed Utf8 columns; host reader only);
        BAM "name", "cigar", "sequence", "quality_score" -- appended behind the default columns in that order."""
        self.lib = L.load()
        self.fmt = fmt
        mask = 0
        for name in project:
            mask |= self.PROJECT[fmt][name]
        opt = L.ScanOptions(L.FORMATS[fmt], L.COMPRESSION[compression], batch_size,
                            info_field.encode() if info_field else None, region.encode() if region else None,
                            1 if use_index else 0, 1 if gpu_parse else 0, mask)
        h = C.c_void_p()
        rc = self.lib.exon_hip_scan_open(str(path).encode(), C.byref(opt), C.byref(h))
        if rc:
            raise ExonHipError(rc, self.lib.exon_hip_last_error(None).decode(errors="replace"))
        self.h = h

    def _check(self, rc):
        if rc < 0:
            raise ExonHipError(rc, self.lib.exon_hip_last_error(None).decode(errors="replace"))
        return rc

    def schema(self):
        import pyarrow as pa
        sch = L.ArrowSchema()
        self._check(self.lib.exon_hip_scan_schema(self.h, C.byref(sch)))
        return pa.DataType._import_from_c(C.addressof(sch))

    def next_raw(self):
        """-> ctypes ArrowArray (caller must release or move it) or None at end of stream."""
        arr = L.ArrowArray()
        rc = self._check(self.lib.exon_hip_scan_next(self.h, C.byref(arr)))
        return None if rc == 1 else arr

    def __iter__(self):
        """Batches as pyarrow StructArrays (imports = moves each batch)."""
        import pyarrow as pa
        while True:
            arr = self.next_raw()
            if arr is None:
                return
            sch = L.ArrowSchema()
            self._check(self.lib.exon_hip_scan_schema(self.h, C.byref(sch)))
            yield pa.Array._import_from_c(C.addressof(arr), C.addressof(sch))

    def bind_ctx(self, ctx):
        """Batches of a scan opened with gpu_parse=True come out of the GPU decode pipeline on `ctx` (exon_hip_scan_bind_ctx): iterate
        the scan as usual afterwards."""
        self._check(self.lib.exon_hip_scan_bind_ctx(self.h, ctx.h))
        return self

    def decoded_on_gpu(self):
        """(decoded, inflated) of the last Stream.consume: were the records decoded / the BGZF blocks inflated on the GPU?"""
        d, i = C.c_int32(), C.c_int32()
        self._check(self.lib.exon_hip_scan_decoded_on_gpu(self.h, C.byref(d), C.byref(i)))
        return bool(d.value), bool(i.value)

    def dictionary_size(self, column):
        n = C.c_int32()
        self._check(self.lib.exon_hip_scan_dictionary_size(self.h, column, C.byref(n)))
        return n.value

    def dictionary(self, column):
        out = []
        for i in range(self.dictionary_size(column)):
            p = C.c_char_p()
            self._check(self.lib.exon_hip_scan_dictionary_value(self.h, column, i, C.byref(p)))
            out.append(p.value.decode(errors="replace"))
        return out

    def intern(self, column, name):
        i = C.c_int32()
        self._check(self.lib.exon_hip_scan_dictionary_intern(self.h, column, name.encode(), C.byref(i)))
        return i.value

    def rows(self):
        n = C.c_int64()
        self._check(self.lib.exon_hip_scan_rows(self.h, C.byref(n)))
        return n.value

    def index_chunks(self):
        n = C.c_int32()
        self._check(self.lib.exon_hip_scan_index_chunks(self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            self.lib.exon_hip_scan_close(self.h)
            self.h = None


class VCFParser:
    """VCF record parsing on the GPU (exon_hip_vcf_parser_*): text slab in HBM -> device-layout columns in HBM."""

    def __init__(self, ctx, contigs, info_field=None, max_slab_bytes=64 << 20):
        self.ctx = ctx
        names = (C.c_char_p * max(len(contigs), 1))(*[c.encode() for c in contigs])
        h = C.c_void_p()
        ctx._check(ctx.lib.exon_hip_vcf_parser_create(ctx.h, names, len(contigs), info_field.encode() if info_field else None,
                                                      max_slab_bytes, C.byref(h)))
        self.h = h
        self.has_info = bool(info_field)

    def parse_device(self, d_text, n_bytes, stream=None):
        cols = L.VCFColumns()
        ptr = d_text.ptr if isinstance(d_text, DeviceBuffer) else int(d_text)
        self.ctx._check(self.ctx.lib.exon_hip_vcf_parser_parse(self.h, stream, ptr, n_bytes, C.byref(cols)))
        return cols

    def parse_host(self, text, misalign=0):
        """Test helper: copy `text` (bytes of complete lines) to HBM (`misalign` bytes past a 16-byte boundary), parse,
        bring the columns back as numpy arrays."""
        buf = np.frombuffer(text, np.uint8)
        d = self.ctx.to_device(np.concatenate([np.full(misalign, 10, np.uint8), buf, np.zeros(64, np.uint8)]))
        cols = self.parse_device(d.ptr + misalign, len(buf))
        n = cols.n_rows
        nb = (n + 7) // 8

        def get(ptr, dtype, count):
            out = np.empty(count, dtype)
            if count:
                self.ctx._check(self.ctx.lib.exon_hip_memcpy_d2h(self.ctx.h, _np_ptr(out), ptr, out.nbytes, None))
            return out

        res = {"n_rows": n, "n_undecided": cols.n_undecided,
               "chrom_id": get(cols.chrom_id, np.int32, n), "pos": get(cols.pos, np.int64, n),
               "pos_valid": get(cols.pos_valid, np.uint8, nb), "qual": get(cols.qual, np.float32, n),
               "qual_valid": get(cols.qual_valid, np.uint8, nb), "filter_id": get(cols.filter_id, np.int32, n)}
        if self.has_info:
            res["info"] = get(cols.info, np.float32, n)
            res["info_valid"] = get(cols.info_valid, np.uint8, nb)
        return res

    def filters(self):
        n = C.c_int32()
        buf = C.create_string_buffer(1 << 20)
        self.ctx._check(self.ctx.lib.exon_hip_vcf_parser_filters(self.h, buf, len(buf), C.byref(n)))
        names, o = [], 0
        raw = buf.raw
        for _ in range(n.value):
            e = raw.index(b"\0", o)
            names.append(raw[o:e].decode(errors="replace"))
            o = e + 1
        return names

    def close(self):
        if self.h:
            self.ctx.lib.exon_hip_vcf_parser_destroy(self.h)
            self.h = None


class BAMParser:
    """BAM record splitting + field extraction on the GPU (exon_hip_bam_parser_*): inflated bytes in HBM -> columns."""

    def __init__(self, ctx, n_references, max_slab_bytes=64 << 20):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx.lib.exon_hip_bam_parser_create(ctx.h, n_references, max_slab_bytes, C.byref(h)))
        self.h = h

    def parse_host(self, data):
        """Test helper: copy `data` (record bytes, starting at a record boundary) to HBM, split + extract, bring the
        columns back as numpy arrays."""
        buf = np.frombuffer(data, np.uint8)
        d = self.ctx.to_device(np.concatenate([buf, np.zeros(64, np.uint8)]))
        cols = L.BAMColumns()
        self.ctx._check(self.ctx.lib.exon_hip_bam_parser_parse(self.h, None, d.ptr, len(buf), C.byref(cols)))
        n = cols.n_rows if cols.n_undecided == 0 else 0
        nb = (n + 7) // 8

        def get(ptr, dtype, count):
            out = np.empty(count, dtype)
            if count:
                self.ctx._check(self.ctx.lib.exon_hip_memcpy_d2h(self.ctx.h, _np_ptr(out), ptr, out.nbytes, None))
            return out

        return {"n_rows": cols.n_rows, "n_undecided": cols.n_undecided, "consumed_bytes": cols.consumed_bytes,
                "flag": get(cols.flag, np.int32, n), "mapq": get(cols.mapq, np.uint8, n),
                "mapq_valid": get(cols.mapq_valid, np.uint8, nb), "ref_id": get(cols.ref_id, np.int32, n),
                "ref_valid": get(cols.ref_valid, np.uint8, nb), "start": get(cols.start, np.int64, n),
                "end": get(cols.end, np.int64, n), "pos_valid": get(cols.pos_valid, np.uint8, nb)}

    def close(self):
        if self.h:
            self.ctx.lib.exon_hip_bam_parser_destroy(self.h)
            self.h = None


class FASTQParser:
    """FASTQ record splitting on the GPU (exon_hip_fastq_parser_*): text slab in HBM -> per-read views into it."""

    def __init__(self, ctx, max_slab_bytes=64 << 20):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx.lib.exon_hip_fastq_parser_create(ctx.h, max_slab_bytes, C.byref(h)))
        self.h = h

    def parse_device(self, d_text, n_bytes, final=True, stream=None):
        v = L.FASTQViews()
        ptr = d_text.ptr if isinstance(d_text, DeviceBuffer) else int(d_text)
        self.ctx._check(self.ctx.lib.exon_hip_fastq_parser_parse(self.h, stream, ptr, n_bytes, 1 if final else 0, C.byref(v)))
        return v

    def parse_host(self, text, final=True, misalign=0):
        """Test helper: copy `text` to HBM (`misalign` bytes past a 16-byte boundary), split it, bring the views back as
        numpy arrays relative to the start of the text (on the device they index views.text_base, which
        exon_hip_qual_pos_hist_views takes)."""
        buf = np.frombuffer(text, np.uint8)
        d = self.ctx.to_device(np.concatenate([np.full(misalign, 10, np.uint8), buf, np.zeros(64, np.uint8)]))
        v = self.parse_device(d.ptr + misalign, len(buf), final)
        n = v.n_reads

        def get(ptr):
            out = np.empty(n, np.int32)
            if n:
                self.ctx._check(self.ctx.lib.exon_hip_memcpy_d2h(self.ctx.h, _np_ptr(out), ptr, out.nbytes, None))
            return out

        shift = (d.ptr + misalign) - (v.text_base or (d.ptr + misalign))
        return {"n_reads": n, "n_undecided": v.n_undecided, "consumed_bytes": v.consumed_bytes, "views": v, "d_text": v.text_base,
                "keepalive": d, "seq_start": get(v.seq_start) - shift, "seq_end": get(v.seq_end) - shift,
                "qual_start": get(v.qual_start) - shift, "qual_end": get(v.qual_end) - shift}

    def close(self):
        if self.h:
            self.ctx.lib.exon_hip_fastq_parser_destroy(self.h)
            self.h = None


class Plan:
    def __init__(self, ctx, desc, columns):
        self.ctx = ctx
        for i, c in enumerate(columns):
            desc.columns[i] = c
        self.desc = desc
        self.n_cols = len(columns)
        h = C.c_void_p()
        ctx._check(ctx.lib.exon_hip_plan_create(ctx.h, C.byref(desc), C.byref(h)))
        self.h = h
        a, b = C.c_int64(), C.c_int64()
        ctx._check(ctx.lib.exon_hip_plan_state_size(h, C.byref(a), C.byref(b)))
        self.n_i64, self.n_f64 = a.value, b.value

    def open(self, partition=0):
        return Stream(self, partition)

    def launch(self, columns, n, d_state, overwrite=False, stream=None):
        """exon_hip_plan_launch: the plan's fused kernel over HBM-resident columns (operator argument order; each a
        (values, validity, offsets) triple of DeviceBuffers / raw device pointers) into the packed state `d_state`
        ([n_i64 int64][n_f64 float64], DeviceBuffer or raw pointer).  overwrite: state := this batch (no zeroing)."""
        cols = (Column * len(columns))(*[_col(v, b, o, n) for (v, b, o) in columns])
        ptr = d_state.ptr if isinstance(d_state, DeviceBuffer) else int(d_state)
        self.ctx._check(self.ctx.lib.exon_hip_plan_launch(self.h, stream, cols, len(columns), n,
                                                          L.LAUNCH_OVERWRITE if overwrite else L.LAUNCH_ACCUMULATE, ptr))

    def prepared(self, columns, n, d_state, overwrite=False, stream=None):
        """launch() with the argument marshalling done once: returns a zero-argument callable that enqueues the same
        launch again (a resident table queried repeatedly; a 10 M-row step is ~20 us of GPU time, so per-call ctypes
        struct building would be what gets measured)."""
        cols = (Column * len(columns))(*[_col(v, b, o, n) for (v, b, o) in columns])
        ptr = C.c_void_p(d_state.ptr if isinstance(d_state, DeviceBuffer) else int(d_state))
        fn, check, h = self.ctx.lib.exon_hip_plan_launch, self.ctx._check, self.h
        ncol, nn = C.c_int32(len(columns)), C.c_int64(n)
        flags = C.c_int32(L.LAUNCH_OVERWRITE if overwrite else L.LAUNCH_ACCUMULATE)
        st = C.c_void_p(stream)

        def go():
            rc = fn(h, st, cols, ncol, nn, flags, ptr)
            if rc:
                check(rc)
        go.keepalive = cols
        return go

    def launch_chunks(self, chunks, d_state, overwrite=False, stream=None):
        """exon_hip_plan_launch_chunks: `chunks` = [(columns, n)], columns as in launch().  A quality-histogram plan
        runs up to 64 chunks per kernel launch; the other kinds launch per chunk."""
        ncol = len(chunks[0][0]) if chunks else self.n_cols
        flat = [_col(v, b, o, n) for (cols, n) in chunks for (v, b, o) in cols]
        arr = (Column * max(1, len(flat)))(*flat)
        ns = (C.c_int64 * max(1, len(chunks)))(*[n for _, n in chunks])
        ptr = d_state.ptr if isinstance(d_state, DeviceBuffer) else int(d_state)
        self.ctx._check(self.ctx.lib.exon_hip_plan_launch_chunks(self.h, stream, arr, ncol, len(chunks), ns,
                                                                 L.LAUNCH_OVERWRITE if overwrite else L.LAUNCH_ACCUMULATE, ptr))

    def close(self):
        if self.h:
            self.ctx.lib.exon_hip_plan_destroy(self.h)
            self.h = None


class Stream:
    """One partition's execution (ExecutionPlan::execute analogue)."""

    def __init__(self, plan, partition):
        self.plan, self.ctx = plan, plan.ctx
        h = C.c_void_p()
        self.ctx._check(self.ctx.lib.exon_hip_stream_open(plan.h, partition, C.byref(h)))
        self.h = h

    def push(self, batch):
        """Push a pyarrow.RecordBatch (host memory) through the Arrow C Data Interface (moved)."""
        arr = L.ArrowArray()
        sch = L.ArrowSchema()
        batch._export_to_c(C.addressof(arr), C.addressof(sch))
        try:
            self.ctx._check(self.ctx.lib.exon_hip_stream_push(self.h, C.byref(arr)))
        finally:
            if sch.release:
                C.CFUNCTYPE(None, C.POINTER(L.ArrowSchema))(sch.release)(C.byref(sch))

    def push_device(self, columns, n_rows):
        """columns: list of (values DeviceBuffer, validity DeviceBuffer|None, offsets DeviceBuffer|None) already
        in HBM, in batch-child order; wrapped as an ArrowDeviceArray (ARROW_DEVICE_ROCM)."""
        keep = []
        kids = (C.POINTER(L.ArrowArray) * len(columns))()
        for i, (values, validity, offsets) in enumerate(columns):
            a = L.ArrowArray()
            if offsets is not None:
                bufs = (C.c_void_p * 3)(validity.ptr if validity else None, offsets.ptr, values.ptr)
                a.n_buffers = 3
            else:
                bufs = (C.c_void_p * 2)(validity.ptr if validity else None, values.ptr)
                a.n_buffers = 2
            a.length = n_rows
            a.null_count = -1 if validity else 0
            a.buffers = C.cast(bufs, C.POINTER(C.c_void_p))
            keep += [a, bufs]
            kids[i] = C.pointer(a)
        top = L.ArrowDeviceArray()
        tb = (C.c_void_p * 1)(None)
        top.array.length = n_rows
        top.array.n_buffers = 1
        top.array.buffers = C.cast(tb, C.POINTER(C.c_void_p))
        top.array.n_children = len(columns)
        top.array.children = C.cast(kids, C.POINTER(C.POINTER(L.ArrowArray)))
        top.device_id = self.ctx.device
        top.device_type = L.ARROW_DEVICE_ROCM
        # no sync_event travels with the batch, which by the Arrow C Device interface means "the data is ready": whatever this
        # harness queued on the context's stream to produce the columns (gen_c4 ...) has to be finished first (the plan runs on
        # its own stream; found by a test that only failed beside three other workers on the same GPU)
        self.ctx.sync()
        self.ctx._check(self.ctx.lib.exon_hip_stream_push_device(self.h, C.byref(top)))
        self._keep = keep + [kids, tb, top]

    def consume(self, scan):
        """Pull every batch of a Scan through this stream (GpuFilterAggExec::execute in native code)."""
        n = C.c_int64()
        rc = self.ctx.lib.exon_hip_stream_consume_scan(self.h, scan.h, C.byref(n))
        if rc < 0:
            msg = self.ctx.lib.exon_hip_last_error(self.ctx.h).decode(errors="replace") or self.ctx.lib.exon_hip_last_error(None).decode(errors="replace")
            raise ExonHipError(rc, msg)
        return n.value

    def state(self):
        a, b, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ctx._check(self.ctx.lib.exon_hip_stream_state(self.h, C.byref(a), C.byref(b), C.byref(s)))
        return a.value, b.value, s.value

    def all_reduce(self, rccl_comm):
        """Merge of the partial state over the ranks of `rccl_comm` (an ncclComm_t as an integer) on the stream: one
        ncclAllGather + fixed-order fold; afterwards every rank's state is the sum over all ranks."""
        self.ctx._check(self.ctx.lib.exon_hip_stream_all_reduce(self.h, C.c_void_p(rccl_comm)))

    # ---- group keys by value (ABI 4): what the state's indexes stand for when the rows came from files
    def keys(self):
        """(names in state-index order, agreed) -- the dictionary exon_hip_stream_consume_scan built up; `agreed` is True once
        set_keys / reconcile_keys ran and no scan was consumed since."""
        n, b, a = C.c_int32(), C.c_size_t(), C.c_int32()
        self.ctx._check(self.ctx.lib.exon_hip_stream_keys(self.h, None, 0, C.byref(n), C.byref(b), C.byref(a)))
        buf = C.create_string_buffer(max(b.value, 1))
        self.ctx._check(self.ctx.lib.exon_hip_stream_keys(self.h, buf, b.value, C.byref(n), C.byref(b), C.byref(a)))
        return unpack_names(buf.raw[:b.value], n.value), bool(a.value)

    def set_keys(self, names):
        """Adopt `names` as the dictionary (every key held must be among them): the device state is permuted into that order."""
        packed = pack_names(names)
        self.ctx._check(self.ctx.lib.exon_hip_stream_set_keys(self.h, packed, len(packed), len(names)))

    def reconcile_keys(self, rccl_comm):
        """Collective: the ranks of `rccl_comm` (an ncclComm_t as an integer) agree on the union dictionary, natively."""
        self.ctx._check(self.ctx.lib.exon_hip_stream_reconcile_keys(self.h, C.c_void_p(rccl_comm)))

    def set_region_contig(self, name):
        """Region plans over files: the contig by name; every consumed file resolves it in its own header order."""
        self.ctx._check(self.ctx.lib.exon_hip_stream_set_region_contig(self.h, name.encode()))

    def snapshot(self):
        """(counts, sums) of the state as it stands, without finishing the stream."""
        a, b, s = self.state()
        self.ctx._check(self.ctx.lib.exon_hip_sync(self.ctx.h, C.c_void_p(s)))
        counts = np.zeros(self.plan.n_i64, np.int64)
        sums = np.zeros(self.plan.n_f64, np.float64)
        if counts.size:
            self.ctx._check(self.ctx.lib.exon_hip_memcpy_d2h(self.ctx.h, _np_ptr(counts), C.c_void_p(a), counts.nbytes, None))
        if sums.size:
            self.ctx._check(self.ctx.lib.exon_hip_memcpy_d2h(self.ctx.h, _np_ptr(sums), C.c_void_p(b), sums.nbytes, None))
        self.ctx.sync()
        return counts, sums

    def reset(self):
        """New query on this stream: the next launch defines the state (overwrite mode, no zeroing kernel)."""
        self.ctx._check(self.ctx.lib.exon_hip_stream_reset(self.h))

    def sync(self):
        self.ctx._check(self.ctx.lib.exon_hip_stream_sync(self.h))

    def finish(self):
        counts = np.zeros(self.plan.n_i64, np.int64)
        sums = np.zeros(self.plan.n_f64, np.float64)
        self.ctx._check(self.ctx.lib.exon_hip_stream_finish(self.h, _np_ptr(counts), _np_ptr(sums)))
        return counts, sums

    def finish_arrow(self):
        import pyarrow as pa
        arr, sch = L.ArrowArray(), L.ArrowSchema()
        self.ctx._check(self.ctx.lib.exon_hip_stream_finish_arrow(self.h, C.byref(arr), C.byref(sch)))
        return pa.Array._import_from_c(C.addressof(arr), C.addressof(sch))

    def close(self):
        if self.h:
            self.ctx.lib.exon_hip_stream_close(self.h)
            self.h = None
