"""ctypes binding of exon_amd/lib/libexon_hip.so (the C ABI of include/exon_hip.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the operators
raises -- the product path never routes through the oracle or numpy.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# EXON_HIP_LIB: another build of the same library (A/B runs of kernel variants: tools/build_variant.sh)
LIB_PATH = os.environ.get("EXON_HIP_LIB") or os.path.join(_HERE, "lib", "libexon_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class ExonHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"exon_hip status {code}: {msg}")
        self.code = code


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))),
    ("dictionary", C.POINTER(ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p),
]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
    ("children", C.POINTER(C.POINTER(ArrowArray))), ("dictionary", C.POINTER(ArrowArray)),
    ("release", C.c_void_p), ("private_data", C.c_void_p),
]


class ArrowDeviceArray(C.Structure):
    _fields_ = [("array", ArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32),
                ("sync_event", C.c_void_p), ("reserved", C.c_int64 * 3)]


ARROW_DEVICE_CPU = 1
ARROW_DEVICE_ROCM = 10


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("gcn_arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("hbm_bytes", C.c_int64), ("clock_khz", C.c_int32),
                ("reserved", C.c_int32)]


class Column(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("offsets", C.c_void_p), ("length", C.c_int64)]


class PlanDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_groups", C.c_int32),
        ("region_chrom_id", C.c_int32), ("x_type", C.c_int32),
        ("region_start", C.c_int64), ("region_end", C.c_int64),
        ("flag_mask", C.c_int32), ("flag_value", C.c_int32), ("mapq_min", C.c_int32),
        ("cmp_op", C.c_int32), ("threshold", C.c_double),
        ("lmax", C.c_int32), ("y_type", C.c_int32),
        ("columns", C.c_int32 * 4),
    ]


class BgzfBlock(C.Structure):
    _fields_ = [("comp_offset", C.c_uint32), ("comp_size", C.c_uint32), ("out_offset", C.c_uint32), ("out_size", C.c_uint32),
                ("crc32", C.c_uint32), ("reserved", C.c_uint32)]


class BAMColumns(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_undecided", C.c_int64), ("consumed_bytes", C.c_int64), ("flag", C.c_void_p),
                ("mapq", C.c_void_p), ("mapq_valid", C.c_void_p), ("ref_id", C.c_void_p), ("ref_valid", C.c_void_p),
                ("start", C.c_void_p), ("end", C.c_void_p), ("pos_valid", C.c_void_p)]


class FASTQViews(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_undecided", C.c_int64), ("consumed_bytes", C.c_int64),
                ("seq_start", C.c_void_p), ("seq_end", C.c_void_p), ("qual_start", C.c_void_p), ("qual_end", C.c_void_p),
                ("text_base", C.c_void_p), ("head_start", C.c_void_p), ("head_end", C.c_void_p)]


class VCFColumns(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_undecided", C.c_int64), ("chrom_id", C.c_void_p), ("pos", C.c_void_p),
                ("pos_valid", C.c_void_p), ("qual", C.c_void_p), ("qual_valid", C.c_void_p), ("filter_id", C.c_void_p),
                ("info", C.c_void_p), ("info_valid", C.c_void_p), ("consumed_bytes", C.c_int64),
                ("n_info", C.c_int32), ("reserved", C.c_int32), ("infos", C.c_void_p * 16), ("infos_valid", C.c_void_p * 16),
                ("info_kinds", C.c_char * 16), ("list_offsets", C.c_void_p * 16), ("list_item_valid", C.c_void_p * 16),
                ("info_nulls", C.c_int32 * 16)]


class ScanOptions(C.Structure):
    _fields_ = [("format", C.c_int32), ("compression", C.c_int32), ("batch_size", C.c_int64),
                ("info_field", C.c_char_p), ("region", C.c_char_p), ("use_index", C.c_int32), ("gpu_parse", C.c_int32),
                ("projection", C.c_uint64)]


class GzipStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("calls", "chunks", "repairs", "overflow_retries", "members", "comp_bytes", "out_bytes")]


FORMATS = {"vcf": 1, "bam": 2, "fastq": 3, "fasta": 4, "sam": 5, "bcf": 6, "cram": 7}
COMPRESSION = {"auto": 0, None: 0, "none": 1, "gzip": 2}

PLAN_REGION_COUNT = 2
PLAN_FLAG_MAPQ_GROUP_COUNT = 3
PLAN_CMP_AVG_BY_GROUP = 4
PLAN_QUAL_POS_HIST = 5
PLAN_OVERLAP_COUNT = 6
PLAN_WITHIN_COUNT = 7
LAUNCH_ACCUMULATE = 0
LAUNCH_OVERWRITE = 1
CMP = {">": 0, ">=": 1, "<": 2, "<=": 3, "=": 4, "==": 4, "!=": 5, "<>": 5}
REGION_OPEN_END = 2**63 - 1

_vp, _i32, _i64, _u64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
_colp = C.POINTER(Column)

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)  # exon_hip_allgather_fn

# every symbol include/exon_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "exon_hip_abi_version": (C.c_int, []),
    "exon_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "exon_hip_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "exon_hip_ctx_destroy": (C.c_int, [_vp]),
    "exon_hip_ctx_info": (C.c_int, [_vp, C.POINTER(DeviceInfo)]),
    "exon_hip_last_error": (C.c_char_p, [_vp]),
    "exon_hip_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "exon_hip_free": (C.c_int, [_vp, _vp]),
    "exon_hip_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "exon_hip_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "exon_hip_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t, _vp]),
    "exon_hip_sync": (C.c_int, [_vp, _vp]),
    "exon_hip_read_probe": (C.c_int, [_vp, _vp, C.POINTER(_vp), _i32, _i64, _i32, C.POINTER(_dbl), C.POINTER(_i64)]),
    "exon_hip_timer_start": (C.c_int, [_vp, _vp]),
    "exon_hip_timer_stop_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "exon_hip_region_count": (C.c_int, [_vp, _vp, _colp, _colp, _i64, _i32, _i64, _i64, _vp]),
    "exon_hip_overlap_count": (C.c_int, [_vp, _vp, _colp, _colp, _colp, _i64, _i32, _i64, _i64, _vp]),
    "exon_hip_within_count": (C.c_int, [_vp, _vp, _colp, _colp, _colp, _i64, _i32, _i64, _i64, _vp]),
    "exon_hip_flag_mapq_group_count": (C.c_int, [_vp, _vp, _colp, _colp, _colp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "exon_hip_cmp_avg_by_group": (C.c_int, [_vp, _vp, _colp, _colp, _colp, _i64, _dbl, _i32, _i32, _vp, _vp]),
    "exon_hip_qual_pos_hist": (C.c_int, [_vp, _vp, _colp, _i64, _i32, _vp]),
    "exon_hip_gen_c6": (C.c_int, [_vp, _vp, _u64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "exon_hip_gen_c2": (C.c_int, [_vp, _vp, _u64, _i64, _i64, _i64, _vp, _vp]),
    "exon_hip_gen_c3": (C.c_int, [_vp, _vp, _u64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "exon_hip_gen_c4": (C.c_int, [_vp, _vp, _u64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "exon_hip_gen_c5": (C.c_int, [_vp, _vp, _u64, _i64, _i64, _i32, _vp, _vp]),
    "exon_hip_parse_region": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(_i64), C.POINTER(_i64)]),
    "exon_hip_regroup_files_by_size": (C.c_int, [C.POINTER(_i64), _i32, _i32, C.POINTER(_i32)]),
    "exon_hip_plan_create": (C.c_int, [_vp, C.POINTER(PlanDesc), C.POINTER(_vp)]),
    "exon_hip_plan_destroy": (C.c_int, [_vp]),
    "exon_hip_plan_state_size": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "exon_hip_plan_launch": (C.c_int, [_vp, _vp, _colp, _i32, _i64, _i32, _vp]),
    "exon_hip_plan_launch_chunks": (C.c_int, [_vp, _vp, _colp, _i32, _i32, _vp, _i32, _vp]),
    "exon_hip_fold_states": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _vp]),
    "exon_hip_merge_states": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "exon_hip_rccl_unique_id": (C.c_int, [_vp]),
    "exon_hip_rccl_comm_init": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "exon_hip_rccl_comm_destroy": (C.c_int, [_vp]),
    "exon_hip_comm_wrap_rccl": (C.c_int, [_vp, C.POINTER(_vp)]),
    "exon_hip_comm_from_callbacks": (C.c_int, [_i32, _i32, ALLGATHER_FN, _vp, C.POINTER(_vp)]),
    "exon_hip_rccl_comm_count": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "exon_hip_stream_reset": (C.c_int, [_vp]),
    "exon_hip_stream_open": (C.c_int, [_vp, _i32, C.POINTER(_vp)]),
    "exon_hip_stream_push": (C.c_int, [_vp, C.POINTER(ArrowArray)]),
    "exon_hip_stream_push_device": (C.c_int, [_vp, C.POINTER(ArrowDeviceArray)]),
    "exon_hip_stream_state": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "exon_hip_stream_all_reduce": (C.c_int, [_vp, _vp]),
    "exon_hip_stream_keys": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(_i32), C.POINTER(C.c_size_t), C.POINTER(_i32)]),
    "exon_hip_stream_set_keys": (C.c_int, [_vp, C.c_char_p, C.c_size_t, _i32]),
    "exon_hip_keys_union": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(_i32), _i32, _vp, C.c_size_t, C.POINTER(_i32),
                                      C.POINTER(C.c_size_t), C.POINTER(_i32)]),
    "exon_hip_stream_reconcile_keys": (C.c_int, [_vp, _vp]),
    "exon_hip_stream_set_region_contig": (C.c_int, [_vp, C.c_char_p]),
    "exon_hip_stream_sync": (C.c_int, [_vp]),
    "exon_hip_stream_finish": (C.c_int, [_vp, _vp, _vp]),
    "exon_hip_stream_finish_arrow": (C.c_int, [_vp, C.POINTER(ArrowArray), C.POINTER(ArrowSchema)]),
    "exon_hip_stream_close": (C.c_int, [_vp]),
    "exon_hip_scan_open": (C.c_int, [C.c_char_p, C.POINTER(ScanOptions), C.POINTER(_vp)]),
    "exon_hip_scan_schema": (C.c_int, [_vp, C.POINTER(ArrowSchema)]),
    "exon_hip_scan_next": (C.c_int, [_vp, C.POINTER(ArrowArray)]),
    "exon_hip_scan_dictionary_size": (C.c_int, [_vp, _i32, C.POINTER(_i32)]),
    "exon_hip_scan_dictionary_intern": (C.c_int, [_vp, _i32, C.c_char_p, C.POINTER(_i32)]),
    "exon_hip_scan_dictionary_value": (C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_char_p)]),
    "exon_hip_scan_rows": (C.c_int, [_vp, C.POINTER(_i64)]),
    "exon_hip_scan_bind_ctx": (C.c_int, [_vp, _vp]),
    "exon_hip_scan_decoded_on_gpu": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "exon_hip_scan_close": (C.c_int, [_vp]),
    "exon_hip_scan_index_chunks": (C.c_int, [_vp, C.POINTER(_i32)]),
    "exon_hip_index_query": (C.c_int, [C.c_char_p, _i32, C.c_char_p, _i32, _i64, _i64, C.POINTER(_u64), C.POINTER(_u64),
                                       _i32, C.POINTER(_i32)]),
    "exon_hip_stream_consume_scan": (C.c_int, [_vp, _vp, C.POINTER(_i64)]),
    "exon_hip_vcf_parser_create": (C.c_int, [_vp, C.POINTER(C.c_char_p), _i32, C.c_char_p, _i64, C.POINTER(_vp)]),
    "exon_hip_vcf_parser_parse": (C.c_int, [_vp, _vp, _vp, _i64, C.POINTER(VCFColumns)]),
    "exon_hip_vcf_parser_filters": (C.c_int, [_vp, C.c_char_p, C.c_size_t, C.POINTER(_i32)]),
    "exon_hip_vcf_parser_info_values": (C.c_int, [_vp, _i32, C.c_char_p, C.c_size_t, C.POINTER(_i32)]),
    "exon_hip_vcf_parser_set_null_key": (C.c_int, [_vp, _i32]),
    "exon_hip_vcf_parser_destroy": (C.c_int, [_vp]),
    "exon_hip_qual_pos_hist_chunks": (C.c_int, [_vp, _vp, _colp, _i32, _vp, _i32, _vp]),
    "exon_hip_qual_pos_hist_views": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "exon_hip_fastq_parser_create": (C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    "exon_hip_fastq_parser_parse": (C.c_int, [_vp, _vp, _vp, _i64, _i32, C.POINTER(FASTQViews)]),
    "exon_hip_fastq_parser_destroy": (C.c_int, [_vp]),
    "exon_hip_bam_parser_create": (C.c_int, [_vp, _i32, _i64, C.POINTER(_vp)]),
    "exon_hip_bam_parser_parse": (C.c_int, [_vp, _vp, _vp, _i64, C.POINTER(BAMColumns)]),
    "exon_hip_bam_parser_destroy": (C.c_int, [_vp]),
    "exon_hip_sam_parser_create": (C.c_int, [_vp, C.POINTER(C.c_char_p), _i32, _i64, C.POINTER(_vp)]),
    "exon_hip_sam_parser_parse": (C.c_int, [_vp, _vp, _vp, _i64, C.POINTER(BAMColumns)]),
    "exon_hip_sam_parser_destroy": (C.c_int, [_vp]),
    "exon_hip_bcf_parser_create": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i64, C.POINTER(_vp)]),
    "exon_hip_bcf_parser_set_info_keys": (C.c_int, [_vp, C.POINTER(_i32), C.c_char_p, _i32]),
    "exon_hip_bcf_parser_parse": (C.c_int, [_vp, _vp, _vp, _i64, C.POINTER(VCFColumns)]),
    "exon_hip_bcf_parser_filters": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), _i32, C.POINTER(_i32)]),
    "exon_hip_bcf_parser_destroy": (C.c_int, [_vp]),
    "exon_hip_bgzf_scan": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.POINTER(BgzfBlock), _i32, C.POINTER(_i32),
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "exon_hip_bgzf_inflate": (C.c_int, [_vp, _vp, _vp, C.POINTER(BgzfBlock), _i32, _vp, _i32, C.POINTER(_i32)]),
    "exon_hip_bgzf_forget_stream": (C.c_int, [_vp]),
    "exon_hip_gzip_stream_create": (C.c_int, [_vp, _i64, _i64, C.POINTER(_vp)]),
    "exon_hip_gzip_stream_decode": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _i64, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "exon_hip_gzip_stream_get_stats": (C.c_int, [_vp, C.POINTER(GzipStats)]),
    "exon_hip_gzip_stream_destroy": (C.c_int, [_vp]),
    "exon_hip_bgzf_inflate_par_stats": (C.c_int, [_vp, _vp]),
}


def build(force=False, verbose=False):
    """Compile libexon_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load():
    """Load the HIP library; raises (no fallback) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(exon_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
