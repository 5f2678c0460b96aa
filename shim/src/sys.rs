//! Raw FFI of include/exon_hip.h (hand-written; bindgen would produce the same).  ABI version 4.
//! `tests/test_shim_layout.py` parses the `#[repr(C)]` structs below and checks field order, offsets and sizes against
//! `include/exon_hip.h` through a gcc-compiled `offsetof` dump, so this file cannot drift from the header unnoticed.
#![allow(non_camel_case_types)]
use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct exon_hip_ctx { _p: [u8; 0] }
#[repr(C)]
pub struct exon_hip_plan { _p: [u8; 0] }
#[repr(C)]
pub struct exon_hip_stream { _p: [u8; 0] }
#[repr(C)]
pub struct exon_hip_scan { _p: [u8; 0] }

pub const EXON_HIP_ABI_VERSION: i32 = 5;
pub const EXON_HIP_PLAN_REGION_COUNT: i32 = 2;
pub const EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT: i32 = 3;
pub const EXON_HIP_PLAN_CMP_AVG_BY_GROUP: i32 = 4;
pub const EXON_HIP_PLAN_QUAL_POS_HIST: i32 = 5;
pub const EXON_HIP_PLAN_OVERLAP_COUNT: i32 = 6;
pub const EXON_HIP_PLAN_WITHIN_COUNT: i32 = 7;
pub const EXON_HIP_GT: i32 = 0;
pub const EXON_HIP_GE: i32 = 1;
pub const EXON_HIP_LT: i32 = 2;
pub const EXON_HIP_LE: i32 = 3;
pub const EXON_HIP_ECAPACITY: i32 = -6; // more distinct group keys than the plan's n_groups; the stream is unchanged
pub const EXON_HIP_EQ: i32 = 4;
pub const EXON_HIP_NE: i32 = 5;
pub const EXON_HIP_FORMAT_VCF: i32 = 1;
pub const EXON_HIP_FORMAT_BAM: i32 = 2;
pub const EXON_HIP_FORMAT_FASTQ: i32 = 3;
pub const EXON_HIP_FORMAT_SAM: i32 = 5;
pub const EXON_HIP_FORMAT_BCF: i32 = 6;
pub const EXON_HIP_FORMAT_CRAM: i32 = 7;
pub const EXON_HIP_COMPRESSION_AUTO: i32 = 0;
pub const EXON_HIP_MAX_GROUPS: i32 = 4096;
pub const EXON_HIP_REGION_OPEN_END: i64 = i64::MAX;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct exon_hip_plan_desc {
    pub kind: i32,
    pub n_groups: i32,
    pub region_chrom_id: i32,
    pub x_type: i32,
    pub region_start: i64,
    pub region_end: i64,
    pub flag_mask: i32,
    pub flag_value: i32,
    pub mapq_min: i32,
    pub cmp_op: i32,
    pub threshold: f64,
    pub lmax: i32,
    pub y_type: i32,
    pub columns: [i32; 4],
}

/// mirrors `exon_hip_scan_options`
#[repr(C)]
pub struct exon_hip_scan_options {
    pub format: i32,
    pub compression: i32,
    pub batch_size: i64,
    pub info_field: *const c_char,
    pub region: *const c_char,
    pub use_index: i32,
    pub gpu_parse: i32,
    pub projection: u64, // EXON_HIP_PROJECT_* (ABI 5): columns beyond the fused kernels' operands; 0 on the aggregate path
}

/// mirrors `exon_hip_column` (an Arrow array already resident in HBM)
#[repr(C)]
pub struct exon_hip_column {
    pub values: *const c_void,
    pub validity: *const u8,
    pub offsets: *const i32,
    pub length: i64,
}

/// mirrors `exon_hip_device_info`
#[repr(C)]
pub struct exon_hip_device_info {
    pub name: [c_char; 64],
    pub gcn_arch: [c_char; 32],
    pub compute_units: i32,
    pub wavefront_size: i32,
    pub hbm_bytes: i64,
    pub clock_khz: i32,
    pub reserved: i32,
}

extern "C" {
    pub fn exon_hip_abi_version() -> c_int;
    pub fn exon_hip_ctx_create(device: c_int, out: *mut *mut exon_hip_ctx) -> c_int;
    pub fn exon_hip_ctx_destroy(ctx: *mut exon_hip_ctx) -> c_int;
    pub fn exon_hip_ctx_info(ctx: *mut exon_hip_ctx, out: *mut exon_hip_device_info) -> c_int;
    pub fn exon_hip_last_error(ctx: *const exon_hip_ctx) -> *const c_char;
    pub fn exon_hip_plan_create(ctx: *mut exon_hip_ctx, desc: *const exon_hip_plan_desc, out: *mut *mut exon_hip_plan) -> c_int;
    pub fn exon_hip_plan_destroy(plan: *mut exon_hip_plan) -> c_int;
    pub fn exon_hip_plan_state_size(plan: *const exon_hip_plan, n_i64: *mut i64, n_f64: *mut i64) -> c_int;
    pub fn exon_hip_stream_open(plan: *mut exon_hip_plan, partition: i32, out: *mut *mut exon_hip_stream) -> c_int;
    /// moves `batch` (the library calls batch.release exactly once, success or not)
    pub fn exon_hip_stream_push(s: *mut exon_hip_stream, batch: *mut FFI_ArrowArray) -> c_int;
    pub fn exon_hip_stream_reset(s: *mut exon_hip_stream) -> c_int;
    pub fn exon_hip_stream_state(s: *mut exon_hip_stream, d_i64: *mut *mut i64, d_f64: *mut *mut f64, hip_stream: *mut *mut c_void) -> c_int;
    /// AggregateExec(Final) across GPUs: one ncclAllGather of the packed state + a fold in rank order (comm: ncclComm_t)
    pub fn exon_hip_stream_all_reduce(s: *mut exon_hip_stream, rccl_comm: *mut c_void) -> c_int;
    // ---- group keys by VALUE (ABI 4): FILTER-list / reference ids are per file; the stream remembers the value behind every
    // state index, re-keys each further scan into its own order, and the ranks agree on one dictionary before the merge
    pub fn exon_hip_stream_keys(s: *mut exon_hip_stream, buf: *mut c_char, cap: usize, n_keys: *mut i32, bytes: *mut usize, agreed: *mut i32) -> c_int;
    pub fn exon_hip_stream_set_keys(s: *mut exon_hip_stream, packed_names: *const c_char, packed_bytes: usize, n_keys: i32) -> c_int;
    pub fn exon_hip_keys_union(packed: *const c_char, packed_bytes: usize, n_keys: *const i32, world: i32, out: *mut c_char, cap: usize, n_out: *mut i32, out_bytes: *mut usize, maps: *mut i32) -> c_int;
    /// collective over the communicator: two small ncclAllGathers move the dictionaries, every rank permutes its state into the union
    pub fn exon_hip_stream_reconcile_keys(s: *mut exon_hip_stream, rccl_comm: *mut c_void) -> c_int;
    /// region plans over files: the contig travels by name, each file resolves it in its own header order
    pub fn exon_hip_stream_set_region_contig(s: *mut exon_hip_stream, name: *const c_char) -> c_int;
    pub fn exon_hip_stream_finish_arrow(s: *mut exon_hip_stream, out: *mut FFI_ArrowArray, out_schema: *mut FFI_ArrowSchema) -> c_int;
    pub fn exon_hip_stream_close(s: *mut exon_hip_stream) -> c_int;
    pub fn exon_hip_rccl_unique_id(id128: *mut u8) -> c_int;
    pub fn exon_hip_rccl_comm_init(ctx: *mut exon_hip_ctx, id128: *const u8, world: i32, rank: i32, comm: *mut *mut c_void) -> c_int;
    pub fn exon_hip_rccl_comm_destroy(comm: *mut c_void) -> c_int;
    pub fn exon_hip_rccl_comm_count(comm: *mut c_void, world: *mut i32, rank: *mut i32) -> c_int;
    pub fn exon_hip_parse_region(region: *const c_char, name_out: *mut c_char, name_cap: usize, start: *mut i64, end: *mut i64) -> c_int;
    pub fn exon_hip_regroup_files_by_size(sizes: *const i64, n_files: i32, target_groups: i32, group_of: *mut i32) -> c_int;

    // ---- whole-partition path: the library opens the file itself (native decoders; with gpu_parse = 1 the file's bytes go
    // to HBM as they are -- BGZF blocks are inflated and VCF / BCF / BAM / SAM / FASTQ records decoded on the GPU) ----
    pub fn exon_hip_scan_open(path: *const c_char, options: *const exon_hip_scan_options, out: *mut *mut exon_hip_scan) -> c_int;
    pub fn exon_hip_scan_dictionary_size(scan: *mut exon_hip_scan, column: i32, size: *mut i32) -> c_int;
    pub fn exon_hip_scan_dictionary_intern(scan: *mut exon_hip_scan, column: i32, name: *const c_char, id: *mut i32) -> c_int;
    pub fn exon_hip_scan_dictionary_value(scan: *mut exon_hip_scan, column: i32, id: i32, name: *mut *const c_char) -> c_int;
    pub fn exon_hip_scan_decoded_on_gpu(scan: *mut exon_hip_scan, decoded: *mut i32, inflated: *mut i32) -> c_int;
    pub fn exon_hip_scan_rows(scan: *mut exon_hip_scan, rows_emitted: *mut i64) -> c_int;
    pub fn exon_hip_scan_bind_ctx(scan: *mut exon_hip_scan, ctx: *mut exon_hip_ctx) -> c_int;
    pub fn exon_hip_scan_close(scan: *mut exon_hip_scan) -> c_int;
    /// GpuFilterAggExec::execute for one file in one call
    pub fn exon_hip_stream_consume_scan(s: *mut exon_hip_stream, scan: *mut exon_hip_scan, rows: *mut i64) -> c_int;
}
