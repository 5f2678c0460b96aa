/*
 * exon_hip.h -- C ABI of the MI355X-native scan -> filter -> aggregate engine (libexon_hip.so).
 *
 * This is the drop-in boundary for Exon's hot path.  Every entry point is
 * extern "C", takes plain pointers / sizes / opaque handles, returns an int status and never
 * throws.  Data crosses as the Arrow C Data Interface (host batches) or the Arrow C Device Data
 * Interface (HBM-resident batches, device_type = ARROW_DEVICE_ROCM).  A Rust binding needs only
 * `extern "C"` declarations of the functions below plus arrow-rs `arrow::ffi` (already enabled in
 * the reference: exon/exon-core/Cargo.toml:73-77, exon/exon-core/src/ffi/mod.rs:58-73); see
 * INTEGRATION.md for the shim.
 *
 * What each group replaces in the reference (paths relative to /root/reference/exon):
 *   exon_hip_plan_* / exon_hip_stream_*   ExecutionPlan::execute of the operator chain
 *        AggregateExec(Partial) <- [CoalesceBatchesExec] <- FilterExec <- {VCF,BAM,FASTQ}Scan
 *        (exon-core/src/datasources/vcf/scanner.rs:142-162, bam/scanner.rs:138-158; the
 *        FilterExec/AggregateExec themselves are DataFusion 44, not vendored)
 *   exon_hip_region_count                 RegionPhysicalExpr::evaluate
 *        (exon-core/src/physical_plan/region_physical_expr.rs:220-240), region_match UDF
 *        (exon-core/src/udfs/vcf/mod.rs:65-131), IndexedAsyncBatchStream::filter
 *        (exon-vcf/src/indexed_async_batch_stream.rs:99-116) + COUNT(*)
 *   exon_hip_flag_mapq_group_count        sam_flag_function (exon-core/src/udfs/sam/samflags.rs:26-47)
 *        + CAST(mapping_quality AS INT) >= q + COUNT(*) GROUP BY reference over the columns of
 *        BAMArrayBuilder::append (exon-bam/src/array_builder.rs:102-218)
 *   exon_hip_cmp_avg_by_group             info."AF" > lit, AVG(qual), COUNT(*) GROUP BY filter over
 *        LazyVCFArrayBuilder columns (exon-vcf/src/array_builder/lazy_array_builder.rs:205-216,
 *        info_builder.rs:152-309)
 *   exon_hip_qual_pos_hist                QualityScoreStringToList::invoke
 *        (exon-core/src/udfs/sequence/quality_score_string_to_list.rs:56-117) + unnest + GROUP BY
 *   exon_hip_regroup_files_by_size        regroup_files_by_size
 *        (exon-core/src/datasources/exon_file_scan_config.rs:79-110) -- the multi-GPU shard rule
 */
#ifndef EXON_HIP_H
#define EXON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Arrow C Data Interface (verbatim ABI; https://arrow.apache.org/docs/format/CDataInterface.html) */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif
#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11
struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event; /* hipEvent_t* or NULL */
  int64_t reserved[3];
};
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define EXON_HIP_OK 0
#define EXON_HIP_EINVAL (-1)       /* bad argument / schema mismatch / misaligned buffer */
#define EXON_HIP_ENOMEM (-2)
#define EXON_HIP_EDEVICE (-3)      /* HIP runtime error (text in exon_hip_last_error) */
#define EXON_HIP_EUNSUPPORTED (-4) /* valid request this build does not implement */
#define EXON_HIP_ESTATE (-5)       /* call out of order (e.g. push after finish) */
#define EXON_HIP_ECAPACITY (-6)    /* more distinct group keys than the plan's n_groups: the stream's state and dictionary are
                                      unchanged (the offending scan's rows were not added) -- finish the stream, emit its state,
                                      and continue on a fresh one, or create the plan for more groups (round 5; EINVAL before) */

typedef struct exon_hip_ctx exon_hip_ctx;
typedef struct exon_hip_plan exon_hip_plan;
typedef struct exon_hip_stream exon_hip_stream;

/* ---- context ------------------------------------------------------------------------------ */
typedef struct exon_hip_device_info {
  char name[64];
  char gcn_arch[32];
  int32_t compute_units;
  int32_t wavefront_size;
  int64_t hbm_bytes;
  int32_t clock_khz;
  int32_t reserved;
} exon_hip_device_info;

int exon_hip_abi_version(void); /* 5 (round 6: the CONTRACT of exon_hip_stream_push changed in round 5 -- batches of up to 131072 rows are held and
                                      their release callback runs at the slot flush, not before push returns; key overflow is ECAPACITY,
                                      not EINVAL -- so a caller built against 4 must not load this library unchanged; also new: scan
                                      projection, plain-gzip inputs on the device, exon_hip_read_probe, collective fault votes;
                                      round 4: group keys by value -- exon_hip_stream_keys / _set_keys / _reconcile_keys, exon_hip_keys_union,
                                      exon_hip_stream_set_region_contig; round 3: plan_desc.x_type / y_type, 16 typed INFO fields, rccl_comm_count) */
int exon_hip_device_count(int* out);
int exon_hip_ctx_create(int device, exon_hip_ctx** out);
int exon_hip_ctx_destroy(exon_hip_ctx* ctx);
int exon_hip_ctx_info(exon_hip_ctx* ctx, exon_hip_device_info* out);
/* Text of the last error on this ctx (or, with ctx == NULL, of the calling thread).  Owned by the
 * library, valid until the next failing call on the same ctx/thread. */
const char* exon_hip_last_error(const exon_hip_ctx* ctx);

/* Device-memory helpers so a host without its own HIP binding can stage columns.  `stream` is a
 * hipStream_t passed as void* (NULL = the ctx's own stream). */
int exon_hip_malloc(exon_hip_ctx* ctx, size_t bytes, void** dptr);
int exon_hip_free(exon_hip_ctx* ctx, void* dptr);
int exon_hip_memcpy_h2d(exon_hip_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);
int exon_hip_memcpy_d2h(exon_hip_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);
int exon_hip_memset(exon_hip_ctx* ctx, void* dst, int value, size_t bytes, void* stream);
int exon_hip_sync(exon_hip_ctx* ctx, void* stream);
/* Wall time of `reps` back-to-back launches measured with hipEvents on `stream` is available to
 * hosts through these two calls (bench.py uses torch events on the same stream instead). */
/* (ABI 5) What this box streams out of HBM through the access pattern of the fused kernels: `reps` bare read passes over the
 * first `bytes_each` bytes (whole 64 KiB tiles) of 1..4 device buffers walked in lock-step -- same persistent grid, same
 * 16 B/lane non-temporal loads, no predicate and no aggregate -- timed by a HIP event pair on `stream`.  bench.py reports
 * bytes_per_pass / ms_per_pass as roofline.box_read_ceiling_GBps next to the kernel's own figure: boxes of one pool differ by
 * a few percent in what their HBM delivers, and a roofline fraction against the spec peak cannot tell a slow box from a slow
 * kernel.  Measurement aid: not on the reference's path. */
int exon_hip_read_probe(exon_hip_ctx* ctx, void* stream, const void* const* buffers, int32_t n_buffers, int64_t bytes_each,
                        int32_t reps, double* ms_per_pass, int64_t* bytes_per_pass);
int exon_hip_timer_start(exon_hip_ctx* ctx, void* stream);
int exon_hip_timer_stop_ms(exon_hip_ctx* ctx, void* stream, float* ms);

/* ---- device column ------------------------------------------------------------------------ */
/* One Arrow array already resident in HBM.  `values` must be 16-byte aligned, `validity` is the
 * Arrow LSB-first bitmap (NULL = no nulls) whose bit 0 is row 0 of `values` (no bit offset),
 * `offsets` is the int32 offsets buffer of a Utf8/Binary column (NULL otherwise). */
typedef struct exon_hip_column {
  const void* values;
  const uint8_t* validity;
  const int32_t* offsets;
  int64_t length;
} exon_hip_column;

/* comparison operators of exon_hip_cmp_avg_by_group */
#define EXON_HIP_GT 0
#define EXON_HIP_GE 1
#define EXON_HIP_LT 2
#define EXON_HIP_LE 3
#define EXON_HIP_EQ 4
#define EXON_HIP_NE 5

#define EXON_HIP_MAX_REG_GROUPS 8   /* group ids kept in registers; ids 8.. use an LDS overflow table */
#define EXON_HIP_MAX_GROUPS 4096    /* group tables kept in LDS */
#define EXON_HIP_MAX_GROUPS_GLOBAL (1 << 24) /* K4: beyond EXON_HIP_MAX_GROUPS the state arrays are updated with global atomics */
#define EXON_HIP_MAX_REFERENCES (1 << 24) /* K3: beyond EXON_HIP_MAX_GROUPS references the counters are global atomics */
#define EXON_HIP_REGION_OPEN_END INT64_MAX

/* ---- operator launches (asynchronous on `stream`; results ACCUMULATE into the state buffers,
 *      which the caller zeroes once per query: hipMemsetAsync / exon_hip_memset -- or see
 *      exon_hip_plan_launch with EXON_HIP_LAUNCH_OVERWRITE, which needs no zeroing pass) ------ */

/* K2.  d_count[0] += |{ i : chrom_id[i] valid && == region_chrom_id && pos[i] valid &&
 *                           start <= pos[i] <= end }|   (1-based inclusive; Kleene AND, keep TRUE) */
int exon_hip_region_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* chrom_id /*i32*/,
                          const exon_hip_column* pos /*i64*/, int64_t n, int32_t region_chrom_id,
                          int64_t start, int64_t end, int64_t* d_count);

/* K3.  rows with (flag & flag_mask) == flag_value AND mapq valid AND mapq >= mapq_min are counted
 *      into d_counts[ref_id] or, when ref_id is NULL, d_counts[n_refs].  d_counts has n_refs+1. */
int exon_hip_flag_mapq_group_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* flag /*i32*/,
                                   const exon_hip_column* mapq /*u8*/, const exon_hip_column* ref_id /*i32*/,
                                   int64_t n, int32_t flag_mask, int32_t flag_value, int32_t mapq_min,
                                   int32_t n_refs, int64_t* d_counts);

/* K6.  *d_count += |{ i : ref_id[i] = region_ref_id AND start[i] <= region_end AND end[i] >= region_start }|, all three
 *      valid (a missing reference / start / end never matches).  SemiLazyRecord::intersects
 *      (exon-bam/src/indexed_async_batch_stream.rs:66-87) = bam_region_filter.  1-based inclusive. */
int exon_hip_overlap_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* ref_id /*i32*/,
                           const exon_hip_column* start /*i64*/, const exon_hip_column* end /*i64*/, int64_t n,
                           int32_t region_ref_id, int64_t region_start, int64_t region_end, int64_t* d_count);
/* K6, strict form: *d_count += |{ i : ref_id[i] = region_ref_id AND start[i] > after AND end[i] < before }|, all three valid.
 *      The BED / GFF interval predicate: StartEndIntervalPhysicalExpr keeps `start > a` / `end < b` BinaryExprs and evaluates
 *      them as written (exon-core/src/physical_plan/start_end_interval_physical_expr.rs:93-139, 186-191): strictly inside
 *      (after, before).  `start > a` alone: before = INT64_MAX; `end < b` alone: after = 0 (its constructor's default). */
int exon_hip_within_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* ref_id /*i32*/,
                          const exon_hip_column* start /*i64*/, const exon_hip_column* end /*i64*/, int64_t n,
                          int32_t region_ref_id, int64_t after, int64_t before, int64_t* d_count);

/* K4.  For rows with x valid AND (double)x <cmp_op> threshold (f32 widened to f64, as DataFusion
 *      coerces Float32 vs a Float64 literal), per dictionary id g = group_id[i] in [0, n_groups):
 *        d_counts[g]            += (y valid)            -- COUNT(y) / AVG denominator
 *        d_counts[n_groups + g] += 1                    -- COUNT(*)
 *        d_sums[g]              += (double)y if y valid -- AVG numerator
 *      Sums are reduced in a fixed order: bit-reproducible run to run for a given launch shape (up to EXON_HIP_MAX_GROUPS
 *      groups; beyond that, up to EXON_HIP_MAX_GROUPS_GLOBAL, global atomics: exact counts, sums in atomic order). */
int exon_hip_cmp_avg_by_group(exon_hip_ctx* ctx, void* stream, const exon_hip_column* x /*f32*/,
                              const exon_hip_column* y /*f32*/, const exon_hip_column* group_id /*i32*/,
                              int64_t n, double threshold, int32_t cmp_op, int32_t n_groups,
                              int64_t* d_counts /*[2*n_groups]*/, double* d_sums /*[n_groups]*/);

/* K5.  d_hist[p*256 + b] += |{ reads r, p < len(r) : bytes[offsets[r] + p] == b }| for p < lmax;
 *      positions >= lmax are an error (status word, reported by exon_hip_stream_finish / sync). */
int exon_hip_qual_pos_hist(exon_hip_ctx* ctx, void* stream, const exon_hip_column* quality_scores /*utf8*/,
                           int64_t n_reads, int32_t lmax, int64_t* d_hist /*[lmax*256]*/);

/* K5 over a column held as several Arrow batches ("chunks": a Utf8 column has int32 offsets, so a quality column beyond
 * 2^31 bytes IS several batches -- the reference streams 8192-row batches into one accumulator).  chunks[c] is a utf8
 * column of n_reads[c] reads; chunks with n_reads[c] == 0 are skipped.  Equivalent to calling exon_hip_qual_pos_hist per
 * chunk, but up to 64 chunks share ONE offsets scan, ONE main kernel (the per-workgroup LDS histograms live across
 * chunks) and ONE fold instead of three launches each. */
int exon_hip_qual_pos_hist_chunks(exon_hip_ctx* ctx, void* stream, const exon_hip_column* chunks /*utf8 each*/,
                                  int32_t n_chunks, const int64_t* n_reads, int32_t lmax, int64_t* d_hist /*[lmax*256]*/);

/* K5 over independent views: read r is bytes[starts[r] .. ends[r]) of `d_bytes` (all device pointers).  Used for
 * the quality (or sequence) lines of raw FASTQ text resident in HBM (exon_hip_fastq_parser_parse): no Arrow
 * column is materialised. */
int exon_hip_qual_pos_hist_views(exon_hip_ctx* ctx, void* stream, const uint8_t* d_bytes, const int32_t* d_starts,
                                 const int32_t* d_ends, int64_t n_reads, int32_t lmax, int64_t* d_hist /*[lmax*256]*/);

/* ---- synthetic inputs generated in HBM (DESIGN.md "Synthetic inputs"; bit-identical to the
 *      oracle's generators, which the parity tests check) ------------------------------------- */
int exon_hip_gen_c2(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t n_total, int64_t lo, int64_t hi,
                    int32_t* d_chrom_id, int64_t* d_pos);
int exon_hip_gen_c3(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, int32_t* d_flag,
                    uint8_t* d_mapq, uint8_t* d_mapq_valid, int32_t* d_ref_id, uint8_t* d_ref_valid);
int exon_hip_gen_c4(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, float* d_af,
                    uint8_t* d_af_valid, float* d_qual, uint8_t* d_qual_valid, int32_t* d_filter_id);
/* alignments for K6: reference id (+validity), start / end (+ one shared validity; unmapped rows have neither) */
int exon_hip_gen_c6(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, int32_t* d_ref_id,
                    uint8_t* d_ref_valid, int64_t* d_start, int64_t* d_end, uint8_t* d_pos_valid);
int exon_hip_gen_c5(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, int32_t read_len,
                    int32_t* d_offsets, uint8_t* d_bytes);

/* ---- host-side planning helpers ------------------------------------------------------------ */
/* noodles Region grammar `name[:start[-end]]`, 1-based inclusive; end = EXON_HIP_REGION_OPEN_END
 * when open (exon-core/src/physical_plan/infer_region.rs:25-42). */
int exon_hip_parse_region(const char* region, char* name_out, size_t name_cap, int64_t* start, int64_t* end);
/* Whole files, ascending size, dealt round-robin into min(target, n) groups.  group_of[i] = group of
 * file i; returns the number of groups (or a negative status). */
int exon_hip_regroup_files_by_size(const int64_t* sizes, int32_t n_files, int32_t target_groups, int32_t* group_of);

/* ---- plan / stream: the ExecutionPlan-shaped surface -------------------------------------- */
#define EXON_HIP_PLAN_REGION_COUNT 2
#define EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT 3
#define EXON_HIP_PLAN_CMP_AVG_BY_GROUP 4
#define EXON_HIP_PLAN_QUAL_POS_HIST 5
#define EXON_HIP_PLAN_OVERLAP_COUNT 6 /* region_chrom_id / region_start / region_end; columns: ref_id, start, end */
#define EXON_HIP_PLAN_WITHIN_COUNT 7  /* same fields, strict form: start > region_start AND end < region_end */

#define EXON_HIP_X_FLOAT32 0
#define EXON_HIP_X_INT32 1
typedef struct exon_hip_plan_desc {
  int32_t kind;          /* EXON_HIP_PLAN_* */
  int32_t n_groups;      /* K3: n_refs; K4: dictionary size; else 0 */
  /* K2 */
  int32_t region_chrom_id;
  /* K4: type of the compared column x.  EXON_HIP_X_FLOAT32 (0): Float32, widened to Float64 and compared in IEEE totalOrder.
   * EXON_HIP_X_INT32 (1): Int32 -- an INFO field of Type=Integer (exon-core/src/datasources/vcf/schema_builder.rs:197-205);
   * compared exactly (DataFusion: as Int64 against an integer literal, as Float64 against a float literal -- every int32
   * is exact in both), never through f32.  exon_hip_stream_consume_scan takes the type from the file's header instead. */
  int32_t x_type;
  int64_t region_start, region_end;
  /* K3 */
  int32_t flag_mask, flag_value, mapq_min;
  /* K4 */
  int32_t cmp_op;
  double threshold;
  /* K5 */
  int32_t lmax;
  /* K4: type of AVG's argument y, EXON_HIP_X_FLOAT32 / EXON_HIP_X_INT32 (DataFusion's avg casts either to Float64) */
  int32_t y_type;
  /* input column indexes into the batch's children, in operator argument order
   * (K2: chrom_id,pos  K3: flag,mapq,ref_id  K4: x,y,group_id  K5: quality_scores  K6: ref_id,start,end) */
  int32_t columns[4];
} exon_hip_plan_desc;

int exon_hip_plan_create(exon_hip_ctx* ctx, const exon_hip_plan_desc* desc, exon_hip_plan** out);
int exon_hip_plan_destroy(exon_hip_plan* plan);
/* number of int64 / float64 words of the partial-aggregate state of this plan */
int exon_hip_plan_state_size(const exon_hip_plan* plan, int64_t* n_i64, int64_t* n_f64);

/* Stateless form of a push: run the plan's fused kernel over HBM-resident columns (`columns[i]` = operator argument i, see
 * exon_hip_plan_desc.columns for the order; n_columns must match the plan) on the caller's hipStream_t into a caller-owned
 * packed partial state [n_i64 x int64][n_f64 x float64] (exon_hip_plan_state_size; 8-byte aligned device memory).
 * EXON_HIP_LAUNCH_ACCUMULATE: state += this batch.  EXON_HIP_LAUNCH_OVERWRITE: state := this batch -- the first batch of a
 * query then needs no zeroing kernel in front of it (AggregateExec(Partial) starting from empty accumulators). */
#define EXON_HIP_LAUNCH_ACCUMULATE 0
#define EXON_HIP_LAUNCH_OVERWRITE 1
int exon_hip_plan_launch(exon_hip_plan* plan, void* stream, const exon_hip_column* columns, int32_t n_columns, int64_t n,
                         int32_t flags, void* d_state);
/* The same over a table held as n_chunks batches: columns[c * n_columns + i] = operator argument i of chunk c, n[c] its rows.
 * `flags` applies to the whole call (OVERWRITE: state := all chunks).  Plans of kind QUAL_POS_HIST fuse up to 64 chunks per
 * kernel launch (exon_hip_qual_pos_hist_chunks); the other kinds launch once per chunk. */
int exon_hip_plan_launch_chunks(exon_hip_plan* plan, void* stream, const exon_hip_column* columns, int32_t n_columns,
                                int32_t n_chunks, const int64_t* n, int32_t flags, void* d_state);

/* ---- merge of partial states across GPUs (AggregateExec(Final) over RCCL / xGMI; one process per GPU) -------------------
 * The reference merges partitions in AggregateExec(Final) behind a RepartitionExec / CoalescePartitionsExec (DataFusion 44,
 * reached from exon-core/src/session_context/exon_context_ext.rs:297-311); file groups come from regroup_files_by_size
 * (exon-core/src/datasources/exon_file_scan_config.rs:79-110).  Here every rank holds one packed state; the merge is ONE
 * ncclAllGather of it plus a fold in rank order 0..world-1, so all ranks end with bit-identical float64 sums whatever
 * algorithm RCCL picks.  librccl.so is dlopen'ed at the first call. */
/* out[v] = sum over r < world of gathered[r][v]  (gathered = `world` packed states, rank-major; d_out may not alias it) */
int exon_hip_fold_states(exon_hip_ctx* ctx, void* stream, const void* d_gathered, int32_t world, int64_t n_i64,
                         int64_t n_f64, void* d_out);
/* all-gather d_state into d_gather (world x state bytes) on `stream`, fold into d_out (may be d_state).  d_gather == NULL:
 * in-place ncclAllReduce of an integer-only state (n_f64 == 0, d_out == d_state) -- for states too large to gather. */
int exon_hip_merge_states(exon_hip_ctx* ctx, void* stream, void* rccl_comm, void* d_state, int64_t n_i64, int64_t n_f64,
                          void* d_gather, void* d_out);
/* ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy for hosts without an RCCL binding: rank 0 makes the 128-byte id and
 * ships it to the other ranks by any means; every rank then calls comm_init (collective) on its own ctx / GPU. */
/* ---- communicators (ABI 5: the handle is OPAQUE -- made by one of the two calls below, never a bare ncclComm_t) --------------------
 * exon_hip_rccl_comm_init: ncclGetUniqueId / ncclCommInitRank for hosts without an RCCL binding of their own (rank 0 creates the id,
 * ships its 128 bytes to the other ranks by any means; one process per GPU).  exon_hip_comm_from_callbacks: the same collectives over
 * an all-gather of HOST buffers the caller supplies (torch.distributed / gloo, MPI ...): `all_gather(user, send, recv, bytes)` copies
 * `bytes` from every rank's `send` into recv[rank * bytes ...] on every rank and returns 0 -- for hosts that already own a transport,
 * and for machines where RCCL cannot form the communicator (several ranks on one GPU: tests/test_collective_faults.py).
 * Every collective entry point of a stream VOTES before it exchanges data: a rank that cannot go on (stream finished, undeclared keys,
 * out of memory) says so in a 16-byte all-gather and EVERY rank returns that error -- none is left waiting in a collective for a peer
 * that has returned.  A peer that never enters at all is met by a bounded wait (EXON_HIP_COLLECTIVE_TIMEOUT_S, default 120):
 * the RCCL communicator is aborted (ncclCommAbort), the call fails with EXON_HIP_EDEVICE, the handle is unusable afterwards.
 * EXON_HIP_FAULT="site@rank[,...]" makes a named site fail on one rank (reconcile_enter, reconcile_malloc, reconcile_rekey,
 * allreduce_enter, allreduce_malloc, stall): how the tests reach every one of those paths. */
typedef int (*exon_hip_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
int exon_hip_comm_from_callbacks(int32_t world, int32_t rank, exon_hip_allgather_fn all_gather, void* user, void** comm);
/* a host that made its ncclComm_t itself wraps it (not owned: exon_hip_rccl_comm_destroy frees the handle, not the ncclComm_t) */
int exon_hip_comm_wrap_rccl(void* nccl_comm, void** comm);
int exon_hip_rccl_unique_id(uint8_t* id128);
int exon_hip_rccl_comm_init(exon_hip_ctx* ctx, const uint8_t* id128, int32_t world, int32_t rank, void** rccl_comm);
int exon_hip_rccl_comm_destroy(void* rccl_comm);
/* ranks of a communicator and this one's index (ncclCommCount / ncclCommUserRank for the RCCL kind; rank may be NULL): what a
 * launcher prints to prove how many GPUs merged */
int exon_hip_rccl_comm_count(void* rccl_comm, int32_t* world, int32_t* rank);

/* One stream per partition (= per file group); single-threaded handle, owns one HIP stream, a
 * device-resident partial state and pinned staging buffers. */
int exon_hip_stream_open(exon_hip_plan* plan, int32_t partition, exon_hip_stream** out);
/* Host Arrow struct batch (children = columns).  The batch is MOVED (Arrow C Data Interface: on return `batch->release` is
 * NULL and the stream owns the array): the library calls the array's release callback exactly once, after its buffers have
 * been copied -- for a batch of up to 131072 rows of fixed-width columns that may be AFTER this call returns: such batches are
 * held and copied together, by a pool of threads, when their staging slot is flushed (the next push that fills it, sync,
 * state, finish, reset or close).  Everything the array points to must therefore stay valid until the callback runs, as the
 * interface requires of any exported array (an array whose children or buffer tables live on the caller's stack is not one).
 * EXON_HIP_HOLD_SMALL_BATCHES=0: copy and release inside the call.  Dictionary-encoded int32 indices are accepted for id
 * columns (the dictionary itself stays with the caller). */
int exon_hip_stream_push(exon_hip_stream* s, struct ArrowArray* batch);
/* HBM-resident batch (device_type must be ARROW_DEVICE_ROCM on this ctx's device).  Not moved:
 * buffers must stay alive until the next exon_hip_stream_sync/finish. */
int exon_hip_stream_push_device(exon_hip_stream* s, const struct ArrowDeviceArray* batch);
/* Device pointers of the partial state ([n_i64] int64 then [n_f64] float64, one allocation) so the
 * host can all-reduce them over RCCL, plus the hipStream_t the kernels run on. */
int exon_hip_stream_state(exon_hip_stream* s, int64_t** d_i64, double** d_f64, void** hip_stream);
/* The merge across GPUs in native code on the stream's hipStream_t: afterwards the state of every rank is the sum over all
 * ranks.  One all-gather + fixed-order fold (exon_hip_merge_states); `rccl_comm` is a communicator of this library (one rank
 * per GPU: exon_hip_rccl_comm_init).
 * (ABI 5) Collective, and it FAILS TOGETHER: what stops one rank -- the stream finished, a state keyed by its own dictionary
 * (EXON_HIP_ESTATE: call exon_hip_stream_reconcile_keys first wherever keys come from files), staged rows that cannot be
 * launched, no memory for the gather buffer -- goes into a 16-byte vote in front of the merge, and every rank returns that
 * error without entering the data exchange.  (Until ABI 4 these checks were rank-local and the peers of a failing rank waited
 * in ncclAllGather for ever.)  The vote synchronises the stream; the merge behind it is enqueued.  exon_hip_merge_states is
 * the bare form for callers that own the buffers and the step budget (bench.py's timed region). */
int exon_hip_stream_all_reduce(exon_hip_stream* s, void* rccl_comm);
/* ---- group keys by VALUE (ABI 4) --------------------------------------------------------------------------------------------
 * K3 / K4 states are indexed by dictionary id, and ids are per FILE: FILTER lists are numbered in order of first appearance in
 * a scan, reference ids follow each BAM's own @SQ order.  The reference merges partitions by key VALUE (AggregateExec(Final)
 * over the file groups of exon-core/src/datasources/exon_file_scan_config.rs:79-110).  So a stream fed by
 * exon_hip_stream_consume_scan remembers the value behind every state index: the first scan's dictionary becomes the stream's,
 * every further scan is aggregated under its own ids and added in under the stream's (new values are appended; more distinct
 * values than the plan's n_groups is EXON_HIP_ECAPACITY and leaves the stream as it was).  Across ranks, exon_hip_stream_all_reduce REFUSES (EXON_HIP_ESTATE) a
 * state keyed by a rank-local dictionary until the ranks have agreed on one -- exon_hip_stream_reconcile_keys, or
 * exon_hip_stream_keys on every rank + exon_hip_keys_union + exon_hip_stream_set_keys when the host moves the names itself.
 * Names are packed as '\0'-terminated strings back to back ("" is a legal key: the empty FILTER list); K3's NULL-reference
 * group is not a key (it stays the last word of the state).  Streams that are only ever pushed to (caller-owned dictionary
 * ids) are not tracked; exon_hip_stream_set_keys lets such a caller declare what its ids mean. */
/* the stream's dictionary in state-index order: *n_keys names, *bytes in all (buf may be NULL to size it);
 * *agreed (optional) = 1 once set_keys / reconcile_keys has run and no scan was consumed since */
int exon_hip_stream_keys(exon_hip_stream* s, char* buf, size_t cap, int32_t* n_keys, size_t* bytes, int32_t* agreed);
/* adopt `names` (n_keys packed strings, no duplicates, at most the plan's n_groups): every key the stream holds must be among
 * them; the state is permuted into the new order on the device.  On a stream without keys: a declaration / a seed. */
int exon_hip_stream_set_keys(exon_hip_stream* s, const char* packed_names, size_t packed_bytes, int32_t n_keys);
/* host-only: union of `world` dictionaries (packed back to back, n_keys[r] names each) in rank order, first appearance first --
 * deterministic, so every rank computes the same.  out (may be NULL to size it) receives the packed union, maps (optional,
 * sum of n_keys entries, rank-major) the union id of every input key. */
int exon_hip_keys_union(const char* packed, size_t packed_bytes, const int32_t* n_keys, int32_t world, char* out, size_t cap,
                        int32_t* n_out, size_t* out_bytes, int32_t* maps);
/* collective over `rccl_comm` (every rank calls it, between its last consume_scan and exon_hip_stream_all_reduce): small
 * all-gathers move the dictionaries, every rank forms the same union and permutes its state into it.  A rank that cannot take
 * part (its stream finished; rows pushed under undeclared ids) says so INSIDE the first exchange, one that runs out of memory for
 * the exchange or the re-keying buffers says so in a vote in front of the second: every rank then returns that error (EXON_HIP_ESTATE
 * / EXON_HIP_ENOMEM), none hangs, no state has been touched.  A union larger than the plan's n_groups is EXON_HIP_ECAPACITY on
 * every rank (all ranks compute the same union). */
int exon_hip_stream_reconcile_keys(exon_hip_stream* s, void* rccl_comm);
/* Region plans (K2 / K6 / K7) fed by files: name the contig instead of fixing exon_hip_plan_desc.region_chrom_id -- every
 * exon_hip_stream_consume_scan then resolves the name in that file's own contig / reference dictionary (a BAM without such
 * a reference contributes no rows). */
int exon_hip_stream_set_region_contig(exon_hip_stream* s, const char* name);
/* Start a new query on this stream: the next launch DEFINES the state (overwrite mode, no zeroing kernel); rows staged but
 * not yet launched are dropped; a finished stream can be pushed to again. */
int exon_hip_stream_reset(exon_hip_stream* s);
int exon_hip_stream_sync(exon_hip_stream* s);
/* Copies the (possibly all-reduced) state to host: counts[n_i64], sums[n_f64]. */
int exon_hip_stream_finish(exon_hip_stream* s, int64_t* counts, double* sums);
/* Same result as one Arrow struct array in DataFusion's partial-aggregate state layout
 * (see DESIGN.md "State schema"); caller releases out/out_schema. */
int exon_hip_stream_finish_arrow(exon_hip_stream* s, struct ArrowArray* out, struct ArrowSchema* out_schema);
int exon_hip_stream_close(exon_hip_stream* s);

/* ---- scan: native decoders + device-layout array builders (host memory) --------------------------------
 * FileOpener::open + BatchReader::read_batch of the reference
 * (exon-core/src/datasources/vcf/file_opener/unindex_file_opener.rs:48-92, exon-vcf/src/async_batch_stream.rs:80-109,
 *  exon-bam/src/batch_reader.rs:44-108, exon-fastq/src/batch_reader.rs:63-82, exon-fasta/src/batch_reader.rs:72-99).
 * Batches are Arrow struct arrays in the DEVICE LAYOUT (dictionary<int32,utf8> keys, u8 mapq) ready for
 * exon_hip_stream_push.  Column order:
 *   VCF   0 chrom(dict) 1 pos:i64? 2 qual:f32? 3 filter(dict of ';'-joined lists, "" = []) [4.. info.<F>: f32? | bool? | dict?]
 *   BAM   0 flag:i32 1 mapping_quality:u8? 2 reference(dict)? 3 start:i64? 4 end:i64?
 *   FASTQ 0 name 1 description? 2 sequence 3 quality_scores      FASTA 0 id 1 description? 2 sequence
 * CPU-only: no ctx needed; errors are reported through exon_hip_last_error(NULL). */
#define EXON_HIP_FORMAT_VCF 1
#define EXON_HIP_FORMAT_BAM 2
#define EXON_HIP_FORMAT_FASTQ 3
#define EXON_HIP_FORMAT_FASTA 4
#define EXON_HIP_FORMAT_SAM 5 /* text SAM: same columns as BAM */
#define EXON_HIP_FORMAT_BCF 6 /* BCF2: same columns as VCF */
#define EXON_HIP_FORMAT_CRAM 7 /* CRAM 3.0 (raw / gzip / bzip2 / lzma / rANS 4x8 blocks), host decoder: same columns as BAM */
#define EXON_HIP_COMPRESSION_AUTO 0 /* sniff the gzip/BGZF magic */
#define EXON_HIP_COMPRESSION_NONE 1
#define EXON_HIP_COMPRESSION_GZIP 2
/* zstd / bzip2 / xz inputs (file_compression_type.convert_stream of the reference's openers) are recognised by their magic
 * numbers and refused by exon_hip_scan_open with EXON_HIP_EUNSUPPORTED and a text naming the codec. */

typedef struct exon_hip_scan exon_hip_scan;
typedef struct exon_hip_scan_options {
  int32_t format;       /* EXON_HIP_FORMAT_* */
  int32_t compression;  /* EXON_HIP_COMPRESSION_* */
  int64_t batch_size;   /* 0 = 8192 (exon-common/src/lib.rs:27) */
  const char* info_field; /* VCF / BCF: typed INFO fields to extract (exon.vcf_parse_info), up to 16 comma-separated header
                             IDs ("AF,DP,DB"), NULL = none.  They become scan columns 4, 5, ..., typed from the header as the
                             reference does (schema_builder.rs:197-249): Number=1 Float -> f32?, Number=1 Integer -> i32?,
                             Flag -> bool? (true when present, NULL when absent), Number=1 String / Character -> dictionary? (VCF text: built on the device too),
                             any other Number -> List<f32 | i32 | dictionary>? (items '.' are NULL items; host decoders);
                             INFO '.' makes all of them NULL (the struct itself is NULL in the reference) */
  const char* region;     /* pushed-down vcf_region_filter / bam_region_filter ("chr1:1-100"), NULL = none */
  int32_t use_index;      /* with `region`: plan BGZF chunks from <path>.tbi / <path>.bai (INDEXED_VCF / INDEXED_BAM) */
  int32_t gpu_parse;      /* VCF, BCF, FASTQ, BAM, SAM: exon_hip_stream_consume_scan ships the file's bytes to HBM and decodes
                             them on the GPU (exon_hip_bgzf_inflate, exon_hip_vcf_parser_* / exon_hip_fastq_parser_* /
                             exon_hip_bam_parser_* ...).  exon_hip_scan_next on such a scan needs exon_hip_scan_bind_ctx first
                             (batches then come out of the same GPU pipeline); without a bound ctx it returns ESTATE */
  uint64_t projection;    /* (ABI 5) EXON_HIP_PROJECT_* bits: columns of the reference's schema beyond the fused kernels' operands,
                             appended BEHIND the default ones in bit order (0 = none: the hot path ships only what a plan reads).
                             VCF text: id List<Utf8>?, ref Utf8, alt List<Utf8>? (lazy_array_builder.rs:169-205; the alt list has no
                             items, as in the reference -- see text_columns.hip); BAM: name Utf8?, cigar Utf8, sequence Utf8,
                             quality_score List<Int64> (exon-bam/src/array_builder.rs:105-201).  Built by the host readers (one
                             thread) and by the GPU pipeline (exon_hip_scan_bind_ctx); EXON_HIP_EUNSUPPORTED for other formats.
                             VCF text also: info Utf8, formats Utf8 -- the unparsed forms of the reference's default schema, which
                             are the parsed entries PRINTED AGAIN, not the fields' bytes ("AF=0.50" -> "AF=0.5", a Flag ->
                             "DB=true", formats = keys TAB samples; lazy_array_builder.rs:216-297, :310-423; host/vcf_text.h).
                             These two are built by the host reader only: a gpu_parse scan that asks for them decodes on the host.
                             BCF: id List<Utf8>, ref Utf8, alt List<Utf8> through the reference's EAGER builder (lists with their
                             items, never NULL: eager_array_builder.rs:112-134); SAM: the BAM columns from the line's fields
                             (exon-sam/src/array_builder.rs:101-185); both from the host readers and from the GPU pipeline */
} exon_hip_scan_options;
#define EXON_HIP_PROJECT_VCF_ID 1ull
#define EXON_HIP_PROJECT_VCF_REF 2ull
#define EXON_HIP_PROJECT_VCF_ALT 4ull
#define EXON_HIP_PROJECT_VCF_INFO 8ull
#define EXON_HIP_PROJECT_VCF_FORMATS 16ull
#define EXON_HIP_PROJECT_BAM_NAME 1ull
#define EXON_HIP_PROJECT_BAM_CIGAR 2ull
#define EXON_HIP_PROJECT_BAM_SEQUENCE 4ull
#define EXON_HIP_PROJECT_BAM_QUALITY_SCORES 8ull

int exon_hip_scan_open(const char* path, const exon_hip_scan_options* options, exon_hip_scan** out);
int exon_hip_scan_schema(exon_hip_scan* scan, struct ArrowSchema* out);
/* 0 = a batch was written to *out (caller releases or moves it); 1 = end of stream; <0 = error */
int exon_hip_scan_next(exon_hip_scan* scan, struct ArrowArray* out);
/* dictionary of a dict-encoded column (VCF 0/3, BAM 2): current size, and id of `name` (interned if new) */
int exon_hip_scan_dictionary_size(exon_hip_scan* scan, int32_t column, int32_t* size);
int exon_hip_scan_dictionary_intern(exon_hip_scan* scan, int32_t column, const char* name, int32_t* id);
int exon_hip_scan_dictionary_value(exon_hip_scan* scan, int32_t column, int32_t id, const char** name);
int exon_hip_scan_rows(exon_hip_scan* scan, int64_t* rows_emitted);
/* (round 5, additive) Batches of a scan opened with gpu_parse = 1 come out of the GPU decode pipeline on `ctx`: after this
 * call exon_hip_scan_next drives the same slab pipeline exon_hip_stream_consume_scan drives (file bytes to HBM, BGZF inflate,
 * record parse, pushed-down region mask on the device), copies every slab's columns back and returns them as batch_size-row
 * batches in the host readers' layout and file order -- for queries that do NOT end in one of the fused kernels (the surface
 * of <Fmt>Scan::execute, exon/exon-core/src/datasources/vcf/scanner.rs:142-162, bam/scanner.rs:138-158).  A producer thread
 * fills a bounded queue; if the device cannot decide a record the host reader takes over behind the last row emitted.
 * VCF / BCF (chrom, pos, qual, filter, Float / Integer / Flag INFO keys), BAM, SAM, and (round 6) FASTQ -- name, description,
 * sequence, quality_scores as Utf8 columns built on the device (exon-fastq/src/array_builder.rs:68-102), plain, gzip or BGZF;
 * EXON_HIP_EUNSUPPORTED for other formats and for scans whose batches the host reader must build (list-valued INFO keys, the
 * info / formats text columns).  Call before the first
 * exon_hip_scan_next; `ctx` must outlive the scan. */
int exon_hip_scan_bind_ctx(exon_hip_scan* scan, exon_hip_ctx* ctx);
/* number of index chunks an indexed scan planned (-1 when the scan is not index-driven) */
int exon_hip_scan_index_chunks(exon_hip_scan* scan, int32_t* n_chunks);
/* Region -> BGZF chunks from a tabix (.tbi, is_bai = 0; `ref_name` resolved through the index' names) or BAI
 * (.bai, is_bai = 1; `ref_id` = BAM header order) index: get_byte_range_for_file
 * (exon-core/src/datasources/indexed_file/indexed_bgzf_file.rs:52-112).  Virtual positions are written to
 * starts/ends (up to `cap`); *n_chunks receives the full count. */
int exon_hip_index_query(const char* index_path, int32_t is_bai, const char* ref_name, int32_t ref_id, int64_t start,
                         int64_t end, uint64_t* starts, uint64_t* ends, int32_t cap, int32_t* n_chunks);
/* After exon_hip_stream_consume_scan: *decoded = 1 when every record was decoded on the GPU (0: the host decoder ran,
 * because gpu_parse was off or the device handed the file back); *inflated (optional) = 1 when the BGZF blocks were
 * inflated on the GPU as well. */
int exon_hip_scan_decoded_on_gpu(exon_hip_scan* scan, int32_t* decoded, int32_t* inflated);
int exon_hip_scan_close(exon_hip_scan* scan);
/* ---- (ABI 5) plain gzip on the GPU: RFC 1952 members that are NOT BGZF (no "BC" extra field, one DEFLATE stream per member) -------
 * The `else` arm of the reference's openers (exon-core/src/datasources/fastq/file_opener.rs:79-92: a gzip file that fails
 * is_bgzip_valid_header goes through file_compression_type.convert_stream).  The compressed bytes of a slab are cut into chunks, one
 * wavefront per chunk finds a block start by itself and decodes with the 32 KiB in front of it unknown (16-bit symbols: a byte, or
 * "byte k of the window"); the host proves the chain of chunks, the windows are resolved by composing the chunks' tail maps, a last
 * pass writes bytes; CRC-32 and ISIZE of every member are checked (exon_amd/csrc/gzip_stream.hip).
 * A stream is one file: create, decode slab after slab, destroy.  Any failure (EXON_HIP_EINVAL with a text) means: inflate this
 * file on the host -- nothing of the failing call has been committed to the stream's state except that it cannot be continued. */
typedef struct exon_hip_gzip_stream exon_hip_gzip_stream;
typedef struct exon_hip_gzip_stats {
  uint64_t calls, chunks, repairs, overflow_retries, members, comp_bytes, out_bytes;
} exon_hip_gzip_stats;
/* max_comp_bytes: the most compressed bytes one decode call is given; scratch_bytes: symbol scratch (2 bytes per output byte of a
 * call; 0 = 16 x max_comp_bytes, at least 64 MiB -- a call whose chunks overflow their share is repeated on a quarter of the slab) */
int exon_hip_gzip_stream_create(exon_hip_ctx* ctx, int64_t max_comp_bytes, int64_t scratch_bytes, exon_hip_gzip_stream** out);
/* d_comp: device memory (any alignment; the up to 3 bytes in front of it down to the 4-byte boundary must be readable), n_comp bytes
 * that start at the byte the previous call stopped in (the file's first byte for the first call) with 4096 readable bytes behind them; final_input: these are the file's last bytes.  Decodes whole DEFLATE blocks
 * while their output fits out_cap; *consumed = compressed bytes used up (pass the rest again in front of the next bytes; the bit
 * position inside the first byte is the stream's business), *produced = bytes written to d_out, *stream_end = 1 once the last
 * member's trailer has been checked at the end of the input.  Synchronises `stream`. */
int exon_hip_gzip_stream_decode(exon_hip_gzip_stream* s, void* stream, const uint8_t* d_comp, int64_t n_comp, int32_t final_input,
                                uint8_t* d_out, int64_t out_cap, int64_t* consumed, int64_t* produced, int32_t* stream_end);
int exon_hip_gzip_stream_get_stats(exon_hip_gzip_stream* s, exon_hip_gzip_stats* out);
int exon_hip_gzip_stream_destroy(exon_hip_gzip_stream* s);
/* ---- VCF record parsing on the GPU (raw text in HBM -> device-layout columns in HBM) ---------------------------
 * Same field rules as LazyVCFArrayBuilder::append (exon-vcf/src/array_builder/lazy_array_builder.rs:159-216) for the
 * device-layout columns.  A parser is bound to one header (contig dictionary) and one optional typed INFO field and
 * keeps the FILTER dictionary it discovers across slabs.  Rows the device cannot decide (float with > 19 significant
 * digits, contig missing from the header, malformed line) are counted in n_undecided: re-decode that slab on the host. */
#define EXON_HIP_MAX_INFO_FIELDS 16 /* typed INFO fields per scan / parser */
typedef struct exon_hip_vcf_parser exon_hip_vcf_parser;
typedef struct exon_hip_vcf_columns {
  int64_t n_rows;
  int64_t n_undecided;
  int32_t* chrom_id;   /* device pointers owned by the parser, overwritten by the next parse call */
  int64_t* pos;
  uint8_t* pos_valid;
  float* qual;
  uint8_t* qual_valid;
  int32_t* filter_id;
  float* info;         /* NULL without an INFO field; = infos[0] */
  uint8_t* info_valid;
  int64_t consumed_bytes; /* bytes up to and including the last newline; a trailing partial line is not parsed */
  /* all typed INFO fields of the parser, in the order given to *_parser_create (scan columns 4 ..), 4-byte values + validity:
   * info_kinds[k] = 'f' Float -> float values; 'i' Integer -> int32_t values (exact; schema_builder.rs:197-205); 'b' Flag ->
   * infos[k] == NULL, the column is the presence bitmap infos_valid[k] (true where set) */
  int32_t n_info;
  int32_t reserved;
  void* infos[EXON_HIP_MAX_INFO_FIELDS];
  uint8_t* infos_valid[EXON_HIP_MAX_INFO_FIELDS];
  char info_kinds[EXON_HIP_MAX_INFO_FIELDS];
  /* list-valued fields, info_kinds[k] = 'F' (List<f32>) / 'I' (List<i32>) -- any Number other than 0 / 1
   * (schema_builder.rs:235-249): Arrow List layout.  infos_valid[k] = validity of the LIST per row (NULL list: key absent,
   * `key=.`, INFO '.'), list_offsets[k] = int32 offsets [n_rows + 1] into the items, infos[k] = the items,
   * list_item_valid[k] = validity of the items (a '.' item is a NULL item: info_builder.rs:258-305).  NULL for scalar kinds. */
  int32_t* list_offsets[EXON_HIP_MAX_INFO_FIELDS];
  uint8_t* list_item_valid[EXON_HIP_MAX_INFO_FIELDS];
  /* (ABI 5) info_kinds[k] = 's' (Number=1 String / Character, VCF text only): infos[k] = int32 dictionary ids, infos_valid[k] =
   * validity; the dictionary is built on the device like the FILTER one: exon_hip_vcf_parser_info_values.  info_nulls[k] = rows
   * of this slab without a value (-1 for other kinds): a GROUP BY over the key has a NULL group unless this is 0. */
  int32_t info_nulls[EXON_HIP_MAX_INFO_FIELDS];
} exon_hip_vcf_columns;
/* info_field: NULL, or up to EXON_HIP_MAX_INFO_FIELDS comma-separated INFO keys "name[:kind]" -- kind f (default): Number=1
 * Float -> f32; kind i: Number=1 Integer -> i32; kind b: Flag -> presence bitmap; kinds F / I: list-valued Float / Integer
 * fields -> List<f32> / List<i32>.  (String INFO fields, scalar or list, are decoded by the host reader only.) */
int exon_hip_vcf_parser_create(exon_hip_ctx* ctx, const char* const* contig_names, int32_t n_contigs,
                               const char* info_field, int64_t max_slab_bytes, exon_hip_vcf_parser** out);
/* d_text: a slab of '\n'-terminated data lines in HBM, any alignment (a trailing partial line is left to the
 * caller: see consumed_bytes).  Synchronises `stream`. */
int exon_hip_vcf_parser_parse(exon_hip_vcf_parser* parser, void* stream, const uint8_t* d_text, int64_t n_bytes,
                              exon_hip_vcf_columns* cols);
/* FILTER dictionary discovered so far, '\0'-separated in id order ("" = the empty list) */
int exon_hip_vcf_parser_filters(exon_hip_vcf_parser* parser, char* buf, size_t cap, int32_t* n_filters);
/* values of the String / Character INFO key `key` (index in the parser's key list) seen so far, '\0'-separated, in id order */
int exon_hip_vcf_parser_info_values(exon_hip_vcf_parser* parser, int32_t key, char* buf, size_t cap, int32_t* n_values);
/* on != 0: rows WITHOUT a value of a String / Character key take the dictionary id of the EMPTY text (a value no row can carry:
 * "key=" is a missing value) and info_nulls stays 0 -- NULL becomes a group key of its own, which is what DataFusion's GROUP BY
 * does with a nullable key.  exon_hip_stream_consume_scan switches it on (a fused plan cannot skip rows by a key's bitmap);
 * batches (exon_hip_scan_next) keep NULL as NULL.  In a keyed stream's dictionary the NULL group is the key "". */
int exon_hip_vcf_parser_set_null_key(exon_hip_vcf_parser* parser, int32_t on);
int exon_hip_vcf_parser_destroy(exon_hip_vcf_parser* parser);

/* ---- BGZF inflate on the GPU (compressed blocks in HBM -> inflated bytes in HBM) ------------------------------------
 * Replaces noodles bgzf::AsyncReader around the byte stream (exon-core/src/datasources/vcf/file_opener/
 * unindex_file_opener.rs:62-70, fastq/file_opener.rs:69-75, streaming_bgzf.rs:56-64): RFC 1951 DEFLATE in RFC 1952
 * members carrying the BGZF "BC" extra field (SAM specification 4.1).  Blocks are independent: one wavefront each. */
typedef struct exon_hip_bgzf_block {
  uint32_t comp_offset; /* of the raw DEFLATE data inside the compressed buffer */
  uint32_t comp_size;
  uint32_t out_offset;  /* where the inflated bytes go inside the output buffer */
  uint32_t out_size;    /* ISIZE */
  uint32_t crc32;
  uint32_t reserved;
} exon_hip_bgzf_block;
/* Host: walk the block headers of `data[0..n)`.  Fills up to `cap` entries (blocks may be NULL to count only);
 * out_offset starts at `out_base` and advances by ISIZE.  *consumed = bytes of whole blocks walked (a trailing
 * partial block is left for the next call), *out_bytes = sum of ISIZE. */
int exon_hip_bgzf_scan(const uint8_t* data, size_t n, size_t out_base, exon_hip_bgzf_block* blocks, int32_t cap,
                       int32_t* n_blocks, size_t* consumed, size_t* out_bytes);
/* Device: inflate `n_blocks` blocks (table in host memory) from d_comp (4-byte aligned, with >= 4 KiB of readable
 * padding behind the last block) into d_out (4-byte aligned); verify_crc != 0 also checks every block's CRC-32.  Synchronises
 * `stream`.  A corrupt block -> EXON_HIP_EINVAL and *first_bad_block (else -1): inflate on the host instead. */
int exon_hip_bgzf_inflate(exon_hip_ctx* ctx, void* stream, const uint8_t* d_comp, const exon_hip_bgzf_block* blocks,
                          int32_t n_blocks, uint8_t* d_out, int32_t verify_crc, int32_t* first_bad_block);
/* The lane-parallel decoder keeps ~0.94 MiB of scratch per resident workgroup in a pool per (device, stream) -- up to ~1.4 GB
 * for a stream that inflated a 1536-member launch -- until the stream's owner lets go: streams the library owns do that
 * themselves; for a caller-owned `stream` passed to exon_hip_bgzf_inflate call this once the stream is idle. */
int exon_hip_bgzf_forget_stream(void* stream);
/* Diagnostics of the lane-parallel block decoder (HISTORY.md section 7e; used for launches of up to 1536 members unless
 * EXON_HIP_INFLATE_PAR says otherwise: 0 = never, 1 = always, 2 = side by side with the serial kernel): out32[0] = DEFLATE blocks
 * it decoded on the current device since the process started, out32[1..15] = blocks it handed back to the serial symbol loop, by
 * reason.  `stream` is ignored.  Synchronises the device. */
int exon_hip_bgzf_inflate_par_stats(void* stream, uint32_t* out32);

/* ---- FASTQ record splitting on the GPU (raw text in HBM -> per-read views into that text) ---------------------
 * Record rules of exon-fastq/src/batch_reader.rs:63-82 (noodles fastq: '@' definition line, sequence line, '+'
 * line, quality line; "\r\n" accepted).  The slab may end anywhere: consumed_bytes tells how many bytes form
 * whole records, the rest is carried into the next slab by the caller.  n_undecided != 0 (a record not starting
 * with '@', a missing '+' line, more lines than the parser was sized for, a partial record in the final slab):
 * decode that input on the host instead. */
typedef struct exon_hip_fastq_parser exon_hip_fastq_parser;
typedef struct exon_hip_fastq_views {
  int64_t n_reads;
  int64_t n_undecided;
  int64_t consumed_bytes;
  const int32_t* seq_start; /* device pointers owned by the parser, overwritten by the next parse call */
  const int32_t* seq_end;
  const int32_t* qual_start;
  const int32_t* qual_end;
  const uint8_t* text_base; /* what the views index: the 16-byte aligned address at or below d_text */
  const int32_t* head_start; /* (appended in round 6) the header line behind its '@' ... */
  const int32_t* head_end;   /* ... up to its end, CR dropped: name [space description] (exon-fastq/src/array_builder.rs:68-102) */
} exon_hip_fastq_views;
int exon_hip_fastq_parser_create(exon_hip_ctx* ctx, int64_t max_slab_bytes, exon_hip_fastq_parser** out);
/* d_text: any alignment, < 2 GiB.  final_slab != 0: the text ends the input (it must end with '\n').
 * Synchronises `stream`. */
int exon_hip_fastq_parser_parse(exon_hip_fastq_parser* parser, void* stream, const uint8_t* d_text, int64_t n_bytes,
                                int32_t final_slab, exon_hip_fastq_views* views);
int exon_hip_fastq_parser_destroy(exon_hip_fastq_parser* parser);

/* ---- BAM record splitting + field extraction on the GPU (inflated BAM bytes in HBM -> device-layout columns) ----
 * Column rules of BAMArrayBuilder::append (exon-bam/src/array_builder.rs:102-218) for flag, mapping_quality (NULL when
 * 255), reference id (NULL when -1), start = pos + 1 and end = start + reference length - 1 (NULL when pos < 0).
 * d_data must start at a record boundary; the chains of 64 KiB segments are walked in parallel from guessed record
 * starts and the guesses are proven by matching every chain's end with the next segment's start.  n_undecided != 0
 * (a proof failed, a record larger than a segment, a malformed record): decode on the host instead. */
typedef struct exon_hip_bam_parser exon_hip_bam_parser;
typedef struct exon_hip_bam_columns {
  int64_t n_rows;
  int64_t n_undecided;
  int64_t consumed_bytes; /* whole records; a record cut off by the end of the slab is left to the caller */
  int32_t* flag;          /* device pointers owned by the parser, overwritten by the next parse call */
  uint8_t* mapq;
  uint8_t* mapq_valid;
  int32_t* ref_id;
  uint8_t* ref_valid;
  int64_t* start;
  int64_t* end;
  uint8_t* pos_valid;     /* validity of start and end */
} exon_hip_bam_columns;
int exon_hip_bam_parser_create(exon_hip_ctx* ctx, int32_t n_references, int64_t max_slab_bytes, exon_hip_bam_parser** out);
/* Synchronises `stream`. */
int exon_hip_bam_parser_parse(exon_hip_bam_parser* parser, void* stream, const uint8_t* d_data, int64_t n_bytes,
                              exon_hip_bam_columns* cols);
int exon_hip_bam_parser_destroy(exon_hip_bam_parser* parser);

/* ---- SAM text parsing on the GPU (alignment lines in HBM -> the BAM device layout; `cols` is the BAM column struct) ----
 * Columns of exon-sam (schema_builder.rs:371-402, same as BAM): RNAME through the @SQ order ('*' / unknown -> NULL),
 * POS 0 -> NULL, MAPQ 255 -> NULL, end from the CIGAR.  d_text: '\n'-terminated alignment lines (no header), any
 * alignment; a trailing partial line is left to the caller (consumed_bytes). */
typedef struct exon_hip_sam_parser exon_hip_sam_parser;
int exon_hip_sam_parser_create(exon_hip_ctx* ctx, const char* const* reference_names, int32_t n_references,
                               int64_t max_slab_bytes, exon_hip_sam_parser** out);
int exon_hip_sam_parser_parse(exon_hip_sam_parser* parser, void* stream, const uint8_t* d_text, int64_t n_bytes,
                              exon_hip_bam_columns* cols);
int exon_hip_sam_parser_destroy(exon_hip_sam_parser* parser);

/* ---- BCF2 record splitting + field extraction on the GPU (inflated BCF bytes in HBM -> the VCF device layout) ----
 * Same Arrow schema as VCF (exon-core/src/datasources/bcf/, exon-bcf); records are found like BAM's (parallel chain
 * walk, guessed starts proven by induction) and decoded one thread each: typed-value walk to FILTER (interned as a list
 * of header-string indexes) and one typed INFO key (dictionary index; Float or Integer value -> f32, missing -> NULL).
 * `cols` is the VCF column struct (pos_valid is NULL: POS is a fixed field). */
typedef struct exon_hip_bcf_parser exon_hip_bcf_parser;
int exon_hip_bcf_parser_create(exon_hip_ctx* ctx, int32_t n_contigs, int32_t n_header_strings, int32_t n_samples,
                               int32_t info_key /* header-string index of the INFO field, -1 = none */, int64_t max_slab_bytes,
                               exon_hip_bcf_parser** out);
/* several typed INFO fields instead of the single key of _create: header-string indexes + kinds ('f' Float -> f32, 'i' Integer
 * -> i32, 'b' Flag -> presence bitmap), up to EXON_HIP_MAX_INFO_FIELDS; call before the first parse */
int exon_hip_bcf_parser_set_info_keys(exon_hip_bcf_parser* parser, const int32_t* keys, const char* kinds, int32_t n);
int exon_hip_bcf_parser_parse(exon_hip_bcf_parser* parser, void* stream, const uint8_t* d_data, int64_t n_bytes,
                              exon_hip_vcf_columns* cols);
/* FILTER lists discovered so far, in id order: lists[8 * i .. 8 * i + counts[i]) are header-string indexes ([] = '.') */
int exon_hip_bcf_parser_filters(exon_hip_bcf_parser* parser, int32_t* lists, int32_t* counts, int32_t cap, int32_t* n_filters);
int exon_hip_bcf_parser_destroy(exon_hip_bcf_parser* parser);

/* GpuFilterAggExec::execute in one call: pull every batch of `scan` and push it through `stream`. */
int exon_hip_stream_consume_scan(exon_hip_stream* s, exon_hip_scan* scan, int64_t* rows);

#ifdef __cplusplus
}
#endif
#endif /* EXON_HIP_H */
