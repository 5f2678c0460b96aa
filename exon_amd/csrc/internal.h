// internal.h -- shared between capi.cpp (ctx + operator launches) and stream.cpp (plan/stream layer).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>

#include "../../include/exon_hip.h"
#include "kernels.h"

struct exon_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;  // used when the caller passes stream == NULL
  exon::LaunchCfg cfg;
  std::mutex mu;
  std::map<hipStream_t, exon::Workspace> workspaces;
  std::string error;
  hipDeviceProp_t props;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // device buffers of the GPU-side parsers are recycled between scans (exon_pool_*): their sizes repeat exactly and
  // hipMalloc / hipFree of gigabytes costs tens of milliseconds per file
  std::multimap<size_t, void*> pool_free;
  std::map<void*, size_t> pool_live;
};

// records the message on the ctx (and the calling thread) and returns `code`
int fail(exon_hip_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
const std::string& exon_hip_tls_error();

#define HIP_TRY(ctx, expr)                                                                             \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_));  \
  } while (0)

inline hipStream_t pick_stream(exon_hip_ctx* ctx, void* s) { return s ? (hipStream_t)s : ctx->stream; }

// Workspace of `words` 8-byte partial words for launches on stream `s` (grown on demand).
int get_workspace(exon_hip_ctx* ctx, hipStream_t s, size_t words, exon::Workspace* out);

// inflate.hip: enqueue the BGZF inflate (+ CRC-32) kernels over device-resident block tables
// a stream that is about to be destroyed: release the inflate scratch kept for it (the stream must be idle)
void exon_bgzf_forget_stream(hipStream_t s);
hipError_t exon_bgzf_inflate_launch(hipStream_t s, const uint8_t* d_comp, const exon_hip_bgzf_block* d_blocks, int n_blocks,
                                    uint8_t* d_out, int* d_status, bool verify_crc, int flavor_hint = -1, bool text_like = true);
const char* exon_bgzf_status_name(int code);

// scan.cpp: slab buffers kept per ctx between scans
void exon_hip_release_ctx_caches(exon_hip_ctx* ctx);
const void* exon_hip_gpu_local_cpus(int device);  // scan.cpp: a cpu_set_t of the GPU's NUMA node (nullptr: nothing to choose), and ...
void exon_hip_run_on(const void* cpus);            // ... the calling thread's affinity set to it (host threads that feed the DMA engine)
void exon_hip_prewarm_ctx(exon_hip_ctx* ctx);  // scan.cpp: the file pipelines' side streams and events, made with the context

// text_columns.hip: the reference's string / list columns of a slab as Arrow buffers on the device (exon_hip_scan_options.projection)
struct ExonTextScratch;
struct ExonVcfText {
  const int32_t* id_list_offsets;  // [n_rows + 1] -> items
  const uint8_t* id_valid;         // bitmap: the ID field is not '.'
  const int32_t* id_item_offsets;  // [n_id_items + 1] -> id_values
  const uint8_t* id_values;
  int64_t n_id_items, n_id_bytes;
  const int32_t* ref_offsets;      // [n_rows + 1]
  const uint8_t* ref_values;
  int64_t n_ref_bytes;
  const uint8_t* alt_valid;        // bitmap: the ALT field is not '.' (the list itself has no items: see text_columns.hip)
};
// BCF id / ref / alt through the reference's EAGER builder (eager_array_builder.rs:112-134): both lists carry their items, neither is
// ever NULL (an empty list when the record has no id / no alternate bases)
struct ExonBcfText {
  const int32_t *id_list_offsets, *id_item_offsets, *ref_offsets, *alt_list_offsets, *alt_item_offsets;
  const uint8_t *id_values, *ref_values, *alt_values;
  int64_t n_id_items, n_id_bytes, n_ref_bytes, n_alt_items, n_alt_bytes;
};
struct ExonBamText {
  const int32_t *name_offsets, *cigar_offsets, *seq_offsets;  // [n_rows + 1] each; seq_offsets are quality_scores' list offsets too
  const uint8_t *name_values, *cigar_values, *seq_values, *name_valid;
  const int64_t* qual_values;
  int64_t n_name_bytes, n_cigar_bytes, n_seq_bytes;
  const int32_t* qual_offsets;  // quality_scores' list offsets (BAM: = seq_offsets; SAM: its own, QUAL may be '*' next to a SEQ)
  int64_t n_qual_items;
};
int exon_text_vcf(exon_hip_ctx* ctx, void* stream, ExonTextScratch** scratch, const uint8_t* d_text, int64_t n_bytes, const unsigned* d_nl, int64_t n_rows, uint64_t projection,
                  ExonVcfText* out);
int exon_text_bam(exon_hip_ctx* ctx, void* stream, ExonTextScratch** scratch, const uint8_t* d_data, int64_t n_bytes, const uint32_t* d_rec_of_row, int64_t n_rows, uint64_t projection,
                  ExonBamText* out);
struct ExonFastqText {  // name, description, sequence, quality_scores (exon-fastq/src/config.rs:79-88), in that order
  const int32_t* offsets[4];  // [n_reads + 1] each
  const uint8_t* values[4];
  int64_t n_bytes[4];
  const uint8_t* desc_valid;  // bitmap: the header has something behind its first space
};
// SAM lines (the parser's newline index) -> the same columns; n_undecided != 0: a line the device does not print the way the reader would
int exon_text_sam(exon_hip_ctx* ctx, void* stream, ExonTextScratch** scratch, const uint8_t* d_text, int64_t n_bytes, const unsigned* d_nl, int64_t n_rows, uint64_t projection,
                  ExonBamText* out, int64_t* n_undecided);
int exon_text_fastq(exon_hip_ctx* ctx, void* stream, ExonTextScratch** scratch, const exon_hip_fastq_views* views, int64_t n_bytes, ExonFastqText* out);
int exon_text_bcf(exon_hip_ctx* ctx, void* stream, ExonTextScratch** scratch, const uint8_t* d_data, int64_t n_bytes, const uint32_t* d_rec_of_row, int64_t n_rows, uint64_t projection,
                  ExonBcfText* out, int64_t* n_undecided);
void exon_text_scratch_destroy(ExonTextScratch* s);
// the parsers' own indexes the text columns are built from (valid until the next parse call)
const unsigned* exon_hip_vcf_parser_newlines(exon_hip_vcf_parser* p);
const unsigned* exon_hip_sam_parser_newlines(exon_hip_sam_parser* p);      // gpu_parse.hip: the same for SAM lines      // gpu_parse.hip: byte offset of every line's '\n' in the aligned slab
const uint32_t* exon_hip_bam_parser_row_records(exon_hip_bam_parser* p);
const uint32_t* exon_hip_bcf_parser_row_records(exon_hip_bcf_parser* p);   // bcf_parse.hip: the same for BCF records   // bam_parse.hip: byte offset of every row's record

// capi.cpp: size-keyed recycling of device buffers (released by exon_hip_ctx_destroy)
void* exon_pool_alloc(exon_hip_ctx* ctx, size_t bytes);
void exon_pool_free(exon_hip_ctx* ctx, void* p);

// internal launch flags next to the public EXON_HIP_LAUNCH_* bits: K4's compared column / AVG argument is Int32
// (exon_hip_plan_desc.x_type / y_type)
#define EXON_LAUNCH_X_INT32 0x100
#define EXON_LAUNCH_Y_INT32 0x200

// capi.cpp: the operator launches with EXON_HIP_LAUNCH_* flags (the extern "C" operators are the ACCUMULATE forms)
int exon_op_region_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* chrom_id, const exon_hip_column* pos,
                         int64_t n, int32_t region_chrom_id, int64_t start, int64_t end, int64_t* d_count, int flags);
int exon_op_overlap_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* ref_id, const exon_hip_column* start,
                          const exon_hip_column* end, int64_t n, int32_t region_ref_id, int64_t region_start,
                          int64_t region_end, int64_t* d_count, int flags, bool strict);
int exon_op_flag_mapq_group_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* flag,
                                  const exon_hip_column* mapq, const exon_hip_column* ref_id, int64_t n,
                                  int32_t flag_mask, int32_t flag_value, int32_t mapq_min, int32_t n_refs,
                                  int64_t* d_counts, int flags);
int exon_op_cmp_avg_by_group(exon_hip_ctx* ctx, void* stream, const exon_hip_column* x, const exon_hip_column* y,
                             const exon_hip_column* group_id, int64_t n, double threshold, int32_t cmp_op,
                             int32_t n_groups, int64_t* d_counts, double* d_sums, int flags);
int exon_op_qual_pos_hist(exon_hip_ctx* ctx, void* stream, const exon_hip_column* q, int64_t n_reads, int32_t lmax,
                          int64_t* d_hist, int flags);
// chunk c is q[c * stride]
int exon_op_qual_pos_hist_chunks(exon_hip_ctx* ctx, void* stream, const exon_hip_column* q, int stride, int32_t n_chunks,
                                 const int64_t* n_reads, int32_t lmax, int64_t* d_hist, int flags);
