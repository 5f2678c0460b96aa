/*
 * exon_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY -- never linked into the product).
 *
 * A plain-C restatement of the reference's scan -> filter -> aggregate path
 * (wheretrue/exon v0.32.4 + DataFusion 44 / arrow 53 / noodles 0.87 semantics),
 * working on the reference's *Arrow layout* (Utf8 chrom/reference, List<Utf8>
 * filter, Utf8 mapping_quality, Utf8 quality_scores) in 8192-row batches with
 * T partitions, exactly the plan shape
 *   AggregateExec(Final) <- AggregateExec(Partial) <- FilterExec <- <Fmt>Scan.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY PINNING: the Rust reference cannot be built here (no rustc/cargo, 576
 * crates, no network), and the filter/aggregate arithmetic lives in third-party
 * DataFusion 44.0.0 / arrow 53.3.0, record parsing in noodles 0.87.0 (Cargo.lock).
 * The oracle is pinned against every value the reference's own tests hold for
 * this path (tests/test_oracle_pins.py): slt counts
 * 621/191/219/211/11/61/7/2, the UDF truth tables, the physical-expr KATs, the
 * quality-score vectors.  Float aggregates / GROUP BY results are NOT pinned by
 * any reference test ("parity unpinned" for those, cross-checked against
 * pyarrow compute instead).
 */
#ifndef EXON_ORACLE_H
#define EXON_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double seconds_materialize; /* building the reference Arrow layout (untimed part of the baseline) */
  double seconds_exec;        /* FilterExec + AggregateExec(Partial) + Final */
  int threads;
} orc_timing;

/* ---- counter-based synthetic inputs (DESIGN.md "Synthetic inputs") -------------------- */
uint64_t orc_rnd(uint64_t seed, uint64_t col, uint64_t i);
void orc_c2_contig_starts(int64_t n_total, int64_t starts[25]);
void orc_gen_c6(uint64_t seed, int64_t lo, int64_t hi, int32_t* ref_id, uint8_t* ref_valid, int64_t* start, int64_t* end,
                uint8_t* pos_valid);
void orc_gen_c2(uint64_t seed, int64_t n_total, int64_t lo, int64_t hi, int32_t* chrom_id, int64_t* pos);
void orc_gen_c3(uint64_t seed, int64_t lo, int64_t hi, int32_t* flag, uint8_t* mapq,
                uint8_t* mapq_valid, int32_t* ref_id, uint8_t* ref_valid);
void orc_gen_c4(uint64_t seed, int64_t lo, int64_t hi, float* af, uint8_t* af_valid, float* qual,
                uint8_t* qual_valid, int32_t* filter_id);
void orc_gen_c5(uint64_t seed, int64_t lo, int64_t hi, int32_t read_len, int32_t* offsets,
                uint8_t* bytes);
int orc_c2_num_contigs(void);
const char* orc_c2_contig_name(int i);
int64_t orc_c2_contig_len(int i);
int orc_c3_num_refs(void);
const char* orc_c3_ref_name(int i);
int orc_c4_num_filters(void);
const char* orc_c4_filter_name(int i); /* ';'-joined list, "" = empty list */

/* ---- region grammar (noodles_core::Region::from_str) --------------- */
/* returns 0 on success; end = INT64_MAX when open; start defaults to 1 */
int orc_parse_region(const char* s, char* name_out, int name_cap, int64_t* start, int64_t* end);
int orc_parse_interval(const char* s, int64_t* start, int64_t* end);

/* ---- UDF restatements (exon-core/src/udfs) -------------------------------------------- */
/* region_match(chrom, pos, region): udfs/vcf/mod.rs:65-131. returns -1 on NULL input (the UDF errors). */
int orc_region_match(const char* chrom, int has_pos, int64_t pos, const char* region);
/* interval_match(pos, interval): udfs/vcf/mod.rs:232-274. NULL pos -> false */
int orc_interval_match(int has_pos, int64_t pos, const char* interval);
int orc_chrom_match(const char* chrom, const char* name);
/* sam flag UDFs: udfs/sam/samflags.rs:26-47: (flag as u16) & bit != 0 */
int orc_sam_flag(int32_t flag, uint16_t bit);
/* quality_scores_to_list: udfs/sequence/quality_score_string_to_list.rs:83-86 (ASCII path) */
int orc_quality_scores_to_list(const char* s, int32_t* out, int cap);
/* BAM range hit: exon-bam/src/indexed_async_batch_stream.rs:66-87 */
int orc_bam_intersects(int has_ref, int32_t ref_id, int has_start, int64_t start, int has_end,
                       int64_t end, int32_t region_ref_id, int64_t rstart, int64_t rend);
/* whole-file round-robin repartition: exon-core/src/datasources/exon_file_scan_config.rs:79-110.
 * sizes[n] -> group_of[n] (group index per ORIGINAL file index), returns number of groups. */
int orc_regroup_files_by_size(const int64_t* sizes, int n, int target, int* group_of);

/* ---- the plan restatements (device-layout in, reference Arrow layout inside) ---------- */
int64_t orc_c2_region_count(const int32_t* chrom_id, const int64_t* pos, const uint8_t* chrom_valid,
                            const uint8_t* pos_valid, int64_t n, const char* const* contig_names,
                            int n_contigs, const char* region, int threads, orc_timing* t);

/* counts[R+1]; index R is the NULL-reference group */
int orc_c3_flag_mapq_group_count(const int32_t* flag, const uint8_t* mapq, const uint8_t* mapq_valid,
                                 const int32_t* ref_id, const uint8_t* ref_valid, int64_t n,
                                 const char* const* ref_names, int n_refs, int32_t flag_mask,
                                 int32_t flag_value, int32_t mapq_min, int threads,
                                 int64_t* counts, orc_timing* t);

/* cmp_op: 0 '>', 1 '>=', 2 '<', 3 '<=', 4 '=', 5 '!='.  Outputs are per dictionary id g < G. */
int orc_c4_cmp_avg_by_group(const float* af, const uint8_t* af_valid, const float* qual,
                            const uint8_t* qual_valid, const int32_t* filter_id, int64_t n,
                            const char* const* filter_names, int G, double thr, int cmp_op,
                            int threads, double* sum, int64_t* cnt_nonnull, int64_t* cnt_rows,
                            orc_timing* t);

/* hist[lmax][256] int64, bin = raw byte (Phred = bin - 33) */
int orc_c5_qual_pos_hist(const int32_t* offsets, const uint8_t* bytes, int64_t n_reads, int lmax,
                         int threads, int64_t* hist, orc_timing* t);

#ifdef __cplusplus
}
#endif
#endif
