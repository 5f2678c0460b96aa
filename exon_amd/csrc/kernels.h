// kernels.h -- launch interface of the gfx950 kernels (internal; the public surface is include/exon_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace exon {

// Per-(ctx, stream) scratch: per-block partial words written by the main kernels and folded into the
// caller's state by finalize_partials in a fixed order (deterministic f64 sums, no atomics).
struct Workspace {
  unsigned long long* partials = nullptr;  // [blocks][words]
  size_t partial_capacity = 0;             // in 8-byte words
  int* status = nullptr;                   // device error word (K5: position >= lmax, bad group id)
  // K4 with more ids than registers + the LDS table hold (tier 3): two record buffers of `tail_capacity` 8-byte records
  // (compacted rows, then the same rows grouped by id range) and a block of counters; absent until such a plan runs
  uint2* tail_rec_a = nullptr;
  uint2* tail_rec_b = nullptr;
  size_t tail_capacity = 0;
  unsigned* tail_u32 = nullptr;
};

struct LaunchCfg {
  int compute_units = 256;
  int blocks_per_cu = 8;  // main-kernel grid = compute_units * blocks_per_cu (persistent, grid-stride)
  bool overwrite = false; // state := result of this launch (EXON_HIP_LAUNCH_OVERWRITE) instead of state += result
  bool x_is_int = false;  // K4: the compared column holds Int32 values (exon_hip_plan_desc.x_type), not Float32
  bool y_is_int = false;  // K4: AVG's argument holds Int32 values (exon_hip_plan_desc.y_type)
};

size_t k2_partial_words(const LaunchCfg&);
size_t k3_partial_words(const LaunchCfg&, int n_refs);
size_t k4_partial_words(const LaunchCfg&, int n_groups);
// 8-byte records of tier-3 scratch a K4 launch over n rows wants (0: no tier 3); per buffer
size_t k4_tail_records(int64_t n, int n_groups);
constexpr size_t K4_TAIL_U32_WORDS = 2048 + 3 * 2048 + 2 + (size_t)2048 * 2048;  // workgroup counts, range totals, offsets, slice starts, [workgroup][range] histogram
size_t k5_partial_words(const LaunchCfg&, int lmax);

hipError_t launch_region_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* chrom,
                               const uint8_t* chrom_valid, const int64_t* pos, const uint8_t* pos_valid, int64_t n,
                               int32_t region_chrom, int64_t start, int64_t end, int64_t* d_count);

// K6: COUNT(*) of rows whose [start, end] interval on reference `region_ref` overlaps [region_start, region_end];
// strict: ... lies strictly inside (region_start, region_end) (start > a AND end < b: the BED / GFF form)
hipError_t launch_overlap_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* ref,
                                const uint8_t* ref_valid, const int64_t* start, const uint8_t* start_valid,
                                const int64_t* end, const uint8_t* end_valid, int64_t n, int32_t region_ref,
                                int64_t region_start, int64_t region_end, int64_t* d_count, bool strict = false);

hipError_t launch_flag_mapq_group_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* flag,
                                        const uint8_t* flag_valid, const uint8_t* mapq, const uint8_t* mapq_valid,
                                        const int32_t* ref_id, const uint8_t* ref_valid, int64_t n, int32_t flag_mask,
                                        int32_t flag_value, int32_t mapq_min, int32_t n_refs, int64_t* d_counts);

hipError_t launch_cmp_avg_by_group(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const float* x,
                                   const uint8_t* x_valid, const float* y, const uint8_t* y_valid, const int32_t* gid,
                                   int64_t n, double thr, int cmp_op, int n_groups, int64_t* d_counts, double* d_sums);

hipError_t launch_qual_pos_hist(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* offsets,
                                const uint8_t* bytes, int64_t n_reads, int lmax, int64_t* d_hist);
// reads given as independent [starts[r], ends[r]) views into `bytes` (e.g. the quality lines of raw FASTQ text)
hipError_t launch_qual_pos_hist_views(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* starts,
                                      const int32_t* ends, const uint8_t* bytes, int64_t n_reads, int lmax,
                                      int64_t* d_hist);

// a quality column held as several Arrow batches (chunks): ONE offsets scan, ONE main kernel (the workgroups' LDS
// histograms live across chunks), ONE finalize for up to K5_MAX_CHUNKS chunks.  Passed to the kernels by value.
constexpr int K5_MAX_CHUNKS = 64;
struct K5Chunks {
  const int32_t* off[K5_MAX_CHUNKS];
  const int32_t* ends[K5_MAX_CHUNKS];
  const uint8_t* bytes[K5_MAX_CHUNKS];
  int64_t n[K5_MAX_CHUNKS];
  int count;
};
// every chunk must have n > 0; 1 <= count <= K5_MAX_CHUNKS
hipError_t launch_qual_pos_hist_chunks(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const K5Chunks& ch, int lmax,
                                       int64_t* d_hist);

// pushed-down region filter as a row mask: out_valid = in_valid AND (row hits the region); *n_pass += rows kept.
// point form (VCF): id_col = chrom id, start = pos; range form (BAM): id_col = reference id, [start, end].
hipError_t launch_region_mask(hipStream_t s, bool range_form, const int32_t* id_col, const uint8_t* id_valid, const int64_t* start,
                              const int64_t* end, const uint8_t* pos_valid, const uint8_t* in_valid, int64_t n, int32_t id,
                              int64_t a, int64_t b, uint8_t* out_valid, unsigned long long* n_pass);

// out[v] = sum over ranks (rank order) of gathered[rank][v]; words [0, n_i64) int64, then n_f64 float64
hipError_t launch_fold_states(hipStream_t s, const void* gathered, int world, int64_t n_i64, int64_t n_f64, void* out);

// Re-keying of a packed partial state: dst[plane][map[g]] += src[plane][g] for every keyed plane (`planes_i64` int64 planes of G
// words, then `tail_i64` unkeyed int64 words that are added in place -- K3's NULL-reference group --, then `planes_f64` float64
// planes of G words); map[g] < 0: key g carries nothing and is skipped.  Used when the dictionary ids of one scan / one rank
// are brought into the order of a shared key dictionary (merge by key VALUE, not by id).
hipError_t launch_permute_add_state(hipStream_t s, const void* src, void* dst, const int32_t* map, int n_map, int G, int planes_i64,
                                    int tail_i64, int planes_f64);

// bare streaming read of up to 4 buffers in lock-step, the access pattern and grid of K2-K6 (bench.py's per-box ceiling)
hipError_t launch_read_probe(hipStream_t s, const LaunchCfg& cfg, const void* const* buffers, int n_buffers, int64_t bytes_each,
                             unsigned* sink);

hipError_t launch_gen_c2(hipStream_t s, uint64_t seed, int64_t n_total, int64_t lo, int64_t hi, int32_t* chrom,
                         int64_t* pos);
hipError_t launch_gen_c3(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, int32_t* flag, uint8_t* mapq,
                         uint8_t* mapq_valid, int32_t* ref_id, uint8_t* ref_valid);
hipError_t launch_gen_c4(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, float* af, uint8_t* af_valid,
                         float* qual, uint8_t* qual_valid, int32_t* filter_id);
hipError_t launch_gen_c6(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, int32_t* ref_id, uint8_t* ref_valid,
                         int64_t* start, int64_t* end, uint8_t* pos_valid);
hipError_t launch_gen_c5(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, int32_t read_len, int32_t* offsets,
                         uint8_t* bytes);

}  // namespace exon
