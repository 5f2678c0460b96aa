/*
 * exon_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see exon_oracle.h header).
 *
 * Restates, in plain C, the reference's CPU path for scan -> filter -> aggregate:
 *   - columns are materialised in the reference's Arrow layout
 *       VCF : chrom Utf8, pos Int64, qual Float32, filter List<Utf8>, info.AF Float32
 *             (exon-vcf/src/array_builder/lazy_array_builder.rs:153-216,
 *              exon-core/src/datasources/vcf/schema_builder.rs:85-129)
 *       BAM : flag Int32, reference Utf8?, mapping_quality Utf8? (255 -> NULL)
 *             (exon-bam/src/array_builder.rs:114-143, exon-sam/src/schema_builder.rs:385-398)
 *       FASTQ: quality_scores Utf8 (exon-fastq/src/array_builder.rs:68-102)
 *   - batches of 8192 rows (exon-common/src/lib.rs:27), T partitions
 *     (target_partitions = num_cpus, exon-core/src/config/mod.rs:44)
 *   - FilterExec keeps rows whose predicate IS TRUE (Kleene AND; NULL drops the row),
 *     Float32-vs-Float64-literal comparisons are done in f64, Utf8 mapping_quality is
 *     CAST to Int32 (DataFusion 44 / arrow 53 semantics; third-party, not in /root/reference)
 *   - AggregateExec(Partial) interns variable-width keys with a hash table and updates
 *     count / avg (f64 sum + u64 count) states; AggregateExec(Final) merges by key.
 * Not shipped, not linked by the product library.
 */
#define _GNU_SOURCE
#include "exon_oracle.h"
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define BATCH 8192 /* exon-common/src/lib.rs:27 DEFAULT_BATCH_SIZE */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static inline int bit_get(const uint8_t* bm, int64_t i) { return bm ? (bm[i >> 3] >> (i & 7)) & 1 : 1; }
static inline void bit_set(uint8_t* bm, int64_t i) { bm[i >> 3] |= (uint8_t)(1u << (i & 7)); }

/* ====================================================================================== */
/* Synthetic inputs (counter-based; DESIGN.md "Synthetic inputs")                          */
/* ====================================================================================== */
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
uint64_t orc_rnd(uint64_t seed, uint64_t col, uint64_t i) {
  return mix64(seed + col * 0xD1B54A32D192ED03ULL + (i + 1) * 0x9E3779B97F4A7C15ULL);
}

static const char* const GR_NAMES[25] = {"1",  "2",  "3",  "4",  "5",  "6",  "7",  "8",  "9",
                                         "10", "11", "12", "13", "14", "15", "16", "17", "18",
                                         "19", "20", "21", "22", "X",  "Y",  "MT"};
static const int64_t GR_LEN[25] = {249250621, 243199373, 198022430, 191154276, 180915260,
                                   171115067, 159138663, 146364022, 141213431, 135534747,
                                   135006516, 133851895, 115169878, 107349540, 102531392,
                                   90354753,  81195210,  78077248,  59128983,  63025520,
                                   48129895,  51304566,  155270560, 59373566,  16569};
int orc_c2_num_contigs(void) { return 24; }
const char* orc_c2_contig_name(int i) { return GR_NAMES[i]; }
int64_t orc_c2_contig_len(int i) { return GR_LEN[i]; }
int orc_c3_num_refs(void) { return 25; }
const char* orc_c3_ref_name(int i) { return GR_NAMES[i]; }
static const char* const C4_FILTERS[5] = {"PASS", "", "q10", "q10;s50", "s50"};
int orc_c4_num_filters(void) { return 5; }
const char* orc_c4_filter_name(int i) { return C4_FILTERS[i]; }

/* rows of contig c are [starts[c], starts[c+1]); proportional to GRCh37 length */
void orc_c2_contig_starts(int64_t n_total, int64_t starts[25]) {
  unsigned __int128 total = 0, cum = 0;
  for (int c = 0; c < 24; c++) total += (unsigned __int128)GR_LEN[c];
  starts[0] = 0;
  for (int c = 0; c < 24; c++) {
    cum += (unsigned __int128)GR_LEN[c];
    starts[c + 1] = (int64_t)(((unsigned __int128)n_total * cum) / total);
  }
  starts[24] = n_total;
}

void orc_gen_c2(uint64_t seed, int64_t n_total, int64_t lo, int64_t hi, int32_t* chrom_id,
                int64_t* pos) {
  int64_t starts[25];
  orc_c2_contig_starts(n_total, starts);
#pragma omp parallel for schedule(static)
  for (int64_t i = lo; i < hi; i++) {
    int c = 0;
    while (c < 23 && i >= starts[c + 1]) c++;
    int64_t j = i - starts[c], nc = starts[c + 1] - starts[c];
    int64_t w = GR_LEN[c] / nc;
    if (w < 1) w = 1;
    int64_t p = 1 + j * w + (int64_t)(orc_rnd(seed, 0, (uint64_t)i) % (uint64_t)w);
    if (p > GR_LEN[c]) p = GR_LEN[c];
    chrom_id[i - lo] = c;
    pos[i - lo] = p;
  }
}

/* C6 (seed 6): alignments for the interval-overlap count; identical to gen_c6_kernel (kernels.hip) */
void orc_gen_c6(uint64_t seed, int64_t lo, int64_t hi, int32_t* ref_id, uint8_t* ref_valid, int64_t* start, int64_t* end,
                uint8_t* pos_valid) {
  uint32_t rthr[24];
  unsigned __int128 total = 0, c128 = 0;
  for (int c = 0; c < 25; c++) total += (unsigned __int128)GR_LEN[c];
  for (int c = 0; c < 24; c++) {
    c128 += (unsigned __int128)GR_LEN[c];
    rthr[c] = (uint32_t)((c128 << 32) / total);
  }
  int64_t n = hi - lo;
  memset(ref_valid, 0, (size_t)((n + 7) / 8));
  memset(pos_valid, 0, (size_t)((n + 7) / 8));
#pragma omp parallel for schedule(static)
  for (int64_t blk = 0; blk < (n + 4095) / 4096; blk++)
  for (int64_t i = lo + blk * 4096; i < hi && i < lo + (blk + 1) * 4096; i++) {
    uint64_t r0 = orc_rnd(seed, 0, (uint64_t)i), r1 = orc_rnd(seed, 1, (uint64_t)i);
    uint32_t u0 = (uint32_t)(r0 >> 32);
    int rc = 0;
    for (int c = 0; c < 24; c++) rc += (u0 >= rthr[c]);
    int mapped = ((uint32_t)r0 % 50u) != 0u;
    int64_t st = 1 + (int64_t)(r1 % 249000000ull);
    ref_id[i - lo] = mapped ? rc : -1;
    start[i - lo] = mapped ? st : 0;
    end[i - lo] = mapped ? st + (int64_t)((r1 >> 40) % 20000ull) : 0;
    if (mapped) {
      bit_set(ref_valid, i - lo);
      bit_set(pos_valid, i - lo);
    }
  }
}

static const int32_t C3_FLAGS[12] = {99, 147, 83, 163, 1123, 1171, 1187, 1107, 77, 141, 355, 65};
static const int C3_FLAG_PCT[12] = {21, 21, 21, 21, 2, 2, 2, 2, 1, 1, 1, 5};
static inline uint32_t pct_thr(int cum_pct) { return (uint32_t)((((uint64_t)cum_pct) << 32) / 100); }

void orc_gen_c3(uint64_t seed, int64_t lo, int64_t hi, int32_t* flag, uint8_t* mapq,
                uint8_t* mapq_valid, int32_t* ref_id, uint8_t* ref_valid) {
  uint32_t fthr[11], rthr[24];
  int cum = 0;
  for (int k = 0; k < 11; k++) {
    cum += C3_FLAG_PCT[k];
    fthr[k] = pct_thr(cum);
  }
  unsigned __int128 total = 0, c128 = 0;
  for (int c = 0; c < 25; c++) total += (unsigned __int128)GR_LEN[c];
  for (int c = 0; c < 24; c++) {
    c128 += (unsigned __int128)GR_LEN[c];
    rthr[c] = (uint32_t)((c128 << 32) / total);
  }
  int64_t n = hi - lo;
  memset(mapq_valid, 0, (size_t)((n + 7) / 8));
  memset(ref_valid, 0, (size_t)((n + 7) / 8));
  const uint32_t m8 = pct_thr(8), m20 = pct_thr(20), m40 = pct_thr(40), m98 = pct_thr(98);
#pragma omp parallel for schedule(static)
  for (int64_t blk = 0; blk < (n + 4095) / 4096; blk++)
  for (int64_t i = lo + blk * 4096; i < hi && i < lo + (blk + 1) * 4096; i++) {
    uint64_t r0 = orc_rnd(seed, 0, (uint64_t)i), r1 = orc_rnd(seed, 1, (uint64_t)i),
             r2 = orc_rnd(seed, 2, (uint64_t)i);
    uint32_t u0 = (uint32_t)(r0 >> 32), u1 = (uint32_t)(r1 >> 32), v1 = (uint32_t)r1,
             u2 = (uint32_t)(r2 >> 32);
    int fi = 0;
    for (int k = 0; k < 11; k++) fi += (u0 >= fthr[k]);
    int32_t f = C3_FLAGS[fi];
    flag[i - lo] = f;
    uint8_t q;
    int qv = 1;
    if (u1 < m8) q = 0;
    else if (u1 < m20) q = (uint8_t)(1 + v1 % 29);
    else if (u1 < m40) q = (uint8_t)(30 + v1 % 30);
    else if (u1 < m98) q = 60;
    else { q = 255; qv = 0; }
    mapq[i - lo] = q;
    if (qv) bit_set(mapq_valid, i - lo);
    int rc = 0;
    for (int c = 0; c < 24; c++) rc += (u2 >= rthr[c]);
    if (f & 4) {
      ref_id[i - lo] = -1;
    } else {
      ref_id[i - lo] = rc;
      bit_set(ref_valid, i - lo);
    }
  }
}

void orc_gen_c4(uint64_t seed, int64_t lo, int64_t hi, float* af, uint8_t* af_valid, float* qual,
                uint8_t* qual_valid, int32_t* filter_id) {
  int64_t n = hi - lo;
  memset(af_valid, 0, (size_t)((n + 7) / 8));
  memset(qual_valid, 0, (size_t)((n + 7) / 8));
  const uint32_t t0 = pct_thr(85), t1 = pct_thr(90), t2 = pct_thr(96), t3 = pct_thr(99);
#pragma omp parallel for schedule(static)
  for (int64_t blk = 0; blk < (n + 4095) / 4096; blk++)
  for (int64_t i = lo + blk * 4096; i < hi && i < lo + (blk + 1) * 4096; i++) {
    uint64_t r0 = orc_rnd(seed, 0, (uint64_t)i), r1 = orc_rnd(seed, 1, (uint64_t)i),
             r2 = orc_rnd(seed, 2, (uint64_t)i);
    uint32_t k = (uint32_t)((((r0 >> 23) & 0xFF) * 14) >> 8);
    uint32_t bits = ((126u - k) << 23) | (uint32_t)(r0 & 0x7FFFFF);
    float a;
    memcpy(&a, &bits, 4);
    if (((r0 >> 31) & 0x3FF) == 0) a = 0.01f; /* f32(0.01) widened to f64 is < 0.01: the coercion trap */
    af[i - lo] = a;
    if ((r0 >> 44) >= 10486) bit_set(af_valid, i - lo); /* ~1 % NULL */
    uint32_t kq = (uint32_t)(r1 & 0xFFFFFFFFu) % 10000u;
    qual[i - lo] = (float)((double)kq / 10.0);
    if ((r1 >> 44) >= 31457) bit_set(qual_valid, i - lo); /* ~3 % NULL */
    uint32_t u = (uint32_t)(r2 >> 32);
    filter_id[i - lo] = (int32_t)((u >= t0) + (u >= t1) + (u >= t2) + (u >= t3));
  }
}

void orc_gen_c5(uint64_t seed, int64_t lo, int64_t hi, int32_t L, int32_t* offsets, uint8_t* bytes) {
  for (int64_t r = lo; r <= hi; r++) offsets[r - lo] = (int32_t)((r - lo) * L);
#pragma omp parallel for schedule(static)
  for (int64_t r = lo; r < hi; r++) {
    for (int32_t p = 0; p < L; p++) {
      uint64_t h = orc_rnd(seed, 0, (uint64_t)(r * L + p));
      int s = (int)(h & 0xFF) + (int)((h >> 8) & 0xFF) + (int)((h >> 16) & 0xFF) + (int)((h >> 24) & 0xFF);
      int d = ((s * 3 + 4096 - 1530) >> 7) - 32;
      int q = 38 - (10 * p) / L + d;
      if (q < 0) q = 0;
      if (q > 41) q = 41;
      bytes[(r - lo) * (int64_t)L + p] = (uint8_t)(33 + q);
    }
  }
}

/* ====================================================================================== */
/* Region grammar: noodles-core 0.15 Region::from_str / Interval::from_str (not in tree;   */
/* call sites exon-core/src/physical_plan/infer_region.rs:31-33, udfs/vcf/mod.rs:86-118)   */
/* ====================================================================================== */
static int parse_position(const char* s, int len, int64_t* out) {
  if (len <= 0 || len > 19) return -1;
  int64_t v = 0;
  int i = 0;
  if (s[0] == '+') { i = 1; if (len == 1) return -1; }
  for (; i < len; i++) {
    if (s[i] < '0' || s[i] > '9') return -1;
    v = v * 10 + (s[i] - '0');
  }
  if (v < 1) return -1; /* Position is NonZeroUsize */
  *out = v;
  return 0;
}
static int parse_interval_n(const char* s, int len, int64_t* start, int64_t* end) {
  *start = 1;
  *end = INT64_MAX;
  if (len == 0) return 0;
  const char* dash = memchr(s, '-', (size_t)len);
  if (!dash) return parse_position(s, len, start);
  if (parse_position(s, (int)(dash - s), start)) return -1;
  return parse_position(dash + 1, (int)(len - (dash - s) - 1), end);
}
int orc_parse_interval(const char* s, int64_t* start, int64_t* end) {
  return parse_interval_n(s, (int)strlen(s), start, end);
}
int orc_parse_region(const char* s, char* name, int cap, int64_t* start, int64_t* end) {
  int len = (int)strlen(s);
  if (len == 0) return -1;
  *start = 1;
  *end = INT64_MAX;
  int nlen = len;
  const char* colon = NULL;
  for (int i = len - 1; i >= 0; i--)
    if (s[i] == ':') { colon = s + i; break; }
  if (colon) {
    int64_t a, b;
    if (parse_interval_n(colon + 1, (int)(len - (colon - s) - 1), &a, &b) == 0) {
      nlen = (int)(colon - s);
      *start = a;
      *end = b;
    }
  }
  if (nlen >= cap) return -1;
  memcpy(name, s, (size_t)nlen);
  name[nlen] = 0;
  return 0;
}

/* ====================================================================================== */
/* UDF restatements                                                                        */
/* ====================================================================================== */
int orc_region_match(const char* chrom, int has_pos, int64_t pos, const char* region) {
  char name[256];
  int64_t a, b;
  if (orc_parse_region(region, name, sizeof name, &a, &b)) return -1;
  if (!chrom || !has_pos) return -1; /* udfs/vcf/mod.rs:107-110: NULL -> Execution error */
  if (pos == 0) return -1;           /* Position::try_from(0) fails (:112-114) */
  uint64_t up = (uint64_t)pos;       /* `pos as usize` */
  return strcmp(name, chrom) == 0 && up >= (uint64_t)a && up <= (uint64_t)b;
}
int orc_interval_match(int has_pos, int64_t pos, const char* interval) {
  int64_t a, b;
  if (orc_parse_interval(interval, &a, &b)) return -1;
  if (!has_pos) return 0; /* udfs/vcf/mod.rs:267: NULL -> Some(false) */
  if (pos == 0) return -1;
  uint64_t up = (uint64_t)pos;
  return up >= (uint64_t)a && up <= (uint64_t)b;
}
int orc_chrom_match(const char* chrom, const char* name) {
  if (!chrom) return -1;
  return strcmp(chrom, name) == 0;
}
int orc_sam_flag(int32_t flag, uint16_t bit) { return (((uint16_t)flag) & bit) != 0; }
int orc_quality_scores_to_list(const char* s, int32_t* out, int cap) {
  int n = 0;
  for (const unsigned char* p = (const unsigned char*)s; *p; p++) {
    if (n >= cap) return -1;
    out[n++] = (int32_t)*p - 33;
  }
  return n;
}
int orc_bam_intersects(int has_ref, int32_t ref_id, int has_start, int64_t start, int has_end,
                       int64_t end, int32_t region_ref_id, int64_t rstart, int64_t rend) {
  if (!has_ref || !has_start || !has_end) return 0;
  /* Interval::intersects: a.start <= b.end && b.start <= a.end */
  return (start <= rend && rstart <= end) && ref_id == region_ref_id;
}
typedef struct { int64_t size; int idx; } sized_t;
static int cmp_sized(const void* a, const void* b) {
  const sized_t *x = a, *y = b;
  if (x->size != y->size) return x->size < y->size ? -1 : 1;
  return x->idx - y->idx; /* itertools sorted_by_key is stable */
}
int orc_regroup_files_by_size(const int64_t* sizes, int n, int target, int* group_of) {
  if (n == 0) return 0;
  sized_t* v = malloc(sizeof(sized_t) * (size_t)n);
  for (int i = 0; i < n; i++) { v[i].size = sizes[i]; v[i].idx = i; }
  qsort(v, (size_t)n, sizeof(sized_t), cmp_sized);
  int tp = target < n ? target : n;
  if (tp < 1) tp = 1;
  for (int i = 0; i < n; i++) group_of[v[i].idx] = i % tp;
  free(v);
  return tp;
}

/* ====================================================================================== */
/* Arrow-layout helpers                                                                    */
/* ====================================================================================== */
typedef struct {
  int32_t* offsets; /* n+1 */
  uint8_t* data;
  uint8_t* valid; /* byte per row (1 = valid); the restatement uses byte maps for booleans */
  int64_t n;
} utf8_col;

static void utf8_free(utf8_col* c) { free(c->offsets); free(c->data); free(c->valid); }

/* materialise a dictionary-id column as the reference's Utf8 column */
static int utf8_from_ids(utf8_col* out, const int32_t* ids, const uint8_t* valid_bm, int64_t lo,
                         int64_t n, const char* const* names, int n_names) {
  int* lens = malloc(sizeof(int) * (size_t)(n_names > 0 ? n_names : 1));
  for (int i = 0; i < n_names; i++) lens[i] = (int)strlen(names[i]);
  int64_t total = 0;
  for (int64_t i = 0; i < n; i++)
    if (bit_get(valid_bm, lo + i)) total += lens[ids[lo + i]];
  if (total >= INT32_MAX) { free(lens); return -1; }
  out->n = n;
  out->offsets = malloc(sizeof(int32_t) * (size_t)(n + 1));
  out->data = malloc((size_t)(total + 1));
  out->valid = malloc((size_t)(n > 0 ? n : 1));
  int32_t o = 0;
  for (int64_t i = 0; i < n; i++) {
    out->offsets[i] = o;
    int v = bit_get(valid_bm, lo + i);
    out->valid[i] = (uint8_t)v;
    if (v) {
      int id = ids[lo + i];
      memcpy(out->data + o, names[id], (size_t)lens[id]);
      o += lens[id];
    }
  }
  out->offsets[n] = o;
  free(lens);
  return 0;
}

static inline uint64_t hash_bytes(const uint8_t* p, int64_t n, uint64_t h) {
  for (int64_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001B3ULL; }
  return mix64(h);
}

/* GroupValues restatement: variable-width byte keys interned to dense group indexes. */
typedef struct {
  int64_t* slots; /* group index + 1, 0 = empty */
  uint64_t* hashes;
  int64_t cap, n_groups;
  uint8_t* keys; /* concatenated key bytes */
  int64_t* key_off; /* n_groups + 1 */
  int64_t keys_cap, koff_cap;
} group_tbl;

static void gt_init(group_tbl* g) {
  g->cap = 64;
  g->slots = calloc((size_t)g->cap, sizeof(int64_t));
  g->hashes = malloc(sizeof(uint64_t) * (size_t)g->cap);
  g->n_groups = 0;
  g->keys_cap = 1024;
  g->keys = malloc((size_t)g->keys_cap);
  g->koff_cap = 64;
  g->key_off = malloc(sizeof(int64_t) * (size_t)g->koff_cap);
  g->key_off[0] = 0;
}
static void gt_free(group_tbl* g) { free(g->slots); free(g->hashes); free(g->keys); free(g->key_off); }
static void gt_grow(group_tbl* g) {
  int64_t ncap = g->cap * 2;
  int64_t* ns = calloc((size_t)ncap, sizeof(int64_t));
  uint64_t* nh = malloc(sizeof(uint64_t) * (size_t)ncap);
  for (int64_t i = 0; i < g->cap; i++)
    if (g->slots[i]) {
      int64_t j = (int64_t)(g->hashes[i] & (uint64_t)(ncap - 1));
      while (ns[j]) j = (j + 1) & (ncap - 1);
      ns[j] = g->slots[i];
      nh[j] = g->hashes[i];
    }
  free(g->slots); free(g->hashes);
  g->slots = ns; g->hashes = nh; g->cap = ncap;
}
static int64_t gt_intern(group_tbl* g, const uint8_t* key, int64_t len) {
  uint64_t h = hash_bytes(key, len, 0xCBF29CE484222325ULL);
  int64_t j = (int64_t)(h & (uint64_t)(g->cap - 1));
  while (g->slots[j]) {
    if (g->hashes[j] == h) {
      int64_t gi = g->slots[j] - 1;
      int64_t kl = g->key_off[gi + 1] - g->key_off[gi];
      if (kl == len && memcmp(g->keys + g->key_off[gi], key, (size_t)len) == 0) return gi;
    }
    j = (j + 1) & (g->cap - 1);
  }
  int64_t gi = g->n_groups++;
  g->slots[j] = gi + 1;
  g->hashes[j] = h;
  int64_t o = g->key_off[gi];
  if (o + len > g->keys_cap) {
    while (o + len > g->keys_cap) g->keys_cap *= 2;
    g->keys = realloc(g->keys, (size_t)g->keys_cap);
  }
  if (gi + 2 > g->koff_cap) {
    g->koff_cap *= 2;
    g->key_off = realloc(g->key_off, sizeof(int64_t) * (size_t)g->koff_cap);
  }
  memcpy(g->keys + o, key, (size_t)len);
  g->key_off[gi + 1] = o + len;
  if (g->n_groups * 2 > g->cap) gt_grow(g);
  return gi;
}

static int clamp_threads(int threads) {
  int mx = omp_get_max_threads();
  if (threads <= 0 || threads > mx) threads = mx;
  return threads;
}
/* partition p of T covers whole 8192-row batches [b0,b1) */
static void part_range(int64_t n, int T, int p, int64_t* lo, int64_t* hi) {
  int64_t nb = (n + BATCH - 1) / BATCH;
  int64_t b0 = nb * p / T, b1 = nb * (p + 1) / T;
  *lo = b0 * BATCH;
  *hi = b1 * BATCH < n ? b1 * BATCH : n;
  if (*lo > n) *lo = n;
}

/* ====================================================================================== */
/* Config 2: SELECT COUNT(*) FROM t WHERE chrom = 'c' AND pos >= a AND pos <= b            */
/* predicate tree = RegionPhysicalExpr (region_physical_expr.rs:91-104,220-240) =          */
/* BinaryExpr(And, RegionNamePhysicalExpr (region_name_physical_expr.rs:94-102),           */
/*            PosIntervalPhysicalExpr (pos_interval_physical_expr.rs:79-98))               */
/* ====================================================================================== */
int64_t orc_c2_region_count(const int32_t* chrom_id, const int64_t* pos, const uint8_t* chrom_valid,
                            const uint8_t* pos_valid, int64_t n, const char* const* contig_names,
                            int n_contigs, const char* region, int threads, orc_timing* t) {
  char name[256];
  int64_t ra, rb;
  if (orc_parse_region(region, name, sizeof name, &ra, &rb)) return -1;
  int nlen = (int)strlen(name);
  int T = clamp_threads(threads);
  utf8_col* cols = calloc((size_t)T, sizeof(utf8_col));
  double t0 = now_s();
  int bad = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int p = 0; p < T; p++) {
    int64_t lo, hi;
    part_range(n, T, p, &lo, &hi);
    if (utf8_from_ids(&cols[p], chrom_id, chrom_valid, lo, hi - lo, contig_names, n_contigs)) bad = 1;
  }
  double t1 = now_s();
  int64_t total = 0;
  if (!bad) {
#pragma omp parallel for num_threads(T) schedule(static, 1) reduction(+ : total)
    for (int p = 0; p < T; p++) {
      int64_t lo, hi;
      part_range(n, T, p, &lo, &hi);
      utf8_col* c = &cols[p];
      uint8_t v_eq[BATCH], n_eq[BATCH], v_ge[BATCH], v_le[BATCH], n_pos[BATCH], v_and[BATCH], n_and[BATCH];
      int64_t partial = 0; /* AggregateExec(Partial) count(*) state */
      for (int64_t b = lo; b < hi; b += BATCH) {
        int m = (int)((hi - b) < BATCH ? (hi - b) : BATCH);
        /* chrom = lit : Utf8 eq scalar */
        for (int i = 0; i < m; i++) {
          int64_t r = b - lo + i;
          int32_t o = c->offsets[r], l = c->offsets[r + 1] - o;
          n_eq[i] = c->valid[r];
          v_eq[i] = (l == nlen) && memcmp(c->data + o, name, (size_t)nlen) == 0;
        }
        /* pos >= a ; pos <= b */
        for (int i = 0; i < m; i++) {
          int64_t x = pos[b + i];
          n_pos[i] = (uint8_t)bit_get(pos_valid, b + i);
          v_ge[i] = x >= ra;
          v_le[i] = x <= rb;
        }
        /* Kleene AND: false dominates NULL */
        for (int i = 0; i < m; i++) {
          int iv_v = v_ge[i] && v_le[i], iv_n = n_pos[i]; /* both sides share validity */
          int a_false = n_eq[i] && !v_eq[i], b_false = iv_n && !iv_v;
          v_and[i] = (uint8_t)(v_eq[i] && iv_v);
          n_and[i] = (uint8_t)((n_eq[i] && iv_n) || a_false || b_false);
        }
        /* FilterExec: keep rows where predicate IS TRUE; COUNT(*) over the filtered batch */
        int kept = 0;
        for (int i = 0; i < m; i++) kept += (v_and[i] & n_and[i]);
        partial += kept;
      }
      total += partial; /* AggregateExec(Final): sum of partial counts */
    }
  }
  double t2 = now_s();
  for (int p = 0; p < T; p++) utf8_free(&cols[p]);
  free(cols);
  if (t) { t->seconds_materialize = t1 - t0; t->seconds_exec = t2 - t1; t->threads = T; }
  return bad ? -2 : total;
}

/* ====================================================================================== */
/* Config 3: SELECT reference, COUNT(*) FROM bam WHERE flag & M = V                        */
/*           AND CAST(mapping_quality AS INT) >= Q GROUP BY reference                      */
/* ====================================================================================== */
typedef struct { group_tbl g; int64_t* cnt; int64_t cap; int64_t null_cnt; int has_null; } cnt_state;

int orc_c3_flag_mapq_group_count(const int32_t* flag, const uint8_t* mapq, const uint8_t* mapq_valid,
                                 const int32_t* ref_id, const uint8_t* ref_valid, int64_t n,
                                 const char* const* ref_names, int n_refs, int32_t flag_mask,
                                 int32_t flag_value, int32_t mapq_min, int threads,
                                 int64_t* counts, orc_timing* t) {
  int T = clamp_threads(threads);
  utf8_col* refc = calloc((size_t)T, sizeof(utf8_col));
  utf8_col* mqc = calloc((size_t)T, sizeof(utf8_col));
  cnt_state* st = calloc((size_t)T, sizeof(cnt_state));
  double t0 = now_s();
  int bad = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int p = 0; p < T; p++) {
    int64_t lo, hi;
    part_range(n, T, p, &lo, &hi);
    int64_t m = hi - lo;
    if (utf8_from_ids(&refc[p], ref_id, ref_valid, lo, m, ref_names, n_refs)) bad = 1;
    /* mapping_quality: decimal string, NULL when 255 (exon-bam/src/array_builder.rs:136-143) */
    utf8_col* q = &mqc[p];
    q->n = m;
    q->offsets = malloc(sizeof(int32_t) * (size_t)(m + 1));
    q->data = malloc((size_t)(3 * m + 1));
    q->valid = malloc((size_t)(m > 0 ? m : 1));
    int32_t o = 0;
    for (int64_t i = 0; i < m; i++) {
      q->offsets[i] = o;
      int v = bit_get(mapq_valid, lo + i);
      q->valid[i] = (uint8_t)v;
      if (v) o += sprintf((char*)q->data + o, "%u", (unsigned)mapq[lo + i]);
    }
    q->offsets[m] = o;
  }
  double t1 = now_s();
  if (!bad) {
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int p = 0; p < T; p++) {
      int64_t lo, hi;
      part_range(n, T, p, &lo, &hi);
      cnt_state* s = &st[p];
      gt_init(&s->g);
      s->cap = 64;
      s->cnt = calloc((size_t)s->cap, sizeof(int64_t));
      uint8_t keep[BATCH];
      int32_t casted[BATCH];
      uint8_t cast_ok[BATCH];
      /* filtered `reference` column (arrow filter kernel materialises survivors) */
      int32_t f_off[BATCH + 1];
      uint8_t f_valid[BATCH];
      uint8_t* f_data = malloc(BATCH * 64);
      for (int64_t b = lo; b < hi; b += BATCH) {
        int m = (int)((hi - b) < BATCH ? (hi - b) : BATCH);
        /* CAST(mapping_quality AS INT): parse decimal, NULL stays NULL */
        for (int i = 0; i < m; i++) {
          int64_t r = b - lo + i;
          cast_ok[i] = mqc[p].valid[r];
          int32_t v = 0;
          for (int32_t k = mqc[p].offsets[r]; k < mqc[p].offsets[r + 1]; k++) v = v * 10 + (mqc[p].data[k] - '0');
          casted[i] = v;
        }
        /* (flag & M) = V  AND  casted >= Q   (flag is never NULL) */
        for (int i = 0; i < m; i++) {
          int a = ((flag[b + i] & flag_mask) == flag_value);
          int bq = casted[i] >= mapq_min;
          /* Kleene: a is non-null; result TRUE iff a && cast_ok && bq */
          keep[i] = (uint8_t)(a && cast_ok[i] && bq);
        }
        /* filter -> take reference */
        int k = 0;
        int32_t fo = 0;
        for (int i = 0; i < m; i++)
          if (keep[i]) {
            int64_t r = b - lo + i;
            int32_t o = refc[p].offsets[r], l = refc[p].offsets[r + 1] - o;
            f_off[k] = fo;
            f_valid[k] = refc[p].valid[r];
            memcpy(f_data + fo, refc[p].data + o, (size_t)l);
            fo += l;
            k++;
          }
        f_off[k] = fo;
        /* AggregateExec(Partial): intern keys, count(*) += 1 */
        for (int i = 0; i < k; i++) {
          if (!f_valid[i]) { s->has_null = 1; s->null_cnt++; continue; }
          int64_t gi = gt_intern(&s->g, f_data + f_off[i], f_off[i + 1] - f_off[i]);
          if (gi >= s->cap) {
            int64_t oc = s->cap;
            while (gi >= s->cap) s->cap *= 2;
            s->cnt = realloc(s->cnt, sizeof(int64_t) * (size_t)s->cap);
            memset(s->cnt + oc, 0, sizeof(int64_t) * (size_t)(s->cap - oc));
          }
          s->cnt[gi]++;
        }
      }
      free(f_data);
    }
  }
  /* AggregateExec(Final): merge partial states by key; report keyed by header reference order */
  for (int i = 0; i <= n_refs; i++) counts[i] = 0;
  if (!bad)
    for (int p = 0; p < T; p++) {
      cnt_state* s = &st[p];
      counts[n_refs] += s->null_cnt;
      for (int64_t gi = 0; gi < s->g.n_groups; gi++) {
        int64_t o = s->g.key_off[gi], l = s->g.key_off[gi + 1] - o;
        for (int r = 0; r < n_refs; r++)
          if ((int64_t)strlen(ref_names[r]) == l && memcmp(ref_names[r], s->g.keys + o, (size_t)l) == 0) {
            counts[r] += s->cnt[gi];
            break;
          }
      }
      gt_free(&s->g);
      free(s->cnt);
    }
  double t2 = now_s();
  for (int p = 0; p < T; p++) { utf8_free(&refc[p]); utf8_free(&mqc[p]); }
  free(refc); free(mqc); free(st);
  if (t) { t->seconds_materialize = t1 - t0; t->seconds_exec = t2 - t1; t->threads = T; }
  return bad ? -2 : 0;
}

/* ====================================================================================== */
/* Config 4: SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" <op> thr            */
/*           GROUP BY filter       (filter is List<Utf8>, group identity order-sensitive)  */
/* ====================================================================================== */
typedef struct {
  int32_t* list_off; /* n+1 : item index */
  int32_t* str_off;  /* items+1 */
  uint8_t* data;
  int64_t n;
} list_utf8_col;

typedef struct { group_tbl g; double* sum; uint64_t* cnt; int64_t* rows; int64_t cap; } avg_state;

/* arrow-rs `cmp::{gt,gt_eq,lt,lt_eq,eq,neq}` (arrow-ord 53.3.0, what DataFusion's BinaryExpr calls) compare
 * floats in IEEE-754 totalOrder: -NaN < -inf < ... < -0 < +0 < ... < +inf < +NaN, so NaN > 0.01 is TRUE and
 * -0.0 >= 0.0 is FALSE.  Restated with the usual sign-magnitude -> two's-complement key. */
static inline int64_t f64_total_key(double d) {
  int64_t b;
  memcpy(&b, &d, 8);
  return b ^ (int64_t)(((uint64_t)(b >> 63)) >> 1);
}
static inline int cmp_f64(double x, double thr, int op) {
  const int64_t a = f64_total_key(x), b = f64_total_key(thr);
  switch (op) {
    case 0: return a > b;
    case 1: return a >= b;
    case 2: return a < b;
    case 3: return a <= b;
    case 4: return a == b;
    default: return a != b;
  }
}

int orc_c4_cmp_avg_by_group(const float* af, const uint8_t* af_valid, const float* qual,
                            const uint8_t* qual_valid, const int32_t* filter_id, int64_t n,
                            const char* const* filter_names, int G, double thr, int cmp_op,
                            int threads, double* sum, int64_t* cnt_nonnull, int64_t* cnt_rows,
                            orc_timing* t) {
  int T = clamp_threads(threads);
  list_utf8_col* fc = calloc((size_t)T, sizeof(list_utf8_col));
  avg_state* st = calloc((size_t)T, sizeof(avg_state));
  /* dictionary id -> list of strings (';' separated, "" = empty list: "." in the file) */
  int* d_items = calloc((size_t)G, sizeof(int));
  for (int g = 0; g < G; g++) {
    if (filter_names[g][0] == 0) continue;
    d_items[g] = 1;
    for (const char* c = filter_names[g]; *c; c++) d_items[g] += (*c == ';');
  }
  double t0 = now_s();
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int p = 0; p < T; p++) {
    int64_t lo, hi;
    part_range(n, T, p, &lo, &hi);
    int64_t m = hi - lo, items = 0, bytes = 0;
    for (int64_t i = 0; i < m; i++) {
      int g = filter_id[lo + i];
      items += d_items[g];
      bytes += (int64_t)strlen(filter_names[g]);
    }
    list_utf8_col* c = &fc[p];
    c->n = m;
    c->list_off = malloc(sizeof(int32_t) * (size_t)(m + 1));
    c->str_off = malloc(sizeof(int32_t) * (size_t)(items + 1));
    c->data = malloc((size_t)(bytes + 1));
    int32_t io = 0, so = 0;
    for (int64_t i = 0; i < m; i++) {
      c->list_off[i] = io;
      const char* s = filter_names[filter_id[lo + i]];
      while (*s) {
        const char* e = strchr(s, ';');
        int l = e ? (int)(e - s) : (int)strlen(s);
        c->str_off[io++] = so;
        memcpy(c->data + so, s, (size_t)l);
        so += l;
        s += l + (e ? 1 : 0);
      }
    }
    c->list_off[m] = io;
    c->str_off[io] = so;
  }
  double t1 = now_s();
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int p = 0; p < T; p++) {
    int64_t lo, hi;
    part_range(n, T, p, &lo, &hi);
    list_utf8_col* c = &fc[p];
    avg_state* s = &st[p];
    gt_init(&s->g);
    s->cap = 16;
    s->sum = calloc((size_t)s->cap, sizeof(double));
    s->cnt = calloc((size_t)s->cap, sizeof(uint64_t));
    s->rows = calloc((size_t)s->cap, sizeof(int64_t));
    uint8_t keep[BATCH];
    float fq[BATCH];
    uint8_t fqv[BATCH];
    int32_t sel[BATCH];
    uint8_t rowbuf[4096];
    for (int64_t b = lo; b < hi; b += BATCH) {
      int m = (int)((hi - b) < BATCH ? (hi - b) : BATCH);
      /* CAST(info.AF AS Float64) <op> thr ; NULL AF -> NULL predicate -> dropped */
      for (int i = 0; i < m; i++) {
        double x = (double)af[b + i];
        keep[i] = (uint8_t)(bit_get(af_valid, b + i) && cmp_f64(x, thr, cmp_op));
      }
      int k = 0;
      for (int i = 0; i < m; i++)
        if (keep[i]) {
          fq[k] = qual[b + i];
          fqv[k] = (uint8_t)bit_get(qual_valid, b + i);
          sel[k] = i;
          k++;
        }
      for (int i = 0; i < k; i++) {
        /* GroupValuesRows: row-encode the list key (item count, then len-prefixed items) */
        int64_t r = b - lo + sel[i];
        int32_t i0 = c->list_off[r], i1 = c->list_off[r + 1];
        int64_t w = 0;
        int32_t cntitems = i1 - i0;
        memcpy(rowbuf + w, &cntitems, 4); w += 4;
        for (int32_t it = i0; it < i1; it++) {
          int32_t l = c->str_off[it + 1] - c->str_off[it];
          memcpy(rowbuf + w, &l, 4); w += 4;
          memcpy(rowbuf + w, c->data + c->str_off[it], (size_t)l); w += l;
        }
        int64_t gi = gt_intern(&s->g, rowbuf, w);
        if (gi >= s->cap) {
          int64_t oc = s->cap;
          while (gi >= s->cap) s->cap *= 2;
          s->sum = realloc(s->sum, sizeof(double) * (size_t)s->cap);
          s->cnt = realloc(s->cnt, sizeof(uint64_t) * (size_t)s->cap);
          s->rows = realloc(s->rows, sizeof(int64_t) * (size_t)s->cap);
          for (int64_t z = oc; z < s->cap; z++) { s->sum[z] = 0; s->cnt[z] = 0; s->rows[z] = 0; }
        }
        s->rows[gi]++; /* COUNT(*) */
        if (fqv[i]) {  /* AVG(Float32 -> Float64): sum f64 + count u64 of non-null */
          s->sum[gi] += (double)fq[i];
          s->cnt[gi]++;
        }
      }
    }
  }
  /* Final: merge by key, map keys back to dictionary ids */
  for (int g = 0; g < G; g++) { sum[g] = 0; cnt_nonnull[g] = 0; cnt_rows[g] = 0; }
  for (int p = 0; p < T; p++) {
    avg_state* s = &st[p];
    for (int64_t gi = 0; gi < s->g.n_groups; gi++) {
      int64_t o = s->g.key_off[gi], l = s->g.key_off[gi + 1] - o;
      for (int g = 0; g < G; g++) {
        uint8_t rb[4096];
        int64_t w = 0;
        int32_t ci = d_items[g];
        memcpy(rb + w, &ci, 4); w += 4;
        const char* q = filter_names[g];
        while (*q) {
          const char* e = strchr(q, ';');
          int32_t ll = e ? (int32_t)(e - q) : (int32_t)strlen(q);
          memcpy(rb + w, &ll, 4); w += 4;
          memcpy(rb + w, q, (size_t)ll); w += ll;
          q += ll + (e ? 1 : 0);
        }
        if (w == l && memcmp(rb, s->g.keys + o, (size_t)l) == 0) {
          sum[g] += s->sum[gi];
          cnt_nonnull[g] += (int64_t)s->cnt[gi];
          cnt_rows[g] += s->rows[gi];
          break;
        }
      }
    }
    gt_free(&s->g);
    free(s->sum); free(s->cnt); free(s->rows);
  }
  double t2 = now_s();
  for (int p = 0; p < T; p++) { free(fc[p].list_off); free(fc[p].str_off); free(fc[p].data); }
  free(fc); free(st); free(d_items);
  if (t) { t->seconds_materialize = t1 - t0; t->seconds_exec = t2 - t1; t->threads = T; }
  return 0;
}

/* ====================================================================================== */
/* Config 5: per-position quality histogram                                                */
/*  SELECT p, score, COUNT(*) FROM (unnest(quality_scores_to_list(quality_scores)) with    */
/*  ordinality) GROUP BY p, score   -- the UDF materialises List<Int32> = char - 33        */
/*  (udfs/sequence/quality_score_string_to_list.rs:71-93), then a hash aggregate on        */
/*  (p, score).  Reported as hist[p][score + 33].                                          */
/* ====================================================================================== */
int orc_c5_qual_pos_hist(const int32_t* offsets, const uint8_t* bytes, int64_t n_reads, int lmax,
                         int threads, int64_t* hist, orc_timing* t) {
  int T = clamp_threads(threads);
  double t0 = now_s();
  int64_t** parts = calloc((size_t)T, sizeof(int64_t*));
  int bad = 0;
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int p = 0; p < T; p++) {
    int64_t lo, hi;
    part_range(n_reads, T, p, &lo, &hi);
    /* partial aggregate state keyed by (p:int64, score:int32): open-addressing hash table */
    int64_t cap = 1 << 16;
    uint64_t* keys = malloc(sizeof(uint64_t) * (size_t)cap);
    int64_t* cnt = calloc((size_t)cap, sizeof(int64_t));
    memset(keys, 0xFF, sizeof(uint64_t) * (size_t)cap);
    int32_t* list_vals = malloc(sizeof(int32_t) * (size_t)BATCH * 512);
    int32_t* list_off = malloc(sizeof(int32_t) * (BATCH + 1));
    int64_t used = 0;
    for (int64_t b = lo; b < hi && !bad; b += BATCH) {
      int m = (int)((hi - b) < BATCH ? (hi - b) : BATCH);
      /* quality_scores_to_list over the batch -> List<Int32> */
      int64_t need = (int64_t)offsets[b + m] - offsets[b];
      if (need > (int64_t)BATCH * 512) { bad = 1; break; }
      int32_t o = 0;
      for (int i = 0; i < m; i++) {
        list_off[i] = o;
        for (int32_t k = offsets[b + i]; k < offsets[b + i + 1]; k++) list_vals[o++] = (int32_t)bytes[k] - 33;
      }
      list_off[m] = o;
      /* unnest with position + hash aggregate */
      for (int i = 0; i < m; i++)
        for (int32_t k = list_off[i]; k < list_off[i + 1]; k++) {
          uint64_t key = ((uint64_t)(uint32_t)(k - list_off[i]) << 32) | (uint32_t)list_vals[k];
          int64_t j = (int64_t)(mix64(key) & (uint64_t)(cap - 1));
          while (keys[j] != key && keys[j] != ~0ULL) j = (j + 1) & (cap - 1);
          if (keys[j] == ~0ULL) { keys[j] = key; used++; if (used * 2 > cap) bad = 1; }
          cnt[j]++;
        }
    }
    int64_t* h = calloc((size_t)lmax * 256, sizeof(int64_t));
    for (int64_t j = 0; j < cap; j++)
      if (keys[j] != ~0ULL) {
        int64_t pos = (int64_t)(keys[j] >> 32);
        int32_t sc = (int32_t)(uint32_t)keys[j];
        if (pos < lmax && sc + 33 >= 0 && sc + 33 < 256) h[pos * 256 + sc + 33] += cnt[j];
        else bad = 1;
      }
    parts[p] = h;
    free(keys); free(cnt); free(list_vals); free(list_off);
  }
  memset(hist, 0, sizeof(int64_t) * (size_t)lmax * 256);
  for (int p = 0; p < T; p++) {
    for (int64_t k = 0; k < (int64_t)lmax * 256; k++) hist[k] += parts[p][k];
    free(parts[p]);
  }
  free(parts);
  double t1 = now_s();
  if (t) { t->seconds_materialize = 0; t->seconds_exec = t1 - t0; t->threads = T; }
  return bad ? -2 : 0;
}
