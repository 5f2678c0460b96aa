/* abi_harness.c -- drives the C ABI of include/exon_hip.h exactly as the Rust shim does (shim/src/lib.rs, ChildBatches source):
 *   exon_hip_ctx_create -> exon_hip_plan_create -> exon_hip_stream_open
 *     -> exon_hip_stream_push (hand-built Arrow C Data Interface struct arrays, MOVED: released once by the library)
 *     -> exon_hip_stream_finish_arrow -> read the state batch -> release out / out_schema -> close / destroy.
 * Plain C11, no Arrow library, no Python: what a cgo / JNI / Rust `extern "C"` host sees.  The expected state is computed by
 * a scalar loop in this file (config 4's query on 3 x 8192 rows with NULLs).
 *   build: gcc -std=c11 -Wall -Werror tests/abi_harness.c -Iinclude -Lexon_amd/lib -lexon_hip -Wl,-rpath,$PWD/exon_amd/lib -lm
 *   exit 0 + "OK ..." on a GPU box; with --allow-no-device it prints "NO_DEVICE ..." and exits 0 when ctx_create reports
 *   that no HIP device is visible (the CPU test: the library links and fails loudly, no fallback). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "exon_hip.h"

#define ROWS 8192
#define BATCHES 3
#define G 5

static int releases = 0;

struct batch_mem { /* everything one exported batch owns */
  float *af, *qual;
  int32_t* fid;
  uint8_t *af_valid, *qual_valid;
  const void* buf_af[2];
  const void* buf_qual[2];
  const void* buf_fid[2];
  const void* buf_top[1];
  struct ArrowArray kids[3];
  struct ArrowArray* kid_ptrs[3];
};

static void release_child(struct ArrowArray* a) { a->release = NULL; }
static void release_batch(struct ArrowArray* a) {
  struct batch_mem* m = (struct batch_mem*)a->private_data;
  for (int i = 0; i < 3; ++i)
    if (m->kids[i].release) m->kids[i].release(&m->kids[i]);
  free(m->af); free(m->qual); free(m->fid); free(m->af_valid); free(m->qual_valid);
  free(m);
  a->release = NULL;
  ++releases;
}

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 20); }

static void child(struct ArrowArray* a, int64_t n, int64_t nulls, const void** bufs) {
  memset(a, 0, sizeof *a);
  a->length = n; a->null_count = nulls; a->n_buffers = 2; a->buffers = bufs; a->release = release_child;
}

/* one struct batch {af: f32?, qual: f32?, filter: i32 (dictionary ids)}; accumulates the expected state */
static void make_batch(struct ArrowArray* top, int64_t* cnn, int64_t* crow, double* sum) {
  struct batch_mem* m = calloc(1, sizeof *m);
  m->af = malloc(ROWS * 4); m->qual = malloc(ROWS * 4); m->fid = malloc(ROWS * 4);
  m->af_valid = calloc(ROWS / 8, 1); m->qual_valid = calloc(ROWS / 8, 1);
  int64_t af_nulls = 0, q_nulls = 0;
  for (int i = 0; i < ROWS; ++i) {
    m->af[i] = (float)(rnd() % 1000) / 5000.0f;          /* 0 .. 0.2, 0.01 itself occurs */
    m->qual[i] = (float)(rnd() % 8000) / 8.0f;           /* eighths: every partial sum is exact */
    m->fid[i] = (int32_t)(rnd() % G);
    const int av = rnd() % 100 != 0, qv = rnd() % 33 != 0;
    if (av) m->af_valid[i >> 3] |= (uint8_t)(1u << (i & 7)); else ++af_nulls;
    if (qv) m->qual_valid[i >> 3] |= (uint8_t)(1u << (i & 7)); else ++q_nulls;
    if (av && (double)m->af[i] > 0.01) {                 /* CAST(af AS DOUBLE) > 0.01, keep TRUE */
      ++crow[m->fid[i]];
      if (qv) { ++cnn[m->fid[i]]; sum[m->fid[i]] += (double)m->qual[i]; }
    }
  }
  m->buf_af[0] = m->af_valid; m->buf_af[1] = m->af;
  m->buf_qual[0] = m->qual_valid; m->buf_qual[1] = m->qual;
  m->buf_fid[0] = NULL; m->buf_fid[1] = m->fid;
  child(&m->kids[0], ROWS, af_nulls, m->buf_af);
  child(&m->kids[1], ROWS, q_nulls, m->buf_qual);
  child(&m->kids[2], ROWS, 0, m->buf_fid);
  for (int i = 0; i < 3; ++i) m->kid_ptrs[i] = &m->kids[i];
  m->buf_top[0] = NULL;
  memset(top, 0, sizeof *top);
  top->length = ROWS; top->n_buffers = 1; top->buffers = m->buf_top; top->n_children = 3; top->children = m->kid_ptrs;
  top->release = release_batch; top->private_data = m;
}

#define CHECK(expr) do { int rc_ = (expr); if (rc_ != EXON_HIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, exon_hip_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int allow_no_device = argc > 1 && strcmp(argv[1], "--allow-no-device") == 0;
  exon_hip_ctx* ctx = NULL;
  if (exon_hip_abi_version() != 5) { fprintf(stderr, "ABI version %d\n", exon_hip_abi_version()); return 1; }
  int rc = exon_hip_ctx_create(0, &ctx);
  if (rc != EXON_HIP_OK) {
    const char* msg = exon_hip_last_error(NULL);
    if (allow_no_device && rc == EXON_HIP_EDEVICE && ctx == NULL) { printf("NO_DEVICE %s\n", msg); return 0; }
    fprintf(stderr, "exon_hip_ctx_create -> %d: %s\n", rc, msg);
    return 1;
  }
  exon_hip_plan_desc d;
  memset(&d, 0, sizeof d);
  d.kind = EXON_HIP_PLAN_CMP_AVG_BY_GROUP; d.n_groups = G; d.cmp_op = EXON_HIP_GT; d.threshold = 0.01;
  d.columns[0] = 0; d.columns[1] = 1; d.columns[2] = 2;
  exon_hip_plan* plan = NULL;
  exon_hip_stream* st = NULL;
  CHECK(exon_hip_plan_create(ctx, &d, &plan));
  int64_t n_i64 = 0, n_f64 = 0;
  CHECK(exon_hip_plan_state_size(plan, &n_i64, &n_f64));
  if (n_i64 != 2 * G || n_f64 != G) { fprintf(stderr, "state size %lld + %lld\n", (long long)n_i64, (long long)n_f64); return 1; }
  CHECK(exon_hip_stream_open(plan, 0, &st));
  int64_t cnn[G] = {0}, crow[G] = {0};
  double sum[G] = {0};
  for (int b = 0; b < BATCHES; ++b) {
    struct ArrowArray batch;
    make_batch(&batch, cnn, crow, sum);
    CHECK(exon_hip_stream_push(st, &batch));
    if (batch.release != NULL) { fprintf(stderr, "batch %d was not released (moved) by the library\n", b); return 1; }
  }
  struct ArrowArray out;
  struct ArrowSchema out_schema;
  CHECK(exon_hip_stream_finish_arrow(st, &out, &out_schema));
  /* a small batch may be HELD by the stream and released when its staging slot is flushed: by now every one is released, once */
  if (releases != BATCHES) { fprintf(stderr, "%d releases for %d batches\n", releases, BATCHES); return 1; }
  /* {group: i32, avg[count]: u64, avg[sum]: f64, count(*)[count]: i64}, observed groups only */
  if (strcmp(out_schema.format, "+s") != 0 || out_schema.n_children != 4 || out.n_children != 4) { fprintf(stderr, "state batch shape\n"); return 1; }
  static const char* want_fmt[4] = {"i", "L", "g", "l"};
  for (int c = 0; c < 4; ++c)
    if (strcmp(out_schema.children[c]->format, want_fmt[c]) != 0) { fprintf(stderr, "state column %d has format %s\n", c, out_schema.children[c]->format); return 1; }
  const int32_t* key = (const int32_t*)out.children[0]->buffers[1];
  const uint64_t* acnt = (const uint64_t*)out.children[1]->buffers[1];
  const double* asum = (const double*)out.children[2]->buffers[1];
  const int64_t* rows = (const int64_t*)out.children[3]->buffers[1];
  int seen = 0;
  for (int64_t r = 0; r < out.length; ++r) {
    const int g = key[r];
    if (g < 0 || g >= G || (int64_t)acnt[r] != cnn[g] || rows[r] != crow[g] || asum[r] != sum[g]) {
      fprintf(stderr, "group %d: got (%llu, %.17g, %lld), want (%lld, %.17g, %lld)\n", g, (unsigned long long)acnt[r], asum[r], (long long)rows[r],
              (long long)cnn[g], sum[g], (long long)crow[g]);
      return 1;
    }
    ++seen;
  }
  int want_groups = 0;
  for (int g = 0; g < G; ++g) want_groups += crow[g] != 0;
  if (seen != want_groups) { fprintf(stderr, "%d groups, expected %d\n", seen, want_groups); return 1; }
  out.release(&out);
  out_schema.release(&out_schema);
  if (out.release != NULL || out_schema.release != NULL) { fprintf(stderr, "release callbacks must clear themselves\n"); return 1; }
  CHECK(exon_hip_stream_close(st));
  CHECK(exon_hip_plan_destroy(plan));
  CHECK(exon_hip_ctx_destroy(ctx));
  printf("OK %d groups, %d batches moved, avg[0] = %.6f\n", seen, releases, cnn[0] ? sum[0] / (double)cnn[0] : 0.0);
  return 0;
}
