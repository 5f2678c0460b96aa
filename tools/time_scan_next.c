/* time_scan_next.c -- what a native caller sees from exon_hip_scan_next on a scan bound to the GPU pipeline (exon_hip_scan_bind_ctx):
 * batches per second without a Python harness in the way (tools/time_scan_batches.py costs ~30 us per batch itself).
 * usage: time_scan_next FILE {vcf|bam|fastq} [runs] [projection-mask] [info_field]   (EXON_TIME_HOST=1: the host reader)
 * Prints rows, batches, seconds per pass (open .. last batch .. close) -- plain C against include/exon_hip.h. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "exon_hip.h"

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: time_scan_next FILE {vcf|bam|fastq} [runs] [projection-mask] [info_field]   (EXON_TIME_HOST=1: the host reader, no GPU)\n");
    return 2;
  }
  const int runs = argc > 3 ? atoi(argv[3]) : 4;
  exon_hip_ctx* ctx = NULL;
  if (exon_hip_ctx_create(0, &ctx)) {
    fprintf(stderr, "%s\n", exon_hip_last_error(NULL));
    return 1;
  }
  double best = 1e30;
  for (int rep = 0; rep < runs; ++rep) {
    exon_hip_scan_options o;
    memset(&o, 0, sizeof o);
    o.format = !strcmp(argv[2], "bam") ? EXON_HIP_FORMAT_BAM : !strcmp(argv[2], "fastq") ? EXON_HIP_FORMAT_FASTQ : EXON_HIP_FORMAT_VCF;
    const int host = getenv("EXON_TIME_HOST") != NULL;
    o.gpu_parse = host ? 0 : 1;
    o.region = getenv("EXON_TIME_REGION");  /* a pushed-down region filter, e.g. "1" or "7:50000000-100000000" */
    o.projection = argc > 4 ? strtoull(argv[4], NULL, 0) : 0;
    o.info_field = argc > 5 ? argv[5] : (o.format == EXON_HIP_FORMAT_VCF ? "AF" : NULL);
    const double t0 = now_s();
    exon_hip_scan* s = NULL;
    if (exon_hip_scan_open(argv[1], &o, &s) || (!host && exon_hip_scan_bind_ctx(s, ctx))) {
      fprintf(stderr, "%s\n", exon_hip_last_error(NULL));
      return 1;
    }
    long long rows = 0, batches = 0;
    for (;;) {
      struct ArrowArray a;
      const int rc = exon_hip_scan_next(s, &a);
      if (rc == 1) break;
      if (rc) {
        fprintf(stderr, "%s\n", exon_hip_last_error(ctx));
        return 1;
      }
      rows += a.length;
      ++batches;
      a.release(&a);
    }
    int32_t dec = 0, inf = 0;
    exon_hip_scan_decoded_on_gpu(s, &dec, &inf);
    exon_hip_scan_close(s);
    const double dt = now_s() - t0;
    if (rep > 0 && dt < best) best = dt;
    printf("pass %d: %lld rows in %lld batches, %.4f s (%.1f Mrows/s), decoded / inflated on the GPU %d / %d\n", rep, rows, batches, dt, rows / dt / 1e6, dec, inf);
  }
  if (runs > 1) printf("best of the warm passes: %.4f s\n", best);
  exon_hip_ctx_destroy(ctx);
  return 0;
}
