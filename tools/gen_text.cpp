// gen_text.cpp -- developer tool: writes the synthetic config-4 table as VCF text (same counter-based generator as
// kernels.hip / oracle) so the decode -> HBM -> kernel pipeline can be timed end to end on real files.
// `gen_text fastq <reads> <out> [read_len=150] [ragged=0]` writes 4-line FASTQ records (config 5 end to end).
// build: g++ -O2 -std=c++17 tools/gen_text.cpp -o tools/bin/gen_text      run: gen_text vcf <rows> <out.vcf>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
static inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static inline uint64_t rnd(uint64_t seed, uint64_t col, uint64_t i) { return mix64(seed + col * 0xD1B54A32D192ED03ULL + (i + 1) * 0x9E3779B97F4A7C15ULL); }
static inline uint32_t pct_thr(int p) { return (uint32_t)((((uint64_t)p) << 32) / 100); }
int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: gen_text vcf|fastq <rows> <out> [read_len] [ragged]\n"); return 2; }
  const int64_t n = (int64_t)atof(argv[2]);
  FILE* f = fopen(argv[3], "wb");
  if (!f) return 1;
  static char buf[1 << 22];
  setvbuf(f, buf, _IOFBF, sizeof buf);
  if (!strcmp(argv[1], "fastq")) {
    const int L = argc > 4 ? atoi(argv[4]) : 150;
    const bool ragged = argc > 5 && atoi(argv[5]) != 0;
    std::string seq((size_t)L, 'A'), qual((size_t)L, '!');
    for (int64_t i = 0; i < n; ++i) {
      int len = L;
      if (ragged) len = L - (int)(rnd(5, 2, (uint64_t)i) % (uint64_t)(L / 4 + 1));
      for (int p = 0; p < len; p += 8) {
        uint64_t a = rnd(5, 0, (uint64_t)(i * ((L + 7) / 8) + p / 8)), b = rnd(5, 1, (uint64_t)(i * ((L + 7) / 8) + p / 8));
        for (int k = 0; k < 8 && p + k < len; ++k) {
          seq[(size_t)(p + k)] = "ACGT"[a & 3];
          qual[(size_t)(p + k)] = (char)(33 + (b & 0xFF) % 42);
          a >>= 8;
          b >>= 8;
        }
      }
      fprintf(f, "@read%lld sample=%d\n", (long long)i, (int)(i % 7));
      fwrite(seq.data(), 1, (size_t)len, f);
      fputs("\n+\n", f);
      fwrite(qual.data(), 1, (size_t)len, f);
      fputc('\n', f);
    }
    fclose(f);
    return 0;
  }
  fputs("##fileformat=VCFv4.3\n##contig=<ID=1>\n##INFO=<ID=AF,Number=1,Type=Float,Description=\"AF\">\n"
        "##FILTER=<ID=q10,Description=\"q\">\n##FILTER=<ID=s50,Description=\"s\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n", f);
  const char* FILT[5] = {"PASS", ".", "q10", "q10;s50", "s50"};
  const uint32_t t0 = pct_thr(85), t1 = pct_thr(90), t2 = pct_thr(96), t3 = pct_thr(99);
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t r0 = rnd(4, 0, i), r1 = rnd(4, 1, i), r2 = rnd(4, 2, i);
    const uint32_t e = (uint32_t)((((r0 >> 23) & 0xFF) * 14) >> 8);
    uint32_t bits = ((126u - e) << 23) | (uint32_t)(r0 & 0x7FFFFF);
    float a; memcpy(&a, &bits, 4);
    if (((r0 >> 31) & 0x3FF) == 0) a = 0.01f;
    const bool av = (r0 >> 44) >= 10486, qv = (r1 >> 44) >= 31457;
    const uint32_t kq = (uint32_t)(r1 & 0xFFFFFFFFu) % 10000u;
    const uint32_t u = (uint32_t)(r2 >> 32);
    const int fid = (u >= t0) + (u >= t1) + (u >= t2) + (u >= t3);
    char q[16];
    if (qv) snprintf(q, sizeof q, "%u.%u", kq / 10, kq % 10); else strcpy(q, ".");
    if (av) fprintf(f, "1\t%lld\t.\tA\tC\t%s\t%s\tAF=%.9g\n", (long long)(i + 1), q, FILT[fid], (double)a);
    else fprintf(f, "1\t%lld\t.\tA\tC\t%s\t%s\t.\n", (long long)(i + 1), q, FILT[fid]);
  }
  fclose(f);
  return 0;
}
