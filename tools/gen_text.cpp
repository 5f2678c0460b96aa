// gen_text.cpp -- developer tool: writes the synthetic config-4 table as VCF text (same counter-based generator as
// kernels.hip / oracle) so the decode -> HBM -> kernel pipeline can be timed end to end on real files.
// `gen_text fastq <reads> <out> [read_len=150] [ragged=0]` writes 4-line FASTQ records (config 5 end to end);
// `gen_text bam <reads> <out> [read_len=100]` an uncompressed BAM stream (bgzip it to get a .bam; config 3 end to end).
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
  if (argc < 4) { fprintf(stderr, "usage: gen_text vcf|bcf|fastq|bam|sam <rows> <out> [read_len] [ragged]\n"); return 2; }
  const int64_t n = (int64_t)atof(argv[2]);
  FILE* f = fopen(argv[3], "wb");
  if (!f) return 1;
  static char buf[1 << 22];
  setvbuf(f, buf, _IOFBF, sizeof buf);
  if (!strcmp(argv[1], "sam")) {
    // text SAM with the same flag / reference / mapq mix as `gen_text bam`: `gen_text sam <reads> <out> [read_len=100]`
    const int L = argc > 4 ? atoi(argv[4]) : 100;
    const int NREF = 25;
    fputs("@HD\tVN:1.6\tSO:unsorted\n", f);
    for (int r = 0; r < NREF; ++r) fprintf(f, "@SQ\tSN:chr%d\tLN:250000000\n", r + 1);
    static const uint16_t FLAGS[12] = {99, 147, 83, 163, 0, 16, 4, 77, 141, 1024 + 99, 256 + 16, 2048 + 0};
    std::string seq((size_t)L, 'A'), qual((size_t)L, 'I');
    for (int64_t i = 0; i < n; ++i) {
      const uint64_t a = rnd(3, 0, (uint64_t)i), b = rnd(3, 1, (uint64_t)i), c = rnd(3, 2, (uint64_t)i);
      const uint16_t flag = FLAGS[a % 12];
      const bool unmapped = (flag & 4) != 0;
      const uint32_t mq = (uint32_t)(b % 100);
      const int mapq = mq < 2 ? 255 : mq < 10 ? 0 : mq < 22 ? (int)(1 + b % 29) : mq < 42 ? (int)(30 + b % 30) : 60;
      const int len = L - (int)((b >> 32) % (uint64_t)(L / 3 + 1));
      char cigar[64] = "*";
      if (!unmapped) {
        const int kind = (int)((c >> 12) % 4);
        if (kind == 0) snprintf(cigar, sizeof cigar, "%dM", len);
        else if (kind == 1) snprintf(cigar, sizeof cigar, "5S%dM", len - 5);
        else if (kind == 2) snprintf(cigar, sizeof cigar, "%dM%dN%dM", len / 2, (int)(100 + (c >> 20) % 5000), len - len / 2);
        else snprintf(cigar, sizeof cigar, "%dM2D3M", len - 3);
      }
      if (unmapped) fprintf(f, "read%lld\t%u\t*\t0\t%d\t*\t*\t0\t0\t", (long long)i, (unsigned)flag, mapq);
      else fprintf(f, "read%lld\t%u\tchr%d\t%d\t%d\t%s\t*\t0\t0\t", (long long)i, (unsigned)flag, (int)((a >> 8) % NREF) + 1,
                   (int)((a >> 16) % 249000000) + 1, mapq, cigar);
      fwrite(seq.data(), 1, (size_t)len, f);
      fputc('\t', f);
      fwrite(qual.data(), 1, (size_t)len, f);
      fputs(i % 3 ? "\tNM:i:1\n" : "\n", f);
    }
    fclose(f);
    return 0;
  }
  if (!strcmp(argv[1], "bcf")) {
    // uncompressed BCF2 stream with the SAME rows as `gen_text vcf` (bgzip it to get a .bcf): `gen_text bcf <rows> <out>`
    std::string text = "##fileformat=VCFv4.3\n##FILTER=<ID=PASS,Description=\"All filters passed\",IDX=0>\n##contig=<ID=1,IDX=0>\n"
                       "##FILTER=<ID=q10,Description=\"q\",IDX=1>\n##FILTER=<ID=s50,Description=\"s\",IDX=2>\n"
                       "##INFO=<ID=AF,Number=1,Type=Float,Description=\"AF\",IDX=3>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
    text.push_back('\0');
    fwrite("BCF\2\2", 1, 5, f);
    const uint32_t lt = (uint32_t)text.size();
    fwrite(&lt, 4, 1, f);
    fwrite(text.data(), 1, text.size(), f);
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
      // the text twin prints AF with 9 significant digits and QUAL as k/10: parse them back the same way
      char tmp[32];
      snprintf(tmp, sizeof tmp, "%.9g", (double)a);
      const float af = strtof(tmp, nullptr);
      snprintf(tmp, sizeof tmp, "%u.%u", kq / 10, kq % 10);
      const float q = strtof(tmp, nullptr);
      uint8_t rec[96];
      size_t o = 8;
      auto p32 = [&](uint32_t v) { memcpy(rec + o, &v, 4); o += 4; };
      p32(0);                      // CHROM
      p32((uint32_t)i);            // POS (0-based) = i  ->  1-based i + 1 as in the text twin
      p32(1);                      // rlen
      uint32_t qb = 0x7F800001u;
      if (qv) memcpy(&qb, &q, 4);
      p32(qb);
      p32((uint32_t)(av ? 1 : 0) | (2u << 16));  // n_info | n_allele << 16
      p32(0);                      // n_fmt << 24 | n_sample
      rec[o++] = 0x07;             // ID: missing
      rec[o++] = 0x17; rec[o++] = 'A';
      rec[o++] = 0x17; rec[o++] = 'C';
      switch (fid) {               // PASS, ".", q10, q10;s50, s50
        case 0: rec[o++] = 0x11; rec[o++] = 0; break;
        case 1: rec[o++] = 0x00; break;
        case 2: rec[o++] = 0x11; rec[o++] = 1; break;
        case 3: rec[o++] = 0x21; rec[o++] = 1; rec[o++] = 2; break;
        default: rec[o++] = 0x11; rec[o++] = 2; break;
      }
      if (av) {
        rec[o++] = 0x11; rec[o++] = 3;  // key AF
        rec[o++] = 0x15;                // one float
        memcpy(rec + o, &af, 4); o += 4;
      }
      const uint32_t ls = (uint32_t)(o - 8), li = 0;
      memcpy(rec, &ls, 4);
      memcpy(rec + 4, &li, 4);
      fwrite(rec, 1, o, f);
    }
    fclose(f);
    return 0;
  }
  if (!strcmp(argv[1], "bam")) {
    // uncompressed BAM stream (header + records; tools/bin/bgzip turns it into a .bam): `gen_text bam <reads> <out> [read_len=100]`
    const int L = argc > 4 ? atoi(argv[4]) : 100;
    auto w32 = [&](int32_t v) { fwrite(&v, 4, 1, f); };
    const int NREF = 25;
    std::string text = "@HD\tVN:1.6\tSO:unsorted\n";
    for (int r = 0; r < NREF; ++r) text += "@SQ\tSN:chr" + std::to_string(r + 1) + "\tLN:250000000\n";
    fwrite("BAM\1", 1, 4, f);
    w32((int32_t)text.size());
    fwrite(text.data(), 1, text.size(), f);
    w32(NREF);
    for (int r = 0; r < NREF; ++r) {
      const std::string nm = "chr" + std::to_string(r + 1);
      w32((int32_t)nm.size() + 1);
      fwrite(nm.c_str(), 1, nm.size() + 1, f);
      w32(250000000);
    }
    static const uint16_t FLAGS[12] = {99, 147, 83, 163, 0, 16, 4, 77, 141, 1024 + 99, 256 + 16, 2048 + 0};
    std::string rec;
    for (int64_t i = 0; i < n; ++i) {
      const uint64_t a = rnd(3, 0, (uint64_t)i), b = rnd(3, 1, (uint64_t)i), c = rnd(3, 2, (uint64_t)i);
      const uint16_t flag = FLAGS[a % 12];
      const bool unmapped = (flag & 4) != 0;
      const int32_t ref = unmapped ? -1 : (int32_t)((a >> 8) % NREF);
      const int32_t pos = unmapped ? -1 : (int32_t)((a >> 16) % 249000000);
      const uint32_t mq = (uint32_t)(b % 100);
      const uint8_t mapq = mq < 2 ? 255 : mq < 10 ? 0 : mq < 22 ? (uint8_t)(1 + b % 29) : mq < 42 ? (uint8_t)(30 + b % 30) : 60;
      const int len = L - (int)((b >> 32) % (uint64_t)(L / 3 + 1));
      char name[40];
      const int ln = snprintf(name, sizeof name, "read%lld:%u", (long long)i, (unsigned)(c & 0xFFF)) + 1;
      uint32_t cig[3];
      int ncig = 0;
      if (!unmapped) {
        const int kind = (int)((c >> 12) % 4);
        if (kind == 0) cig[ncig++] = (uint32_t)len << 4 | 0;                                    // lenM
        else if (kind == 1) { cig[ncig++] = 5u << 4 | 4; cig[ncig++] = (uint32_t)(len - 5) << 4 | 0; }  // 5S..M
        else if (kind == 2) { cig[ncig++] = (uint32_t)(len / 2) << 4 | 0; cig[ncig++] = (uint32_t)(100 + (c >> 20) % 5000) << 4 | 3; cig[ncig++] = (uint32_t)(len - len / 2) << 4 | 0; }  // M N M
        else { cig[ncig++] = (uint32_t)(len - 3) << 4 | 0; cig[ncig++] = 2u << 4 | 2; cig[ncig++] = 3u << 4 | 0; }  // M 2D M
      }
      const int aux = (int)((c >> 40) % 3) * 7;  // 0, 7 or 14 bytes of aux (NM:i / AS:i as 'C' + padding tags)
      const int32_t bs = 32 + ln + 4 * ncig + (len + 1) / 2 + len + aux;
      rec.assign((size_t)bs + 4, '\0');
      auto p32 = [&](size_t o, int32_t v) { memcpy(&rec[o], &v, 4); };
      p32(0, bs); p32(4, ref); p32(8, pos);
      rec[12] = (char)ln; rec[13] = (char)mapq;
      const uint16_t bin = 4680, nc = (uint16_t)ncig;
      memcpy(&rec[14], &bin, 2); memcpy(&rec[16], &nc, 2); memcpy(&rec[18], &flag, 2);
      p32(20, len); p32(24, -1); p32(28, -1); p32(32, 0);
      memcpy(&rec[36], name, (size_t)ln);
      size_t o = 36 + (size_t)ln;
      memcpy(&rec[o], cig, 4u * (size_t)ncig); o += 4u * (size_t)ncig;
      // bases: 4-bit codes of A/C/G/T; qualities: binned Illumina-like values in runs (what real BAMs look like to DEFLATE)
      static const uint8_t NIB[4] = {1, 2, 4, 8}, QBIN[4] = {37, 37, 25, 11};
      for (int k = 0; k < (len + 1) / 2; ++k) {
        const uint64_t x = rnd(3, 3, (uint64_t)(i * 128 + k));
        rec[o + (size_t)k] = (char)(NIB[x & 3] << 4 | NIB[(x >> 2) & 3]);
      }
      o += (size_t)(len + 1) / 2;
      for (int k = 0; k < len;) {
        const uint64_t x = rnd(3, 4, (uint64_t)(i * 256 + k));
        const int run = 1 + (int)((x >> 8) % 12);
        for (int j = 0; j < run && k < len; ++j, ++k) rec[o + (size_t)k] = (char)QBIN[x & 3];
      }
      o += (size_t)len;
      for (int k = 0; k < aux; k += 7) memcpy(&rec[o + (size_t)k], "NMC\5ASC", 7);
      fwrite(rec.data(), 1, rec.size(), f);
    }
    fclose(f);
    return 0;
  }
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
