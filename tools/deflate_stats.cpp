// deflate_stats.cpp -- token statistics of the DEFLATE streams inside a BGZF file (developer harness, CPU only).
// What a "wide" decode round of inflate.hip (64 lanes = 64 consecutive bit offsets) would see: symbols per 64-bit round, matches per
// round, how many matches reach beyond the LDS ring, overlap their own output or the output of their own round.
//   g++ -O2 -o tools/bin/deflate_stats tools/deflate_stats.cpp && tools/bin/deflate_stats file.gz [members]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Bits {
  const uint8_t* p;
  size_t n, bit = 0;
  uint32_t peek(int k) const {
    uint64_t v = 0;
    const size_t b = bit >> 3;
    for (int i = 0; i < 8 && b + i < n; ++i) v |= (uint64_t)p[b + i] << (8 * i);
    return (uint32_t)((v >> (bit & 7)) & ((1ull << k) - 1));
  }
  uint32_t take(int k) {
    const uint32_t v = peek(k);
    bit += k;
    return v;
  }
};
struct Huff {
  uint16_t count[16], sym[320];
  int maxlen = 0;
  bool build(const uint8_t* lens, int n) {
    memset(count, 0, sizeof count);
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    for (int i = 0; i < n; ++i)
      if (lens[i]) sym[offs[lens[i]]++] = (uint16_t)i;
    for (int l = 15; l > 0; --l)
      if (count[l]) { maxlen = l; break; }
    return true;
  }
  int decode(Bits& b, int* len_out) const {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
      code |= (int)b.take(1);
      const int c = count[len];
      if (code - c < first) { *len_out = len; return sym[index + (code - first)]; }
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};
static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Tok { uint32_t bitpos, bits, len, dist; bool longcode, longlen, longdist; };

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  const int want = argc > 2 ? atoi(argv[2]) : 200;
  std::vector<uint8_t> raw(64u << 20);
  raw.resize(fread(raw.data(), 1, raw.size(), f));
  fclose(f);
  size_t o = 0;
  int members = 0;
  uint64_t nsym = 0, nlit = 0, nmatch = 0, nbits = 0, nout = 0, long_lit = 0, long_dist = 0, len_gt64 = 0, overlap = 0, nblocks = 0;
  uint64_t dist_le[5] = {0, 0, 0, 0, 0};
  const uint32_t dist_cut[5] = {1024 - 258, 2048 - 258, 4096 - 258, 8192 - 258, 32768};
  uint64_t rounds = 0, round_syms = 0, round_matches = 0, round_dep = 0, round_hist[20] = {0}, rm_hist[12] = {0}, round_out = 0, round_out_gt64 = 0;
  uint64_t matchbytes = 0, lit_run_hist[10] = {0}, long_len = 0, r_longlen = 0, r_longdist = 0, r_overlap = 0, r_any_slow = 0, r_longlit = 0;
  static uint64_t cl_hist[16] = {0};
  while (o + 18 <= raw.size() && members < want) {
    const uint8_t* h = raw.data() + o;
    const size_t xlen = h[10] | (h[11] << 8);
    const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1;
    if (o + bsize > raw.size()) break;
    Bits b{h + 12 + xlen, bsize - 12 - xlen - 8};
    std::vector<Tok> toks;
    bool last = false;
    while (!last) {
      last = b.take(1);
      const int bt = b.take(2);
      ++nblocks;
      if (bt == 0) {
        b.bit = (b.bit + 7) & ~size_t(7);
        const uint32_t len = b.take(16);
        b.take(16);
        b.bit += 8 * (size_t)len;
        nout += len;
        continue;
      }
      uint8_t lens[320];
      Huff lit, dist;
      if (bt == 1) {
        for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        lit.build(lens, 288);
        for (int s = 0; s < 32; ++s) lens[s] = 5;
        dist.build(lens, 32);
      } else {
        const int hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
        static const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) cl[ord[i]] = (uint8_t)b.take(3);
        Huff clh;
        clh.build(cl, 19);
        int i = 0, prev = 0;
        while (i < hlit + hdist) {
          int l;
          const int s = clh.decode(b, &l);
          if (s < 16) { lens[i++] = (uint8_t)s; prev = s; }
          else {
            int rep, val = 0;
            if (s == 16) { val = prev; rep = 3 + b.take(2); }
            else if (s == 17) rep = 3 + b.take(3);
            else rep = 11 + b.take(7);
            while (rep--) lens[i++] = (uint8_t)val;
            prev = val;
          }
        }
        lit.build(lens, hlit);
        dist.build(lens + hlit, hdist);
      }
      for (;;) {
        Tok t{};
        t.bitpos = (uint32_t)b.bit;
        int l;
        const int s = lit.decode(b, &l);
        if (s < 0) { fprintf(stderr, "bad code\n"); return 2; }
        if (s == 256) break;
        cl_hist[l]++;
        if (s < 256) {
          t.len = 1;
          t.dist = 0;
          t.longcode = l > 9;
        } else {
          t.longcode = l > 9;
          t.longlen = l > 9;
          t.len = LBASE[s - 257] + b.take(LEXT[s - 257]);
          int dl;
          const int d = dist.decode(b, &dl);
          if (dl > 8) t.longcode = true, t.longdist = true, ++long_dist;
          t.dist = DBASE[d] + b.take(DEXT[d]);
        }
        if (t.longcode && !t.dist) ++long_lit;
        t.bits = (uint32_t)b.bit - t.bitpos;
        toks.push_back(t);
      }
    }
    // statistics over the member's tokens
    uint32_t litrun = 0;
    for (const Tok& t : toks) {
      ++nsym;
      nbits += t.bits;
      nout += t.len;
      if (!t.dist) { ++nlit; ++litrun; continue; }
      lit_run_hist[litrun > 9 ? 9 : litrun]++;
      litrun = 0;
      ++nmatch;
      matchbytes += t.len;
      if (t.len > 64) ++len_gt64;
      if (t.longlen) ++long_len;
      if (t.dist < t.len) ++overlap;
      for (int k = 0; k < 5; ++k)
        if (t.dist <= dist_cut[k]) dist_le[k]++;
    }
    // wide rounds: a round takes every symbol that STARTS within 64 bits of the round's first bit
    for (size_t i = 0; i < toks.size();) {
      const uint32_t p0 = toks[i].bitpos;
      uint32_t ns = 0, nm = 0, outb = 0;
      bool dep = false, s_ll = false, s_ld = false, s_ov = false;
      while (i < toks.size() && toks[i].bitpos - p0 < 64) {
        if (toks[i].longcode && !toks[i].dist) ++r_longlit;
        if (toks[i].dist) {
          ++nm;
          if (toks[i].longlen) s_ll = true;
          if (toks[i].longdist) s_ld = true;
          if (toks[i].dist < toks[i].len) s_ov = true;
          if (toks[i].dist < outb + toks[i].len) dep = true;  // reads bytes this round wrote (or its own)
        }
        outb += toks[i].len;
        ++ns;
        ++i;
      }
      ++rounds;
      round_syms += ns;
      round_matches += nm;
      round_dep += dep;
      r_longlen += s_ll; r_longdist += s_ld; r_overlap += s_ov; r_any_slow += (s_ll || s_ld || s_ov);
      round_out += outb;
      round_out_gt64 += outb > 64;
      round_hist[ns > 19 ? 19 : ns]++;
      rm_hist[nm > 11 ? 11 : nm]++;
    }
    ++members;
    o += bsize;
  }
  printf("%d members, %llu deflate blocks, %.0f out bytes / member, %.0f symbols / member, %.2f bits / symbol, %.2f out bytes / symbol\n", members,
         (unsigned long long)nblocks, (double)nout / members, (double)nsym / members, (double)nbits / nsym, (double)nout / nsym);
  printf("literals %.1f %% of symbols, matches %.1f %% (avg len %.1f, %.1f %% of bytes); long-code literals/lengths %.2f %%, long distance codes %.2f %% of matches\n",
         100.0 * nlit / nsym, 100.0 * nmatch / nsym, (double)matchbytes / (nmatch ? nmatch : 1), 100.0 * matchbytes / nout, 100.0 * long_lit / nsym,
         100.0 * long_dist / (nmatch ? nmatch : 1));
  printf("matches: len > 64 %.2f %%, overlapping (dist < len) %.2f %%; dist <= ring-258 for ring 1K %.1f %%, 2K %.1f %%, 4K %.1f %%, 8K %.1f %%\n",
         100.0 * len_gt64 / (nmatch ? nmatch : 1), 100.0 * overlap / (nmatch ? nmatch : 1), 100.0 * dist_le[0] / (nmatch ? nmatch : 1),
         100.0 * dist_le[1] / (nmatch ? nmatch : 1), 100.0 * dist_le[2] / (nmatch ? nmatch : 1), 100.0 * dist_le[3] / (nmatch ? nmatch : 1));
  printf("wide rounds (64 bit offsets): %.0f / member, %.2f symbols, %.2f matches, %.1f out bytes per round; rounds with a match that reads its own round's output %.2f %%, rounds with > 64 out bytes %.2f %%\n",
         (double)rounds / members, (double)round_syms / rounds, (double)round_matches / rounds, (double)round_out / rounds, 100.0 * round_dep / rounds,
         100.0 * round_out_gt64 / rounds);
  printf("rounds holding a match with a long LENGTH code %.2f %%, a long distance code %.2f %%, an overlapping match %.2f %%, any of the three %.2f %%; long-code literals %.3f per round; long length codes %.2f %% of matches\n",
         100.0 * r_longlen / rounds, 100.0 * r_longdist / rounds, 100.0 * r_overlap / rounds, 100.0 * r_any_slow / rounds, (double)r_longlit / rounds, 100.0 * long_len / (nmatch ? nmatch : 1));
  printf("symbols per round:");
  for (int k = 0; k < 20; ++k) printf(" %d:%.1f%%", k, 100.0 * round_hist[k] / rounds);
  printf("\nmatches per round:");
  for (int k = 0; k < 12; ++k) printf(" %d:%.1f%%", k, 100.0 * rm_hist[k] / rounds);
  printf("\nliterals in front of a match:");
  for (int k = 0; k < 10; ++k) printf(" %d:%.1f%%", k, 100.0 * lit_run_hist[k] / (nmatch ? nmatch : 1));
  printf("\nliteral/length code lengths:");
  for (int k = 1; k < 16; ++k) printf(" %d:%.2f%%", k, 100.0 * cl_hist[k] / nsym);
  printf("\n");
  return 0;
}
