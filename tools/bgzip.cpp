// bgzip.cpp -- developer tool: BGZF-compress a file (65280-byte blocks, zlib level given) with a few threads, so the
// GPU inflate path can be timed on large inputs.  build: g++ -O2 -std=c++17 tools/bgzip.cpp -lz -lpthread -o tools/bin/bgzip
// run: bgzip <in> <out> [level=6]
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static std::vector<uint8_t> block(const uint8_t* p, size_t n, int level) {
  std::vector<uint8_t> out(18 + compressBound(n) + 8);
  z_stream z;
  memset(&z, 0, sizeof z);
  deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  z.next_in = const_cast<uint8_t*>(p);
  z.avail_in = (uInt)n;
  z.next_out = out.data() + 18;
  z.avail_out = (uInt)(out.size() - 18);
  deflate(&z, Z_FINISH);
  const size_t c = z.total_out;
  deflateEnd(&z);
  const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
  memcpy(out.data(), head, 16);
  const uint16_t bs = (uint16_t)(18 + c + 8 - 1);
  memcpy(out.data() + 16, &bs, 2);
  const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), p, (uInt)n), isz = (uint32_t)n;
  memcpy(out.data() + 18 + c, &crc, 4);
  memcpy(out.data() + 18 + c + 4, &isz, 4);
  out.resize(18 + c + 8);
  return out;
}
int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: bgzip <in> <out> [level]\n"); return 2; }
  const int level = argc > 3 ? atoi(argv[3]) : 6;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  fseek(f, 0, SEEK_END);
  const size_t n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> in(n);
  if (fread(in.data(), 1, n, f) != n) return 1;
  fclose(f);
  const size_t B = 65280, nb = (n + B - 1) / B;
  std::vector<std::vector<uint8_t>> out(nb);
  const unsigned T = std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t)
    th.emplace_back([&, t] { for (size_t b = t; b < nb; b += T) out[b] = block(in.data() + b * B, std::min(B, n - b * B), level); });
  for (auto& x : th) x.join();
  FILE* g = fopen(argv[2], "wb");
  if (!g) return 1;
  for (auto& o : out) fwrite(o.data(), 1, o.size(), g);
  const auto eof = block(nullptr, 0, level);
  fwrite(eof.data(), 1, eof.size(), g);
  fclose(g);
  return 0;
}
