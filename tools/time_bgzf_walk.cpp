// time_bgzf_walk.cpp -- cost of the BGZF header walk (exon_hip_bgzf_scan) over a file held in memory, piece by piece the way the
// pipeline's reader thread takes it (8 MB pieces filled by other threads, walked cold).  Developer harness, CPU only.
//   g++ -O2 -pthread -Iinclude -o tools/bin/time_bgzf_walk tools/time_bgzf_walk.cpp -Lexon_amd/lib -lexon_hip -Wl,-rpath,$PWD/exon_amd/lib
//   EXON_HIP_BGZF_WALK_WARM=0 tools/bin/time_bgzf_walk file.gz      (the plain dependent walk)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "exon_hip.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  fseek(f, 0, SEEK_END);
  const size_t size = (size_t)ftell(f);
  fclose(f);
  std::vector<uint8_t> buf(size + 64);
  {  // eight threads fill the buffer (as the read pool does): the walker's core has none of it in its caches
    std::vector<std::thread> th;
    const size_t per = (size + 7) / 8;
    for (int t = 0; t < 8; ++t)
      th.emplace_back([&, t] {
        FILE* g = fopen(argv[1], "rb");
        const size_t lo = std::min(size, per * t), hi = std::min(size, lo + per);
        fseek(g, (long)lo, SEEK_SET);
        if (fread(buf.data() + lo, 1, hi - lo, g) != hi - lo) abort();
        fclose(g);
      });
    for (auto& t : th) t.join();
  }
  const size_t PIECE = 8u << 20;
  std::vector<exon_hip_bgzf_block> blocks(1 << 20);
  for (int rep = 0; rep < 3; ++rep) {
    {  // push the file out of the caches
      std::vector<uint8_t> junk(512u << 20, 1);
      volatile uint64_t sink = 0;
      for (size_t i = 0; i < junk.size(); i += 64) sink += junk[i];
    }
    size_t o = 0, out = 0;
    long nb = 0;
    const double t0 = now_s();
    while (o < size) {
      const size_t n = std::min(PIECE + (1u << 16), size - o);
      int32_t got = 0;
      size_t consumed = 0, ob = 0;
      if (exon_hip_bgzf_scan(buf.data() + o, n, 0, blocks.data(), (int32_t)blocks.size(), &got, &consumed, &ob) != EXON_HIP_OK) {
        fprintf(stderr, "%s\n", exon_hip_last_error(nullptr));
        return 2;
      }
      if (consumed == 0) break;
      o += consumed;
      out += ob;
      nb += got;
    }
    const double t = now_s() - t0;
    printf("%ld blocks, %zu -> %zu bytes: walk %.2f ms, %.0f ns per block\n", nb, o, out, t * 1e3, t * 1e9 / (double)nb);
  }
  return 0;
}
