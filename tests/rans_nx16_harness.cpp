// Stream-level driver of the product's rANS Nx16 decoder (exon_amd/csrc/host/cram.h rans_nx16) for tests/test_cram_rans_nx16.py:
// the same header the library compiles, no GPU.  stdin: records of { u32 expected size, u32 stream size, stream bytes };
// stdout: per record { i32 status (0 ok, 1 error), u32 size, bytes (decoded data, or the error text) }.
//   g++ -std=c++17 -O2 -Iexon_amd/csrc -Iinclude tests/rans_nx16_harness.cpp -o <out> -lz -ldl -lpthread
#include "host/cram.h"

#include <cstdio>

int main() {
  std::vector<uint8_t> in;
  for (;;) {
    uint32_t head[2];
    if (fread(head, 4, 2, stdin) != 2) break;
    in.resize(head[1]);
    if (head[1] && fread(in.data(), 1, head[1], stdin) != head[1]) return 2;
    int32_t status = 0;
    std::vector<uint8_t> out;
    try {
      out = exon::cram::rans_nx16(in.data(), in.size(), head[0]);
      if (out.size() != head[0]) throw std::runtime_error("CRAM: block size mismatch");
    } catch (const std::exception& e) {
      status = 1;
      const std::string w = e.what();
      out.assign(w.begin(), w.end());
    }
    const uint32_t n = (uint32_t)out.size();
    fwrite(&status, 4, 1, stdout);
    fwrite(&n, 4, 1, stdout);
    if (n) fwrite(out.data(), 1, n, stdout);
  }
  return 0;
}
