#include "../exon_amd/csrc/host/decimal_f32.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
int main(int argc, char** argv) {
  std::mt19937_64 rng(12345);
  long bad = 0, total = 0, skipped = 0;
  auto check = [&](const std::string& s) {
    uint32_t b; int ok = exon::dec::parse_f32(s.data(), (int)s.size(), &b);
    if (!ok) { ++skipped; return; }
    float f = strtof(s.c_str(), nullptr); uint32_t w; memcpy(&w, &f, 4);
    ++total;
    if (w != b) { if (bad < 20) printf("MISMATCH %s got %08x want %08x\n", s.c_str(), b, w); ++bad; }
  };
  const char* fixed[] = {"0","1","0.1","0.01","16777217","16777216","3.4028235e38","3.4028236e38","1e39","1.17549435e-38","1e-45","1.4e-45","7e-46","1e-46","123456.7",
     "0.30000001192092896","8.5e-7","1.00000017881393421514957253748434595763683319091796875","9007199254740993","4.7019774032891500318749461488889827112746622270883500860350068251e-38","1e-64","1e-65","1e-66","99999999999999999999"};
  for (auto s : fixed) check(s);
  for (long it = 0; it < (argc > 1 ? atol(argv[1]) : 30000000); ++it) {
    int nd = 1 + rng() % 19; std::string m;
    for (int i = 0; i < nd; ++i) m += char('0' + rng() % 10);
    int kind = rng() % 5; std::string s;
    if (kind == 0) s = m;
    else if (kind == 1) { int k = rng() % (nd + 1); s = (k ? m.substr(0, k) : "0") + "." + (k < nd ? m.substr(k) : "0"); }
    else if (kind == 2) { int e = (int)(rng() % 100) - 60; s = m.substr(0,1) + "." + (nd > 1 ? m.substr(1) : "0") + "e" + std::to_string(e); }
    else if (kind == 3) { s = "0." + std::string(rng() % 45, '0') + m; }
    else { int e = (int)(rng() % 90) - 50; s = m + "e" + std::to_string(e); }
    check(s);
  }
  // every float32 printed with 9 significant digits must round-trip
  for (uint64_t bits = 1; bits < 0x7F800000ull; bits += 977) { uint32_t b32 = (uint32_t)bits; float f; memcpy(&f, &b32, 4); char buf[64]; snprintf(buf, sizeof buf, "%.9g", (double)f); check(buf); }
  printf("checked %ld, skipped %ld, mismatches %ld\n", total, skipped, bad);
  return bad != 0;
}
