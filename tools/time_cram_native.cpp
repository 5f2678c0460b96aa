// Native drain of the host CRAM reader (no Python, no GPU): records/s by containers in flight and batch size.
//   g++ -std=c++17 -O2 -Iexon_amd/csrc -Iinclude tools/time_cram_native.cpp -o /tmp/time_cram_native -lz -lpthread -ldl
//   /tmp/time_cram_native FILE [threads=0 (default)] [batch_rows=8192] [reps=3]
#include "host/cram.h"

#include <chrono>
#include <cstdio>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const int threads = argc > 2 ? atoi(argv[2]) : 0, batch = argc > 3 ? atoi(argv[3]) : 8192, reps = argc > 4 ? atoi(argv[4]) : 3;
  for (int r = 0; r < reps; ++r) {
    const auto t0 = std::chrono::steady_clock::now();
    exon::BAMConfig c;
    c.threads = threads;
    c.batch_size = batch;
    exon::CRAMBatchReader rd(argv[1], c);
    long rows = 0, batches = 0;
    struct ArrowArray a;
    for (;;) {
      memset(&a, 0, sizeof a);
      if (!rd.read_batch(&a)) break;
      rows += a.length;
      ++batches;
      if (a.release) a.release(&a);
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads %d batch %d: %ld records in %ld batches, %.1f ms = %.1f M records/s\n", threads, batch, rows, batches, dt * 1e3, rows / dt / 1e6);
  }
  return 0;
}
