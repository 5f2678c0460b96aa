// read_rate.cpp -- how fast a pool of threads moves a page-cached file into pinned host memory (pread, slices of S bytes taken off a
// shared counter: no per-piece barrier), alone and with the H2D copy of the same bytes running beside it.  The ceiling the BGZF
// pipelines' reader thread works under.   hipcc -O2 -o /tmp/read_rate tools/read_rate.cpp -lpthread ; /tmp/read_rate FILE
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <cstring>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// copy with non-temporal stores: the destination lines never enter the caches the DMA engine would have to snoop
__attribute__((target("avx2"))) static void copy_nt(uint8_t* dst, const uint8_t* src, size_t n) {
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
    _mm256_stream_si256((__m256i*)(dst + i), a);
    _mm256_stream_si256((__m256i*)(dst + i + 32), b);
    _mm256_stream_si256((__m256i*)(dst + i + 64), c);
    _mm256_stream_si256((__m256i*)(dst + i + 96), d);
  }
  memcpy(dst + i, src + i, n - i);
  _mm_sfence();
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  const int fd = open(argv[1], O_RDONLY);
  if (fd < 0) return 1;
  struct stat st;
  fstat(fd, &st);
  const size_t size = (size_t)st.st_size;
  const size_t RING = 256u << 20;
  uint8_t *h = nullptr, *d = nullptr;
  if (hipHostMalloc((void**)&h, RING) != hipSuccess || hipMalloc((void**)&d, RING) != hipSuccess) return 2;
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int only_mode = argc > 2 ? atoi(argv[2]) : -1;
  // 0 pread; 1 mmap + memcpy, pages faulted in by the copy; 2 mmap + non-temporal stores; 3 = 2 with MADV_POPULATE_READ per slice first;
  // 4 = 3 and the slice is unmapped (MADV_DONTNEED is not needed: clean file pages) by one munmap at the end on a side thread
  for (int mode = 0; mode < 4; ++mode)
    for (int with_h2d = 0; with_h2d < 2; ++with_h2d)
      for (int T : {8, 12}) {
        if (only_mode >= 0 && mode != only_mode) continue;
        const size_t slice = 1u << 20;
        double best = 1e9, best_map = 0, best_unmap = 0;
        for (int rep = 0; rep < 3; ++rep) {
          const uint8_t* map = nullptr;
          const double tm0 = now_s();
          if (mode > 0) {
            map = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) return 3;
          }
          const double t_map = now_s() - tm0;
          std::atomic<size_t> next{0};
          std::atomic<bool> stop{false};
          std::thread copier;
          if (with_h2d)
            copier = std::thread([&] {
              hipSetDevice(0);
              while (!stop.load()) {
                for (size_t o = 0; o < RING; o += 8u << 20) hipMemcpyAsync(d + o, h + o, 8u << 20, hipMemcpyHostToDevice, s);
                hipStreamSynchronize(s);
              }
            });
          const double t0 = now_s();
          std::vector<std::thread> th;
          for (int t = 0; t < T; ++t)
            th.emplace_back([&] {
              for (;;) {
                const size_t o = next.fetch_add(slice);
                if (o >= size) break;
                const size_t n = std::min(slice, size - o);
                if (mode == 1) { memcpy(h + (o % RING), map + o, n); continue; }
                if (mode == 3) madvise((void*)(map + o), n, 22 /* MADV_POPULATE_READ */);
                if (mode >= 2) { copy_nt(h + (o % RING), map + o, n); continue; }
                size_t have = 0;
                while (have < n) {
                  const ssize_t g = pread(fd, h + (o % RING) + have, n - have, (off_t)(o + have));
                  if (g <= 0) break;
                  have += (size_t)g;
                }
              }
            });
          for (auto& t : th) t.join();
          const double t = now_s() - t0;
          stop = true;
          if (copier.joinable()) copier.join();
          const double tu0 = now_s();
          if (map) munmap((void*)map, size);
          const double t_unmap = now_s() - tu0;
          if (t < best) best = t, best_map = t_map, best_unmap = t_unmap;
        }
        printf("%-28s %s %2d threads: copy %.1f ms = %.1f GB/s (mmap %.2f ms, munmap %.2f ms)\n",
               mode == 0 ? "pread" : mode == 1 ? "mmap, memcpy" : mode == 2 ? "mmap, nt stores" : "mmap, populate + nt stores",
               with_h2d ? "with H2D beside it," : "alone,", T, best * 1e3, size / best / 1e9, best_map * 1e3, best_unmap * 1e3);
        fflush(stdout);
      }
  return 0;
}
