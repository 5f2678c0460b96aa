// parallel.h -- multi-threaded text decode: one reader thread cuts the (decompressed) byte stream into
// record-aligned slabs, N workers parse slabs into column vectors, the consumer re-keys slab-local dictionary
// ids and emits batches in file order.
//
// The reference decodes one record at a time on one tokio task per partition
// (exon-vcf/src/async_batch_stream.rs:59-109, exon-fastq/src/batch_reader.rs:63-82); at ~4 Mrows/s per core that
// is five orders of magnitude below what the GPU consumes, so the decoders here use every host core of the
// partition's process.  Results are identical to the sequential readers (tests/test_scan_decoders.py).
#pragma once
#include <sched.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "arrow_build.h"
#include "bgzf_block.h"
#include "io.h"

namespace exon {

struct TextSlab {
  uint64_t id = 0;
  std::unique_ptr<char[]> buf;  // whole records, '\n'-terminated except possibly the last (uninitialised storage:
  size_t len = 0;               // no zero-fill pass over every slab)
  const char* data() const { return buf.get(); }
  std::exception_ptr error;
  virtual ~TextSlab() = default;
};

// Parses slabs on worker threads.  `Result` derives from TextSlab; `parse(Result&)` fills it from `text`.
template <typename Result>
class SlabPipeline {
 public:
  using ParseFn = void (*)(Result&, const void* ctx);

  SlabPipeline(std::unique_ptr<ByteSource> src, std::string carry, int lines_per_record, int threads, ParseFn parse,
               const void* ctx, size_t slab_bytes = 4u << 20)
      : src_(std::move(src)), carry_(std::move(carry)), lpr_(lines_per_record), parse_(parse), ctx_(ctx),
        slab_bytes_(slab_bytes), max_inflight_((size_t)threads * 2 + 2) {
    reader_ = std::thread([this] { read_loop(); });
    for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work_loop(); });
  }
  ~SlabPipeline() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_todo_.notify_all();
    cv_done_.notify_all();
    cv_space_.notify_all();
    if (reader_.joinable()) reader_.join();
    for (auto& w : workers_) w.join();
  }

  // next parsed slab in file order; nullptr at end of input.  Rethrows reader / parser errors.
  std::unique_ptr<Result> next() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      auto it = done_.find(next_id_);
      if (it != done_.end()) {
        std::unique_ptr<Result> r = std::move(it->second);
        done_.erase(it);
        ++next_id_;
        --inflight_;
        cv_space_.notify_one();
        lk.unlock();
        if (r->error) std::rethrow_exception(r->error);
        return r;
      }
      if (read_error_) std::rethrow_exception(read_error_);
      if (reader_finished_ && next_id_ >= produced_) return nullptr;
      cv_done_.wait(lk);
    }
  }

 private:
  void read_loop() {
    try {
      std::string carry = std::move(carry_);  // bytes after the last whole record of the previous slab
      uint64_t lines_before = 0;              // complete lines handed out so far (for multi-line records)
      bool eof = false;
      while (!eof) {
        // read straight into the slab's own buffer: [carry | up to slab_bytes_ fresh bytes]
        std::unique_ptr<Result> slab(new Result());
        size_t cap = carry.size() + slab_bytes_;
        slab->buf.reset(new char[cap]);
        memcpy(slab->buf.get(), carry.data(), carry.size());
        size_t have = carry.size();
        carry.clear();
        size_t cut = 0;
        for (;;) {
          while (have < cap) {
            const size_t k = src_->read(reinterpret_cast<uint8_t*>(slab->buf.get() + have), cap - have);
            if (k == 0) {
              eof = true;
              break;
            }
            have += k;
          }
          const char* base = slab->buf.get();
          if (eof) {
            cut = have;
          } else if (lpr_ == 1) {
            cut = have;
            while (cut > 0 && base[cut - 1] != '\n') --cut;
          } else {
            uint64_t lines = lines_before;
            const char* p = base;
            const char* end = base + have;
            cut = 0;
            while (p < end) {
              const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
              if (!nl) break;
              ++lines;
              p = nl + 1;
              if (lines % (uint64_t)lpr_ == 0) cut = (size_t)(p - base);
            }
          }
          if (cut > 0 || eof) break;
          // not a single whole record in the buffer: grow it and keep reading
          std::unique_ptr<char[]> bigger(new char[cap * 2]);
          memcpy(bigger.get(), slab->buf.get(), have);
          slab->buf = std::move(bigger);
          cap *= 2;
        }
        carry.assign(slab->buf.get() + cut, have - cut);
        slab->len = cut;
        if (lpr_ > 1) {
          const char* p = slab->buf.get();
          const char* end = p + cut;
          while (p < end) {
            const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            if (!nl) break;
            ++lines_before;
            p = nl + 1;
          }
        }
        if (slab->len == 0) continue;
        std::unique_lock<std::mutex> lk(mu_);
        cv_space_.wait(lk, [this] { return stop_ || inflight_ < max_inflight_; });
        if (stop_) return;
        slab->id = produced_++;
        ++inflight_;
        todo_.push_back(std::move(slab));
        lk.unlock();
        cv_todo_.notify_one();
      }
    } catch (...) {
      std::lock_guard<std::mutex> g(mu_);
      read_error_ = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> g(mu_);
      reader_finished_ = true;
    }
    cv_todo_.notify_all();
    cv_done_.notify_all();
  }

  void work_loop() {
    for (;;) {
      std::unique_ptr<Result> slab;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_todo_.wait(lk, [this] { return stop_ || !todo_.empty() || reader_finished_; });
        if (stop_) return;
        if (todo_.empty()) {
          if (reader_finished_) return;
          continue;
        }
        slab = std::move(todo_.front());
        todo_.pop_front();
      }
      try {
        parse_(*slab, ctx_);
      } catch (...) {
        slab->error = std::current_exception();
      }
      slab->buf.reset();  // the text is no longer needed
      {
        std::lock_guard<std::mutex> g(mu_);
        const uint64_t id = slab->id;
        done_[id] = std::move(slab);
      }
      cv_done_.notify_all();
    }
  }

  std::unique_ptr<ByteSource> src_;
  std::string carry_;
  const int lpr_;
  ParseFn parse_;
  const void* ctx_;
  const size_t slab_bytes_, max_inflight_;
  std::thread reader_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_todo_, cv_done_, cv_space_;
  std::deque<std::unique_ptr<Result>> todo_;
  std::map<uint64_t, std::unique_ptr<Result>> done_;
  uint64_t produced_ = 0, next_id_ = 0;
  size_t inflight_ = 0;
  bool stop_ = false, reader_finished_ = false;
  std::exception_ptr read_error_;
};

// BGZF blocks are independent gzip members of <= 64 KiB: one thread walks the block headers (BSIZE) and hands
// groups of compressed blocks to N inflate workers; read() returns the inflated bytes in file order.
// (the reference inflates on the partition's single task via noodles::bgzf::AsyncReader.)
class BgzfParallelSource : public ByteSource {
 public:
  static bool is_bgzf(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t h[18];
    const bool ok = fread(h, 1, 18, f) == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
    fclose(f);
    return ok;
  }
  BgzfParallelSource(const std::string& path, int threads) : path_(path), max_inflight_((size_t)threads * 3 + 2) {
    f_ = fopen(path.c_str(), "rb");
    if (!f_) throw std::runtime_error("cannot open " + path);
    reader_ = std::thread([this] { read_loop(); });
    for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work_loop(); });
  }
  ~BgzfParallelSource() override {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_todo_.notify_all();
    cv_done_.notify_all();
    cv_space_.notify_all();
    if (reader_.joinable()) reader_.join();
    for (auto& w : workers_) w.join();
    if (f_) fclose(f_);
  }
  size_t read(uint8_t* dst, size_t n) override {
    size_t got = 0;
    while (got < n) {
      if (!cur_ || pos_ == cur_->out.size()) {
        if (!advance()) break;
        continue;
      }
      const size_t k = std::min(n - got, cur_->out.size() - pos_);
      memcpy(dst + got, cur_->out.data() + pos_, k);
      pos_ += k;
      got += k;
    }
    return got;
  }

 private:
  struct Job {
    uint64_t id = 0;
    std::vector<uint8_t> comp;              // whole BGZF blocks, back to back
    std::vector<std::pair<size_t, size_t>> blocks;  // (offset, size) of each block inside comp
    std::vector<uint8_t> out;
    std::exception_ptr error;
  };
  bool advance() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      auto it = done_.find(next_id_);
      if (it != done_.end()) {
        cur_ = std::move(it->second);
        done_.erase(it);
        ++next_id_;
        --inflight_;
        pos_ = 0;
        cv_space_.notify_one();
        lk.unlock();
        if (cur_->error) std::rethrow_exception(cur_->error);
        return true;
      }
      if (read_error_) std::rethrow_exception(read_error_);
      if (reader_finished_ && next_id_ >= produced_) return false;
      cv_done_.wait(lk);
    }
  }
  void read_loop() {
    try {
      bool eof = false;
      while (!eof) {
        std::unique_ptr<Job> job(new Job());
        while (job->comp.size() < (1u << 20)) {
          BgzfBlockInfo info;  // every length checked before use (host/bgzf_block.h)
          if (!read_bgzf_block(f_, &blk_, &info, path_)) {
            eof = true;
            break;
          }
          const size_t off = job->comp.size();
          job->comp.insert(job->comp.end(), blk_.begin(), blk_.end());
          job->blocks.emplace_back(off, info.total);
        }
        if (job->blocks.empty()) continue;
        std::unique_lock<std::mutex> lk(mu_);
        cv_space_.wait(lk, [this] { return stop_ || inflight_ < max_inflight_; });
        if (stop_) return;
        job->id = produced_++;
        ++inflight_;
        todo_.push_back(std::move(job));
        lk.unlock();
        cv_todo_.notify_one();
      }
    } catch (...) {
      std::lock_guard<std::mutex> g(mu_);
      read_error_ = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> g(mu_);
      reader_finished_ = true;
    }
    cv_todo_.notify_all();
    cv_done_.notify_all();
  }
  void work_loop() {
    z_stream z;
    memset(&z, 0, sizeof z);
    const bool zok = inflateInit2(&z, -15) == Z_OK;
    for (;;) {
      std::unique_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_todo_.wait(lk, [this] { return stop_ || !todo_.empty() || reader_finished_; });
        if (stop_) break;
        if (todo_.empty()) {
          if (reader_finished_) break;
          continue;
        }
        job = std::move(todo_.front());
        todo_.pop_front();
      }
      try {
        if (!zok) throw std::runtime_error("inflateInit2 failed");
        size_t total = 0;
        std::vector<BgzfBlockInfo> infos(job->blocks.size());
        for (size_t i = 0; i < job->blocks.size(); ++i) {
          bgzf_block_info(job->comp.data() + job->blocks[i].first, job->blocks[i].second, &infos[i], path_);
          total += infos[i].isize;
        }
        job->out.resize(total);
        size_t o = 0;
        for (size_t i = 0; i < job->blocks.size(); ++i) {  // inflate + CRC-32 check
          inflate_bgzf_block(&z, job->comp.data() + job->blocks[i].first, infos[i], job->out.data() + o, path_);
          o += infos[i].isize;
        }
        std::vector<uint8_t>().swap(job->comp);
      } catch (...) {
        job->error = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        const uint64_t id = job->id;
        done_[id] = std::move(job);
      }
      cv_done_.notify_all();
    }
    if (zok) inflateEnd(&z);
  }

  std::string path_;
  FILE* f_ = nullptr;
  std::vector<uint8_t> blk_;  // reader thread: the block being read
  const size_t max_inflight_;
  std::thread reader_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_todo_, cv_done_, cv_space_;
  std::deque<std::unique_ptr<Job>> todo_;
  std::map<uint64_t, std::unique_ptr<Job>> done_;
  std::unique_ptr<Job> cur_;
  size_t pos_ = 0;
  uint64_t produced_ = 0, next_id_ = 0;
  size_t inflight_ = 0;
  bool stop_ = false, reader_finished_ = false;
  std::exception_ptr read_error_;
};

// CPUs this process can really keep busy: the logical CPUs it may run on (affinity mask), capped by the container's CFS
// quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  The GPU boxes of this pool show 256 logical CPUs
// and a quota of 16: 128 decode threads there ran 12.6 cores' worth of work (throttled every 100 ms period), 16 threads run
// 15.6 (tools/host_scaling.py, profiles/r4_host_scaling.log).
inline int usable_cpus() {
  static const int n = [] {
    int hc = (int)std::thread::hardware_concurrency();
    if (hc < 1) hc = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) hc = std::min(hc, (int)CPU_COUNT(&set));
    double quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64] = {0};
      double period = 0;
      if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / period;
      fclose(f);
    } else {
      double q = -1, period = 0;
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lf", &q) != 1) q = -1;
        fclose(g);
      }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(g, "%lf", &period) != 1) period = 0;
        fclose(g);
      }
      if (q > 0 && period > 0) quota = q / period;
    }
    if (quota > 0) hc = std::min(hc, std::max(1, (int)(quota + 0.999)));
    return hc;
  }();
  return n;
}

inline int decode_threads() {
  if (const char* v = getenv("EXON_HIP_DECODE_THREADS")) {
    const int t = atoi(v);
    if (t >= 1) return t;
  }
  return usable_cpus();  // target_partitions = num_cpus in the reference (exon-core/src/config/mod.rs:44)
}

inline long file_size(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return -1;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fclose(f);
  return n;
}

// the byte source for a file: parallel BGZF inflate for big BGZF inputs, sequential (gzip / plain) otherwise
inline std::unique_ptr<ByteSource> open_source(const std::string& path, Compression c, int threads) {
  if (threads <= 0) threads = decode_threads();
  if (c != Compression::None && threads > 1 && file_size(path) >= (8 << 20) && BgzfParallelSource::is_bgzf(path))
    return std::unique_ptr<ByteSource>(new BgzfParallelSource(path, std::min(threads, 32)));
  return std::unique_ptr<ByteSource>(new ByteReader(path, c));
}

}  // namespace exon
