// stream.cpp -- plan / stream layer of the C ABI: the ExecutionPlan-shaped surface.
//
// Replaces, for one partition, the reference's operator chain
//   AggregateExec(Partial) <- CoalesceBatchesExec <- FilterExec <- <Fmt>Scan batches
// (exon-core/src/datasources/vcf/scanner.rs:142-162 hands 8192-row RecordBatches to DataFusion).
// Host batches arrive through the Arrow C Data Interface, are appended column-wise into a pinned
// staging slot (the CoalesceBatches step: 8192-row batches are far too small for a 256-CU launch),
// copied to HBM asynchronously and consumed by one fused filter+aggregate launch per slot; two slots
// double-buffer so the host keeps appending while the previous slot is in flight.  HBM-resident
// batches (Arrow C Device Data Interface, what the device-layout builders emit) skip staging.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <functional>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "host/arrow_build.h"
#include "host/raw_batch.h"
#include "internal.h"

namespace {

struct ColSpec {
  int elem = 4;        // bytes per value (1 for utf8 payload bytes)
  bool utf8 = false;   // offsets + bytes
};

struct ColStage {
  uint8_t* h_values = nullptr;
  uint8_t* h_valid = nullptr;
  int32_t* h_offsets = nullptr;
  uint8_t* d_values = nullptr;
  uint8_t* d_valid = nullptr;
  int32_t* d_offsets = nullptr;
  bool any_null_bitmap = false;  // at least one appended batch carried a validity bitmap
};

// A small pushed batch (the reference's 8192 rows = ~100 KB) is not copied when it arrives: it is HELD -- the stream owns it,
// its rows have their place in the slot -- and all held batches of a slot are copied together, by the whole pool, when the
// slot is flushed; then they are released.  One thread copies 32 KB pieces at ~10 GB/s; the pool takes them at the rate of one
// large batch (bench.py extras.h2d_inclusive: 8192-row batches 29 -> the 4 Mi-row rate).
struct CopySpan {
  uint8_t* dst;
  const uint8_t* src;
  size_t n;
};
struct HeldBatch {
  struct ArrowArray array;           // moved in (the caller's copy is marked released); released after the copy
  const struct ArrowArray* ch[4];
  int64_t off[4];
  int64_t rows, row0;                // its rows and where they go in the slot
};
struct Slot {
  std::vector<ColStage> cols;
  int64_t rows = 0;
  int64_t bytes = 0;  // utf8 payload bytes appended
  hipEvent_t done = nullptr;
  bool in_flight = false;
  std::vector<HeldBatch> held;
  // held batches are handed to the copy pool in runs of a few MB WHILE further batches arrive (the pushing thread only keeps
  // books); the slot's flush waits for the last run
  size_t submitted = 0;                                        // held[0 .. submitted) are with the pool
  int64_t unsubmitted_bytes = 0;
  std::vector<std::unique_ptr<std::vector<CopySpan>>> span_runs;  // the pool reads these until `pending` is 0
  std::atomic<size_t> pending{0};
};

}  // namespace

struct exon_hip_plan {
  exon_hip_ctx* ctx;
  exon_hip_plan_desc d;
  int64_t n_i64, n_f64;
  int n_cols;
  ColSpec cols[4];
};

struct exon_hip_stream {
  exon_hip_plan* plan;
  exon_hip_ctx* ctx;
  hipStream_t stream = nullptr;
  uint8_t* d_state = nullptr;
  Slot slots[2];
  int cur = 0;
  int64_t cap_rows = 0, cap_bytes = 0;
  bool closed = false;
  int64_t rows_pushed = 0;
  double t_copy = 0, t_wait = 0, t_enqueue = 0;  // EXON_HIP_STAGE_TRACE: staging copies / waits for a free slot / H2D + launch calls
  bool overwrite_next = false;  // exon_hip_stream_reset: the next launch DEFINES the state (no zeroing kernel)
  int x_type = -1, y_type = -1; // K4 fed by a scan: the INFO fields' types from the file's header (-1: the plan's)
  // K4 fed by a scan whose group key is nullable (a String INFO key): the id the scan's dictionary gives the NULL group, asked for
  // when the first NULL key is staged (exon_hip_stream_set_null_group)
  std::function<int32_t()> null_group;
  uint8_t* d_gather = nullptr;  // [world][state words] receive buffer of the all-gather merge
  size_t gather_bytes = 0;
  // ---- group keys by VALUE (ABI 4).  A plan that groups by a dictionary-encoded key (K3 reference, K4 filter) indexes its
  // state by dictionary id; ids are per file (FILTER lists are numbered in order of first appearance, @SQ order differs between
  // BAMs), so a stream that consumes scans remembers which key VALUE each state index stands for, re-keys every further scan
  // into that order, and refuses a cross-rank merge until the ranks have agreed on one dictionary (AggregateExec(Final) merges
  // by key value: SURVEY section 8e "dictionaries identical across shards, else union").
  std::vector<std::string> keys;  // keys[g] = value of state index g
  int keys_state = 0;             // KEYS_NONE: ids are the caller's business; KEYS_LOCAL: from scans, this rank only; KEYS_AGREED
  uint8_t* d_rekey_state = nullptr;  // scratch state of the scan being consumed / of a reorder
  int32_t* d_map = nullptr;
  size_t map_cap = 0;
  uint8_t* saved_state = nullptr;  // the real state while a scan writes into d_rekey_state
  bool saved_overwrite = false;
  // region plans (K2 / K6 / K7) over files: the contig is resolved by NAME in every scan's own dictionary
  std::string region_contig;
  bool has_region_contig = false;
  int32_t region_id_override = INT32_MIN;
};
enum { KEYS_NONE = 0, KEYS_LOCAL = 1, KEYS_AGREED = 2 };
constexpr size_t COLL_SCRATCH = 8192;  // device words for the collectives' votes (up to 511 ranks x 16 bytes), allocated with the state
constexpr int64_t HOLD_MAX_ROWS = 1 << 17;  // batches up to this many rows are held until their slot is flushed

// keyed layout of a plan's packed state: [planes_i x G int64][tail int64][planes_f x G float64]
static bool key_layout(const exon_hip_plan* p, int* G, int* planes_i, int* tail, int* planes_f) {
  switch (p->d.kind) {
    case EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT:  // count[R] + the NULL-reference group
      *G = p->d.n_groups, *planes_i = 1, *tail = 1, *planes_f = 0;
      return true;
    case EXON_HIP_PLAN_CMP_AVG_BY_GROUP:  // count(y)[G], count(*)[G], sum(y)[G]
      *G = p->d.n_groups, *planes_i = 2, *tail = 0, *planes_f = 1;
      return true;
  }
  return false;
}

// ---- staging copies on several cores -----------------------------------------------------------------------------------
// One thread copies into pinned memory at ~10-12 GB/s, a fifth of what PCIe Gen5 takes; the staging copy of a large batch is
// therefore cut into 1 MiB pieces for a small pool of helper threads (lazily started, shared by all streams of the process;
// EXON_HIP_STAGE_THREADS overrides the count, 1 = the caller copies alone).  Small batches (the reference's 8192 rows =
// ~100 KB) stay on the calling thread: handing them over would cost more than the copy.
namespace {
class CopyPool {
 public:
  static CopyPool& get() {
    static CopyPool p;
    return p;
  }
  typedef CopySpan Span;
  // many small copies, asynchronously: the list (which must stay alive until *left is 0) is cut into runs of ~1 MiB for the
  // helper threads; the caller goes on and later calls help_until(left).  Without helpers the copies are done here.
  void submit(const Span* v, size_t count, std::atomic<size_t>* left) {
    if (threads_.empty()) {
      for (size_t i = 0; i < count; ++i) memcpy(v[i].dst, v[i].src, v[i].n);
      return;
    }
    {
      std::lock_guard<std::mutex> g(mu_);
      size_t i = 0;
      while (i < count) {
        size_t j = i, bytes = 0;
        while (j < count && bytes < (1u << 20)) bytes += v[j++].n;
        left->fetch_add(1, std::memory_order_relaxed);
        jobs_.push_back(Job{nullptr, reinterpret_cast<const uint8_t*>(v + i), j - i, left});  // dst == nullptr: a run of spans
        i = j;
      }
    }
    cv_.notify_all();
  }
  // the caller takes jobs too (anybody's), then waits for the runs the helpers hold
  void help_until(std::atomic<size_t>* left) {
    while (left->load(std::memory_order_acquire) != 0) {
      Job j;
      bool have = false;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (!jobs_.empty()) {
          j = jobs_.front();
          jobs_.pop_front();
          have = true;
        }
      }
      if (have) do_job(j);
      else std::this_thread::yield();
    }
  }
  void copy(void* dst, const void* src, size_t n) {
    constexpr size_t PIECE = 1u << 20;
    if (threads_.empty() || n < 4 * PIECE) {
      memcpy(dst, src, n);
      return;
    }
    std::atomic<size_t> left{0};
    size_t pieces = 0;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t o = 0; o < n; o += PIECE) {
        jobs_.push_back(Job{static_cast<uint8_t*>(dst) + o, static_cast<const uint8_t*>(src) + o, std::min(PIECE, n - o), &left});
        ++pieces;
      }
      left.store(pieces, std::memory_order_relaxed);
    }
    cv_.notify_all();
    // the caller works too, then waits for the pieces the helpers took
    for (;;) {
      Job j;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (jobs_.empty()) break;
        j = jobs_.front();
        jobs_.pop_front();
      }
      do_job(j);
    }
    while (left.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }

 private:
  struct Job {
    uint8_t* dst;        // nullptr: `src` points at a run of `n` Spans
    const uint8_t* src;
    size_t n;
    std::atomic<size_t>* left;
  };
  static void do_job(const Job& j) {
    if (j.dst) {
      memcpy(j.dst, j.src, j.n);
    } else {
      const Span* v = reinterpret_cast<const Span*>(j.src);
      for (size_t i = 0; i < j.n; ++i) memcpy(v[i].dst, v[i].src, v[i].n);
    }
    j.left->fetch_sub(1, std::memory_order_acq_rel);
  }
  CopyPool() {
    int t = 8;
    if (const char* v = getenv("EXON_HIP_STAGE_THREADS")) t = atoi(v);
    const int hc = (int)std::thread::hardware_concurrency();
    if (hc > 0) t = std::min(t, std::max(1, hc / 2));
    // (the pool's threads copy host batches into the pinned staging blocks the DMA engine reads next: they run on the GPU's NUMA
    //  node like the file pipelines' reader, scan.cpp; one process drives one GPU, so the device current at first use decides)
    int dev = 0;
    const void* cpus = hipGetDevice(&dev) == hipSuccess ? exon_hip_gpu_local_cpus(dev) : nullptr;
    for (int i = 1; i < t; ++i)
      threads_.emplace_back([this, cpus] {
        exon_hip_run_on(cpus);
        run();
      });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& th : threads_) th.join();
  }
  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
        if (stop_) return;
        j = jobs_.front();
        jobs_.pop_front();
      }
      do_job(j);
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Job> jobs_;
  std::vector<std::thread> threads_;
  bool stop_ = false;
};
}  // namespace

// ---- bit utilities (Arrow LSB-first bitmaps) -------------------------------------------------------
static void append_bits(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t src_off, int64_t n) {
  if (n <= 0) return;
  if (src == nullptr) {  // all valid
    int64_t i = 0;
    while (i < n && ((dst_off + i) & 7)) {
      dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
      ++i;
    }
    const int64_t full = (n - i) / 8;
    if (full > 0) memset(dst + ((dst_off + i) >> 3), 0xFF, (size_t)full);
    i += full * 8;
    for (; i < n; ++i) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
    return;
  }
  if (((dst_off | src_off) & 7) == 0) {
    const int64_t full = n / 8;
    memcpy(dst + (dst_off >> 3), src + (src_off >> 3), (size_t)full);
    for (int64_t i = full * 8; i < n; ++i)
      if ((src[(src_off + i) >> 3] >> ((src_off + i) & 7)) & 1) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
    return;
  }
  for (int64_t i = 0; i < n; ++i)
    if ((src[(src_off + i) >> 3] >> ((src_off + i) & 7)) & 1) dst[(dst_off + i) >> 3] |= (uint8_t)(1u << ((dst_off + i) & 7));
}

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void free_slot(Slot& s) {
  CopyPool::get().help_until(&s.pending);
  for (auto& h : s.held)
    if (h.array.release) h.array.release(&h.array);
  s.held.clear();
  s.span_runs.clear();
  s.submitted = 0;
  s.unsubmitted_bytes = 0;
  for (auto& c : s.cols) {
    if (c.h_values) hipHostFree(c.h_values);
    if (c.h_valid) hipHostFree(c.h_valid);
    if (c.h_offsets) hipHostFree(c.h_offsets);
    if (c.d_values) hipFree(c.d_values);
    if (c.d_valid) hipFree(c.d_valid);
    if (c.d_offsets) hipFree(c.d_offsets);
  }
  s.cols.clear();
  if (s.done) hipEventDestroy(s.done);
  s.done = nullptr;
}

static int alloc_slot(exon_hip_stream* st, Slot& s) {
  exon_hip_plan* p = st->plan;
  s.cols.assign((size_t)p->n_cols, ColStage());
  for (int c = 0; c < p->n_cols; ++c) {
    ColStage& cs = s.cols[(size_t)c];
    const size_t vbytes = p->cols[c].utf8 ? (size_t)st->cap_bytes : (size_t)st->cap_rows * (size_t)p->cols[c].elem;
    const size_t bbytes = (size_t)(st->cap_rows + 7) / 8 + 64;
    if (hipHostMalloc((void**)&cs.h_values, vbytes + 64) != hipSuccess) return EXON_HIP_ENOMEM;
    if (hipMalloc((void**)&cs.d_values, vbytes + 64) != hipSuccess) return EXON_HIP_ENOMEM;
    if (hipHostMalloc((void**)&cs.h_valid, bbytes) != hipSuccess) return EXON_HIP_ENOMEM;
    if (hipMalloc((void**)&cs.d_valid, bbytes) != hipSuccess) return EXON_HIP_ENOMEM;
    memset(cs.h_valid, 0, bbytes);
    if (p->cols[c].utf8) {
      const size_t ob = ((size_t)st->cap_rows + 1) * 4;
      if (hipHostMalloc((void**)&cs.h_offsets, ob) != hipSuccess) return EXON_HIP_ENOMEM;
      if (hipMalloc((void**)&cs.d_offsets, ob) != hipSuccess) return EXON_HIP_ENOMEM;
      cs.h_offsets[0] = 0;
    }
  }
  if (!s.done && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return EXON_HIP_EDEVICE;
  s.rows = 0;
  s.bytes = 0;
  s.in_flight = false;
  return EXON_HIP_OK;
}

// launch the plan's kernel over device columns (operator argument order) into the packed state [n_i64][n_f64]
// x_type / y_type: EXON_HIP_X_* of K4's compared column / AVG argument, or -1 = what the plan says
static int run_plan(const exon_hip_plan* p, void* stream, const exon_hip_column* cols, int64_t n, int flags, void* d_state, int x_type = -1,
                    int y_type = -1, int32_t region_id = INT32_MIN) {
  exon_hip_plan_desc d = p->d;
  if (region_id != INT32_MIN) d.region_chrom_id = region_id;
  if (x_type < 0) x_type = d.x_type;
  if (y_type < 0) y_type = d.y_type;
  if (x_type == EXON_HIP_X_INT32) flags |= EXON_LAUNCH_X_INT32;
  if (y_type == EXON_HIP_X_INT32) flags |= EXON_LAUNCH_Y_INT32;
  exon_hip_ctx* ctx = p->ctx;
  int64_t* counts = reinterpret_cast<int64_t*>(d_state);
  double* sums = reinterpret_cast<double*>(static_cast<uint8_t*>(d_state) + p->n_i64 * 8);
  switch (d.kind) {
    case EXON_HIP_PLAN_REGION_COUNT:
      return exon_op_region_count(ctx, stream, &cols[0], &cols[1], n, d.region_chrom_id, d.region_start, d.region_end, counts,
                                  flags);
    case EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT:
      return exon_op_flag_mapq_group_count(ctx, stream, &cols[0], &cols[1], &cols[2], n, d.flag_mask, d.flag_value, d.mapq_min,
                                           d.n_groups, counts, flags);
    case EXON_HIP_PLAN_CMP_AVG_BY_GROUP:
      return exon_op_cmp_avg_by_group(ctx, stream, &cols[0], &cols[1], &cols[2], n, d.threshold, d.cmp_op, d.n_groups, counts,
                                      sums, flags);
    case EXON_HIP_PLAN_QUAL_POS_HIST:
      return exon_op_qual_pos_hist(ctx, stream, &cols[0], n, d.lmax, counts, flags);
    case EXON_HIP_PLAN_OVERLAP_COUNT:
    case EXON_HIP_PLAN_WITHIN_COUNT:
      return exon_op_overlap_count(ctx, stream, &cols[0], &cols[1], &cols[2], n, d.region_chrom_id, d.region_start,
                                   d.region_end, counts, flags, d.kind == EXON_HIP_PLAN_WITHIN_COUNT);
  }
  return fail(ctx, EXON_HIP_EINVAL, "unknown plan kind %d", d.kind);
}
static int launch_plan(exon_hip_stream* st, const exon_hip_column* cols, int64_t n) {
  const int flags = st->overwrite_next ? EXON_HIP_LAUNCH_OVERWRITE : EXON_HIP_LAUNCH_ACCUMULATE;
  int rc = run_plan(st->plan, st->stream, cols, n, flags, st->d_state, st->x_type, st->y_type, st->region_id_override);
  if (!rc) st->overwrite_next = false;
  return rc;
}
// a reset that no launch has followed yet: the state is all zeros
static int settle_reset(exon_hip_stream* st) {
  if (!st->overwrite_next) return EXON_HIP_OK;
  HIP_TRY(st->ctx, hipMemsetAsync(st->d_state, 0, (size_t)(st->plan->n_i64 + st->plan->n_f64) * 8, st->stream));
  st->overwrite_next = false;
  return EXON_HIP_OK;
}

static void release_held(Slot& s) {
  CopyPool::get().help_until(&s.pending);  // (nobody may still be reading a batch that is released)
  for (auto& h : s.held)
    if (h.array.release) h.array.release(&h.array);
  s.held.clear();
  s.span_runs.clear();
  s.submitted = 0;
  s.unsubmitted_bytes = 0;
}
// held[submitted ..) go to the copy pool (values) and into the bitmaps (in arrival order, on this thread)
static void submit_held(exon_hip_stream* st, Slot& s) {
  if (s.submitted == s.held.size()) return;
  exon_hip_plan* p = st->plan;
  auto run = std::make_unique<std::vector<CopySpan>>();
  run->reserve((s.held.size() - s.submitted) * (size_t)p->n_cols);
  for (size_t k = s.submitted; k < s.held.size(); ++k) {
    const HeldBatch& h = s.held[k];
    for (int c = 0; c < p->n_cols; ++c) {
      ColStage& cs = s.cols[(size_t)c];
      const int e = p->cols[c].elem;
      run->push_back(CopySpan{cs.h_values + (size_t)h.row0 * e, (const uint8_t*)h.ch[c]->buffers[1] + (size_t)h.off[c] * e, (size_t)h.rows * e});
      const uint8_t* valid = (const uint8_t*)h.ch[c]->buffers[0];
      if (valid && h.ch[c]->null_count != 0) {
        if (!cs.any_null_bitmap) {
          append_bits(cs.h_valid, 0, nullptr, 0, h.row0);  // the rows in front of it had no bitmap: mark them valid
          cs.any_null_bitmap = true;
        }
        append_bits(cs.h_valid, h.row0, valid, h.off[c], h.rows);
      } else if (cs.any_null_bitmap) {
        append_bits(cs.h_valid, h.row0, nullptr, 0, h.rows);
      }
    }
  }
  CopyPool::get().submit(run->data(), run->size(), &s.pending);
  s.span_runs.push_back(std::move(run));
  s.submitted = s.held.size();
  s.unsubmitted_bytes = 0;
}
// every held batch of the slot is in its staging buffers; then they are released
static void materialise_held(exon_hip_stream* st, Slot& s) {
  if (s.held.empty()) return;
  const double t_c0 = now_s();
  submit_held(st, s);
  release_held(s);
  st->t_copy += now_s() - t_c0;
}

static int flush_slot(exon_hip_stream* st) {
  Slot& s = st->slots[st->cur];
  if (s.rows == 0) return EXON_HIP_OK;
  materialise_held(st, s);
  exon_hip_plan* p = st->plan;
  const double t_f0 = now_s();
  exon_hip_column cols[4];
  for (int c = 0; c < p->n_cols; ++c) {
    ColStage& cs = s.cols[(size_t)c];
    const size_t vbytes = p->cols[c].utf8 ? (size_t)s.bytes : (size_t)s.rows * (size_t)p->cols[c].elem;
    bool null_key_rewritten = false;
    if (cs.any_null_bitmap && c == 2 && p->d.kind == EXON_HIP_PLAN_CMP_AVG_BY_GROUP && st->null_group) {
      // a NULL group key becomes the scan's id of the NULL group (in the staging copy, before it goes to HBM): no bitmap
      const int32_t id = st->null_group();
      if (id < 0) return fail(st->ctx, EXON_HIP_ESTATE, "the scan has no dictionary for its nullable group key");
      int32_t* ids = reinterpret_cast<int32_t*>(cs.h_values);
      for (int64_t i = 0; i < s.rows; ++i)
        if (!((cs.h_valid[i >> 3] >> (i & 7)) & 1)) ids[i] = id;
      null_key_rewritten = true;
    }
    if (vbytes) HIP_TRY(st->ctx, hipMemcpyAsync(cs.d_values, cs.h_values, vbytes, hipMemcpyHostToDevice, st->stream));
    cols[c].values = cs.d_values;
    cols[c].validity = nullptr;
    cols[c].offsets = nullptr;
    cols[c].length = s.rows;
    if (cs.any_null_bitmap && !null_key_rewritten) {
      HIP_TRY(st->ctx, hipMemcpyAsync(cs.d_valid, cs.h_valid, (size_t)(s.rows + 7) / 8, hipMemcpyHostToDevice, st->stream));
      cols[c].validity = cs.d_valid;
    }
    if (p->cols[c].utf8) {
      HIP_TRY(st->ctx, hipMemcpyAsync(cs.d_offsets, cs.h_offsets, ((size_t)s.rows + 1) * 4, hipMemcpyHostToDevice, st->stream));
      cols[c].offsets = cs.d_offsets;
    }
  }
  int rc = launch_plan(st, cols, s.rows);
  if (rc) return rc;
  HIP_TRY(st->ctx, hipEventRecord(s.done, st->stream));
  s.in_flight = true;
  const double t_f1 = now_s();
  st->t_enqueue += t_f1 - t_f0;
  // switch to the other slot; wait until its previous contents have been consumed
  st->cur ^= 1;
  Slot& n = st->slots[st->cur];
  if (n.in_flight) {
    HIP_TRY(st->ctx, hipEventSynchronize(n.done));
    n.in_flight = false;
  }
  st->t_wait += now_s() - t_f1;
  n.rows = 0;
  n.bytes = 0;
  for (auto& cs : n.cols) {
    if (cs.any_null_bitmap) memset(cs.h_valid, 0, (size_t)(st->cap_rows + 7) / 8 + 64);
    cs.any_null_bitmap = false;
    if (cs.h_offsets) cs.h_offsets[0] = 0;
  }
  return EXON_HIP_OK;
}

static int ensure_capacity(exon_hip_stream* st, int64_t rows, int64_t bytes) {
  if (st->cap_rows >= rows && st->cap_bytes >= bytes && !st->slots[0].cols.empty()) return EXON_HIP_OK;
  // (re)allocate both slots; pending rows must be flushed and consumed first
  int rc = flush_slot(st);
  if (rc) return rc;
  HIP_TRY(st->ctx, hipStreamSynchronize(st->stream));
  for (auto& s : st->slots) free_slot(s);
  int64_t def_rows = 4 << 20, def_bytes = 256ll << 20;
  if (const char* v = getenv("EXON_HIP_COALESCE_ROWS")) {
    long long x = atoll(v);
    if (x >= 1) def_rows = x;
  }
  st->cap_rows = std::max(st->cap_rows, std::max(rows, def_rows));
  st->cap_bytes = std::max(st->cap_bytes, std::max(bytes, def_bytes));
  for (auto& s : st->slots)
    if ((rc = alloc_slot(st, s))) return fail(st->ctx, rc, "staging allocation failed (%lld rows)", (long long)st->cap_rows);
  st->cur = 0;
  return EXON_HIP_OK;
}

extern "C" {

int exon_hip_plan_create(exon_hip_ctx* ctx, const exon_hip_plan_desc* desc, exon_hip_plan** out) {
  if (!ctx || !desc || !out) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_plan_create: NULL argument");
  *out = nullptr;
  exon_hip_plan* p = new (std::nothrow) exon_hip_plan();
  if (!p) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  p->ctx = ctx;
  p->d = *desc;
  p->n_i64 = p->n_f64 = 0;
  switch (desc->kind) {
    case EXON_HIP_PLAN_REGION_COUNT:
      p->n_cols = 2;
      p->cols[0].elem = 4;
      p->cols[1].elem = 8;
      p->n_i64 = 1;
      if (desc->region_start < 1 || desc->region_end < desc->region_start) {
        delete p;
        return fail(ctx, EXON_HIP_EINVAL, "region interval must satisfy 1 <= start <= end");
      }
      break;
    case EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT:
      p->n_cols = 3;
      p->cols[0].elem = 4;
      p->cols[1].elem = 1;
      p->cols[2].elem = 4;
      if (desc->n_groups < 0 || desc->n_groups >= EXON_HIP_MAX_REFERENCES) {
        delete p;
        return fail(ctx, EXON_HIP_EUNSUPPORTED, "n_groups %d out of range", desc->n_groups);
      }
      p->n_i64 = desc->n_groups + 1;
      break;
    case EXON_HIP_PLAN_CMP_AVG_BY_GROUP:
      p->n_cols = 3;
      p->cols[0].elem = p->cols[1].elem = p->cols[2].elem = 4;
      if (desc->n_groups < 1 || desc->n_groups > EXON_HIP_MAX_GROUPS_GLOBAL) {
        delete p;
        return fail(ctx, EXON_HIP_EUNSUPPORTED, "n_groups %d outside [1, %d]", desc->n_groups, EXON_HIP_MAX_GROUPS_GLOBAL);
      }
      if (desc->cmp_op < EXON_HIP_GT || desc->cmp_op > EXON_HIP_NE) {
        delete p;
        return fail(ctx, EXON_HIP_EINVAL, "bad cmp_op %d", desc->cmp_op);
      }
      p->n_i64 = 2 * desc->n_groups;
      p->n_f64 = desc->n_groups;
      break;
    case EXON_HIP_PLAN_OVERLAP_COUNT:
      p->n_cols = 3;
      p->cols[0].elem = 4;
      p->cols[1].elem = p->cols[2].elem = 8;
      p->n_i64 = 1;
      if (desc->region_start < 1 || desc->region_end < desc->region_start) {
        delete p;
        return fail(ctx, EXON_HIP_EINVAL, "region interval must satisfy 1 <= start <= end");
      }
      break;
    case EXON_HIP_PLAN_WITHIN_COUNT:  // start > region_start AND end < region_end: any pair of bounds is a valid predicate
      p->n_cols = 3;
      p->cols[0].elem = 4;
      p->cols[1].elem = p->cols[2].elem = 8;
      p->n_i64 = 1;
      break;
    case EXON_HIP_PLAN_QUAL_POS_HIST:
      p->n_cols = 1;
      p->cols[0].elem = 1;
      p->cols[0].utf8 = true;
      if (desc->lmax < 1 || desc->lmax > 65536) {
        delete p;
        return fail(ctx, EXON_HIP_EINVAL, "lmax %d out of range", desc->lmax);
      }
      p->n_i64 = (int64_t)desc->lmax * 256;
      break;
    default:
      delete p;
      return fail(ctx, EXON_HIP_EINVAL, "unknown plan kind %d", desc->kind);
  }
  for (int c = 0; c < p->n_cols; ++c)
    if (desc->columns[c] < 0) {
      delete p;
      return fail(ctx, EXON_HIP_EINVAL, "negative column index");
    }
  *out = p;
  return EXON_HIP_OK;
}

int exon_hip_plan_destroy(exon_hip_plan* plan) {
  delete plan;
  return EXON_HIP_OK;
}

int exon_hip_plan_state_size(const exon_hip_plan* plan, int64_t* n_i64, int64_t* n_f64) {
  if (!plan || !n_i64 || !n_f64) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_plan_state_size: NULL argument");
  *n_i64 = plan->n_i64;
  *n_f64 = plan->n_f64;
  return EXON_HIP_OK;
}

int exon_hip_stream_open(exon_hip_plan* plan, int32_t partition, exon_hip_stream** out) {
  (void)partition;
  if (!plan || !out) return fail(plan ? plan->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_open: NULL argument");
  *out = nullptr;
  exon_hip_ctx* ctx = plan->ctx;
  exon_hip_stream* st = new (std::nothrow) exon_hip_stream();
  if (!st) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  st->plan = plan;
  st->ctx = ctx;
  hipSetDevice(ctx->device);
  hipError_t e;
  const size_t sbytes = (size_t)(plan->n_i64 + plan->n_f64) * 8;
  // The partition's stream carries the parse + filter/aggregate kernels of the file pipelines: many short kernels that
  // must not queue behind the thousands of workgroups of the NEXT slab's inflate (its own, lower-priority stream in
  // scan.cpp) -- rocprofv3 showed 16-byte fills and single-workgroup scans "taking" milliseconds there.
  // EXON_HIP_STREAM_PRIORITY=0: plain streams (A/B runs).
  int prio_least = 0, prio_greatest = 0;
  const char* pv = getenv("EXON_HIP_STREAM_PRIORITY");
  const bool use_prio = !(pv && pv[0] == '0') && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_least != prio_greatest;
  if ((e = use_prio ? hipStreamCreateWithPriority(&st->stream, hipStreamNonBlocking, prio_greatest)
                    : hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking)) != hipSuccess ||
      // (+ COLL_SCRATCH bytes behind the state: the votes of the collectives -- a rank that is out of memory can still say so)
      (e = hipMalloc((void**)&st->d_state, ((sbytes + 63) & ~(size_t)63) + COLL_SCRATCH)) != hipSuccess ||
      (e = hipMemsetAsync(st->d_state, 0, sbytes, st->stream)) != hipSuccess) {
    if (st->stream) hipStreamDestroy(st->stream);
    if (st->d_state) hipFree(st->d_state);
    delete st;
    return fail(ctx, EXON_HIP_EDEVICE, "stream open: %s", hipGetErrorString(e));
  }
  *out = st;
  return EXON_HIP_OK;
}

static int column_child(exon_hip_stream* st, const struct ArrowArray* batch, int c, const struct ArrowArray** child,
                        int64_t* eff_off) {
  const int idx = st->plan->d.columns[c];
  if (idx >= batch->n_children) return fail(st->ctx, EXON_HIP_EINVAL, "batch has %lld columns, plan needs index %d", (long long)batch->n_children, idx);
  const struct ArrowArray* ch = batch->children[idx];
  if (!ch) return fail(st->ctx, EXON_HIP_EINVAL, "column %d is NULL", idx);
  if (ch->n_children != 0)  // a List<item> column (a list-valued INFO field): no fused kernel takes one as an operand
    return fail(st->ctx, EXON_HIP_EUNSUPPORTED, "column %d is a nested (list) column; the plan's operands are scalar columns", idx);
  const int need = st->plan->cols[c].utf8 ? 3 : 2;
  if (ch->n_buffers < need) return fail(st->ctx, EXON_HIP_EINVAL, "column %d has %lld buffers, expected %d", idx, (long long)ch->n_buffers, need);
  *eff_off = batch->offset + ch->offset;
  if (ch->length + ch->offset < batch->offset + batch->length)
    return fail(st->ctx, EXON_HIP_EINVAL, "column %d shorter than the batch", idx);
  *child = ch;
  return EXON_HIP_OK;
}

int exon_hip_stream_push(exon_hip_stream* st, struct ArrowArray* batch) {
  if (!st || !batch) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_push: NULL argument");
  if (st->closed) return fail(st->ctx, EXON_HIP_ESTATE, "push after finish/close");
  if (!batch->release) return fail(st->ctx, EXON_HIP_EINVAL, "batch already released");
  exon_hip_plan* p = st->plan;
  const int64_t rows = batch->length;
  int rc = EXON_HIP_OK;
  const struct ArrowArray* ch[4];
  int64_t off[4];
  int64_t need_bytes = 0;
  for (int c = 0; c < p->n_cols && !rc; ++c) {
    rc = column_child(st, batch, c, &ch[c], &off[c]);
    if (!rc && p->cols[c].utf8 && rows > 0) {
      const int32_t* o = (const int32_t*)ch[c]->buffers[1];
      need_bytes = std::max<int64_t>(need_bytes, (int64_t)o[off[c] + rows] - o[off[c]]);
    }
  }
  if (!rc && rows > 0) {
    rc = ensure_capacity(st, rows, need_bytes);
    if (!rc && (st->slots[st->cur].rows + rows > st->cap_rows || st->slots[st->cur].bytes + need_bytes > st->cap_bytes))
      rc = flush_slot(st);
    bool fixed = true;
    for (int c = 0; c < p->n_cols; ++c) fixed = fixed && !p->cols[c].utf8;
    static const bool hold_on = [] {
      const char* v = getenv("EXON_HIP_HOLD_SMALL_BATCHES");
      return !(v && v[0] == '0');
    }();
    if (!rc && fixed && hold_on && rows <= HOLD_MAX_ROWS) {  // a small batch: held, copied with the slot's others when the slot is flushed
      Slot& s = st->slots[st->cur];
      HeldBatch h;
      h.array = *batch;
      batch->release = nullptr;  // moved: the stream releases it after the copy
      for (int c = 0; c < p->n_cols; ++c) {
        // (the children pointers stay valid: they belong to the moved array's private data, not to the caller's struct)
        h.ch[c] = ch[c];
        h.off[c] = off[c];
      }
      h.rows = rows;
      h.row0 = s.rows;
      s.held.push_back(h);
      s.rows += rows;
      st->rows_pushed += rows;
      for (int c = 0; c < p->n_cols; ++c) s.unsubmitted_bytes += rows * p->cols[c].elem;
      if (s.unsubmitted_bytes >= (4 << 20)) submit_held(st, s);  // the pool copies them while the next batches arrive
      return EXON_HIP_OK;
    }
    if (!rc) {
      Slot& s = st->slots[st->cur];
      materialise_held(st, s);  // (keeps the bitmap bookkeeping in arrival order)
      const double t_c0 = now_s();
      for (int c = 0; c < p->n_cols; ++c) {
        ColStage& cs = s.cols[(size_t)c];
        const uint8_t* valid = (const uint8_t*)ch[c]->buffers[0];
        if (valid && ch[c]->null_count != 0) {
          if (!cs.any_null_bitmap) {
            // rows appended so far had no bitmap: mark them valid
            append_bits(cs.h_valid, 0, nullptr, 0, s.rows);
            cs.any_null_bitmap = true;
          }
          append_bits(cs.h_valid, s.rows, valid, off[c], rows);
        } else if (cs.any_null_bitmap) {
          append_bits(cs.h_valid, s.rows, nullptr, 0, rows);
        }
        if (p->cols[c].utf8) {
          const int32_t* o = (const int32_t*)ch[c]->buffers[1] + off[c];
          const uint8_t* data = (const uint8_t*)ch[c]->buffers[2];
          const int32_t base = o[0], len = o[rows] - base;
          if (len) CopyPool::get().copy(cs.h_values + s.bytes, data + base, (size_t)len);
          const int32_t shift = (int32_t)s.bytes - base;
          for (int64_t i = 1; i <= rows; ++i) cs.h_offsets[s.rows + i] = o[i] + shift;
        } else {
          const int e = p->cols[c].elem;
          CopyPool::get().copy(cs.h_values + (size_t)s.rows * e, (const uint8_t*)ch[c]->buffers[1] + (size_t)off[c] * e, (size_t)rows * e);
        }
      }
      s.rows += rows;
      s.bytes += need_bytes;
      st->rows_pushed += rows;
      st->t_copy += now_s() - t_c0;
    }
  }
  batch->release(batch);  // moved: released exactly once, success or not
  return rc;
}

}  // extern "C"

// Internal (C++) hooks for the GPU decode pipeline (scan.cpp): launch the plan over HBM-resident columns addressed
// in the SCAN's column order, and snapshot / restore the partial state around a speculative GPU-parsed file.
// `row_mask` (optional): validity bitmap that REPLACES the one of the plan's first operand (the pushed-down region filter:
// the caller has already ANDed that operand's own validity into it; every fused kernel drops rows whose first operand is NULL).
int exon_hip_stream_launch_scan_columns(exon_hip_stream* st, const exon_hip_column* scan_cols, int n_scan_cols, int64_t n,
                                        const uint8_t* row_mask) {
  exon_hip_plan* p = st->plan;
  if (st->closed) return fail(st->ctx, EXON_HIP_ESTATE, "push after finish/close");
  int rc = flush_slot(st);  // keep stream order with rows staged earlier
  if (rc) return rc;
  exon_hip_column cols[4];
  for (int c = 0; c < p->n_cols; ++c) {
    const int idx = p->d.columns[c];
    if (idx >= n_scan_cols || !scan_cols[idx].values) return fail(st->ctx, EXON_HIP_EINVAL, "plan needs scan column %d which this scan does not produce", idx);
    cols[c] = scan_cols[idx];
  }
  if (row_mask) cols[0].validity = row_mask;
  rc = launch_plan(st, cols, n);
  if (!rc) st->rows_pushed += n;
  return rc;
}
int exon_hip_stream_plan_first_column(exon_hip_stream* st) { return st->plan->d.columns[0]; }
int exon_hip_stream_plan_kind(exon_hip_stream* st) { return st->plan->d.kind; }
int exon_hip_stream_plan_column(exon_hip_stream* st, int arg) { return arg >= 0 && arg < st->plan->n_cols ? st->plan->d.columns[arg] : -1; }
void exon_hip_stream_set_value_types(exon_hip_stream* st, int x_type, int y_type) {
  st->x_type = x_type;
  st->y_type = y_type;
}
// (taking the provider away flushes what is staged first: those rows' NULL keys still need it)
int exon_hip_stream_set_null_group(exon_hip_stream* st, std::function<int32_t()> id_of_null) {
  int rc = EXON_HIP_OK;
  if (!id_of_null && st->null_group && !st->closed) rc = flush_slot(st);
  st->null_group = std::move(id_of_null);
  return rc;
}
// K5 over views into text resident in HBM (FASTQ slabs): scan column 2 = sequence lines, 3 = quality lines
int exon_hip_stream_launch_views(exon_hip_stream* st, const uint8_t* d_text, const exon_hip_fastq_views& v) {
  exon_hip_plan* p = st->plan;
  if (st->closed) return fail(st->ctx, EXON_HIP_ESTATE, "push after finish/close");
  if (p->d.kind != EXON_HIP_PLAN_QUAL_POS_HIST) return fail(st->ctx, EXON_HIP_EUNSUPPORTED, "GPU-side FASTQ splitting serves the per-position histogram plan only");
  const int idx = p->d.columns[0];
  if (idx != 2 && idx != 3) return fail(st->ctx, EXON_HIP_EINVAL, "plan needs scan column %d; FASTQ views exist for 2 (sequence) and 3 (quality_scores)", idx);
  int rc = flush_slot(st);
  if (!rc) rc = settle_reset(st);
  if (rc) return rc;
  (void)d_text;  // the views index v.text_base (the aligned address at or below the slab)
  rc = exon_hip_qual_pos_hist_views(st->ctx, st->stream, v.text_base, idx == 2 ? v.seq_start : v.qual_start,
                                    idx == 2 ? v.seq_end : v.qual_end, v.n_reads, p->d.lmax,
                                    reinterpret_cast<int64_t*>(st->d_state));
  if (!rc) st->rows_pushed += v.n_reads;
  return rc;
}
void* exon_hip_stream_hip_stream(exon_hip_stream* st) { return (void*)st->stream; }
exon_hip_ctx* exon_hip_stream_ctx(exon_hip_stream* st) { return st->ctx; }
// snapshot (restore = false) / roll back (restore = true) the partial state around a speculative GPU decode.  Rows staged
// by earlier pushes are launched BEFORE the snapshot is taken (they belong to it), and the row counter travels with it.
int exon_hip_stream_state_copy(exon_hip_stream* st, void* d_snapshot, bool restore, int64_t* rows_pushed) {
  if (!restore) {
    int rc = flush_slot(st);
    if (!rc) rc = settle_reset(st);
    if (rc) return rc;
    if (rows_pushed) *rows_pushed = st->rows_pushed;
  } else {
    st->overwrite_next = false;
    if (rows_pushed) st->rows_pushed = *rows_pushed;
  }
  const size_t bytes = (size_t)(st->plan->n_i64 + st->plan->n_f64) * 8;
  HIP_TRY(st->ctx, hipMemcpyAsync(restore ? (void*)st->d_state : d_snapshot, restore ? d_snapshot : (void*)st->d_state, bytes,
                                  hipMemcpyDeviceToDevice, st->stream));
  return EXON_HIP_OK;
}
size_t exon_hip_stream_state_bytes(exon_hip_stream* st) { return (size_t)(st->plan->n_i64 + st->plan->n_f64) * 8; }

// Internal (C++): append decoder output (raw vectors + validity bytes) to the staging slot.  Fixed-width plans only.
int exon_hip_stream_push_raw(exon_hip_stream* st, const exon::RawBatch& rb) {
  exon_hip_plan* p = st->plan;
  if (st->closed) return fail(st->ctx, EXON_HIP_ESTATE, "push after finish/close");
  for (int c = 0; c < p->n_cols; ++c) {
    if (p->cols[c].utf8) return fail(st->ctx, EXON_HIP_EUNSUPPORTED, "raw push of a Utf8 column");
    const int idx = p->d.columns[c];
    if (idx >= (int)rb.cols.size()) return fail(st->ctx, EXON_HIP_EINVAL, "scan has %zu columns, plan needs index %d", rb.cols.size(), idx);
    if (rb.cols[(size_t)idx].elem != p->cols[c].elem) return fail(st->ctx, EXON_HIP_EINVAL, "column %d: element size %d, plan expects %d", idx, rb.cols[(size_t)idx].elem, p->cols[c].elem);
  }
  int64_t done = 0;
  while (done < rb.rows) {
    int rc = ensure_capacity(st, 1, 0);
    if (rc) return rc;
    if (st->slots[st->cur].rows >= st->cap_rows && (rc = flush_slot(st))) return rc;
    Slot& s = st->slots[st->cur];
    materialise_held(st, s);  // (batches held by exon_hip_stream_push land first: the bitmap bookkeeping below is in arrival order)
    const int64_t n = std::min<int64_t>(rb.rows - done, st->cap_rows - s.rows);
    for (int c = 0; c < p->n_cols; ++c) {
      const exon::RawColumn& rc2 = rb.cols[(size_t)p->d.columns[c]];
      ColStage& cs = s.cols[(size_t)c];
      const int e = p->cols[c].elem;
      CopyPool::get().copy(cs.h_values + (size_t)s.rows * e, static_cast<const uint8_t*>(rc2.values) + (size_t)done * e, (size_t)n * e);
      if (rc2.valid_bytes) {
        if (!cs.any_null_bitmap) {
          append_bits(cs.h_valid, 0, nullptr, 0, s.rows);
          cs.any_null_bitmap = true;
        }
        const uint8_t* v = rc2.valid_bytes + done;
        for (int64_t i = 0; i < n; ++i)
          if (v[i]) cs.h_valid[(s.rows + i) >> 3] |= (uint8_t)(1u << ((s.rows + i) & 7));
      } else if (cs.any_null_bitmap) {
        append_bits(cs.h_valid, s.rows, nullptr, 0, n);
      }
    }
    s.rows += n;
    st->rows_pushed += n;
    done += n;
  }
  return EXON_HIP_OK;
}

extern "C" {

int exon_hip_stream_push_device(exon_hip_stream* st, const struct ArrowDeviceArray* dbatch) {
  if (!st || !dbatch) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_push_device: NULL argument");
  if (st->closed) return fail(st->ctx, EXON_HIP_ESTATE, "push after finish/close");
  if (dbatch->device_type != ARROW_DEVICE_ROCM)
    return fail(st->ctx, EXON_HIP_EINVAL, "device_type %d is not ARROW_DEVICE_ROCM", (int)dbatch->device_type);
  if (dbatch->device_id != st->ctx->device)
    return fail(st->ctx, EXON_HIP_EINVAL, "batch lives on device %lld, stream on %d", (long long)dbatch->device_id, st->ctx->device);
  exon_hip_plan* p = st->plan;
  const struct ArrowArray* batch = &dbatch->array;
  const int64_t rows = batch->length;
  exon_hip_column cols[4];
  for (int c = 0; c < p->n_cols; ++c) {
    const struct ArrowArray* ch;
    int64_t off;
    int rc = column_child(st, batch, c, &ch, &off);
    if (rc) return rc;
    cols[c].length = rows;
    cols[c].validity = nullptr;
    cols[c].offsets = nullptr;
    if (ch->buffers[0] && ch->null_count != 0) {
      if (off & 7) return fail(st->ctx, EXON_HIP_EINVAL, "device batch: validity bit offset must be a multiple of 8");
      cols[c].validity = (const uint8_t*)ch->buffers[0] + (off >> 3);
    }
    if (p->cols[c].utf8) {
      cols[c].offsets = (const int32_t*)ch->buffers[1] + off;
      cols[c].values = ch->buffers[2];
    } else {
      cols[c].values = (const uint8_t*)ch->buffers[1] + (size_t)off * p->cols[c].elem;
    }
  }
  if (dbatch->sync_event) HIP_TRY(st->ctx, hipStreamWaitEvent(st->stream, *(hipEvent_t*)dbatch->sync_event, 0));
  int rc = launch_plan(st, cols, rows);
  if (!rc) st->rows_pushed += rows;
  return rc;
}

int exon_hip_stream_state(exon_hip_stream* st, int64_t** d_i64, double** d_f64, void** hip_stream) {
  if (!st) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_stream_state: NULL stream");
  int rc = flush_slot(st);
  if (!rc) rc = settle_reset(st);
  if (rc) return rc;
  if (d_i64) *d_i64 = reinterpret_cast<int64_t*>(st->d_state);
  if (d_f64) *d_f64 = reinterpret_cast<double*>(st->d_state + st->plan->n_i64 * 8);
  if (hip_stream) *hip_stream = (void*)st->stream;
  return EXON_HIP_OK;
}

}  // extern "C"


// ---- group keys by value: internal machinery ------------------------------------------------------------------------------
static size_t state_bytes_of(const exon_hip_stream* st) { return (size_t)(st->plan->n_i64 + st->plan->n_f64) * 8; }

static int ensure_rekey_buffers(exon_hip_stream* st, size_t map_entries) {
  if (!st->d_rekey_state && hipMalloc((void**)&st->d_rekey_state, std::max<size_t>(state_bytes_of(st), 16)) != hipSuccess)
    return fail(st->ctx, EXON_HIP_ENOMEM, "re-keying scratch state of %zu bytes", state_bytes_of(st));
  if (st->map_cap < map_entries) {
    if (st->d_map) hipFree(st->d_map);
    st->d_map = nullptr;
    st->map_cap = 0;
    if (hipMalloc((void**)&st->d_map, std::max<size_t>(map_entries, 64) * 4) != hipSuccess)
      return fail(st->ctx, EXON_HIP_ENOMEM, "re-keying map of %zu entries", map_entries);
    st->map_cap = std::max<size_t>(map_entries, 64);
  }
  return EXON_HIP_OK;
}

// dst += src with src's index g landing on map[g]; synchronises the stream (the map is staged from pageable memory)
static int permute_add(exon_hip_stream* st, const uint8_t* src, uint8_t* dst, const std::vector<int32_t>& map) {
  int G = 0, pi = 0, tail = 0, pf = 0;
  if (!key_layout(st->plan, &G, &pi, &tail, &pf)) return fail(st->ctx, EXON_HIP_EINVAL, "this plan has no group keys");
  int rc = ensure_rekey_buffers(st, map.size());
  if (rc) return rc;
  if (!map.empty()) HIP_TRY(st->ctx, hipMemcpyAsync(st->d_map, map.data(), map.size() * 4, hipMemcpyHostToDevice, st->stream));
  HIP_TRY(st->ctx, exon::launch_permute_add_state(st->stream, src, dst, st->d_map, (int)map.size(), G, pi, tail, pf));
  HIP_TRY(st->ctx, hipStreamSynchronize(st->stream));
  return EXON_HIP_OK;
}

// Internal (scan.cpp): bracket one consumed scan.  begin: when the stream's state is already keyed by earlier scans (or by an
// agreed dictionary), the scan's kernels write into a scratch state under the SCAN's ids; end: the scan's dictionary is
// interned into the stream's, and the scratch state is added into the real one under the stream's ids.
bool exon_hip_stream_is_keyed(exon_hip_stream* st) {
  int G, a, b, c;
  return key_layout(st->plan, &G, &a, &b, &c);
}
int exon_hip_stream_begin_scan(exon_hip_stream* st, bool* tracked, bool* redirected) {
  *tracked = *redirected = false;
  if (!exon_hip_stream_is_keyed(st)) return EXON_HIP_OK;
  int rc = flush_slot(st);
  if (rc) return rc;
  if (st->keys_state == KEYS_NONE) {
    // rows pushed earlier came with the caller's own ids: the stream cannot know their values and keeps out of it
    *tracked = st->rows_pushed == 0;
    return EXON_HIP_OK;
  }
  *tracked = true;
  rc = settle_reset(st);
  if (!rc) rc = ensure_rekey_buffers(st, 0);
  if (rc) return rc;
  st->saved_state = st->d_state;
  st->saved_overwrite = st->overwrite_next;
  st->d_state = st->d_rekey_state;
  st->overwrite_next = true;  // the first launch of the scan defines the scratch state
  *redirected = true;
  return EXON_HIP_OK;
}
int exon_hip_stream_end_scan(exon_hip_stream* st, const std::vector<std::string>* scan_keys, bool tracked, bool redirected, bool ok) {
  int rc = EXON_HIP_OK;
  if (redirected) {
    rc = flush_slot(st);
    if (!rc) rc = settle_reset(st);  // a scan that launched nothing leaves an all-zero scratch state
    st->d_state = st->saved_state;
    st->overwrite_next = st->saved_overwrite;
    st->saved_state = nullptr;
  }
  if (rc || !ok || !tracked) return rc;
  int G = 0, pi, tail, pf;
  key_layout(st->plan, &G, &pi, &tail, &pf);
  if (!scan_keys) {  // the key column of this scan is not dictionary-encoded: nothing to re-key by
    if (redirected) return fail(st->ctx, EXON_HIP_EUNSUPPORTED, "the stream is keyed by value but this scan's group column has no dictionary");
    return EXON_HIP_OK;
  }
  if ((int64_t)scan_keys->size() > G) {
    if (!redirected) {
      // first scan of the stream: its rows went straight into the state (there was nothing to protect and no scratch to pay
      // for) -- under ids beyond n_groups that the kernels dropped or folded.  "Unchanged" means EMPTY here: the stream had no
      // rows and no keys before this scan (tracked && !redirected <=> rows_pushed was 0, KEYS_NONE), so it gets that back.
      const int rc2 = flush_slot(st);
      (void)exon_hip_sync(st->ctx, st->stream);  // the kernels flagged the ids beyond n_groups in the device status word: take it (and clear it) here
      st->overwrite_next = true;  // the next launch defines the state; exon_hip_stream_state / finish zero it (settle_reset)
      st->rows_pushed = 0;
      if (rc2) return rc2;
    }
    return fail(st->ctx, EXON_HIP_ECAPACITY, "the scan's group-key dictionary has %zu entries, the plan was created for n_groups = %d; this scan's rows were not added, the stream's state and dictionary are unchanged", scan_keys->size(), G);
  }
  if (!redirected) {  // first scan of the stream: its ids ARE the stream's
    st->keys = *scan_keys;
    st->keys_state = KEYS_LOCAL;
    return EXON_HIP_OK;
  }
  // The map and the enlarged dictionary are worked out on the side and adopted only when the scan's state has been added in: a
  // scan that does not fit (or whose add fails) leaves the stream's dictionary and state exactly as they were -- the scan's rows
  // are lost (the call fails), but nothing half-done stays behind.
  std::vector<int32_t> map(scan_keys->size());
  std::vector<std::string> grown = st->keys;
  std::unordered_map<std::string, int32_t> index;
  for (size_t i = 0; i < grown.size(); ++i) index.emplace(grown[i], (int32_t)i);  // first occurrence wins
  for (size_t i = 0; i < scan_keys->size(); ++i) {
    auto it = index.find((*scan_keys)[i]);
    if (it == index.end()) {
      if ((int64_t)grown.size() >= G)
        return fail(st->ctx, EXON_HIP_ECAPACITY,
                    "the scans' group keys hold more than the %d distinct values the plan was created for (n_groups); this scan's "
                    "rows were not added, the stream's state and dictionary are unchanged", G);
      it = index.emplace((*scan_keys)[i], (int32_t)grown.size()).first;
      grown.push_back((*scan_keys)[i]);
    }
    map[i] = it->second;
  }
  rc = permute_add(st, st->d_rekey_state, st->d_state, map);
  if (rc) return rc;
  st->keys.swap(grown);
  st->keys_state = KEYS_LOCAL;
  return EXON_HIP_OK;
}
// region plans over files: the stream's contig NAME (exon_hip_stream_set_region_contig) -> this scan's dictionary id
bool exon_hip_stream_region_contig(exon_hip_stream* st, std::string* name) {
  if (!st->has_region_contig) return false;
  *name = st->region_contig;
  return true;
}
// rows staged by the previous file were decoded under ITS header's ids: they are launched before the id changes
int exon_hip_stream_set_region_id(exon_hip_stream* st, int32_t id) {
  if (st->region_id_override != id) {
    const int rc = flush_slot(st);
    if (rc) return rc;
  }
  st->region_id_override = id;
  return EXON_HIP_OK;
}

// ---- RCCL (loaded on first use: hosts that never merge across GPUs -- and machines without RCCL -- do not need it) ----
namespace {
struct Rccl {
  typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
  typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
  typedef int (*count_fn)(void*, int*);
  typedef int (*uid_fn)(void*);
  typedef int (*destroy_fn)(void*);
  void* lib = nullptr;
  allreduce_fn all_reduce = nullptr;
  allgather_fn all_gather = nullptr;
  count_fn comm_count = nullptr, comm_rank = nullptr;
  uid_fn get_unique_id = nullptr;
  void* init_rank = nullptr;  // ncclCommInitRank(ncclComm_t*, int, ncclUniqueId /* 128 bytes BY VALUE */, int)
  destroy_fn comm_destroy = nullptr;
  Rccl() {
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return;
    all_reduce = (allreduce_fn)dlsym(lib, "ncclAllReduce");
    all_gather = (allgather_fn)dlsym(lib, "ncclAllGather");
    comm_count = (count_fn)dlsym(lib, "ncclCommCount");
    comm_rank = (count_fn)dlsym(lib, "ncclCommUserRank");
    get_unique_id = (uid_fn)dlsym(lib, "ncclGetUniqueId");
    init_rank = dlsym(lib, "ncclCommInitRank");
    comm_destroy = (destroy_fn)dlsym(lib, "ncclCommDestroy");
  }
  bool ok() const { return all_reduce && all_gather && comm_count && comm_rank && get_unique_id && init_rank && comm_destroy; }
};
const Rccl& rccl() {
  static Rccl r;
  return r;
}
struct UniqueId {
  char internal[128];  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
};
constexpr int NCCL_INT64 = 4, NCCL_SUM = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t
constexpr size_t GATHER_MERGE_MAX_STATE = 1 << 20;  // larger states (K3 with millions of references) are all-reduced
}  // namespace

extern "C" {

// ---- communicators ----------------------------------------------------------------------------------------------------------
// The handle the collectives take (exon_hip_rccl_comm_init, exon_hip_comm_from_callbacks) wraps one of two transports:
//   RCCL       ncclAllGather / ncclAllReduce on the stream's hipStream_t (the product path: one process per GPU over xGMI);
//   callbacks  an all-gather over HOST buffers supplied by the caller (torch.distributed / gloo, MPI, a Rust host's own runtime):
//              the same vote / reconcile / merge logic runs on machines where RCCL cannot form the communicator -- e.g. several
//              ranks on ONE GPU, which is how tests/test_collective_faults.py drives 2 and 8 ranks through every failure path.
// Every collective entry point VOTES first: each rank contributes one word (0, or why it cannot go on), all ranks see all words,
// and either all go on or all return the same error.  No rank-local early return sits between two collectives.
struct ExonComm {
  uint32_t magic = 0x45584343u;  // "EXCC"
  int kind = 0;                  // 0 RCCL, 1 callbacks
  void* nccl = nullptr;
  int world = 1, rank = 0;
  exon_hip_allgather_fn fn = nullptr;
  void* user = nullptr;
  bool aborted = false;
  bool owned = true;  // the ncclComm_t was made by exon_hip_rccl_comm_init (destroyed with the handle); false: the caller's own
};
static ExonComm* comm_of(void* h) {
  ExonComm* c = static_cast<ExonComm*>(h);
  return c && c->magic == 0x45584343u ? c : nullptr;
}
// EXON_HIP_FAULT="site@rank[,site@rank...]": the named site fails on that rank (tests of the fail-together paths).
// Sites: reconcile_enter, reconcile_malloc, reconcile_rekey, allreduce_enter, allreduce_malloc, stall (the rank sleeps instead of entering)
static bool fault_at(const char* site, int rank) {
  const char* v = getenv("EXON_HIP_FAULT");
  if (!v || !*v) return false;
  const std::string all(v);
  size_t i = 0;
  while (i < all.size()) {
    size_t j = all.find(',', i);
    if (j == std::string::npos) j = all.size();
    const std::string item = all.substr(i, j - i);
    const size_t at = item.find('@');
    if (at != std::string::npos && item.compare(0, at, site) == 0 && atoi(item.c_str() + at + 1) == rank) return true;
    i = j + 1;
  }
  return false;
}
static double collective_timeout_s() {
  const char* v = getenv("EXON_HIP_COLLECTIVE_TIMEOUT_S");
  const double t = v ? atof(v) : 120.0;
  return t > 0 ? t : 120.0;
}
// hipStreamSynchronize with a bound: a peer that never enters a collective leaves this rank's kernel spinning for ever; after the
// bound the communicator is aborted (ncclCommAbort) and the call fails.  The communicator is unusable afterwards.
static int bounded_sync(exon_hip_ctx* ctx, ExonComm* c, hipStream_t s, const char* what) {
  const double limit = collective_timeout_s();
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return EXON_HIP_OK;
    if (q != hipErrorNotReady) return fail(ctx, EXON_HIP_EDEVICE, "%s: %s", what, hipGetErrorString(q));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt > limit) {
      if (c && c->kind == 0 && c->nccl && !c->aborted) {
        typedef int (*abort_fn)(void*);
        if (abort_fn f = (abort_fn)dlsym(rccl().lib, "ncclCommAbort")) (void)f(c->nccl);
        c->aborted = true;
        c->nccl = nullptr;
      }
      return fail(ctx, EXON_HIP_EDEVICE, "%s: no answer from the other ranks within %.0f s (EXON_HIP_COLLECTIVE_TIMEOUT_S); the communicator was aborted", what, limit);
    }
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
// all-gather of `bytes` (a multiple of 8) per rank between DEVICE buffers, enqueued on `s` (callbacks: staged through the host)
static int comm_all_gather_dev(exon_hip_ctx* ctx, ExonComm* c, hipStream_t s, const void* d_send, void* d_recv, size_t bytes) {
  if (c->aborted) return fail(ctx, EXON_HIP_ESTATE, "the communicator was aborted by an earlier time-out");
  if (c->kind == 0) {
    const int e = rccl().all_gather(d_send, d_recv, bytes / 8, NCCL_INT64, c->nccl, s);
    return e ? fail(ctx, EXON_HIP_EDEVICE, "ncclAllGather failed with ncclResult_t %d", e) : EXON_HIP_OK;
  }
  std::vector<uint8_t> mine(bytes), all(bytes * (size_t)c->world);
  HIP_TRY(ctx, hipMemcpyAsync(mine.data(), d_send, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (c->fn(c->user, mine.data(), all.data(), bytes)) return fail(ctx, EXON_HIP_EDEVICE, "the caller's all-gather failed");
  HIP_TRY(ctx, hipMemcpyAsync(d_recv, all.data(), all.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));  // (pageable staging: `all` must outlive the copy)
  return EXON_HIP_OK;
}
// all-gather of `bytes` (a multiple of 8, world x bytes <= COLL_SCRATCH / 2) per rank between HOST buffers, through `d_scratch`
static int comm_all_gather_host(exon_hip_ctx* ctx, ExonComm* c, hipStream_t s, uint8_t* d_scratch, const void* h_send, void* h_recv, size_t bytes, const char* what) {
  if (c->aborted) return fail(ctx, EXON_HIP_ESTATE, "the communicator was aborted by an earlier time-out");
  if (c->kind == 1) {
    if (c->fn(c->user, h_send, h_recv, bytes)) return fail(ctx, EXON_HIP_EDEVICE, "%s: the caller's all-gather failed", what);
    return EXON_HIP_OK;
  }
  uint8_t* d_all = d_scratch;
  uint8_t* d_mine = d_scratch + COLL_SCRATCH / 2;
  HIP_TRY(ctx, hipMemcpyAsync(d_mine, h_send, bytes, hipMemcpyHostToDevice, s));
  const int e = rccl().all_gather(d_mine, d_all, bytes / 8, NCCL_INT64, c->nccl, s);
  if (e) return fail(ctx, EXON_HIP_EDEVICE, "%s: ncclAllGather failed with ncclResult_t %d", what, e);
  HIP_TRY(ctx, hipMemcpyAsync(h_recv, d_all, bytes * (size_t)c->world, hipMemcpyDeviceToHost, s));
  return bounded_sync(ctx, c, s, what);
}
// The vote: my_code = 0, or an EXON_HIP_E* code with `why`.  Returns 0 when every rank said 0; otherwise every rank returns the
// code of the LOWEST failing rank, with a text that names it.
static int comm_vote(exon_hip_ctx* ctx, ExonComm* c, hipStream_t s, uint8_t* d_scratch, int my_code, const char* why, const char* what) {
  if ((size_t)c->world * 16 > COLL_SCRATCH / 2) return fail(ctx, EXON_HIP_EUNSUPPORTED, "%s: more than %zu ranks", what, COLL_SCRATCH / 32);
  if (fault_at("stall", c->rank)) {  // (test hook) this rank never enters: the others must time out, not hang
    std::this_thread::sleep_for(std::chrono::duration<double>(collective_timeout_s() * 3));
    return fail(ctx, EXON_HIP_EDEVICE, "%s: stalled by EXON_HIP_FAULT", what);
  }
  int64_t mine[2] = {my_code, 0};
  std::vector<int64_t> all((size_t)c->world * 2, 0);
  const int rc = comm_all_gather_host(ctx, c, s, d_scratch, mine, all.data(), 16, what);
  if (rc) return rc;
  for (int r = 0; r < c->world; ++r)
    if (all[(size_t)r * 2] != 0) {
      const int code = (int)all[(size_t)r * 2];
      if (r == c->rank) return fail(ctx, code, "%s: %s (this rank, %d; every rank of the communicator returns this error, none has entered the data exchange)", what, why ? why : "failed", r);
      return fail(ctx, code, "%s: rank %d could not go on (status %d: its own error text says why); no rank has entered the data exchange", what, r, code);
    }
  return EXON_HIP_OK;
}

// ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy for hosts without an RCCL binding of their own: rank 0 creates the id,
// ships its 128 bytes to the other ranks by any means, every rank (one process per GPU) calls comm_init on its ctx.
int exon_hip_rccl_unique_id(uint8_t* id128) {
  if (!id128) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_rccl_unique_id: NULL argument");
  if (!rccl().ok()) return fail(nullptr, EXON_HIP_EUNSUPPORTED, "librccl.so could not be loaded");
  UniqueId id;
  const int e = rccl().get_unique_id(&id);
  if (e) return fail(nullptr, EXON_HIP_EDEVICE, "ncclGetUniqueId failed with ncclResult_t %d", e);
  memcpy(id128, id.internal, 128);
  return EXON_HIP_OK;
}
int exon_hip_rccl_comm_init(exon_hip_ctx* ctx, const uint8_t* id128, int32_t world, int32_t rank, void** comm) {
  if (!ctx || !id128 || !comm) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_rccl_comm_init: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, EXON_HIP_EINVAL, "rank %d outside a world of %d", rank, world);
  if (!rccl().ok()) return fail(ctx, EXON_HIP_EUNSUPPORTED, "librccl.so could not be loaded");
  UniqueId id;
  memcpy(id.internal, id128, 128);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  typedef int (*init_fn)(void**, int, UniqueId, int);
  *comm = nullptr;
  void* nccl = nullptr;
  const int e = ((init_fn)rccl().init_rank)(&nccl, world, id, rank);
  if (e) return fail(ctx, EXON_HIP_EDEVICE, "ncclCommInitRank(world %d, rank %d) failed with ncclResult_t %d", world, rank, e);
  ExonComm* c = new ExonComm();
  c->kind = 0;
  c->nccl = nccl;
  c->world = world;
  c->rank = rank;
  *comm = c;
  return EXON_HIP_OK;
}
int exon_hip_comm_wrap_rccl(void* nccl_comm, void** comm) {
  if (!nccl_comm || !comm) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_comm_wrap_rccl: NULL argument");
  if (!rccl().ok()) return fail(nullptr, EXON_HIP_EUNSUPPORTED, "librccl.so could not be loaded");
  int w = 0, r = 0;
  int e = rccl().comm_count(nccl_comm, &w);
  if (!e) e = rccl().comm_rank(nccl_comm, &r);
  if (e || w < 1) return fail(nullptr, EXON_HIP_EDEVICE, "ncclCommCount / ncclCommUserRank failed with ncclResult_t %d", e);
  ExonComm* c = new ExonComm();
  c->kind = 0;
  c->nccl = nccl_comm;
  c->world = w;
  c->rank = r;
  c->owned = false;
  *comm = c;
  return EXON_HIP_OK;
}
int exon_hip_comm_from_callbacks(int32_t world, int32_t rank, exon_hip_allgather_fn all_gather, void* user, void** comm) {
  if (!all_gather || !comm) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_comm_from_callbacks: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(nullptr, EXON_HIP_EINVAL, "rank %d outside a world of %d", rank, world);
  ExonComm* c = new ExonComm();
  c->kind = 1;
  c->world = world;
  c->rank = rank;
  c->fn = all_gather;
  c->user = user;
  *comm = c;
  return EXON_HIP_OK;
}
int exon_hip_rccl_comm_destroy(void* comm) {
  if (!comm) return EXON_HIP_OK;
  ExonComm* c = comm_of(comm);
  if (!c) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_rccl_comm_destroy: not a communicator of this library");
  int e = 0;
  if (c->kind == 0 && c->nccl && c->owned) e = rccl().comm_destroy(c->nccl);
  c->magic = 0;
  delete c;
  return e ? fail(nullptr, EXON_HIP_EDEVICE, "ncclCommDestroy failed with ncclResult_t %d", e) : EXON_HIP_OK;
}

int exon_hip_rccl_comm_count(void* comm, int32_t* world, int32_t* rank) {
  if (!comm || !world) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_rccl_comm_count: NULL argument");
  ExonComm* c = comm_of(comm);
  if (!c) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_rccl_comm_count: not a communicator of this library");
  int w = c->world, r = c->rank;
  if (c->kind == 0 && c->nccl) {  // what RCCL itself says (a launcher prints it to prove how many GPUs merged)
    int e = rccl().comm_count(c->nccl, &w);
    if (!e && rank) e = rccl().comm_rank(c->nccl, &r);
    if (e) return fail(nullptr, EXON_HIP_EDEVICE, "ncclCommCount / ncclCommUserRank failed with ncclResult_t %d", e);
  }
  *world = w;
  if (rank) *rank = r;
  return EXON_HIP_OK;
}

// Merge of packed partial states across the ranks of `comm`, enqueued on `stream`: ONE all-gather of the state
// (d_state -> d_gather[world][words]) + the fixed-order fold into d_out (may be d_state itself).  The f64 sums come out
// bit-identical on every rank and for every collective algorithm RCCL may pick.  States above 1 MiB are integer counters
// only (K3 with millions of references) or too large to gather 8x: those are all-reduced in place (integer sums are exact).
// No vote here (the caller owns the buffers: nothing in this call can fail on one rank alone but the collective itself).
int exon_hip_merge_states(exon_hip_ctx* ctx, void* stream, void* comm, void* d_state, int64_t n_i64, int64_t n_f64,
                          void* d_gather, void* d_out) {
  if (!ctx || !comm || !d_state || !d_out) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_merge_states: NULL argument");
  ExonComm* c = comm_of(comm);
  if (!c) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_merge_states: not a communicator of this library (exon_hip_rccl_comm_init / exon_hip_comm_from_callbacks)");
  if (c->kind == 0 && !rccl().ok()) return fail(ctx, EXON_HIP_EUNSUPPORTED, "librccl.so could not be loaded");
  if (n_i64 < 0 || n_f64 < 0 || n_i64 + n_f64 < 1) return fail(ctx, EXON_HIP_EINVAL, "empty state");
  hipStream_t s = pick_stream(ctx, stream);
  const size_t words = (size_t)(n_i64 + n_f64);
  if (d_gather != nullptr && n_i64 + n_f64 > INT32_MAX)  // the fold kernel indexes words with an int (same limit as exon_hip_fold_states)
    return fail(ctx, EXON_HIP_EINVAL, "state of %lld words is too large for the gather + fold form", (long long)(n_i64 + n_f64));
  if (d_gather == nullptr) {  // in-place all-reduce form (large integer states)
    if (n_f64) return fail(ctx, EXON_HIP_EINVAL, "a state with float64 sums is merged by gather + fold: pass d_gather");
    if (d_out != d_state) return fail(ctx, EXON_HIP_EINVAL, "the all-reduce form is in place");
    if (c->aborted) return fail(ctx, EXON_HIP_ESTATE, "the communicator was aborted by an earlier time-out");
    if (c->kind == 0) {
      const int e = rccl().all_reduce(d_state, d_state, words, NCCL_INT64, NCCL_SUM, c->nccl, s);
      return e ? fail(ctx, EXON_HIP_EDEVICE, "ncclAllReduce failed with ncclResult_t %d", e) : EXON_HIP_OK;
    }
    std::vector<int64_t> mine(words), all(words * (size_t)c->world);
    HIP_TRY(ctx, hipMemcpyAsync(mine.data(), d_state, words * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (c->fn(c->user, mine.data(), all.data(), words * 8)) return fail(ctx, EXON_HIP_EDEVICE, "the caller's all-gather failed");
    for (size_t w = 0; w < words; ++w) {
      int64_t t = 0;
      for (int r = 0; r < c->world; ++r) t += all[(size_t)r * words + w];
      mine[w] = t;
    }
    HIP_TRY(ctx, hipMemcpyAsync(d_state, mine.data(), words * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return EXON_HIP_OK;
  }
  const int world = c->world;
  {  // the fold reads every rank's copy while it writes d_out: d_out inside the gather buffer would race
    const uintptr_t g0 = reinterpret_cast<uintptr_t>(d_gather), g1 = g0 + (size_t)world * words * 8, o0 = reinterpret_cast<uintptr_t>(d_out);
    if (o0 + words * 8 > g0 && o0 < g1) return fail(ctx, EXON_HIP_EINVAL, "d_out must not lie inside d_gather");
  }
  const int rc = comm_all_gather_dev(ctx, c, s, d_state, d_gather, words * 8);  // 8-byte words; no arithmetic in flight
  if (rc) return rc;
  HIP_TRY(ctx, exon::launch_fold_states(s, d_gather, world, n_i64, n_f64, d_out));
  return EXON_HIP_OK;
}

// AggregateExec(Final) across GPUs in native code, on the stream's hipStream_t: afterwards every rank's state holds the
// sum over all ranks (the name is kept from ABI 1; since ABI 2 it is one all-gather + a fixed-order fold, see above).
// Collective: every rank calls it.  Whatever stops ONE rank -- the stream finished, a state keyed by its own dictionary, rows
// that cannot be launched, no memory for the gather buffer -- is put to the vote first: all ranks return that error, none waits
// in the all-gather for a peer that has left.  (Costs one 16-byte all-gather in front of the merge; exon_hip_merge_states is
// the form without it.)
int exon_hip_stream_all_reduce(exon_hip_stream* st, void* comm) {
  if (!st || !comm) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_all_reduce: NULL argument");
  ExonComm* c = comm_of(comm);
  if (!c) return fail(st->ctx, EXON_HIP_EINVAL, "exon_hip_stream_all_reduce: not a communicator of this library");
  if (c->kind == 0 && !rccl().ok()) return fail(st->ctx, EXON_HIP_EUNSUPPORTED, "librccl.so could not be loaded");  // (the same on every rank of a node)
  hipSetDevice(st->ctx->device);
  const exon_hip_plan* p = st->plan;
  const size_t sbytes = (size_t)(p->n_i64 + p->n_f64) * 8;
  uint8_t* d_scratch = st->d_state + ((sbytes + 63) & ~(size_t)63);
  int code = EXON_HIP_OK;
  std::string why;
  auto local = [&](int rc_, const char* text) {
    if (code == EXON_HIP_OK) {
      code = rc_;
      why = text;
    }
  };
  if (fault_at("allreduce_enter", c->rank)) local(EXON_HIP_ESTATE, "EXON_HIP_FAULT=allreduce_enter");
  if (st->closed) local(EXON_HIP_ESTATE, "all_reduce after finish/close");
  if (st->keys_state == KEYS_LOCAL)  // ids of file-derived keys are per rank: adding states by id would merge different groups
    local(EXON_HIP_ESTATE, "the state is keyed by this rank's own dictionary: call exon_hip_stream_reconcile_keys (or exon_hip_stream_set_keys with the agreed "
                           "dictionary) on every rank before the merge");
  if (code == EXON_HIP_OK) {
    int rc = flush_slot(st);
    if (!rc) rc = settle_reset(st);
    if (rc) local(rc, exon_hip_last_error(st->ctx));
  }
  const bool big = sbytes > GATHER_MERGE_MAX_STATE && p->n_f64 == 0;
  const int world = c->world;
  if (code == EXON_HIP_OK && !big && st->gather_bytes < sbytes * (size_t)world) {
    if (st->d_gather) {
      hipStreamSynchronize(st->stream);
      hipFree(st->d_gather);
      st->d_gather = nullptr;
      st->gather_bytes = 0;
    }
    if (fault_at("allreduce_malloc", c->rank) || hipMalloc((void**)&st->d_gather, sbytes * (size_t)world) != hipSuccess) {
      (void)hipGetLastError();
      st->d_gather = nullptr;
      local(EXON_HIP_ENOMEM, "no memory for the gather buffer");
    } else {
      st->gather_bytes = sbytes * (size_t)world;
    }
  }
  const int v = comm_vote(st->ctx, c, st->stream, d_scratch, code, why.c_str(), "exon_hip_stream_all_reduce");
  if (v) return v;
  if (big) return exon_hip_merge_states(st->ctx, st->stream, comm, st->d_state, p->n_i64, 0, nullptr, st->d_state);
  return exon_hip_merge_states(st->ctx, st->stream, comm, st->d_state, p->n_i64, p->n_f64, st->d_gather, st->d_state);
}

// ---- group keys by value: the C ABI (include/exon_hip.h "group keys") ---------------------------------------------------------
// '\0'-terminated names back to back -> vector; false when the buffer ends before n names do
static bool unpack_names(const char* packed, size_t bytes, int32_t n, std::vector<std::string>* out) {
  size_t o = 0;
  for (int32_t i = 0; i < n; ++i) {
    const void* z = o < bytes ? memchr(packed + o, 0, bytes - o) : nullptr;
    if (!z) return false;
    const size_t len = (size_t)(static_cast<const char*>(z) - (packed + o));
    out->emplace_back(packed + o, len);
    o += len + 1;
  }
  return true;
}
static size_t packed_size(const std::vector<std::string>& names) {
  size_t b = 0;
  for (const auto& k : names) b += k.size() + 1;
  return b;
}
static void pack_names(const std::vector<std::string>& names, char* out) {
  for (const auto& k : names) {
    memcpy(out, k.data(), k.size());
    out[k.size()] = 0;
    out += k.size() + 1;
  }
}
// union of `world` dictionaries in rank order, first appearance first (rank 0's keys keep their ids: its frequent keys stay
// early, where the kernels keep them in registers); maps (optional): union id of every input key, rank-major
static void union_in_rank_order(const std::vector<std::vector<std::string>>& dicts, std::vector<std::string>* uni, std::vector<int32_t>* maps) {
  std::unordered_map<std::string, int32_t> index;
  for (const auto& d : dicts)
    for (const auto& k : d) {
      auto it = index.find(k);
      if (it == index.end()) {
        it = index.emplace(k, (int32_t)uni->size()).first;
        uni->push_back(k);
      }
      if (maps) maps->push_back(it->second);
    }
}

int exon_hip_keys_union(const char* packed, size_t packed_bytes, const int32_t* n_keys, int32_t world, char* out, size_t cap,
                        int32_t* n_out, size_t* out_bytes, int32_t* maps) {
  if ((!packed && packed_bytes) || !n_keys || world < 1 || !n_out || !out_bytes) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_keys_union: bad argument");
  std::vector<std::vector<std::string>> dicts((size_t)world);
  size_t o = 0;
  for (int32_t r = 0; r < world; ++r) {
    if (n_keys[r] < 0) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_keys_union: n_keys[%d] < 0", r);
    if (!unpack_names(packed + o, packed_bytes - o, n_keys[r], &dicts[(size_t)r]))
      return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_keys_union: the packed names end before rank %d's %d keys do", r, n_keys[r]);
    o += packed_size(dicts[(size_t)r]);
  }
  std::vector<std::string> uni;
  std::vector<int32_t> m;
  union_in_rank_order(dicts, &uni, maps ? &m : nullptr);
  *n_out = (int32_t)uni.size();
  *out_bytes = packed_size(uni);
  if (maps) memcpy(maps, m.data(), m.size() * 4);
  if (out) {
    if (cap < *out_bytes) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_keys_union: output needs %zu bytes, %zu given", *out_bytes, cap);
    pack_names(uni, out);
  }
  return EXON_HIP_OK;
}

int exon_hip_stream_keys(exon_hip_stream* st, char* buf, size_t cap, int32_t* n_keys, size_t* bytes, int32_t* agreed) {
  if (!st || !n_keys || !bytes) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_keys: NULL argument");
  *n_keys = (int32_t)st->keys.size();
  *bytes = packed_size(st->keys);
  if (agreed) *agreed = st->keys_state == KEYS_AGREED ? 1 : 0;
  if (buf) {
    if (cap < *bytes) return fail(st->ctx, EXON_HIP_EINVAL, "exon_hip_stream_keys: the names need %zu bytes, %zu given", *bytes, cap);
    pack_names(st->keys, buf);
  }
  return EXON_HIP_OK;
}

// adopt `names` as the stream's dictionary: state index g stands for names[g] afterwards.  A stream without keys takes the
// names as a declaration of what its ids mean; a keyed stream must find every key it holds in `names` and has its state
// permuted into the new order.
static int adopt_keys(exon_hip_stream* st, const std::vector<std::string>& names) {
  int G = 0, pi, tail, pf;
  if (!key_layout(st->plan, &G, &pi, &tail, &pf)) return fail(st->ctx, EXON_HIP_EINVAL, "this plan does not group by a key");
  if ((int64_t)names.size() > G)
    return fail(st->ctx, EXON_HIP_ECAPACITY, "%zu group keys do not fit the plan's n_groups = %d (create the plan for the union's size)", names.size(), G);
  {
    std::unordered_map<std::string, int> seen;
    for (const auto& k : names)
      if (!seen.emplace(k, 0).second) return fail(st->ctx, EXON_HIP_EINVAL, "group key '%s' appears twice in the dictionary", k.c_str());
  }
  if (st->keys_state != KEYS_NONE && !st->keys.empty()) {
    std::unordered_map<std::string, int32_t> index;
    for (size_t i = 0; i < names.size(); ++i) index.emplace(names[i], (int32_t)i);
    std::vector<int32_t> map(st->keys.size());
    bool identity = true;
    for (size_t i = 0; i < st->keys.size(); ++i) {
      auto it = index.find(st->keys[i]);
      if (it == index.end()) return fail(st->ctx, EXON_HIP_EINVAL, "the new dictionary lacks the key '%s' this stream holds rows for", st->keys[i].c_str());
      map[i] = it->second;
      identity = identity && map[i] == (int32_t)i;
    }
    if (!identity) {
      int rc = flush_slot(st);
      if (!rc) rc = settle_reset(st);
      if (!rc) rc = ensure_rekey_buffers(st, map.size());
      if (rc) return rc;
      const size_t sb = state_bytes_of(st);
      HIP_TRY(st->ctx, hipMemsetAsync(st->d_rekey_state, 0, sb, st->stream));
      rc = permute_add(st, st->d_state, st->d_rekey_state, map);
      if (rc) return rc;
      HIP_TRY(st->ctx, hipMemcpyAsync(st->d_state, st->d_rekey_state, sb, hipMemcpyDeviceToDevice, st->stream));
    }
  }
  st->keys = names;
  st->keys_state = KEYS_AGREED;
  return EXON_HIP_OK;
}

int exon_hip_stream_set_keys(exon_hip_stream* st, const char* packed, size_t packed_bytes, int32_t n_keys) {
  if (!st || (!packed && n_keys) || n_keys < 0) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_set_keys: bad argument");
  if (st->closed) return fail(st->ctx, EXON_HIP_ESTATE, "set_keys after finish/close");
  std::vector<std::string> names;
  if (!unpack_names(packed, packed_bytes, n_keys, &names)) return fail(st->ctx, EXON_HIP_EINVAL, "exon_hip_stream_set_keys: the packed names end before %d keys do", n_keys);
  return adopt_keys(st, names);
}

int exon_hip_stream_set_region_contig(exon_hip_stream* st, const char* name) {
  if (!st || !name) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_set_region_contig: NULL argument");
  const int k = st->plan->d.kind;
  if (k != EXON_HIP_PLAN_REGION_COUNT && k != EXON_HIP_PLAN_OVERLAP_COUNT && k != EXON_HIP_PLAN_WITHIN_COUNT)
    return fail(st->ctx, EXON_HIP_EINVAL, "plan kind %d has no region", k);
  st->region_contig = name;
  st->has_region_contig = true;
  return EXON_HIP_OK;
}

// The ranks of `comm` agree on ONE dictionary: the sizes (with the vote folded in), one more vote behind the local allocations, the
// packed names padded to the longest, the union in rank order on every rank, each state permuted into it.  Collective: every rank
// must call it -- and every rank leaves it with the same verdict: nothing that can stop one rank alone (the stream finished, rows
// pushed under undeclared ids, no memory for the exchange or the re-keying buffers) sits between two exchanges unannounced.
int exon_hip_stream_reconcile_keys(exon_hip_stream* st, void* comm) {
  if (!st || !comm) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_reconcile_keys: NULL argument");
  ExonComm* c = comm_of(comm);
  if (!c) return fail(st->ctx, EXON_HIP_EINVAL, "exon_hip_stream_reconcile_keys: not a communicator of this library");
  if (!exon_hip_stream_is_keyed(st)) return EXON_HIP_OK;  // plans without group keys have nothing to agree on (the plan is the same on every rank)
  if (c->kind == 0 && !rccl().ok()) return fail(st->ctx, EXON_HIP_EUNSUPPORTED, "librccl.so could not be loaded");
  const int world = c->world;
  if ((size_t)world * 32 > COLL_SCRATCH / 2) return fail(st->ctx, EXON_HIP_EUNSUPPORTED, "exon_hip_stream_reconcile_keys: more than %zu ranks", COLL_SCRATCH / 64);
  hipSetDevice(st->ctx->device);
  const size_t sbytes = state_bytes_of(st);
  uint8_t* d_scratch = st->d_state + ((sbytes + 63) & ~(size_t)63);
  // (1) sizes + vote.  A rank that may not take part (finished; rows pushed under the caller's own ids, never declared) still ENTERS
  // and says so: every rank then fails together.
  const bool unkeyed = st->keys_state == KEYS_NONE && st->rows_pushed > 0;
  int code = EXON_HIP_OK;
  if (st->closed) code = EXON_HIP_ESTATE;
  if (fault_at("reconcile_enter", c->rank)) code = EXON_HIP_ESTATE;
  const bool mute = unkeyed || code != EXON_HIP_OK;
  if (fault_at("stall", c->rank)) {
    std::this_thread::sleep_for(std::chrono::duration<double>(collective_timeout_s() * 3));
    return fail(st->ctx, EXON_HIP_EDEVICE, "exon_hip_stream_reconcile_keys: stalled by EXON_HIP_FAULT");
  }
  int64_t mine[4] = {mute ? 0 : (int64_t)st->keys.size(), mute ? 0 : (int64_t)packed_size(st->keys), unkeyed ? 1 : 0, code};
  std::vector<int64_t> sizes4((size_t)world * 4, 0);
  int rc = comm_all_gather_host(st->ctx, c, st->stream, d_scratch, mine, sizes4.data(), 32, "exon_hip_stream_reconcile_keys (sizes)");
  if (rc) return rc;
  for (int r = 0; r < world; ++r) {  // decided by all ranks on the same data: all of them leave here
    if (sizes4[(size_t)r * 4 + 3])
      return fail(st->ctx, (int)sizes4[(size_t)r * 4 + 3], "exon_hip_stream_reconcile_keys: rank %d cannot take part (%s); no rank's state was touched", r,
                  r == c->rank && st->closed ? "its stream has finished" : "status from that rank");
    if (sizes4[(size_t)r * 4 + 2])
      return fail(st->ctx, EXON_HIP_ESTATE,
                  "rank %d pushed rows under the caller's own dictionary ids and never declared their values (exon_hip_stream_set_keys): "
                  "no rank's state was touched", r);
  }
  size_t slot = 8;
  for (int r = 0; r < world; ++r) slot = std::max(slot, (size_t)sizes4[(size_t)r * 4 + 1]);
  slot = (slot + 7) / 8 * 8;
  // (2) what this rank needs for the exchange and for the re-keying behind it, BEFORE the names travel; then the second vote
  std::vector<char> all(slot * (size_t)world), own(slot, 0);
  pack_names(st->keys, own.data());
  char* d_txt = nullptr;
  std::string why;
  if (fault_at("reconcile_malloc", c->rank) || (c->kind == 0 && hipMalloc((void**)&d_txt, slot * (size_t)(world + 1)) != hipSuccess)) {
    (void)hipGetLastError();
    d_txt = nullptr;
    code = EXON_HIP_ENOMEM;
    why = "no memory for the key exchange";
  }
  if (code == EXON_HIP_OK && (fault_at("reconcile_rekey", c->rank) || ensure_rekey_buffers(st, st->keys.size()) != EXON_HIP_OK)) {
    code = EXON_HIP_ENOMEM;
    why = "no memory for the re-keying buffers";
  }
  if (code == EXON_HIP_OK) {  // rows staged by earlier pushes belong to the state that is about to be permuted
    int rf = flush_slot(st);
    if (!rf) rf = settle_reset(st);
    if (rf) {
      code = rf;
      why = exon_hip_last_error(st->ctx);
    }
  }
  rc = comm_vote(st->ctx, c, st->stream, d_scratch, code, why.c_str(), "exon_hip_stream_reconcile_keys");
  if (rc) {
    if (d_txt) hipFree(d_txt);
    return rc;
  }
  // (3) the packed names, one padded slot per rank
  if (c->kind == 0) {
    hipError_t he = hipMemcpyAsync(d_txt + slot * (size_t)world, own.data(), slot, hipMemcpyHostToDevice, st->stream);
    int e = 0;
    if (he == hipSuccess) e = rccl().all_gather(d_txt + slot * (size_t)world, d_txt, slot / 8, NCCL_INT64, c->nccl, st->stream);
    if (he == hipSuccess && !e) he = hipMemcpyAsync(all.data(), d_txt, all.size(), hipMemcpyDeviceToHost, st->stream);
    int rs = EXON_HIP_OK;
    if (he == hipSuccess && !e) rs = bounded_sync(st->ctx, c, st->stream, "exon_hip_stream_reconcile_keys (names)");
    hipFree(d_txt);
    if (e) return fail(st->ctx, EXON_HIP_EDEVICE, "ncclAllGather (key dictionaries) failed with ncclResult_t %d", e);
    if (he != hipSuccess) return fail(st->ctx, EXON_HIP_EDEVICE, "key dictionaries: %s", hipGetErrorString(he));
    if (rs) return rs;
  } else if (c->fn(c->user, own.data(), all.data(), slot)) {
    return fail(st->ctx, EXON_HIP_EDEVICE, "exon_hip_stream_reconcile_keys: the caller's all-gather failed");
  }
  std::vector<std::vector<std::string>> dicts((size_t)world);
  for (int r = 0; r < world; ++r)
    if (!unpack_names(all.data() + slot * (size_t)r, (size_t)sizes4[(size_t)r * 4 + 1], (int32_t)sizes4[(size_t)r * 4], &dicts[(size_t)r]))
      return fail(st->ctx, EXON_HIP_EINVAL, "rank %d sent a malformed key dictionary", r);  // (the same bytes on every rank: all of them leave here)
  std::vector<std::string> uni;
  union_in_rank_order(dicts, &uni, nullptr);
  if (st->keys_state == KEYS_NONE) st->keys_state = KEYS_LOCAL;  // a rank that scanned nothing holds an empty dictionary
  // the union and n_groups are the same on every rank (ECAPACITY is everybody's); the buffers it needs exist since (2)
  return adopt_keys(st, uni);
}

// The next launch on this stream defines the state instead of adding to it: a new query on the same stream without a
// zeroing kernel (finalize writes in overwrite mode).  Rows staged but not yet launched belong to the old query and are
// dropped with it.
int exon_hip_stream_reset(exon_hip_stream* st) {
  if (!st) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_stream_reset: NULL stream");
  Slot& s = st->slots[st->cur];
  release_held(s);  // (batches held for the old query: dropped with it, released uncopied)
  if (!s.cols.empty()) {
    s.rows = 0;
    s.bytes = 0;
    for (auto& cs : s.cols) {
      if (cs.any_null_bitmap) memset(cs.h_valid, 0, (size_t)(st->cap_rows + 7) / 8 + 64);
      cs.any_null_bitmap = false;
      if (cs.h_offsets) cs.h_offsets[0] = 0;
    }
  }
  st->overwrite_next = true;
  st->closed = false;
  st->rows_pushed = 0;
  st->keys.clear();  // a new query: its scans define the key dictionary afresh
  st->keys_state = KEYS_NONE;
  return EXON_HIP_OK;
}

// Stateless form of a push: run the plan's fused kernel over HBM-resident columns (operator argument order, see
// exon_hip_plan_desc.columns) on the caller's hipStream_t into a caller-owned packed state [n_i64 x int64][n_f64 x float64].
int exon_hip_plan_launch(exon_hip_plan* plan, void* stream, const exon_hip_column* columns, int32_t n_columns, int64_t n,
                         int32_t flags, void* d_state) {
  if (!plan || !columns || !d_state) return fail(plan ? plan->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_plan_launch: NULL argument");
  if (n_columns != plan->n_cols) return fail(plan->ctx, EXON_HIP_EINVAL, "plan takes %d columns, %d given", plan->n_cols, n_columns);
  if (flags & ~EXON_HIP_LAUNCH_OVERWRITE) return fail(plan->ctx, EXON_HIP_EINVAL, "unknown launch flags 0x%x", flags);
  if (reinterpret_cast<uintptr_t>(d_state) & 7) return fail(plan->ctx, EXON_HIP_EINVAL, "d_state must be 8-byte aligned");
  return run_plan(plan, stream, columns, n, flags, d_state);
}

int exon_hip_plan_launch_chunks(exon_hip_plan* plan, void* stream, const exon_hip_column* columns, int32_t n_columns,
                                int32_t n_chunks, const int64_t* n, int32_t flags, void* d_state) {
  if (!plan || !d_state) return fail(plan ? plan->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_plan_launch_chunks: NULL argument");
  if (n_chunks < 0 || (n_chunks && (!columns || !n))) return fail(plan->ctx, EXON_HIP_EINVAL, "exon_hip_plan_launch_chunks: bad chunk list");
  if (n_columns != plan->n_cols) return fail(plan->ctx, EXON_HIP_EINVAL, "plan takes %d columns, %d given", plan->n_cols, n_columns);
  if (flags & ~EXON_HIP_LAUNCH_OVERWRITE) return fail(plan->ctx, EXON_HIP_EINVAL, "unknown launch flags 0x%x", flags);
  if (reinterpret_cast<uintptr_t>(d_state) & 7) return fail(plan->ctx, EXON_HIP_EINVAL, "d_state must be 8-byte aligned");
  if (plan->d.kind == EXON_HIP_PLAN_QUAL_POS_HIST)
    return exon_op_qual_pos_hist_chunks(plan->ctx, stream, columns, n_columns, n_chunks, n, plan->d.lmax,
                                        reinterpret_cast<int64_t*>(d_state), flags);
  bool first = true;
  for (int c = 0; c < n_chunks; ++c) {
    if (n[c] < 0) return fail(plan->ctx, EXON_HIP_EINVAL, "n < 0");
    if (n[c] == 0) continue;
    int rc = run_plan(plan, stream, columns + (size_t)c * n_columns, n[c], first ? flags : EXON_HIP_LAUNCH_ACCUMULATE, d_state);
    if (rc) return rc;
    first = false;
  }
  if (first && (flags & EXON_HIP_LAUNCH_OVERWRITE))  // nothing launched: an overwrite of nothing is the empty state
    HIP_TRY(plan->ctx, hipMemsetAsync(d_state, 0, (size_t)(plan->n_i64 + plan->n_f64) * 8, pick_stream(plan->ctx, stream)));
  return EXON_HIP_OK;
}

int exon_hip_stream_sync(exon_hip_stream* st) {
  if (!st) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_stream_sync: NULL stream");
  int rc = flush_slot(st);
  if (!rc) rc = settle_reset(st);
  if (rc) return rc;
  return exon_hip_sync(st->ctx, st->stream);
}

int exon_hip_stream_finish(exon_hip_stream* st, int64_t* counts, double* sums) {
  if (!st) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_stream_finish: NULL stream");
  int rc = exon_hip_stream_sync(st);
  if (rc) return rc;
  st->closed = true;
  if (st->plan->n_i64 && !counts) return fail(st->ctx, EXON_HIP_EINVAL, "counts is NULL");
  if (st->plan->n_f64 && !sums) return fail(st->ctx, EXON_HIP_EINVAL, "sums is NULL");
  if (st->plan->n_i64) HIP_TRY(st->ctx, hipMemcpy(counts, st->d_state, (size_t)st->plan->n_i64 * 8, hipMemcpyDeviceToHost));
  if (st->plan->n_f64)
    HIP_TRY(st->ctx, hipMemcpy(sums, st->d_state + st->plan->n_i64 * 8, (size_t)st->plan->n_f64 * 8, hipMemcpyDeviceToHost));
  return EXON_HIP_OK;
}

}  // extern "C"

// ---- Arrow export of the partial-aggregate state ---------------------------------------------------
namespace {
using exon::make_schema;
using exon::make_struct;
template <typename T>
struct ArrowArray* prim(const std::vector<T>& v, const std::vector<uint8_t>* valid = nullptr) {
  struct ArrowArray* a = (struct ArrowArray*)malloc(sizeof *a);
  exon::make_primitive(a, v.data(), (int64_t)v.size(), (int)sizeof(T), valid ? *valid : std::vector<uint8_t>());
  return a;
}
struct ArrowSchema* field(const char* fmt, const char* name, bool nullable) { return exon::new_field(fmt, name, nullable); }
}  // namespace

extern "C" {

int exon_hip_stream_finish_arrow(exon_hip_stream* st, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  if (!st || !out || !out_schema) return fail(st ? st->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_stream_finish_arrow: NULL argument");
  const exon_hip_plan* p = st->plan;
  std::vector<int64_t> counts((size_t)p->n_i64);
  std::vector<double> sums((size_t)p->n_f64);
  int rc = exon_hip_stream_finish(st, counts.data(), sums.data());
  if (rc) return rc;
  const exon_hip_plan_desc& d = p->d;
  switch (d.kind) {
    case EXON_HIP_PLAN_OVERLAP_COUNT:
    case EXON_HIP_PLAN_WITHIN_COUNT:
    case EXON_HIP_PLAN_REGION_COUNT: {
      make_struct(out, 1, {prim(counts)});
      make_schema(out_schema, "+s", "", false, {field("l", "count(*)[count]", false)});
      break;
    }
    case EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT: {
      std::vector<int32_t> key;
      std::vector<uint8_t> valid;
      std::vector<int64_t> cnt;
      for (int g = 0; g <= d.n_groups; ++g)
        if (counts[(size_t)g]) {
          key.push_back(g < d.n_groups ? g : 0);
          valid.push_back(g < d.n_groups);
          cnt.push_back(counts[(size_t)g]);
        }
      make_struct(out, (int64_t)key.size(), {prim(key, &valid), prim(cnt)});
      make_schema(out_schema, "+s", "", false, {field("i", "reference", true), field("l", "count(*)[count]", false)});
      break;
    }
    case EXON_HIP_PLAN_CMP_AVG_BY_GROUP: {
      const int G = d.n_groups;
      std::vector<int32_t> key;
      std::vector<uint64_t> acnt;
      std::vector<double> asum;
      std::vector<int64_t> rows;
      for (int g = 0; g < G; ++g)
        if (counts[(size_t)(G + g)]) {
          key.push_back(g);
          acnt.push_back((uint64_t)counts[(size_t)g]);
          asum.push_back(sums[(size_t)g]);
          rows.push_back(counts[(size_t)(G + g)]);
        }
      make_struct(out, (int64_t)key.size(), {prim(key), prim(acnt), prim(asum), prim(rows)});
      make_schema(out_schema, "+s", "", false,
                  {field("i", "group", false), field("L", "avg[count]", false), field("g", "avg[sum]", false),
                   field("l", "count(*)[count]", false)});
      break;
    }
    case EXON_HIP_PLAN_QUAL_POS_HIST: {
      std::vector<int32_t> pos, score;
      std::vector<int64_t> cnt;
      for (int pp = 0; pp < d.lmax; ++pp)
        for (int b = 0; b < 256; ++b)
          if (counts[(size_t)pp * 256 + b]) {
            pos.push_back(pp);
            score.push_back(b - 33);
            cnt.push_back(counts[(size_t)pp * 256 + b]);
          }
      make_struct(out, (int64_t)pos.size(), {prim(pos), prim(score), prim(cnt)});
      make_schema(out_schema, "+s", "", false,
                  {field("i", "position", false), field("i", "quality_score", false), field("l", "count(*)[count]", false)});
      break;
    }
    default:
      return fail(st->ctx, EXON_HIP_EINVAL, "unknown plan kind");
  }
  return EXON_HIP_OK;
}

int exon_hip_stream_close(exon_hip_stream* st) {
  if (!st) return EXON_HIP_OK;
  if (getenv("EXON_HIP_STAGE_TRACE"))
    fprintf(stderr, "[exon-hip stage] %lld rows: staging copies %.1f ms, H2D + launch calls %.1f ms, waiting for a free slot %.1f ms\n",
            (long long)st->rows_pushed, st->t_copy * 1e3, st->t_enqueue * 1e3, st->t_wait * 1e3);
  hipSetDevice(st->ctx->device);
  if (st->stream) hipStreamSynchronize(st->stream);
  for (auto& s : st->slots) free_slot(s);
  {
    // drop this stream's workspace
    std::lock_guard<std::mutex> g(st->ctx->mu);
    auto it = st->ctx->workspaces.find(st->stream);
    if (it != st->ctx->workspaces.end()) {
      if (it->second.partials) hipFree(it->second.partials);
      if (it->second.status) hipFree(it->second.status);
      if (it->second.tail_rec_a) hipFree(it->second.tail_rec_a);
      if (it->second.tail_u32) hipFree(it->second.tail_u32);
      st->ctx->workspaces.erase(it);
    }
  }
  if (st->d_state) hipFree(st->d_state);
  if (st->d_gather) hipFree(st->d_gather);
  if (st->d_rekey_state) hipFree(st->d_rekey_state);
  if (st->d_map) hipFree(st->d_map);
  if (st->stream) hipStreamDestroy(st->stream);
  delete st;
  return EXON_HIP_OK;
}

}  // extern "C"
