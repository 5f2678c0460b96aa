// capi.cpp -- the extern "C" boundary declared in include/exon_hip.h.
//
// Layers:
//   ctx     device, default stream, per-stream workspaces, launch shape, last-error text
//   raw     exon_hip_<operator>(): validate, pick the workspace, launch on the caller's hipStream_t
//   plan    immutable operator description (which fused filter+aggregate kernel, its literals)
//   stream  one partition's execution: owns a HIP stream, the device-resident partial state, pinned
//           staging; host Arrow batches are coalesced into large device batches before a launch
// No exceptions cross the boundary; every failure returns a negative status and records its text.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/exon_hip.h"
#include "host/region.h"
#include "kernels.h"

#include "internal.h"

using exon::LaunchCfg;
using exon::Workspace;


static thread_local std::string tls_error;
const std::string& exon_hip_tls_error() { return tls_error; }

int fail(exon_hip_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  tls_error = buf;
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->error = buf;
  }
  return code;
}

int get_workspace(exon_hip_ctx* ctx, hipStream_t s, size_t words, Workspace* out) {
  std::lock_guard<std::mutex> g(ctx->mu);
  Workspace& ws = ctx->workspaces[s];
  if (!ws.status) {
    // status[0] = device error word, status[1..7] = scratch flags (K5 path selection), then 64 bytes of 0xFF
    // (stand-in for absent validity bitmaps)
    if (hipMalloc(&ws.status, 8 * sizeof(int) + 64) != hipSuccess) return EXON_HIP_ENOMEM;
    if (hipMemset(ws.status, 0, 8 * sizeof(int)) != hipSuccess || hipMemset(ws.status + 8, 0xFF, 64) != hipSuccess) {
      hipFree(ws.status);
      ws.status = nullptr;
      return EXON_HIP_EDEVICE;
    }
  }
  if (ws.partial_capacity < words) {
    if (ws.partials) {
      hipStreamSynchronize(s);
      hipFree(ws.partials);
      ws.partials = nullptr;
      ws.partial_capacity = 0;
    }
    size_t cap = std::max<size_t>(words, 1 << 16);
    if (hipMalloc(&ws.partials, cap * 8) != hipSuccess) return EXON_HIP_ENOMEM;
    ws.partial_capacity = cap;
  }
  *out = ws;
  return EXON_HIP_OK;
}

// K4's tier-3 scratch (plans with more group ids than registers + the LDS table hold): grown on demand, kept with the
// stream's workspace.  Failure to allocate is not an error: the kernel then keeps the atomic form of tier 3.
static void ensure_tail_scratch(exon_hip_ctx* ctx, hipStream_t s, size_t records, Workspace* out) {
  if (records == 0) return;
  std::lock_guard<std::mutex> g(ctx->mu);
  Workspace& ws = ctx->workspaces[s];
  if (ws.tail_capacity < records) {
    if (ws.tail_rec_a) {
      hipStreamSynchronize(s);
      hipFree(ws.tail_rec_a);
      ws.tail_rec_a = ws.tail_rec_b = nullptr;
      ws.tail_capacity = 0;
    }
    // ONE allocation, rec_b = its second half: the direct partition (kernels.hip, k4_one_launch) uses both halves as one
    // pool of chunks
    if (hipMalloc((void**)&ws.tail_rec_a, records * 16) != hipSuccess) {
      (void)hipGetLastError();
      ws.tail_rec_a = ws.tail_rec_b = nullptr;
    } else {
      ws.tail_rec_b = ws.tail_rec_a + records;
      ws.tail_capacity = records;
    }
  }
  if (!ws.tail_u32 && hipMalloc((void**)&ws.tail_u32, exon::K4_TAIL_U32_WORDS * 4) != hipSuccess) {
    (void)hipGetLastError();
    ws.tail_u32 = nullptr;
  }
  *out = ws;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int exon_hip_abi_version(void) { return 5; }

int exon_hip_device_count(int* out) {
  if (!out) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_device_count: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *out = 0;
    return fail(nullptr, EXON_HIP_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *out = n;
  return EXON_HIP_OK;
}

int exon_hip_ctx_create(int device, exon_hip_ctx** out) {
  if (!out) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_ctx_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(nullptr, EXON_HIP_EDEVICE, "no HIP device visible (%s): the MI355X path has no CPU fallback",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(nullptr, EXON_HIP_EINVAL, "device %d out of range [0,%d)", device, n);
  exon_hip_ctx* ctx = new (std::nothrow) exon_hip_ctx();
  if (!ctx) return fail(nullptr, EXON_HIP_ENOMEM, "out of host memory");
  ctx->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->props, device)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess) {
    delete ctx;
    return fail(nullptr, EXON_HIP_EDEVICE, "ctx init: %s", hipGetErrorString(e));
  }
  ctx->cfg.compute_units = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
  ctx->cfg.blocks_per_cu = 8;
  if (const char* v = getenv("EXON_HIP_BLOCKS_PER_CU")) {
    int b = atoi(v);
    if (b >= 1 && b <= 32) ctx->cfg.blocks_per_cu = b;
  }
  exon_hip_prewarm_ctx(ctx);
  *out = ctx;
  return EXON_HIP_OK;
}

}  // extern "C"

void* exon_pool_alloc(exon_hip_ctx* ctx, size_t bytes) {
  if (bytes == 0) bytes = 16;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    auto it = ctx->pool_free.find(bytes);
    if (it != ctx->pool_free.end()) {
      void* p = it->second;
      ctx->pool_free.erase(it);
      ctx->pool_live[p] = bytes;
      return p;
    }
  }
  void* p = nullptr;
  hipSetDevice(ctx->device);
  if (hipMalloc(&p, bytes) != hipSuccess) {
    // make room: drop everything that is only cached, then try once more
    std::lock_guard<std::mutex> g(ctx->mu);
    for (auto& kv : ctx->pool_free) hipFree(kv.second);
    ctx->pool_free.clear();
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  }
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->pool_live[p] = bytes;
  return p;
}

void exon_pool_free(exon_hip_ctx* ctx, void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> g(ctx->mu);
  auto it = ctx->pool_live.find(p);
  if (it == ctx->pool_live.end()) {
    hipFree(p);
    return;
  }
  const size_t bytes = it->second;
  ctx->pool_live.erase(it);
  if (ctx->pool_free.size() >= 96) {  // bounded: beyond that, really free
    hipFree(p);
    return;
  }
  ctx->pool_free.emplace(bytes, p);
}

extern "C" {

int exon_hip_ctx_destroy(exon_hip_ctx* ctx) {
  if (!ctx) return EXON_HIP_OK;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  exon_hip_release_ctx_caches(ctx);
  for (auto& kv : ctx->pool_free) hipFree(kv.second);  // buffers still held by live parsers stay theirs
  for (auto& kv : ctx->workspaces) {
    if (kv.second.partials) hipFree(kv.second.partials);
    if (kv.second.status) hipFree(kv.second.status);
    if (kv.second.tail_rec_a) hipFree(kv.second.tail_rec_a);
    if (kv.second.tail_u32) hipFree(kv.second.tail_u32);
  }
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  if (ctx->stream) {
    exon_bgzf_forget_stream(ctx->stream);
    hipStreamDestroy(ctx->stream);
  }
  delete ctx;
  return EXON_HIP_OK;
}

int exon_hip_ctx_info(exon_hip_ctx* ctx, exon_hip_device_info* out) {
  if (!ctx || !out) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_ctx_info: NULL argument");
  memset(out, 0, sizeof *out);
  snprintf(out->name, sizeof out->name, "%s", ctx->props.name);
  snprintf(out->gcn_arch, sizeof out->gcn_arch, "%s", ctx->props.gcnArchName);
  out->compute_units = ctx->props.multiProcessorCount;
  out->wavefront_size = ctx->props.warpSize;
  out->hbm_bytes = (int64_t)ctx->props.totalGlobalMem;
  out->clock_khz = ctx->props.clockRate;
  return EXON_HIP_OK;
}

const char* exon_hip_last_error(const exon_hip_ctx* ctx) { return ctx ? ctx->error.c_str() : tls_error.c_str(); }

int exon_hip_malloc(exon_hip_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_malloc: NULL argument");
  *dptr = nullptr;
  if (bytes == 0) bytes = 16;
  hipSetDevice(ctx->device);
  hipError_t e = hipMalloc(dptr, bytes);
  if (e != hipSuccess) return fail(ctx, EXON_HIP_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return EXON_HIP_OK;
}
int exon_hip_free(exon_hip_ctx* ctx, void* dptr) {
  if (!ctx) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_free: NULL ctx");
  if (dptr) HIP_TRY(ctx, hipFree(dptr));
  return EXON_HIP_OK;
}
int exon_hip_memcpy_h2d(exon_hip_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
  if (!ctx) return fail(ctx, EXON_HIP_EINVAL, "NULL ctx");
  if (bytes == 0) return EXON_HIP_OK;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pick_stream(ctx, stream)));
  return EXON_HIP_OK;
}
int exon_hip_memcpy_d2h(exon_hip_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
  if (!ctx) return fail(ctx, EXON_HIP_EINVAL, "NULL ctx");
  if (bytes == 0) return EXON_HIP_OK;
  hipStream_t s = pick_stream(ctx, stream);
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return EXON_HIP_OK;
}
int exon_hip_memset(exon_hip_ctx* ctx, void* dst, int value, size_t bytes, void* stream) {
  if (!ctx) return fail(ctx, EXON_HIP_EINVAL, "NULL ctx");
  if (bytes == 0) return EXON_HIP_OK;
  HIP_TRY(ctx, hipMemsetAsync(dst, value, bytes, pick_stream(ctx, stream)));
  return EXON_HIP_OK;
}

// Synchronises `stream` and surfaces any device-side error word raised by kernels launched on it.
int exon_hip_sync(exon_hip_ctx* ctx, void* stream) {
  if (!ctx) return fail(ctx, EXON_HIP_EINVAL, "NULL ctx");
  hipStream_t s = pick_stream(ctx, stream);
  HIP_TRY(ctx, hipStreamSynchronize(s));
  int* d_status = nullptr;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    auto it = ctx->workspaces.find(s);
    if (it != ctx->workspaces.end()) d_status = it->second.status;
  }
  if (d_status) {
    int st = 0;
    HIP_TRY(ctx, hipMemcpy(&st, d_status, sizeof st, hipMemcpyDeviceToHost));
    if (st) {
      HIP_TRY(ctx, hipMemset(d_status, 0, sizeof st));
      return fail(ctx, EXON_HIP_EINVAL, "device status 0x%x:%s%s%s", st, (st & 2) ? " reference id out of range" : "",
                  (st & 4) ? " group id out of range" : "", (st & 8) ? " read longer than lmax" : "");
    }
  }
  return EXON_HIP_OK;
}

int exon_hip_timer_start(exon_hip_ctx* ctx, void* stream) {
  if (!ctx) return fail(ctx, EXON_HIP_EINVAL, "NULL ctx");
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, pick_stream(ctx, stream)));
  return EXON_HIP_OK;
}
int exon_hip_timer_stop_ms(exon_hip_ctx* ctx, void* stream, float* ms) {
  if (!ctx || !ms) return fail(ctx, EXON_HIP_EINVAL, "NULL argument");
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, pick_stream(ctx, stream)));
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
  HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return EXON_HIP_OK;
}

// Bare streaming read of `n_buffers` (1..4) device buffers, the first `bytes_each` bytes of each (whole 64 KiB tiles), walked
// in lock-step with the grid and the 16 B/lane non-temporal loads of the fused kernels: `reps` passes on `stream`, timed with a
// HIP event pair around them; *ms_per_pass = the average, *bytes_per_pass = what one pass read.
int exon_hip_read_probe(exon_hip_ctx* ctx, void* stream, const void* const* buffers, int32_t n_buffers, int64_t bytes_each,
                        int32_t reps, double* ms_per_pass, int64_t* bytes_per_pass) {
  if (!ctx || !buffers || !ms_per_pass) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_read_probe: NULL argument");
  if (n_buffers < 1 || n_buffers > 4 || reps < 1 || bytes_each < (64 << 10)) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_read_probe: 1..4 buffers of at least 64 KiB, reps >= 1");
  for (int i = 0; i < n_buffers; ++i)
    if (!buffers[i] || !aligned16(buffers[i])) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_read_probe: buffer %d is NULL or not 16-byte aligned", i);
  hipStream_t s = pick_stream(ctx, stream);
  unsigned* sink = static_cast<unsigned*>(exon_pool_alloc(ctx, (size_t)std::max(ctx->cfg.compute_units, 1) * 16 * 4));
  if (!sink) return fail(ctx, EXON_HIP_ENOMEM, "exon_hip_read_probe: sink");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = exon::launch_read_probe(s, ctx->cfg, buffers, n_buffers, bytes_each, sink);  // (untimed: code object load, clocks)
  if (e == hipSuccess) e = hipEventRecord(e0, s);
  for (int r = 0; r < reps && e == hipSuccess; ++r) e = exon::launch_read_probe(s, ctx->cfg, buffers, n_buffers, bytes_each, sink);
  if (e == hipSuccess) e = hipEventRecord(e1, s);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  exon_pool_free(ctx, sink);
  if (e != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "exon_hip_read_probe: %s", hipGetErrorString(e));
  *ms_per_pass = (double)ms / reps;
  if (bytes_per_pass) *bytes_per_pass = (bytes_each / (64 << 10)) * (64 << 10) * (int64_t)n_buffers;
  return EXON_HIP_OK;
}

// ---- operator launches ---------------------------------------------------------------------------
// `flags`: EXON_HIP_LAUNCH_*.  OVERWRITE makes the launch DEFINE the state (finalize writes instead of adding), so a
// query needs no zeroing pass; an empty input then still has to leave zeros behind.
static LaunchCfg cfg_for(const exon_hip_ctx* ctx, int flags) {
  LaunchCfg c = ctx->cfg;
  c.overwrite = (flags & EXON_HIP_LAUNCH_OVERWRITE) != 0;
  c.x_is_int = (flags & EXON_LAUNCH_X_INT32) != 0;
  c.y_is_int = (flags & EXON_LAUNCH_Y_INT32) != 0;
  return c;
}
static int empty_input(exon_hip_ctx* ctx, hipStream_t s, int flags, void* state, size_t bytes) {
  if ((flags & EXON_HIP_LAUNCH_OVERWRITE) && bytes) HIP_TRY(ctx, hipMemsetAsync(state, 0, bytes, s));
  return EXON_HIP_OK;
}

static int check_col(exon_hip_ctx* ctx, const char* what, const exon_hip_column* c, int64_t n, bool need_offsets) {
  if (!c) return fail(ctx, EXON_HIP_EINVAL, "%s: column is NULL", what);
  if (n > 0 && !c->values) return fail(ctx, EXON_HIP_EINVAL, "%s: values is NULL", what);
  if (c->length < n) return fail(ctx, EXON_HIP_EINVAL, "%s: length %lld < n %lld", what, (long long)c->length, (long long)n);
  if (!aligned16(c->values)) return fail(ctx, EXON_HIP_EINVAL, "%s: values must be 16-byte aligned", what);
  if (need_offsets && n > 0 && !c->offsets) return fail(ctx, EXON_HIP_EINVAL, "%s: offsets is NULL", what);
  return EXON_HIP_OK;
}

}  // extern "C"

int exon_op_region_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* chrom_id, const exon_hip_column* pos,
                         int64_t n, int32_t region_chrom_id, int64_t start, int64_t end, int64_t* d_count, int flags) {
  if (!ctx || !d_count) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_region_count: NULL argument");
  if (n < 0) return fail(ctx, EXON_HIP_EINVAL, "n < 0");
  int rc;
  if ((rc = check_col(ctx, "chrom_id", chrom_id, n, false)) || (rc = check_col(ctx, "pos", pos, n, false))) return rc;
  hipStream_t s = pick_stream(ctx, stream);
  if (n == 0) return empty_input(ctx, s, flags, d_count, 8);
  Workspace ws;
  if ((rc = get_workspace(ctx, s, exon::k2_partial_words(ctx->cfg), &ws))) return fail(ctx, rc, "workspace allocation failed");
  HIP_TRY(ctx, exon::launch_region_count(s, cfg_for(ctx, flags), ws, (const int32_t*)chrom_id->values, chrom_id->validity,
                                         (const int64_t*)pos->values, pos->validity, n, region_chrom_id, start, end,
                                         d_count));
  return EXON_HIP_OK;
}

int exon_op_overlap_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* ref_id, const exon_hip_column* start,
                          const exon_hip_column* end, int64_t n, int32_t region_ref_id, int64_t region_start,
                          int64_t region_end, int64_t* d_count, int flags, bool strict) {
  if (!ctx || !d_count) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_overlap_count: NULL argument");
  if (n < 0) return fail(ctx, EXON_HIP_EINVAL, "n < 0");
  int rc;
  if ((rc = check_col(ctx, "ref_id", ref_id, n, false)) || (rc = check_col(ctx, "start", start, n, false)) ||
      (rc = check_col(ctx, "end", end, n, false)))
    return rc;
  hipStream_t s = pick_stream(ctx, stream);
  if (n == 0) return empty_input(ctx, s, flags, d_count, 8);
  Workspace ws;
  if ((rc = get_workspace(ctx, s, exon::k2_partial_words(ctx->cfg), &ws))) return fail(ctx, rc, "workspace allocation failed");
  HIP_TRY(ctx, exon::launch_overlap_count(s, cfg_for(ctx, flags), ws, (const int32_t*)ref_id->values, ref_id->validity,
                                          (const int64_t*)start->values, start->validity, (const int64_t*)end->values,
                                          end->validity, n, region_ref_id, region_start, region_end, d_count, strict));
  return EXON_HIP_OK;
}

int exon_op_flag_mapq_group_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* flag,
                                  const exon_hip_column* mapq, const exon_hip_column* ref_id, int64_t n,
                                  int32_t flag_mask, int32_t flag_value, int32_t mapq_min, int32_t n_refs,
                                  int64_t* d_counts, int flags) {
  if (!ctx || !d_counts) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_flag_mapq_group_count: NULL argument");
  if (n < 0) return fail(ctx, EXON_HIP_EINVAL, "n < 0");
  if (n_refs < 0 || n_refs >= EXON_HIP_MAX_REFERENCES)
    return fail(ctx, EXON_HIP_EUNSUPPORTED, "n_refs %d outside [0, %d)", n_refs, EXON_HIP_MAX_REFERENCES);
  int rc;
  if ((rc = check_col(ctx, "flag", flag, n, false)) || (rc = check_col(ctx, "mapq", mapq, n, false)) ||
      (rc = check_col(ctx, "ref_id", ref_id, n, false)))
    return rc;
  hipStream_t s = pick_stream(ctx, stream);
  if (n == 0) return empty_input(ctx, s, flags, d_counts, (size_t)(n_refs + 1) * 8);
  Workspace ws;
  if ((rc = get_workspace(ctx, s, exon::k3_partial_words(ctx->cfg, n_refs), &ws))) return fail(ctx, rc, "workspace allocation failed");
  HIP_TRY(ctx, exon::launch_flag_mapq_group_count(s, cfg_for(ctx, flags), ws, (const int32_t*)flag->values, flag->validity,
                                                  (const uint8_t*)mapq->values, mapq->validity,
                                                  (const int32_t*)ref_id->values, ref_id->validity, n, flag_mask,
                                                  flag_value, mapq_min, n_refs, d_counts));
  return EXON_HIP_OK;
}

int exon_op_cmp_avg_by_group(exon_hip_ctx* ctx, void* stream, const exon_hip_column* x, const exon_hip_column* y,
                             const exon_hip_column* group_id, int64_t n, double threshold, int32_t cmp_op,
                             int32_t n_groups, int64_t* d_counts, double* d_sums, int flags) {
  if (!ctx || !d_counts || !d_sums) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_cmp_avg_by_group: NULL argument");
  if (n < 0) return fail(ctx, EXON_HIP_EINVAL, "n < 0");
  if (cmp_op < EXON_HIP_GT || cmp_op > EXON_HIP_NE) return fail(ctx, EXON_HIP_EINVAL, "bad cmp_op %d", cmp_op);
  if (n_groups < 1) return fail(ctx, EXON_HIP_EINVAL, "n_groups must be >= 1");
  if (n_groups > EXON_HIP_MAX_GROUPS_GLOBAL)
    return fail(ctx, EXON_HIP_EUNSUPPORTED, "n_groups %d > %d", n_groups, EXON_HIP_MAX_GROUPS_GLOBAL);
  if (group_id && group_id->validity)
    return fail(ctx, EXON_HIP_EUNSUPPORTED, "nullable group ids: encode NULL as its own dictionary id");
  int rc;
  if ((rc = check_col(ctx, "x", x, n, false)) || (rc = check_col(ctx, "y", y, n, false)) ||
      (rc = check_col(ctx, "group_id", group_id, n, false)))
    return rc;
  hipStream_t s = pick_stream(ctx, stream);
  if (n == 0) {
    if ((rc = empty_input(ctx, s, flags, d_counts, (size_t)n_groups * 16))) return rc;
    return empty_input(ctx, s, flags, d_sums, (size_t)n_groups * 8);
  }
  Workspace ws;
  if ((rc = get_workspace(ctx, s, exon::k4_partial_words(ctx->cfg, n_groups), &ws))) return fail(ctx, rc, "workspace allocation failed");
  ensure_tail_scratch(ctx, s, exon::k4_tail_records(n, n_groups), &ws);
  HIP_TRY(ctx, exon::launch_cmp_avg_by_group(s, cfg_for(ctx, flags), ws, (const float*)x->values, x->validity,
                                             (const float*)y->values, y->validity, (const int32_t*)group_id->values,
                                             n, threshold, cmp_op, n_groups, d_counts, d_sums));
  return EXON_HIP_OK;
}

int exon_op_qual_pos_hist(exon_hip_ctx* ctx, void* stream, const exon_hip_column* q, int64_t n_reads, int32_t lmax,
                          int64_t* d_hist, int flags) {
  if (!q) return fail(ctx, EXON_HIP_EINVAL, "quality_scores column is NULL");
  return exon_op_qual_pos_hist_chunks(ctx, stream, q, 1, 1, &n_reads, lmax, d_hist, flags);
}

int exon_op_qual_pos_hist_chunks(exon_hip_ctx* ctx, void* stream, const exon_hip_column* q, int stride, int32_t n_chunks,
                                 const int64_t* n_reads, int32_t lmax, int64_t* d_hist, int flags) {
  if (!ctx || !d_hist) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_qual_pos_hist: NULL argument");
  if (lmax < 1 || lmax > (1 << 20)) return fail(ctx, EXON_HIP_EINVAL, "lmax %d out of range", lmax);
  if (n_chunks < 0) return fail(ctx, EXON_HIP_EINVAL, "n_chunks < 0");
  if (n_chunks && (!q || !n_reads)) return fail(ctx, EXON_HIP_EINVAL, "quality_scores column is NULL");
  for (int c = 0; c < n_chunks; ++c) {
    const exon_hip_column& col = q[(size_t)c * stride];
    if (n_reads[c] < 0) return fail(ctx, EXON_HIP_EINVAL, "n_reads < 0");
    if (col.validity) return fail(ctx, EXON_HIP_EUNSUPPORTED, "nullable quality_scores (the reference column is non-null)");
    if (n_reads[c] == 0) continue;
    if (!col.offsets || !col.values) return fail(ctx, EXON_HIP_EINVAL, "quality_scores: offsets/values NULL");
    if (col.length < n_reads[c]) return fail(ctx, EXON_HIP_EINVAL, "quality_scores: length < n_reads");
  }
  hipStream_t s = pick_stream(ctx, stream);
  Workspace ws;
  int rc;
  bool launched = false;
  exon::K5Chunks ch;
  ch.count = 0;
  int64_t reads = 0;
  auto flush = [&]() -> int {
    if (!ch.count) return EXON_HIP_OK;
    if (!launched && (rc = get_workspace(ctx, s, exon::k5_partial_words(ctx->cfg, lmax), &ws))) return fail(ctx, rc, "workspace allocation failed");
    HIP_TRY(ctx, exon::launch_qual_pos_hist_chunks(s, cfg_for(ctx, launched ? EXON_HIP_LAUNCH_ACCUMULATE : flags), ws, ch, lmax, d_hist));
    launched = true;
    ch.count = 0;
    reads = 0;
    return EXON_HIP_OK;
  };
  // a workgroup's u32 counters see at most (reads of a launch) / (workgroups) + a wave's share increments per bin
  const int64_t reads_cap = (int64_t)ctx->cfg.compute_units << 30;
  for (int c = 0; c < n_chunks; ++c) {
    if (n_reads[c] == 0) continue;
    if (ch.count == exon::K5_MAX_CHUNKS || (ch.count && reads + n_reads[c] > reads_cap))
      if ((rc = flush())) return rc;
    const exon_hip_column& col = q[(size_t)c * stride];
    ch.off[ch.count] = col.offsets;
    ch.ends[ch.count] = col.offsets + 1;
    ch.bytes[ch.count] = (const uint8_t*)col.values;
    ch.n[ch.count] = n_reads[c];
    ++ch.count;
    reads += n_reads[c];
  }
  if ((rc = flush())) return rc;
  if (!launched) return empty_input(ctx, s, flags, d_hist, (size_t)lmax * 256 * 8);
  return EXON_HIP_OK;
}

extern "C" {

int exon_hip_region_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* chrom_id, const exon_hip_column* pos,
                          int64_t n, int32_t region_chrom_id, int64_t start, int64_t end, int64_t* d_count) {
  return exon_op_region_count(ctx, stream, chrom_id, pos, n, region_chrom_id, start, end, d_count, EXON_HIP_LAUNCH_ACCUMULATE);
}
int exon_hip_overlap_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* ref_id, const exon_hip_column* start,
                           const exon_hip_column* end, int64_t n, int32_t region_ref_id, int64_t region_start,
                           int64_t region_end, int64_t* d_count) {
  return exon_op_overlap_count(ctx, stream, ref_id, start, end, n, region_ref_id, region_start, region_end, d_count,
                               EXON_HIP_LAUNCH_ACCUMULATE, false);
}
int exon_hip_within_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* ref_id, const exon_hip_column* start,
                          const exon_hip_column* end, int64_t n, int32_t region_ref_id, int64_t after, int64_t before,
                          int64_t* d_count) {
  return exon_op_overlap_count(ctx, stream, ref_id, start, end, n, region_ref_id, after, before, d_count, EXON_HIP_LAUNCH_ACCUMULATE,
                               true);
}
int exon_hip_flag_mapq_group_count(exon_hip_ctx* ctx, void* stream, const exon_hip_column* flag,
                                   const exon_hip_column* mapq, const exon_hip_column* ref_id, int64_t n,
                                   int32_t flag_mask, int32_t flag_value, int32_t mapq_min, int32_t n_refs,
                                   int64_t* d_counts) {
  return exon_op_flag_mapq_group_count(ctx, stream, flag, mapq, ref_id, n, flag_mask, flag_value, mapq_min, n_refs, d_counts,
                                       EXON_HIP_LAUNCH_ACCUMULATE);
}
int exon_hip_cmp_avg_by_group(exon_hip_ctx* ctx, void* stream, const exon_hip_column* x, const exon_hip_column* y,
                              const exon_hip_column* group_id, int64_t n, double threshold, int32_t cmp_op,
                              int32_t n_groups, int64_t* d_counts, double* d_sums) {
  return exon_op_cmp_avg_by_group(ctx, stream, x, y, group_id, n, threshold, cmp_op, n_groups, d_counts, d_sums,
                                  EXON_HIP_LAUNCH_ACCUMULATE);
}
int exon_hip_qual_pos_hist(exon_hip_ctx* ctx, void* stream, const exon_hip_column* q, int64_t n_reads, int32_t lmax,
                           int64_t* d_hist) {
  return exon_op_qual_pos_hist(ctx, stream, q, n_reads, lmax, d_hist, EXON_HIP_LAUNCH_ACCUMULATE);
}

// Fixed-order fold of `world` gathered packed states (AggregateExec(Final) across GPUs after one all-gather).
int exon_hip_fold_states(exon_hip_ctx* ctx, void* stream, const void* d_gathered, int32_t world, int64_t n_i64,
                         int64_t n_f64, void* d_out) {
  if (!ctx || !d_gathered || !d_out) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_fold_states: NULL argument");
  if (world < 1 || n_i64 < 0 || n_f64 < 0 || n_i64 + n_f64 < 1 || n_i64 + n_f64 > INT32_MAX)
    return fail(ctx, EXON_HIP_EINVAL, "exon_hip_fold_states: bad sizes (world %d, %lld + %lld words)", world, (long long)n_i64, (long long)n_f64);
  HIP_TRY(ctx, exon::launch_fold_states(pick_stream(ctx, stream), d_gathered, world, n_i64, n_f64, d_out));
  return EXON_HIP_OK;
}

int exon_hip_qual_pos_hist_chunks(exon_hip_ctx* ctx, void* stream, const exon_hip_column* chunks, int32_t n_chunks,
                                  const int64_t* n_reads, int32_t lmax, int64_t* d_hist) {
  return exon_op_qual_pos_hist_chunks(ctx, stream, chunks, 1, n_chunks, n_reads, lmax, d_hist, EXON_HIP_LAUNCH_ACCUMULATE);
}

int exon_hip_qual_pos_hist_views(exon_hip_ctx* ctx, void* stream, const uint8_t* d_bytes, const int32_t* d_starts,
                                 const int32_t* d_ends, int64_t n_reads, int32_t lmax, int64_t* d_hist) {
  if (!ctx || !d_hist) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_qual_pos_hist_views: NULL argument");
  if (n_reads < 0) return fail(ctx, EXON_HIP_EINVAL, "n_reads < 0");
  if (lmax < 1 || lmax > (1 << 20)) return fail(ctx, EXON_HIP_EINVAL, "lmax %d out of range", lmax);
  if (n_reads == 0) return EXON_HIP_OK;
  if (!d_bytes || !d_starts || !d_ends) return fail(ctx, EXON_HIP_EINVAL, "views: bytes/starts/ends NULL");
  hipStream_t s = pick_stream(ctx, stream);
  Workspace ws;
  int rc;
  if ((rc = get_workspace(ctx, s, exon::k5_partial_words(ctx->cfg, lmax), &ws))) return fail(ctx, rc, "workspace allocation failed");
  HIP_TRY(ctx, exon::launch_qual_pos_hist_views(s, ctx->cfg, ws, d_starts, d_ends, d_bytes, n_reads, lmax, d_hist));
  return EXON_HIP_OK;
}

// ---- synthetic inputs ----------------------------------------------------------------------------
int exon_hip_gen_c2(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t n_total, int64_t lo, int64_t hi,
                    int32_t* d_chrom_id, int64_t* d_pos) {
  if (!ctx || !d_chrom_id || !d_pos) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gen_c2: NULL argument");
  if (lo < 0 || hi < lo || hi > n_total) return fail(ctx, EXON_HIP_EINVAL, "bad row range");
  HIP_TRY(ctx, exon::launch_gen_c2(pick_stream(ctx, stream), seed, n_total, lo, hi, d_chrom_id, d_pos));
  return EXON_HIP_OK;
}
int exon_hip_gen_c3(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, int32_t* d_flag,
                    uint8_t* d_mapq, uint8_t* d_mapq_valid, int32_t* d_ref_id, uint8_t* d_ref_valid) {
  if (!ctx || !d_flag || !d_mapq || !d_mapq_valid || !d_ref_id || !d_ref_valid)
    return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gen_c3: NULL argument");
  if (lo < 0 || hi < lo) return fail(ctx, EXON_HIP_EINVAL, "bad row range");
  HIP_TRY(ctx, exon::launch_gen_c3(pick_stream(ctx, stream), seed, lo, hi, d_flag, d_mapq, d_mapq_valid, d_ref_id, d_ref_valid));
  return EXON_HIP_OK;
}
int exon_hip_gen_c4(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, float* d_af,
                    uint8_t* d_af_valid, float* d_qual, uint8_t* d_qual_valid, int32_t* d_filter_id) {
  if (!ctx || !d_af || !d_af_valid || !d_qual || !d_qual_valid || !d_filter_id)
    return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gen_c4: NULL argument");
  if (lo < 0 || hi < lo) return fail(ctx, EXON_HIP_EINVAL, "bad row range");
  HIP_TRY(ctx, exon::launch_gen_c4(pick_stream(ctx, stream), seed, lo, hi, d_af, d_af_valid, d_qual, d_qual_valid, d_filter_id));
  return EXON_HIP_OK;
}
int exon_hip_gen_c6(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, int32_t* d_ref_id,
                    uint8_t* d_ref_valid, int64_t* d_start, int64_t* d_end, uint8_t* d_pos_valid) {
  if (!ctx || !d_ref_id || !d_ref_valid || !d_start || !d_end || !d_pos_valid) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gen_c6: NULL argument");
  if (lo < 0 || hi < lo) return fail(ctx, EXON_HIP_EINVAL, "bad row range");
  HIP_TRY(ctx, exon::launch_gen_c6(pick_stream(ctx, stream), seed, lo, hi, d_ref_id, d_ref_valid, d_start, d_end, d_pos_valid));
  return EXON_HIP_OK;
}
int exon_hip_gen_c5(exon_hip_ctx* ctx, void* stream, uint64_t seed, int64_t lo, int64_t hi, int32_t read_len,
                    int32_t* d_offsets, uint8_t* d_bytes) {
  if (!ctx || !d_offsets || !d_bytes) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gen_c5: NULL argument");
  if (lo < 0 || hi < lo || read_len < 1) return fail(ctx, EXON_HIP_EINVAL, "bad arguments");
  if ((hi - lo) * (int64_t)read_len > INT32_MAX) return fail(ctx, EXON_HIP_EINVAL, "batch exceeds int32 offsets");
  HIP_TRY(ctx, exon::launch_gen_c5(pick_stream(ctx, stream), seed, lo, hi, read_len, d_offsets, d_bytes));
  return EXON_HIP_OK;
}

// ---- host-side planning helpers ------------------------------------------------------------------
int exon_hip_parse_region(const char* region, char* name_out, size_t name_cap, int64_t* start, int64_t* end) {
  if (!region || !name_out || !start || !end) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_parse_region: NULL argument");
  exon::Region r;
  std::string err;
  if (!exon::parse_region(region, &r, &err)) return fail(nullptr, EXON_HIP_EINVAL, "invalid region '%s': %s", region, err.c_str());
  if (r.name.size() + 1 > name_cap) return fail(nullptr, EXON_HIP_EINVAL, "region name buffer too small");
  memcpy(name_out, r.name.c_str(), r.name.size() + 1);
  *start = r.start;
  *end = r.end;
  return EXON_HIP_OK;
}

int exon_hip_regroup_files_by_size(const int64_t* sizes, int32_t n_files, int32_t target_groups, int32_t* group_of) {
  if (n_files < 0 || (n_files > 0 && (!sizes || !group_of)))
    return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_regroup_files_by_size: bad argument");
  if (target_groups < 1) return fail(nullptr, EXON_HIP_EINVAL, "target_groups must be >= 1");
  std::vector<int64_t> sz(sizes, sizes + n_files);
  std::vector<int32_t> g = exon::regroup_files_by_size(sz, target_groups);
  int32_t ng = 0;
  for (int32_t i = 0; i < n_files; ++i) {
    group_of[i] = g[i];
    ng = std::max(ng, g[i] + 1);
  }
  return ng;
}

}  // extern "C"
