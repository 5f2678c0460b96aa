// internal.h -- shared between capi.cpp (ctx + operator launches) and stream.cpp (plan/stream layer).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>

#include "../../include/exon_hip.h"
#include "kernels.h"

struct exon_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;  // used when the caller passes stream == NULL
  exon::LaunchCfg cfg;
  std::mutex mu;
  std::map<hipStream_t, exon::Workspace> workspaces;
  std::string error;
  hipDeviceProp_t props;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // device buffers of the GPU-side parsers are recycled between scans (exon_pool_*): their sizes repeat exactly and
  // hipMalloc / hipFree of gigabytes costs tens of milliseconds per file
  std::multimap<size_t, void*> pool_free;
  std::map<void*, size_t> pool_live;
};

// records the message on the ctx (and the calling thread) and returns `code`
int fail(exon_hip_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
const std::string& exon_hip_tls_error();

#define HIP_TRY(ctx, expr)                                                                             \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_));  \
  } while (0)

inline hipStream_t pick_stream(exon_hip_ctx* ctx, void* s) { return s ? (hipStream_t)s : ctx->stream; }

// Workspace of `words` 8-byte partial words for launches on stream `s` (grown on demand).
int get_workspace(exon_hip_ctx* ctx, hipStream_t s, size_t words, exon::Workspace* out);

// inflate.hip: enqueue the BGZF inflate (+ CRC-32) kernels over device-resident block tables
hipError_t exon_bgzf_inflate_launch(hipStream_t s, const uint8_t* d_comp, const exon_hip_bgzf_block* d_blocks, int n_blocks,
                                    uint8_t* d_out, int* d_status, bool verify_crc);
const char* exon_bgzf_status_name(int code);

// scan.cpp: slab buffers kept per ctx between scans
void exon_hip_release_ctx_caches(exon_hip_ctx* ctx);

// capi.cpp: size-keyed recycling of device buffers (released by exon_hip_ctx_destroy)
void* exon_pool_alloc(exon_hip_ctx* ctx, size_t bytes);
void exon_pool_free(exon_hip_ctx* ctx, void* p);
