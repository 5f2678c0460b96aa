// bam_parse.hip -- BAM record splitting + field extraction ON THE GPU: inflated BAM bytes in HBM -> device-layout
// columns (flag, mapping_quality, reference id, start, end) in HBM.
//
// Reference semantics: BAMArrayBuilder::append (exon-bam/src/array_builder.rs:102-218) restricted to the columns of
// the device layout -- flag = raw u16 bits as Int32 (:114-117), reference = refID (NULL if -1, :118-127), start =
// pos + 1 (NULL if pos < 0), end = start + (sum of M/D/N/=/X lengths) - 1, mapping_quality NULL when 255 (:136-143);
// record layout: SAM specification section 4.2.
//
// Records are found by the parallel chain walk of chain_walk.h (64 KiB segments, guessed starts proven by induction);
//   k_bam_extract  one thread per record: fixed fields, CIGAR walk for the reference length, columns + validity bits.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

#include "chain_walk.h"
#include "internal.h"

namespace {

using chain::ld16;
using chain::ld32;
using chain::SEG;
using chain::SegInfo;

// BAM record: int32 block_size, then block_size bytes (SAM spec 4.2)
struct BamFormat {
  static constexpr uint32_t MIN_HEADER = 36, MIN_RECORD = 36, LEN_BYTES = 4;
  int32_t n_ref;
  __device__ __forceinline__ uint32_t record_bytes(const uint8_t* d, uint32_t r) const {
    const uint32_t bs = ld32(d + r);
    return bs < 32 || bs > (1u << 28) ? 0u : 4u + bs;
  }
  // Is there a well-formed record header at offset r?  Only what is needed to tell record starts from arbitrary bytes;
  // the walk itself trusts block_size alone, as the host decoder does.
  __device__ bool plausible(const uint8_t* d, uint32_t n, uint32_t r) const {
    if ((uint64_t)r + 36 > n) return false;
    const uint32_t bs = ld32(d + r);
    if (bs < 32 || bs > (1u << 24)) return false;
    const int32_t ref = (int32_t)ld32(d + r + 4), pos = (int32_t)ld32(d + r + 8);
    if (ref < -1 || ref >= n_ref || pos < -1) return false;
    const uint32_t l_name = d[r + 12], n_cigar = ld16(d + r + 16);
    const int32_t l_seq = (int32_t)ld32(d + r + 20);
    const int32_t mref = (int32_t)ld32(d + r + 24), mpos = (int32_t)ld32(d + r + 28);
    if (l_name == 0 || l_seq < 0 || mref < -1 || mref >= n_ref || mpos < -1) return false;
    const uint64_t need = 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > bs) return false;
    if ((uint64_t)r + 36 + l_name <= n && d[r + 36 + l_name - 1] != 0) return false;  // read name is NUL-terminated
    return true;
  }
};
constexpr uint32_t SEG_CAP = chain::seg_cap<BamFormat>();

struct BamOut {
  int32_t* flag;
  uint8_t* mapq;
  uint32_t* mapq_valid;  // validity bitmaps as 32-bit words
  int32_t* ref_id;
  uint32_t* ref_valid;
  int64_t* start;
  int64_t* end;
  uint32_t* pos_valid;
  uint32_t* rec_of_row;  // byte offset of every row's record (the string columns are built from it: text_columns.hip)
};

__global__ __launch_bounds__(256) void k_bam_extract(const uint8_t* __restrict__ d, const SegInfo* __restrict__ seg,
                                                     const uint32_t* __restrict__ base, const uint32_t* __restrict__ rec_off, BamOut o,
                                                     unsigned* __restrict__ scalars) {
  const uint32_t s = blockIdx.x;
  if (scalars[1] != 0) return;  // the segmentation was not proven: nothing here can be trusted
  const uint32_t cnt = seg[s].count, row0 = base[s];
  const uint32_t* offs = rec_off + (size_t)s * SEG_CAP;
  // validity bits: ONE atomicOr per 32-row word and wave (a wave's 64 consecutive rows touch at most three words; words at segment
  // seams are shared with the neighbouring workgroup) instead of one per row and column -- three global atomics per record were
  // most of this kernel's time
  const uint32_t lane = threadIdx.x & 63u;
  auto publish = [&](uint32_t* bitmap, bool bit, uint32_t wave_row) {
    const unsigned long long b = __ballot(bit);
    if (b == 0) return;
    const uint32_t sh = wave_row & 31u;
    const unsigned long long x = b << sh;
    const uint32_t part = lane == 0 ? (uint32_t)x : lane == 1 ? (uint32_t)(x >> 32) : (sh ? (uint32_t)(b >> (64u - sh)) : 0u);
    if (lane < 3 && part) atomicOr(&bitmap[(wave_row >> 5) + lane], part);
  };
  for (uint32_t k = threadIdx.x; k < cnt; k += 256) {
    const uint32_t r = offs[k], row = row0 + k;
    const uint32_t bs = ld32(d + r);
    const int32_t ref = (int32_t)ld32(d + r + 4), pos = (int32_t)ld32(d + r + 8);
    const uint32_t l_name = d[r + 12], mapq = d[r + 13], n_cigar = ld16(d + r + 16), flag = ld16(d + r + 18);
    const uint32_t co = 32 + l_name;
    const bool ok = co + 4 * n_cigar <= bs;  // else "corrupt BAM cigar" on the host
    if (!ok) atomicAdd(&scalars[1], 1u);
    int64_t ref_len = 0;
    for (uint32_t c = 0; ok && c < n_cigar; ++c) {
      const uint32_t op = ld32(d + r + 4 + co + 4 * c);
      const uint32_t code = op & 0xF;
      if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) ref_len += op >> 4;
    }
    const bool pv = pos >= 0;
    if (ok) {
      o.flag[row] = (int32_t)flag;
      o.mapq[row] = (uint8_t)mapq;
      o.ref_id[row] = ref < 0 ? -1 : ref;
      o.start[row] = pv ? (int64_t)pos + 1 : 0;
      o.end[row] = pv ? (int64_t)pos + ref_len : 0;
      o.rec_of_row[row] = r;
    }
    const uint32_t wave_row = row - lane;  // the row of the wave's lane 0 (rows are consecutive across a wave's lanes)
    publish(o.mapq_valid, ok && mapq != 255, wave_row);
    publish(o.ref_valid, ok && ref >= 0, wave_row);
    publish(o.pos_valid, ok && pv, wave_row);
  }
}

}  // namespace

struct exon_hip_bam_parser {
  exon_hip_ctx* ctx = nullptr;
  int32_t n_ref = 0;
  int64_t max_bytes = 0, max_rows = 0;
  uint32_t max_seg = 0;
  SegInfo* d_seg = nullptr;
  uint32_t *d_base = nullptr, *d_rec_off = nullptr, *d_scalars = nullptr;
  void* bufs[9] = {nullptr};
  BamOut out{};
  unsigned* h_scalars = nullptr;
};

extern "C" {

int exon_hip_bam_parser_create(exon_hip_ctx* ctx, int32_t n_references, int64_t max_bytes, exon_hip_bam_parser** outp) {
  if (!ctx || !outp || max_bytes < 64 || n_references < 0) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_bam_parser_create: bad argument");
  if (max_bytes > 0xF0000000LL) return fail(ctx, EXON_HIP_EINVAL, "slab size must stay below 4 GiB (32-bit record offsets)");
  *outp = nullptr;
  exon_hip_bam_parser* p = new (std::nothrow) exon_hip_bam_parser();
  if (!p) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  p->ctx = ctx;
  p->n_ref = n_references;
  p->max_bytes = max_bytes;
  p->max_seg = (uint32_t)((max_bytes + SEG - 1) / SEG);
  p->max_rows = max_bytes / 36 + 1;
  hipSetDevice(ctx->device);
  hipError_t e = hipSuccess;
  auto dalloc = [&](void** ptr, size_t bytes) {
    if (e == hipSuccess && !(*ptr = exon_pool_alloc(ctx, bytes))) e = hipErrorOutOfMemory;
  };
  const size_t r = (size_t)p->max_rows, w = (r + 31) / 32 * 4 + 64;
  dalloc((void**)&p->d_seg, (size_t)p->max_seg * sizeof(SegInfo));
  dalloc((void**)&p->d_base, (size_t)p->max_seg * 4);
  dalloc((void**)&p->d_rec_off, (size_t)p->max_seg * SEG_CAP * 4);
  dalloc((void**)&p->d_scalars, 16);
  dalloc(&p->bufs[0], r * 4);
  dalloc(&p->bufs[1], r + 64);
  dalloc(&p->bufs[2], w);
  dalloc(&p->bufs[3], r * 4);
  dalloc(&p->bufs[4], w);
  dalloc(&p->bufs[5], r * 8);
  dalloc(&p->bufs[6], r * 8);
  dalloc(&p->bufs[7], w);
  dalloc(&p->bufs[8], r * 4);
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_scalars, 16);
  if (e != hipSuccess) {
    const std::string msg = hipGetErrorString(e);
    exon_hip_bam_parser_destroy(p);
    return fail(ctx, EXON_HIP_ENOMEM, "bam parser allocation: %s", msg.c_str());
  }
  p->out = BamOut{(int32_t*)p->bufs[0], (uint8_t*)p->bufs[1], (uint32_t*)p->bufs[2], (int32_t*)p->bufs[3],
                  (uint32_t*)p->bufs[4], (int64_t*)p->bufs[5], (int64_t*)p->bufs[6], (uint32_t*)p->bufs[7], (uint32_t*)p->bufs[8]};
  *outp = p;
  return EXON_HIP_OK;
}

int exon_hip_bam_parser_destroy(exon_hip_bam_parser* p) {
  if (!p) return EXON_HIP_OK;
  for (void* b : p->bufs)
    if (b) exon_pool_free(p->ctx, b);
  if (p->d_seg) exon_pool_free(p->ctx, p->d_seg);
  if (p->d_base) exon_pool_free(p->ctx, p->d_base);
  if (p->d_rec_off) exon_pool_free(p->ctx, p->d_rec_off);
  if (p->d_scalars) exon_pool_free(p->ctx, p->d_scalars);
  if (p->h_scalars) hipHostFree(p->h_scalars);
  delete p;
  return EXON_HIP_OK;
}

int exon_hip_bam_parser_parse(exon_hip_bam_parser* p, void* stream, const uint8_t* d_data, int64_t n_bytes, exon_hip_bam_columns* cols) {
  if (!p || !cols || (n_bytes > 0 && !d_data)) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_bam_parser_parse: NULL argument");
  exon_hip_ctx* ctx = p->ctx;
  if (n_bytes > p->max_bytes) return fail(ctx, EXON_HIP_EINVAL, "slab of %lld bytes exceeds the parser's %lld", (long long)n_bytes, (long long)p->max_bytes);
  memset(cols, 0, sizeof *cols);
  if (n_bytes == 0) return EXON_HIP_OK;
  hipStream_t s = pick_stream(ctx, stream);
  const uint32_t n = (uint32_t)n_bytes, n_seg = (n + SEG - 1) / SEG;
  // rows of this slab <= n / 36: the validity words they can touch are cleared by k_chain_check, the scalars by k_chain_walk
  chain::ZeroList zl{};
  zl.p[0] = p->out.mapq_valid;
  zl.p[1] = p->out.ref_valid;
  zl.p[2] = p->out.pos_valid;
  zl.n = 3;
  zl.words = (uint32_t)(((size_t)n / 36 + 1 + 31) / 32 + 1);
  hipLaunchKernelGGL(chain::k_chain_walk<BamFormat>, dim3(n_seg), dim3(64), 0, s, d_data, n, BamFormat{p->n_ref}, p->d_seg, p->d_rec_off,
                     p->d_scalars);
  hipLaunchKernelGGL(chain::k_chain_check<0>, dim3(1), dim3(256), 0, s, p->d_seg, n_seg, p->d_base, p->d_scalars, zl);
  hipLaunchKernelGGL(k_bam_extract, dim3(n_seg), dim3(256), 0, s, d_data, p->d_seg, p->d_base, p->d_rec_off, p->out, p->d_scalars);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(p->h_scalars, p->d_scalars, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  if (getenv("EXON_HIP_CHAIN_TRACE")) fprintf(stderr, "[exon-hip chain] %u segments, %u rows, plain proof %u\n", n_seg, p->h_scalars[0], p->h_scalars[3]);
  cols->n_rows = p->h_scalars[0];
  cols->n_undecided = p->h_scalars[1];
  cols->consumed_bytes = p->h_scalars[2];
  cols->flag = p->out.flag;
  cols->mapq = p->out.mapq;
  cols->mapq_valid = (uint8_t*)p->out.mapq_valid;
  cols->ref_id = p->out.ref_id;
  cols->ref_valid = (uint8_t*)p->out.ref_valid;
  cols->start = p->out.start;
  cols->end = p->out.end;
  cols->pos_valid = (uint8_t*)p->out.pos_valid;
  return EXON_HIP_OK;
}

}  // extern "C"

const uint32_t* exon_hip_bam_parser_row_records(exon_hip_bam_parser* p) { return p ? p->out.rec_of_row : nullptr; }
