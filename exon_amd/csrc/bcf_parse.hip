// bcf_parse.hip -- BCF2 record splitting + field extraction ON THE GPU: inflated BCF bytes in HBM -> the VCF device
// layout (chrom id, pos, qual, filter-list id, one typed INFO field) in HBM.
//
// Reference: exon-bcf behind exon-core/src/datasources/bcf/ (same Arrow schema as VCF; pinned by
// exon-core/src/session_context/exon_context_ext.rs:1053-1090: index.bcf has 621 records, 191 on chromosome "1").
// Record layout: VCF 4.x specification section 6 (BCF2): `l_shared, l_indiv`, fixed fields (CHROM id, 0-based POS,
// rlen, QUAL with the 0x7F800001 missing sentinel, n_info | n_allele << 16, n_fmt << 24 | n_sample), typed ID / alleles
// / FILTER (dictionary indexes) / INFO (dictionary index -> typed value).  Same field rules as host/bcf.h.
//
// Records are found by the parallel chain walk of chain_walk.h (64 KiB segments, guessed starts proven by induction);
//   k_bcf_extract  one thread per record: fixed fields, typed-value walk to FILTER and the wanted INFO key; FILTER index
//                  lists are interned in a persistent device table (identity = 64-bit hash of the list)
//   k_bcf_assign / k_bcf_remap   dense ids for lists seen for the first time, provisional slot -> id
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "chain_walk.h"
#include "list_kernels.h"
#include "internal.h"

namespace {

using chain::ld16;
using chain::ld32;
using chain::SEG;
using chain::SegInfo;

constexpr int FSLOTS = 8192;  // open addressing; at most EXON_HIP_MAX_GROUPS distinct lists are supported
constexpr int FLIST = 8;      // longest FILTER list kept (longer ones: undecided)

struct BcfFormat {
  static constexpr uint32_t MIN_HEADER = 32, MIN_RECORD = 32, LEN_BYTES = 8;
  int32_t n_contigs, n_samples;
  __device__ __forceinline__ uint32_t record_bytes(const uint8_t* d, uint32_t r) const {
    const uint32_t ls = ld32(d + r), li = ld32(d + r + 4);
    return (ls < 24 || ls > (1u << 28) || li > (1u << 28)) ? 0u : 8u + ls + li;
  }
  __device__ bool plausible(const uint8_t* d, uint32_t n, uint32_t r) const {
    if ((uint64_t)r + 32 > n) return false;
    const uint32_t ls = ld32(d + r), li = ld32(d + r + 4);
    if (ls < 24 || ls > (1u << 24) || li > (1u << 26)) return false;
    const int32_t chrom = (int32_t)ld32(d + r + 8), pos = (int32_t)ld32(d + r + 12), rlen = (int32_t)ld32(d + r + 16);
    if (chrom < 0 || chrom >= n_contigs || pos < -1 || rlen < 0) return false;
    const uint32_t nia = ld32(d + r + 24), nfs = ld32(d + r + 28);
    if ((nia >> 16) == 0) return false;                          // at least REF
    if ((int32_t)(nfs & 0xFFFFFFu) != n_samples) return false;   // every record carries the header's sample count
    if (n_samples == 0 && li != 0) return false;
    if (ls > 24 && (uint64_t)r + 33 <= n && (d[r + 32] & 0xF) != 7) return false;  // ID is a typed string
    return true;
  }
};
constexpr uint32_t SEG_CAP = chain::seg_cap<BcfFormat>();

struct FilterLists {  // persistent across slabs
  unsigned long long* keys;  // 0 = empty
  int32_t* ids;              // -1 until assigned
  int32_t* lists;            // [FSLOTS][FLIST]
  int32_t* counts;           // [FSLOTS]
  int32_t* counters;         // [0] ids assigned, [1] table overflow
};

struct BcfOut {
  int32_t* chrom_id;
  int64_t* pos;
  float* qual;
  uint32_t* qual_valid;
  int32_t* filter_id;
  float* info[EXON_HIP_MAX_INFO_FIELDS];  // typed INFO fields (4-byte values), in the order of exon_hip_bcf_parser_set_info_keys
  uint32_t* info_valid[EXON_HIP_MAX_INFO_FIELDS];
  uint32_t* lv_off[EXON_HIP_MAX_INFO_FIELDS];  // list kinds ('F' / 'I'): where the row's typed vector starts (| value type << 29)
  uint32_t* lv_cnt[EXON_HIP_MAX_INFO_FIELDS];  // ... and how many items it holds (0: NULL list)
  uint32_t* pos_valid;  // POS 0 (BCF pos0 = -1, the telomere) is NULL like in the VCF path
  uint32_t* rec_of_row;  // byte offset of every row's record (id / ref / alt are built from it: text_columns.hip)
};
struct BcfInfoKeys {  // header-string indexes of the INFO fields to extract; kind 'f' -> f32, 'i' -> i32, 'b' Flag -> presence,
                      // 'F' / 'I' -> List<f32> / List<i32> (typed vectors: list_kernels.h + k_bcf_list_fill)
  int n;
  int32_t key[EXON_HIP_MAX_INFO_FIELDS];
  char kind[EXON_HIP_MAX_INFO_FIELDS];
};

__device__ __forceinline__ int type_size(int t) { return t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 4 : t == 5 ? 4 : t == 7 ? 1 : 0; }

struct Cursor {
  const uint8_t* d;
  uint32_t o, end;
  bool bad;
  __device__ int64_t read_int(int type) {
    const int sz = type_size(type);
    if (type < 1 || type > 3 || o + (uint32_t)sz > end) { bad = true; return 0; }
    int64_t v;
    if (type == 1) v = (int8_t)d[o];
    else if (type == 2) v = (int16_t)ld16(d + o);
    else v = (int32_t)ld32(d + o);
    o += (uint32_t)sz;
    return v;
  }
  __device__ void typed_header(int* type, int* count) {
    if (o >= end) { bad = true; *type = 0; *count = 0; return; }
    const uint8_t b = d[o++];
    *type = b & 0xF;
    *count = b >> 4;
    if (*count == 15) {  // the real count follows as a typed integer
      if (o >= end) { bad = true; return; }
      const uint8_t c = d[o++];
      *count = (int)read_int(c & 0xF);
      if (*count < 0) bad = true;
    }
  }
  __device__ void skip_typed() {
    int t, c;
    typed_header(&t, &c);
    o += (uint32_t)c * (uint32_t)type_size(t);
    if (o > end) bad = true;
  }
};

__global__ __launch_bounds__(256) void k_bcf_extract(const uint8_t* __restrict__ d, const SegInfo* __restrict__ seg,
                                                     const uint32_t* __restrict__ base, const uint32_t* __restrict__ rec_off, BcfOut out,
                                                     FilterLists f, int32_t n_contigs, int32_t n_strings, BcfInfoKeys ik,
                                                     unsigned* __restrict__ scalars) {
  const uint32_t s = blockIdx.x;
  if (scalars[1] != 0) return;  // the segmentation was not proven: nothing here can be trusted
  const uint32_t cnt = seg[s].count, row0 = base[s];
  const uint32_t* offs = rec_off + (size_t)s * SEG_CAP;
  // validity bits: one atomicOr per 32-row word and wave (a wave's 64 consecutive rows touch at most three words, words at segment
  // seams are shared with the neighbouring workgroup) instead of one per row and column (bam_parse.hip: 3.3 x on that kernel)
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
    out.rec_of_row[row] = r;
    const uint32_t wave_row = row - lane;  // the row of the wave's lane 0 (rows are consecutive across a wave's lanes)
    bool row_ok = false;                   // the record was decided: its validity bits count
    const uint32_t ls = ld32(d + r);
    const int32_t chrom = (int32_t)ld32(d + r + 8), pos0 = (int32_t)ld32(d + r + 12);
    const uint32_t qbits = ld32(d + r + 20), nia = ld32(d + r + 24);
    const int n_info = (int)(nia & 0xFFFF), n_allele = (int)(nia >> 16);
    Cursor c{d, r + 32, r + 8 + ls, false};
    c.skip_typed();                                      // ID
    for (int a = 0; a < n_allele && !c.bad; ++a) c.skip_typed();  // REF + ALTs
    // FILTER: typed int vector of dictionary indexes; empty = '.'
    int ft, fc;
    c.typed_header(&ft, &fc);
    int32_t list[FLIST];
    bool undecided = chrom < 0 || chrom >= n_contigs || fc > FLIST;
    unsigned long long h = 0xCBF29CE484222325ULL ^ (unsigned long long)(fc < 0 ? 0 : fc);
    for (int i = 0; i < fc && !c.bad && !undecided; ++i) {
      const int64_t v = c.read_int(ft);
      if (v < 0 || v >= n_strings) undecided = true;
      list[i] = (int32_t)v;
      h = (h ^ (unsigned long long)(v + 1)) * 0x100000001B3ULL;
    }
    // INFO: (typed key, typed value) pairs; the first occurrence of a key wins
    unsigned have = 0;  // bit w: INFO field w has a value (bit mask + direct stores: up to 16 keys without per-thread arrays)
    for (int q = 0; q < n_info && !c.bad && !undecided; ++q) {
      int kt, kc;
      c.typed_header(&kt, &kc);
      const int64_t key = kc ? c.read_int(kt) : -1;
      int vt, vc;
      c.typed_header(&vt, &vc);
      if (c.bad) break;
      for (int w = 0; w < ik.n; ++w) {
        if (key != ik.key[w] || (have >> w & 1u)) continue;
        if (ik.kind[w] == 'b') {
          have |= 1u << w;  // a Flag is true by being there
        } else if (ik.kind[w] == 'F' || ik.kind[w] == 'I') {
          // a typed vector -> List<item>: count the items up to the type's end-of-vector value; ONE item that is the type's
          // 'missing' value is `key=.` (NULL list).  The items are parsed by k_bcf_list_fill behind the offsets scan.
          const bool ints = vt >= 1 && vt <= 3, flts = vt == 5 && ik.kind[w] == 'F';
          if (vc >= 1 && (ints || flts)) {
            const uint32_t sz = (uint32_t)type_size(vt);
            if ((uint64_t)c.o + (uint64_t)vc * sz > c.end) { c.bad = true; break; }
            const int64_t missing = vt == 1 ? -128 : vt == 2 ? -32768 : (int64_t)INT32_MIN;
            unsigned items = 0;
            bool first_missing = false;
            for (int e = 0; e < vc; ++e) {
              Cursor t = c;
              t.o = c.o + (uint32_t)e * sz;
              if (flts) {
                const uint32_t bb = ld32(d + t.o);
                if (bb == 0x7F800002u) break;
                if (e == 0) first_missing = bb == 0x7F800001u;
              } else {
                const int64_t v = t.read_int(vt);
                if (v == missing + 1) break;
                if (e == 0) first_missing = v == missing;
              }
              ++items;
            }
            if (items > 0 && !(vc == 1 && first_missing)) {
              out.lv_off[w][row] = c.o | ((uint32_t)vt << 29);
              out.lv_cnt[w][row] = items;
              have |= 1u << w;
            }
          }
        } else if (ik.kind[w] == 'i') {
          // Type=Integer: int8 / int16 / int32 widened to Int32 exactly (bit pattern in the 4-byte column)
          if (vc >= 1 && vt >= 1 && vt <= 3) {
            Cursor t = c;
            const int64_t v = t.read_int(vt);
            if (t.bad) { c.bad = true; break; }
            const int64_t missing = vt == 1 ? -128 : vt == 2 ? -32768 : (int64_t)INT32_MIN;
            if (v != missing) {
              out.info[w][row] = __int_as_float((int32_t)v);
              have |= 1u << w;
            }
          }
        } else if (vc >= 1) {
          if (vt == 5) {
            if (c.o + 4 > c.end) { c.bad = true; break; }
            const uint32_t bb = ld32(d + c.o);
            if (bb != 0x7F800001u && bb != 0x7F800002u) {
              out.info[w][row] = __uint_as_float(bb);
              have |= 1u << w;
            }
          } else if (vt >= 1 && vt <= 3) {
            Cursor t = c;
            const int64_t v = t.read_int(vt);
            if (t.bad) { c.bad = true; break; }
            const int64_t missing = vt == 1 ? -128 : vt == 2 ? -32768 : (int64_t)INT32_MIN;
            if (v != missing) {
              out.info[w][row] = (float)v;
              have |= 1u << w;
            }
          }
        }
      }
      if (c.bad) break;
      c.o += (uint32_t)vc * (uint32_t)type_size(vt);
      if (c.o > c.end) c.bad = true;
    }
    if (c.bad || undecided) {
      atomicAdd(&scalars[1], 1u);
    } else {
    row_ok = true;
    // intern the FILTER list
    h |= 1ull;
    int slot = (int)(h & (unsigned long long)(FSLOTS - 1)), probes = 0;
    for (;; slot = (slot + 1) & (FSLOTS - 1)) {
      // plain read first: almost every record finds its list already there (device atomics on one address serialise)
      unsigned long long prev = *reinterpret_cast<volatile unsigned long long*>(&f.keys[slot]);
      if (prev == 0ull) prev = atomicCAS(&f.keys[slot], 0ull, h);
      if (prev == 0ull) {  // ours: publish the list (read by k_bcf_assign / the host only after this kernel)
        f.counts[slot] = fc;
        for (int i = 0; i < fc; ++i) f.lists[slot * FLIST + i] = list[i];
        break;
      }
      if (prev == h) break;
      if (++probes >= FSLOTS) {
        f.counters[1] = 1;
        slot = 0;
        break;
      }
    }
    out.chrom_id[row] = chrom;
    out.pos[row] = pos0 >= 0 ? (int64_t)pos0 + 1 : 0;
    out.qual[row] = qbits == 0x7F800001u ? 0.f : __uint_as_float(qbits);
    out.filter_id[row] = slot;
    for (int w = 0; w < ik.n; ++w) {
      if (have >> w & 1u) continue;
      if (ik.kind[w] == 'F' || ik.kind[w] == 'I') out.lv_cnt[w][row] = 0;  // NULL list: no items (summed by the offsets scan)
      else if (ik.kind[w] != 'b') out.info[w][row] = 0.f;  // NULL slots hold a defined value
    }
    }  // decided record
    publish(out.qual_valid, row_ok && qbits != 0x7F800001u, wave_row);
    publish(out.pos_valid, row_ok && pos0 >= 0, wave_row);
    for (int w = 0; w < ik.n; ++w) publish(out.info_valid[w], row_ok && (have >> w & 1u) != 0, wave_row);
  }
}

// list-valued INFO fields: offsets[row] = exclusive prefix of the item counts, every row copies its typed vector into
// values[offsets[row] ..] ('I': int8 / int16 / int32 -> int32 exactly; 'F': float bits, or integers converted); the type's
// 'missing' value is a NULL item (flag 0).  rows = scalars[0] (written by k_chain_check).
__global__ __launch_bounds__(LIST_TPB) void k_bcf_list_fill(const uint8_t* __restrict__ d, const uint32_t* __restrict__ lv_off,
                                                            const uint32_t* __restrict__ cnt, const unsigned* __restrict__ block_offsets,
                                                            const unsigned* __restrict__ n_rows_p, unsigned cap, unsigned cap_items, char kind,
                                                            int32_t* __restrict__ offsets, float* __restrict__ values,
                                                            uint8_t* __restrict__ item_flags, unsigned* __restrict__ exceptions) {
  const unsigned n_rows = min(*n_rows_p, cap);
  const unsigned row = blockIdx.x * LIST_TPB + threadIdx.x;
  const unsigned c = row < n_rows ? cnt[row] : 0u;
  const unsigned first = list_first_item(c, block_offsets);
  if (row < n_rows) offsets[row] = (int32_t)first;
  if (row + 1 == n_rows) offsets[n_rows] = (int32_t)(first + c);
  if (row == 0 && n_rows == 0) offsets[0] = 0;
  if (row >= n_rows || c == 0) return;
  if ((uint64_t)first + c > cap_items) {
    atomicAdd(exceptions, 1u);
    return;
  }
  const uint32_t o = lv_off[row] & 0x1FFFFFFFu;
  const int vt = (int)(lv_off[row] >> 29);
  const int64_t missing = vt == 1 ? -128 : vt == 2 ? -32768 : (int64_t)INT32_MIN;
  for (unsigned i = 0; i < c; ++i) {
    uint32_t bits = 0;
    bool ok;
    if (vt == 5) {
      bits = ld32(d + o + 4 * i);
      ok = bits != 0x7F800001u;
    } else {
      int64_t v;
      if (vt == 1) v = (int8_t)d[o + i];
      else if (vt == 2) v = (int16_t)ld16(d + o + 2 * i);
      else v = (int32_t)ld32(d + o + 4 * i);
      ok = v != missing;
      bits = kind == 'I' ? (uint32_t)(int32_t)v : __float_as_uint((float)v);
    }
    values[first + i] = __uint_as_float(ok ? bits : 0u);
    item_flags[first + i] = ok ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void k_bcf_assign(FilterLists f) {
  for (int s = threadIdx.x; s < FSLOTS; s += 256)
    if (f.keys[s] != 0ull && f.ids[s] < 0) f.ids[s] = atomicAdd(&f.counters[0], 1);
}

__global__ __launch_bounds__(256) void k_bcf_remap(int32_t* __restrict__ filter_id, const unsigned* __restrict__ scalars,
                                                   const int32_t* __restrict__ ids) {
  if (scalars[1] != 0) return;
  const int64_t n = scalars[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned slot = (unsigned)filter_id[i];
    filter_id[i] = slot < (unsigned)FSLOTS ? ids[slot] : 0;
  }
}

}  // namespace

struct exon_hip_bcf_parser {
  exon_hip_ctx* ctx = nullptr;
  int32_t n_contigs = 0, n_strings = 0, n_samples = 0, info_key = -1;
  BcfInfoKeys ik{};
  void* ibufs[2 * (EXON_HIP_MAX_INFO_FIELDS - 1)] = {nullptr};  // value / validity buffers of INFO fields 1 .. 15 (field 0 lives in bufs[5] / bufs[6])
  // list kinds: per key { vector offsets per row, item counts per row, Arrow offsets [rows + 1], item flags, item bitmap, items }
  void* lbufs[6 * EXON_HIP_MAX_INFO_FIELDS] = {nullptr};
  unsigned* d_list_blocks = nullptr;
  size_t words = 0;
  int64_t max_bytes = 0, max_rows = 0;
  uint32_t max_seg = 0;
  SegInfo* d_seg = nullptr;
  uint32_t *d_base = nullptr, *d_rec_off = nullptr, *d_scalars = nullptr;
  void* bufs[8] = {nullptr};
  void* fbufs[5] = {nullptr};
  uint32_t* d_rec_of_row = nullptr;
  BcfOut out{};
  FilterLists filters{};
  unsigned* h_scalars = nullptr;
};

extern "C" {

int exon_hip_bcf_parser_create(exon_hip_ctx* ctx, int32_t n_contigs, int32_t n_strings, int32_t n_samples, int32_t info_key,
                               int64_t max_bytes, exon_hip_bcf_parser** outp) {
  if (!ctx || !outp || max_bytes < 64 || n_contigs < 0 || n_strings < 0 || n_samples < 0)
    return fail(ctx, EXON_HIP_EINVAL, "exon_hip_bcf_parser_create: bad argument");
  if (max_bytes > 0xF0000000LL) return fail(ctx, EXON_HIP_EINVAL, "slab size must stay below 4 GiB (32-bit record offsets)");
  *outp = nullptr;
  exon_hip_bcf_parser* p = new (std::nothrow) exon_hip_bcf_parser();
  if (!p) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  p->ctx = ctx;
  p->n_contigs = n_contigs;
  p->n_strings = n_strings;
  p->n_samples = n_samples;
  p->info_key = info_key;
  p->max_bytes = max_bytes;
  p->max_seg = (uint32_t)((max_bytes + SEG - 1) / SEG);
  p->max_rows = max_bytes / 32 + 1;
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
  dalloc(&p->bufs[1], r * 8);
  dalloc(&p->bufs[2], r * 4);
  dalloc(&p->bufs[3], w);
  dalloc(&p->bufs[4], r * 4);
  dalloc(&p->bufs[5], r * 4);
  dalloc(&p->bufs[6], w);
  dalloc(&p->bufs[7], w);
  dalloc((void**)&p->d_rec_of_row, r * 4 + 64);
  dalloc(&p->fbufs[0], (size_t)FSLOTS * 8);
  dalloc(&p->fbufs[1], (size_t)FSLOTS * 4);
  dalloc(&p->fbufs[2], (size_t)FSLOTS * FLIST * 4);
  dalloc(&p->fbufs[3], (size_t)FSLOTS * 4);
  dalloc(&p->fbufs[4], 16);
  if (e == hipSuccess) e = hipMemset(p->fbufs[0], 0, (size_t)FSLOTS * 8);
  if (e == hipSuccess) e = hipMemset(p->fbufs[1], 0xFF, (size_t)FSLOTS * 4);
  if (e == hipSuccess) e = hipMemset(p->fbufs[4], 0, 16);
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_scalars, 16);
  if (e != hipSuccess) {
    const std::string msg = hipGetErrorString(e);
    exon_hip_bcf_parser_destroy(p);
    return fail(ctx, EXON_HIP_ENOMEM, "bcf parser allocation: %s", msg.c_str());
  }
  p->out = BcfOut{};
  p->out.chrom_id = (int32_t*)p->bufs[0];
  p->out.pos = (int64_t*)p->bufs[1];
  p->out.qual = (float*)p->bufs[2];
  p->out.qual_valid = (uint32_t*)p->bufs[3];
  p->out.filter_id = (int32_t*)p->bufs[4];
  p->out.info[0] = (float*)p->bufs[5];
  p->out.info_valid[0] = (uint32_t*)p->bufs[6];
  p->out.pos_valid = (uint32_t*)p->bufs[7];
  p->out.rec_of_row = p->d_rec_of_row;
  p->words = w;
  if (info_key >= 0) {
    p->ik.n = 1;
    p->ik.key[0] = info_key;
    p->ik.kind[0] = 'f';
  }
  p->filters = FilterLists{(unsigned long long*)p->fbufs[0], (int32_t*)p->fbufs[1], (int32_t*)p->fbufs[2], (int32_t*)p->fbufs[3],
                           (int32_t*)p->fbufs[4]};
  *outp = p;
  return EXON_HIP_OK;
}

// Several typed INFO fields (InfosBuilder children): header-string indexes + kinds ('f' numeric -> f32, 'b' Flag).  Replaces
// the single key given to _create; call before the first parse.
int exon_hip_bcf_parser_set_info_keys(exon_hip_bcf_parser* p, const int32_t* keys, const char* kinds, int32_t n) {
  if (!p || n < 0 || n > EXON_HIP_MAX_INFO_FIELDS || (n > 0 && (!keys || !kinds))) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_bcf_parser_set_info_keys: bad argument");
  p->ik = BcfInfoKeys{};
  p->ik.n = n;
  hipSetDevice(p->ctx->device);
  for (int q = 0; q < n; ++q) {
    if (kinds[q] != 'f' && kinds[q] != 'b' && kinds[q] != 'i' && kinds[q] != 'F' && kinds[q] != 'I')
      return fail(p->ctx, EXON_HIP_EUNSUPPORTED, "INFO kind '%c' is not decoded on the device", kinds[q]);
    p->ik.key[q] = keys[q];
    p->ik.kind[q] = kinds[q];
    if (kinds[q] == 'F' || kinds[q] == 'I') {  // an item takes at least one byte of the slab
      const size_t r = (size_t)p->max_rows, items = (size_t)p->max_bytes + 1;
      const size_t sizes[6] = {r * 4, r * 4, (r + 1) * 4, items, items / 8 + 64, items * 4};
      for (int k = 0; k < 6; ++k)
        if (!p->lbufs[6 * q + k] && !(p->lbufs[6 * q + k] = exon_pool_alloc(p->ctx, sizes[k]))) return fail(p->ctx, EXON_HIP_ENOMEM, "INFO list buffers");
      if (!p->d_list_blocks && !(p->d_list_blocks = (unsigned*)exon_pool_alloc(p->ctx, (r / LIST_TPB + 2) * 4))) return fail(p->ctx, EXON_HIP_ENOMEM, "INFO list buffers");
      p->out.lv_off[q] = (uint32_t*)p->lbufs[6 * q + 0];
      p->out.lv_cnt[q] = (uint32_t*)p->lbufs[6 * q + 1];
    }
    if (q == 0) {
      if (kinds[q] == 'F' || kinds[q] == 'I') p->out.info[0] = (float*)p->lbufs[5];
      continue;
    }
    if (!p->ibufs[2 * (q - 1)]) p->ibufs[2 * (q - 1)] = exon_pool_alloc(p->ctx, (size_t)p->max_rows * 4);
    if (!p->ibufs[2 * (q - 1) + 1]) p->ibufs[2 * (q - 1) + 1] = exon_pool_alloc(p->ctx, p->words);
    if (!p->ibufs[2 * (q - 1)] || !p->ibufs[2 * (q - 1) + 1]) return fail(p->ctx, EXON_HIP_ENOMEM, "INFO column buffers");
    p->out.info[q] = (kinds[q] == 'F' || kinds[q] == 'I') ? (float*)p->lbufs[6 * q + 5] : (float*)p->ibufs[2 * (q - 1)];
    p->out.info_valid[q] = (uint32_t*)p->ibufs[2 * (q - 1) + 1];
  }
  p->info_key = n ? keys[0] : -1;
  return EXON_HIP_OK;
}

int exon_hip_bcf_parser_destroy(exon_hip_bcf_parser* p) {
  if (!p) return EXON_HIP_OK;
  for (void* b : p->ibufs) exon_pool_free(p->ctx, b);
  for (void* b : p->lbufs)
    if (b) exon_pool_free(p->ctx, b);
  if (p->d_list_blocks) exon_pool_free(p->ctx, p->d_list_blocks);
  for (void* b : p->bufs) exon_pool_free(p->ctx, b);
  for (void* b : p->fbufs) exon_pool_free(p->ctx, b);
  exon_pool_free(p->ctx, p->d_seg);
  exon_pool_free(p->ctx, p->d_base);
  exon_pool_free(p->ctx, p->d_rec_off);
  if (p->d_rec_of_row) exon_pool_free(p->ctx, p->d_rec_of_row);
  exon_pool_free(p->ctx, p->d_scalars);
  if (p->h_scalars) hipHostFree(p->h_scalars);
  delete p;
  return EXON_HIP_OK;
}

int exon_hip_bcf_parser_parse(exon_hip_bcf_parser* p, void* stream, const uint8_t* d_data, int64_t n_bytes, exon_hip_vcf_columns* cols) {
  if (!p || !cols || (n_bytes > 0 && !d_data)) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_bcf_parser_parse: NULL argument");
  exon_hip_ctx* ctx = p->ctx;
  if (n_bytes > p->max_bytes) return fail(ctx, EXON_HIP_EINVAL, "slab of %lld bytes exceeds the parser's %lld", (long long)n_bytes, (long long)p->max_bytes);
  memset(cols, 0, sizeof *cols);
  if (n_bytes == 0) return EXON_HIP_OK;
  hipStream_t s = pick_stream(ctx, stream);
  const uint32_t n = (uint32_t)n_bytes, n_seg = (n + SEG - 1) / SEG;
  // rows of this slab <= n / 32: the validity words they can touch are cleared by k_chain_check, the scalars by k_chain_walk
  chain::ZeroList zl{};
  zl.p[zl.n++] = p->out.qual_valid;
  zl.p[zl.n++] = p->out.pos_valid;
  for (int q = 0; q < (p->ik.n ? p->ik.n : 1); ++q) zl.p[zl.n++] = p->out.info_valid[q];
  zl.words = (uint32_t)(((size_t)n / 32 + 1 + 31) / 32 + 1);
  hipLaunchKernelGGL(chain::k_chain_walk<BcfFormat>, dim3(n_seg), dim3(64), 0, s, d_data, n, BcfFormat{p->n_contigs, p->n_samples}, p->d_seg,
                     p->d_rec_off, p->d_scalars);
  hipLaunchKernelGGL(chain::k_chain_check<0>, dim3(1), dim3(256), 0, s, p->d_seg, n_seg, p->d_base, p->d_scalars, zl);
  hipLaunchKernelGGL(k_bcf_extract, dim3(n_seg), dim3(256), 0, s, d_data, p->d_seg, p->d_base, p->d_rec_off, p->out, p->filters,
                     p->n_contigs, p->n_strings, p->ik, p->d_scalars);
  hipLaunchKernelGGL(k_bcf_assign, dim3(1), dim3(256), 0, s, p->filters);
  hipLaunchKernelGGL(k_bcf_remap, dim3(std::min<uint32_t>(n_seg * 4 + 1, 4096)), dim3(256), 0, s, p->out.filter_id, p->d_scalars, p->filters.ids);
  for (int q = 0; q < p->ik.n; ++q) {  // list-valued fields: counts -> offsets -> items -> child validity
    const char kind = p->ik.kind[q];
    if (kind != 'F' && kind != 'I') continue;
    const unsigned row_bound = (unsigned)p->max_rows;
    const int lblocks = (int)((std::min<int64_t>(p->max_rows, n_bytes / 32 + 1) + LIST_TPB - 1) / LIST_TPB);  // a BCF record is >= 32 bytes
    int32_t* offsets = (int32_t*)p->lbufs[6 * q + 2];
    hipLaunchKernelGGL(k_list_block_sums, dim3(lblocks), dim3(LIST_TPB), 0, s, p->out.lv_cnt[q], p->d_scalars, row_bound, p->d_list_blocks);
    hipLaunchKernelGGL(k_list_scan_blocks, dim3(1), dim3(256), 0, s, p->d_list_blocks, lblocks, p->d_scalars + 3);
    hipLaunchKernelGGL(k_bcf_list_fill, dim3(lblocks), dim3(LIST_TPB), 0, s, d_data, p->out.lv_off[q], p->out.lv_cnt[q], p->d_list_blocks, p->d_scalars,
                       row_bound, (unsigned)std::min<int64_t>(p->max_bytes + 1, 0xFFFFFFFFLL), kind, offsets, p->out.info[q],
                       (uint8_t*)p->lbufs[6 * q + 3], p->d_scalars + 1);
    hipLaunchKernelGGL(k_pack_bits, dim3(1024), dim3(256), 0, s, (const uint8_t*)p->lbufs[6 * q + 3], offsets, p->d_scalars, row_bound,
                       (unsigned)std::min<int64_t>(p->max_bytes + 1, 0xFFFFFFFFLL), (uint8_t*)p->lbufs[6 * q + 4]);
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(p->h_scalars, p->d_scalars, 12, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  cols->n_rows = p->h_scalars[0];
  cols->n_undecided = p->h_scalars[1];
  cols->consumed_bytes = p->h_scalars[2];
  cols->chrom_id = p->out.chrom_id;
  cols->pos = p->out.pos;
  cols->pos_valid = (uint8_t*)p->out.pos_valid;  // pos0 = -1 (POS 0) -> NULL
  cols->qual = p->out.qual;
  cols->qual_valid = (uint8_t*)p->out.qual_valid;
  cols->filter_id = p->out.filter_id;
  cols->info = p->ik.n ? p->out.info[0] : nullptr;
  cols->info_valid = p->ik.n ? (uint8_t*)p->out.info_valid[0] : nullptr;
  cols->n_info = p->ik.n;
  for (int q = 0; q < p->ik.n; ++q) {
    cols->infos[q] = p->ik.kind[q] != 'b' ? p->out.info[q] : nullptr;
    cols->infos_valid[q] = (uint8_t*)p->out.info_valid[q];
    cols->info_kinds[q] = p->ik.kind[q];
    if (p->ik.kind[q] == 'F' || p->ik.kind[q] == 'I') {
      cols->list_offsets[q] = (int32_t*)p->lbufs[6 * q + 2];
      cols->list_item_valid[q] = (uint8_t*)p->lbufs[6 * q + 4];
    }
  }
  return EXON_HIP_OK;
}

// FILTER lists discovered so far, in id order: lists[i * 8 .. i * 8 + counts[i]) are dictionary (header string) indexes
int exon_hip_bcf_parser_filters(exon_hip_bcf_parser* p, int32_t* lists, int32_t* counts, int32_t cap, int32_t* n_filters) {
  if (!p || !n_filters) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_bcf_parser_filters: NULL argument");
  exon_hip_ctx* ctx = p->ctx;
  int32_t counters[4];
  HIP_TRY(ctx, hipMemcpy(counters, p->filters.counters, 16, hipMemcpyDeviceToHost));
  if (counters[1] || counters[0] > EXON_HIP_MAX_GROUPS) return fail(ctx, EXON_HIP_EUNSUPPORTED, "more than %d distinct FILTER lists", EXON_HIP_MAX_GROUPS);
  *n_filters = counters[0];
  if (!lists || !counts) return EXON_HIP_OK;
  if (counters[0] > cap) return fail(ctx, EXON_HIP_EINVAL, "filter list buffer too small (%d needed)", counters[0]);
  std::vector<unsigned long long> keys(FSLOTS);
  std::vector<int32_t> ids(FSLOTS), l((size_t)FSLOTS * FLIST), c(FSLOTS);
  HIP_TRY(ctx, hipMemcpy(keys.data(), p->filters.keys, (size_t)FSLOTS * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(ids.data(), p->filters.ids, (size_t)FSLOTS * 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(l.data(), p->filters.lists, (size_t)FSLOTS * FLIST * 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(c.data(), p->filters.counts, (size_t)FSLOTS * 4, hipMemcpyDeviceToHost));
  for (int s = 0; s < FSLOTS; ++s)
    if (keys[(size_t)s] != 0 && ids[(size_t)s] >= 0 && ids[(size_t)s] < counters[0]) {
      const int id = ids[(size_t)s];
      counts[id] = c[(size_t)s];
      for (int i = 0; i < FLIST; ++i) lists[(size_t)id * FLIST + i] = l[(size_t)s * FLIST + i];
    }
  return EXON_HIP_OK;
}

}  // extern "C"

const uint32_t* exon_hip_bcf_parser_row_records(exon_hip_bcf_parser* p) { return p ? p->out.rec_of_row : nullptr; }
