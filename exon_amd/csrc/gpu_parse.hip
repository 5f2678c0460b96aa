// gpu_parse.hip -- VCF record parsing ON THE GPU: raw text in HBM -> device-layout columns in HBM.
//
// The host decoders (host/formats.h, host/parallel.h) top out at ~150 Mrows/s on the GPU boxes' 16-CPU quota while the
// filter+aggregate kernels consume 500 000 Mrows/s; text crossing PCIe as-is and being parsed on the device lifts
// the decode stage to the PCIe rate ("GPU-side parse later").  Semantics are those of
// LazyVCFArrayBuilder::append (exon-vcf/src/array_builder/lazy_array_builder.rs:159-216) restricted to the columns
// of the device layout: chrom -> dictionary id, pos (0 / '.' -> NULL), qual ('.' -> NULL, correctly rounded f32),
// filter -> dictionary id of the ';'-joined list ('.' -> empty list), one typed INFO field (Number=1 Float/Integer,
// missing key / '.' -> NULL).
//
// Pipeline for one slab of complete lines:
//   k_index_lines      positions of all newlines, in order (line i = (nl[i-1], nl[i])), in ONE pass: decoupled look-back over
//                      the tiles' counts (round 5; k_count_newlines / k_scan_blocks / k_fill_newlines, the three-launch form of
//                      rounds 1-4, stay behind EXON_HIP_LINE_INDEX_PASSES=2)
//   k_parse_lines      one thread per line: split on tabs, parse, look names up in hash tables, ballot the validity
//                      bitmaps; FILTER lists not seen before are inserted with atomicCAS (slot = provisional id)
//   k_assign_filters   dense ids for newly inserted FILTER lists, their text copied to a persistent pool
//   k_remap_filters    provisional slot -> dense id
// Rows the device cannot decide (a float with > 19 significant digits, a contig missing from the header, a malformed
// line) are counted; the caller then re-decodes that slab on the host, so results never differ from the CPU path.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "host/decimal_f32.h"
#include "internal.h"
#include "list_kernels.h"

namespace {

constexpr int TPB = 256;
constexpr int BYTES_PER_THREAD = 64;  // four 16-byte loads per thread (one cache line): 4 KiB workgroups were launch-bound
constexpr int BYTES_PER_BLOCK = TPB * BYTES_PER_THREAD;
constexpr int FILTER_SLOTS = 8192;  // open addressing; at most EXON_HIP_MAX_GROUPS distinct lists are supported
constexpr int FILTER_POOL = 1 << 20;

__host__ __device__ inline uint64_t fnv1a(const uint8_t* p, int n) {
  uint64_t h = 0xCBF29CE484222325ULL;
  for (int i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 0x100000001B3ULL;
  }
  return h | 1ULL;  // never 0 (0 = empty slot)
}

__device__ __forceinline__ int count_nl16(uint4 v) {
  int c = 0;
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c += ((w[i] & 0xFF) == 0x0A) + (((w[i] >> 8) & 0xFF) == 0x0A) + (((w[i] >> 16) & 0xFF) == 0x0A) + ((w[i] >> 24) == 0x0A);
  }
  return c;
}

// `text` is 16-byte aligned; the slab proper starts `skip` (< 16) bytes into it: those bytes read as zeros
__device__ __forceinline__ uint4 load16(const uint8_t* text, int64_t n, int64_t off, unsigned skip) {
  uint4 v = {0, 0, 0, 0};
  unsigned* w = &v.x;
  if (off + 16 <= n) {
    v = *reinterpret_cast<const uint4*>(text + off);
  } else {
    for (int i = 0; i < 16 && off + i < n; ++i) w[i >> 2] |= (unsigned)text[off + i] << (8 * (i & 3));
  }
  if (off == 0 && skip) {
    for (unsigned i = 0; i < skip; ++i) w[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
  }
  return v;
}

__global__ __launch_bounds__(TPB) void k_count_newlines(const uint8_t* __restrict__ text, int64_t n, unsigned skip,
                                                        unsigned* __restrict__ block_counts) {
  __shared__ unsigned red[TPB / 64];
  const int64_t off = ((int64_t)blockIdx.x * TPB + threadIdx.x) * BYTES_PER_THREAD;
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < BYTES_PER_THREAD / 16; ++j)
    if (off + 16 * j < n) c += (unsigned)count_nl16(load16(text, n, off + 16 * j, skip));
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of `nb` workgroup totals in place; total -> *n_lines.  zero_rest: n_lines is the slab's scalar block
// {lines, exceptions, consumed, -}: words 1..3 are cleared here (the kernels behind this one add to them), which replaces
// the 16-byte hipMemsetAsync every slab used to start with -- a tiny kernel of its own, queued behind the inflate
// (One workgroup of 256 threads, not 1024: a workgroup starts only when ONE CU has wave slots for all of it, and beside the
//  inflate of the next slab -- 30 of a CU's 32 slots -- sixteen free slots took 0.5-1.3 ms to appear: round 4.)
__global__ __launch_bounds__(256) void k_scan_blocks(unsigned* __restrict__ counts, int nb, unsigned* __restrict__ n_lines, int zero_rest) {
  if (zero_rest && threadIdx.x >= 1 && threadIdx.x <= 3) n_lines[threadIdx.x] = 0;
  __shared__ unsigned part[256];
  const int per = (nb + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
  unsigned s = 0;
  for (int b = b0; b < b1; ++b) s += counts[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan
    unsigned v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int b = b0; b < b1; ++b) {
    const unsigned c = counts[b];
    counts[b] = run;
    run += c;
  }
  if (threadIdx.x == 255) *n_lines = part[255];
}

__global__ __launch_bounds__(TPB) void k_fill_newlines(const uint8_t* __restrict__ text, int64_t n, unsigned skip,
                                                       const unsigned* __restrict__ block_offsets,
                                                       unsigned* __restrict__ nl_pos, unsigned cap) {
  __shared__ unsigned wave_tot[TPB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t off = ((int64_t)blockIdx.x * TPB + threadIdx.x) * BYTES_PER_THREAD;
  constexpr int Q = BYTES_PER_THREAD / 16;
  uint4 v[Q];
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    v[j] = uint4{0, 0, 0, 0};
    if (off + 16 * j < n) {
      v[j] = load16(text, n, off + 16 * j, skip);
      c += (unsigned)count_nl16(v[j]);
    }
  }
  unsigned incl = c;  // inclusive scan within the wave
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned base = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  unsigned k = base + incl - c;
  if (c) {
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      if ((w[0] | w[1] | w[2] | w[3]) == 0) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (((w[i >> 2] >> (8 * (i & 3))) & 0xFF) == 0x0A) {
          if (k < cap) nl_pos[k] = (unsigned)(off + 16 * j + i);  // more lines than VCF records can fill: reported by the caller
          ++k;
        }
    }
  }
}

// ---- the line index in ONE pass over the text (round 5) -------------------------------------------------------------------------
// k_count_newlines + k_scan_blocks + k_fill_newlines read the slab twice and are three launches (each of which queues behind the
// next slab's inflate).  Here a workgroup takes its tile off a counter (so tile t runs only after tiles 0 .. t-1 have started),
// counts its newlines, publishes the count, finds the number of newlines in front of its tile by looking BACK over its
// predecessors' published words -- a word is (generation, value, flag): flag 1 = the tile's own count, flag 2 = the count of
// everything up to and including the tile; a wave reads 64 predecessors at a time and stops at the first "inclusive" word -- and
// writes the positions of its newlines.  Decoupled look-back; the words are written and read with agent-scope atomics (the 8
// XCDs do not share an L2), the generation makes last slab's words read as "not there yet" (no clearing pass), and the counter
// wraps to 0 by itself (atomicInc).  A look-back that does not see its predecessor within LOOKBACK_SPINS reads gives up, poisons its
// own word (the tiles behind it give up at once) and marks the slab "one undecided record": the host decoder takes it -- a hang is
// not possible.
constexpr unsigned LOOKBACK_SPINS = 1u << 22;
__device__ __forceinline__ void st_agent_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long* p) {
  return __hip_atomic_load(const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Tiles are 64 KiB (1024 threads x 64 bytes): the look-back advances 64 tiles per round trip through the L2 atomics (~1 us), so with
// the 16 KiB tiles of the first version a 516 MB slab's 31.5 k tiles took 0.47 ms -- 1.1 TB/s, bound by the chain, not by HBM.
#ifndef EXON_IDX_TPB
#define EXON_IDX_TPB 1024
#endif
#ifndef EXON_IDX_BPT
#define EXON_IDX_BPT 64
#endif
constexpr int IDX_TPB = EXON_IDX_TPB, IDX_BPT = EXON_IDX_BPT;
__global__ __launch_bounds__(IDX_TPB) void k_index_lines(const uint8_t* __restrict__ text, int64_t n, unsigned skip, unsigned long long* __restrict__ words,
                                                         unsigned* __restrict__ tile_ctr, unsigned nblocks, unsigned gen, unsigned* __restrict__ nl_pos,
                                                         unsigned cap, unsigned* __restrict__ scalars) {
  constexpr int TPB = IDX_TPB, BYTES_PER_THREAD = IDX_BPT;  // (this kernel's own tile: shadows the other kernels' constants)
  __shared__ unsigned wave_tot[TPB / 64];
  __shared__ unsigned s_tile, s_prefix, s_bad;
  if (threadIdx.x == 0) {
    s_tile = atomicInc(tile_ctr, nblocks - 1u);
    s_bad = 0;
  }
  __syncthreads();
  const unsigned tile = s_tile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t off = ((int64_t)tile * TPB + threadIdx.x) * BYTES_PER_THREAD;
  constexpr int Q = BYTES_PER_THREAD / 16;
  uint4 v[Q];
  unsigned c = 0;
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    v[j] = uint4{0, 0, 0, 0};
    if (off + 16 * j < n) {
      v[j] = load16(text, n, off + 16 * j, skip);
      c += (unsigned)count_nl16(v[j]);
    }
  }
  unsigned incl = c;  // inclusive scan within the wave
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned T = 0;
#pragma unroll
  for (int w = 0; w < TPB / 64; ++w) T += wave_tot[w];
  const unsigned long long g = (unsigned long long)(gen & 0x3FFFFFFFu) << 34;
  if (wave == 0) {
    if (lane == 0) st_agent_u64(&words[tile], g | ((unsigned long long)T << 2) | (tile == 0 ? 2ull : 1ull));
    unsigned prefix = 0;
    bool bad = false;
    if (tile != 0) {
      for (int64_t j0 = (int64_t)tile - 1;; j0 -= 64) {
        const int64_t j = j0 - lane;
        unsigned long long w = 0;
        if (j >= 0) {
          unsigned spins = 0;
          do {
            w = ld_agent_u64(&words[j]);
          } while (((w >> 34) != (g >> 34) || (w & 3ull) == 0) && ++spins < LOOKBACK_SPINS);
          if ((w >> 34) != (g >> 34) || (w & 3ull) == 0 || (w & 3ull) == 3ull) bad = true;  // (3: a predecessor gave up)
        }
        if (__any(bad)) {
          bad = true;
          break;
        }
        const unsigned long long incl_lanes = __ballot(j >= 0 && (w & 3ull) == 2ull);
        const int first = incl_lanes ? __ffsll((long long)incl_lanes) - 1 : 64;
        unsigned add = (j >= 0 && lane <= first) ? (unsigned)(w >> 2) : 0u;
        for (int o = 32; o > 0; o >>= 1) add += __shfl_xor(add, o, 64);
        prefix += add;
        if (incl_lanes || j0 < 64) break;
      }
    }
    if (lane == 0) {
      s_prefix = prefix;
      s_bad = bad ? 1u : 0u;
      if (bad) st_agent_u64(&words[tile], g | 3ull);  // the tiles behind this one give up at once
      else if (tile != 0) st_agent_u64(&words[tile], g | ((unsigned long long)(prefix + T) << 2) | 2ull);
    }
  }
  __syncthreads();
  if (s_bad) {
    // no lines, one undecided record: the kernels behind do nothing, the caller hands the slab to the host decoder.  The verdict
    // is STICKY in tile_ctr[1] (= this launch's generation; no tile ever clears it): a tile further on may have read this tile's
    // aggregate before it gave up, finish its look-back, and -- as the last tile -- publish a total and zero scalars[1..3] AFTER
    // the two stores below.  k_index_verdict, ordered behind this kernel, re-applies the verdict from the sticky word.
    if (threadIdx.x == 0) {
      atomicExch(&tile_ctr[1], gen);
      scalars[0] = 0;
      scalars[1] = 1;
    }
    return;
  }
  unsigned base = s_prefix;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  unsigned k = base + incl - c;
  if (c) {
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      if ((w[0] | w[1] | w[2] | w[3]) == 0) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (((w[i >> 2] >> (8 * (i & 3))) & 0xFF) == 0x0A) {
          if (k < cap) nl_pos[k] = (unsigned)(off + 16 * j + i);
          ++k;
        }
    }
  }
  // the last tile in tile order holds the total; it also clears words 1..3 of the slab's scalar block {lines, exceptions,
  // consumed, -}, which only the kernels BEHIND this one add to (what k_scan_blocks did; no 16-byte memset per slab)
  if (tile == nblocks - 1u && threadIdx.x < 4) {
    if (threadIdx.x == 0) scalars[0] = s_prefix + T;
    else scalars[threadIdx.x] = 0;
  }
}
// behind k_index_lines on the same stream: a tile that gave up wins over whatever the last tile published
__global__ void k_index_verdict(const unsigned* __restrict__ tile_ctr, unsigned gen, unsigned* __restrict__ scalars) {
  if (tile_ctr[1] == gen) {
    scalars[0] = 0;
    scalars[1] = 1;
  }
}
// one launch instead of count / scan / fill; EXON_HIP_LINE_INDEX_PASSES=2 keeps the three kernels (A/B)
static void launch_line_index(hipStream_t s, const uint8_t* d_text, int64_t n_bytes, unsigned skip, unsigned* d_block_counts, int64_t max_blocks, int nblocks, unsigned* gen,
                              unsigned* d_nl, unsigned cap, unsigned* d_scalars) {
  static const bool two_pass = [] {
    const char* v = getenv("EXON_HIP_LINE_INDEX_PASSES");
    return v && v[0] == '2';
  }();
  if (two_pass) {
    hipLaunchKernelGGL(k_count_newlines, dim3(nblocks), dim3(TPB), 0, s, d_text, n_bytes, skip, d_block_counts);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, d_block_counts, nblocks, d_scalars, 1);
    hipLaunchKernelGGL(k_fill_newlines, dim3(nblocks), dim3(TPB), 0, s, d_text, n_bytes, skip, d_block_counts, d_nl, cap);
    return;
  }
  unsigned long long* words = reinterpret_cast<unsigned long long*>(d_block_counts);  // [max blocks] words, then the tile counter
  *gen = (*gen + 1u) & 0x3FFFFFFFu;
  if (*gen == 0) *gen = 1;
  const int64_t tile_bytes = (int64_t)IDX_TPB * IDX_BPT;
  const unsigned ntiles = (unsigned)std::max<int64_t>(1, (n_bytes + tile_bytes - 1) / tile_bytes);  // (<= nblocks <= max_blocks: the words fit)
  hipLaunchKernelGGL(k_index_lines, dim3(ntiles), dim3(IDX_TPB), 0, s, d_text, n_bytes, skip, words, reinterpret_cast<unsigned*>(words + max_blocks), ntiles, *gen, d_nl,
                     cap, d_scalars);
  hipLaunchKernelGGL(k_index_verdict, dim3(1), dim3(1), 0, s, reinterpret_cast<const unsigned*>(words + max_blocks), *gen, d_scalars);
}

// scalars[2] = bytes up to and including the last newline (what a caller may discard after this slab)
__global__ void k_last_newline(const unsigned* __restrict__ nl_pos, unsigned* __restrict__ scalars, unsigned cap) {
  const unsigned n = scalars[0];
  scalars[2] = (n && n <= cap) ? nl_pos[n - 1] + 1u : 0u;
}

struct NameTable {  // open-addressing table of strings (contigs): hash -> id, text verified
  const uint64_t* keys;
  const int32_t* ids;
  const uint32_t* text_off;
  const uint32_t* text_len;
  const uint8_t* pool;
  int mask;
};

struct FilterTable {
  unsigned long long* keys;  // 0 = empty
  int32_t* ids;              // -1 until k_assign_filters has run
  uint32_t* text_off;        // provisional: offset into the CURRENT slab; after assignment: offset into pool
  uint32_t* text_len;
  uint8_t* pool;
  int32_t* counters;  // [0] = number of dense ids, [1] = pool bytes used, [2] = overflow flag
};

constexpr int MAX_INFO = EXON_HIP_MAX_INFO_FIELDS;  // 16: the by-value key table below is 16 x 9 bytes of kernel arguments
// the typed INFO fields a parser extracts (InfosBuilder children: exon-vcf/src/array_builder/info_builder.rs:152-309):
// kind 'f' = Number=1 Float -> f32 + validity; 'i' = Number=1 Integer -> i32 + validity (the 4-byte column holds the bit
// pattern); 'b' = Flag -> presence bitmap (value true where valid); 'F' / 'I' = any other Number of Float / Integer ->
// List<f32> / List<i32> (info_builder.rs:258-305): k_parse_lines records where the value text is and how many items it has,
// k_list_fill parses the items behind an exclusive scan of the counts (offsets), k_pack_bits turns the per-item flags
// into the child validity bitmap.  info_valid[q] is the LIST validity (NULL list: key absent, `key=.`, INFO '.')
struct InfoKeys {
  int n;
  int len[MAX_INFO];
  int off[MAX_INFO];  // into `text`
  char kind[MAX_INFO];
  const uint8_t* text;
};
struct ParseOut {
  int32_t* chrom_id;
  int64_t* pos;
  uint8_t* pos_valid;
  float* qual;
  uint8_t* qual_valid;
  int32_t* filter_id;
  float* info[MAX_INFO];
  uint8_t* info_valid[MAX_INFO];
  uint32_t* lv_off[MAX_INFO];  // list kinds: offset of the value text in the slab / number of items, per row
  uint32_t* lv_cnt[MAX_INFO];
  unsigned* exceptions;  // [0] = count of rows the device could not decide
};

__device__ __forceinline__ void store_valid(uint8_t* bm, int64_t row0_of_wave, int64_t n_rows, bool v, int lane) {
  const unsigned long long m = __ballot(v);
  if (lane < 8) {
    const int64_t r = row0_of_wave + lane * 8;
    if (r < n_rows) bm[r >> 3] = (uint8_t)(m >> (lane * 8));
  }
}

// bit b = byte b of the 16-byte group equals `c` (splat as c * 0x01010101).  ONE mask per group and one loop over its bits: an inner
// loop per dword -- four divergent regions per group -- is exec-mask bookkeeping on the CU's scalar unit, which bounds k_parse_lines.
__device__ __forceinline__ unsigned eq_mask16(const uint4& v, uint32_t splat) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  unsigned mask = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t x = w[k] ^ splat;  // equal bytes become 0
    const uint32_t m = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 exactly in the zero bytes
    const uint32_t t = m >> 7;                                                 // bits 0, 8, 16, 24
    mask |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu) << (4 * k);
  }
  return mask;
}
// ... restricted to the bytes [begin, end) of the slab, the group starting at byte a
__device__ __forceinline__ unsigned clip_mask16(unsigned mask, unsigned a, unsigned begin, unsigned end) {
  if (begin > a) mask &= 0xFFFFu << (begin - a);
  if (end - a < 16u) mask &= (1u << (end - a)) - 1u;
  return mask;
}

__global__ __launch_bounds__(TPB) void k_parse_lines(const uint8_t* __restrict__ text, const unsigned* __restrict__ nl_pos,
                                                     const unsigned* __restrict__ n_lines_p, NameTable contigs,
                                                     FilterTable filters, InfoKeys ik, ParseOut out, unsigned cap, unsigned skip,
                                                     unsigned n_total) {
  const int64_t n_rows = min(*n_lines_p, cap);
  // an aligned 16-byte group of the slab; the last one is read byte by byte (nothing behind n_total is touched)
  auto group16 = [&](unsigned a) {
    uint4 v = {0, 0, 0, 0};
    if (a + 16u <= n_total) {
      v = *reinterpret_cast<const uint4*>(text + a);
    } else {
      unsigned* w = &v.x;
      for (unsigned i = 0; a + i < n_total; ++i) w[i >> 2] |= (unsigned)text[a + i] << (8 * (i & 3));
    }
    return v;
  };
  const int64_t row = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool pos_ok = false, qual_ok = false, bad = false;
  unsigned info_ok = 0;  // bit q: INFO field q has a value in this row (bit masks, not arrays: up to 16 keys stay in registers)
  if (row < n_rows) {
    const unsigned begin = row ? nl_pos[row - 1] + 1 : skip;
    unsigned end = nl_pos[row];
    if (end > begin && text[end - 1] == '\r') --end;
    // split the first 8 fields
    unsigned fs[9];
    int nf = 0;
    fs[0] = begin;
    // tabs, 16 bytes per load (aligned groups; `text` is 16-byte aligned and padded): the byte-at-a-time version of this
    // loop was a chain of ~35 dependent loads per line, a quarter of the kernel's memory waits
    for (unsigned a = begin & ~15u; a < end && nf < 8; a += 16) {
      unsigned m = clip_mask16(eq_mask16(group16(a), 0x09090909u), a, begin, end);  // the group's tabs
      while (m && nf < 8) {
        fs[++nf] = a + (unsigned)__ffs((int)m);  // the field starts behind the tab
        m &= m - 1;
      }
    }
    // field f spans [fs[f], fs[f+1] - 1) for f < nf, the last one ends at `end` (INFO may be followed by FORMAT...)
    auto fbeg = [&](int f) { return fs[f]; };
    auto fend = [&](int f) { return f < nf ? fs[f + 1] - 1 : end; };
    if (nf < 7 || begin == end || text[begin] == '#') {
      bad = true;  // not a data line with 8 fields
      // the row's slots still get defined values: k_remap_filters indexes a table with filter_id, and the buffers are
      // recycled between scans (a corrupted file after other scans faulted there: found by tools/fuzz_gpu_decode.py)
      out.chrom_id[row] = 0;
      out.pos[row] = 0;
      out.qual[row] = 0.f;
      out.filter_id[row] = 0;
      for (int q = 0; q < ik.n; ++q)
        if (ik.kind[q] == 'F' || ik.kind[q] == 'I') out.lv_cnt[q][row] = 0;  // summed by the offsets scan
    } else {
#ifndef EXON_PARSE_SKIP  // (profiling builds: bit 0 CHROM, 1 POS, 2 QUAL, 3 FILTER, 4 INFO left out)
#define EXON_PARSE_SKIP 0
#endif
      // CHROM
      if (!(EXON_PARSE_SKIP & 1)) {
        const uint8_t* p = text + fbeg(0);
        const int len = (int)(fend(0) - fbeg(0));
        const uint64_t h = fnv1a(p, len);
        int slot = (int)(h & (uint64_t)contigs.mask), id = -1;
        for (int probe = 0; probe <= contigs.mask; ++probe) {
          const uint64_t k = contigs.keys[slot];
          if (k == 0) break;
          if (k == h && (int)contigs.text_len[slot] == len) {
            bool same = true;
            for (int i = 0; i < len && same; ++i) same = contigs.pool[contigs.text_off[slot] + i] == p[i];
            if (same) {
              id = contigs.ids[slot];
              break;
            }
          }
          slot = (slot + 1) & contigs.mask;
        }
        if (id < 0) bad = true;  // contig not in the header: host decides the id
        out.chrom_id[row] = id < 0 ? 0 : id;
      }
      // POS
      if (!(EXON_PARSE_SKIP & 2)) {
        uint64_t v = 0;  // (unsigned: a spoiled or over-long value wraps, it never overflows a signed integer)
        unsigned pb = fbeg(1), pn = fend(1) - fbeg(1);
        if (pn && text[pb] == '+') ++pb, --pn;  // usize::from_str takes one leading '+' (host/formats.h parse_pos)
        bool ok = pn > 0;
        if (pn <= 16 && pb + 16u <= n_total) {  // the digits from two (unaligned) 8-byte loads instead of a chain of byte loads
          uint64_t w[2];
          __builtin_memcpy(w, text + pb, 16);
          for (unsigned k = 0; k < pn; ++k) {  // (no branch inside: a non-digit spoils v, which is then not used)
            const unsigned d = ((unsigned)(w[k >> 3] >> (8 * (k & 7))) & 0xFFu) - (unsigned)'0';
            ok &= d <= 9u;
            v = v * 10 + d;
          }
        } else {
          for (unsigned i = pb; i < pb + pn && ok; ++i) {
            const uint8_t c = text[i];
            if (c < '0' || c > '9') ok = false;
            else v = v * 10 + (c - '0');
          }
        }
        if (pn > 18) ok = false;  // beyond 18 digits the host decides (usize overflow is an error there)
        pos_ok = ok && v > 0;
        if (!ok) bad = true;  // not a number: `variant_start().transpose()?` is an error in the reference -- the host reports it
        out.pos[row] = pos_ok ? (int64_t)v : 0;
      }
      // QUAL
      if (!(EXON_PARSE_SKIP & 4)) {
        const int len = (int)(fend(5) - fbeg(5));
        float q = 0.f;
        if (!(len == 1 && text[fbeg(5)] == '.')) {
          uint32_t bits;
          if (exon::dec::parse_f32(reinterpret_cast<const char*>(text + fbeg(5)), len, &bits)) {
            q = __uint_as_float(bits);
            qual_ok = true;
          } else {
            bad = true;
          }
        }
        out.qual[row] = q;
      }
      // FILTER: '.' -> the empty list
      if (!(EXON_PARSE_SKIP & 8)) {
        const uint8_t* p = text + fbeg(6);
        int len = (int)(fend(6) - fbeg(6));
        if (len == 1 && p[0] == '.') len = 0;
        const unsigned long long h = fnv1a(p, len);
        int slot = (int)(h & (FILTER_SLOTS - 1));
        int found = -1;
        for (int probe = 0; probe < FILTER_SLOTS; ++probe) {
          unsigned long long k = filters.keys[slot];
          if (k == 0) {
            k = atomicCAS(&filters.keys[slot], 0ull, h);
            if (k == 0) {  // this thread inserted the key: remember where its text lives in this slab
              filters.text_off[slot] = fbeg(6);
              filters.text_len[slot] = (uint32_t)len;
              found = slot;
              break;
            }
          }
          if (k == h) {
            found = slot;
            break;
          }
          slot = (slot + 1) & (FILTER_SLOTS - 1);
        }
        if (found < 0) {
          atomicExch(&filters.counters[2], 1);
          found = 0;
        }
        out.filter_id[row] = found;  // provisional: slot index
      }
      // INFO: `key=value` (or a bare Flag key) among ';'-separated entries; the first occurrence of a key wins
      if (ik.n > 0 && !(EXON_PARSE_SKIP & 16)) {
        unsigned seen = 0;  // bit q: key q was met (the first occurrence wins); values are stored as they are parsed
        int left = ik.n;
        const unsigned ib = fbeg(7), ie = fend(7);
        if (!(ie - ib == 1 && text[ib] == '.')) {  // INFO '.': the whole struct is NULL
          // entries are separated by ';': find the separators 16 bytes per load, test the keys at every entry start
          unsigned i = ib;  // start of the current entry
          auto entry = [&](unsigned j) {  // the entry [i, j)
            for (int q = 0; q < ik.n; ++q) {
              const int kl = ik.len[q];
              if ((seen >> q & 1u) || (int)(j - i) < kl) continue;
              const bool valued = (int)(j - i) > kl && text[i + kl] == '=';
              if (!valued && (int)(j - i) != kl) continue;
              bool same = true;
              for (int k = 0; k < kl && same; ++k) same = text[i + k] == ik.text[ik.off[q] + k];
              if (!same) continue;
              seen |= 1u << q;
              --left;
              if (ik.kind[q] == 'b') {
                info_ok |= 1u << q;  // a Flag is true by being there
              } else if (valued) {
                const unsigned vb = i + kl + 1;
                const int vl = (int)(j - vb);
                if (!(vl == 0 || (vl == 1 && text[vb] == '.'))) {
                  uint32_t bits;
                  if (ik.kind[q] == 's') {  // String / Character: the value's text; its dictionary id comes from k_info_string_ids
                    out.lv_off[q][row] = vb;
                    out.lv_cnt[q][row] = (uint32_t)vl;
                    info_ok |= 1u << q;
                  } else if (ik.kind[q] == 'F' || ik.kind[q] == 'I') {
                    unsigned items = 1;  // items are separated by ','; they are parsed by k_list_fill
                    for (int k = 0; k < vl; ++k) items += text[vb + k] == ',';
                    out.lv_off[q][row] = vb;
                    out.lv_cnt[q][row] = items;
                    info_ok |= 1u << q;
                  } else if (ik.kind[q] == 'i') {
                    // Type=Integer: exact int32 ([+-] digits); the value travels as its bit pattern in the 4-byte column.
                    // Anything else (including out of range) is the reference's parse error: the row is left to the host
                    int k = 0;
                    const bool neg = text[vb] == '-';
                    if (neg || text[vb] == '+') k = 1;
                    int64_t iv = 0;
                    bool ok = k < vl && vl - k <= 10;
                    for (; k < vl && ok; ++k) {
                      const unsigned d = (unsigned)text[vb + k] - '0';
                      ok = d <= 9u;
                      iv = iv * 10 + d;
                    }
                    if (neg) iv = -iv;
                    if (ok && iv >= INT32_MIN && iv <= INT32_MAX) {
                      out.info[q][row] = __int_as_float((int32_t)iv);
                      info_ok |= 1u << q;
                    } else {
                      bad = true;
                    }
                  } else if (exon::dec::parse_f32(reinterpret_cast<const char*>(text + vb), vl, &bits)) {
                    out.info[q][row] = __uint_as_float(bits);
                    info_ok |= 1u << q;
                  } else {
                    bad = true;
                  }
                }
              }
            }
            i = j + 1;
          };
          for (unsigned a = ib & ~15u; a < ie && left > 0; a += 16) {
            unsigned m = clip_mask16(eq_mask16(group16(a), 0x3B3B3B3Bu), a, ib, ie);  // the group's ';'
            while (m && left > 0) {
              entry(a + (unsigned)__ffs((int)m) - 1u);
              m &= m - 1;
            }
          }
          if (left > 0 && i < ie) entry(ie);  // the last entry has no ';' behind it
        }
        for (int q = 0; q < ik.n; ++q) {
          if (info_ok >> q & 1u) continue;
          if (ik.kind[q] == 'F' || ik.kind[q] == 'I') out.lv_cnt[q][row] = 0;  // NULL list: no items
          else if (ik.kind[q] != 'b') out.info[q][row] = 0.f;                   // NULL slots hold a defined value
        }
      }
    }
  }
  const int64_t wave_row0 = row - lane;
  store_valid(out.pos_valid, wave_row0, n_rows, pos_ok, lane);
  store_valid(out.qual_valid, wave_row0, n_rows, qual_ok, lane);
  for (int q = 0; q < ik.n; ++q) store_valid(out.info_valid[q], wave_row0, n_rows, (info_ok >> q & 1u) != 0, lane);
  const unsigned long long nb = __ballot(bad);
  if (lane == 0 && nb) atomicAdd(out.exceptions, (unsigned)__popcll(nb));
}

// Number=1 String / Character INFO key (info_builder.rs:152-309 builds a Utf8 column; here: dictionary ids + the dictionary, like
// FILTER): every row with a value hashes its text into the key's table -- provisional slot in ids[row], replaced by the dense id
// by k_remap_filters once k_assign_filters has numbered the new values.  counters[3] += rows WITHOUT a value (a consumer that
// groups by the key needs to know whether there is a NULL group).
__global__ __launch_bounds__(TPB) void k_info_string_ids(const uint8_t* __restrict__ text, const uint32_t* __restrict__ voff, const uint32_t* __restrict__ vlen,
                                                         const uint8_t* __restrict__ valid, const unsigned* __restrict__ n_lines_p, unsigned cap, FilterTable t,
                                                         int32_t* __restrict__ ids, int null_as_value) {
  const unsigned n = min(*n_lines_p, cap);
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  bool has = false;
  if (row < n) {
    has = (valid[row >> 3] >> (row & 7)) & 1;
    int found = 0;
    // null_as_value (a fused plan groups by this key): a row without a value takes the id of the EMPTY text -- a value no row can
    // carry ("key=" is a missing value) -- so that NULL is a group of its own, as in DataFusion's GROUP BY
    if (has || null_as_value) {
      const uint8_t* p = text + (has ? voff[row] : 0u);
      const int len = has ? (int)vlen[row] : 0;
      const unsigned long long h = fnv1a(p, len);
      int slot = (int)(h & (FILTER_SLOTS - 1));
      found = -1;
      for (int probe = 0; probe < FILTER_SLOTS; ++probe) {
        unsigned long long k = t.keys[slot];
        if (k == 0) {
          k = atomicCAS(&t.keys[slot], 0ull, h);
          if (k == 0) {
            t.text_off[slot] = has ? voff[row] : 0u;
            t.text_len[slot] = (uint32_t)len;
            found = slot;
            break;
          }
        }
        if (k == h) {
          found = slot;
          break;
        }
        slot = (slot + 1) & (FILTER_SLOTS - 1);
      }
      if (found < 0) {
        atomicExch(&t.counters[2], 1);
        found = 0;
      }
    }
    ids[row] = found;
  }
  const unsigned long long miss = __ballot(row < n && !has && !null_as_value);
  if ((threadIdx.x & 63) == 0 && miss) atomicAdd(&t.counters[3], __popcll(miss));
  // (null_as_value: counters[3] stays 0 = "no row without an id": the consumer then takes the ids without the bitmap)
}

// dense ids for FILTER lists inserted during the last parse, text copied into the persistent pool.  New lists are
// rare, so every thread scans its share of the slots and claims ids / pool space with atomics (ids are arbitrary
// but stable; names are recovered through exon_hip_vcf_parser_filters).
__global__ __launch_bounds__(256) void k_assign_filters(const uint8_t* __restrict__ text, FilterTable f) {
  for (int s = threadIdx.x; s < FILTER_SLOTS; s += 256)
    if (f.keys[s] != 0 && f.ids[s] < 0) {
      const uint32_t len = f.text_len[s], src = f.text_off[s];
      const int id = atomicAdd(&f.counters[0], 1);
      const int po = atomicAdd(&f.counters[1], (int)len);
      if (id >= EXON_HIP_MAX_GROUPS || po + (int)len > FILTER_POOL) {
        f.counters[2] = 1;
        f.ids[s] = 0;
        continue;
      }
      for (uint32_t i = 0; i < len; ++i) f.pool[po + i] = text[src + i];
      f.text_off[s] = (uint32_t)po;
      f.ids[s] = id;
    }
}

__global__ __launch_bounds__(TPB) void k_remap_filters(int32_t* __restrict__ filter_id, const unsigned* __restrict__ n_lines_p,
                                                       const int32_t* __restrict__ ids, unsigned cap) {
  const int64_t n = min(*n_lines_p, cap);
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const unsigned slot = (unsigned)filter_id[i];
    filter_id[i] = slot < (unsigned)FILTER_SLOTS ? ids[slot] : 0;
  }
}

}  // namespace

// Upload an open-addressing name table (names[i] -> id i).  bufs[0..5) receive the device allocations.
static hipError_t build_name_table(exon_hip_ctx* ctx, const char* const* names_in, int32_t n, void** bufs, NameTable* out) {
  int cap = 16;
  while (cap < 2 * n + 1) cap <<= 1;
  std::vector<uint64_t> keys((size_t)cap, 0);
  std::vector<int32_t> ids((size_t)cap, -1);
  std::vector<uint32_t> toff((size_t)cap, 0), tlen((size_t)cap, 0);
  std::string pool;
  for (int i = 0; i < n; ++i) {
    const std::string nm = names_in[i];
    const uint64_t h = fnv1a(reinterpret_cast<const uint8_t*>(nm.data()), (int)nm.size());
    int slot = (int)(h & (uint64_t)(cap - 1));
    while (keys[(size_t)slot] != 0) slot = (slot + 1) & (cap - 1);
    keys[(size_t)slot] = h;
    ids[(size_t)slot] = i;
    toff[(size_t)slot] = (uint32_t)pool.size();
    tlen[(size_t)slot] = (uint32_t)nm.size();
    pool += nm;
  }
  hipError_t e = hipSuccess;
  auto dalloc = [&](void** ptr, size_t bytes) {
    if (e == hipSuccess && !(*ptr = exon_pool_alloc(ctx, bytes))) e = hipErrorOutOfMemory;
  };
  dalloc(&bufs[0], (size_t)cap * 8);
  dalloc(&bufs[1], (size_t)cap * 4);
  dalloc(&bufs[2], (size_t)cap * 4);
  dalloc(&bufs[3], (size_t)cap * 4);
  dalloc(&bufs[4], pool.size() + 16);
  if (e == hipSuccess) e = hipMemcpy(bufs[0], keys.data(), (size_t)cap * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(bufs[1], ids.data(), (size_t)cap * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(bufs[2], toff.data(), (size_t)cap * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(bufs[3], tlen.data(), (size_t)cap * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && !pool.empty()) e = hipMemcpy(bufs[4], pool.data(), pool.size(), hipMemcpyHostToDevice);
  *out = NameTable{(const uint64_t*)bufs[0], (const int32_t*)bufs[1], (const uint32_t*)bufs[2], (const uint32_t*)bufs[3],
                   (const uint8_t*)bufs[4], cap - 1};
  return e;
}

// ------------------------------------------------------------------------------------------------------------
// ---- list-valued INFO fields ('F' / 'I'): offsets by a scan of the per-row item counts (list_kernels.h), then the items ------
// offsets[row] = exclusive prefix of cnt (block_offsets = scanned block sums), offsets[n_rows] = total; every row parses its
// items into values[offsets[row] ..]: '.' or an empty item -> NULL item (flag 0), anything unparsable -> undecided (host)
__global__ __launch_bounds__(LIST_TPB) void k_list_fill(const uint8_t* __restrict__ text, unsigned n_total, const uint32_t* __restrict__ lv_off,
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
  if ((uint64_t)first + c > cap_items) {  // comma-dense values ("AF=,,,,": one EMPTY item per byte) can exceed slab bytes / 2 + 1: host decoder
    atomicAdd(exceptions, 1u);
    return;
  }
  unsigned a = lv_off[row];
  bool bad = false;
  for (unsigned i = 0; i < c; ++i) {
    unsigned e = a;
    while (e < n_total && text[e] != ',' && text[e] != ';' && text[e] != '\t' && text[e] != '\n' && text[e] != '\r') ++e;
    const int len = (int)(e - a);
    uint32_t bits = 0;
    bool ok = false;
    if (!(len == 0 || (len == 1 && text[a] == '.'))) {
      if (kind == 'I') {
        int k = 0;
        const bool neg = text[a] == '-';
        if (neg || text[a] == '+') k = 1;
        int64_t iv = 0;
        ok = k < len && len - k <= 10;
        for (; k < len && ok; ++k) {
          const unsigned d = (unsigned)text[a + k] - '0';
          ok = d <= 9u;
          iv = iv * 10 + d;
        }
        if (neg) iv = -iv;
        ok = ok && iv >= INT32_MIN && iv <= INT32_MAX;
        bits = (uint32_t)(int32_t)iv;
      } else {
        ok = exon::dec::parse_f32(reinterpret_cast<const char*>(text + a), len, &bits);
      }
      if (!ok) bad = true;
    }
    values[first + i] = __uint_as_float(ok ? bits : 0u);
    item_flags[first + i] = ok ? 1 : 0;
    a = e + 1;
  }
  if (bad) atomicAdd(exceptions, 1u);
}
struct exon_hip_vcf_parser {
  exon_hip_ctx* ctx = nullptr;
  int64_t max_bytes = 0, max_rows = 0;
  std::string info_field;  // "name[:kind],..." as given; kinds f (default) / b
  InfoKeys ik{};
  // device state
  uint8_t* d_info_key = nullptr;
  unsigned *d_block_counts = nullptr, *d_nl = nullptr, *d_scalars = nullptr;  // scalars: [0] n_lines, [1] exceptions
  int64_t index_blocks = 0;
  unsigned index_gen = 0;
  NameTable contigs{};
  void* contig_bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  FilterTable filters{};
  ParseOut out{};
  void* out_bufs[6 + 2 * MAX_INFO] = {nullptr};
  // list kinds: per key { value offsets per row, item counts per row, Arrow offsets [rows + 1], item flags, item validity bitmap }
  void* list_bufs[5 * MAX_INFO] = {nullptr};
  unsigned* d_list_blocks = nullptr;  // per-workgroup sums of the item counts (scanned in place)
  int64_t cap_items = 0;
  unsigned* h_scalars = nullptr;  // pinned mirror of d_scalars
  FilterTable str_tables[MAX_INFO] = {};
  int null_as_value = 0;  // exon_hip_vcf_parser_set_null_key: rows without a value of a String key take the id of the empty text
  int32_t h_str_stat[MAX_INFO][2] = {{0, 0}};  // per 's' key, last slab: {dictionary overflow, rows without a value}  // kind 's': the key's value dictionary, built on the device like the FILTER dictionary
};

extern "C" {

int exon_hip_vcf_parser_create(exon_hip_ctx* ctx, const char* const* contig_names, int32_t n_contigs,
                               const char* info_field, int64_t max_bytes, exon_hip_vcf_parser** outp) {
  if (!ctx || !outp || (n_contigs > 0 && !contig_names) || max_bytes < 16)
    return fail(ctx, EXON_HIP_EINVAL, "exon_hip_vcf_parser_create: bad argument");
  if (max_bytes > 0xF0000000LL) return fail(ctx, EXON_HIP_EINVAL, "slab size must stay below 4 GiB (32-bit line offsets)");
  *outp = nullptr;
  exon_hip_vcf_parser* p = new (std::nothrow) exon_hip_vcf_parser();
  if (!p) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  p->ctx = ctx;
  p->max_bytes = max_bytes;
  p->max_rows = max_bytes / 16 + 1;  // a VCF data line has 8 fields: >= 15 bytes + newline
  p->info_field = info_field ? info_field : "";
  hipSetDevice(ctx->device);
  hipError_t e = hipSuccess;
  auto dalloc = [&](void** ptr, size_t bytes) {
    if (e == hipSuccess && !(*ptr = exon_pool_alloc(ctx, bytes))) e = hipErrorOutOfMemory;
  };
  // contig table
  e = build_name_table(ctx, contig_names, n_contigs, p->contig_bufs, &p->contigs);
  // filter table
  dalloc((void**)&p->filters.keys, FILTER_SLOTS * 8);
  dalloc((void**)&p->filters.ids, FILTER_SLOTS * 4);
  dalloc((void**)&p->filters.text_off, FILTER_SLOTS * 4);
  dalloc((void**)&p->filters.text_len, FILTER_SLOTS * 4);
  dalloc((void**)&p->filters.pool, FILTER_POOL);
  dalloc((void**)&p->filters.counters, 16);
  if (e == hipSuccess) e = hipMemset(p->filters.keys, 0, FILTER_SLOTS * 8);
  if (e == hipSuccess) e = hipMemset(p->filters.ids, 0xFF, FILTER_SLOTS * 4);
  if (e == hipSuccess) e = hipMemset(p->filters.counters, 0, 16);
  // scratch + outputs
  const int64_t nblocks = (max_bytes + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK;
  dalloc((void**)&p->d_block_counts, ((size_t)nblocks + 1) * 8);  // the line index's words + its tile counter (zeroed below)
  p->index_blocks = nblocks;
  if (e == hipSuccess && p->d_block_counts) e = hipMemset(p->d_block_counts, 0, ((size_t)nblocks + 1) * 8);
  dalloc((void**)&p->d_nl, (size_t)p->max_rows * 4);
  dalloc((void**)&p->d_scalars, 16);
  // INFO keys: "AF,DP:f,DB:b" -> names back to back + (offset, length, kind) per key
  std::string key_text;
  {
    size_t i = 0;
    const std::string& f = p->info_field;
    while (i < f.size()) {
      size_t j = f.find(',', i);
      if (j == std::string::npos) j = f.size();
      std::string item = f.substr(i, j - i);
      char kind = 'f';
      const size_t c = item.rfind(':');
      if (c != std::string::npos && c + 2 == item.size() && (item[c + 1] == 'f' || item[c + 1] == 'b' || item[c + 1] == 'i' || item[c + 1] == 'F' || item[c + 1] == 'I' || item[c + 1] == 's')) {
        kind = item[c + 1];
        item.resize(c);
      }
      if (!item.empty()) {
        if (p->ik.n == MAX_INFO) {
          exon_hip_vcf_parser_destroy(p);
          return fail(ctx, EXON_HIP_EUNSUPPORTED, "at most %d INFO fields per parser", MAX_INFO);
        }
        p->ik.off[p->ik.n] = (int)key_text.size();
        p->ik.len[p->ik.n] = (int)item.size();
        p->ik.kind[p->ik.n] = kind;
        key_text += item;
        ++p->ik.n;
      }
      i = j + 1;
    }
  }
  dalloc((void**)&p->d_info_key, key_text.size() + 16);
  if (e == hipSuccess && !key_text.empty()) e = hipMemcpy(p->d_info_key, key_text.data(), key_text.size(), hipMemcpyHostToDevice);
  p->ik.text = p->d_info_key;
  const size_t r = (size_t)p->max_rows, rb = r / 8 + 64;
  dalloc(&p->out_bufs[0], r * 4);
  dalloc(&p->out_bufs[1], r * 8);
  dalloc(&p->out_bufs[2], rb);
  dalloc(&p->out_bufs[3], r * 4);
  dalloc(&p->out_bufs[4], rb);
  dalloc(&p->out_bufs[5], r * 4);
  p->cap_items = max_bytes / 2 + 1;  // a non-empty item and its separator take at least two bytes of the slab; a slab of mostly EMPTY
                                     // items (legal: "AF=,,,,") overflows this and is decoded by the host reader (k_list_fill / k_pack_bits clamp)
  for (int q = 0; q < p->ik.n; ++q) {
    const char kind = p->ik.kind[q];
    if (kind == 'f' || kind == 'i' || kind == 's') dalloc(&p->out_bufs[6 + 2 * q], r * 4);
    dalloc(&p->out_bufs[7 + 2 * q], rb);
    if (kind == 's') {  // Number=1 String / Character: where the value text is (k_parse_lines), then dictionary ids (k_info_string_ids)
      dalloc(&p->list_bufs[5 * q + 0], r * 4);
      dalloc(&p->list_bufs[5 * q + 1], r * 4);
      FilterTable& t = p->str_tables[q];
      dalloc((void**)&t.keys, FILTER_SLOTS * 8);
      dalloc((void**)&t.ids, FILTER_SLOTS * 4);
      dalloc((void**)&t.text_off, FILTER_SLOTS * 4);
      dalloc((void**)&t.text_len, FILTER_SLOTS * 4);
      dalloc((void**)&t.pool, FILTER_POOL);
      dalloc((void**)&t.counters, 16);
      if (e == hipSuccess) e = hipMemset(t.keys, 0, FILTER_SLOTS * 8);
      if (e == hipSuccess) e = hipMemset(t.ids, 0xFF, FILTER_SLOTS * 4);
      if (e == hipSuccess) e = hipMemset(t.counters, 0, 16);
    }
    if (kind == 'F' || kind == 'I') {
      dalloc(&p->out_bufs[6 + 2 * q], (size_t)p->cap_items * 4);  // the items
      dalloc(&p->list_bufs[5 * q + 0], r * 4);
      dalloc(&p->list_bufs[5 * q + 1], r * 4);
      dalloc(&p->list_bufs[5 * q + 2], (r + 1) * 4);
      dalloc(&p->list_bufs[5 * q + 3], (size_t)p->cap_items);
      dalloc(&p->list_bufs[5 * q + 4], (size_t)p->cap_items / 8 + 64);
      if (!p->d_list_blocks) dalloc((void**)&p->d_list_blocks, (r / LIST_TPB + 2) * 4);
    }
  }
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_scalars, 16);
  if (e != hipSuccess) {
    const std::string msg = hipGetErrorString(e);
    exon_hip_vcf_parser_destroy(p);
    return fail(ctx, EXON_HIP_ENOMEM, "vcf parser allocation: %s", msg.c_str());
  }
  p->out = ParseOut{};
  p->out.chrom_id = (int32_t*)p->out_bufs[0];
  p->out.pos = (int64_t*)p->out_bufs[1];
  p->out.pos_valid = (uint8_t*)p->out_bufs[2];
  p->out.qual = (float*)p->out_bufs[3];
  p->out.qual_valid = (uint8_t*)p->out_bufs[4];
  p->out.filter_id = (int32_t*)p->out_bufs[5];
  for (int q = 0; q < p->ik.n; ++q) {
    p->out.info[q] = (float*)p->out_bufs[6 + 2 * q];
    p->out.info_valid[q] = (uint8_t*)p->out_bufs[7 + 2 * q];
    p->out.lv_off[q] = (uint32_t*)p->list_bufs[5 * q + 0];
    p->out.lv_cnt[q] = (uint32_t*)p->list_bufs[5 * q + 1];
  }
  p->out.exceptions = p->d_scalars + 1;
  *outp = p;
  return EXON_HIP_OK;
}

int exon_hip_vcf_parser_destroy(exon_hip_vcf_parser* p) {
  if (!p) return EXON_HIP_OK;
  for (void* b : p->contig_bufs)
    if (b) exon_pool_free(p->ctx, b);
  for (void* b : p->out_bufs)
    if (b) exon_pool_free(p->ctx, b);
  if (p->filters.keys) exon_pool_free(p->ctx, p->filters.keys);
  if (p->filters.ids) exon_pool_free(p->ctx, p->filters.ids);
  if (p->filters.text_off) exon_pool_free(p->ctx, p->filters.text_off);
  if (p->filters.text_len) exon_pool_free(p->ctx, p->filters.text_len);
  if (p->filters.pool) exon_pool_free(p->ctx, p->filters.pool);
  if (p->filters.counters) exon_pool_free(p->ctx, p->filters.counters);
  if (p->d_block_counts) exon_pool_free(p->ctx, p->d_block_counts);
  if (p->d_nl) exon_pool_free(p->ctx, p->d_nl);
  if (p->d_scalars) exon_pool_free(p->ctx, p->d_scalars);
  if (p->d_info_key) exon_pool_free(p->ctx, p->d_info_key);
  for (FilterTable& t : p->str_tables) {
    if (t.keys) exon_pool_free(p->ctx, t.keys);
    if (t.ids) exon_pool_free(p->ctx, t.ids);
    if (t.text_off) exon_pool_free(p->ctx, t.text_off);
    if (t.text_len) exon_pool_free(p->ctx, t.text_len);
    if (t.pool) exon_pool_free(p->ctx, t.pool);
    if (t.counters) exon_pool_free(p->ctx, t.counters);
  }
  for (void* b : p->list_bufs)
    if (b) exon_pool_free(p->ctx, b);
  if (p->d_list_blocks) exon_pool_free(p->ctx, p->d_list_blocks);
  if (p->h_scalars) hipHostFree(p->h_scalars);
  delete p;
  return EXON_HIP_OK;
}

int exon_hip_vcf_parser_parse(exon_hip_vcf_parser* p, void* stream, const uint8_t* d_text, int64_t n_bytes,
                              exon_hip_vcf_columns* cols) {
  if (!p || !cols || (n_bytes > 0 && !d_text)) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_vcf_parser_parse: NULL argument");
  exon_hip_ctx* ctx = p->ctx;
  if (n_bytes > p->max_bytes) return fail(ctx, EXON_HIP_EINVAL, "slab of %lld bytes exceeds the parser's %lld", (long long)n_bytes, (long long)p->max_bytes);
  memset(cols, 0, sizeof *cols);
  if (n_bytes == 0) return EXON_HIP_OK;
  // the kernels read aligned 16-byte groups: start at the aligned address at or below d_text and ignore the bytes before it
  const unsigned skip = (unsigned)(reinterpret_cast<uintptr_t>(d_text) & 15);
  d_text -= skip;
  n_bytes += skip;
  if (n_bytes > p->max_bytes) return fail(ctx, EXON_HIP_EINVAL, "slab of %lld bytes exceeds the parser's %lld", (long long)n_bytes, (long long)p->max_bytes);
  hipStream_t s = pick_stream(ctx, stream);
  const int nblocks = (int)((n_bytes + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK);
  launch_line_index(s, d_text, n_bytes, skip, p->d_block_counts, p->index_blocks, nblocks, &p->index_gen, p->d_nl, (unsigned)p->max_rows, p->d_scalars);
  hipLaunchKernelGGL(k_last_newline, dim3(1), dim3(1), 0, s, p->d_nl, p->d_scalars, (unsigned)p->max_rows);
  // the number of lines is bounded by n_bytes / 16 + 1 for well-formed data lines; launch for that bound
  const int64_t row_bound = std::min<int64_t>(p->max_rows, n_bytes / 16 + 1);
  const int pblocks = (int)((row_bound + TPB - 1) / TPB);
  hipLaunchKernelGGL(k_parse_lines, dim3(pblocks), dim3(TPB), 0, s, d_text, p->d_nl, p->d_scalars, p->contigs, p->filters,
                     p->ik, p->out, (unsigned)row_bound, skip, (unsigned)n_bytes);
  hipLaunchKernelGGL(k_assign_filters, dim3(1), dim3(256), 0, s, d_text, p->filters);
  hipLaunchKernelGGL(k_remap_filters, dim3(std::min(pblocks, 4096)), dim3(TPB), 0, s, p->out.filter_id, p->d_scalars, p->filters.ids, (unsigned)row_bound);
  for (int q = 0; q < p->ik.n; ++q) {  // list-valued fields: counts -> offsets -> items -> child validity
    const char kind = p->ik.kind[q];
    if (kind != 'F' && kind != 'I') continue;
    const int lblocks = (int)((row_bound + LIST_TPB - 1) / LIST_TPB);
    int32_t* offsets = (int32_t*)p->list_bufs[5 * q + 2];
    hipLaunchKernelGGL(k_list_block_sums, dim3(lblocks), dim3(LIST_TPB), 0, s, p->out.lv_cnt[q], p->d_scalars, (unsigned)row_bound, p->d_list_blocks);
    hipLaunchKernelGGL(k_list_scan_blocks, dim3(1), dim3(256), 0, s, p->d_list_blocks, lblocks, p->d_scalars + 3);
    hipLaunchKernelGGL(k_list_fill, dim3(lblocks), dim3(LIST_TPB), 0, s, d_text, (unsigned)n_bytes, p->out.lv_off[q], p->out.lv_cnt[q], p->d_list_blocks,
                       p->d_scalars, (unsigned)row_bound, (unsigned)std::min<int64_t>(p->cap_items, 0xFFFFFFFFLL), kind, offsets, p->out.info[q],
                       (uint8_t*)p->list_bufs[5 * q + 3], p->out.exceptions);
    hipLaunchKernelGGL(k_pack_bits, dim3(1024), dim3(256), 0, s, (const uint8_t*)p->list_bufs[5 * q + 3], offsets, p->d_scalars, (unsigned)row_bound,
                       (unsigned)std::min<int64_t>(p->cap_items, 0xFFFFFFFFLL), (uint8_t*)p->list_bufs[5 * q + 4]);
  }
  for (int q = 0; q < p->ik.n; ++q) {  // String / Character keys: value text -> dictionary ids
    if (p->ik.kind[q] != 's') continue;
    FilterTable& t = p->str_tables[q];
    HIP_TRY(ctx, hipMemsetAsync(t.counters + 3, 0, 4, s));
    hipLaunchKernelGGL(k_info_string_ids, dim3(pblocks), dim3(TPB), 0, s, d_text, p->out.lv_off[q], p->out.lv_cnt[q], p->out.info_valid[q], p->d_scalars, (unsigned)row_bound, t,
                       (int32_t*)p->out.info[q], p->null_as_value);
    hipLaunchKernelGGL(k_assign_filters, dim3(1), dim3(256), 0, s, d_text, t);
    hipLaunchKernelGGL(k_remap_filters, dim3(std::min(pblocks, 4096)), dim3(TPB), 0, s, (int32_t*)p->out.info[q], p->d_scalars, t.ids, (unsigned)row_bound);
    HIP_TRY(ctx, hipMemcpyAsync(p->h_str_stat[q], t.counters + 2, 8, hipMemcpyDeviceToHost, s));  // {overflow, rows without a value}
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(p->h_scalars, p->d_scalars, 12, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  const int64_t n_lines = p->h_scalars[0];
  if (n_lines > row_bound) return fail(ctx, EXON_HIP_EINVAL, "slab has %lld lines, more than its byte size allows for VCF records", (long long)n_lines);
  cols->n_rows = n_lines;
  cols->n_undecided = p->h_scalars[1];
  cols->consumed_bytes = p->h_scalars[2] > skip ? (int64_t)p->h_scalars[2] - skip : 0;
  cols->chrom_id = p->out.chrom_id;
  cols->pos = p->out.pos;
  cols->pos_valid = p->out.pos_valid;
  cols->qual = p->out.qual;
  cols->qual_valid = p->out.qual_valid;
  cols->filter_id = p->out.filter_id;
  cols->info = p->ik.n ? p->out.info[0] : nullptr;
  cols->info_valid = p->ik.n ? p->out.info_valid[0] : nullptr;
  cols->n_info = p->ik.n;
  for (int q = 0; q < p->ik.n; ++q) {
    cols->infos[q] = p->out.info[q];  // NULL for a Flag: its column IS the presence bitmap
    cols->infos_valid[q] = p->out.info_valid[q];
    cols->info_kinds[q] = p->ik.kind[q];
    cols->info_nulls[q] = p->ik.kind[q] == 's' ? p->h_str_stat[q][1] : -1;
    if (p->ik.kind[q] == 's' && p->h_str_stat[q][0]) ++cols->n_undecided;  // more distinct values than the dictionary holds: the host reader's
    if (p->ik.kind[q] == 'F' || p->ik.kind[q] == 'I') {
      cols->list_offsets[q] = (int32_t*)p->list_bufs[5 * q + 2];
      cols->list_item_valid[q] = (uint8_t*)p->list_bufs[5 * q + 4];
    }
  }
  return EXON_HIP_OK;
}

}  // extern "C"
const unsigned* exon_hip_vcf_parser_newlines(exon_hip_vcf_parser* p) { return p ? p->d_nl : nullptr; }
extern "C" {

// a device-built dictionary (FILTER lists, or the values of a String INFO key) in id order: names '\0'-separated into `buf`
static int table_names(exon_hip_ctx* ctx, const FilterTable& t, const char* what, char* buf, size_t cap, int32_t* n_names) {
  int32_t counters[4];
  HIP_TRY(ctx, hipMemcpy(counters, t.counters, 16, hipMemcpyDeviceToHost));
  if (counters[2]) return fail(ctx, EXON_HIP_EUNSUPPORTED, "more than %d distinct %s (or their text pool exhausted)", EXON_HIP_MAX_GROUPS, what);
  std::vector<unsigned long long> keys(FILTER_SLOTS);
  std::vector<int32_t> ids(FILTER_SLOTS);
  std::vector<uint32_t> toff(FILTER_SLOTS), tlen(FILTER_SLOTS);
  std::vector<uint8_t> pool((size_t)std::max(counters[1], 1));
  HIP_TRY(ctx, hipMemcpy(keys.data(), t.keys, FILTER_SLOTS * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(ids.data(), t.ids, FILTER_SLOTS * 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(toff.data(), t.text_off, FILTER_SLOTS * 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(tlen.data(), t.text_len, FILTER_SLOTS * 4, hipMemcpyDeviceToHost));
  if (counters[1] > 0) HIP_TRY(ctx, hipMemcpy(pool.data(), t.pool, (size_t)counters[1], hipMemcpyDeviceToHost));
  std::vector<std::string> names((size_t)counters[0]);
  for (int s = 0; s < FILTER_SLOTS; ++s)
    if (keys[(size_t)s] != 0 && ids[(size_t)s] >= 0 && ids[(size_t)s] < counters[0])
      names[(size_t)ids[(size_t)s]] = std::string(reinterpret_cast<const char*>(pool.data()) + toff[(size_t)s], tlen[(size_t)s]);
  size_t need = 0;
  for (const auto& nm : names) need += nm.size() + 1;
  *n_names = counters[0];
  if (buf) {
    if (need > cap) return fail(ctx, EXON_HIP_EINVAL, "name buffer too small (%zu needed)", need);
    size_t o = 0;
    for (const auto& nm : names) {
      memcpy(buf + o, nm.c_str(), nm.size() + 1);
      o += nm.size() + 1;
    }
  }
  return EXON_HIP_OK;
}
// FILTER dictionary discovered so far: names are written '\0'-separated into `buf` (id order); returns the count
int exon_hip_vcf_parser_filters(exon_hip_vcf_parser* p, char* buf, size_t cap, int32_t* n_filters) {
  if (!p || !n_filters) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_vcf_parser_filters: NULL argument");
  return table_names(p->ctx, p->filters, "FILTER lists", buf, cap, n_filters);
}
// on != 0: a row without a value of a String / Character key gets the dictionary id of the EMPTY text (no row can carry it) and
// counts as valid: NULL becomes a group key of its own for a plan that groups by the key.  Off (default): NULL stays NULL.
int exon_hip_vcf_parser_set_null_key(exon_hip_vcf_parser* p, int32_t on) {
  if (!p) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_vcf_parser_set_null_key: NULL argument");
  p->null_as_value = on ? 1 : 0;
  return EXON_HIP_OK;
}
// the value dictionary of INFO key `key` (its index in the parser's key list; kind 's') in id order
int exon_hip_vcf_parser_info_values(exon_hip_vcf_parser* p, int32_t key, char* buf, size_t cap, int32_t* n_values) {
  if (!p || !n_values) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_vcf_parser_info_values: NULL argument");
  if (key < 0 || key >= p->ik.n || p->ik.kind[key] != 's') return fail(p->ctx, EXON_HIP_EINVAL, "exon_hip_vcf_parser_info_values: key %d is not a String / Character key of this parser", key);
  return table_names(p->ctx, p->str_tables[key], "values of a String INFO key", buf, cap, n_values);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------
// FASTQ: the newline index above is all the "parsing" a histogram over quality (or sequence) lines needs.  Read r
// is lines 4r .. 4r+3; its views are byte ranges into the slab itself, consumed in place by K5's ragged / fixed
// paths (kernels.hip) -- the text is read from HBM once for the index and once for the histogram.
namespace {

// scalars: [0] n_lines (in), [1] undecided (+=), [2] consumed bytes (out)
__global__ __launch_bounds__(TPB) void k_fastq_views(const uint8_t* __restrict__ text, const unsigned* __restrict__ nl,
                                                     unsigned* __restrict__ scalars, unsigned cap_lines, int final_slab,
                                                     int32_t* __restrict__ seq_s, int32_t* __restrict__ seq_e,
                                                     int32_t* __restrict__ qual_s, int32_t* __restrict__ qual_e, int32_t* __restrict__ head_s,
                                                     int32_t* __restrict__ head_e, unsigned skip) {
  const unsigned n_lines = scalars[0];
  const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (n_lines > cap_lines) {  // the index was truncated: nothing can be trusted
    if (r == 0) atomicAdd(&scalars[1], 1u);
    return;
  }
  const unsigned n_reads = n_lines / 4;
  if (r == 0) {
    scalars[2] = n_reads ? nl[4 * n_reads - 1] + 1u : 0u;
    if (final_slab && (n_lines & 3u)) atomicAdd(&scalars[1], 1u);
  }
  if (r >= n_reads) return;
  const unsigned l0 = r ? nl[4 * r - 1] + 1u : skip;
  const unsigned e0 = nl[4 * r], e1 = nl[4 * r + 1], e2 = nl[4 * r + 2], e3 = nl[4 * r + 3];
  const bool bad = text[l0] != '@' || text[e1 + 1] != '+';
  unsigned se = e1, qe = e3;
  if (se > e0 + 1 && text[se - 1] == '\r') --se;
  if (qe > e2 + 1 && text[qe - 1] == '\r') --qe;
  seq_s[r] = (int32_t)(e0 + 1);
  seq_e[r] = (int32_t)se;
  qual_s[r] = (int32_t)(e2 + 1);
  qual_e[r] = (int32_t)qe;
  unsigned he = e0;  // the header line behind its '@', CR dropped (name + description: exon-fastq/src/array_builder.rs:68-102)
  if (he > l0 + 1 && text[he - 1] == '\r') --he;
  head_s[r] = (int32_t)(l0 + 1);
  head_e[r] = (int32_t)he;
  if (bad) atomicAdd(&scalars[1], 1u);
}

}  // namespace

struct exon_hip_fastq_parser {
  exon_hip_ctx* ctx = nullptr;
  int64_t max_bytes = 0, max_lines = 0;
  unsigned *d_block_counts = nullptr, *d_nl = nullptr, *d_scalars = nullptr;
  int64_t index_blocks = 0;
  unsigned index_gen = 0;
  int32_t* d_views = nullptr;  // 6 arrays of max_lines / 4 + 1
  unsigned* h_scalars = nullptr;
};

extern "C" {

int exon_hip_fastq_parser_create(exon_hip_ctx* ctx, int64_t max_bytes, exon_hip_fastq_parser** outp) {
  if (!ctx || !outp || max_bytes < 16) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_fastq_parser_create: bad argument");
  if (max_bytes > 0x7FFF0000LL) return fail(ctx, EXON_HIP_EINVAL, "slab size must stay below 2 GiB (32-bit views)");
  *outp = nullptr;
  exon_hip_fastq_parser* p = new (std::nothrow) exon_hip_fastq_parser();
  if (!p) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  p->ctx = ctx;
  p->max_bytes = max_bytes;
  p->max_lines = max_bytes / 4 + 8;  // records of >= 16 bytes; denser text is handed back to the host decoder
  hipSetDevice(ctx->device);
  const int64_t nblocks = (max_bytes + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK;
  const size_t per = (size_t)(p->max_lines / 4 + 1);
  hipError_t e = hipSuccess;
  auto dalloc = [&](void** ptr, size_t bytes) {
    if (e == hipSuccess && !(*ptr = exon_pool_alloc(ctx, bytes))) e = hipErrorOutOfMemory;
  };
  dalloc((void**)&p->d_block_counts, ((size_t)nblocks + 1) * 8);  // the line index's words + its tile counter (zeroed below)
  p->index_blocks = nblocks;
  if (e == hipSuccess && p->d_block_counts) e = hipMemset(p->d_block_counts, 0, ((size_t)nblocks + 1) * 8);
  dalloc((void**)&p->d_nl, (size_t)p->max_lines * 4);
  dalloc((void**)&p->d_scalars, 16);
  dalloc((void**)&p->d_views, per * 6 * 4);
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_scalars, 16);
  if (e != hipSuccess) {
    const std::string msg = hipGetErrorString(e);
    exon_hip_fastq_parser_destroy(p);
    return fail(ctx, EXON_HIP_ENOMEM, "fastq parser allocation: %s", msg.c_str());
  }
  *outp = p;
  return EXON_HIP_OK;
}

int exon_hip_fastq_parser_destroy(exon_hip_fastq_parser* p) {
  if (!p) return EXON_HIP_OK;
  if (p->d_block_counts) exon_pool_free(p->ctx, p->d_block_counts);
  if (p->d_nl) exon_pool_free(p->ctx, p->d_nl);
  if (p->d_scalars) exon_pool_free(p->ctx, p->d_scalars);
  if (p->d_views) exon_pool_free(p->ctx, p->d_views);
  if (p->h_scalars) hipHostFree(p->h_scalars);
  delete p;
  return EXON_HIP_OK;
}

int exon_hip_fastq_parser_parse(exon_hip_fastq_parser* p, void* stream, const uint8_t* d_text, int64_t n_bytes,
                                int32_t final_slab, exon_hip_fastq_views* views) {
  if (!p || !views || (n_bytes > 0 && !d_text)) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_fastq_parser_parse: NULL argument");
  exon_hip_ctx* ctx = p->ctx;
  if (n_bytes > p->max_bytes) return fail(ctx, EXON_HIP_EINVAL, "slab of %lld bytes exceeds the parser's %lld", (long long)n_bytes, (long long)p->max_bytes);
  memset(views, 0, sizeof *views);
  if (n_bytes == 0) return EXON_HIP_OK;
  // the kernels read aligned 16-byte groups: start at the aligned address at or below d_text and ignore the bytes before it
  const unsigned skip = (unsigned)(reinterpret_cast<uintptr_t>(d_text) & 15);
  d_text -= skip;
  n_bytes += skip;
  if (n_bytes > p->max_bytes) return fail(ctx, EXON_HIP_EINVAL, "slab of %lld bytes exceeds the parser's %lld", (long long)n_bytes, (long long)p->max_bytes);
  hipStream_t s = pick_stream(ctx, stream);
  const int nblocks = (int)((n_bytes + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK);
  const size_t per = (size_t)(p->max_lines / 4 + 1);
  int32_t* v = p->d_views;
  launch_line_index(s, d_text, n_bytes, skip, p->d_block_counts, p->index_blocks, nblocks, &p->index_gen, p->d_nl, (unsigned)p->max_lines, p->d_scalars);
  const int64_t read_bound = std::min<int64_t>((int64_t)per, n_bytes / 4 + 1);  // a record holds 4 newlines
  hipLaunchKernelGGL(k_fastq_views, dim3((unsigned)((read_bound + TPB - 1) / TPB)), dim3(TPB), 0, s, d_text, p->d_nl, p->d_scalars,
                     (unsigned)p->max_lines, (int)final_slab, v, v + per, v + 2 * per, v + 3 * per, v + 4 * per, v + 5 * per, skip);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(p->h_scalars, p->d_scalars, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  views->n_undecided = p->h_scalars[1];
  views->n_reads = p->h_scalars[0] > (unsigned)p->max_lines ? 0 : p->h_scalars[0] / 4;
  views->consumed_bytes = p->h_scalars[2] > skip ? (int64_t)p->h_scalars[2] - skip : 0;
  views->text_base = d_text;
  views->seq_start = v;
  views->seq_end = v + per;
  views->qual_start = v + 2 * per;
  views->qual_end = v + 3 * per;
  views->head_start = v + 4 * per;
  views->head_end = v + 5 * per;
  return EXON_HIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------
// SAM text: the first six tab-separated fields of every alignment line -> the BAM device layout (flag, mapq, reference
// id, start, end).  Field rules of host/formats.h's SAM reader (schema of exon-sam/src/schema_builder.rs:371-402, same
// columns as BAM): RNAME through the header's @SQ order ('*' or unknown -> NULL), POS 0 -> NULL, MAPQ 255 -> NULL,
// end = POS + (sum of M/D/N/=/X lengths) - 1.  Lines the device cannot decide (fewer than 6 fields, non-digits in a
// numeric field, header or empty lines in the middle) are counted: the caller decodes on the host instead.
namespace {

struct SamOut {
  int32_t* flag;
  uint8_t* mapq;
  uint8_t* mapq_valid;
  int32_t* ref_id;
  uint8_t* ref_valid;
  int64_t* start;
  int64_t* end;
  uint8_t* pos_valid;
};

__device__ __forceinline__ bool parse_uint(const uint8_t* text, unsigned b, unsigned e, int64_t* out) {
  if (e <= b || e - b > 18) return false;
  int64_t v = 0;
  for (unsigned i = b; i < e; ++i) {
    const uint8_t c = text[i];
    if (c < '0' || c > '9') return false;
    v = v * 10 + (c - '0');
  }
  *out = v;
  return true;
}

__global__ __launch_bounds__(TPB) void k_parse_sam_lines(const uint8_t* __restrict__ text, const unsigned* __restrict__ nl_pos,
                                                         unsigned* __restrict__ scalars, NameTable refs, SamOut out, unsigned cap,
                                                         unsigned skip) {
  const int64_t n_rows = min(scalars[0], cap);
  const int64_t row = (int64_t)blockIdx.x * TPB + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool mq_ok = false, ref_ok = false, pos_ok = false, bad = false;
  if (row < n_rows) {
    const unsigned begin = row ? nl_pos[row - 1] + 1 : skip;
    unsigned end = nl_pos[row];
    if (end > begin && text[end - 1] == '\r') --end;
    unsigned fs[7];
    int nf = 0;
    fs[0] = begin;
    for (unsigned i = begin; i < end && nf < 6; ++i)
      if (text[i] == '\t') fs[++nf] = i + 1;
    auto fbeg = [&](int f) { return fs[f]; };
    auto fend = [&](int f) { return f < nf ? fs[f + 1] - 1 : end; };
    int64_t flag = 0, pos1 = 0, mapq = 0;
    if (begin == end || text[begin] == '@' || nf < 5) {
      bad = true;
    } else if (!parse_uint(text, fbeg(1), fend(1), &flag) || !parse_uint(text, fbeg(3), fend(3), &pos1) ||
               !parse_uint(text, fbeg(4), fend(4), &mapq) || flag > 0xFFFF || mapq > 255 || pos1 > 0x7FFFFFFF) {
      bad = true;
    } else {
      // RNAME
      int id = -1;
      const unsigned rb = fbeg(2), re = fend(2);
      const int len = (int)(re - rb);
      if (!(len == 1 && text[rb] == '*')) {
        const uint64_t h = fnv1a(text + rb, len);
        int slot = (int)(h & (uint64_t)refs.mask);
        for (int probe = 0; probe <= refs.mask; ++probe) {
          const uint64_t k = refs.keys[slot];
          if (k == 0) break;
          if (k == h && (int)refs.text_len[slot] == len) {
            bool same = true;
            for (int i = 0; i < len && same; ++i) same = refs.pool[refs.text_off[slot] + i] == text[rb + i];
            if (same) {
              id = refs.ids[slot];
              break;
            }
          }
          slot = (slot + 1) & refs.mask;
        }
      }
      // CIGAR: reference length
      int64_t ref_len = 0, num = 0;
      const unsigned cb = fbeg(5), ce = fend(5);
      if (!(ce - cb == 1 && text[cb] == '*'))
        for (unsigned i = cb; i < ce; ++i) {
          const uint8_t ch = text[i];
          if (ch >= '0' && ch <= '9') num = num * 10 + (ch - '0');
          else {
            if (ch == 'M' || ch == 'D' || ch == 'N' || ch == '=' || ch == 'X') ref_len += num;
            num = 0;
          }
        }
      out.flag[row] = (int32_t)flag;
      out.mapq[row] = (uint8_t)mapq;
      out.ref_id[row] = id;
      mq_ok = mapq != 255;
      ref_ok = id >= 0;
      pos_ok = pos1 >= 1;
      out.start[row] = pos_ok ? pos1 : 0;
      out.end[row] = pos_ok ? pos1 + ref_len - 1 : 0;
    }
  }
  const int64_t row0 = ((int64_t)blockIdx.x * TPB + threadIdx.x) - lane;
  store_valid(out.mapq_valid, row0, n_rows, mq_ok, lane);
  store_valid(out.ref_valid, row0, n_rows, ref_ok, lane);
  store_valid(out.pos_valid, row0, n_rows, pos_ok, lane);
  const unsigned long long bm = __ballot(bad);
  if (lane == 0 && bm) atomicAdd(&scalars[1], (unsigned)__popcll(bm));
}

}  // namespace

struct exon_hip_sam_parser {
  exon_hip_ctx* ctx = nullptr;
  int64_t max_bytes = 0, max_rows = 0;
  unsigned *d_block_counts = nullptr, *d_nl = nullptr, *d_scalars = nullptr;
  int64_t index_blocks = 0;
  unsigned index_gen = 0;
  NameTable refs{};
  void* ref_bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  void* bufs[8] = {nullptr};
  SamOut out{};
  unsigned* h_scalars = nullptr;
};
const unsigned* exon_hip_sam_parser_newlines(exon_hip_sam_parser* p) { return p ? p->d_nl : nullptr; }

extern "C" {

int exon_hip_sam_parser_create(exon_hip_ctx* ctx, const char* const* ref_names, int32_t n_refs, int64_t max_bytes,
                               exon_hip_sam_parser** outp) {
  if (!ctx || !outp || (n_refs > 0 && !ref_names) || max_bytes < 16) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_sam_parser_create: bad argument");
  if (max_bytes > 0xF0000000LL) return fail(ctx, EXON_HIP_EINVAL, "slab size must stay below 4 GiB (32-bit line offsets)");
  *outp = nullptr;
  exon_hip_sam_parser* p = new (std::nothrow) exon_hip_sam_parser();
  if (!p) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  p->ctx = ctx;
  p->max_bytes = max_bytes;
  p->max_rows = max_bytes / 12 + 1;  // 11 fields: >= 21 bytes + newline; be generous
  hipSetDevice(ctx->device);
  hipError_t e = build_name_table(ctx, ref_names, n_refs, p->ref_bufs, &p->refs);
  auto dalloc = [&](void** ptr, size_t bytes) {
    if (e == hipSuccess && !(*ptr = exon_pool_alloc(ctx, bytes))) e = hipErrorOutOfMemory;
  };
  const int64_t nblocks = (max_bytes + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK;
  const size_t r = (size_t)p->max_rows, rb = r / 8 + 64;
  dalloc((void**)&p->d_block_counts, ((size_t)nblocks + 1) * 8);  // the line index's words + its tile counter (zeroed below)
  p->index_blocks = nblocks;
  if (e == hipSuccess && p->d_block_counts) e = hipMemset(p->d_block_counts, 0, ((size_t)nblocks + 1) * 8);
  dalloc((void**)&p->d_nl, r * 4);
  dalloc((void**)&p->d_scalars, 16);
  dalloc(&p->bufs[0], r * 4);
  dalloc(&p->bufs[1], r + 64);
  dalloc(&p->bufs[2], rb);
  dalloc(&p->bufs[3], r * 4);
  dalloc(&p->bufs[4], rb);
  dalloc(&p->bufs[5], r * 8);
  dalloc(&p->bufs[6], r * 8);
  dalloc(&p->bufs[7], rb);
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_scalars, 16);
  if (e != hipSuccess) {
    const std::string msg = hipGetErrorString(e);
    exon_hip_sam_parser_destroy(p);
    return fail(ctx, EXON_HIP_ENOMEM, "sam parser allocation: %s", msg.c_str());
  }
  p->out = SamOut{(int32_t*)p->bufs[0], (uint8_t*)p->bufs[1], (uint8_t*)p->bufs[2], (int32_t*)p->bufs[3],
                  (uint8_t*)p->bufs[4], (int64_t*)p->bufs[5], (int64_t*)p->bufs[6], (uint8_t*)p->bufs[7]};
  *outp = p;
  return EXON_HIP_OK;
}

int exon_hip_sam_parser_destroy(exon_hip_sam_parser* p) {
  if (!p) return EXON_HIP_OK;
  for (void* b : p->ref_bufs) exon_pool_free(p->ctx, b);
  for (void* b : p->bufs) exon_pool_free(p->ctx, b);
  exon_pool_free(p->ctx, p->d_block_counts);
  exon_pool_free(p->ctx, p->d_nl);
  exon_pool_free(p->ctx, p->d_scalars);
  if (p->h_scalars) hipHostFree(p->h_scalars);
  delete p;
  return EXON_HIP_OK;
}

int exon_hip_sam_parser_parse(exon_hip_sam_parser* p, void* stream, const uint8_t* d_text, int64_t n_bytes, exon_hip_bam_columns* cols) {
  if (!p || !cols || (n_bytes > 0 && !d_text)) return fail(p ? p->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_sam_parser_parse: NULL argument");
  exon_hip_ctx* ctx = p->ctx;
  memset(cols, 0, sizeof *cols);
  if (n_bytes == 0) return EXON_HIP_OK;
  const unsigned skip = (unsigned)(reinterpret_cast<uintptr_t>(d_text) & 15);
  d_text -= skip;
  n_bytes += skip;
  if (n_bytes > p->max_bytes) return fail(ctx, EXON_HIP_EINVAL, "slab of %lld bytes exceeds the parser's %lld", (long long)n_bytes, (long long)p->max_bytes);
  hipStream_t s = pick_stream(ctx, stream);
  const int nblocks = (int)((n_bytes + BYTES_PER_BLOCK - 1) / BYTES_PER_BLOCK);
  launch_line_index(s, d_text, n_bytes, skip, p->d_block_counts, p->index_blocks, nblocks, &p->index_gen, p->d_nl, (unsigned)p->max_rows, p->d_scalars);
  hipLaunchKernelGGL(k_last_newline, dim3(1), dim3(1), 0, s, p->d_nl, p->d_scalars, (unsigned)p->max_rows);
  const int64_t row_bound = std::min<int64_t>(p->max_rows, n_bytes / 12 + 1);
  const int pblocks = (int)((row_bound + TPB - 1) / TPB);
  hipLaunchKernelGGL(k_parse_sam_lines, dim3(pblocks), dim3(TPB), 0, s, d_text, p->d_nl, p->d_scalars, p->refs, p->out, (unsigned)row_bound, skip);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(p->h_scalars, p->d_scalars, 12, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  const int64_t n_lines = p->h_scalars[0];
  cols->n_rows = n_lines;
  cols->n_undecided = p->h_scalars[1] + (n_lines > row_bound ? 1 : 0);
  cols->consumed_bytes = p->h_scalars[2] > skip ? (int64_t)p->h_scalars[2] - skip : 0;
  cols->flag = p->out.flag;
  cols->mapq = p->out.mapq;
  cols->mapq_valid = p->out.mapq_valid;
  cols->ref_id = p->out.ref_id;
  cols->ref_valid = p->out.ref_valid;
  cols->start = p->out.start;
  cols->end = p->out.end;
  cols->pos_valid = p->out.pos_valid;
  return EXON_HIP_OK;
}

}  // extern "C"
