// gzip_stream.hip -- plain gzip (RFC 1952 members around ONE RFC 1951 DEFLATE stream each; NOT BGZF) inflated ON THE GPU.
//
// Reference path replaced: the `else` arm of the reference's FASTQ / VCF openers -- a file that carries the gzip magic but no
// BGZF "BC" extra field goes through `file_compression_type.convert_stream` (async-compression's GzipDecoder over the object
// store's byte stream): exon-core/src/datasources/fastq/file_opener.rs:79-92 (`is_bgzip_valid_header` decides), the same shape in
// vcf/file_opener/unindex_file_opener.rs:62-92.  Third-party arithmetic: RFC 1951 / 1952 (flate2 / miniz_oxide under
// async-compression; zlib is the byte-exact checker in tests/test_gpu_gzip_stream.py).
//
// A gzip member is one DEFLATE stream: no block sizes anywhere, and every block may copy from the 32 KiB in front of it.  The
// decode is made parallel the way pugz / rapidgzip do it on CPU threads, restated for one WAVEFRONT per chunk:
//   1. k_gz_decode: the compressed bytes of a slab are cut into chunks (64 KiB by default).  Chunk 0 starts at a position the
//      caller knows (the member header, or the block boundary the previous slab stopped at).  Every other chunk SEARCHES: 64 lanes
//      test 64 consecutive bit offsets for a dynamic-block header (BFINAL = 0, BTYPE = 2, HLIT / HDIST in range, a COMPLETE
//      code-length code), and each survivor is decoded for real -- header, tables, symbols, block after block up to the first
//      block boundary at or behind the next chunk's nominal start.  Anything zlib would reject kills the candidate and the search
//      goes on one bit later.  The window in front of a chunk is unknown, so the output is 16-bit SYMBOLS: a byte, or
//      256 + k = "byte k of the 32 KiB in front of this chunk".  Copies of markers stay markers.
//   2. the host walks the chunks in order (a 32-byte record each): chunk i + 1 must have started exactly where chunk i stopped.
//      That is the proof -- by induction from chunk 0 -- that every accepted chunk decoded real blocks; a chunk that started
//      somewhere else (a header-shaped pattern inside compressed data that survived its whole range, or a range without a dynamic
//      header: stored / fixed blocks) is decoded again from the proven position.
//   3. k_gz_compose / k_gz_groups / k_gz_windows: the 32 KiB window in front of every chunk.  A chunk's last 32 Ki symbols are a
//      gather map over the window in front of it; maps compose, so groups of 32 chunks compose theirs in parallel, one workgroup
//      chains the groups, and the groups then resolve their chunks' windows in parallel (a sequential walk would be ~4 us x
//      chunks; this is ~0.1 ms per slab).
//   4. k_gz_emit: symbols -> bytes at their final offsets (exclusive scan of the chunks' sizes), markers through the chunk's
//      window.  k_gz_crc: CRC-32 of the output in pieces that never straddle a member end; the host chains them
//      (x^(8 n) mod P) against every member's trailer, ISIZE too.
// Whatever does not prove (a corrupt stream, a block larger than a slab, more than 8 member ends in one chunk) fails the call:
// the caller decodes the file on the host, which reports the error the reference's decoder would.
// Test infrastructure is elsewhere (tests/test_gpu_gzip_stream.py: zlib equality over strategies / levels / block types /
// multi-member files / planted headers, a 3000-stream fuzz); nothing here calls zlib.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "internal.h"

namespace {

constexpr int LIT_BITS = 9, DIST_BITS = 8, CL_BITS = 7;
constexpr uint32_t E_LIT = 1u << 8, E_EOB = 1u << 9, E_INVALID = 1u << 10;
// LDS per wavefront decides how many chunks a CU decodes at once, and the kernel waits (LDS lookups, the window's reads) for about
// half of its cycles (profiles/r6_gz_decode_pmc_*.txt): 8988 bytes = 17 waves per CU at first; a 1 Ki-symbol ring, the
// code-length table laid over the distance table and the code lengths over the literal/length table make it 6108 = 26.
constexpr int GZ_WAVES_PER_CU = 26;  // resident wavefronts of k_gz_decode per CU (160 KiB of LDS / 6108 bytes)
constexpr int RING = 1024;           // symbols of recent output kept in LDS per wavefront
constexpr uint32_t RM = RING - 1;
constexpr uint32_t DRAIN = 256;      // symbols per drain step (64 lanes x 8 bytes)
constexpr uint32_t NEAR = DRAIN + 264;  // sources at most this far back are read from the ring; everything older has been drained
static_assert(RING >= (int)NEAR + 258, "a match may not overwrite ring entries that count as near");
constexpr int WIN = 32768;
constexpr int MAX_MEMBER_ENDS = 8;   // member trailers one chunk may cross
constexpr uint32_t GAP_CAP = 8192;   // symbols a batched gap decode may produce (a gap is what the header search skipped: an empty stored block, a short fixed block)
constexpr int SPARE_REGIONS = 64;    // scratch regions beyond one per chunk: a repair decodes the gap in front of a chunk into one

enum : uint32_t {
  GZ_OK = 0,
  GZ_NOT_FOUND = 1,      // search: no block start in the chunk's range
  GZ_BAD_BTYPE = 2,
  GZ_BAD_STORED = 3,
  GZ_BAD_LENGTHS = 4,
  GZ_BAD_CODE = 5,
  GZ_BAD_DISTANCE = 6,
  GZ_SYM_OVERFLOW = 7,   // more symbols than the chunk's scratch holds: the call is repeated with a larger ratio
  GZ_BAD_HEADER = 8,     // member header: magic / method / reserved flags
  GZ_MEMBER_OVERFLOW = 9,
  GZ_TRUNCATED = 10,
};
enum : uint32_t { START_SEARCH = 0, START_BLOCK = 1, START_MEMBER = 2 };
enum : uint32_t { F_EXHAUSTED = 1, F_STREAM_END = 2, F_AT_MEMBER = 4 /* end_bit is a member header (the input ended inside it) */ };

struct GzTask {        // one wavefront's work
  uint64_t bit;        // where to start (START_SEARCH: where the search starts)
  uint64_t stop;       // decode up to the first block boundary at or behind this bit (~0: to the end of the input); the search ends here too
  uint32_t kind;
  uint32_t region;     // which region of the symbol scratch / which result record
};
struct GzChunk {         // 40 bytes
  uint64_t start_bit;    // where the accepted decode began
  uint64_t end_bit;      // the block boundary it stopped at
  uint32_t n_out;        // symbols
  uint32_t status, flags, n_members;
  uint32_t ticks, slot;  // diagnostics (EXON_HIP_GZ_TRACE): 100 MHz ticks this wavefront spent on the task, when it started
};
struct GzMember {        // a member trailer crossed by a chunk: `out_off` symbols of the chunk belong to the member that ends
  uint32_t out_off, crc, isize, pad;
};

struct Lds {
  union {  // the code lengths of a header are dead once the literal/length table -- built LAST, from lengths its builder has in registers -- is written
    uint32_t lit_lut[1 << LIT_BITS];
    uint8_t lens[320];
  };
  union {  // the code-length code is dead once the lengths are read; the distance table is built after that
    uint32_t dist_lut[1 << DIST_BITS];
    uint32_t cl_lut[1 << CL_BITS];
  };
  uint16_t ring[RING];
  uint16_t lit_sym[288], dist_sym[32], cl_sym[20];
  uint16_t count[3][16], first[3][16], offs[3][16];
  uint8_t cl_lens[20];
};
enum { C_LIT = 0, C_DIST = 1, C_CL = 2 };
__shared__ __attribute__((aligned(16))) Lds g_lds;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uniu((uint32_t)(v >> 32)) << 32) | uniu((uint32_t)v); }
template <class T>
__device__ __forceinline__ T* unip(T* p) {
  return reinterpret_cast<T*>(uni64(reinterpret_cast<uint64_t>(p)));
}
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// RFC 1951 section 3.2.5 in closed form
__device__ __forceinline__ void length_code(int s, uint32_t* base, int* extra) {
  if (s < 8) {
    *base = 3u + (uint32_t)s;
    *extra = 0;
  } else if (s == 28) {
    *base = 258u;
    *extra = 0;
  } else {
    const int e = (s - 4) >> 2;
    *base = 3u + ((4u + (uint32_t)(s & 3)) << e);
    *extra = e;
  }
}
__device__ __forceinline__ void distance_code(int d, uint32_t* base, int* extra) {
  if (d < 4) {
    *base = 1u + (uint32_t)d;
    *extra = 0;
  } else {
    const int e = (d - 2) >> 1;
    *base = 1u + ((2u + (uint32_t)(d & 1)) << e);
    *extra = e;
  }
}
__device__ __forceinline__ int cl_order(int i) {
  constexpr uint64_t LO = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
  constexpr uint64_t HI = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
  return i < 12 ? (int)((LO >> (5 * i)) & 31) : (int)((HI >> (5 * (i - 12))) & 31);
}

// table entry of symbol s (without the code length): value << 16 | extra bits << 4 | flags
__device__ __forceinline__ uint32_t entry_for(int which, int s) {
  if (which == C_CL) return (uint32_t)s << 16;
  uint32_t base;
  int extra;
  if (which == C_LIT) {
    if (s < 256) return E_LIT | ((uint32_t)s << 16);
    if (s == 256) return E_EOB;
    if (s > 285) return E_INVALID;
    length_code(s - 257, &base, &extra);
  } else {
    if (s > 29) return E_INVALID;
    distance_code(s, &base, &extra);
  }
  return (base << 16) | ((uint32_t)extra << 4);
}

// ---- bit reader: 64-bit positions over a dword-aligned buffer; a 256-byte window lives in one VGPR (a dword per lane) ----------
struct Bits {
  const uint32_t* base;
  uint32_t widx;  // next dword to take
  uint32_t cur;   // per lane: base[(widx & ~63) + lane]
  uint64_t buf;
  int cnt;
  __device__ __forceinline__ void init(const uint32_t* b, uint64_t bitpos) {
    base = b;
    widx = (uint32_t)(bitpos >> 5);
    cur = base[(widx & ~63u) + lane_id()];
    buf = 0;
    cnt = 0;
    refill();
    refill();
    const int skip = (int)(bitpos & 31);
    buf >>= skip;
    cnt -= skip;
  }
  __device__ __forceinline__ void refill() {
    if (cnt <= 32) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(widx & 63u));
      buf |= (uint64_t)d << cnt;
      cnt += 32;
      ++widx;
      if ((widx & 63u) == 0) cur = base[widx + lane_id()];
    }
  }
  __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
  __device__ __forceinline__ void drop(int n) {
    buf >>= n;
    cnt -= n;
  }
  __device__ __forceinline__ uint32_t take(int n) {
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  __device__ __forceinline__ uint64_t pos() const { return (uint64_t)widx * 32u - (uint64_t)cnt; }
};

// Build code `which` from lens[0..n) (LDS): per-length counts, canonical order, first-level table.  0: zlib would reject the set
// (over-subscribed; incomplete unless it is the one-code case of a literal/length or distance code -- inflate_table's rule).
__device__ __noinline__ int build_code(int which, const uint8_t* lens, int n) {
  which = uni(which);
  n = uni(n);
  Lds& L = g_lds;
  uint32_t* lut = which == C_LIT ? L.lit_lut : which == C_DIST ? L.dist_lut : L.cl_lut;
  uint16_t* sym = which == C_LIT ? L.lit_sym : which == C_DIST ? L.dist_sym : L.cl_sym;
  const int bits = which == C_LIT ? LIT_BITS : which == C_DIST ? DIST_BITS : CL_BITS;
  const int lane = (int)lane_id();
  int myl[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int s = k * 64 + lane;
    myl[k] = s < n ? (int)lens[s] : 0;
  }
  wave_fence();  // (the literal/length table lies over the lengths just read)
  for (int i = lane; i < (1 << bits); i += 64) lut[i] = 0;
  int cnt_l = 0;  // lane q: symbols of length q
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (k * 64 >= n) break;
#pragma unroll 1
    for (int q = 1; q < 16; ++q) {
      const int c = __popcll(__ballot(myl[k] == q));
      if (lane == q) cnt_l += c;
    }
  }
  int left = 1, total = 0, code = 0, off = 0, first_l = 0, offs_l = 0, maxlen = 0;
#pragma unroll 1
  for (int q = 1; q < 16; ++q) {
    const int rq = __builtin_amdgcn_readlane(cnt_l, q);
    code <<= 1;
    if (lane == q) {
      first_l = code;
      offs_l = off;
    }
    code += rq;
    off += rq;
    left = (left << 1) - rq;
    if (left < 0) return 0;
    total += rq;
    if (rq) maxlen = q;
  }
  if (total == 0) {
    if (which != C_DIST) return 0;  // (a block without distance codes is legal: any match in it is then an invalid code)
  } else if (left > 0 && (which == C_CL || maxlen != 1)) {
    return 0;
  }
  if (lane < 16) {
    L.count[which][lane] = (uint16_t)cnt_l;
    L.first[which][lane] = (uint16_t)first_l;
    L.offs[which][lane] = (uint16_t)offs_l;
  }
  wave_fence();
  int seen_l = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (k * 64 >= n) break;
    const int s = k * 64 + lane;
    const int l = myl[k];
    int rank = 0;
#pragma unroll 1
    for (int q = 1; q < 16; ++q) {
      const unsigned long long m = __ballot(l == q);
      const int sq = __builtin_amdgcn_readlane(seen_l, q);
      if (l == q) rank = sq + __popcll(m & lt);
      if (lane == q) seen_l += __popcll(m);
    }
    if (l > 0) {
      const int c = (int)L.first[which][l] + rank;
      sym[(int)L.offs[which][l] + rank] = (uint16_t)s;
      if (l <= bits) {
        const unsigned rev = __brev((unsigned)c) >> (32 - l);
        const uint32_t e = entry_for(which, s) | (uint32_t)l;
        for (unsigned k2 = rev; k2 < (1u << bits); k2 += 1u << l) lut[k2] = e;
      }
    }
  }
  wave_fence();
  return 1;
}

// a code longer than the first-level table (or one that does not exist): all 15 candidate lengths at once, lane q takes length q
__device__ __noinline__ int decode_long(int which, uint32_t bits) {
  which = uni(which);
  bits = uniu(bits);
  const Lds& L = g_lds;
  const uint16_t* sym = which == C_LIT ? L.lit_sym : which == C_DIST ? L.dist_sym : L.cl_sym;
  const int len = (int)lane_id() & 15;
  const uint32_t code = len ? __brev(bits) >> (32 - len) : 0u;
  const uint32_t rel = code - (uint32_t)L.first[which][len];
  const bool hit = len != 0 && rel < (uint32_t)L.count[which][len];
  const uint32_t m = (uint32_t)__ballot(hit) & 0xFFFEu;
  if (m == 0) return -1;
  const int l = __ffs((int)m) - 1;
  const int idx = __builtin_amdgcn_readlane((int)((uint32_t)L.offs[which][len] + rel), l);
  return ((int)sym[idx] << 8) | l;
}
template <int WHICH>
__device__ __forceinline__ uint32_t decode_symbol(Bits& br) {
  const Lds& L = g_lds;
  const uint32_t* lut = WHICH == C_LIT ? L.lit_lut : WHICH == C_DIST ? L.dist_lut : L.cl_lut;
  constexpr int BITS = WHICH == C_LIT ? LIT_BITS : WHICH == C_DIST ? DIST_BITS : CL_BITS;
  uint32_t e = uniu(lut[br.peek(BITS)]);
  if (__builtin_expect((e & 15u) == 0, 0)) {
    const int r = uni(decode_long(WHICH, (uint32_t)br.buf));
    e = r < 0 ? (uint32_t)E_INVALID : entry_for(WHICH, r >> 8) | (uint32_t)(r & 255);
  }
  br.drop((int)(e & 15u));
  return e;
}

// ---- output cursor of one chunk: the last symbols in the LDS ring, everything older in the chunk's scratch region ------------------
struct Out {
  uint16_t* sym;     // the chunk's region of the symbol scratch
  uint32_t cap;      // symbols it holds (a multiple of DRAIN)
  uint32_t pos;      // symbols produced
  uint32_t drained;  // [0, drained) are in `sym`; a multiple of DRAIN
};
__device__ __forceinline__ bool drain_full(Out& o) {  // false: the scratch region is full
  while (o.pos - o.drained >= DRAIN) {
    if (o.drained + DRAIN > o.cap) return false;
    wave_fence();
    const uint2 v = *reinterpret_cast<const uint2*>(&g_lds.ring[(o.drained & RM) + lane_id() * 4u]);
    *reinterpret_cast<uint2*>(&o.sym[o.drained + lane_id() * 4u]) = v;
    o.drained += DRAIN;
  }
  return true;
}
__device__ __forceinline__ bool drain_rest(Out& o) {
  if (o.drained >= o.pos) return true;  // (a rolled-back block may have drained beyond the position the chain returns to)
  if (!drain_full(o)) return false;
  if (o.pos > o.cap) return false;
  wave_fence();
  for (uint32_t i = o.drained + lane_id(); i < o.pos; i += 64) o.sym[i] = g_lds.ring[i & RM];
  wave_fence();
  return true;
}
// `len` symbols from `dist` back.  Sources in front of the chunk become markers; sources of the last NEAR symbols are in the ring,
// older ones in the scratch region (drain_full keeps pos - drained < DRAIN + 258, so everything older than NEAR has been stored).
__device__ __forceinline__ void copy_match(Out& o, uint32_t len, uint32_t dist) {
  wave_fence();
  const int32_t src0 = (int32_t)o.pos - (int32_t)dist;
  for (uint32_t j = lane_id(); j < len; j += 64) {
    const uint32_t jj = dist < len ? j % dist : j;  // an overlapping match repeats its period: every source lies below pos
    const int32_t idx = src0 + (int32_t)jj;
    uint16_t v;
    if (idx < 0) v = (uint16_t)(256 + WIN + idx);
    else if ((uint32_t)idx + NEAR >= o.pos) v = g_lds.ring[(uint32_t)idx & RM];
#ifdef EXON_GZ_NOFAR  // (timing experiment only: what the far sources' global round trips cost -- the output is wrong)
    else v = g_lds.ring[(uint32_t)idx & RM];
#else
    else v = o.sym[idx];
#endif
    g_lds.ring[(o.pos + j) & RM] = v;
  }
  o.pos += len;
}

struct ChainResult {
  uint64_t end_bit;
  uint32_t n_out, status, flags, n_members;
};

// gzip member header at a byte boundary (RFC 1952 section 2.3).  0 ok, 1 not a gzip header, 2 ran out of input
__device__ __forceinline__ int member_header(Bits& br, uint64_t n_bits) {
  auto need = [&](uint32_t bytes) { return br.pos() + 8ull * bytes <= n_bits; };
  auto byte = [&]() {
    br.refill();
    return br.take(8);
  };
  if (!need(10)) return 2;
  if (byte() != 0x1f || byte() != 0x8b) return 1;
  if (byte() != 8) return 1;
  const uint32_t flg = byte();
  if (flg & 0xE0) return 1;
  for (int i = 0; i < 6; ++i) (void)byte();
  if (flg & 4) {  // FEXTRA
    if (!need(2)) return 2;
    uint32_t xlen = byte();
    xlen |= byte() << 8;
    if (!need(xlen)) return 2;
    for (uint32_t i = 0; i < xlen; ++i) (void)byte();
  }
  for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
    if (flg & f) {
      for (;;) {
        if (!need(1)) return 2;
        if (byte() == 0) break;
      }
    }
  if (flg & 2) {  // FHCRC
    if (!need(2)) return 2;
    (void)byte();
    (void)byte();
  }
  return 0;
}

// Decode blocks from `start` (a block boundary, or a member header) up to the first block boundary at or behind `stop_bit`.
// Input that ends inside a block (or inside the trailer / next header behind a final block) rolls the chain back to that block's
// start: F_EXHAUSTED.  Everything is wave-uniform.
__device__ __noinline__ ChainResult decode_chain(const uint32_t* comp, uint64_t n_bits, uint64_t start_bit, uint32_t start_kind, uint64_t stop_bit, int input_final,
                                                 uint16_t* sym, uint32_t cap, GzMember* members) {
  // (arguments of a non-inlined function arrive in vector registers: make them scalar again)
  comp = unip(comp);
  sym = unip(sym);
  members = unip(members);
  n_bits = uni64(n_bits);
  start_bit = uni64(start_bit);
  stop_bit = uni64(stop_bit);
  start_kind = uniu(start_kind);
  input_final = uni(input_final);
  cap = uniu(cap);
  Lds& L = g_lds;
  ChainResult r;
  r.end_bit = start_bit;
  r.n_out = 0;
  r.status = GZ_OK;
  r.flags = 0;
  r.n_members = 0;
  Bits br;
  br.init(comp, start_bit);
  Out o{sym, cap, 0, 0};
  if (start_kind == START_MEMBER) {
    const int h = uni(member_header(br, n_bits));
    if (h == 1) {
      r.status = GZ_BAD_HEADER;
      return r;
    }
    if (h == 2) {
      r.flags = F_EXHAUSTED | F_AT_MEMBER;
      return r;
    }
  }
  for (;;) {
    const uint64_t block_bit = br.pos();
    const uint32_t block_pos = o.pos, block_members = r.n_members;
    auto exhausted = [&]() {  // roll back to this block's start
      r.end_bit = block_bit;
      r.n_out = block_pos;
      r.n_members = block_members;
      r.flags |= F_EXHAUSTED;
      r.status = GZ_OK;
      o.pos = block_pos;
    };
    if (block_bit >= stop_bit) {
      r.end_bit = block_bit;
      r.n_out = o.pos;
      break;
    }
    if (block_bit + 3 > n_bits) {
      exhausted();
      break;
    }
    br.refill();
    const uint32_t bfinal = br.take(1), btype = br.take(2);
    if (btype == 3) {
      r.status = GZ_BAD_BTYPE;
      return r;
    }
    if (btype == 0) {
      br.drop(br.cnt & 7);  // to the byte boundary (cnt counts the bits left of an aligned dword stream)
      br.refill();
      const uint32_t len = br.take(16);
      br.refill();
      const uint32_t nlen = br.take(16);
      if ((len ^ nlen) != 0xFFFFu) {
        if (br.pos() > n_bits) {
          exhausted();
          break;
        }
        r.status = GZ_BAD_STORED;
        return r;
      }
      const uint64_t data_bit = br.pos();
      if (data_bit + 8ull * len > n_bits) {
        exhausted();
        break;
      }
      const uint8_t* bytes = reinterpret_cast<const uint8_t*>(comp) + (data_bit >> 3);
      for (uint32_t i = 0; i < len; i += 64) {
        const uint32_t j = i + lane_id();
        if (j < len) L.ring[(o.pos + (j - i)) & RM] = (uint16_t)bytes[j];
        o.pos += min(64u, len - i);
        if (!drain_full(o)) {
          r.status = GZ_SYM_OVERFLOW;
          return r;
        }
      }
      br.init(comp, data_bit + 8ull * len);
    } else {
      if (btype == 1) {
        for (int i = (int)lane_id(); i < 320; i += 64) L.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
        wave_fence();
        (void)build_code(C_DIST, L.lens + 288, 32);  // (the lengths lie under the literal/length table: it is built last)
        (void)build_code(C_LIT, L.lens, 288);
      } else {
        br.refill();
        const int hlit = (int)br.take(5) + 257, hdist = (int)br.take(5) + 1, hclen = (int)br.take(4) + 4;
        if (hlit > 286 || hdist > 30) {
          if (br.pos() > n_bits) {  // (bits behind the end of the input are padding, not a verdict)
            exhausted();
            goto chain_done;
          }
          r.status = GZ_BAD_LENGTHS;
          return r;
        }
        if (lane_id() < 19) L.cl_lens[lane_id()] = 0;
        wave_fence();
        for (int i = 0; i < hclen; ++i) {
          br.refill();
          const uint32_t v = br.take(3);
          if (lane_id() == 0) L.cl_lens[cl_order(i)] = (uint8_t)v;
        }
        wave_fence();
        if (!uni(build_code(C_CL, L.cl_lens, 19))) {
          if (br.pos() > n_bits) {  // (bits behind the end of the input are padding, not a verdict)
            exhausted();
            goto chain_done;
          }
          r.status = GZ_BAD_LENGTHS;
          return r;
        }
        const int total = hlit + hdist;
        int i = 0;
        uint32_t prev = 0;
        bool bad = false;
        while (i < total) {
          br.refill();
          const uint32_t e = decode_symbol<C_CL>(br);
          if (e & E_INVALID) {
            bad = true;
            break;
          }
          const uint32_t s = e >> 16;
          if (s < 16) {
            if (lane_id() == 0) L.lens[i] = (uint8_t)s;
            prev = s;
            ++i;
            continue;
          }
          uint32_t rep, val = 0;
          if (s == 16) {
            if (i == 0) {
              bad = true;
              break;
            }
            rep = 3 + br.take(2);
            val = prev;
          } else if (s == 17) {
            rep = 3 + br.take(3);
            prev = 0;
          } else {
            rep = 11 + br.take(7);
            prev = 0;
          }
          if (i + (int)rep > total) {
            bad = true;
            break;
          }
          for (uint32_t k = lane_id(); k < rep; k += 64) L.lens[i + (int)k] = (uint8_t)val;
          i += (int)rep;
        }
        wave_fence();
        if (bad || br.pos() > n_bits + 64) {
          if (br.pos() > n_bits) {
            exhausted();
            break;
          }
          if (br.pos() > n_bits) {  // (bits behind the end of the input are padding, not a verdict)
            exhausted();
            goto chain_done;
          }
          r.status = GZ_BAD_LENGTHS;
          return r;
        }
        if (uniu(L.lens[256]) == 0) {  // no end-of-block code
          if (br.pos() > n_bits) {  // (bits behind the end of the input are padding, not a verdict)
            exhausted();
            goto chain_done;
          }
          r.status = GZ_BAD_LENGTHS;
          return r;
        }
        // the distance lengths move behind the 288 literal/length slots (build_code reads them from there; the literal/length build
        // must not see them as symbols 286, 287)
        uint8_t dl = 0;
        if ((int)lane_id() < hdist) dl = L.lens[hlit + (int)lane_id()];
        wave_fence();
        if ((int)lane_id() < 32) L.lens[288 + lane_id()] = (int)lane_id() < hdist ? dl : 0;
        wave_fence();
        if (!uni(build_code(C_DIST, L.lens + 288, hdist)) || !uni(build_code(C_LIT, L.lens, hlit))) {
          if (br.pos() > n_bits) {  // (bits behind the end of the input are padding, not a verdict)
            exhausted();
            goto chain_done;
          }
          r.status = GZ_BAD_LENGTHS;
          return r;
        }
      }
      // symbols.  The kernel is bound by instruction issue (profiles/r6_gz_decode_pmc_*.txt: ~15 scalar + 7 vector instructions
      // per output byte at first), so the two common cases are kept short: a run of literals is a loop of lookup / store / shift
      // with the drain test reduced to one compare against a precomputed position (FASTQ +7 %, VCF text unchanged).
      bool ran_out = false;
      uint32_t drain_at = o.drained + DRAIN;  // (pos - drained < DRAIN holds here: a drain is due when pos reaches drain_at)
      for (;;) {
        uint32_t e;
        for (;;) {  // literals
          br.refill();
          e = uniu(L.lit_lut[br.peek(LIT_BITS)]);
          if (!(e & E_LIT)) break;  // (a literal entry always carries its code length: long codes and everything else leave)
          L.ring[o.pos & RM] = (uint16_t)(e >> 16);  // all lanes store the same value to the same address
          ++o.pos;
          br.drop((int)(e & 15u));
          if (o.pos == drain_at) break;
        }
        if (e & E_LIT) {
          // a drain is due (checked below)
        } else {
          if ((e & 15u) == 0) {  // longer than the table, or no such code
            const int rl = uni(decode_long(C_LIT, (uint32_t)br.buf));
            e = rl < 0 ? (uint32_t)E_INVALID : entry_for(C_LIT, rl >> 8) | (uint32_t)(rl & 255);
          }
          br.drop((int)(e & 15u));
          if (e & (E_LIT | E_EOB | E_INVALID)) {
            if (e & E_LIT) {
              L.ring[o.pos & RM] = (uint16_t)(e >> 16);
              ++o.pos;
            } else if (e & E_EOB) {
              break;
            } else {
              r.status = GZ_BAD_CODE;
              break;
            }
          } else {
            const uint32_t len = (e >> 16) + br.take((int)((e >> 4) & 15u));
            br.refill();
            const uint32_t de = decode_symbol<C_DIST>(br);
            if (de & E_INVALID) {
              r.status = GZ_BAD_DISTANCE;
              break;
            }
            const uint32_t dist = (de >> 16) + br.take((int)((de >> 4) & 15u));
            copy_match(o, len, dist);  // (a special case for short matches whose sources all lie in the ring measured SLOWER on VCF text: profiles/r6_gz_loop_ab.log)
          }
        }
        if (o.pos >= drain_at) {
          if (!drain_full(o)) {
            r.status = GZ_SYM_OVERFLOW;
            break;
          }
          drain_at = o.drained + DRAIN;
          if (br.pos() > n_bits) {  // (checked once per drain: zero padding decodes as symbols for ever)
            ran_out = true;
            break;
          }
        }
      }
      if (ran_out || (br.pos() > n_bits && r.status != GZ_SYM_OVERFLOW)) {
        exhausted();
        break;
      }
      if (r.status != GZ_OK) return r;
    }
    if (bfinal) {
      // member trailer: CRC-32 and ISIZE at the next byte boundary; then the stream ends, or another member follows
      br.drop(br.cnt & 7);
      if (br.pos() + 64 > n_bits) {
        if (input_final) {
          r.status = GZ_TRUNCATED;
          return r;
        }
        exhausted();
        break;
      }
      br.refill();  // (take(n) is for n < 32)
      uint32_t crc = br.take(16);
      crc |= br.take(16) << 16;
      br.refill();
      uint32_t isize = br.take(16);
      isize |= br.take(16) << 16;
      if (r.n_members >= (uint32_t)MAX_MEMBER_ENDS) {
        r.status = GZ_MEMBER_OVERFLOW;
        return r;
      }
      if (lane_id() == 0) members[r.n_members] = GzMember{o.pos, crc, isize, 0};
      ++r.n_members;
      if (br.pos() == n_bits && input_final) {
        r.end_bit = n_bits;
        r.n_out = o.pos;
        r.flags |= F_STREAM_END;
        break;
      }
      const int h = uni(member_header(br, n_bits));
      if (h == 1) {
        r.status = GZ_BAD_HEADER;  // bytes behind the last member that are not a member: the host reader decides what they are
        return r;
      }
      if (h == 2) {
        if (input_final) {
          r.status = GZ_TRUNCATED;
          return r;
        }
        exhausted();
        break;
      }
    }
  }
chain_done:
  if (r.status == GZ_OK) {
    o.pos = r.n_out;
    if (!drain_rest(o)) r.status = GZ_SYM_OVERFLOW;
  }
  return r;
}

// One wavefront per task (a chunk of the slab, or a repair).
__global__ __launch_bounds__(64) void k_gz_decode(const uint32_t* __restrict__ comp, uint64_t n_bits, const GzTask* __restrict__ tasks, uint16_t* __restrict__ sym, uint32_t cap,
                                                  GzChunk* __restrict__ res, GzMember* __restrict__ members, int input_final) {
  const uint64_t t_begin = wall_clock64();
  const GzTask st = tasks[blockIdx.x];
  const int c = (int)uniu(st.region);
  const uint64_t stop_bit = uni64(st.stop);
  uint16_t* my_sym = sym + (size_t)c * cap;
  GzMember* my_members = members + (size_t)c * MAX_MEMBER_ENDS;
  ChainResult r;
  uint64_t began = st.bit;
  if (uniu(st.kind) != START_SEARCH) {
    r = decode_chain(comp, n_bits, uni64(st.bit), uniu(st.kind), stop_bit, input_final, my_sym, cap, my_members);
  } else {
    r.status = GZ_NOT_FOUND;
    r.end_bit = 0;
    r.n_out = 0;
    r.flags = 0;
    r.n_members = 0;
    const uint64_t lo = uni64(st.bit);
    const uint64_t hi = min(stop_bit, n_bits);
    bool done = false;
    for (uint64_t p = lo; p < hi && !done; p += 64) {
      // lane l: is bit p + l a plausible dynamic-block header?  BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29, and the code-length
      // code complete (Kraft sum of its 3-bit lengths exactly 1)
      const uint64_t q = p + lane_id();
      const uint32_t di = (uint32_t)(q >> 5), sh = (uint32_t)(q & 31);
      const uint64_t w01 = (uint64_t)comp[di] | ((uint64_t)comp[di + 1] << 32);
      const uint64_t w23 = (uint64_t)comp[di + 2] | ((uint64_t)comp[di + 3] << 32);
      const uint64_t b0 = sh ? (w01 >> sh) | (w23 << (64 - sh)) : w01;  // bits q .. q + 63
      const uint64_t b1 = w23 >> sh;                                     // bits q + 64 .. (at least 33 of them)
      bool ok = q < hi && (b0 & 7u) == 4u && ((b0 >> 3) & 31u) <= 29u && ((b0 >> 8) & 31u) <= 29u;
      if (ok) {
        const int hclen = (int)((b0 >> 13) & 15u) + 4;
        const uint64_t t = (b0 >> 17) | (b1 << 47);  // the 3-bit lengths
        uint32_t kraft = 0;
        for (int k = 0; k < hclen; ++k) {
          const uint32_t l = (uint32_t)(t >> (3 * k)) & 7u;
          if (l) kraft += 128u >> l;
        }
        ok = kraft == 128u;
      }
      unsigned long long m = __ballot(ok);
      while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint64_t cand = p + (uint64_t)l;
        const ChainResult t = decode_chain(comp, n_bits, cand, START_BLOCK, stop_bit, input_final, my_sym, cap, my_members);
        // a candidate stands when its chain reached the next chunk's range (or the end of the input) without anything zlib would
        // reject; one that produced nothing before running out of input proves nothing and is left to the chunk in front
        if (t.status == GZ_OK && (t.n_out > 0 || t.end_bit > cand)) {
          r = t;
          began = cand;
          done = true;
          break;
        }
        if (t.status == GZ_SYM_OVERFLOW) {  // cannot be judged with this scratch: the host repeats the call with more room
          r = t;
          began = cand;
          done = true;
          break;
        }
      }
    }
  }
  if (lane_id() == 0) {
    GzChunk out;
    out.start_bit = began;
    out.end_bit = r.end_bit;
    out.n_out = r.n_out;
    out.status = r.status;
    out.flags = r.flags;
    out.n_members = r.n_members;
    out.ticks = (uint32_t)(wall_clock64() - t_begin);
    out.slot = (uint32_t)t_begin;
    res[c] = out;
  }
}

// ---- windows ------------------------------------------------------------------------------------------------------------------------
// accepted chunk a: symbols sym[chunk[a] * cap ...], n_out[a] of them.  Its TAIL map T over the window W in front of it
// (32 Ki entries, W[32767] = the byte right in front of the chunk): T[j] = the symbol that becomes byte j of the window behind it.
struct Accepted {
  uint64_t sym_off;  // where its symbols start in the scratch
  uint64_t out_off;  // bytes in front of it in the slab's output
  uint32_t n_out, pad;
};
__device__ __forceinline__ uint16_t tail_symbol(const uint16_t* __restrict__ s, uint32_t n_out, int j) {
  if (n_out >= (uint32_t)WIN) return s[n_out - WIN + j];
  const int keep = WIN - (int)n_out;  // the window's last `keep` bytes slide to the front
  return j < keep ? (uint16_t)(256 + j + (int)n_out) : s[j - keep];
}
constexpr int GROUP = 64;  // chunks per group (the chain of groups is serial: ~6.6 us each; 13 k chunks = 205 groups)
// level A: the composed map of every group (over the window in front of the group's first chunk)
__global__ __launch_bounds__(1024) void k_gz_compose(const uint16_t* __restrict__ sym, uint32_t cap, const Accepted* __restrict__ acc, int n_acc, uint16_t* __restrict__ group_map) {
  extern __shared__ uint16_t m_lds[];  // two maps of 32 Ki symbols
  uint16_t* cur = m_lds;
  uint16_t* nxt = m_lds + WIN;
  const int g = blockIdx.x;
  const int a0 = g * GROUP, a1 = min(n_acc, a0 + GROUP);
  for (int j = threadIdx.x; j < WIN; j += 1024) cur[j] = (uint16_t)(256 + j);
  __syncthreads();
  for (int a = a0; a < a1; ++a) {
    const uint16_t* s = sym + acc[a].sym_off;
    const uint32_t n = acc[a].n_out;
    for (int j = threadIdx.x; j < WIN; j += 1024) {
      const uint16_t t = tail_symbol(s, n, j);
      nxt[j] = t < 256 ? t : cur[t - 256];
    }
    __syncthreads();
    uint16_t* x = cur;
    cur = nxt;
    nxt = x;
  }
  uint16_t* dst = group_map + (size_t)g * WIN;
  for (int j = threadIdx.x; j < WIN; j += 1024) dst[j] = cur[j];
}
// level B: one workgroup chains the groups: group_win[g] = the window in front of group g; win_out = the window behind the last
__global__ __launch_bounds__(1024) void k_gz_groups(const uint16_t* __restrict__ group_map, int n_groups, const uint8_t* __restrict__ win_in, uint8_t* __restrict__ group_win,
                                                    uint8_t* __restrict__ win_out) {
  __shared__ uint8_t w[2][WIN];
  int k = 0;
  for (int j = threadIdx.x; j < WIN; j += 1024) w[0][j] = win_in[j];
  __syncthreads();
  for (int g = 0; g < n_groups; ++g) {
    uint8_t* dst = group_win + (size_t)g * WIN;
    const uint16_t* m = group_map + (size_t)g * WIN;
    for (int j = threadIdx.x; j < WIN; j += 1024) {
      dst[j] = w[k][j];
      const uint16_t t = m[j];
      w[k ^ 1][j] = t < 256 ? (uint8_t)t : w[k][t - 256];
    }
    __syncthreads();
    k ^= 1;
  }
  for (int j = threadIdx.x; j < WIN; j += 1024) win_out[j] = w[k][j];
}
// level C: every group resolves the window in front of each of its chunks
__global__ __launch_bounds__(1024) void k_gz_windows(const uint16_t* __restrict__ sym, uint32_t cap, const Accepted* __restrict__ acc, int n_acc,
                                                     const uint8_t* __restrict__ group_win, uint8_t* __restrict__ chunk_win) {
  __shared__ uint8_t w[2][WIN];
  const int g = blockIdx.x;
  const int a0 = g * GROUP, a1 = min(n_acc, a0 + GROUP);
  int k = 0;
  for (int j = threadIdx.x; j < WIN; j += 1024) w[0][j] = group_win[(size_t)g * WIN + j];
  __syncthreads();
  for (int a = a0; a < a1; ++a) {
    uint8_t* dst = chunk_win + (size_t)a * WIN;
    const uint16_t* s = sym + acc[a].sym_off;
    const uint32_t n = acc[a].n_out;
    const bool last = a + 1 == a1;
    for (int j = threadIdx.x; j < WIN; j += 1024) {
      dst[j] = w[k][j];
      if (!last) {
        const uint16_t t = tail_symbol(s, n, j);
        w[k ^ 1][j] = t < 256 ? (uint8_t)t : w[k][t - 256];
      }
    }
    __syncthreads();
    k ^= 1;
  }
}
// symbols -> bytes.  One workgroup per (accepted chunk, piece of 16 Ki symbols); a thread takes 8 symbols per step with one aligned
// 16-byte load (a piece starts on a 32 KiB boundary of the chunk's region) and writes their 8 bytes with one store (the output
// address has whatever alignment the chunk's offset gives it: the hardware takes unaligned dword stores).
constexpr int EMIT_PIECE = 16384;
__global__ __launch_bounds__(256) void k_gz_emit(const uint16_t* __restrict__ sym, uint32_t cap, const Accepted* __restrict__ acc, int pieces_per_chunk,
                                                 const uint8_t* __restrict__ chunk_win, uint8_t* __restrict__ out) {
  const int a = blockIdx.x / pieces_per_chunk, piece = blockIdx.x % pieces_per_chunk;
  const Accepted A = acc[a];
  const uint32_t lo = (uint32_t)piece * EMIT_PIECE;
  if (lo >= A.n_out) return;
  const uint32_t hi = min(A.n_out, lo + EMIT_PIECE);
  const uint16_t* s = sym + A.sym_off;
  const uint8_t* w = chunk_win + (size_t)a * WIN;
  uint8_t* o = out + A.out_off;
  for (uint32_t i = lo + threadIdx.x * 8u; i < hi; i += 256u * 8u) {
    if (i + 8u <= hi) {
      const uint4 q = *reinterpret_cast<const uint4*>(s + i);
      const uint32_t t[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
      uint32_t lo4 = 0, hi4 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        lo4 |= (uint32_t)(t[k] < 256u ? t[k] : (uint32_t)w[t[k] - 256u]) << (8 * k);
        hi4 |= (uint32_t)(t[k + 4] < 256u ? t[k + 4] : (uint32_t)w[t[k + 4] - 256u]) << (8 * k);
      }
      typedef uint32_t u32_unaligned __attribute__((aligned(1)));
      *reinterpret_cast<u32_unaligned*>(o + i) = lo4;
      *reinterpret_cast<u32_unaligned*>(o + i + 4) = hi4;
    } else {
      for (uint32_t k = i; k < hi; ++k) {
        const uint16_t t = s[k];
        o[k] = t < 256 ? (uint8_t)t : w[t - 256];
      }
    }
  }
}

// ---- CRC-32 ---------------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t gf2_mulmod(uint32_t a, uint32_t b) {  // a * b mod P, reflected
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) {
    if (a & 0x80000000u) r ^= b;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
  }
  return r;
}
__host__ __device__ inline uint32_t x_pow_8n(uint64_t n_bytes) {  // x^(8 n) mod P
  uint32_t result = 0x80000000u, sq = 0x00800000u;
  while (n_bytes) {
    if (n_bytes & 1u) result = gf2_mulmod(result, sq);
    sq = gf2_mulmod(sq, sq);
    n_bytes >>= 1;
  }
  return result;
}
struct CrcPiece {
  uint64_t off;
  uint32_t len, raw;  // raw: the CRC register over the piece from a ZERO initial value, no final xor
};
// one wavefront per piece (<= 64 KiB): every lane its slice, combined with x^(8 bytes behind the slice)
__global__ __launch_bounds__(256) void k_gz_crc(const uint8_t* __restrict__ out, CrcPiece* __restrict__ pieces, int n_pieces) {
  __shared__ uint32_t table[4][256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    uint32_t c = (uint32_t)i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
    table[0][i] = c;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    uint32_t c = table[0][i];
    for (int k = 1; k < 4; ++k) {
      c = table[0][c & 0xFFu] ^ (c >> 8);
      table[k][i] = c;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pi = blockIdx.x * 4 + wave;
  if (pi >= n_pieces) return;
  const uint32_t n = pieces[pi].len;
  const uint8_t* p = out + pieces[pi].off;
  const uint32_t per = (((n + 63u) / 64u) + 15u) & ~15u;
  const uint32_t lo = min(n, (uint32_t)lane * per), hi = min(n, lo + per);
  uint32_t c = 0, i = lo;
  auto step4 = [&](uint32_t w) {
    c ^= w;
    c = table[3][c & 0xFFu] ^ table[2][(c >> 8) & 0xFFu] ^ table[1][(c >> 16) & 0xFFu] ^ table[0][c >> 24];
  };
  while (i < hi && ((reinterpret_cast<uintptr_t>(p + i)) & 15u)) c = table[0][(c ^ p[i++]) & 0xFFu] ^ (c >> 8);
  for (; i + 64 <= hi; i += 64) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4*>(p + i + 16 * k);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      step4(v[k].x);
      step4(v[k].y);
      step4(v[k].z);
      step4(v[k].w);
    }
  }
  for (; i + 4 <= hi; i += 4) step4(*reinterpret_cast<const uint32_t*>(p + i));
  for (; i < hi; ++i) c = table[0][(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  uint32_t term = hi > lo ? gf2_mulmod(c, x_pow_8n(n - hi)) : 0u;
  for (int o = 32; o > 0; o >>= 1) term ^= __shfl_xor(term, o, 64);
  if (lane == 0) pieces[pi].raw = term;
}

const char* gz_status_name(uint32_t s) {
  switch (s) {
    case GZ_OK: return "ok";
    case GZ_NOT_FOUND: return "no block start found";
    case GZ_BAD_BTYPE: return "invalid block type";
    case GZ_BAD_STORED: return "invalid stored block lengths";
    case GZ_BAD_LENGTHS: return "invalid code lengths set";
    case GZ_BAD_CODE: return "invalid literal/length code";
    case GZ_BAD_DISTANCE: return "invalid distance code";
    case GZ_SYM_OVERFLOW: return "symbol scratch overflow";
    case GZ_BAD_HEADER: return "not a gzip member header";
    case GZ_MEMBER_OVERFLOW: return "too many members in one chunk";
    case GZ_TRUNCATED: return "truncated gzip stream";
  }
  return "?";
}

}  // namespace

// ---- host driver ------------------------------------------------------------------------------------------------------------------
struct exon_hip_gzip_stream {
  exon_hip_ctx* ctx = nullptr;
  uint32_t chunk_bytes = 65536;     // the largest chunk (EXON_HIP_GZ_CHUNK_KB fixes the size)
  uint32_t min_chunk_bytes = 16384; // a call picks its chunk size in between so that the chip's wavefront slots are filled twice
  bool fixed_chunk = false;
  int64_t max_comp = 0;
  int max_chunks = 0;
  size_t sym_words = 0;  // symbols of scratch (the chunks' regions)
  size_t gap_words = 0;  // ... and behind them GAP_CAP symbols per batched gap
  uint16_t* d_sym = nullptr;
  GzChunk* d_res = nullptr;
  GzChunk* h_res = nullptr;  // pinned
  GzMember* d_members = nullptr;
  GzMember* h_members = nullptr;
  GzTask* d_tasks = nullptr;
  GzTask* h_tasks = nullptr;
  Accepted* d_acc = nullptr;
  Accepted* h_acc = nullptr;
  uint16_t* d_group_map = nullptr;
  uint8_t* d_group_win = nullptr;
  uint8_t* d_chunk_win = nullptr;
  uint8_t* d_win[2] = {nullptr, nullptr};
  int win_k = 0;
  CrcPiece* d_pieces = nullptr;
  CrcPiece* h_pieces = nullptr;
  size_t max_pieces = 0;
  // state carried from call to call
  uint32_t start_kind = START_MEMBER;
  uint32_t start_bit = 0;       // bit offset of the next call's start inside its first byte
  bool ended = false;           // the last member's trailer was the end of the input
  uint32_t member_raw = 0;      // running CRC register of the member being decoded (zero initial value, no xors)
  uint64_t member_len = 0;
  bool verify_crc = true;
  // statistics
  exon_hip_gzip_stats stats{};
};

namespace {
void gz_free(exon_hip_gzip_stream* s) {
  auto dfree = [](void* p) { if (p) hipFree(p); };
  auto hfree = [](void* p) { if (p) hipHostFree(p); };
  dfree(s->d_sym), dfree(s->d_res), dfree(s->d_members), dfree(s->d_tasks), dfree(s->d_acc), dfree(s->d_group_map), dfree(s->d_group_win),
      dfree(s->d_chunk_win), dfree(s->d_win[0]), dfree(s->d_win[1]), dfree(s->d_pieces);
  hfree(s->h_res), hfree(s->h_members), hfree(s->h_tasks), hfree(s->h_acc), hfree(s->h_pieces);
}
}  // namespace

extern "C" {

int exon_hip_gzip_stream_create(exon_hip_ctx* ctx, int64_t max_comp_bytes, int64_t scratch_bytes, exon_hip_gzip_stream** out) {
  if (!ctx || !out) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gzip_stream_create: NULL argument");
  *out = nullptr;
  if (max_comp_bytes < 1) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gzip_stream_create: max_comp_bytes must be positive");
  std::unique_ptr<exon_hip_gzip_stream> s(new exon_hip_gzip_stream());
  s->ctx = ctx;
  if (const char* v = getenv("EXON_HIP_GZ_CHUNK_KB")) {
    const long kb = atol(v);
    if (kb >= 1 && kb <= 4096) {
      s->chunk_bytes = s->min_chunk_bytes = (uint32_t)kb << 10;
      s->fixed_chunk = true;
    }
  }
  if (const char* v = getenv("EXON_HIP_GZ_VERIFY_CRC")) s->verify_crc = !(v[0] == '0');
  s->max_comp = max_comp_bytes;
  // adaptive chunks: a call never has more than two rounds of wavefront slots' worth of them, or max_comp / 64 KiB
  s->max_chunks = s->fixed_chunk ? (int)((max_comp_bytes + s->chunk_bytes - 1) / s->chunk_bytes) + 1
                                 : (int)std::max<int64_t>(2ll * GZ_WAVES_PER_CU * std::max(ctx->cfg.compute_units, 1) + 2, (max_comp_bytes + s->chunk_bytes - 1) / s->chunk_bytes + 2);
  const size_t nr = 2 * (size_t)s->max_chunks + SPARE_REGIONS;  // records: one per chunk + the spares of sequential repairs + one per batched gap
  // symbol scratch: 2 bytes per output byte.  Default: room for a ratio of 8 over the largest slab (at least 1 Mi symbols per
  // chunk are never needed: a chunk's region is scratch / chunks, the call shrinks its slab when a region overflows)
  if (scratch_bytes <= 0) scratch_bytes = std::max<int64_t>(max_comp_bytes * 16, 64 << 20);
  s->sym_words = (size_t)scratch_bytes / 2;
  s->gap_words = (size_t)s->max_chunks * GAP_CAP;  // the batched gaps' own area behind the chunks' regions
  hipSetDevice(ctx->device);
  const size_t nc = nr;
  const size_t ng = (nc + GROUP - 1) / GROUP;
  s->max_pieces = s->sym_words / 65536 + 2 * nc * (MAX_MEMBER_ENDS + 1) + 16;
  bool ok = hipMalloc((void**)&s->d_sym, (s->sym_words + s->gap_words) * 2 + 64) == hipSuccess && hipMalloc((void**)&s->d_res, nc * sizeof(GzChunk)) == hipSuccess &&
            hipMalloc((void**)&s->d_members, nc * MAX_MEMBER_ENDS * sizeof(GzMember)) == hipSuccess && hipMalloc((void**)&s->d_tasks, nc * sizeof(GzTask)) == hipSuccess &&
            hipMalloc((void**)&s->d_acc, nc * sizeof(Accepted)) == hipSuccess &&
            hipMalloc((void**)&s->d_group_map, ng * WIN * 2) == hipSuccess && hipMalloc((void**)&s->d_group_win, ng * WIN) == hipSuccess &&
            hipMalloc((void**)&s->d_chunk_win, nc * WIN) == hipSuccess && hipMalloc((void**)&s->d_win[0], WIN) == hipSuccess && hipMalloc((void**)&s->d_win[1], WIN) == hipSuccess &&
            hipMalloc((void**)&s->d_pieces, s->max_pieces * sizeof(CrcPiece)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&s->h_res, nc * sizeof(GzChunk)) == hipSuccess && hipHostMalloc((void**)&s->h_members, nc * MAX_MEMBER_ENDS * sizeof(GzMember)) == hipSuccess &&
       hipHostMalloc((void**)&s->h_tasks, nc * sizeof(GzTask)) == hipSuccess &&
       hipHostMalloc((void**)&s->h_acc, nc * sizeof(Accepted)) == hipSuccess && hipHostMalloc((void**)&s->h_pieces, s->max_pieces * sizeof(CrcPiece)) == hipSuccess;
  if (ok) ok = hipMemset(s->d_win[0], 0, WIN) == hipSuccess && hipMemset(s->d_win[1], 0, WIN) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    gz_free(s.get());
    return fail(ctx, EXON_HIP_ENOMEM, "exon_hip_gzip_stream_create: buffers for %d chunks and %zu symbols of scratch", s->max_chunks, s->sym_words);
  }
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gz_compose), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WIN * 2);
  *out = s.release();
  return EXON_HIP_OK;
}

int exon_hip_gzip_stream_destroy(exon_hip_gzip_stream* s) {
  if (!s) return EXON_HIP_OK;
  gz_free(s);
  delete s;
  return EXON_HIP_OK;
}

int exon_hip_gzip_stream_get_stats(exon_hip_gzip_stream* s, exon_hip_gzip_stats* out) {
  if (!s || !out) return fail(s ? s->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_gzip_stream_get_stats: NULL argument");
  *out = s->stats;
  return EXON_HIP_OK;
}

// One slab.  d_comp: 4-byte aligned, n_comp compressed bytes starting at the byte the previous call stopped in (or the file's first
// byte), 4096 readable bytes behind them.  Decodes as many whole blocks as fit `out_cap`; *consumed = whole bytes used up (the caller
// passes the rest again, in front of fresh bytes); *produced = bytes written to d_out.  final_input: these are the file's last bytes.
int exon_hip_gzip_stream_decode(exon_hip_gzip_stream* s, void* stream, const uint8_t* d_comp, int64_t n_comp, int32_t final_input, uint8_t* d_out, int64_t out_cap,
                                int64_t* consumed, int64_t* produced, int32_t* stream_end) {
  if (!s || !consumed || !produced) return fail(s ? s->ctx : nullptr, EXON_HIP_EINVAL, "exon_hip_gzip_stream_decode: NULL argument");
  exon_hip_ctx* ctx = s->ctx;
  *consumed = 0;
  *produced = 0;
  if (stream_end) *stream_end = s->ended ? 1 : 0;
  if (n_comp == 0 || s->ended) {
    if (s->ended && n_comp > 0) return fail(ctx, EXON_HIP_EINVAL, "gzip: bytes behind the end of the stream");
    if (!s->ended && final_input) return fail(ctx, EXON_HIP_EINVAL, "gzip: truncated stream");
    return EXON_HIP_OK;
  }
  if (!d_comp || !d_out) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_gzip_stream_decode: NULL buffer");
  // the bit reader takes aligned dwords: positions are counted from the aligned address at or below d_comp
  const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(d_comp) & 3);
  d_comp -= lead;
  hipStream_t hs = pick_stream(ctx, stream);
  int64_t n_use = std::min<int64_t>(n_comp, s->fixed_chunk ? (int64_t)(s->max_chunks - 1) * s->chunk_bytes - 4 : s->max_comp);
  for (int attempt = 0; attempt < 6; ++attempt) {
    const bool final_here = final_input && n_use == n_comp;
    // Chunk size of this call.  The kernel's throughput is flat in the chunk size once every wavefront slot is taken (26 per CU),
    // but a call lasts at least as long as ONE chunk takes one wavefront (~0.3 us per compressed byte): a slab of well-compressing
    // text holds few 64 KiB chunks (VCF: 128 MiB -> 2000 chunks = half the slots, 25 ms whatever the count).  So: enough chunks
    // for two rounds of the chip's slots, between 16 and 64 KiB (profiles/r6_gz_chunk_sweep.log).
    uint32_t chunk_bytes = s->chunk_bytes;
    if (!s->fixed_chunk) {
      const int64_t slots2 = 2ll * GZ_WAVES_PER_CU * std::max(ctx->cfg.compute_units, 1);
      const int64_t want = ((lead + n_use) / slots2 + 4095) & ~4095ll;
      chunk_bytes = (uint32_t)std::min<int64_t>(s->chunk_bytes, std::max<int64_t>(s->min_chunk_bytes, want));
    }
    while ((lead + n_use + chunk_bytes - 1) / chunk_bytes > (int64_t)s->max_chunks - 1) chunk_bytes += 4096;  // (never more chunks than the tables hold)
    const int n_chunks = (int)((lead + n_use + chunk_bytes - 1) / chunk_bytes);
    const int n_spare = n_chunks > 1 ? std::min(SPARE_REGIONS, n_chunks) : 0;  // (one chunk: nothing to repair, the whole scratch is its region)
    const uint32_t cap = (uint32_t)std::min<size_t>((s->sym_words / (size_t)(n_chunks + n_spare)) & ~(size_t)(DRAIN - 1), 1u << 30);
    if (cap < 2 * DRAIN) return fail(ctx, EXON_HIP_EINVAL, "gzip: symbol scratch too small");
    const uint64_t n_bits = 8ull * (uint64_t)(lead + n_use);
    const uint64_t chunk_bits = 8ull * chunk_bytes;
    const uint64_t first_bit = 8ull * lead + s->start_bit;
    int n_records = n_chunks + n_spare;  // result records / member slots in use (chunks, sequential spares, batched gaps)
    for (int c = 0; c < n_chunks; ++c) {
      const uint64_t stop = c + 1 < n_chunks ? (uint64_t)(c + 1) * chunk_bits : ~0ull;
      s->h_tasks[c] = c == 0 ? GzTask{first_bit, stop, s->start_kind, 0} : GzTask{(uint64_t)c * chunk_bits, stop, START_SEARCH, (uint32_t)c};
    }
    HIP_TRY(ctx, hipMemcpyAsync(s->d_tasks, s->h_tasks, (size_t)n_chunks * sizeof(GzTask), hipMemcpyHostToDevice, hs));
    hipLaunchKernelGGL(k_gz_decode, dim3(n_chunks), dim3(64), 0, hs, reinterpret_cast<const uint32_t*>(d_comp), n_bits, (const GzTask*)s->d_tasks, s->d_sym, cap, s->d_res, s->d_members,
                       final_here ? 1 : 0);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(s->h_res, s->d_res, (size_t)n_chunks * sizeof(GzChunk), hipMemcpyDeviceToHost, hs));
    HIP_TRY(ctx, hipStreamSynchronize(hs));
    s->stats.chunks += n_chunks;
    if (getenv("EXON_HIP_GZ_TRACE")) {  // how even is the work? (ticks of the constant 100 MHz counter)
      uint64_t sum = 0;
      uint32_t mx = 0, t_lo = ~0u, t_hi = 0;
      std::vector<uint32_t> tk((size_t)n_chunks);
      for (int c = 0; c < n_chunks; ++c) {
        const GzChunk& r = s->h_res[c];
        tk[(size_t)c] = r.ticks;
        sum += r.ticks;
        mx = std::max(mx, r.ticks);
        t_lo = std::min(t_lo, r.slot);
        t_hi = std::max(t_hi, r.slot + r.ticks);
      }
      std::sort(tk.begin(), tk.end());
      fprintf(stderr, "[exon-hip gz] %d chunks of %u bytes: wavefront time sum %.1f ms, median %.0f us, p90 %.0f us, max %.0f us; first start .. last end %.2f ms\n", n_chunks, chunk_bytes,
              sum / 1e5, tk[tk.size() / 2] / 100.0, tk[tk.size() * 9 / 10] / 100.0, mx / 100.0, (t_hi - t_lo) / 1e5);
    }
    // Gaps, all at once.  The header search only looks for DYNAMIC blocks: where a stored or fixed block stands at the chain's
    // position (pigz ends every 128 KiB of input with an empty stored block, zlib's Z_SYNC_FLUSH the same) the chunk behind it began
    // a few bytes late.  One walk over the records as they are -- trusting every usable chunk -- lists those gaps, ONE launch decodes
    // them (a wavefront each, into small regions of their own), and the proving walk below takes them where they fit.  (One by one
    // they cost a launch and a host round trip each: ~40 us x a third of the chunks of a pigz file.)
    struct Gap {
      uint64_t from, to;
      int record;  // index of its result record / member slots
    };
    std::vector<Gap> gap_of((size_t)n_chunks, Gap{0, 0, -1});
    {
      uint64_t cur0 = first_bit;
      int n_gaps = 0;
      const int gap_base = n_chunks + n_spare;
      for (int c = 0; c < n_chunks; ++c) {
        const uint64_t stop = c + 1 < n_chunks ? (uint64_t)(c + 1) * chunk_bits : ~0ull;
        if (c > 0 && cur0 >= stop) continue;
        const GzChunk& f = s->h_res[c];
        const bool usable = f.status == GZ_OK || f.status == GZ_SYM_OVERFLOW;
        if (c > 0 && !(usable && f.start_bit == cur0)) {
          if (!(usable && f.start_bit > cur0 && f.start_bit - cur0 <= chunk_bits) || n_gaps >= s->max_chunks) break;  // a real repair: the walk below
          gap_of[(size_t)c] = Gap{cur0, f.start_bit, gap_base + n_gaps};
          s->h_tasks[gap_base + n_gaps] = GzTask{cur0, f.start_bit, START_BLOCK, (uint32_t)n_gaps};
          ++n_gaps;
        }
        if (!usable || f.status == GZ_SYM_OVERFLOW) break;
        cur0 = f.end_bit;
        if (f.flags & (F_EXHAUSTED | F_STREAM_END)) break;
      }
      if (n_gaps > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(s->d_tasks + gap_base, s->h_tasks + gap_base, (size_t)n_gaps * sizeof(GzTask), hipMemcpyHostToDevice, hs));
        hipLaunchKernelGGL(k_gz_decode, dim3(n_gaps), dim3(64), 0, hs, reinterpret_cast<const uint32_t*>(d_comp), n_bits, (const GzTask*)(s->d_tasks + gap_base), s->d_sym + s->sym_words, GAP_CAP,
                           s->d_res + gap_base, s->d_members + (size_t)gap_base * MAX_MEMBER_ENDS, final_here ? 1 : 0);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(s->h_res + gap_base, s->d_res + gap_base, (size_t)n_gaps * sizeof(GzChunk), hipMemcpyDeviceToHost, hs));
        HIP_TRY(ctx, hipStreamSynchronize(hs));
        s->stats.repairs += (uint64_t)n_gaps;
      }
      n_records = gap_base + n_gaps;
    }
    // The chain: chunk i + 1 must have started where chunk i stopped.  `chain` lists result RECORDS in stream order.
    std::vector<int> chain;
    uint64_t cur = first_bit;
    bool overflow = false, at_end = false;
    uint32_t bad = GZ_OK;
    int spares = 0;
    const bool trace = getenv("EXON_HIP_GZ_TRACE") != nullptr;
    // one more wavefront: decode from `bit` up to `stop` into `region`
    auto redo = [&](uint64_t bit, uint64_t stop, int region) -> int {
      s->h_tasks[region] = GzTask{bit, stop, START_BLOCK, (uint32_t)region};
      HIP_TRY(ctx, hipMemcpyAsync(s->d_tasks + region, s->h_tasks + region, sizeof(GzTask), hipMemcpyHostToDevice, hs));
      hipLaunchKernelGGL(k_gz_decode, dim3(1), dim3(64), 0, hs, reinterpret_cast<const uint32_t*>(d_comp), n_bits, (const GzTask*)(s->d_tasks + region), s->d_sym, cap, s->d_res,
                         s->d_members, final_here ? 1 : 0);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipMemcpyAsync(s->h_res + region, s->d_res + region, sizeof(GzChunk), hipMemcpyDeviceToHost, hs));
      HIP_TRY(ctx, hipStreamSynchronize(hs));
      ++s->stats.repairs;
      return EXON_HIP_OK;
    };
    for (int c = 0; c < n_chunks && !at_end; ++c) {
      const uint64_t stop = c + 1 < n_chunks ? (uint64_t)(c + 1) * chunk_bits : ~0ull;
      if (c > 0 && cur >= stop) continue;  // the chunk in front decoded through this one's whole range
      int region = c;
      const GzChunk found = s->h_res[c];
      const bool usable = found.status == GZ_OK || found.status == GZ_SYM_OVERFLOW;
      if (c > 0 && !(usable && found.start_bit == cur)) {
        // Not proven.  Either the search skipped what stands at the chain's position (an empty stored block of a full flush, a fixed
        // block, the stream's last block: it only looks for dynamic headers) and found the next header behind it -- then the GAP is
        // decoded into a spare region and, when it ends exactly where this chunk began, both stand -- or the chunk began somewhere
        // false (or nowhere): it is decoded again from the proven position.
        if (trace)
          fprintf(stderr, "[exon-hip gz] repair: chunk %d of %d began at %llu (status %u flags %u n_out %u end %llu), the chain is at %llu\n", c, n_chunks,
                  (unsigned long long)found.start_bit, found.status, found.flags, found.n_out, (unsigned long long)found.end_bit, (unsigned long long)cur);
        bool spliced = false;
        const Gap& bg = gap_of[(size_t)c];
        if (usable && bg.record >= 0 && bg.from == cur && bg.to == found.start_bit) {
          const GzChunk& g = s->h_res[bg.record];
          if (g.status == GZ_OK && g.flags == 0 && g.end_bit == found.start_bit) {
            chain.push_back(bg.record);
            spliced = true;
          }
        }
        if (!spliced && usable && found.start_bit > cur && found.start_bit - cur <= chunk_bits && spares < n_spare) {
          const int gap = n_chunks + spares;  // (the regions right behind this call's chunks)
          const int rc = redo(cur, found.start_bit, gap);
          if (rc) return rc;
          const GzChunk& g = s->h_res[gap];
          if (g.status == GZ_OK && g.flags == 0 && g.end_bit == found.start_bit) {
            ++spares;
            chain.push_back(gap);
            spliced = true;
          }
        }
        if (!spliced) {
          const int rc = redo(cur, stop, c);
          if (rc) return rc;
        }
      }
      const GzChunk& r = s->h_res[region];
      if (r.status == GZ_SYM_OVERFLOW) {
        overflow = true;
        break;
      }
      if (r.status != GZ_OK) {
        bad = r.status;
        break;
      }
      chain.push_back(region);
      cur = r.end_bit;
      if (r.flags & (F_EXHAUSTED | F_STREAM_END)) at_end = true;
    }
    if (bad != GZ_OK) return fail(ctx, EXON_HIP_EINVAL, "gzip stream: %s", gz_status_name(bad));
    // what fits the caller's buffer (whole chunks)
    size_t n_acc = 0;
    uint64_t total = 0;
    for (int c : chain) {
      if (total + s->h_res[c].n_out > (uint64_t)out_cap) break;
      total += s->h_res[c].n_out;
      ++n_acc;
    }
    if (overflow && n_acc == 0) {
      // a chunk's symbols did not fit its region: fewer chunks share the scratch next time
      if (n_chunks == 1) return fail(ctx, EXON_HIP_EINVAL, "gzip stream: one chunk inflates to more than the symbol scratch holds");
      ++s->stats.overflow_retries;
      n_use = std::min<int64_t>(n_use, std::max<int64_t>((int64_t)chunk_bytes, (n_use / 4) & ~(int64_t)(chunk_bytes - 1)));
      continue;
    }
    if (n_acc == 0) {
      if (!chain.empty()) {
        if (s->h_res[chain[0]].n_out > (uint64_t)out_cap) return fail(ctx, EXON_HIP_EINVAL, "gzip stream: the output buffer is smaller than one chunk's output");
      }
      return fail(ctx, EXON_HIP_EINVAL, "gzip stream: no progress (a block larger than the slab?)");
    }
    const bool whole_chain = n_acc == chain.size() && !overflow;
    chain.resize(n_acc);
    const GzChunk& last = s->h_res[chain.back()];
    const bool ends_stream = (last.flags & F_STREAM_END) != 0;
    if (final_here && whole_chain && !ends_stream && (last.flags & F_EXHAUSTED)) return fail(ctx, EXON_HIP_EINVAL, "gzip stream: truncated");
    // accepted chunks, their output offsets
    uint64_t off = 0;
    int total_members = 0;
    for (size_t a = 0; a < n_acc; ++a) {
      const GzChunk& r = s->h_res[chain[a]];
      const int rec = chain[a];
      const uint64_t sym_off = rec < n_chunks + n_spare ? (uint64_t)rec * cap : (uint64_t)s->sym_words + (uint64_t)(rec - (n_chunks + n_spare)) * GAP_CAP;
      s->h_acc[a] = Accepted{sym_off, off, r.n_out, 0};
      off += r.n_out;
      total_members += (int)r.n_members;
    }
    HIP_TRY(ctx, hipMemcpyAsync(s->d_acc, s->h_acc, n_acc * sizeof(Accepted), hipMemcpyHostToDevice, hs));
    const int n_groups = (int)((n_acc + GROUP - 1) / GROUP);
    uint8_t* win_in = s->d_win[s->win_k];
    uint8_t* win_out = s->d_win[s->win_k ^ 1];
    hipLaunchKernelGGL(k_gz_compose, dim3(n_groups), dim3(1024), 2 * WIN * 2, hs, s->d_sym, cap, s->d_acc, (int)n_acc, s->d_group_map);
    hipLaunchKernelGGL(k_gz_groups, dim3(1), dim3(1024), 0, hs, s->d_group_map, n_groups, win_in, s->d_group_win, win_out);
    hipLaunchKernelGGL(k_gz_windows, dim3(n_groups), dim3(1024), 0, hs, s->d_sym, cap, s->d_acc, (int)n_acc, s->d_group_win, s->d_chunk_win);
    uint32_t max_out = 1;
    for (size_t a = 0; a < n_acc; ++a) max_out = std::max(max_out, s->h_acc[a].n_out);
    const int ppc = (int)((max_out + EMIT_PIECE - 1) / EMIT_PIECE);  // (pieces of the LARGEST accepted chunk: smaller ones leave theirs at once)
    // (one launch per 2^31 / ppc chunks would be needed beyond that; slabs are far smaller)
    hipLaunchKernelGGL(k_gz_emit, dim3((unsigned)(n_acc * (size_t)ppc)), dim3(256), 0, hs, s->d_sym, cap, s->d_acc, ppc, s->d_chunk_win, d_out);
    HIP_TRY(ctx, hipGetLastError());
    // CRC pieces: <= 64 KiB each, cut at every member end
    size_t n_pieces = 0;
    std::vector<std::pair<uint64_t, const GzMember*>> ends;  // (absolute output offset of a member end, its trailer)
    if (s->verify_crc) {
      if (total_members) {
        HIP_TRY(ctx, hipMemcpyAsync(s->h_members, s->d_members, (size_t)n_records * MAX_MEMBER_ENDS * sizeof(GzMember), hipMemcpyDeviceToHost, hs));
        HIP_TRY(ctx, hipStreamSynchronize(hs));
        for (size_t a = 0; a < n_acc; ++a) {
          const GzChunk& r = s->h_res[chain[a]];
          for (uint32_t m = 0; m < r.n_members; ++m) {
            const GzMember* me = &s->h_members[(size_t)chain[a] * MAX_MEMBER_ENDS + m];
            ends.emplace_back(s->h_acc[a].out_off + me->out_off, me);
          }
        }
      }
      uint64_t p = 0;
      size_t e = 0;
      while (p < off) {
        uint64_t lim = off;
        while (e < ends.size() && ends[e].first <= p) ++e;
        if (e < ends.size()) lim = std::min(lim, ends[e].first);
        const uint32_t len = (uint32_t)std::min<uint64_t>(65536, lim - p);
        if (n_pieces >= s->max_pieces) return fail(ctx, EXON_HIP_EINVAL, "gzip stream: CRC piece table overflow");
        s->h_pieces[n_pieces++] = CrcPiece{p, len, 0};
        p += len;
      }
      if (n_pieces) {
        HIP_TRY(ctx, hipMemcpyAsync(s->d_pieces, s->h_pieces, n_pieces * sizeof(CrcPiece), hipMemcpyHostToDevice, hs));
        hipLaunchKernelGGL(k_gz_crc, dim3((unsigned)((n_pieces + 3) / 4)), dim3(256), 0, hs, d_out, s->d_pieces, (int)n_pieces);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(s->h_pieces, s->d_pieces, n_pieces * sizeof(CrcPiece), hipMemcpyDeviceToHost, hs));
      }
    }
    HIP_TRY(ctx, hipStreamSynchronize(hs));
    if (s->verify_crc) {
      // chain the pieces' registers; at every member end compare with the trailer: crc = raw ^ init * x^(8 n) ^ 0xFFFFFFFF
      size_t e = 0;
      uint32_t xp64k = x_pow_8n(65536);
      for (size_t i = 0; i <= n_pieces; ++i) {
        const uint64_t at = i < n_pieces ? s->h_pieces[i].off : off;
        while (e < ends.size() && ends[e].first == at) {
          const uint32_t crc = s->member_raw ^ gf2_mulmod(0xFFFFFFFFu, x_pow_8n(s->member_len)) ^ 0xFFFFFFFFu;
          const GzMember* me = ends[e].second;
          if (crc != me->crc || (uint32_t)s->member_len != me->isize)
            return fail(ctx, EXON_HIP_EINVAL, "gzip stream: %s mismatch at the end of a member (decoded on the GPU)", crc != me->crc ? "CRC-32" : "ISIZE");
          s->member_raw = 0;
          s->member_len = 0;
          ++s->stats.members;
          ++e;
        }
        if (i == n_pieces) break;
        const CrcPiece& pc = s->h_pieces[i];
        s->member_raw = gf2_mulmod(s->member_raw, pc.len == 65536 ? xp64k : x_pow_8n(pc.len)) ^ pc.raw;
        s->member_len += pc.len;
      }
    }
    s->win_k ^= 1;
    const uint64_t end_bit = last.end_bit;
    *consumed = (int64_t)(end_bit >> 3) - (int64_t)lead;
    *produced = (int64_t)off;
    s->start_bit = (uint32_t)(end_bit & 7);
    s->start_kind = (last.flags & F_AT_MEMBER) ? START_MEMBER : START_BLOCK;
    s->stats.out_bytes += off;
    s->stats.comp_bytes += (uint64_t)*consumed;
    ++s->stats.calls;
    if (ends_stream) {
      s->ended = true;
      *consumed = n_use;
      if (n_use != n_comp) return fail(ctx, EXON_HIP_EINVAL, "gzip: bytes behind the end of the stream");
    }
    if (stream_end) *stream_end = s->ended ? 1 : 0;
    return EXON_HIP_OK;
  }
  return fail(ctx, EXON_HIP_EINVAL, "gzip stream: the symbol scratch keeps overflowing");
}

}  // extern "C"
