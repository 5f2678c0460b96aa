// inflate.hip -- BGZF block inflate ON THE GPU: compressed blocks in HBM -> text / records in HBM.
//
// Reference path replaced: noodles bgzf::AsyncReader wrapped around the object-store byte stream in
// exon-core/src/datasources/vcf/file_opener/unindex_file_opener.rs:62-70, fastq/file_opener.rs:69-75 and the
// block-by-block AsyncBGZFReader of exon-core/src/streaming_bgzf.rs:56-64 (third-party arithmetic: RFC 1951 DEFLATE
// inside RFC 1952 members with the BGZF "BC" extra field, SAM specification section 4.1).  Nearly every VCF / BAM /
// FASTQ in the wild is BGZF-compressed, and host zlib is what bounds the scan then; BGZF blocks are independent
// (<= 64 KiB of text each), so a slab of compressed blocks crosses PCIe as it is (3-5x fewer bytes than text) and
// is inflated by one wavefront per block.
//
// One wavefront per block (k_inflate_w / k_inflate, one wavefront per workgroup, 4928 B of static LDS, 64 VGPRs, <= 79 SGPRs -> 32
// members per CU):
//   * Huffman tables live in LDS: a 9-bit (literal/length) and an 8-bit (distance) first-level table whose entries are complete
//     decode results (literal byte / length or distance base + extra-bit count + code length), built lane-parallel from the code
//     lengths (ballot ranks give the canonical codes); per-length first-code / count / offset arrays serve the longer codes;
//   * the last 1 KiB of output stays in an LDS ring: literals and near matches are LDS stores / LDS->LDS copies, farther matches
//     read the output already drained to HBM; the ring is drained in aligned 256-byte rows, one dword per lane;
//   * the symbol loop (round 5, default): wide_run -- 64 lanes decode the symbols that would start at 64 consecutive BIT offsets,
//     the true chain is followed with one v_readlane per symbol and the round's output (<= 64 bytes) is produced in one step;
//     its ordinary rounds are hand-written (wide_rounds_asm), and so are the common rare ones: literal and distance codes longer than
//     the tables are decoded inside the walk, a match that overlaps its own output takes its sources whole periods back, a round
//     ends in front of a symbol that starts beyond its 64 bytes.  Rounds 1-4's loops -- one symbol per step on the scalar unit
//     (symbol_run), on the vector unit with software pipelining (symbol_run_v) -- stay selectable (EXON_HIP_INFLATE_FLAVOR=0|1|2)
//     and serve the lane-parallel kernel's hand-backs.  Tuning logs: profiles/r1_tuning.md ... HISTORY.md section 7e.
// Every block reports a status; any failure makes the caller inflate on the host instead.  The CRC-32 of every
// inflated block is checked by a second kernel (k_crc32: 64 slices per block, combined in GF(2)[x] mod P with powers of x from a table).
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <algorithm>
#include <utility>

#include <cstring>
#include <vector>

#include "internal.h"

namespace {

#ifndef EXON_LIT_BITS
#define EXON_LIT_BITS 9
#endif
#ifndef EXON_DIST_BITS
#define EXON_DIST_BITS 8
#endif
constexpr int LIT_BITS = EXON_LIT_BITS, DIST_BITS = EXON_DIST_BITS, CL_BITS = 7;
constexpr int WAVES_PER_WG = 4;     // k_crc32
#ifndef EXON_INFLATE_RING
#define EXON_INFLATE_RING 2048
#endif
#ifndef EXON_INFLATE_LIT
#define EXON_INFLATE_LIT 2  // symbol loop: 0 = the compiler's (readable reference), 1 = literal_run, 2 = symbol_run
#endif
constexpr int INFLATE_RING = EXON_INFLATE_RING;  // bytes of recent output kept in LDS per wavefront (k_inflate_par: also its window)
// The serial kernel's ring is 1 KiB since round 4: 1024 + 3840 bytes of LDS per member are under the 5120-byte tier that lets a
// CU hold 32 workgroups (tools/occupancy_probe.hip); with the kernel at 64 VGPRs (__launch_bounds__(64, 8)) and 78 SGPRs (pinned
// registers s50-s71, amdgpu_num_sgpr(72)) a CU holds 32 members instead of 24.  One launch: 6144 members 4.3 ms, 7168 4.77,
// 8192 5.26 (+9.5 % throughput); the far copies the smaller ring adds are deferred ones.  With 7680-member slabs: .vcf.gz
// 58.6 -> 56 ms, BAM 63.9 -> 61, .fastq.gz 128 -> 119 on one box.  -DEXON_INFLATE_RING_SERIAL=2048 restores the old ring.
#ifndef EXON_INFLATE_RING_SERIAL
#define EXON_INFLATE_RING_SERIAL 1024
#endif
constexpr int INFLATE_RING_SERIAL = EXON_INFLATE_RING_SERIAL;

enum : int {
  INF_OK = 0,
  INF_BAD_BTYPE = 1,
  INF_BAD_STORED = 2,
  INF_BAD_LENGTHS = 3,
  INF_BAD_CODE = 4,
  INF_BAD_DISTANCE = 5,
  INF_OUTPUT_OVERRUN = 6,
  INF_INPUT_OVERRUN = 7,
  INF_SIZE_MISMATCH = 8,
  INF_BAD_CRC = 9,
};

// Length / distance base values and extra-bit counts of RFC 1951 section 3.2.5 in closed form (table lookups would be
// vector-memory loads on the serial decode chain):
//   length symbol s (0..28, = code - 257): s < 8: 3 + s, 0 bits; s = 28: 258; else e = (s - 4) >> 2, 3 + ((4 + (s & 3)) << e)
//   distance symbol d (0..29): d < 4: 1 + d, 0 bits; else e = (d - 2) >> 1, 1 + ((2 + (d & 1)) << e)
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
// order of the code-length code lengths {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15}, 5 bits each
constexpr uint64_t CL_ORDER_LO = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                                 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
constexpr uint64_t CL_ORDER_HI = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
__device__ __forceinline__ int cl_order(int i) {
  return i < 12 ? (int)((CL_ORDER_LO >> (5 * i)) & 31) : (int)((CL_ORDER_HI >> (5 * (i - 12))) & 31);
}

// ---- LDS layout: one wavefront per workgroup; [0, RING) recent output, then the tables -----------------------------
// A file-scope dynamic LDS array keeps the address space visible to the non-inlined helpers (ds_* instructions, not
// flat_*).

// First-level table entries are complete decode results, so the symbol loop does no arithmetic on symbol numbers:
//   bits 0-3 code length (0 = longer than the table, or no such code), 4-7 number of extra bits, 8 literal, 9 end of
//   block, 10 a code that exists but may not be used (286, 287, distance 30, 31); 16-31 literal byte / length base /
//   distance base / code-length symbol
constexpr uint32_t E_LIT = 1u << 8, E_EOB = 1u << 9, E_INVALID = 1u << 10;
// bit 11: a length / distance symbol the hand-written vector loop may take as it is (in the table, neither end of block nor
// invalid): with the three flags above in the same byte, "byte 1 == 8" is ONE sdwa compare instead of mask, add, compare
constexpr uint32_t E_FAST = 1u << 11;

struct ClTables {  // the code-length code is dead once the two main codes are built: it shares dist_lut's space
  uint32_t cl_lut[1 << CL_BITS];
  uint16_t cl_sym[20];
  uint16_t cl_count[16];
};
struct WaveLds {
  // The code lengths read from a block header are only needed until the tables are built, and the literal/length table -- built
  // LAST, from lengths its builder has taken into registers before it writes the first entry -- is dead from the header on:
  // they share its bytes (round 4: 320 bytes less per member = 26 instead of 24 resident waves per CU; the inflate's
  // throughput grows with them, profiles/r4_inflate_waves_per_cu.log)
  union {
    uint32_t lit_lut[1 << LIT_BITS];
    uint8_t lens[288 + 32];
  };
  union {
    uint32_t dist_lut[1 << DIST_BITS];
    ClTables cl;
  };
  uint16_t lit_sym[288];             // symbols in canonical order (bit-serial decode of long codes)
  uint16_t dist_sym[32];
  uint16_t lit_count[16], dist_count[16];  // codes per length
  // per length: first canonical code / symbols of shorter lengths.  Scratch of every build_code; the literal/length
  // code is built LAST, so between table builds these hold ITS values, which decode_long uses.
  uint16_t first[16], offs[16];
  // the same two of the DISTANCE code, kept when its table is built: the wide loop decodes distance codes longer than the table
  // itself (they are the largest group of symbols that used to leave its rounds: 0.8-1.6 % of the rounds), and decode_long takes
  // all candidate lengths at once for them too.  (first / offs / dfirst / doffs and lit_count / dist_count are read as two arrays
  // of 32 by the wide loop: keep them adjacent.)
  uint16_t dfirst[16], doffs[16];
};
static_assert(__builtin_offsetof(WaveLds, offs) == __builtin_offsetof(WaveLds, first) + 32 && __builtin_offsetof(WaveLds, dfirst) == __builtin_offsetof(WaveLds, first) + 64 &&
                  __builtin_offsetof(WaveLds, doffs) == __builtin_offsetof(WaveLds, first) + 96 && __builtin_offsetof(WaveLds, dist_count) == __builtin_offsetof(WaveLds, lit_count) + 32,
              "the wide loop's prologue reads these as arrays of 32");
// (the code-length tables share the bytes of dist_lut: they are dead once the lengths are read)
enum { CODE_LIT = 0, CODE_DIST = 1, CODE_CL = 2 };

constexpr int INF_WAVES = 1;  // wavefronts (= BGZF blocks) per workgroup (1: LDS addresses need no per-wave base)
// static (not dynamic) LDS: with one kernel using it the addresses are compile-time constants in every function
__shared__ __attribute__((aligned(16))) uint8_t smem[INF_WAVES * ((INFLATE_RING + sizeof(WaveLds) + 15) & ~size_t(15))];
#if EXON_INFLATE_RING_SERIAL != EXON_INFLATE_RING  // a kernel is charged for the arrays it references: each kernel its own
__shared__ __attribute__((aligned(16))) uint8_t smem_serial[INF_WAVES * ((INFLATE_RING_SERIAL + sizeof(WaveLds) + 15) & ~size_t(15))];
#else
#define smem_serial smem
#endif
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
template <int RING>
__device__ __forceinline__ uint8_t* wave_ring();
template <int RING>
__device__ __forceinline__ WaveLds* wave_lds() { return reinterpret_cast<WaveLds*>(wave_ring<RING>() + RING); }

template <int RING>
__device__ __forceinline__ uint8_t* wave_ring() {
  constexpr uint32_t STRIDE = (RING + (uint32_t)sizeof(WaveLds) + 15u) & ~15u;
  uint8_t* base = RING == INFLATE_RING ? smem : smem_serial;
  if (INF_WAVES == 1) return base;
  return base + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * STRIDE;
}

struct Code {
  uint32_t* lut;
  uint16_t* sym;
  uint16_t* count;
  uint16_t* first;  // per length: first canonical code
  uint16_t* offs;   // per length: symbols of shorter lengths
  int bits;
};
template <int RING>
__device__ __forceinline__ Code code_of(int which) {
  WaveLds* L = wave_lds<RING>();
  if (which == CODE_LIT) return Code{L->lit_lut, L->lit_sym, L->lit_count, L->first, L->offs, LIT_BITS};
  if (which == CODE_DIST) return Code{L->dist_lut, L->dist_sym, L->dist_count, L->first, L->offs, DIST_BITS};
  return Code{L->cl.cl_lut, L->cl.cl_sym, L->cl.cl_count, L->first, L->offs, CL_BITS};
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <class T>
__device__ __forceinline__ T* unip(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  return reinterpret_cast<T*>(((uint64_t)uniu((uint32_t)(v >> 32)) << 32) | uniu((uint32_t)v));
}

// Bit reader over dword-aligned global memory; all fields wave-uniform except the window register (one dword per
// lane: the aligned 256-byte window that holds dword `widx`).  A window is reloaded synchronously when exhausted: once
// per 256 input bytes (~250 symbols).  Reading past the block's end is detected by the callers (`overrun()`, checked
// once per 256 output bytes), so up to ~2 KiB behind a corrupt block may be read: buffers carry 4 KiB of padding.
struct BitReader {
  const uint32_t* base;
  uint32_t widx;   // next dword to take
  uint32_t cur;    // per lane: base[(widx & ~63) + lane]
  uint64_t buf;
  int cnt;
  uint32_t limit;  // dword index one past the last dword that may hold this block's bits

  __device__ __forceinline__ void init(const uint8_t* data, uint32_t byte_off, uint32_t byte_end) {
    base = reinterpret_cast<const uint32_t*>(data);
    widx = byte_off >> 2;
    limit = (byte_end + 3) >> 2;
    cur = base[(widx & ~63u) + lane_id()];
    buf = 0;
    cnt = 0;
    refill();
    refill();
    const int skip = (int)(byte_off & 3) * 8;
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
  // bytes of the input consumed so far (whole bytes; the partial byte counts as consumed)
  __device__ __forceinline__ uint32_t byte_pos() const { return widx * 4u - (uint32_t)(cnt >> 3); }
  __device__ __forceinline__ bool overrun() const { return widx > limit + 2u; }  // two dwords of look-ahead are legitimate
  // values that crossed a function boundary arrive in VGPRs: make them scalar again
  __device__ __forceinline__ void make_uniform() {
    base = unip(base);
    widx = uniu(widx);
    buf = ((uint64_t)uniu((uint32_t)(buf >> 32)) << 32) | uniu((uint32_t)buf);
    cnt = uni(cnt);
    limit = uniu(limit);
  }
};

// Table entry (without the code length) of symbol s of code `which`
__device__ __forceinline__ uint32_t entry_for(int which, int s) {
  if (which == CODE_CL) return (uint32_t)s << 16;
  uint32_t base;
  int extra;
  if (which == CODE_LIT) {
    if (s < 256) return E_LIT | ((uint32_t)s << 16);
    if (s == 256) return E_EOB;
    if (s > 285) return E_INVALID;
    length_code(s - 257, &base, &extra);
  } else {
    if (s > 29) return E_INVALID;
    distance_code(s, &base, &extra);
  }
  return (base << 16) | ((uint32_t)extra << 4) | E_FAST;
}

// Build code `which` from lens[0..n): counts, canonical order, first-level table.  Returns 0 for an over-subscribed
// code, or an incomplete one that is not the single-code case DEFLATE allows.
// (96 SGPRs in all -- 90 + VCC, flat scratch, XNACK -- is what lets a SIMD hold 7 waves of a kernel, 80 what lets it hold 8:
//  tools/occupancy_probe.hip.  The per-length totals of this function are wave-uniform; as four 16-entry arrays they took 102
//  scalar registers and capped the whole kernel at 6 waves per SIMD.  They live in the LANES of vector registers now: lane q
//  holds the value for code length q.)
template <int RING>
__device__ __noinline__ int build_code_impl(int which, int lens_off, int n) {
  which = uni(which);
  lens_off = uni(lens_off);
  n = uni(n);
  WaveLds* L = wave_lds<RING>();
  const Code c = code_of<RING>(which);
  const uint8_t* lens = L->lens + lens_off;
  const int lane = (int)lane_id();
  const int size = 1 << c.bits;
  // this lane's lengths (symbols lane, lane + 64, ...: n <= 288) BEFORE the table is touched: the literal/length table
  // overlays the lengths
  int myl[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int s = k * 64 + lane;
    myl[k] = s < n ? (int)lens[s] : 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int i = lane; i < size; i += 64) c.lut[i] = 0;
  // totals per length, by ballots: lane q accumulates the symbols of length q
  int run_l = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (k * 64 >= n) break;
    const int l = myl[k];
#pragma unroll 1  // (rolled: unrolled, the 5 x 15 ballots kept ~90 scalar registers live)
    for (int q = 1; q < 16; ++q) {
      const int cq = __popcll(__ballot(l == q));
      if (lane == q) run_l += cq;
    }
  }
  int left = 1, total = 0;
  int first_l = 0, offs_l = 0;  // lane q: first canonical code of length q / symbols of shorter lengths
  {
    int code = 0, off = 0;
#pragma unroll 1
    for (int q = 1; q < 16; ++q) {
      const int rq = __builtin_amdgcn_readlane(run_l, q);
      code <<= 1;
      if (lane == q) {
        first_l = code;
        offs_l = off;
      }
      code += rq;
      off += rq;
      left = (left << 1) - rq;
      if (left < 0) return 0;  // over-subscribed
      total += rq;
    }
  }
  if (left > 0 && total > 1) return 0;  // incomplete (RFC 1951 allows it only for a single distance code)
  if (lane < 16) {
    c.count[lane] = (uint16_t)run_l;
    c.first[lane] = (uint16_t)first_l;
    c.offs[lane] = (uint16_t)offs_l;
    if (which == CODE_DIST) {
      L->dfirst[lane] = (uint16_t)first_l;
      L->doffs[lane] = (uint16_t)offs_l;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // every symbol gets its canonical code (rank inside its length class, in symbol order); fill the tables
  int seen_l = 0;  // lane q: symbols of length q placed so far
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
      const int code = (int)c.first[l] + rank;
      c.sym[(int)c.offs[l] + rank] = (uint16_t)s;
      if (l <= c.bits) {
        const unsigned rev = __brev((unsigned)code) >> (32 - l);
        const uint32_t e = entry_for(which, s) | (uint32_t)l;
        for (unsigned k2 = rev; k2 < (unsigned)size; k2 += 1u << l) c.lut[k2] = e;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return 1;
}
template <int RING>
__device__ __forceinline__ bool build_code(int which, int lens_off, int n) {
  return uni(build_code_impl<RING>(which, lens_off, n)) != 0;
}

// Canonical decode of a code longer than the first-level table.  Returns symbol << 8 | length, or -1 for a code that
// does not exist.  Literal/length codes: all candidate lengths at once -- lane L (1..15) takes the first L bits as a
// code of that length and tests it against the codes of its length (first[L] <= code < first[L] + count[L]); codes
// are prefix-free, so at most one length matches.  (High-entropy payloads -- BAM -- come here for ~2 % of their bytes;
// the bit-serial loop cost ~150 instructions a time: +21 % on BAM.)  Distance and code-length codes, which hardly
// ever get here, kept the bit-serial loop until round 5 (the distance code's first[] / offs[] now outlive its build: dfirst / doffs).
template <int RING>
__device__ __noinline__ int decode_long(int which, uint32_t bits) {
  which = uni(which);
  bits = uniu(bits);
  const Code c = code_of<RING>(which);
  if (which == CODE_LIT || which == CODE_DIST) {
    const WaveLds* L = wave_lds<RING>();
    const uint16_t* first = which == CODE_LIT ? c.first : L->dfirst;  // (c.first / c.offs are the literal/length code's between builds)
    const uint16_t* offs = which == CODE_LIT ? c.offs : L->doffs;
    const int len = (int)lane_id() & 15;  // lanes 16.. repeat 0..15: only the ballot's low 16 bits are used
    const uint32_t code = len ? __brev(bits) >> (32 - len) : 0u;  // DEFLATE packs Huffman codes starting from their MSB
    const uint32_t rel = code - (uint32_t)first[len];
    const bool hit = len != 0 && rel < (uint32_t)c.count[len];
    const uint32_t m = (uint32_t)__ballot(hit) & 0xFFFEu;
    if (m == 0) return -1;
    const int l = __ffs((int)m) - 1;
    const int idx = __builtin_amdgcn_readlane((int)((uint32_t)offs[len] + rel), l);
    return ((int)c.sym[idx] << 8) | l;
  }
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= 15; ++len) {
    code |= (int)(bits & 1u);
    bits >>= 1;
    const int n = uni((int)c.count[len]);
    if (code - n < first) return (uni((int)c.sym[index + (code - first)]) << 8) | len;
    index += n;
    first += n;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// Decode one symbol (wave-uniform): the table entry with the code's bits consumed (the extra bits are not).
// E_INVALID for a code that does not exist.
template <int RING, int WHICH>
__device__ __forceinline__ uint32_t decode_symbol(BitReader& br, const WaveLds* L) {
  const uint32_t* lut = WHICH == CODE_LIT ? L->lit_lut : WHICH == CODE_DIST ? L->dist_lut : L->cl.cl_lut;
  constexpr int BITS = WHICH == CODE_LIT ? LIT_BITS : WHICH == CODE_DIST ? DIST_BITS : CL_BITS;
  uint32_t e = uniu(lut[br.peek(BITS)]);
  if (__builtin_expect((e & 15u) == 0, 0)) {
    const int r = uni(decode_long<RING>(WHICH, (uint32_t)br.buf));
    e = r < 0 ? (uint32_t)E_INVALID : entry_for(WHICH, r >> 8) | (uint32_t)(r & 255);  // lengths <= 15 fit the length field
  }
  br.drop((int)(e & 15u));  // one merge point for the short and the long code (no flag threaded through the hot path)
  return e;
}

struct Block {  // == exon_hip_bgzf_block
  uint32_t comp_offset, comp_size, out_offset, out_size, crc32, reserved;
};

struct Out {  // output cursor of one BGZF block (wave-uniform)
  uint8_t* out;
  uint32_t begin, end;  // [begin, end) absolute indexes into `out`
  uint32_t pos;         // next output byte
  uint32_t drained;     // output below this index is in HBM; [drained, pos) is only in the ring
  __device__ __forceinline__ void make_uniform() {
    out = unip(out);
    begin = uniu(begin);
    end = uniu(end);
    pos = uniu(pos);
    drained = uniu(drained);
  }
};

// ring -> HBM for [drained, limit), in aligned 256-byte rows (one dword per lane) where possible
template <int RING>
__device__ __forceinline__ void drain_to(Out& o, uint32_t limit) {
  constexpr uint32_t M = RING - 1;
  const uint32_t lane = lane_id();
  const uint8_t* ring = wave_ring<RING>();
  while (o.drained < limit) {
    const uint32_t row_end = min(limit, (o.drained | 255u) + 1u);
    if (((o.drained | row_end) & 3u) == 0) {
      const uint32_t p = o.drained + 4u * lane;
      if (p < row_end) *reinterpret_cast<uint32_t*>(o.out + p) = *reinterpret_cast<const uint32_t*>(ring + (p & M));
    } else {
      for (uint32_t p = o.drained + lane; p < row_end; p += 64) o.out[p] = ring[p & M];
    }
    o.drained = row_end;
  }
}

// Rare paths of the symbol loop live in their own functions so that the hot loop stays small for the compiler
// (fewer live values and exits = fewer scalar copies per iteration).
template <int RING>
__device__ __noinline__ uint32_t drain_rows(uint8_t* out, uint32_t drained, uint32_t limit) {
  Out o{unip(out), 0, 0, 0, uniu(drained)};
  drain_to<RING>(o, uniu(limit));
  return o.drained;
}
template <int RING>
__device__ __noinline__ void copy_overlapping(uint32_t pos, uint32_t d, uint32_t len) {
  constexpr uint32_t M = RING - 1;
  pos = uniu(pos);
  d = uniu(d);
  len = uniu(len);
  uint8_t* ring = wave_ring<RING>();
  const uint32_t lane = lane_id();
  // 64 bytes per step read strictly older bytes (modular source), then write
  for (uint32_t j0 = 0; j0 < len; j0 += 64) {
    const uint32_t j = j0 + lane;
    const uint8_t v = ring[(pos - d + j % d) & M];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (j < len) ring[(pos + j) & M] = v;
  }
}
struct SymResult {
  BitReader br;
  Out o;
  int err;
};

// The symbols of one DEFLATE block.  RING bytes of the most recent output stay in LDS: matches whose distance fits are
// LDS -> LDS copies (no memory latency on the decode chain); farther ones read the output already drained to HBM.
// Output bounds are checked when rows are drained (the ring wraps harmlessly), input bounds when a window is reloaded.
// LDS byte address of a __shared__ object
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// The literal loop, hand-written: refill -> first-level lookup -> store -> next in 22 instructions per literal (the
// compiler's version of the same loop, EXON_INFLATE_LIT=0: 26, with two more taken branches).  Table index and ring
// address are computed on the vector ALU, the row check rides on the SCC of s_and.  Worth +2 % (VCF text), +5 % (BAM),
// +13 % (FASTQ) on the same box; a batched variant (one lookup for all 64 bit offsets, the chain of literals followed
// with v_readlane) was no faster on literal-heavy data and slower on match-heavy data (profiles/r1_tuning.md).
// Leaves with   0: `e` is a first-level entry that is not a literal (its code bits are NOT consumed; 0 = long code)
//               1: a literal was stored and o.pos entered a new 256-byte row
// `vpos` is the per-lane copy of `pos` (ring addresses), `lane4` = 4 * lane.  Relies on the ring sitting at LDS
// address 0 (checked by the kernel).
template <int RING>
__device__ __forceinline__ uint32_t literal_run(BitReader& br, uint32_t& pos, uint32_t& vpos, uint32_t lane4, uint32_t& e) {
  uint32_t why, vt, ve;
  uint64_t buf = br.buf;
  int cnt = br.cnt;
  uint32_t widx = br.widx;
  asm volatile(
      "L_lit_loop%=:\n"
      "  s_cmp_gt_i32 s52, 32\n"
      "  s_cbranch_scc1 L_lit_have%=\n"
      "  s_waitcnt vmcnt(0)\n"
      "  v_readlane_b32 s60, %[cur], s53\n"
      "  s_mov_b32 s61, 0\n"
      "  s_lshl_b64 s[60:61], s[60:61], s52\n"
      "  s_or_b64 s[50:51], s[50:51], s[60:61]\n"
      "  s_add_i32 s52, s52, 32\n"
      "  s_add_i32 s53, s53, 1\n"
      "  s_and_b32 s57, s53, 63\n"
      "  s_cbranch_scc1 L_lit_have%=\n"
      "  v_lshl_add_u32 %[vt], s53, 2, %[lane4]\n"
      "  global_load_dword %[cur], %[vt], s[58:59]\n"
      "L_lit_have%=:\n"
      "  v_lshlrev_b32 %[vt], 2, s50\n"
      "  v_and_b32 %[vt], %[lutmask], %[vt]\n"
      "  ds_read_b32 %[ve], %[vt] offset:%[lutoff]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_readfirstlane_b32 s55, %[ve]\n"
      "  s_bitcmp1_b32 s55, 8\n"
      "  s_cbranch_scc0 L_lit_other%=\n"
      "  s_and_b32 s57, s55, 15\n"
      "  s_lshr_b64 s[50:51], s[50:51], s57\n"
      "  s_sub_i32 s52, s52, s57\n"
      "  v_and_b32 %[vt], %[ringmask], %[vpos]\n"
      "  ds_write_b8_d16_hi %[vt], %[ve]\n"
      "  v_add_u32 %[vpos], 1, %[vpos]\n"
      "  s_add_i32 s54, s54, 1\n"
      "  s_and_b32 s57, s54, 0xff\n"
      "  s_cbranch_scc1 L_lit_loop%=\n"
      "  s_mov_b32 s56, 1\n"
      "  s_branch L_lit_out%=\n"
      "L_lit_other%=:\n"
      "  s_mov_b32 s56, 0\n"
      "L_lit_out%=:\n"
      "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      : [buf] "+{s[50:51]}"(buf), [cnt] "+{s52}"(cnt), [widx] "+{s53}"(widx), [pos] "+{s54}"(pos), [e] "={s55}"(e), [why] "={s56}"(why),
        [cur] "+v"(br.cur), [vpos] "+v"(vpos), [vt] "=&v"(vt), [ve] "=&v"(ve)
      : [base] "{s[58:59]}"(br.base), [lane4] "v"(lane4), [lutmask] "i"(((1 << LIT_BITS) - 1) << 2), [ringmask] "i"(RING - 1),
        [lutoff] "i"(RING + (int)__builtin_offsetof(WaveLds, lit_lut))
      : "s57", "s60", "s61", "scc", "memory");
  // asm results count as divergent for the compiler even in scalar registers: say otherwise (folds to plain copies)
  br.buf = ((uint64_t)uniu((uint32_t)(buf >> 32)) << 32) | uniu((uint32_t)buf);
  br.cnt = uni(cnt);
  br.widx = uniu(widx);
  pos = uniu(pos);
  e = uniu(e);
  return uniu(why);
}

// literal_run plus the common match, in one block: a length code and a distance code that both sit in the first-level
// tables, at most 64 bytes, source either in the ring without overlap or far enough back to be in HBM already --
// decoded, copied (one masked read + write) and advanced without leaving the loop.  Anything else leaves with what has
// been decoded so far:
//   0: `e` is a first-level entry the loop does not handle (long code, end of block, invalid); nothing consumed
//   1: o.pos entered a new 256-byte row (after a literal or a match)
//   2: a match is decoded (len, d; all its bits consumed) but not copied: overlapping, longer than 64, or a bad distance
//   3: a length is decoded (len) and consumed, the buffer refilled; the distance code needs the slow path
template <int RING>
__device__ __forceinline__ uint32_t symbol_run(BitReader& br, uint32_t& pos, uint32_t& vpos, uint32_t lane, uint32_t lane4, uint32_t begin,
                                               const uint8_t* out, uint32_t& e, uint32_t& len, uint32_t& d) {
  uint32_t why, vt, ve;
  uint64_t buf = br.buf;
  int cnt = br.cnt;
  uint32_t widx = br.widx;
#define EXON_REFILL(tag)                                \
  "  s_cmp_gt_i32 s52, 32\n"                            \
  "  s_cbranch_scc1 L_have_" tag "%=\n"                 \
  "  s_waitcnt vmcnt(0)\n"                              \
  "  v_readlane_b32 s60, %[cur], s53\n"                 \
  "  s_mov_b32 s61, 0\n"                                \
  "  s_lshl_b64 s[60:61], s[60:61], s52\n"              \
  "  s_or_b64 s[50:51], s[50:51], s[60:61]\n"           \
  "  s_add_i32 s52, s52, 32\n"                          \
  "  s_add_i32 s53, s53, 1\n"                           \
  "  s_and_b32 s57, s53, 63\n"                          \
  "  s_cbranch_scc1 L_have_" tag "%=\n"                 \
  "  v_lshl_add_u32 %[vt], s53, 2, %[lane4]\n"          \
  "  global_load_dword %[cur], %[vt], s[58:59]\n"       \
  "L_have_" tag "%=:\n"
  asm volatile(
      "L_sym_loop%=:\n" EXON_REFILL("l")
      "  v_lshlrev_b32 %[vt], 2, s50\n"
      "  v_and_b32 %[vt], %[lutmask], %[vt]\n"
      "  ds_read_b32 %[ve], %[vt] offset:%[lutoff]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_readfirstlane_b32 s55, %[ve]\n"
      "  s_bitcmp1_b32 s55, 8\n"
      "  s_cbranch_scc0 L_sym_match%=\n"
      "  s_and_b32 s57, s55, 15\n"
      "  s_lshr_b64 s[50:51], s[50:51], s57\n"
      "  s_sub_i32 s52, s52, s57\n"
      "  v_and_b32 %[vt], %[ringmask], %[vpos]\n"
      "  ds_write_b8_d16_hi %[vt], %[ve]\n"
      "  v_add_u32 %[vpos], 1, %[vpos]\n"
      "  s_add_i32 s54, s54, 1\n"
      "  s_and_b32 s57, s54, 0xff\n"
      "  s_cbranch_scc1 L_sym_loop%=\n"
      "  s_branch L_sym_row%=\n"
      // ---- not a literal: a length code?
      "L_sym_match%=:\n"
      "  s_and_b32 s57, s55, 15\n"            // code length; SCC = (it is in the table)
      "  s_cbranch_scc0 L_sym_exit0%=\n"
      "  s_and_b32 s64, s55, 0x600\n"         // end of block / invalid
      "  s_cbranch_scc1 L_sym_exit0%=\n"
      "  s_lshr_b64 s[50:51], s[50:51], s57\n"
      "  s_sub_i32 s52, s52, s57\n"
      "  s_lshr_b32 s62, s55, 16\n"           // length base
      "  s_bfe_u32 s57, s55, 0x40004\n"       // extra bits; SCC = (any)
      "  s_cbranch_scc0 L_sym_len%=\n"
      "  s_bfm_b32 s64, s57, 0\n"
      "  s_and_b32 s64, s50, s64\n"
      "  s_add_i32 s62, s62, s64\n"
      "  s_lshr_b64 s[50:51], s[50:51], s57\n"
      "  s_sub_i32 s52, s52, s57\n"
      "L_sym_len%=:\n" EXON_REFILL("m")
      // ---- the distance
      "  v_lshlrev_b32 %[vt], 2, s50\n"
      "  v_and_b32 %[vt], %[dmask], %[vt]\n"
      "  ds_read_b32 %[ve], %[vt] offset:%[dlut]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_readfirstlane_b32 s65, %[ve]\n"
      "  s_and_b32 s57, s65, 15\n"
      "  s_cbranch_scc0 L_sym_exit3%=\n"      // long or nonexistent code
      "  s_bitcmp1_b32 s65, 10\n"
      "  s_cbranch_scc1 L_sym_exit3%=\n"      // invalid symbol (30, 31)
      "  s_lshr_b64 s[50:51], s[50:51], s57\n"
      "  s_sub_i32 s52, s52, s57\n"
      "  s_lshr_b32 s63, s65, 16\n"           // distance base
      "  s_bfe_u32 s57, s65, 0x40004\n"       // extra bits; SCC = (any)
      "  s_cbranch_scc0 L_sym_dist%=\n"
      "  s_bfm_b32 s64, s57, 0\n"
      "  s_and_b32 s64, s50, s64\n"
      "  s_add_i32 s63, s63, s64\n"
      "  s_lshr_b64 s[50:51], s[50:51], s57\n"
      "  s_sub_i32 s52, s52, s57\n"
      "L_sym_dist%=:\n"
      // ---- the copies the loop does itself: d <= history, len <= 64, and either len <= d <= NEAR (ring -> ring) or
      //      d > NEAR (the source is below `drained`, i.e. in HBM already)
      "  s_sub_i32 s57, s54, s66\n"
      "  s_cmp_gt_u32 s63, s57\n"
      "  s_cbranch_scc1 L_sym_exit2%=\n"
      "  s_cmp_gt_u32 s62, 64\n"
      "  s_cbranch_scc1 L_sym_exit2%=\n"
      "  v_cmp_gt_u32 vcc, s62, %[lane]\n"     // lanes below len
      "  s_sub_i32 s57, s54, s63\n"            // first source byte
      "  s_cmp_gt_u32 s63, %[near]\n"
      "  s_cbranch_scc1 L_sym_far%=\n"
      "  s_cmp_gt_u32 s62, s63\n"
      "  s_cbranch_scc1 L_sym_exit2%=\n"       // overlapping run
      "  s_and_saveexec_b64 s[60:61], vcc\n"
      "  v_add_u32 %[vt], s57, %[lane]\n"
      "  v_and_b32 %[vt], %[ringmask], %[vt]\n"
      "  ds_read_u8 %[ve], %[vt]\n"
      "  v_add_u32 %[vt], s54, %[lane]\n"
      "  v_and_b32 %[vt], %[ringmask], %[vt]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  ds_write_b8 %[vt], %[ve]\n"
      "  s_mov_b64 exec, s[60:61]\n"
      "  s_branch L_sym_adv%=\n"
      "L_sym_far%=:\n"
      "  s_and_saveexec_b64 s[60:61], vcc\n"
      "  v_add_u32 %[vt], s57, %[lane]\n"
      "  global_load_ubyte %[ve], %[vt], s[68:69]\n"
      "  v_add_u32 %[vt], s54, %[lane]\n"
      "  v_and_b32 %[vt], %[ringmask], %[vt]\n"
      "  s_waitcnt vmcnt(0)\n"
      "  ds_write_b8 %[vt], %[ve]\n"
      "  s_mov_b64 exec, s[60:61]\n"
      "L_sym_adv%=:\n"
      "  s_add_i32 s57, s54, s62\n"
      "  s_xor_b32 s64, s57, s54\n"
      "  s_mov_b32 s54, s57\n"
      "  v_mov_b32 %[vpos], s57\n"
      "  s_lshr_b32 s64, s64, 8\n"            // SCC = a 256-byte row boundary was crossed
      "  s_cbranch_scc0 L_sym_loop%=\n"
      "L_sym_row%=:\n"
      "  s_mov_b32 s56, 1\n"
      "  s_branch L_sym_out%=\n"
      "L_sym_exit0%=:\n"
      "  s_mov_b32 s56, 0\n"
      "  s_branch L_sym_out%=\n"
      "L_sym_exit2%=:\n"
      "  s_mov_b32 s56, 2\n"
      "  s_branch L_sym_out%=\n"
      "L_sym_exit3%=:\n"
      "  s_mov_b32 s56, 3\n"
      "L_sym_out%=:\n"
      "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      : [buf] "+{s[50:51]}"(buf), [cnt] "+{s52}"(cnt), [widx] "+{s53}"(widx), [pos] "+{s54}"(pos), [e] "={s55}"(e), [why] "={s56}"(why),
        [len] "={s62}"(len), [d] "={s63}"(d), [cur] "+v"(br.cur), [vpos] "+v"(vpos), [vt] "=&v"(vt), [ve] "=&v"(ve)
      : [base] "{s[58:59]}"(br.base), [begin] "{s66}"(begin), [out] "{s[68:69]}"(out), [lane] "v"(lane), [lane4] "v"(lane4), [lutmask] "i"(((1 << LIT_BITS) - 1) << 2),
        [dmask] "i"(((1 << DIST_BITS) - 1) << 2), [ringmask] "i"(RING - 1), [near] "i"(RING - 258),
        [lutoff] "i"(RING + (int)__builtin_offsetof(WaveLds, lit_lut)), [dlut] "i"(RING + (int)__builtin_offsetof(WaveLds, dist_lut))
      : "s57", "s60", "s61", "s64", "s65", "vcc", "scc", "memory");
#undef EXON_REFILL
  br.buf = ((uint64_t)uniu((uint32_t)(buf >> 32)) << 32) | uniu((uint32_t)buf);
  br.cnt = uni(cnt);
  br.widx = uniu(widx);
  pos = uniu(pos);
  e = uniu(e);
  len = uniu(len);
  d = uniu(d);
  return uniu(why);
}

// symbol_run on the VECTOR unit, software-pipelined (round 4).
// (1) Where the instructions issue.  The loop above keeps the bit buffer in scalar registers, so nearly every instruction of a
// symbol goes through the CU's ONE scalar ALU, which issues one instruction per clock for the whole CU (tools/issue_rate.hip:
// 1.02 per clock and CU from 16 waves up).  The four SIMD-32 units of a gfx950 CU issue a wave64 vector instruction every two
// clocks EACH (measured 1.8-1.95 per clock and CU on the same dependent chains; a v_cmp + s_cbranch_vccnz pair costs what
// s_cmp + s_cbranch_scc costs), and a wave-uniform value can just as well live in a vector register with all 64 lanes
// computing the same thing.  So here the bit buffer, the bit count, the table entries, length and distance are VGPRs; what
// stays scalar is what is cheap there: the output position (ring address, row test), the window index of the refill, the
// exec mask of the copy.  A plain translation of symbol_run (8 vector + 3 scalar instructions per literal instead of 5 + 10)
// measured the SAME throughput as the scalar loop on every format (profiles/r4_inflate_flavor_v1.log), and so did making the
// far copies free (r4_inflate_far_nowait.log: +6..12 %): 24 waves per CU are bound by each wave's own dependency chain
// buffer -> table index -> LDS lookup (~100 clocks) -> code length -> shift, not by an issue port.
// (2) So the chain is what this version shortens.  A first-level lookup needs 9 valid bits and the buffer never holds fewer
// than 12 where one is issued, so the NEXT symbol's lookup goes out as soon as the current code's bits are shifted away --
// before the literal is stored, the counters move, the buffer is refilled or the match is copied; all of that now runs under
// the lookup's latency.  Length + extra bits (and distance + extra bits) leave the buffer with ONE 64-bit shift, the extra
// value comes from a v_bfe with register operands.  The refill is out of line (the common no-refill case falls through).
// Same contract as symbol_run: same `why` codes, same registers in and out; a lookup in flight at an exit is dropped.
// Hazards the assembler does not see inside an asm block (gfx940 family): a VALU-written SGPR needs 2 wait states before a
// VALU reads it (v_readlane -> v_lshlrev_b64 in the refill: two scalar instructions in between), a VALU-written VGPR 1 before
// v_readfirstlane reads it (the exits start with scalar instructions).  VALU-written VCC / SGPRs read by the scalar unit or by
// s_cbranch_vcc* are interlocked.
template <int RING>
__device__ __forceinline__ uint32_t symbol_run_v(BitReader& br, uint32_t& pos, uint32_t& vpos, uint32_t lane, uint32_t lane4, uint32_t begin,
                                                 const uint8_t* out, uint32_t& e, uint32_t& len, uint32_t& d) {
  uint32_t why, vt, ve, vfa, vfb;
  uint64_t buf = br.buf;
  int cnt = br.cnt;
  uint32_t widx = br.widx;
  // v[48:49] bit buffer, v50 bit count, v51 literal/length entry, v52 distance entry, v53 length, v54 distance, v55 code
  // length, v[32:33] / v34 / v35 / v38 scratch, v37 ring address of `pos`, v39 first source byte -- below v64, so that the kernel
  // fits the 64 registers that let 8 waves share a SIMD (v32-v39 are callee-saved in the AMDGPU calling convention: the
  // non-inlined decode_symbols saves them once per DEFLATE block)
#define EXON_REFILL_BODY(tag)                           \
  "  s_and_b32 s57, s53, 63\n"                          \
  "  s_cbranch_scc1 L_vwin_" tag "%=\n"                 \
  "  s_waitcnt vmcnt(0)\n"                              \
  "L_vwin_" tag "%=:\n"                                 \
  "  v_readlane_b32 s60, %[cur], s53\n"                 \
  "  s_mov_b32 s61, 0\n"                                \
  "  s_add_i32 s53, s53, 1\n"                           \
  "  v_lshlrev_b64 v[32:33], v50, s[60:61]\n"           \
  "  v_or_b32 v48, v48, v32\n"                          \
  "  v_or_b32 v49, v49, v33\n"                          \
  "  v_add_u32 v50, 32, v50\n"                          \
  "  s_and_b32 s57, s53, 63\n"
#define EXON_REFILL_V(tag)                              \
  "  v_cmp_lt_i32 vcc, 32, v50\n"                       \
  "  s_cbranch_vccnz L_vhave_" tag "%=\n"               \
  EXON_REFILL_BODY(tag)                                 \
  "  s_cbranch_scc1 L_vhave_" tag "%=\n"                \
  "  v_lshl_add_u32 %[vt], s53, 2, %[lane4]\n"          \
  "  global_load_dword %[cur], %[vt], s[58:59]\n"       \
  "L_vhave_" tag "%=:\n"
#define EXON_REFILL_OUT(tag)                            \
  "L_vrefill_" tag "%=:\n"                              \
  EXON_REFILL_BODY(tag)                                 \
  "  s_cbranch_scc1 L_vback_" tag "%=\n"                \
  "  v_lshl_add_u32 %[vt], s53, 2, %[lane4]\n"          \
  "  global_load_dword %[cur], %[vt], s[58:59]\n"       \
  "  s_branch L_vback_" tag "%=\n"
// a deferred far copy lands in the ring: the lanes below its length (s71) write their byte at s70 + lane
#define EXON_FAR_COMPLETE(vx)                           \
  "  v_cmp_gt_u32_e64 s[60:61], s71, %[lane]\n"        \
  "  s_and_saveexec_b64 s[64:65], s[60:61]\n"           \
  "  v_add_u32 %[vt], s70, %[lane]\n"                  \
  "  v_and_b32 %[vt], %[ringmask], %[vt]\n"             \
  "  ds_write_b8 %[vt], " vx "\n"                       \
  "  s_mov_b64 exec, s[64:65]\n"
#define EXON_FAR_COMPLETE_PENDING(tag)                  \
  "  s_waitcnt vmcnt(0)\n"                              \
  "  s_cmp_eq_u32 s67, 1\n"                             \
  "  s_cbranch_scc0 L_vcpb_" tag "%=\n"                 \
  EXON_FAR_COMPLETE("%[vfa]")                           \
  "  s_branch L_vcpd_" tag "%=\n"                       \
  "L_vcpb_" tag "%=:\n"                                 \
  EXON_FAR_COMPLETE("%[vfb]")                           \
  "L_vcpd_" tag "%=:\n"                                 \
  "  s_mov_b32 s67, 0\n"
#define EXON_LOOKUP_LIT                                 \
  "  v_lshlrev_b32 %[vt], 2, v48\n"                     \
  "  v_and_b32 %[vt], %[lutmask], %[vt]\n"              \
  "  ds_read_b32 v51, %[vt] offset:%[lutoff]\n"
  asm volatile(
      "  v_mov_b32 v48, s50\n"
      "  v_mov_b32 v49, s51\n"
      "  v_mov_b32 v50, s52\n"
      "  s_mov_b32 s67, 0\n"                    // far copies in flight: 0 none, 1 one in vfa, 2 one in vfb
      EXON_REFILL_V("e")
      EXON_LOOKUP_LIT
      "  s_and_b32 s57, s54, %[ringmask]\n"
      "  v_mov_b32 v37, s57\n"
      // ---- invariant at the loop head: the lookup of the current symbol is in flight into v51, v37 = pos & (RING - 1)
      "L_vsym_loop%=:\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cmp_eq_u32_sdwa vcc, v51, 1 src0_sel:BYTE_1 src1_sel:DWORD\n"  // bits 8-15 == 1: a literal (E_LIT alone)
      "  s_cbranch_vccz L_vsym_match%=\n"
      "  v_and_b32 v55, 15, v51\n"
      "  v_lshrrev_b64 v[48:49], v55, v[48:49]\n"
      "  ds_write_b8_d16_hi v37, v51\n"
      EXON_LOOKUP_LIT                          // the next symbol; everything below runs under its latency
      "  v_sub_u32 v50, v50, v55\n"
      "  s_add_i32 s54, s54, 1\n"
      "  s_and_b32 s57, s54, %[ringmask]\n"
      "  v_mov_b32 v37, s57\n"
      "  s_and_b32 s57, s54, 0xff\n"
      "  s_cbranch_scc0 L_vsym_row%=\n"
      // a second symbol without a look at the bit count: the loop head had more than 32 bits, a literal took at most 9, and 24 are
      // enough for another literal or for a length code with its extra bits (the match path refills behind the length itself)
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cmp_eq_u32_sdwa vcc, v51, 1 src0_sel:BYTE_1 src1_sel:DWORD\n"
      "  s_cbranch_vccz L_vsym_match%=\n"
      "  v_and_b32 v55, 15, v51\n"
      "  v_lshrrev_b64 v[48:49], v55, v[48:49]\n"
      "  ds_write_b8_d16_hi v37, v51\n"
      EXON_LOOKUP_LIT
      "  v_sub_u32 v50, v50, v55\n"
      "  s_add_i32 s54, s54, 1\n"
      "  s_and_b32 s57, s54, %[ringmask]\n"
      "  v_mov_b32 v37, s57\n"
      "  v_cmp_lt_i32 vcc, 32, v50\n"
      "  s_cbranch_vccz L_vrefill_l%=\n"
      "L_vback_l%=:\n"
      "  s_and_b32 s57, s54, 0xff\n"
      "  s_cbranch_scc1 L_vsym_loop%=\n"
      "  s_branch L_vsym_row%=\n"
      // ---- not a literal: a length code in the table (length field 1..15, neither end of block nor invalid)?
      "L_vsym_match%=:\n"
      "  v_cmp_eq_u32_sdwa vcc, v51, 8 src0_sel:BYTE_1 src1_sel:DWORD\n"  // E_FAST alone (long codes are 0 entries)
      "  s_cbranch_vccz L_vsym_exit0%=\n"
      "  v_and_b32 v55, 15, v51\n"             // code length
      "  v_bfe_u32 v34, v51, 4, 4\n"           // extra bits
      "  v_add_u32 v38, v55, v34\n"
      "  v_bfe_u32 v35, v48, v55, v34\n"       // their value (0 bits: 0)
      "  v_lshrrev_b64 v[48:49], v38, v[48:49]\n"
      "  v_lshlrev_b32 %[vt], 2, v48\n"        // the distance lookup (8 valid bits are there; the refill comes under it)
      "  v_and_b32 %[vt], %[dmask], %[vt]\n"
      "  ds_read_b32 v52, %[vt] offset:%[dlut]\n"
      "  v_lshrrev_b32 v53, 16, v51\n"
      "  v_add_u32 v53, v53, v35\n"            // length
      "  v_sub_u32 v50, v50, v38\n"
      EXON_REFILL_V("m")
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cmp_eq_u32_sdwa vcc, v52, 8 src0_sel:BYTE_1 src1_sel:DWORD\n"  // a usable distance code (not long, not 30 / 31)
      "  s_cbranch_vccz L_vsym_exit3%=\n"
      "  v_and_b32 v55, 15, v52\n"
      "  v_bfe_u32 v34, v52, 4, 4\n"
      "  v_add_u32 v38, v55, v34\n"
      "  v_bfe_u32 v35, v48, v55, v34\n"
      "  v_lshrrev_b64 v[48:49], v38, v[48:49]\n"
      EXON_LOOKUP_LIT                          // the symbol behind the match, under the copy
      "  v_lshrrev_b32 v54, 16, v52\n"
      "  v_add_u32 v54, v54, v35\n"            // distance
      "  v_sub_u32 v50, v50, v38\n"
      // ---- the copies the loop does itself (as in symbol_run): d <= history, len <= 64, and either len <= d <= NEAR or d > NEAR
      "  s_sub_i32 s57, s54, s66\n"
      "  v_cmp_lt_u32 vcc, s57, v54\n"
      "  s_cbranch_vccnz L_vsym_exit2%=\n"
      "  v_cmp_lt_u32 vcc, 64, v53\n"
      "  s_cbranch_vccnz L_vsym_exit2%=\n"
      "  v_readfirstlane_b32 s62, v53\n"
      "  v_sub_u32 v39, s54, v54\n"                  // first source byte
      "  v_cmp_lt_u32 vcc, %[near], v54\n"
      "  s_cbranch_vccnz L_vsym_far%=\n"
      "  v_cmp_gt_u32 vcc, v53, v54\n"
      "  s_cbranch_vccnz L_vsym_exit2%=\n"           // overlapping run
      // near: ring -> ring.  A far copy still in flight must land first if this source reaches into its bytes
      // (source end > its first byte; it ends at or below pos, where every source starts below)
      "  s_cmp_eq_u32 s67, 0\n"
      "  s_cbranch_scc1 L_vsym_near%=\n"
      "  v_add_u32 v35, v39, v53\n"
      "  v_cmp_lt_u32 vcc, s70, v35\n"
      "  s_cbranch_vccz L_vsym_near%=\n"
      EXON_FAR_COMPLETE_PENDING("n")
      "L_vsym_near%=:\n"
      "  v_cmp_gt_u32_e64 s[60:61], v53, %[lane]\n"  // lanes below len
      "  s_and_saveexec_b64 s[64:65], s[60:61]\n"
      "  v_add_u32 %[vt], v39, %[lane]\n"
      "  v_and_b32 %[vt], %[ringmask], %[vt]\n"
      "  ds_read_u8 %[ve], %[vt]\n"
      "  v_add_u32 %[vt], s54, %[lane]\n"
      "  v_and_b32 %[vt], %[ringmask], %[vt]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  ds_write_b8 %[vt], %[ve]\n"
      "  s_mov_b64 exec, s[64:65]\n"
      "  s_branch L_vsym_adv%=\n"
      // far: the source is in HBM already (d > NEAR: below `drained`).  The load is ISSUED here and its bytes are put into
      // the ring later -- when the next far copy has issued its own load (two data registers take turns), when a near copy
      // reads them, when the block is left (row drains, slow paths) -- so the round trip to L2 / HBM runs under the decoding
      // of the symbols behind it.  Nothing reads those ring bytes earlier: literals and copies only write at `pos` and above.
      "L_vsym_far%=:\n"
#ifdef EXON_INFLATE_FAR_NOWAIT  // timing experiment only (wrong bytes): what the loop would cost if far copies were free
      "  s_branch L_vsym_adv%=\n"
#endif
      "  v_cmp_gt_u32_e64 s[60:61], v53, %[lane]\n"
      "  s_and_saveexec_b64 s[64:65], s[60:61]\n"
      "  v_add_u32 %[vt], v39, %[lane]\n"
      "  s_cmp_eq_u32 s67, 1\n"
      "  s_cbranch_scc1 L_vsym_far_b%=\n"
      "  global_load_ubyte %[vfa], %[vt], s[68:69]\n"
      "  s_mov_b64 exec, s[64:65]\n"
      "  s_cmp_eq_u32 s67, 0\n"
      "  s_cbranch_scc1 L_vsym_far_a1%=\n"
      "  s_waitcnt vmcnt(1)\n"                       // the older one (in vfb) has landed; loads return in order
      EXON_FAR_COMPLETE("%[vfb]")
      "L_vsym_far_a1%=:\n"
      "  s_mov_b32 s67, 1\n"
      "  s_branch L_vsym_far_rec%=\n"
      "L_vsym_far_b%=:\n"
      "  global_load_ubyte %[vfb], %[vt], s[68:69]\n"
      "  s_mov_b64 exec, s[64:65]\n"
      "  s_waitcnt vmcnt(1)\n"
      EXON_FAR_COMPLETE("%[vfa]")
      "  s_mov_b32 s67, 2\n"
      "L_vsym_far_rec%=:\n"
      "  s_mov_b32 s70, s54\n"
      "  s_mov_b32 s71, s62\n"
      "L_vsym_adv%=:\n"
      "  s_add_i32 s57, s54, s62\n"
      "  s_xor_b32 s64, s57, s54\n"
      "  s_mov_b32 s54, s57\n"
      "  s_and_b32 s57, s57, %[ringmask]\n"
      "  v_mov_b32 v37, s57\n"
      "  v_cmp_lt_i32 vcc, 32, v50\n"
      "  s_cbranch_vccz L_vrefill_a%=\n"
      "L_vback_a%=:\n"
      "  s_lshr_b32 s64, s64, 8\n"             // SCC = a 256-byte row boundary was crossed
      "  s_cbranch_scc0 L_vsym_loop%=\n"
      "L_vsym_row%=:\n"
      "  s_mov_b32 s56, 1\n"
      "  s_branch L_vsym_out%=\n"
      EXON_REFILL_OUT("l")
      EXON_REFILL_OUT("a")
      "L_vsym_exit0%=:\n"
      "  s_mov_b32 s56, 0\n"
      "  s_branch L_vsym_out%=\n"
      "L_vsym_exit2%=:\n"
      "  s_mov_b32 s56, 2\n"
      "  s_branch L_vsym_out%=\n"
      "L_vsym_exit3%=:\n"
      "  s_mov_b32 s56, 3\n"
      "L_vsym_out%=:\n"
      "  s_cmp_eq_u32 s67, 0\n"
      "  s_cbranch_scc1 L_vsym_fin%=\n"
      EXON_FAR_COMPLETE_PENDING("x")
      "L_vsym_fin%=:\n"
      "  s_nop 0\n"
      "  v_readfirstlane_b32 s50, v48\n"
      "  v_readfirstlane_b32 s51, v49\n"
      "  v_readfirstlane_b32 s52, v50\n"
      "  v_readfirstlane_b32 s55, v51\n"
      "  v_readfirstlane_b32 s62, v53\n"
      "  v_readfirstlane_b32 s63, v54\n"
      "  v_mov_b32 %[vpos], s54\n"
      "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      "  s_nop 1\n"  // the compiler does not know a VALU instruction wrote s50-s63: keep its next VALU read two states away
      : [buf] "+{s[50:51]}"(buf), [cnt] "+{s52}"(cnt), [widx] "+{s53}"(widx), [pos] "+{s54}"(pos), [e] "={s55}"(e), [why] "={s56}"(why),
        [len] "={s62}"(len), [d] "={s63}"(d), [cur] "+v"(br.cur), [vpos] "+v"(vpos), [vt] "=&v"(vt), [ve] "=&v"(ve), [vfa] "=&v"(vfa), [vfb] "=&v"(vfb)
      : [base] "{s[58:59]}"(br.base), [begin] "{s66}"(begin), [out] "{s[68:69]}"(out), [lane] "v"(lane), [lane4] "v"(lane4), [lutmask] "i"(((1 << LIT_BITS) - 1) << 2),
        [dmask] "i"(((1 << DIST_BITS) - 1) << 2), [ringmask] "i"(RING - 1), [near] "i"(RING - 258),
        [lutoff] "i"(RING + (int)__builtin_offsetof(WaveLds, lit_lut)), [dlut] "i"(RING + (int)__builtin_offsetof(WaveLds, dist_lut))
      : "s57", "s60", "s61", "s64", "s65", "s67", "s70", "s71", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",
        "vcc", "scc", "memory");
#undef EXON_REFILL_V
#undef EXON_REFILL_OUT
#undef EXON_REFILL_BODY
#undef EXON_LOOKUP_LIT
#undef EXON_FAR_COMPLETE
#undef EXON_FAR_COMPLETE_PENDING
  br.buf = ((uint64_t)uniu((uint32_t)(buf >> 32)) << 32) | uniu((uint32_t)buf);
  br.cnt = uni(cnt);
  br.widx = uniu(widx);
  pos = uniu(pos);
  e = uniu(e);
  len = uniu(len);
  d = uniu(d);
  return uniu(why);
}


// ================================================================================================================
// The WIDE symbol loop (round 5): 64 lanes = 64 consecutive BIT OFFSETS of the stream.
// The loops above decode one symbol per step with one useful lane: every step pays an LDS round trip (table lookup) on the
// wave's own dependency chain, ~300 clocks per symbol with 8 waves sharing a SIMD.  Here a ROUND takes 64 bits of the stream
// at once: lane l decodes the complete symbol that WOULD start at bit l (literal, or length + extra bits + distance + extra
// bits: two table lookups for all 64 offsets together), so after two LDS round trips every possible symbol start of the
// round is decoded.  The true chain 0 -> next(0) -> next(next(0)) ... is then followed with one v_readlane per symbol (no
// memory on the chain), and while it is walked every OUTPUT BYTE of the round finds its owner: byte lane b keeps the record of
// the last symbol that starts at or before output byte b.  The round's output (<= 64 bytes) is then produced in ONE step for
// all its symbols -- literal lanes store their byte, near-match lanes copy ring -> ring, far-match lanes load from the output
// already in HBM (the load is deferred by one round) -- instead of one masked copy per match.
//   * a match that does not fit the round's 64 output bytes is carried into the next round as (rest, distance): rounds split
//     matches freely, a copy is byte-sequential;
//   * a byte whose source lies in its own round (distance <= lane: a match right behind the literals it repeats, runs): the round
//     is cut in front of the symbol that owns the first such byte (what precedes it has its sources in earlier rounds, and so has
//     that symbol as the first of the next round); only a match that overlaps ITSELF (distance < length) takes the slow path of
//     the round, where the bytes are produced in dependency order, as many per step as are ready;
//   * a symbol the tables do not resolve ends the chain in front of it (its lane's record is 0, the chain goes to 64 + lane).  A
//     LITERAL with a code longer than the table is decoded right there (all candidate lengths at once, one per lane) and the walk
//     goes on; end of block, a long length / distance code or an invalid code leave the loop: the round consumes what precedes
//     it and the caller's slow path takes the symbol, as for the loops above.
// Same contract as symbol_run: why 0 / 1 / 2 as there (e / len / d), 4 = a distance reaches before the member's output.
// Window: `cur` = dwords [wb, wb + 64) of the compressed data, one per lane, wb a multiple of 32; a round needs the dwords
// k .. k + 4 with k = bp >> 5 < 32 + 4; when bp passes 1024 the window moves by 32 dwords (upper half of cur + lower half of
// nxt, nxt = dwords [wb + 64, wb + 128) loaded long before).
// tools/deflate_stats.cpp has what a round sees: VCF text 5.4 symbols (3.4 matches) and 31 output bytes per round, BAM 6.7
// (2.3) and 27, FASTQ 8.7 (1.4) and 16; 0.5 % / 8 % / 0.7 % of the rounds take the slow path.
// ================================================================================================================

#ifndef EXON_WIDE_ASM
#define EXON_WIDE_ASM 1
#endif
// The rounds of wide_run that need nothing special, hand-written.  Why by hand: the compiler's version of the round spends 78
// scalar + 83 vector instructions (VCF text; rocprofv3 PMC, profiles/r5_inflate_wide_v0_pmc.txt) around divergent branches; the first
// hand-written version moved everything wave-uniform onto the vector unit (44 scalar + 86 vector per round) -- and PMC then showed the
// VECTOR unit 95 % busy and the scalar one 45 % (profiles/r5_inflate_wide_v3_pmc.txt: a wave64 integer instruction holds its SIMD for 4
// clocks, so 8 waves x 86 instructions x 4 clocks were the round).  Now the two units carry about the same load: the chain walk is
// scalar (position, byte count, loop test) with ONE v_readlane and one masked v_mov per symbol -- EXEC shrinks to the byte lanes at
// or beyond the current symbol, so no compare-and-select per symbol -- masks are taken by v_cmpx straight into EXEC, and per-lane
// arithmetic uses the fused forms (v_and_or, v_xad, v_lshl_or, v_add3, v_alignbit instead of a 64-bit shift, SDWA byte / word selects).
// Order of a round: window (its three ds_bpermute were issued in the previous round, under that round's output) -> two table
// lookups -> the chain -> the next round's window is requested -> the far copy deferred in the previous round lands (its load had
// this round's decode and walk to come back) -> a row of the ring completed by the PREVIOUS round is drained (one round late, so
// that nothing waits for a load just issued) -> literals, near copies, this round's far loads -> advance.
// Wait states the assembler does not insert inside an asm block (gfx940 family, LLVM's GCNHazardRecognizer): a VALU-written SGPR
// or VCC needs 2 states before a VALU reads it as an operand or a v_cndmask mask, 4 before v_readlane uses it as the lane select; a
// VALU-written VGPR 1 before v_readlane / v_readfirstlane reads it; a VALU-written EXEC 4 before v_readlane.  SALU reads of
// VALU-written SGPRs and s_cbranch_vcc* are interlocked.
// Registers: s40-s71 and v33-v63 are this block's; the interface values travel in the operands.
// Comes back with 0 / 4 as wide_run's `why`; 1: a completed row needs the caller (something to report, or not one whole aligned
// row) BEFORE the next round; 5: a round the caller finishes from the chain walk on (rec / nextp / ev are that round's, nothing
// of it is consumed).
template <int RING>
__device__ __forceinline__ uint32_t wide_rounds_asm(const __attribute__((address_space(1))) uint32_t* base, const __attribute__((address_space(1))) uint8_t* out,
                                                    uint32_t& bp, uint32_t& wb, uint32_t& pos, uint32_t& drained, uint32_t& carry_len, uint32_t& carry_rec,
                                                    uint32_t begin, uint32_t end, uint32_t limit, uint32_t lane, uint32_t& cur, uint32_t& nxt, uint32_t& fdata,
                                                    uint32_t& faddr, uint32_t& rec, uint32_t& nextp, uint32_t& ev, uint32_t& e_out) {
  uint32_t code;
  asm volatile(
      "  v_add_u32 v60, 64, %[lane]\n"                   // 64 + lane: where the chain goes from a symbol the tables do not resolve
      "  v_mov_b32 v50, 0xff0000\n"                      // (constants of the literal's record: one v_and_or instead of two instructions)
      "  v_mov_b32 v54, 0x80000001\n"
      // lane q (1..15): first canonical code / number of codes / symbols of shorter lengths, of literal/length codes q bits long;
      // lane 16 + q: the same of distance codes q bits long (first | offs | dfirst | doffs and lit_count | dist_count are adjacent)
      "  v_and_b32 v48, 31, %[lane]\n"
      "  v_lshlrev_b32 v49, 1, v48\n"                    // counts: 2 bytes x (lane & 31)
      "  v_and_b32 v47, 16, %[lane]\n"
      "  v_lshl_add_u32 v48, v47, 1, v49\n"              // first / offs: the distance code's sit 64 bytes further on (32 + 2 x 16)
      "  ds_read_u16 v61, v48 offset:%[firstoff]\n"
      "  ds_read_u16 v62, v49 offset:%[countoff]\n"
      "  ds_read_u16 v63, v48 offset:%[offsoff]\n"
      "  v_and_b32 v47, 15, %[lane]\n"
      "  v_cmp_gt_u32 vcc, 32, %[lane]\n"
      "  v_cmp_ne_u32 s[58:59], 0, v47\n"                // (no code of length 0; the distance code's slot for it is the far copy's dummy byte)
      "  s_and_b64 vcc, vcc, s[58:59]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cndmask_b32 v62, 0, v62, vcc\n"
      "  s_cmpk_lt_u32 s44, 0x400\n"
      "  s_cbranch_scc1 L_wr_first%=\n"
      "  s_mov_b32 s71, 0\n"                              // (return address of the window switch: 0 = the first window)
      "  s_branch L_wr_switch%=\n"
      "L_wr_first%=:\n"
      "  v_add_u32 v33, s44, %[lane]\n"
      "  v_lshrrev_b32 v34, 3, v33\n"
      "  ds_bpermute_b32 v35, v34, %[cur]\n"
      "  ds_bpermute_b32 v36, v34, %[cur] offset:4\n"
      "  ds_bpermute_b32 v37, v34, %[cur] offset:8\n"
      // ---- a round: every lane's 64 bits of the stream from bit bp + lane (v33) are on their way.  (The round's first instruction
      // sits on a 64-byte line: the chain walk's loop, 268 bytes further on, then lies inside one line -- the kernel is 1.2 % faster on
      // FASTQ and VCF members than with the loop 4 bytes further along, profiles/r5_inflate_code_alignment.log.  The padding is only
      // run through when the rounds are entered.)
#ifndef EXON_WIDE_NO_ALIGN
      "  .p2align 6\n"
#endif
      "L_wr_round%=:\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_alignbit_b32 v38, v36, v35, v33\n"
      "  v_alignbit_b32 v39, v37, v36, v33\n"
      "  v_and_b32 v34, 0x1ff, v38\n"
      "  v_lshlrev_b32 v34, 2, v34\n"
      "  ds_read_b32 %[ev], v34 offset:%[lutoff]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      // ---- the symbol that would start there: literal, or length + extra bits + distance + extra bits
      "  v_and_b32 v42, 15, %[ev]\n"                      // code length
      "  v_bfe_u32 v43, %[ev], 4, 4\n"                    // extra bits
      "  v_add_u32 v44, v42, v43\n"
      "  v_alignbit_b32 v46, v39, v38, v44\n"            // the 32 bits behind the length code and its extra bits (at most 20)
      "  v_lshlrev_b32 v34, 2, v46\n"
      "  v_and_b32 v34, 0x3fc, v34\n"
      "  ds_read_b32 v41, v34 offset:%[dlut]\n"
      "  v_bfe_u32 v45, v38, v42, v43\n"                  // value of the extra bits
      "  v_cmp_eq_u32_sdwa s[58:59], %[ev], 1 src0_sel:BYTE_1 src1_sel:DWORD\n"  // E_LIT alone: a literal
      "  v_cmp_eq_u32_sdwa s[60:61], %[ev], 8 src0_sel:BYTE_1 src1_sel:DWORD\n"  // E_FAST alone: a length the tables resolve
      "  v_add_u32_sdwa v48, v45, %[ev] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"  // length
      "  v_and_or_b32 v49, %[ev], v50, v54\n"            // a literal's record
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cmp_eq_u32_sdwa vcc, v41, 8 src0_sel:BYTE_1 src1_sel:DWORD\n"         // ... and a distance they resolve
      "  s_and_b64 s[60:61], s[60:61], vcc\n"
      "  v_and_b32 v35, 15, v41\n"
      "  v_bfe_u32 v36, v41, 4, 4\n"
      "  v_bfe_u32 v37, v46, v35, v36\n"
      "  v_add_u32_sdwa v37, v37, v41 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"    // distance
      "  v_add_u32 v37, -1, v37\n"
      "  v_lshl_or_b32 %[rec], v37, 16, v48\n"           // a match's record
      "  v_add3_u32 %[nextp], v44, v35, v36\n"           // its bits
      "  s_or_b64 s[60:61], s[60:61], s[58:59]\n"        // resolved at all
      "  v_cndmask_b32 %[nextp], %[nextp], v42, s[58:59]\n"
      "  v_cndmask_b32 %[rec], %[rec], v49, s[58:59]\n"
      "  v_add_u32 %[nextp], %[nextp], %[lane]\n"
      "  v_cndmask_b32 %[nextp], v60, %[nextp], s[60:61]\n"   // not resolved: the chain goes to 64 + lane ...
      "  v_cndmask_b32 %[rec], 0, %[rec], s[60:61]\n"         // ... with a record of no output bytes
      "  v_lshl_or_b32 v51, %[nextp], 9, %[rec]\n"            // the walk's word: record + next position in bits 9-15
      // ---- the chain.  s65 = output bytes so far; v53 = per byte lane the record of the last symbol that starts at or before it:
      // EXEC shrinks to the byte lanes at or beyond the current symbol's first byte (it only ever shrinks: the bytes below keep
      // their records), so a symbol's record is ONE masked v_mov.  The scalar unit has the headroom (PMC: the vector unit is 95 %
      // busy, the scalar one 45 %): position, byte count and loop test are scalar, 3 vector + 5 scalar instructions per symbol.
      "  v_mov_b32 v53, s52\n"
      "  s_mov_b32 s65, s51\n"
      "  s_mov_b32 s53, 0\n"
      "  s_mov_b32 s54, s52\n"
      "  s_mov_b32 s57, 1\n"                             // (last symbol's bytes: not 0 = no unresolved symbol met)
      "  s_cmp_ge_u32 s51, 64\n"
      "  s_cbranch_scc1 L_wr_walked%=\n"
      "L_wr_walk%=:\n"
      "  v_readlane_b32 s54, v51, s53\n"
      "  v_cmpx_le_u32 vcc, s65, %[lane]\n"
      "  s_bfe_u32 s53, s54, 0x70009\n"                  // where the next symbol starts
      "  s_and_b32 s57, s54, 0x1ff\n"                    // this symbol's bytes
      "  v_mov_b32 v53, s54\n"
      "  s_add_u32 s65, s65, s57\n"
      "  s_cmp_lt_u32 s53, 64\n"
      "  s_cbranch_scc1 L_wr_walk%=\n"
      "  s_cmp_eq_u32 s57, 0\n"                          // the last symbol gave no bytes: one the tables do not resolve
      "  s_cbranch_scc1 L_wr_long%=\n"
      "L_wr_postwalk%=:\n"
      "  s_mov_b64 exec, -1\n"
      // a symbol that STARTS beyond the 64 output bytes of the round was walked over (a long match with more symbols behind it in
      // the same 64 bits: 2 % of the rounds of BAM payloads): the byte lanes are right -- such a symbol found EXEC empty -- but
      // the round must end in front of the first of them.  A second, scalar-only walk finds it.
      "  s_sub_u32 s66, s65, s57\n"
      "  s_cmp_ge_u32 s66, 64\n"
      "  s_cbranch_scc1 L_wr_walk64_start%=\n"
      "L_wr_walked%=:\n"
      "  s_mov_b32 s70, s53\n"                           // bits consumed ...
      "  s_mov_b32 s67, s57\n"                           // (0: the round ends in front of an unresolved symbol; kept until the advance)
      "  s_cmp_eq_u32 s57, 0\n"
      "  s_cbranch_scc0 L_wr_consumed%=\n"
      "  s_sub_u32 s70, s53, 64\n"                       // ... up to the unresolved symbol (it sits at 64 + its lane)
      "L_wr_consumed%=:\n"
      "  s_min_u32 s64, s65, 64\n"                       // bytes of the round
      // ---- the far copy deferred in the previous round lands (lanes without one write a byte nobody reads)
      "  s_waitcnt vmcnt(0)\n"
      "  ds_write_b8 %[faddr], %[fdata]\n"
      "  v_mov_b32 %[faddr], %[dummy]\n"
      // ---- the next round's window is requested now: it arrives under this round's output
      "L_wr_rewin%=:\n"
      "  s_add_u32 s44, s44, s70\n"
      "  s_cmpk_lt_u32 s44, 0x400\n"
      "  s_cbranch_scc0 L_wr_switch1%=\n"
      "L_wr_win%=:\n"
      "  v_add_u32 v33, s44, %[lane]\n"
      "  v_lshrrev_b32 v34, 3, v33\n"
      "  ds_bpermute_b32 v35, v34, %[cur]\n"
      "  ds_bpermute_b32 v36, v34, %[cur] offset:4\n"
      "  ds_bpermute_b32 v37, v34, %[cur] offset:8\n"
      // ---- a row of the ring completed before this round goes to HBM
      "  s_and_b32 s66, s46, 0xffffff00\n"
      "  s_cmp_gt_u32 s66, s50\n"
      "  s_cbranch_scc1 L_wr_drain%=\n"
      "L_wr_drained%=:\n"
      // ---- the output bytes
      "  v_ashrrev_i32 v55, 16, v53\n"                    // distance - 1 (a literal: negative)
      "  v_add_u32 v56, s46, %[lane]\n"                   // where the byte goes
      "  v_xad_u32 v58, v55, -1, v56\n"                  // its source: where it goes - distance = ~(distance - 1) + where it goes
      "  v_and_b32 v57, %[ringmask], v56\n"
      "  v_cmpx_gt_u32 vcc, s64, %[lane]\n"               // EXEC = the round's byte lanes
      "  s_sub_u32 s66, s46, s47\n"                      // (no distance reaches back 32 KiB: nothing to check from there on)
      "  s_cmpk_ge_u32 s66, 0x8000\n"
      "  s_cbranch_scc1 L_wr_nobad%=\n"
      "  v_subrev_u32 v48, s47, v56\n"
      "  v_cmp_ge_i32 vcc, v55, v48\n"                    // a distance that reaches before the member's output
      "  s_cbranch_vccnz L_wr_bad%=\n"
      "L_wr_nobad%=:\n"
      "  v_cmp_lt_u32 vcc, v55, %[lane]\n"                // a source inside the round's own output
      "  s_cbranch_vccnz L_wr_dep%=\n"
      "L_wr_copy%=:\n"
      "  s_mov_b64 s[62:63], exec\n"
      "  v_cmpx_gt_i32 vcc, 0, v53\n"                     // literals
      "  ds_write_b8_d16_hi v57, v53\n"
      "  s_andn2_b64 exec, s[62:63], exec\n"              // matches
      "  s_cbranch_execz L_wr_nomatch%=\n"
      "  s_mov_b64 s[62:63], exec\n"
      "  v_cmpx_gt_u32 vcc, %[nearw], v55\n"              // near: ring -> ring
      "  s_cbranch_execz L_wr_nonear%=\n"
      "  v_and_b32 v48, %[ringmask], v58\n"
      "  ds_read_u8 v59, v48\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  ds_write_b8 v57, v59\n"
      "L_wr_nonear%=:\n"
      "  s_andn2_b64 exec, s[62:63], exec\n"              // far: the source is in HBM already; the load lands in the next round
      "  s_cbranch_execz L_wr_nomatch%=\n"
      "  global_load_ubyte %[fdata], v58, s[42:43]\n"
      "  v_mov_b32 %[faddr], v57\n"
      "L_wr_nomatch%=:\n"
      "  s_mov_b64 exec, -1\n"
      // ---- advance
      "  s_sub_u32 s51, s65, s64\n"                       // the rest of a match that did not fit
      "  s_mov_b32 s52, s54\n"
      "  s_add_u32 s46, s46, s64\n"
      "  s_cmp_eq_u32 s67, 0\n"
      "  s_cbranch_scc0 L_wr_round%=\n"
      "  s_nop 0\n"                                       // a symbol the tables do not resolve: its first-level entry for the caller
      "  v_readlane_b32 s56, %[ev], s70\n"               // (s70 = bits consumed = its lane)
      "  s_mov_b32 s55, 0\n"
      "  s_branch L_wr_out%=\n"
      // ---- rows completed by earlier rounds: drained here when it is exactly one aligned row and there is nothing to report
      "L_wr_drain%=:\n"
      "  s_cmp_gt_u32 s46, s48\n"
      "  s_cbranch_scc1 L_wr_exit1%=\n"
      "  s_lshr_b32 s57, s44, 5\n"
      "  s_add_u32 s57, s57, s45\n"
      "  s_cmp_gt_u32 s57, s49\n"
      "  s_cbranch_scc1 L_wr_exit1%=\n"
      "  s_and_b32 s57, s50, 0xff\n"
      "  s_cbranch_scc1 L_wr_exit1%=\n"
      "  s_sub_u32 s57, s66, s50\n"
      "  s_cmp_eq_u32 s57, 0x100\n"
      "  s_cbranch_scc0 L_wr_exit1%=\n"
      "  s_and_b32 s57, s50, %[ringmask]\n"
      "  v_lshl_add_u32 v48, %[lane], 2, s57\n"
      "  ds_read_b32 v49, v48\n"
      "  v_lshl_add_u32 v48, %[lane], 2, s50\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  global_store_dword v48, v49, s[42:43]\n"
      "  s_mov_b32 s50, s66\n"
      "  s_branch L_wr_drained%=\n"
      "L_wr_exit1%=:\n"                                   // (nothing of this round is consumed: the position goes back)
      "  s_sub_u32 s44, s44, s70\n"
      "  s_mov_b32 s55, 1\n"
      "  s_branch L_wr_out%=\n"
      // ---- the window moves by 32 dwords: upper half of cur + lower half of nxt
      "L_wr_switch1%=:\n"
      "  s_mov_b32 s71, 1\n"
      "L_wr_switch%=:\n"
      "  v_xor_b32 v34, 32, %[lane]\n"
      "  v_lshlrev_b32 v34, 2, v34\n"
      "  s_waitcnt vmcnt(0)\n"
      "  ds_bpermute_b32 v35, v34, %[cur]\n"
      "  ds_bpermute_b32 v36, v34, %[nxt]\n"
      "  v_cmp_gt_u32 vcc, 32, %[lane]\n"
      "  s_add_u32 s45, s45, 32\n"
      "  s_sub_u32 s44, s44, 0x400\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_cndmask_b32 %[cur], v36, v35, vcc\n"
      "  v_add_u32 v34, s45, %[lane]\n"
      "  v_lshlrev_b32 v34, 2, v34\n"
      // (no far copy is in flight here: inside the loop it has just landed, at the entry the caller's loads are waited for)
      "  global_load_dword %[nxt], v34, s[40:41] offset:256\n"
      "  s_cmp_eq_u32 s71, 0\n"
      "  s_cbranch_scc1 L_wr_first%=\n"
      "  s_branch L_wr_win%=\n"
      // ---- the chain stopped at a symbol the first-level table does not resolve.  A literal with a code longer than the table
      // (7 % of the symbols of BAM payloads) is decoded here and the walk goes on: all candidate lengths at once -- lane q takes the
      // first q bits as a code of that length and tests it against the codes of its length; codes are prefix-free, so at most one
      // length matches (decode_long's method).  Anything else (end of block, a long length code, no such code) stays the caller's.
      "L_wr_long%=:\n"
      "  s_sub_u32 s67, s53, 64\n"                       // its lane
      "  s_mov_b64 s[62:63], exec\n"
      "  s_mov_b64 exec, -1\n"
      "  v_and_b32 v49, 15, %[lane]\n"
      "  v_readlane_b32 s66, %[ev], s67\n"
      "  v_sub_u32 v49, 32, v49\n"
      "  s_cmp_eq_u32 s66, 0\n"
      "  s_cbranch_scc0 L_wr_longdist%=\n"
      "  v_readlane_b32 s66, v38, s67\n"
      "  s_nop 1\n"
      "  v_bfrev_b32 v48, s66\n"
      "  v_lshrrev_b32 v48, v49, v48\n"
      "  v_sub_u32 v48, v48, v61\n"
      "  v_cmp_lt_u32 vcc, v48, v62\n"
      "  v_add_u32 v48, v48, v63\n"
      "  s_and_b32 s66, vcc_lo, 0xfffe\n"               // (lanes 16-31 test the distance code's lengths)
      "  s_ff1_i32_b32 s66, s66\n"
      "  s_cmp_lt_i32 s66, 0\n"
      "  s_cbranch_scc1 L_wr_postwalk%=\n"
      "  v_readlane_b32 s71, v48, s66\n"
      "  s_lshl_b32 s71, s71, 1\n"
      "  v_mov_b32 v49, s71\n"
      "  ds_read_u16 v49, v49 offset:%[symoff]\n"
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_readfirstlane_b32 s71, v49\n"
      "  s_cmp_lt_u32 s71, 0x100\n"
      "  s_cbranch_scc0 L_wr_postwalk%=\n"
      "  s_lshl_b32 s71, s71, 16\n"
      "  s_or_b32 s54, s71, 0x80000001\n"                // the literal's record, and the walk's step for it
      "  s_mov_b64 exec, s[62:63]\n"
      "  v_cmpx_le_u32 vcc, s65, %[lane]\n"
      "  s_add_u32 s53, s67, s66\n"
      "  s_mov_b32 s57, 1\n"
      "  v_mov_b32 v53, s54\n"
      "  s_add_u32 s65, s65, 1\n"
      "  s_cmp_lt_u32 s53, 64\n"
      "  s_cbranch_scc1 L_wr_walk%=\n"
      "  s_branch L_wr_postwalk%=\n"
      // ---- ... or at a length the tables resolve whose DISTANCE code is longer than its table (1 % of the matches of FASTQ members,
      // the largest group of symbols that left the rounds): the same test of all candidate lengths in lanes 17-31, the symbol's
      // base and extra bits from its number, and the walk goes on with the match's record.
      "L_wr_longdist%=:\n"
      "  s_bfe_u32 s71, s66, 0x80008\n"
      "  s_cmp_eq_u32 s71, 8\n"                          // E_FAST alone
      "  s_cbranch_scc0 L_wr_postwalk%=\n"
      "  v_readlane_b32 s71, v41, s67\n"
      "  s_cmp_eq_u32 s71, 0\n"                          // (long codes are 0 entries)
      "  s_cbranch_scc0 L_wr_postwalk%=\n"
      "  v_readlane_b32 s71, v46, s67\n"                // the bits behind the length code and its extra bits
      "  s_nop 1\n"
      "  v_bfrev_b32 v48, s71\n"
      "  v_lshrrev_b32 v48, v49, v48\n"
      "  v_sub_u32 v48, v48, v61\n"
      "  v_cmp_lt_u32 vcc, v48, v62\n"
      "  v_add_u32 v48, v48, v63\n"
      "  s_lshr_b32 s58, vcc_lo, 16\n"
      "  s_ff1_i32_b32 s58, s58\n"                       // the code's length
      "  s_cmp_lt_i32 s58, 0\n"
      "  s_cbranch_scc1 L_wr_postwalk%=\n"
      "  s_add_u32 s59, s58, 16\n"
      "  v_readlane_b32 s60, v48, s59\n"
      "  s_lshl_b32 s60, s60, 1\n"
      "  v_mov_b32 v48, s60\n"
      "  ds_read_u16 v48, v48 offset:%[dsymoff]\n"
      "  v_readlane_b32 s69, v45, s67\n"                // value of the length's extra bits
      "  v_readlane_b32 s59, v44, s67\n"                // bits of the length code and its extra bits
      "  s_waitcnt lgkmcnt(0)\n"
      "  v_readfirstlane_b32 s60, v48\n"                // the distance symbol
      "  s_cmp_gt_u32 s60, 29\n"
      "  s_cbranch_scc1 L_wr_postwalk%=\n"
      "  s_lshr_b32 s61, s60, 1\n"
      "  s_sub_u32 s61, s61, 1\n"
      "  s_max_i32 s61, s61, 0\n"                        // its extra bits
      "  s_and_b32 s68, s60, 1\n"
      "  s_or_b32 s68, s68, 2\n"
      "  s_lshl_b32 s68, s68, s61\n"                     // distance - 1 of the symbol's first distance (symbols 2 ..)
      "  s_cmp_lt_u32 s60, 2\n"
      "  s_cselect_b32 s68, s60, s68\n"                  // (symbols 0 and 1: distances 1 and 2)
      "  s_lshr_b32 s71, s71, s58\n"
      "  s_bfm_b32 s60, s61, 0\n"
      "  s_and_b32 s71, s71, s60\n"                      // value of the distance's extra bits
      "  s_add_u32 s68, s68, s71\n"                      // distance - 1
      "  s_lshr_b32 s66, s66, 16\n"
      "  s_add_u32 s57, s66, s69\n"                      // length
      "  s_add_u32 s59, s59, s58\n"
      "  s_add_u32 s59, s59, s61\n"
      "  s_add_u32 s53, s59, s67\n"                      // where the next symbol starts
      "  s_lshl_b32 s68, s68, 16\n"
      "  s_or_b32 s54, s68, s57\n"                       // the match's record
      "  s_mov_b64 exec, s[62:63]\n"
      "  v_cmpx_le_u32 vcc, s65, %[lane]\n"
      "  s_add_u32 s65, s65, s57\n"
      "  s_nop 0\n"
      "  v_mov_b32 v53, s54\n"
      "  s_cmp_lt_u32 s53, 64\n"
      "  s_cbranch_scc1 L_wr_walk%=\n"
      "  s_branch L_wr_postwalk%=\n"
      // ---- a byte of the round has its source inside the round (8 % of the rounds of BAM payloads: a match right behind the
      // literals it repeats, runs).  The round is cut in front of the symbol that owns the first such byte: what precedes it has
      // all its sources in earlier rounds, and that symbol, first of the next round, has too -- unless it overlaps ITSELF
      // (distance < length), which only the caller's byte-ordered slow path resolves.
      // First the round's FIRST symbol, carried in or not: a match that overlaps its own output (distance < length: runs, 1.5 % of the
      // matches of BAM payloads) repeats with period `distance`, so its bytes take their sources floor(lane / distance) periods
      // further back -- in front of the round -- and depend on nothing.  (They used to go to the caller's byte-ordered loop: an
      // iteration per byte of a run, ~20 rounds' worth of instructions for one round, 3.4 % of the rounds of BAM payloads.)
      "L_wr_dep%=:\n"
      "  s_mov_b64 s[58:59], vcc\n"                      // the dependent byte lanes
      "  s_mov_b32 s57, s52\n"
      "  s_min_u32 s62, s51, 64\n"                       // bytes of the first symbol in this round: of a match carried in ...
      "  s_cmp_eq_u32 s51, 0\n"
      "  s_cbranch_scc0 L_wr_fold_first%=\n"
      "  s_nop 3\n"
      "  v_readlane_b32 s57, v51, 0\n"                   // ... or of the symbol at the round's first bit
      "  s_and_b32 s62, s57, 0x1ff\n"
      "L_wr_fold_first%=:\n"
      "  s_ashr_i32 s66, s57, 16\n"                      // its distance - 1 (a literal: negative)
      "  s_cmp_lt_i32 s66, 0\n"
      "  s_cbranch_scc1 L_wr_dep_others%=\n"
      "  s_mov_b64 s[60:61], -1\n"
      "  s_cmp_ge_u32 s62, 64\n"
      "  s_cbranch_scc1 L_wr_fold_mask%=\n"
      "  s_bfm_b64 s[60:61], s62, 0\n"                   // the byte lanes of that symbol
      "L_wr_fold_mask%=:\n"
      "  s_and_b64 vcc, s[60:61], s[58:59]\n"
      "  s_cbranch_scc0 L_wr_dep_others%=\n"             // none of them dependent
      "  s_add_u32 s66, s66, 1\n"                        // the distance
      "  v_cvt_f32_u32 v48, %[lane]\n"
      "  v_cvt_f32_u32 v49, s66\n"
      "  v_rcp_f32 v49, v49\n"
      "  v_add_f32 v48, 0.5, v48\n"                      // (lane + 1/2) / distance is never within rounding of a whole number
      "  s_nop 0\n"
      "  v_mul_f32 v48, v48, v49\n"
      "  v_cvt_u32_f32 v48, v48\n"                       // whole periods between the byte and the symbol's start in this round
      "  v_mul_u32_u24 v48, s66, v48\n"
      "  v_cndmask_b32 v48, 0, v48, s[60:61]\n"
      "  v_sub_u32 v58, v58, v48\n"
      "  s_andn2_b64 vcc, s[58:59], s[60:61]\n"          // dependent bytes of the symbols behind it
      "  s_cbranch_scc0 L_wr_copy%=\n"
      "  s_branch L_wr_dep_cut%=\n"
      "L_wr_dep_others%=:\n"
      "  s_mov_b64 vcc, s[58:59]\n"
      // A source that is a LITERAL of this round is no dependence at all: the copy below stores the round's literals before it reads
      // the near sources (the usual shape in BAM payloads: fresh literals, then a match that repeats them).  Only a source that is
      // itself a match byte of the round has to wait for it.
      "L_wr_dep_cut%=:\n"
      "  v_cmp_gt_i32 s[60:61], 0, v53\n"                // the round's literal lanes ...
      "  s_andn2_b64 s[60:61], exec, s[60:61]\n"         // ... and its match lanes
      "  v_subrev_u32 v48, s46, v58\n"                   // the lane a dependent byte's source was written by
      "  v_lshrrev_b64 v[48:49], v48, s[60:61]\n"
      "  v_and_b32 v48, 1, v48\n"
      "  v_cmp_ne_u32 s[60:61], 0, v48\n"
      "  s_and_b64 vcc, vcc, s[60:61]\n"                 // dependent bytes whose source is a match byte of the round
      "  s_cbranch_scc0 L_wr_copy%=\n"
      "  s_ff1_i32_b64 s66, vcc\n"                       // the first byte whose source lies inside the round
      "  s_mov_b64 exec, -1\n"
      "  s_sub_u32 s44, s44, s70\n"                      // back to the round's first bit (a borrow: the window has moved, the caller rewinds)
      "  s_cbranch_scc1 L_wr_slow%=\n"
      "  s_cmp_gt_u32 s51, s66\n"                        // the byte belongs to the match carried in: an overlapping one
      "  s_cbranch_scc1 L_wr_slow%=\n"
      "  v_mov_b32 v53, s52\n"
      "  s_mov_b32 s65, s51\n"
      "  s_mov_b32 s53, 0\n"
      "  s_mov_b32 s54, s52\n"
      "L_wr_walk2%=:\n"
      "  v_readlane_b32 s71, v51, s53\n"
      "  s_and_b32 s62, s71, 0x1ff\n"                    // its bytes; none: an unresolved symbol, the round ends in front of it too
      "  s_cbranch_scc0 L_wr_walk2_done%=\n"
      "  s_add_u32 s57, s65, s62\n"
      "  s_cmp_gt_u32 s57, s66\n"                        // it owns the byte
      "  s_cbranch_scc1 L_wr_walk2_done%=\n"
      "  v_cmpx_le_u32 vcc, s65, %[lane]\n"
      "  s_mov_b32 s54, s71\n"
      "  s_bfe_u32 s53, s71, 0x70009\n"
      "  v_mov_b32 v53, s71\n"
      "  s_mov_b32 s65, s57\n"
      "  s_branch L_wr_walk2%=\n"
      "L_wr_walk2_done%=:\n"
      "  s_mov_b64 exec, -1\n"
      "  s_or_b32 s57, s53, s51\n"                       // no symbol taken and nothing carried in: no progress this way
      "  s_cbranch_scc0 L_wr_slow%=\n"
      "  s_mov_b32 s57, 1\n"
      "  s_mov_b32 s67, 1\n"
      "  s_mov_b32 s70, s53\n"
      "  s_mov_b32 s64, s65\n"
      "  s_branch L_wr_rewin%=\n"
      "L_wr_walk64_start%=:\n"
      "  s_mov_b32 s65, s51\n"
      "  s_mov_b32 s53, 0\n"
      "  s_mov_b32 s54, s52\n"
      "  s_nop 3\n"
      "L_wr_walk64%=:\n"
      "  v_readlane_b32 s71, v51, s53\n"
      "  s_cmp_ge_u32 s65, 64\n"                         // this one starts beyond the round's bytes
      "  s_cbranch_scc1 L_wr_walk64_done%=\n"
      "  s_and_b32 s57, s71, 0x1ff\n"
      "  s_cbranch_scc0 L_wr_slow%=\n"                   // (no bytes: a long literal the first walk decoded on the way -- the caller's)
      "  s_mov_b32 s54, s71\n"
      "  s_bfe_u32 s53, s71, 0x70009\n"
      "  s_add_u32 s65, s65, s57\n"
      "  s_branch L_wr_walk64%=\n"
      "L_wr_walk64_done%=:\n"
      "  s_mov_b32 s57, 1\n"                             // (the round does not end in front of an unresolved symbol)
      "  s_branch L_wr_walked%=\n"
      "L_wr_bad%=:\n"
      "  s_mov_b64 exec, -1\n"
      "  s_sub_u32 s44, s44, s70\n"
      "  s_mov_b32 s55, 4\n"
      "  s_branch L_wr_out%=\n"
      "L_wr_slow_unwind%=:\n"                             // (the position was moved for the next window: it goes back)
      "  s_mov_b64 exec, -1\n"
      "  s_sub_u32 s44, s44, s70\n"
      "L_wr_slow%=:\n"
      "  s_mov_b32 s55, 5\n"
      "L_wr_out%=:\n"
      "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      "  s_nop 1\n"
      : [code] "={s55}"(code), [e] "={s56}"(e_out), [bp] "+{s44}"(bp), [wb] "+{s45}"(wb), [pos] "+{s46}"(pos), [drained] "+{s50}"(drained),
        [cl] "+{s51}"(carry_len), [cr] "+{s52}"(carry_rec), [cur] "+v"(cur), [nxt] "+v"(nxt), [fdata] "+v"(fdata), [faddr] "+v"(faddr),
        [rec] "=&v"(rec), [nextp] "=&v"(nextp), [ev] "=&v"(ev)
      : [base] "{s[40:41]}"(base), [out] "{s[42:43]}"(out), [begin] "{s47}"(begin), [end] "{s48}"(end), [limit] "{s49}"(limit), [lane] "v"(lane),
        [ringmask] "i"(RING - 1), [nearw] "i"(RING - 258), [lutoff] "i"(RING + (int)__builtin_offsetof(WaveLds, lit_lut)),
        [dlut] "i"(RING + (int)__builtin_offsetof(WaveLds, dist_lut)), [firstoff] "i"(RING + (int)__builtin_offsetof(WaveLds, first)),
        [countoff] "i"(RING + (int)__builtin_offsetof(WaveLds, lit_count)), [offsoff] "i"(RING + (int)__builtin_offsetof(WaveLds, offs)),
        [symoff] "i"(RING + (int)__builtin_offsetof(WaveLds, lit_sym)), [dsymoff] "i"(RING + (int)__builtin_offsetof(WaveLds, dist_sym)), [dummy] "i"(RING + (int)__builtin_offsetof(WaveLds, dist_count))
      : "s53", "s54", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "v33", "v34", "v35", "v36", "v37",
        "v38", "v39", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",
        "v62", "v63", "vcc", "scc", "memory");
  bp = uniu(bp);
  wb = uniu(wb);
  pos = uniu(pos);
  drained = uniu(drained);
  carry_len = uniu(carry_len);
  carry_rec = uniu(carry_rec);
  e_out = uniu(e_out);
  return uniu(code);
}

#ifdef EXON_WIDE_STATS  // developer build (tools/build_variant.sh stats -DEXON_WIDE_STATS): where a member's clocks go
// [0] clocks inside wide_rounds_asm, [1] clocks of whole members, [2..7] returns of wide_rounds_asm by code 0..5, [8] members
__device__ unsigned long long g_wide_stats[16];
#endif
template <int RING>
__device__ __forceinline__ uint32_t wide_run(BitReader& br, Out& o, uint32_t& e_out, uint32_t& len_out, uint32_t& d_out) {
  constexpr uint32_t RM = RING - 1, NEARW = RING - 258;
  static_assert(NEARW >= 64 + 255 + 64, "a far source must lie below the drained rows");
  const uint32_t lane = lane_id();
  uint8_t* ring = wave_ring<RING>();
  const WaveLds* L = wave_lds<RING>();
  typedef const __attribute__((address_space(1))) uint32_t* gptr32;  // global_load, not flat_load (a flat load also counts as an LDS operation)
  typedef const __attribute__((address_space(1))) uint8_t* gptr8;
  gptr32 base = (gptr32)br.base;
  gptr8 gout = (gptr8)o.out;
  uint32_t drained = o.drained;
  auto bperm = [](uint32_t byte_addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v); };
  auto rdl = [](uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
  // the next symbol's bit: `back` dwords behind the reader's next dword (no absolute bit index: it would pass 2^32 at 512 MB)
  const uint32_t back = ((uint32_t)br.cnt + 31u) >> 5, dw0 = br.widx - back;
  uint32_t wb = dw0 & ~31u;
  uint32_t bp = (dw0 - wb) * 32u + (back * 32u - (uint32_t)br.cnt);
  uint32_t cur = base[wb + lane];
  uint32_t nxt = base[wb + 64u + lane];
  uint32_t pos = o.pos;
  uint32_t carry_len = 0, carry_rec = 0;
  // the deferred far copy: bytes in flight and their LDS addresses -- a lane without one writes a byte nobody reads (the
  // never-used "codes of length 0" slot of the distance code's per-length counts).  (Keeping the lanes as a scalar mask instead
  // -- one vector instruction less, three scalar ones more per round -- measured 3 % slower: both units are equally loaded.)
  constexpr uint32_t FAR_NONE = RING + (uint32_t)__builtin_offsetof(WaveLds, dist_count);
  uint32_t fdata = 0, faddr = FAR_NONE;
  uint32_t why;
  auto complete_far = [&]() {
    ring[faddr] = (uint8_t)fdata;  // (the ring sits at LDS address 0: `faddr` is an LDS address)
    faddr = FAR_NONE;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  for (;;) {
    uint32_t rec, nextp, ev;
#if EXON_WIDE_ASM
    // ---- the rounds that need nothing special, hand-written (wide_rounds_asm): comes back with
    //   0 / 1 / 4: as `why` (1: `p` still holds the special flag of the round that crossed the row)
    //   5: a round this C++ code has to finish from the chain walk on (a source inside the round's own output, or a symbol
    //      that starts beyond the round's 64 output bytes): rec / nextp / ev are that round's, nothing of it is consumed
    {
      uint32_t e_asm = 0;
#ifdef EXON_WIDE_STATS
      const unsigned long long ws0 = __builtin_readcyclecounter();
#endif
      const uint32_t code = wide_rounds_asm<RING>(base, gout, bp, wb, pos, drained, carry_len, carry_rec, o.begin, o.end, br.limit, lane, cur, nxt,
                                                  fdata, faddr, rec, nextp, ev, e_asm);
#ifdef EXON_WIDE_STATS
      if (lane == 0) {
        atomicAdd(&g_wide_stats[0], __builtin_readcyclecounter() - ws0);
        atomicAdd(&g_wide_stats[2 + (code < 6 ? code : 5)], 1ull);
      }
#endif
      if (code == 0) {
        why = 0;
        e_out = e_asm;
        break;
      }
      if ((int32_t)bp < 0) {  // the window moved for a round that then was not consumed: it moves back
        wb -= 32u;
        bp += 1024u;
        cur = base[wb + lane];
        nxt = base[wb + 64u + lane];
      }
      if (code == 4) {
        why = 4;
        break;
      }
      if (code == 1) {  // rows are complete and the drain is not the plain one
        if (pos > o.end || wb + (bp >> 5) > br.limit) {
          why = 1;
          break;
        }
        complete_far();
        Out od{o.out, 0, 0, 0, drained};
        drain_to<RING>(od, pos & ~255u);
        drained = od.drained;
        continue;
      }
    }
#else
    if (bp >= 1024u) {  // move the window by 32 dwords
      const uint32_t c1 = bperm((lane ^ 32u) << 2, cur), c2 = bperm((lane ^ 32u) << 2, nxt);
      cur = lane < 32u ? c1 : c2;
      wb += 32u;
      bp -= 1024u;
      nxt = base[wb + 64u + lane];
    }
    // ---- every lane's 64 bits of the stream, from bit bp + lane
    const uint32_t t = bp + lane;
    const uint32_t a = (t >> 5) << 2;
    const uint32_t d0 = bperm(a, cur), d1 = bperm(a + 4u, cur), d2 = bperm(a + 8u, cur);
    const uint32_t wlo = __builtin_amdgcn_alignbit(d1, d0, t & 31u), whi = __builtin_amdgcn_alignbit(d2, d1, t & 31u);
    // ---- the symbol that would start there
    ev = L->lit_lut[wlo & ((1u << LIT_BITS) - 1u)];
    const uint32_t l1 = ev & 15u, xb = (ev >> 4) & 15u, lx = l1 + xb;
    const uint32_t exv = __builtin_amdgcn_ubfe(wlo, l1, xb);
    const uint32_t w2lo = (uint32_t)((((uint64_t)whi << 32) | wlo) >> lx);
    const uint32_t dv = L->dist_lut[w2lo & ((1u << DIST_BITS) - 1u)];
    const bool is_lit = ((ev >> 8) & 0xFFu) == 1u;                                  // E_LIT alone
    const bool mok = ((ev >> 8) & 0xFFu) == 8u && ((dv >> 8) & 0xFFu) == 8u;        // E_FAST alone, twice
    const uint32_t dl = dv & 15u, dxb = (dv >> 4) & 15u;
    const uint32_t dm1 = (dv >> 16) + __builtin_amdgcn_ubfe(w2lo, dl, dxb) - 1u;
    const uint32_t tot = is_lit ? l1 : lx + dl + dxb;
    // record: bits 0-8 output bytes; literal: bit 31 + the byte in bits 16-23; match: distance - 1 in bits 16-30
    rec = is_lit ? (0x80000001u | (ev & 0x00FF0000u)) : (((ev >> 16) + exv) | (dm1 << 16));
    const bool ok = is_lit || mok;
    if (!ok) rec = 0;
    nextp = ok ? lane + tot : 64u + lane;
#endif
    // ---- the chain; byte lane b keeps the record of the last symbol that starts at or before output byte b
    uint32_t p = 0, vo = carry_len, lastrec = carry_rec, rb = carry_rec;
    bool special = false;
    while (p < 64u && vo < 64u) {
      const uint32_t r = rdl(rec, p), pn = rdl(nextp, p);
      if (r == 0) {  // a symbol the tables do not resolve: the round ends in front of it
        special = true;
        break;
      }
      rb = lane >= vo ? r : rb;
      vo += r & 0x1FFu;
      lastrec = r;
      p = pn;
    }
    const uint32_t consumed = p;
    // ---- the round's output bytes
    const uint32_t nb = min(vo, 64u);
    const bool valid = lane < nb;
    const int32_t dm1b = (int32_t)rb >> 16;
    const bool blit = (int32_t)rb < 0;
    const bool bmatch = valid && !blit;
    const uint32_t dst = pos + lane, ra = dst & RM;
    uint32_t src1 = dst - (uint32_t)dm1b;  // source index + 1
    if (__any(bmatch && (uint32_t)dm1b >= dst - o.begin)) {
      why = 4;
      break;
    }
    const bool bfar = bmatch && dm1b >= (int32_t)NEARW;
    const bool bnear = bmatch && !bfar;
    // A match that overlaps its own output (distance < length: runs; 1.5 % of the matches of BAM payloads) repeats with period
    // `distance`: the bytes of the round's FIRST symbol -- carried in or not -- take their sources that many whole periods further
    // back, in front of the round, and wait for nothing.  (Left to the dependency loop below, a run of one byte costs an iteration
    // per byte: ~20 rounds' worth of instructions for one round.)
    {
      const uint32_t first_rec = carry_len ? carry_rec : rdl(rec, 0);
      const uint32_t F = carry_len ? min(carry_len, 64u) : (first_rec & 0x1FFu);
      const int32_t fd1 = (int32_t)first_rec >> 16;
      if (fd1 >= 0 && (uint32_t)fd1 + 1u < 64u && lane < F && bnear) src1 -= (lane / ((uint32_t)fd1 + 1u)) * ((uint32_t)fd1 + 1u);
    }
    const bool bdep = bnear && src1 > dst - lane;  // the source (index src1 - 1) lies inside the round's own output
    if (__builtin_expect(__any(bdep), 0)) {
      // a source inside this round's own output: produce the bytes in dependency order
      complete_far();
      if (valid && blit) ring[ra] = (uint8_t)(rb >> 16);
      if (bfar) ring[ra] = gout[src1 - 1u];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const uint32_t need = bdep ? src1 - (dst - lane) : 0u;  // bytes of the round this one waits for
      for (uint32_t D = 0; D < nb;) {
        const unsigned long long blocked = __ballot(valid && need > D);
        const uint32_t Dn = blocked ? (uint32_t)__ffsll((long long)blocked) - 1u : nb;
        if (bnear && lane >= D && lane < Dn) {
          const uint8_t v = ring[(src1 - 1u) & RM];
          ring[ra] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        D = Dn;
      }
    } else {
      complete_far();  // (a near source may lie inside the bytes still in flight)
      if (valid && blit) ring[ra] = (uint8_t)(rb >> 16);
      uint8_t nv = 0;
      if (bnear) nv = ring[(src1 - 1u) & RM];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (bnear) ring[ra] = nv;
      if (__any(bfar)) {
        uint32_t fnew = 0;
        if (bfar) fnew = gout[src1 - 1u];  // below `drained`: in HBM already
        fdata = fnew;
        faddr = bfar ? ra : FAR_NONE;
#ifdef EXON_WIDE_SYNC_FAR  // debugging: no deferral
        complete_far();
#endif
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    const uint32_t npos = pos + nb;
    const bool crossed = ((pos ^ npos) >> 8) != 0;
    pos = npos;
    carry_len = vo - nb;
    carry_rec = lastrec;
    bp += consumed;
    if (crossed) {  // rows of the ring are complete: drain them here unless there is something to report (the caller's business)
      if (pos > o.end || wb + (bp >> 5) > br.limit) {
        why = 1;
        break;
      }
      complete_far();
      Out od{o.out, 0, 0, 0, drained};
      drain_to<RING>(od, pos & ~255u);
      drained = od.drained;
    }
    if (special) {
      why = 0;
      e_out = rdl(ev, p);
      break;
    }
  }
  complete_far();
  o.drained = drained;
  if (why != 4 && carry_len != 0) {  // the rest of a match: the caller copies it (any length, any distance)
    why = 2;
    len_out = carry_len;
    d_out = ((carry_rec >> 16) & 0x7FFFu) + 1u;
  }
  // ---- back to the caller's bit reader: two whole dwords from bit bp on
  o.pos = pos;
  const uint32_t k = bp >> 5;  // < 36
  const uint32_t w0 = rdl(cur, k), w1 = rdl(cur, k + 1u);
  br.buf = (((uint64_t)w1 << 32) | w0) >> (bp & 31u);
  br.cnt = 64 - (int)(bp & 31u);
  br.widx = wb + k + 2u;
  const uint32_t w64 = br.widx & ~63u;
  if (w64 == wb) {
    br.cur = cur;
  } else {  // wb is an odd multiple of 32: the reader's window is [wb - 32, wb + 32) (its lower half is behind the reader) or [wb + 32, wb + 96)
    const uint32_t c1 = bperm((lane ^ 32u) << 2, cur), c2 = bperm((lane ^ 32u) << 2, nxt);
    br.cur = (w64 > wb && lane >= 32u) ? c2 : c1;
  }
  return why;
}

// `vflav` (wave-uniform): the hand-written symbol loop runs on the vector unit (symbol_run_v) instead of the scalar one
template <int RING, bool WIDE>
__device__ __noinline__ SymResult decode_symbols(BitReader br, Out o, int vflav) {
  constexpr uint32_t M = RING - 1;
  constexpr uint32_t NEAR = RING - 258;  // largest distance served from the ring (the copy must not overwrite its source)
  static_assert(RING >= 1024, "far matches rely on NEAR >= 258 + 255");
  br.make_uniform();
  o.make_uniform();
  vflav = uni(vflav);
  uint8_t* ring = wave_ring<RING>();
  const WaveLds* L = wave_lds<RING>();
  const uint32_t lane = lane_id();
  uint32_t vpos = o.pos;  // per-lane copy of o.pos: ring addresses of literals come from the vector ALU (literal_run)
  const uint32_t lane4 = lane * 4u;
  (void)vpos;
  (void)lane4;
  int err;  // every way out of the loop goes through ONE exit (several exit blocks cost a state variable on the back edge)
  for (;;) {
    uint32_t e;
#if EXON_INFLATE_LIT == 0
    br.refill();
    e = uniu(L->lit_lut[br.peek(LIT_BITS)]);
    if (e & E_LIT) {  // literal with a first-level code (its length field is never 0)
      br.drop((int)(e & 15u));
      ring[o.pos & M] = (uint8_t)(e >> 16);  // every lane stores the same byte
      ++o.pos;
      if (__builtin_expect((o.pos & 255u) == 0, 0)) {
        if (o.pos > o.end || br.overrun()) { err = br.overrun() ? INF_INPUT_OVERRUN : INF_OUTPUT_OVERRUN; break; }
        o.drained = uniu(drain_rows<RING>(o.out, o.drained, o.pos));
      }
      continue;
    }
#elif EXON_INFLATE_LIT == 1
    if (literal_run<RING>(br, o.pos, vpos, lane4, e)) {  // a 256-byte row of the ring is complete
      if (o.pos > o.end || br.overrun()) { err = br.overrun() ? INF_INPUT_OVERRUN : INF_OUTPUT_OVERRUN; break; }
      o.drained = uniu(drain_rows<RING>(o.out, o.drained, o.pos & ~255u));
      continue;
    }
#endif
    uint32_t len, d;
#if EXON_INFLATE_LIT == 2
    uint32_t why;
    if (WIDE) {
      why = uniu(wide_run<RING>(br, o, e, len, d));
      br.make_uniform();
      o.pos = uniu(o.pos);
      o.drained = uniu(o.drained);
      e = uniu(e);
      len = uniu(len);
      d = uniu(d);
      vpos = o.pos;
      if (__builtin_expect(why == 4, 0)) { err = INF_BAD_DISTANCE; break; }
      // rows completed inside the rounds are drained here, whatever else the loop left with (a carried match would otherwise
      // hide the crossing: the next drain would come a row late, behind far sources and, after two, behind the ring)
      if ((o.pos & ~255u) > o.drained) {
        if (o.pos > o.end || br.overrun()) { err = br.overrun() ? INF_INPUT_OVERRUN : INF_OUTPUT_OVERRUN; break; }
        o.drained = uniu(drain_rows<RING>(o.out, o.drained, o.pos & ~255u));
      }
      if (why == 1) continue;
    } else {
      why = vflav ? symbol_run_v<RING>(br, o.pos, vpos, lane, lane4, o.begin, o.out, e, len, d)
                  : symbol_run<RING>(br, o.pos, vpos, lane, lane4, o.begin, o.out, e, len, d);
    }
    if (why == 1) {  // a 256-byte row of the ring is complete
      if (o.pos > o.end || br.overrun()) { err = br.overrun() ? INF_INPUT_OVERRUN : INF_OUTPUT_OVERRUN; break; }
      o.drained = uniu(drain_rows<RING>(o.out, o.drained, o.pos & ~255u));
      continue;
    }
    if (why != 2) {
      if (why == 0) {
#else
    {
      {
#endif
        if (__builtin_expect((e & 15u) == 0, 0)) {  // a code longer than the table, or no such code
          const int r = uni(decode_long<RING>(CODE_LIT, (uint32_t)br.buf));
          // lengths are <= 15: the entry's length field holds it; no such code -> E_INVALID (caught below, no exit from here)
          e = r < 0 ? (uint32_t)E_INVALID : entry_for(CODE_LIT, r >> 8) | (uint32_t)(r & 255);
        }
        br.drop((int)(e & 15u));
        if (__builtin_expect((e & (E_LIT | E_EOB | E_INVALID)) != 0, 0)) {
          if (!(e & E_LIT)) { err = (e & E_EOB) ? INF_OK : INF_BAD_CODE; break; }
          ring[o.pos & M] = (uint8_t)(e >> 16);  // a literal with a long code; every lane stores the same byte
          ++o.pos;
          vpos = o.pos;
          if ((o.pos & 255u) == 0) {
            if (o.pos > o.end || br.overrun()) { err = br.overrun() ? INF_INPUT_OVERRUN : INF_OUTPUT_OVERRUN; break; }
            o.drained = uniu(drain_rows<RING>(o.out, o.drained, o.pos));
          }
          continue;
        }
        len = (e >> 16) + br.take((int)((e >> 4) & 15u));
        br.refill();
      }
      const uint32_t de = decode_symbol<RING, CODE_DIST>(br, L);
      if (__builtin_expect((de & E_INVALID) != 0, 0)) { err = INF_BAD_CODE; break; }
      d = (de >> 16) + br.take((int)((de >> 4) & 15u));
    }
    if (__builtin_expect(d > o.pos - o.begin, 0)) { err = INF_BAD_DISTANCE; break; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (d - len <= NEAR - len) {  // len <= d <= NEAR (unsigned wrap-around when d < len)
      if (len <= 64) {  // nearly all matches: one masked read + write, no loop bookkeeping
        if (lane < len) ring[(o.pos + lane) & M] = ring[(o.pos - d + lane) & M];
      } else {
        for (uint32_t j = lane; j < len; j += 64) ring[(o.pos + j) & M] = ring[(o.pos - d + j) & M];
      }
    } else if (d <= NEAR) {
      copy_overlapping<RING>(o.pos, d, len);
    } else {
      // far: d > NEAR >= 258 + 255, so the source ends below `drained` (pos - drained < 256): it is in HBM already
      // (common: DEFLATE windows are 32 KiB, the ring holds 2) -- inline, a call costs ~40 scalar instructions
      if (len <= 64) {
        if (lane < len) ring[(o.pos + lane) & M] = o.out[o.pos - d + lane];
      } else {
        for (uint32_t j = lane; j < len; j += 64) ring[(o.pos + j) & M] = o.out[o.pos - d + j];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t npos = o.pos + len;
    if (__builtin_expect(((o.pos ^ npos) >> 8) != 0, 0)) {  // crossed a 256-byte row
      if (npos > o.end || br.overrun()) { err = br.overrun() ? INF_INPUT_OVERRUN : INF_OUTPUT_OVERRUN; break; }
      o.drained = uniu(drain_rows<RING>(o.out, o.drained, npos & ~255u));
    }
    o.pos = npos;
    vpos = npos;
  }
  return SymResult{br, o, err};
}


// ================================================================================================================
// Lane-parallel decode of one DEFLATE block (speculative decoding, round 2).  The symbol chain of a block is serial, but
// Huffman streams SELF-SYNCHRONISE: a decoder started at a wrong bit offset falls into step with the true symbol boundaries
// after a few symbols (measured on VCF text: median 110 bits, 95 % within 512, 99.6 % within 1024).  So the payload bits
// [P0, P1) are cut into 64 dword-aligned ranges, one per lane:
//   phase 1  every lane decodes from its range start to its range end (lane 0 from the true start), one symbol per step in
//            lockstep: token records go to scratch ([step][lane]: coalesced), a bitmap marks each symbol start;
//   phase 2  every lane keeps decoding past its range end until it lands on a marked bit (= it merged into the chain of a
//            later lane) or meets the end-of-block code;
//   phase 3  the true chain is followed lane to lane from lane 0: lane -> its continuation -> the lane it merged into (from
//            the merge index on) -> ...; lanes that were stepped over are dropped;
//   phase 4  output offsets by prefix sums over the surviving token ranges; literals are stored by their lanes, matches
//            are listed in output order;
//   phase 5  matches are copied 64 at a time, one per lane, in rounds: a match runs once its source lies below the first
//            unfinished match of its batch (far matches -- most of them in text -- all run in the first round).
// Anything unusual (caps exceeded, a dead end on the true chain, an invalid code) returns nonzero WITHOUT committing:
// the caller then decodes the block with the serial symbol loop, which also owns all error reporting.
// ================================================================================================================
constexpr int PAR_T1 = 512, PAR_T2 = 256;            // token steps per lane: own range / continuation
constexpr int PAR_BM_WORDS = 16384 + 64;             // one bit per payload bit of a member (<= 64 KiB of DEFLATE data)
constexpr int PAR_OL = 65536 + 64;                   // ordered output tokens of one member (each yields >= 1 byte)
constexpr size_t PAR_SLOT_BYTES = (size_t)(PAR_T1 + PAR_T2) * 64 * 8 + (size_t)PAR_BM_WORDS * 4 + (size_t)PAR_OL * 8;
constexpr uint32_t PAR_MIN_BITS = 64 * 96;           // smaller payloads are not worth the set-up
constexpr uint32_t PAR_MAX_LANE_BITS = 3072;         // = 24 KiB of DEFLATE data per member
constexpr uint32_t PAR_WIN = EXON_INFLATE_RING, PAR_SPAN = 1024;  // the window IS the serial path's ring (idle while a block is decoded in parallel)  // phase 5: LDS window of recent output / output bytes per batch of tokens
struct ParSlot {
  uint64_t* t1;   // [PAR_T1][64]
  uint64_t* t2;   // [PAR_T2][64]
  uint32_t* bm;   // [PAR_BM_WORDS]
  uint64_t* ol;   // [PAR_OL]  dst (32) | literal byte, or T_MATCH | (len - 1) (len <= 16) | dist << 8
};
__device__ __forceinline__ ParSlot par_slot(uint8_t* base, uint32_t slot) {
  uint8_t* p = base + (size_t)slot * PAR_SLOT_BYTES;
  ParSlot s;
  s.t1 = reinterpret_cast<uint64_t*>(p);
  s.t2 = s.t1 + (size_t)PAR_T1 * 64;
  s.ol = s.t2 + (size_t)PAR_T2 * 64;
  s.bm = reinterpret_cast<uint32_t*>(s.ol + PAR_OL);
  return s;
}
// token: lo = literal byte | T_MATCH | (len - 3) | (dist - 1) << 8 | T_EOB ; hi = cum (18 bits) | relpos << 18
constexpr uint32_t T_MATCH = 1u << 31, T_EOB = 1u << 30;
enum { PE_RUN = 0, PE_EXIT = 1, PE_EOB = 2, PE_DEAD = 3, PE_MERGED = 4 };

// per-lane canonical decode of a code longer than the first-level table, resumed after its first `kbits` lengths
// (first_k / index_k = the bit-serial loop's state after kbits lengths, wave-uniform).  Returns entry | length, or E_INVALID.
__device__ __forceinline__ uint32_t par_long(int which, const uint16_t* count, const uint16_t* sym, uint64_t w, int kbits, int first_k,
                                             int index_k) {
  int code = (int)(__brev((uint32_t)w) >> (32 - kbits));
  int first = first_k, index = index_k;
  uint32_t bits = (uint32_t)(w >> kbits);
  for (int len = kbits + 1; len <= 15; ++len) {
    code = (code << 1) | (int)(bits & 1u);
    bits >>= 1;
    const int n = (int)count[len];
    if (code - n < first) return entry_for(which, (int)sym[index + (code - first)]) | (uint32_t)len;
    index += n;
    first += n;
    first <<= 1;
  }
  return E_INVALID;
}


// A match (len, dist) as pieces of <= 16 bytes that never overlap their own source (piece dist >= piece len), so that phase 5
// copies every piece with one aligned read and one write.  dist >= 16 (or dist >= len): consecutive 16-byte cuts with the
// match's distance.  A run (dist < len, dist < 16) repeats its `dist` bytes: the first piece copies one period, and every
// later piece copies from a whole number of periods back, as many bytes as are already there (d, 2d, 4d ... up to 16).
template <class F>
__device__ __forceinline__ uint32_t par_pieces(uint32_t len, uint32_t dist, F&& emit) {
  uint32_t n = 0;
  if (dist >= 16u || dist >= len) {
    for (uint32_t q = 0; q < len; q += 16u, ++n) emit(q, min(16u, len - q), dist);
    return n;
  }
  uint32_t w = 0, avail = dist;  // avail = the largest multiple of dist <= w + dist
  while (w < len) {
    const uint32_t pl = min(min(16u, len - w), avail);
    uint32_t pd = dist;
    while (pd < pl) pd += dist;
    emit(w, pl, pd);
    ++n;
    w += pl;
    while (avail + dist <= w + dist) avail += dist;
  }
  return n;
}

struct ParSym {
  uint32_t tok;   // token low word
  uint32_t olen;  // output bytes
  uint32_t bits;  // bits consumed
  int kind;       // PE_RUN (ordinary) / PE_EOB / PE_DEAD
};
struct ParCodes {
  int lit_first, lit_index, dist_first, dist_index;  // par_long resume state (wave-uniform)
};
// A lane's window on the compressed bits: three aligned 8-byte words, the third one always in flight -- a symbol takes at
// most 48 bits, so the word loaded when the window moves is first needed two moves later and the memory latency stays off
// the symbol chain (with one unaligned load per symbol every step waited a full L2 round trip).
struct ParBits {
  const uint64_t* base;  // member start rounded down to 8 bytes
  uint32_t skew;         // bits between `base` and bit 0 of the member
  uint32_t wbit;         // bit index (from base) of w0, a multiple of 64
  uint64_t w0, w1, w2;
  __device__ __forceinline__ void init(const uint8_t* comp, uint32_t p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(comp);
    base = reinterpret_cast<const uint64_t*>(a & ~uintptr_t(7));
    skew = (uint32_t)(a & 7u) * 8u;
    wbit = (p + skew) & ~63u;
    const uint64_t* q = base + (wbit >> 6);
    w0 = q[0];
    w1 = q[1];
    w2 = q[2];
  }
  // >= 57 valid bits starting at member bit p; p never moves backwards and by at most 48 bits between calls
  __device__ __forceinline__ uint64_t at(uint32_t p) {
    uint32_t off = p + skew - wbit;
    if (off >= 64u) {
      w0 = w1;
      w1 = w2;
      wbit += 64u;
      w2 = base[(wbit >> 6) + 2];
      off -= 64u;
    }
    return off ? (w0 >> off) | (w1 << (64u - off)) : w0;
  }
};
// one symbol (literal, or length + distance) at bit position p of the member (per lane)
template <int RING>
__device__ __forceinline__ ParSym par_symbol(ParBits& pb, uint32_t p, const WaveLds* L, const ParCodes& pc) {
  uint64_t w = pb.at(p);  // code 15 + extra 5 + code 15 + extra 13 = 48 bits at most
  ParSym r;
  uint32_t e = L->lit_lut[(uint32_t)w & ((1u << LIT_BITS) - 1u)];
  if (__builtin_expect((e & 15u) == 0, 0)) e = par_long(CODE_LIT, L->lit_count, L->lit_sym, w, LIT_BITS, pc.lit_first, pc.lit_index);
  uint32_t used = e & 15u;
  r.kind = PE_RUN;
  if (e & E_LIT) {
    r.tok = e >> 16;
    r.olen = 1;
    r.bits = used;
    return r;
  }
  if (__builtin_expect((e & (E_EOB | E_INVALID)) != 0, 0)) {
    r.kind = (e & E_INVALID) ? PE_DEAD : PE_EOB;
    r.tok = T_EOB;
    r.olen = 0;
    r.bits = used;
    return r;
  }
  w >>= used;
  const uint32_t xb = (e >> 4) & 15u;
  const uint32_t len = (e >> 16) + ((uint32_t)w & ((1u << xb) - 1u));
  w >>= xb;
  used += xb;
  uint32_t de = L->dist_lut[(uint32_t)w & ((1u << DIST_BITS) - 1u)];
  if (__builtin_expect((de & 15u) == 0, 0)) de = par_long(CODE_DIST, L->dist_count, L->dist_sym, w, DIST_BITS, pc.dist_first, pc.dist_index);
  if (__builtin_expect((de & E_INVALID) != 0, 0)) r.kind = PE_DEAD;
  const uint32_t dl = de & 15u;
  w >>= dl;
  const uint32_t dxb = (de >> 4) & 15u;
  const uint32_t dist = (de >> 16) + ((uint32_t)w & ((1u << dxb) - 1u));
  r.tok = T_MATCH | (len - 3u) | ((dist - 1u) << 8);
  r.olen = len;
  r.bits = used + dl + dxb;
  return r;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
  return uniu(v);
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t* total) {
  const int lane = (int)lane_id();
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, d);
    if (lane >= d) x += y;
  }
  *total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  return x - v;
}

// Decodes the DEFLATE block whose symbols start at bit P0 of the member (tables are built) straight into out[o.pos ..].
// On success returns 0, advances o.pos / o.drained and sets *end_bit to the bit after the end-of-block code.
template <int RING>
__device__ __noinline__ int par_decode_block(const uint8_t* __restrict__ comp /* member base */, uint32_t P0, uint32_t P1, Out& o,
                                             const ParSlot sl, uint32_t* end_bit, unsigned* __restrict__ stats) {
  const WaveLds* L = wave_lds<RING>();
  const int lane = (int)lane_id();
  uint64_t tq = __builtin_amdgcn_s_memtime();
  auto lap = [&](int idx) {  // EXON_INFLATE_PROFILE: kilo-clocks per phase into stats[8 + idx]
#ifdef EXON_INFLATE_PROFILE
    const uint64_t now = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicAdd(&stats[8 + idx], (unsigned)((now - tq) >> 10));
    tq = now;
#endif
  };
  (void)tq;
  P0 = uniu(P0);
  P1 = uniu(P1);
  if (P1 <= P0 || P1 - P0 < PAR_MIN_BITS || (P1 >> 5) + 2 >= (uint32_t)PAR_BM_WORDS) return 1;
  // resume state of the long-code decoders
  ParCodes pc;
  {
    int first = 0, index = 0;
    for (int q = 1; q <= LIT_BITS; ++q) {
      const int n = uni((int)L->lit_count[q]);
      index += n;
      first += n;
      first <<= 1;
    }
    pc.lit_first = first;
    pc.lit_index = index;
    first = index = 0;
    for (int q = 1; q <= DIST_BITS; ++q) {
      const int n = uni((int)L->dist_count[q]);
      index += n;
      first += n;
      first <<= 1;
    }
    pc.dist_first = first;
    pc.dist_index = index;
  }
  // ---- ranges: A = P0 rounded down to a dword; lane l owns bits [A + l S, A + (l + 1) S), S a multiple of 32
  const uint32_t A = P0 & ~31u;
  const uint32_t S = (((P1 - A + 63u) / 64u) + 31u) & ~31u;
  if (S + 64u >= (1u << 14)) return 1;  // relpos field
  if (S > PAR_MAX_LANE_BITS) return 1;   // literal-heavy members (FASTQ) would run out of token steps: the serial loop is faster there
  const uint32_t rs = lane == 0 ? P0 : A + (uint32_t)lane * S;
  const uint32_t re = min(P1, A + (uint32_t)(lane + 1) * S);
  const int nlanes = (int)((P1 - A + S - 1u) / S);
  // zero the bitmap over the payload
  for (uint32_t wi = (A >> 5) + (uint32_t)lane; wi <= (P1 >> 5) + 1u; wi += 64) sl.bm[wi] = 0;

  lap(0);
  // ---- phase 1 ---------------------------------------------------------------------------------------------------
  uint32_t p = rs, cum = 0, n1 = 0, tc = 0;  // tc: output tokens so far (a match counts once per 16 bytes)
  int kind = lane < nlanes && rs < P1 ? PE_RUN : PE_DEAD;
  ParBits pb;
  pb.init(comp, kind == PE_RUN ? rs : P0);
  uint32_t bm_wi = rs >> 5, bm_acc = 0;
  for (int step = 0; step < PAR_T1; ++step) {
    const bool act = kind == PE_RUN;
    if (!__any(act)) break;
    if (act) {
      const ParSym sy = par_symbol<RING>(pb, p, L, pc);
      const uint32_t wi = p >> 5;
      if (wi != bm_wi) {
        sl.bm[bm_wi] = bm_acc;
        bm_acc = 0;
        bm_wi = wi;
      }
      bm_acc |= 1u << (p & 31u);
      sl.t1[(size_t)step * 64 + lane] = (uint64_t)sy.tok | ((uint64_t)(cum | ((p - rs) << 18)) << 32);
      ++n1;
      cum += sy.olen;
      tc += (sy.tok & T_MATCH) ? par_pieces(sy.olen, ((sy.tok >> 8) & 0x7FFFu) + 1u, [](uint32_t, uint32_t, uint32_t) {}) : sy.olen;
      p += sy.bits;
      if (sy.kind != PE_RUN) kind = sy.kind;
      else if (cum >= (1u << 18)) kind = PE_DEAD;
      else if (p >= re) kind = p > P1 ? PE_DEAD : PE_EXIT;
    }
  }
  if (kind == PE_RUN) kind = PE_DEAD;  // out of token space
#ifdef EXON_INFLATE_PROFILE
  { const uint32_t mx = wave_max_u32(n1); if (lane == 0) atomicAdd(&stats[14], mx); }
#endif
  if (lane < nlanes && rs < P1) sl.bm[bm_wi] = bm_acc;
  const uint32_t cum1 = cum;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  lap(1);
  // ---- phase 2: continue until the chain lands on a symbol start of a later lane --------------------------------
  int ckind = kind == PE_EXIT ? PE_RUN : PE_DEAD;
  uint32_t cn = 0, mpos = 0;
  for (int step = 0; step < PAR_T2; ++step) {
    bool act = ckind == PE_RUN;
    if (!__any(act)) break;
    if (act) {
      if (p >= P1) {
        ckind = PE_DEAD;
      } else if ((sl.bm[p >> 5] >> (p & 31u)) & 1u) {
        ckind = PE_MERGED;
        mpos = p;
      } else {
        const ParSym sy = par_symbol<RING>(pb, p, L, pc);
        sl.t2[(size_t)step * 64 + lane] = (uint64_t)sy.tok | ((uint64_t)cum << 32);
        ++cn;
        cum += sy.olen;
        tc += (sy.tok & T_MATCH) ? par_pieces(sy.olen, ((sy.tok >> 8) & 0x7FFFu) + 1u, [](uint32_t, uint32_t, uint32_t) {}) : sy.olen;
        p += sy.bits;
        if (sy.kind != PE_RUN) ckind = sy.kind;
        else if (cum >= (1u << 18) || p > P1) ckind = PE_DEAD;
      }
    }
  }
  if (ckind == PE_RUN) ckind = PE_DEAD;
#ifdef EXON_INFLATE_PROFILE
  { const uint32_t mx = wave_max_u32(cn); if (lane == 0) atomicAdd(&stats[15], mx); }
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  lap(2);
  // ---- phase 3: follow the true chain ---------------------------------------------------------------------------
  unsigned long long valid = 0;
  uint32_t k0 = 0;  // per lane: first valid token of its own list
  int last = -1;
  bool end_in_cont = false;
  {
    int cur = 0;
    uint32_t kcur = 0;
    for (int it = 0; it < 64; ++it) {
      valid |= 1ull << cur;
      if (lane == cur) k0 = kcur;
      const int ek = __builtin_amdgcn_readlane(kind, cur);
      if (ek == PE_EOB) {
        last = cur;
        break;
      }
      if (ek != PE_EXIT) return 2;
      const int ck = __builtin_amdgcn_readlane(ckind, cur);
      if (ck == PE_EOB) {
        last = cur;
        end_in_cont = true;
        break;
      }
      if (ck != PE_MERGED) return 3;
      const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mpos, cur);
      const int nxt = (int)((m - A) / S);
      if (nxt <= cur || nxt >= nlanes) return 4;
      const uint32_t n1n = (uint32_t)__builtin_amdgcn_readlane((int)n1, nxt);
      const uint32_t target = m - (A + (uint32_t)nxt * S);
      int found = -1;
      for (uint32_t kb = 0; kb < n1n; kb += 64) {
        const uint32_t k = kb + (uint32_t)lane;
        const bool hit = k < n1n && (uint32_t)(sl.t1[(size_t)k * 64 + nxt] >> 50) == target;
        const unsigned long long bal = __ballot(hit);
        if (bal) {
          found = (int)kb + __ffsll((long long)bal) - 1;
          break;
        }
      }
      if (found < 0) return 5;
      cur = nxt;
      kcur = (uint32_t)found;
    }
    if (last < 0) return 6;
  }
  const bool mine = (valid >> lane) & 1ull;
  // the end-of-block token is the last one of lane `last` (own list, or continuation)
  *end_bit = (uint32_t)__builtin_amdgcn_readlane((int)p, last);

  lap(3);
  // ---- phase 4: output offsets, literals, the ordered match list -------------------------------------------------
  const uint32_t cum_k0 = mine ? (uint32_t)(sl.t1[(size_t)k0 * 64 + lane] >> 32) & 0x3FFFFu : 0u;
  // a lane that ended with EOB inside its own range has no continuation (cn = 0); every other valid lane's continuation counts
  const uint32_t obytes = mine ? cum - cum_k0 : 0u;
  uint32_t total;
  const uint32_t obase = wave_excl_scan(obytes, &total);
  if (o.pos + total > o.end) return 7;
  (void)cum1;
  (void)end_in_cont;
  // ---- phase 4: the tokens in output order.  Matches are cut into pieces of <= 16 bytes (a copy is byte-sequential, so the
  // pieces are ordinary matches with the same distance): phase 5 then never runs a long per-lane loop.
  uint32_t tcount = 0;
  const uint32_t n1max = wave_max_u32(mine ? n1 : 0u), cnmax = wave_max_u32(mine ? cn : 0u);
  // visit(lo, cum) for every valid token of this lane, own list then continuation; 8 coalesced loads in flight per lane
  auto walk = [&](auto&& visit) {
    for (uint32_t kb = 0; kb < n1max; kb += 8) {
      uint64_t t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = sl.t1[(size_t)min(kb + j, (uint32_t)PAR_T1 - 1u) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t k = kb + j;
        if (mine && k >= k0 && k < n1) visit((uint32_t)t[j], (uint32_t)(t[j] >> 32) & 0x3FFFFu);
      }
    }
    for (uint32_t kb = 0; kb < cnmax; kb += 8) {
      uint64_t t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = sl.t2[(size_t)min(kb + j, (uint32_t)PAR_T2 - 1u) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t k = kb + j;
        if (mine && k < cn) visit((uint32_t)t[j], (uint32_t)(t[j] >> 32));
      }
    }
  };
  {  // tokens of this lane's list before k0 are not on the true chain: count them out (k0 is small: chains merge early)
    uint32_t before = 0;
    const uint32_t k0max = wave_max_u32(mine ? k0 : 0u);
    for (uint32_t kb = 0; kb < k0max; kb += 8) {
      uint64_t t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = sl.t1[(size_t)min(kb + j, (uint32_t)PAR_T1 - 1u) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t lo = (uint32_t)t[j];
        if (mine && kb + j < k0)
          before += (lo & T_MATCH) ? par_pieces((lo & 255u) + 3u, ((lo >> 8) & 0x7FFFu) + 1u, [](uint32_t, uint32_t, uint32_t) {}) : ((lo & T_EOB) ? 0u : 1u);
      }
    }
    tcount = mine ? tc - before : 0u;
  }
  uint32_t T;
  const uint32_t tbase = wave_excl_scan(tcount, &T);
  if (T > (uint32_t)PAR_OL - 64u) return 8;
  {
    uint32_t ti = tbase;
    const uint32_t dpos0 = o.pos + obase - cum_k0;
    walk([&](uint32_t lo, uint32_t c) {
      if (lo & T_MATCH) {
        par_pieces((lo & 255u) + 3u, ((lo >> 8) & 0x7FFFu) + 1u, [&](uint32_t q, uint32_t pl, uint32_t pd) {
          sl.ol[ti++] = (uint64_t)(dpos0 + c + q) | ((uint64_t)(T_MATCH | (pl - 1u) | (pd << 8)) << 32);
        });
      } else if (!(lo & T_EOB)) {
        sl.ol[ti++] = (uint64_t)(dpos0 + c) | ((uint64_t)(lo & 255u) << 32);
      }
    });
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  lap(4);
  // ---- phase 5: replay through an LDS window, 64 tokens at a time -------------------------------------------------
  // The window holds output bytes [batch end - PW, batch end) (slot = position mod PW); rows of 256 complete bytes are
  // flushed to HBM as the serial kernel's ring does.  Literals store their byte; a match whose source starts below the
  // window ("far") copies from HBM -- such a source always lies below `drained` (PW - PSPAN > 258 + 255) -- the others copy
  // inside the window in dependency rounds: a piece runs once its source ends below the first unfinished piece.
  {
    constexpr uint32_t PM = PAR_WIN - 1u;
    uint8_t* const out = o.out;
    uint8_t* const win = wave_ring<RING>();
    static_assert(PAR_WIN - PAR_SPAN > 258 + 255 + 16, "far sources must end below the drained rows");
    uint32_t pos = o.pos, drained = o.pos;
    // bytes of earlier DEFLATE blocks of this member that a near match may reach (they are in HBM: the caller drained)
    const uint32_t hist = min(o.pos - o.begin, PAR_WIN - PAR_SPAN - 530u);
    const uint32_t wvalid = o.pos - hist;
    for (uint32_t x = wvalid + (uint32_t)lane; x < o.pos; x += 64) win[x & PM] = out[x];
    auto flush_to = [&](uint32_t limit) {  // window -> HBM for [drained, limit)
      while (drained < limit) {
        const uint32_t row_end = min(limit, (drained | 255u) + 1u);
        if (((drained | row_end) & 3u) == 0) {
          const uint32_t x = drained + 4u * (uint32_t)lane;
          if (x < row_end) *reinterpret_cast<uint32_t*>(out + x) = *reinterpret_cast<const uint32_t*>(win + (x & PM));
        } else {
          for (uint32_t x = drained + (uint32_t)lane; x < row_end; x += 64) out[x] = win[x & PM];
        }
        drained = row_end;
      }
    };
    // the window is the serial path's ring: a late hand-back (only corrupt data gets here) must leave it as it was found,
    // i.e. holding the last bytes in front of this block, all of which the caller has drained to HBM
    auto restore_ring = [&]() {
      __builtin_amdgcn_s_waitcnt(0);
      const uint32_t lo = o.pos - o.begin > PAR_WIN ? o.pos - PAR_WIN : o.begin;
      for (uint32_t x = lo + (uint32_t)lane; x < o.pos; x += 64) win[x & PM] = out[x];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    int bad = 0;
#ifdef EXON_INFLATE_PROFILE
    uint64_t t_wait = 0, t_rounds = 0;
    uint32_t n_batches = 0, n_rounds = 0;
#endif
    uint64_t rec = (uint32_t)lane < T ? sl.ol[lane] : ~0ull;
    for (uint32_t i0 = 0; i0 < T;) {
      const uint32_t dst = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
      const bool is_match = (hi & T_MATCH) != 0;
      const uint32_t len = is_match ? (hi & 15u) + 1u : 1u, dist = (hi >> 8) & 0xFFFFu;
      const bool fits = i0 + (uint32_t)lane < T && dst + len - pos <= PAR_SPAN;
      const uint32_t nb = (uint32_t)__popcll(__ballot(fits));  // dst increases with the lane: the fitting tokens are a prefix; >= 1
      const bool have = (uint32_t)lane < nb;
      const uint32_t bend = (uint32_t)__builtin_amdgcn_readlane((int)(dst + len), (int)nb - 1);
      const uint32_t lo_valid = max(wvalid, bend > PAR_WIN ? bend - PAR_WIN : 0u);
      if (have && is_match && dist > dst - o.begin) bad = 1;  // reaches before the start of the member's output
      if (__any(bad)) {
        restore_ring();
        return 9;
      }
      const uint32_t src = dst - dist;
      const bool far = have && is_match && src < lo_valid;
      bool pending = have && is_match && !far;
      // vector memory, in this order: rows completed by the previous batch -> HBM (stores); the next batch's records and
      // this batch's far sources (loads).  ONE wait then covers all of it (the counter is in-order: a load waits for
      // every older store anyway).
      if ((pos & ~255u) > drained) flush_to(pos & ~255u);
      const uint32_t inext = i0 + nb;
      const uint64_t rec_next = inext + (uint32_t)lane < T ? sl.ol[inext + lane] : ~0ull;
      uint32_t v[4] = {0, 0, 0, 0};
      if (far) {  // the source ends below `drained`
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((uint32_t)(4 * i) < len) __builtin_memcpy(&v[i], out + src + 4 * i, 4);
      }
#ifdef EXON_INFLATE_PROFILE
      const uint64_t tw0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0);
      const uint64_t tw1 = __builtin_amdgcn_s_memtime();
      t_wait += tw1 - tw0;
      ++n_batches;
#endif
      // `len` (<= 16) bytes held in v[0..3] -> window at dst: bytes up to the first dword boundary, whole dwords, tail bytes
      // (10 LDS stores at most instead of 16; a read-modify-write of the edge dwords would race with the neighbour piece)
      auto put = [&](const uint32_t* v, uint32_t dst, uint32_t len) {
        const uint32_t head = min(len, (0u - dst) & 3u);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if ((uint32_t)i < head) win[(dst + i) & PM] = (uint8_t)(v[0] >> (8 * i));
        // the stream re-aligned to the first dword boundary: w[j] = bytes head + 4 j .. of v
        const uint32_t body = (len - head) >> 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((uint32_t)j < body) {
            const uint32_t lo_w = v[j], hi_w = j + 1 < 4 ? v[j + 1] : 0u;
            const uint32_t w = head == 0 ? lo_w : (lo_w >> (8 * head)) | (hi_w << (32 - 8 * head));
            *reinterpret_cast<uint32_t*>(win + ((dst + head + 4 * j) & PM)) = w;
          }
        }
        const uint32_t tail0 = head + 4 * body;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const uint32_t q = tail0 + i;
          if (q < len) win[(dst + q) & PM] = (uint8_t)(v[q >> 2 & 3] >> (8 * (q & 3)));
        }
      };
      // `len` (<= 16) bytes of the window starting at src -> v[0..3]: five aligned dwords, byte-aligned in registers
      auto get = [&](uint32_t src, uint32_t* v) {
        const uint32_t a = src & ~3u, sh = src & 3u;
        uint32_t d[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) d[i] = *reinterpret_cast<const uint32_t*>(win + ((a + 4 * i) & PM));
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
      };
      if (have && !is_match) win[dst & PM] = (uint8_t)hi;
      if (far) put(v, dst, len);
#ifdef EXON_INFLATE_PROFILE
      const uint64_t tr0 = __builtin_amdgcn_s_memtime();
#endif
      while (__any(pending)) {
#ifdef EXON_INFLATE_PROFILE
        ++n_rounds;
#endif
        const int f = __ffsll((long long)__ballot(pending)) - 1;
        const uint32_t dst_f = (uint32_t)__builtin_amdgcn_readlane((int)dst, f);
        if (pending && (lane == f || src + len <= dst_f)) {
          uint32_t w[4];
          get(src, w);  // pieces never overlap their source (par_pieces)
          put(w, dst, len);
          pending = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
#ifdef EXON_INFLATE_PROFILE
      __builtin_amdgcn_s_waitcnt(0);
      t_rounds += __builtin_amdgcn_s_memtime() - tr0;
#endif
      pos = bend;
      i0 = inext;
      rec = rec_next;
    }
#ifdef EXON_INFLATE_PROFILE
    if (lane == 0) {
      atomicAdd(&stats[16], (unsigned)(t_wait >> 10));
      atomicAdd(&stats[17], (unsigned)(t_rounds >> 10));
      atomicAdd(&stats[18], n_batches);
      atomicAdd(&stats[19], n_rounds);
    }
#endif
    flush_to(pos);
    if (pos != o.pos + total) {
      restore_ring();
      return 10;
    }
  }
  lap(5);
  o.pos += total;
  o.drained = o.pos;
  return 0;
}

// One BGZF member.  PAR: dynamic / fixed DEFLATE blocks first try par_decode_block (scratch slot `sl`, fallback counters `stats`).
template <int RING, bool PAR, bool WIDE = false>
__device__ __forceinline__ void inflate_member(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks, int b, uint8_t* out,
                                               int* __restrict__ status, const ParSlot sl, unsigned* __restrict__ stats, int vflav) {
  constexpr uint32_t M = RING - 1;
  const int lane = (int)lane_id();
  WaveLds* L = wave_lds<RING>();
  uint8_t* ring = wave_ring<RING>();
  if (lds_addr(wave_ring<RING>()) != 0) {  // literal_run addresses the ring and the table with immediates (folds away when true)
    if (lane == 0) status[b] = INF_BAD_BTYPE;
    return;
  }
  const Block blk = blocks[b];
  const uint32_t comp_end = blk.comp_offset + blk.comp_size;

  BitReader br;
  br.init(comp, blk.comp_offset, comp_end);
  Out o{out, blk.out_offset, blk.out_offset + blk.out_size, blk.out_offset, blk.out_offset};
  int err = INF_OK;

  bool last = false;
  while (!last && err == INF_OK) {
    // a corrupt member made of empty non-final blocks must not keep reading: stop at the end of the compressed data
    // instead of relying on what lies behind it
    if (br.overrun()) { err = INF_INPUT_OVERRUN; break; }
    br.refill();
    last = br.take(1) != 0;
    const int btype = (int)br.take(2);
    if (btype == 0) {  // stored: input -> HBM directly, the tail also into the ring for later matches
      if (o.pos > o.end) { err = INF_OUTPUT_OVERRUN; break; }
      drain_to<RING>(o, o.pos);
      br.drop(br.cnt & 7);
      br.refill();
      const uint32_t len = br.take(16);
      br.refill();
      const uint32_t nlen = br.take(16);
      if ((len ^ nlen) != 0xFFFFu) { err = INF_BAD_STORED; break; }
      const uint32_t src = br.byte_pos();
      if (src + len > comp_end) { err = INF_INPUT_OVERRUN; break; }
      if (o.pos + len > o.end) { err = INF_OUTPUT_OVERRUN; break; }
      const uint32_t keep = len > (uint32_t)RING ? len - (uint32_t)RING : 0u;
#pragma unroll 2  // (unrolled by 8 this loop alone took the kernel from 64 to 70 VGPRs = from 8 to 7 waves per SIMD)
      for (uint32_t j = (uint32_t)lane; j < len; j += 64) {
        const uint8_t v = comp[src + j];
        out[o.pos + j] = v;
        if (j >= keep) ring[(o.pos + j) & M] = v;
      }
      o.pos += len;
      o.drained = o.pos;
      br.init(comp, src + len, comp_end);
      continue;
    }
    if (btype == 3) { err = INF_BAD_BTYPE; break; }
    if (btype == 1) {  // fixed codes
      for (int s = lane; s < 288; s += 64) L->lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
      if (lane < 32) L->lens[288 + lane] = 5;  // 30 and 31 complete the code; using them is an error
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // the literal/length code LAST: its first[] / offs[] must survive the build (decode_long)
      if (!build_code<RING>(CODE_DIST, 288, 32) || !build_code<RING>(CODE_LIT, 0, 288)) { err = INF_BAD_LENGTHS; break; }
    } else {  // dynamic codes
      br.refill();
      const int hlit = (int)br.take(5) + 257, hdist = (int)br.take(5) + 1, hclen = (int)br.take(4) + 4;
      if (hlit > 286 || hdist > 30) { err = INF_BAD_LENGTHS; break; }
      if (lane < 19) L->lens[lane] = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int i = 0; i < hclen; ++i) {
        br.refill();
        const uint8_t v = (uint8_t)br.take(3);
        L->lens[cl_order(i)] = v;  // every lane writes the same byte
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (!build_code<RING>(CODE_CL, 0, 19)) { err = INF_BAD_LENGTHS; break; }
      // code lengths of the literal/length and distance alphabets, run-length coded
      int i = 0, prev = 0;
      const int total = hlit + hdist;
      while (i < total) {
        br.refill();
        const uint32_t ce = decode_symbol<RING, CODE_CL>(br, L);
        if (ce & E_INVALID) { err = INF_BAD_CODE; break; }
        const int s = (int)(ce >> 16);
        if (s < 16) {
          L->lens[i++] = (uint8_t)s;
          prev = s;
        } else {
          int rep, val = 0;
          if (s == 16) {
            if (i == 0) { err = INF_BAD_LENGTHS; break; }
            val = prev;
            rep = 3 + (int)br.take(2);
          } else if (s == 17) {
            rep = 3 + (int)br.take(3);
          } else {
            rep = 11 + (int)br.take(7);
          }
          if (i + rep > total) { err = INF_BAD_LENGTHS; break; }
          for (int k = lane; k < rep; k += 64) L->lens[i + k] = (uint8_t)val;
          i += rep;
          prev = val;
        }
      }
      if (err) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (uni((int)L->lens[256]) == 0) { err = INF_BAD_LENGTHS; break; }  // no end-of-block code
      // the distance lengths follow the literal/length lengths directly: move them behind the 288 slots
      const int dl = lane < hdist ? L->lens[hlit + lane] : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane < 32) L->lens[288 + lane] = (uint8_t)dl;
      for (int s = hlit + lane; s < 288; s += 64) L->lens[s] = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // the literal/length code LAST: its first[] / offs[] must survive the build (decode_long)
      if (!build_code<RING>(CODE_DIST, 288, 30) || !build_code<RING>(CODE_LIT, 0, 288)) { err = INF_BAD_LENGTHS; break; }
    }
    if (PAR) {
      // everything decoded so far must be in HBM: matches of the parallel path read the output buffer
      drain_to<RING>(o, o.pos);
      const uint32_t P0 = br.widx * 32u - (uint32_t)br.cnt - blk.comp_offset * 8u;  // bit position inside the member
      uint32_t end_bit = 0;
      const int why = uni(par_decode_block<RING>(comp + blk.comp_offset, P0, blk.comp_size * 8u, o, sl, &end_bit, stats));
      if (lane == 0) atomicAdd(&stats[why & 15], 1u);
      if (why == 0) {
        o.make_uniform();
        end_bit = uniu(end_bit);
        br.init(comp, blk.comp_offset + (end_bit >> 3), comp_end);
        br.drop((int)(end_bit & 7u));
        if (!last) {  // a serial block may follow: it finds the last RING bytes of output in the ring
          __builtin_amdgcn_s_waitcnt(0);
          const uint32_t lo = o.pos - o.begin > (uint32_t)RING ? o.pos - (uint32_t)RING : o.begin;
          for (uint32_t j = lo + (uint32_t)lane; j < o.pos; j += 64) ring[j & M] = out[j];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        continue;
      }
    }
    const SymResult r = decode_symbols<RING, WIDE>(br, o, vflav);
    br = r.br;
    br.make_uniform();
    o = r.o;
    o.make_uniform();
    err = uni(r.err);
  }
  if (err == INF_OK && (o.pos > o.end || br.overrun())) err = br.overrun() ? INF_INPUT_OVERRUN : INF_OUTPUT_OVERRUN;
  if (err != INF_OK) o.pos = min(o.pos, o.end);
  drain_to<RING>(o, o.pos);
  if (err == INF_OK && o.pos != o.end) err = INF_SIZE_MISMATCH;
  if (lane == 0) status[b] = err;
}

// Which symbol loop a wave runs: 0 scalar (symbol_run), 1 vector (symbol_run_v), 2 both on every CU -- workgroup b lands on
// XCD b % 8 and the XCD deals its workgroups over its 32 CUs, so bit 8 of b splits the waves of a CU, not the CUs
__device__ __forceinline__ int flavor_of(int flavor) { return flavor == 2 ? (int)((blockIdx.x >> 8) & 1u) : flavor; }

#ifdef EXON_INFLATE_WPE  // A/B builds: waves per SIMD the register allocation aims for (8 = 64 VGPRs)
#define EXON_INFLATE_WPE_ATTR __attribute__((amdgpu_waves_per_eu(EXON_INFLATE_WPE, EXON_INFLATE_WPE)))
#else
#define EXON_INFLATE_WPE_ATTR
#endif
template <int RING>
__global__ __launch_bounds__(64 * INF_WAVES, 8) EXON_INFLATE_WPE_ATTR __attribute__((amdgpu_num_sgpr(72))) void k_inflate(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks, int n_blocks,
                                                uint8_t* out, int* __restrict__ status, int flavor) {
  const int b = uni((int)(blockIdx.x * INF_WAVES + (threadIdx.x >> 6)));
  if (b >= n_blocks) return;
  inflate_member<RING, false>(comp, blocks, b, out, status, ParSlot{}, nullptr, flavor_of(flavor));
}


// The wide symbol loop (wide_run) as its own kernel: its register budget is not the serial loops'
template <int RING>
__global__ __launch_bounds__(64 * INF_WAVES, 8) __attribute__((amdgpu_num_sgpr(72))) void k_inflate_w(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks, int n_blocks, uint8_t* out,
                                                              int* __restrict__ status) {
  const int b = uni((int)(blockIdx.x * INF_WAVES + (threadIdx.x >> 6)));
  if (b >= n_blocks) return;
#ifdef EXON_WIDE_STATS
  const unsigned long long ws0 = __builtin_readcyclecounter();
#endif
  inflate_member<RING, false, true>(comp, blocks, b, out, status, ParSlot{}, nullptr, 1);
#ifdef EXON_WIDE_STATS
  if (lane_id() == 0) {
    atomicAdd(&g_wide_stats[1], __builtin_readcyclecounter() - ws0);
    atomicAdd(&g_wide_stats[8], 1ull);
  }
#endif
}

// The lane-parallel variant: a fixed set of workgroups (one scratch slot each) takes members off a shared counter.
template <int RING>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_inflate_par(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks, int n_blocks, uint8_t* out,
                                                    int* __restrict__ status, uint8_t* __restrict__ scratch, unsigned* __restrict__ counter,
                                                    unsigned* __restrict__ stats, int flavor) {
  const ParSlot sl = par_slot(scratch, blockIdx.x);
  const int vflav = flavor_of(flavor);
  for (;;) {
    unsigned b = 0;
    if (lane_id() == 0) b = atomicAdd(counter, 1u);
    b = uniu(b);
    if (b >= (unsigned)n_blocks) return;
    inflate_member<RING, true>(comp, blocks, (int)b, out, status, sl, stats, vflav);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// CRC-32 (IEEE, reflected) of every inflated block: one wavefront per block.  Each lane runs the byte-serial CRC
// over its own contiguous slice with a zero initial value; slices are then chained with the "append zeros" operator
// realised as a carry-less exponentiation -- here simply by processing the slices' CRCs through GF(2) matrix-free
// shifting: crc(A || B) = shift(crc(A), |B|) ^ crc0(B), where shift multiplies by x^(8|B|) mod P.
__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b) {  // a * b mod P, reflected representation
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) {
    if (a & 0x80000000u) r ^= b;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
  }
  return r;
}
__device__ __forceinline__ uint32_t x_pow_8n(uint32_t n_bytes) {  // x^(8 n) mod P
  uint32_t result = 0x80000000u;                                  // x^0
  uint32_t sq = 0x00800000u;                                      // x^8
  while (n_bytes) {
    if (n_bytes & 1u) result = gf2_mulmod(result, sq);
    sq = gf2_mulmod(sq, sq);
    n_bytes >>= 1;
  }
  return result;
}

// Powers of x mod P the combine needs, made once on the host (crc_powers): slices are `per` = 128 m bytes (m = 1..8) long, so the
// multiplier of lane l's CRC is x^(8 per k) x^(8 t) with k = whole slices behind it and t = the bytes of the last, partial one.
// (Computing it per lane by square-and-multiply -- ~40 multiplications of 32 steps, divergent, and lane 0's x^(8 n) on top -- was
// three times the table phase's vector instructions.)
struct CrcPowers {
  uint32_t xp[8][64];   // [m - 1][k]  = x^(8 * 128 m * k)
  uint32_t ixp[8][64];  // 0xFFFFFFFF times that: the initial value's term
  uint32_t xt[1025];    // [t] = x^(8 t)
};

__global__ __launch_bounds__(WAVES_PER_WG * 64) void k_crc32(const uint8_t* __restrict__ out, const Block* __restrict__ blocks, int n_blocks,
                                                             int* __restrict__ status, const CrcPowers* __restrict__ pw) {
  __shared__ uint32_t table[4][256];  // slicing-by-4: table[k][b] = CRC of byte b followed by k zero bytes
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
  const int b = blockIdx.x * WAVES_PER_WG + wave;
  if (b >= n_blocks) return;
  const Block blk = blocks[b];
  if (status[b] != INF_OK) return;
  const uint32_t n = blk.out_size;
  // whole 128-byte lines per slice (65280 bytes: 1024 per lane): a lane reads full cache lines -- with 64-byte steps half of every
  // 128-byte line fetched belonged to nobody's current step and was fetched again later
  const uint32_t per = (((n + 63u) / 64u) + 127u) & ~127u;
  const uint32_t lo = min(n, (uint32_t)lane * per), hi = min(n, lo + per);
  const uint8_t* p = out + blk.out_offset;
  uint32_t c = 0;  // raw CRC register over the slice, zero initial value, no final xor
  uint32_t i = lo;
  auto step4 = [&](uint32_t w) {
    c ^= w;
    c = table[3][c & 0xFFu] ^ table[2][(c >> 8) & 0xFFu] ^ table[1][(c >> 16) & 0xFFu] ^ table[0][c >> 24];
  };
  while (i < hi && ((reinterpret_cast<uintptr_t>(p + i)) & 15u)) c = table[0][(c ^ p[i++]) & 0xFFu] ^ (c >> 8);
  // the lanes' slices lie ~1 KiB apart, so a load instruction touches 64 different cache lines: take a whole 64-byte
  // line per lane per iteration (4 x 16 B), or the lines are evicted between the 16 dword loads that share them
  for (; i + 128 <= hi; i += 128) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const uint4*>(p + i + 16 * k);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      step4(v[k].x);
      step4(v[k].y);
      step4(v[k].z);
      step4(v[k].w);
    }
  }
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
  // combine: total = sum over lanes of crc_l * x^(8 * bytes after slice l); the 0xFFFFFFFF initial value is the
  // CRC of a virtual prefix: init * x^(8 n)
  uint32_t crc;
  if (pw && per >= 128u && per <= 1024u && n > 0) {
    // n = last * per + t: the lanes in front of the last non-empty one share the factor x^(8 t)
    const uint32_t m1 = (per >> 7) - 1u, last = (n - 1u) / per, t = n - last * per;
    uint32_t term = (uint32_t)lane < last ? gf2_mulmod(c, pw->xp[m1][last - 1u - (uint32_t)lane]) : 0u;
    if (lane == 0) term ^= pw->ixp[m1][last];
    for (int o = 32; o > 0; o >>= 1) term ^= __shfl_xor(term, o, 64);
    crc = gf2_mulmod(term, pw->xt[t]) ^ (uint32_t)__shfl((int)c, (int)last, 64) ^ 0xFFFFFFFFu;
  } else {
    uint32_t term = (hi > lo) ? gf2_mulmod(c, x_pow_8n(n - hi)) : 0u;
    if (lane == 0) term ^= gf2_mulmod(0xFFFFFFFFu, x_pow_8n(n));
    for (int o = 32; o > 0; o >>= 1) term ^= __shfl_xor(term, o, 64);
    crc = term ^ 0xFFFFFFFFu;
  }
  if (lane == 0 && crc != blk.crc32) status[b] = INF_BAD_CRC;
}

}  // namespace

// The header walk is a chain of dependent loads, one block header per ~18 KB of a buffer that other cores (the read pool) or nobody
// has touched: every step is a cache miss (0.12-0.24 us per block measured on the reader thread: 16 ms of a 100 M-row .vcf.gz).
// Memory-level parallelism instead: K walkers start at the first plausible header behind K evenly spaced offsets and advance round
// robin, each prefetching its next header while the others take their step.  They only WARM the lines the exact walk below then
// reads (their guesses are never trusted: a wrong one costs time, not correctness).  EXON_HIP_BGZF_WALK_WARM=0 turns it off.
static void bgzf_warm_headers(const uint8_t* data, size_t n) {
  constexpr int K = 8;
  if (n < (1u << 20)) return;
  static const bool on = [] {
    const char* v = getenv("EXON_HIP_BGZF_WALK_WARM");
    return !(v && v[0] == '0');
  }();
  if (!on) return;
  size_t pos[K], lim[K];
  const size_t last = n - 18;
  for (int j = 0; j < K; ++j) {
    size_t s = (size_t)j * (n / K);
    if (j > 0) {  // first canonical header (XLEN 6, one BC subfield) at or behind s, within two blocks' worth of bytes
      const size_t stop = std::min(last, s + (1u << 17));
      size_t found = n;
      while (s < stop) {
        const uint8_t* q = static_cast<const uint8_t*>(memchr(data + s, 0x1f, stop - s));
        if (!q) break;
        if (q[1] == 0x8b && q[2] == 8 && q[3] == 4 && q[12] == 'B' && q[13] == 'C' && q[14] == 2 && q[15] == 0) {
          found = (size_t)(q - data);
          break;
        }
        s = (size_t)(q - data) + 1;
      }
      s = found;
    }
    pos[j] = s;
  }
  // a walker stops where the next one that found a start began; one that found none idles
  size_t next_start = n;
  for (int j = K - 1; j >= 0; --j) {
    if (pos[j] >= n) {
      pos[j] = lim[j] = 0;
      continue;
    }
    lim[j] = next_start;
    next_start = pos[j];
  }
  bool active = true;
  while (active) {
    active = false;
    for (int j = 0; j < K; ++j) {
      const size_t o = pos[j];
      if (o >= lim[j] || o > last) continue;
      const uint8_t* h = data + o;
      if (h[0] != 0x1f || h[1] != 0x8b || h[12] != 'B' || h[13] != 'C' || (h[10] | (h[11] << 8)) != 6) {
        lim[j] = 0;  // not the canonical layout (or a wrong guess): the exact walk deals with it
        continue;
      }
      const size_t next = o + ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;
      pos[j] = next;
      if (next <= last) {
        __builtin_prefetch(data + next - 8);  // CRC32 + ISIZE of this block
        __builtin_prefetch(data + next + 17);  // header of the next
      }
      active = true;
    }
  }
}

extern "C" {

int exon_hip_bgzf_scan(const uint8_t* data, size_t n, size_t out_base, exon_hip_bgzf_block* blocks, int32_t cap,
                       int32_t* n_blocks, size_t* consumed, size_t* out_bytes) {
  if (!n_blocks || !consumed || !out_bytes || (n && !data)) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_bgzf_scan: NULL argument");
  size_t o = 0, out = out_base;
  int32_t k = 0;
  if (cap >= 64) bgzf_warm_headers(data, n);
  while (o + 18 <= n) {
    const uint8_t* h = data + o;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail(nullptr, EXON_HIP_EINVAL, "not a BGZF block at byte %zu", o);
    const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
    if (o + 12 + xlen > n) break;
    size_t bsize = 0;
    for (size_t x = 12; x + 4 <= 12 + xlen;) {  // extra subfields: SI1 SI2 SLEN data
      const size_t slen = (size_t)h[x + 2] | ((size_t)h[x + 3] << 8);
      if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = ((size_t)h[x + 4] | ((size_t)h[x + 5] << 8)) + 1;
      x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8) return fail(nullptr, EXON_HIP_EINVAL, "BGZF block at byte %zu has no valid BC field", o);
    if (o + bsize > n) break;  // partial block: the caller reads more
    if (k == cap) break;
    uint32_t crc, isize;
    memcpy(&crc, h + bsize - 8, 4);
    memcpy(&isize, h + bsize - 4, 4);
    if (isize > 65536u) return fail(nullptr, EXON_HIP_EINVAL, "BGZF block at byte %zu claims %u bytes", o, isize);
    if (out + isize > 0xFFFFFFFFull || o + bsize > 0xFFFFFFFFull) break;  // 32-bit offsets per call
    if (blocks) {
      blocks[k].comp_offset = (uint32_t)(o + 12 + xlen);
      blocks[k].comp_size = (uint32_t)(bsize - 12 - xlen - 8);
      blocks[k].out_offset = (uint32_t)out;
      blocks[k].out_size = isize;
      blocks[k].crc32 = crc;
      blocks[k].reserved = 0;
    }
    ++k;
    out += isize;
    o += bsize;
  }
  *n_blocks = k;
  *consumed = o;
  *out_bytes = out - out_base;
  return EXON_HIP_OK;
}

int exon_hip_bgzf_inflate(exon_hip_ctx* ctx, void* stream, const uint8_t* d_comp, const exon_hip_bgzf_block* blocks,
                          int32_t n_blocks, uint8_t* d_out, int32_t verify_crc, int32_t* first_bad_block) {
  if (!ctx || (n_blocks > 0 && (!d_comp || !blocks || !d_out))) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_bgzf_inflate: NULL argument");
  if (first_bad_block) *first_bad_block = -1;
  if (n_blocks <= 0) return EXON_HIP_OK;
  if ((reinterpret_cast<uintptr_t>(d_comp) & 3) != 0 || (reinterpret_cast<uintptr_t>(d_out) & 3) != 0)
    return fail(ctx, EXON_HIP_EINVAL, "compressed and output buffers must be 4-byte aligned");
  hipStream_t s = pick_stream(ctx, stream);
  hipSetDevice(ctx->device);
  // block table + status words: one scratch allocation per call (freed after the synchronise below)
  Block* d_blocks = nullptr;
  const size_t tb = (size_t)n_blocks * sizeof(Block), sb = (size_t)n_blocks * sizeof(int);
  if (hipMalloc((void**)&d_blocks, tb + sb) != hipSuccess) return fail(ctx, EXON_HIP_ENOMEM, "block table allocation failed");
  int* d_status = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(d_blocks) + tb);
  std::vector<int> h_status((size_t)n_blocks);
  hipError_t e = hipMemcpyAsync(d_blocks, blocks, tb, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = exon_bgzf_inflate_launch(s, d_comp, reinterpret_cast<const exon_hip_bgzf_block*>(d_blocks), n_blocks, d_out, d_status, verify_crc != 0);
  if (e == hipSuccess) e = hipMemcpyAsync(h_status.data(), d_status, sb, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(d_blocks);
  if (e != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "bgzf inflate: %s", hipGetErrorString(e));
  for (int32_t b = 0; b < n_blocks; ++b)
    if (h_status[(size_t)b] != INF_OK) {
      if (first_bad_block) *first_bad_block = b;
      return fail(ctx, EXON_HIP_EINVAL, "BGZF block %d: %s", b, exon_bgzf_status_name(h_status[(size_t)b]));
    }
  return EXON_HIP_OK;
}

}  // extern "C"

// Internal (C++): enqueue the inflate (+ CRC) kernels; d_blocks / d_status are device arrays of n_blocks entries.
namespace {
// Which decoder a launch uses (EXON_HIP_INFLATE_PAR):
//   unset  auto: the lane-parallel decoder for launches of at most PAR_AUTO_MAX members, the serial one above that.  A
//          member alone takes ~2.5 ms through the serial symbol loop and ~0.5 ms through the parallel one, so small files,
//          index chunks and first slabs finish 3x sooner; a machine-full of members (a pipeline slab) is the serial
//          kernel's best case, and far above that the parallel one wins again by 1.26x (profiles/r2_tuning.md)
//          Round 4 (profiles/r4_inflate_par_crossover.log, one launch of K members through either decoder): VCF text 0.9 ms
//          against 2.9 ms up to 256 members (one per CU) and 2.97-3.2 against 2.9-3.0 ms from 512 on; BAM and FASTQ members
//          (mostly handed back to the serial loop inside the parallel kernel) 10 % slower at every size.  The limit was 1536.
//   0      serial always      1  lane-parallel always      2  both side by side (a share of the members each)
constexpr int PAR_AUTO_MAX = 256;
int par_mode() {
  static const int m = [] {
    const char* e = getenv("EXON_HIP_INFLATE_PAR");
    return e && *e ? atoi(e) : -1;
  }();
  return m;
}
double par_serial_share() {  // hybrid mode: fraction of a launch's members given to the serial kernel
  static const double r = [] {
    const char* e = getenv("EXON_HIP_INFLATE_PAR_SERIAL_SHARE");
    const double v = e ? atof(e) : 0.4;
    return v < 0 ? 0.0 : v > 0.95 ? 0.95 : v;
  }();
  return r;
}
int par_slots() {  // resident workgroups of the parallel kernel = scratch slots (16 per CU on 256 CUs)
  static const int n = [] {
    const char* e = getenv("EXON_HIP_INFLATE_PAR_SLOTS");
    const int v = e ? atoi(e) : 4096;
    return v < 64 ? 64 : v > 16384 ? 16384 : v;
  }();
  return n;
}
// per device: outcome counters of par_decode_block (64 words, never freed) and, for the hybrid mode, a side stream per
// caller stream with its fork / join events
struct ParDev {
  unsigned* stats = nullptr;
};
struct ParSide {
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
// scratch of the parallel kernel: 256 bytes (work counter) + one slot per workgroup, per caller stream (two scans inflating
// at the same time must not share slots).  Grows to the largest launch seen on the stream, is never freed otherwise.
// (hipMallocAsync / hipFreeAsync per launch would be the natural fit; on ROCm 7.0 the file pipeline then stalls with an idle
// GPU after its second slab -- the inflate stream never runs again -- so the pools are managed here.)
struct ParPool {
  uint8_t* mem = nullptr;
  int slots = 0;
};
std::map<std::pair<int, hipStream_t>, ParPool> g_par_pools;
std::mutex g_par_mu;
std::map<int, ParDev> g_par_dev;
std::map<std::pair<int, hipStream_t>, ParSide> g_par_side;
unsigned* par_stats_buffer(int dev) {
  std::lock_guard<std::mutex> g(g_par_mu);
  ParDev& d = g_par_dev[dev];
  if (!d.stats) {
    if (hipMalloc((void**)&d.stats, 64 * sizeof(unsigned)) != hipSuccess || hipMemset(d.stats, 0, 64 * sizeof(unsigned)) != hipSuccess) {
      (void)hipGetLastError();
      d.stats = nullptr;
    }
  }
  return d.stats;
}
uint8_t* par_pool(int dev, hipStream_t s, int want) {
  std::lock_guard<std::mutex> g(g_par_mu);
  auto key = std::make_pair(dev, s);
  if (!g_par_pools.count(key)) {  // at most 8 pools PER DEVICE (a 9th stream on a device inflates serially)
    int on_dev = 0;
    for (const auto& kv : g_par_pools) on_dev += kv.first.first == dev;
    if (on_dev >= 8) return nullptr;
  }
  ParPool& p = g_par_pools[key];
  if (p.slots < want) {
    // sized to the launch (rounded to 64 slots), not to the next power of two: a slot is ~0.94 MiB, so the AUTO mode's
    // largest launch (1536 members) keeps ~1.4 GB per inflating stream instead of 2 GB, EXON_HIP_INFLATE_PAR=1's 4096
    // slots 3.8 GB.  The pool stays with the stream until exon_hip_bgzf_forget_stream / the owner's destructor.
    const int n = (want + 63) / 64 * 64;
    if (p.mem) {  // kernels queued on `s` may still use the old block
      if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
      hipFree(p.mem);
      p.mem = nullptr;
      p.slots = 0;
    }
    if (hipMalloc((void**)&p.mem, 256 + (size_t)n * PAR_SLOT_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      p.mem = nullptr;
      return nullptr;
    }
    p.slots = n;
  }
  return p.mem;
}
bool par_side(int dev, hipStream_t s, ParSide* out) {
  std::lock_guard<std::mutex> g(g_par_mu);
  auto key = std::make_pair(dev, s);
  if (!g_par_side.count(key)) {
    int on_dev = 0;
    for (const auto& kv : g_par_side) on_dev += kv.first.first == dev;
    if (on_dev >= 8) return false;
  }
  ParSide& p = g_par_side[key];
  if (!p.side) {
    if (hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&p.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p.ev_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      g_par_side.erase(key);
      return false;
    }
  }
  *out = p;
  return true;
}
}  // namespace

// outcome counters of the lane-parallel path on the current device since the process started (index 0 = blocks decoded in
// parallel, 1..15 = blocks handed back to the serial loop, by reason); `stream` is ignored (kept for the ABI)
extern "C" int exon_hip_bgzf_inflate_par_stats(void* stream, uint32_t* out32) {
  (void)stream;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || !out32) return -1;
  memset(out32, 0, 32 * sizeof(uint32_t));
  unsigned* st = nullptr;
  {
    std::lock_guard<std::mutex> g(g_par_mu);
    auto it = g_par_dev.find(dev);
    if (it != g_par_dev.end()) st = it->second.stats;
  }
#ifdef EXON_WIDE_STATS  // the wide loop's clock split instead: 16 x u64 in the 32 words, then zeroed
  {
    unsigned long long w[16] = {0}, z[16] = {0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(w, HIP_SYMBOL(g_wide_stats), sizeof w) != hipSuccess) return -1;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wide_stats), z, sizeof z);
    memcpy(out32, w, sizeof w);
    return 0;
  }
#endif
  if (!st) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpy(out32, st, 32 * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

void exon_bgzf_forget_stream(hipStream_t s) {
  if (!s) return;
  std::lock_guard<std::mutex> g(g_par_mu);
  // keyed on the STREAM, whatever device is current here: a stream belongs to exactly one device
  for (auto it = g_par_pools.begin(); it != g_par_pools.end();) {
    if (it->first.second == s) {
      if (it->second.mem) hipFree(it->second.mem);
      it = g_par_pools.erase(it);
    } else {
      ++it;
    }
  }
  for (auto sd = g_par_side.begin(); sd != g_par_side.end();) {
    if (sd->first.second == s) {
      if (sd->second.side) hipStreamDestroy(sd->second.side);
      if (sd->second.ev_fork) hipEventDestroy(sd->second.ev_fork);
      if (sd->second.ev_join) hipEventDestroy(sd->second.ev_join);
      sd = g_par_side.erase(sd);
    } else {
      ++sd;
    }
  }
}

// public form for caller-owned streams handed to exon_hip_bgzf_inflate: releases the scratch kept for `stream` (idle stream)
extern "C" int exon_hip_bgzf_forget_stream(void* stream) {
  exon_bgzf_forget_stream((hipStream_t)stream);
  return EXON_HIP_OK;
}

// Which symbol loop a launch runs: 3 the wide loop (wide_run: 64 bit offsets per round, round 5), 1 the software-pipelined
// vector-unit loop with deferred far copies (symbol_run_v, round 4), 0 round 1's scalar loop (symbol_run), 2 loops 0 and 1 side by
// side.  EXON_HIP_INFLATE_FLAVOR forces one; otherwise the caller's hint, otherwise 3.  One resident launch, same box
// (profiles/r5_inflate_wide_v3_32waves.log): VCF text 113 -> 178 GB/s, BAM payloads 102 -> 127, FASTQ 81 -> 82; the file pipelines
// .vcf.gz 59 -> 41 ms, BAM 59 -> 48-51 ms, .fastq.gz 119 -> 115-120 ms.  (Round 4's history of loops 0 / 1 / 2: HISTORY.md section 7e.)
static int inflate_flavor(int hint) {
  static const int forced = [] {
    const char* e = getenv("EXON_HIP_INFLATE_FLAVOR");
    return e && e[0] >= '0' && e[0] <= '3' ? e[0] - '0' : -1;
  }();
  return forced >= 0 ? forced : hint >= 0 && hint <= 3 ? hint : 3;
}

// The CRC combine's powers of x: computed once per process, one copy per device (never freed).
namespace {
uint32_t gf2_mulmod_host(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) {
    if (a & 0x80000000u) r ^= b;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
  }
  return r;
}
const CrcPowers* crc_powers(int dev) {
  static const CrcPowers* host = [] {
    CrcPowers* t = new CrcPowers;
    t->xt[0] = 0x80000000u;
    for (int i = 1; i <= 1024; ++i) t->xt[i] = gf2_mulmod_host(t->xt[i - 1], 0x00800000u);
    for (int m = 1; m <= 8; ++m) {
      const uint32_t X = t->xt[128 * m];
      t->xp[m - 1][0] = 0x80000000u;
      for (int k = 1; k < 64; ++k) t->xp[m - 1][k] = gf2_mulmod_host(t->xp[m - 1][k - 1], X);
      for (int k = 0; k < 64; ++k) t->ixp[m - 1][k] = gf2_mulmod_host(0xFFFFFFFFu, t->xp[m - 1][k]);
    }
    return t;
  }();
  static std::mutex mu;
  static std::map<int, const CrcPowers*> per_dev;
  std::lock_guard<std::mutex> g(mu);
  auto it = per_dev.find(dev);
  if (it != per_dev.end()) return it->second;
  CrcPowers* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(CrcPowers)) != hipSuccess || hipMemcpy(d, host, sizeof(CrcPowers), hipMemcpyHostToDevice) != hipSuccess) {
    if (d) hipFree(d);
    (void)hipGetLastError();
    d = nullptr;
  }
  per_dev[dev] = d;
  return d;
}
}  // namespace

hipError_t exon_bgzf_inflate_launch(hipStream_t s, const uint8_t* d_comp, const exon_hip_bgzf_block* d_blocks, int n_blocks,
                                    uint8_t* d_out, int* d_status, bool verify_crc, int flavor_hint, bool text_like) {
  if (n_blocks <= 0) return hipSuccess;
  const int flavor = inflate_flavor(flavor_hint);
  static_assert(sizeof(Block) == sizeof(exon_hip_bgzf_block), "block layouts must agree");
  const Block* blocks = reinterpret_cast<const Block*>(d_blocks);
  const int mode = par_mode();
  // (auto: small launches of TEXT only -- members of BAM / BCF records and FASTQ reads are literal-heavy, the lane-parallel decoder
  //  hands most of them back to a serial loop: 50 k BAM records 3.9 ms through it against 2.1 ms through the wide loop alone,
  //  where 100 k rows of VCF text take 0.93 against 1.32 ms: profiles/r5_inflate_small_launches.log)
  bool parallel = mode >= 1 || (mode < 0 && n_blocks <= PAR_AUTO_MAX && text_like);
  int dev = 0;
  unsigned* stats = nullptr;
  uint8_t* scratch = nullptr;
  int n_ser = 0, n_wg = 0;
  if (parallel) {
    // scratch: one slot per workgroup + the work counter; anything that fails here just means "decode serially"
    parallel = hipGetDevice(&dev) == hipSuccess && (stats = par_stats_buffer(dev)) != nullptr;
    ParSide side;
    if (parallel && mode == 2 && n_blocks >= 256 && par_side(dev, s, &side)) n_ser = (int)((double)n_blocks * par_serial_share());
    n_wg = std::min(n_blocks - n_ser, par_slots());
    if (parallel && !(scratch = par_pool(dev, s, n_wg))) parallel = false;
    if (parallel) {
      hipError_t e;
      if ((e = hipMemsetAsync(scratch, 0, 256, s)) != hipSuccess) return e;
      if (n_ser > 0) {
        if ((e = hipEventRecord(side.ev_fork, s)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(side.side, side.ev_fork, 0)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_inflate<INFLATE_RING_SERIAL>, dim3(n_ser), dim3(64), 0, side.side, d_comp, blocks, n_ser, d_out, d_status, flavor);
        if ((e = hipEventRecord(side.ev_join, side.side)) != hipSuccess) return e;
      }
      hipLaunchKernelGGL(k_inflate_par<INFLATE_RING>, dim3(n_wg), dim3(64), 0, s, d_comp, blocks + n_ser, n_blocks - n_ser, d_out, d_status + n_ser,
                         scratch + 256, reinterpret_cast<unsigned*>(scratch), stats, flavor);
      if (n_ser > 0 && (e = hipStreamWaitEvent(s, side.ev_join, 0)) != hipSuccess) return e;
    }
  }
  // (experiment knob: EXON_HIP_INFLATE_PAD_LDS=<bytes> of unused dynamic LDS per workgroup = fewer resident waves per CU)
  static const size_t pad_lds = [] {
    const char* e = getenv("EXON_HIP_INFLATE_PAD_LDS");
    const long v = e ? atol(e) : 0;
    return (size_t)(v > 0 && v <= 32768 ? v : 0);
  }();
  if (!parallel && flavor == 3)
    hipLaunchKernelGGL(k_inflate_w<INFLATE_RING_SERIAL>, dim3((n_blocks + INF_WAVES - 1) / INF_WAVES), dim3(64 * INF_WAVES), pad_lds, s, d_comp, blocks,
                       n_blocks, d_out, d_status);
  else if (!parallel)
    hipLaunchKernelGGL(k_inflate<INFLATE_RING_SERIAL>, dim3((n_blocks + INF_WAVES - 1) / INF_WAVES), dim3(64 * INF_WAVES), pad_lds, s, d_comp, blocks, n_blocks,
                       d_out, d_status, flavor);
  if (verify_crc) {
    int cdev = 0;
    const CrcPowers* pw = hipGetDevice(&cdev) == hipSuccess ? crc_powers(cdev) : nullptr;  // (nullptr: the kernel computes the powers itself)
    hipLaunchKernelGGL(k_crc32, dim3((n_blocks + WAVES_PER_WG - 1) / WAVES_PER_WG), dim3(WAVES_PER_WG * 64), 0, s, d_out, blocks, n_blocks,
                       d_status, pw);
  }
  return hipGetLastError();
}

// resident workgroups per CU of the serial kernel, as the runtime computes it (tools / DESIGN only)
extern "C" int exon_hip_debug_inflate_occupancy(int dynamic_lds_bytes) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inflate<INFLATE_RING_SERIAL>, 64, (size_t)dynamic_lds_bytes) != hipSuccess) return -1;
  return nb;
}

const char* exon_bgzf_status_name(int code) {
  static const char* const names[] = {"ok", "reserved block type", "stored-block length check", "invalid code lengths",
                                      "invalid Huffman code", "distance before the block start", "more output than ISIZE",
                                      "compressed data exhausted", "less output than ISIZE", "CRC-32 mismatch"};
  return code >= 0 && code < 10 ? names[code] : "unknown";
}
