// chain_walk.h -- splitting a slab of length-prefixed binary records (BAM, BCF) in HBM, in parallel.
//
// The records form a chain (each starts where the previous one ends), so the slab is cut into 64 KiB segments and the
// chains are walked by one wavefront per segment:
//   k_chain_walk<F>  segment 0 starts at byte 0 (the caller guarantees a record boundary); every other segment GUESSES its
//                    first record start (smallest offset where three consecutive record headers look plausible), then walks
//                    its chain, storing record offsets, until it leaves the segment -> (start, landing, count)
//   k_chain_check    the guesses are PROVEN by induction: the chain that leaves segment s must land exactly on the guessed
//                    start of the segment it lands in (segments it jumps over -- records longer than a segment -- are
//                    ignored).  Any mismatch or a malformed record -> undecided: the caller decodes on the host instead.
//                    Exclusive scan of the counts -> first row of each segment.
// A format F provides: MIN_HEADER (bytes a plausibility check needs), MIN_RECORD (smallest record), record_bytes(d, r)
// (total size of the record at r, 0 = malformed) and plausible(d, n, r).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chain {

constexpr uint32_t SEG = 65536;  // segment size in bytes
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);  // unaligned dword load
  return v;
}
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

struct SegInfo {
  uint32_t start;    // first record start inside the segment (NONE = none found)
  uint32_t landing;  // where the chain leaves the segment (start of the first record not walked)
  uint32_t count;    // records walked
  uint32_t bad;      // 1 = malformed record met, 2 = stopped at a record that the slab cuts off
};

template <class F>
constexpr uint32_t seg_cap() { return SEG / F::MIN_RECORD + 2; }  // most records that can start inside one segment

template <class F>
__global__ __launch_bounds__(64) void k_chain_walk(const uint8_t* __restrict__ d, uint32_t n, F fmt, SegInfo* __restrict__ seg,
                                                   uint32_t* __restrict__ rec_off, unsigned* __restrict__ scalars) {
  const uint32_t s = blockIdx.x, lane = threadIdx.x;
  // the slab's scalars start at zero: nothing in this kernel touches them, the kernels behind it (stream order) add
  if (s == 0 && lane < 4) scalars[lane] = 0;
  const uint32_t lo = s * SEG, hi = min(n, lo + SEG);
  uint32_t start = NONE;
  if (s == 0) {
    start = 0;
  } else {
    for (uint32_t c0 = lo; c0 < hi && start == NONE; c0 += 64) {
      const uint32_t c = c0 + lane;
      bool ok = c < hi && fmt.plausible(d, n, c);
      if (ok) {  // two more records down the chain (fewer when the slab ends first)
        uint32_t r = c;
        for (int k = 0; k < 2 && ok; ++k) {
          r += fmt.record_bytes(d, r);
          if ((uint64_t)r + F::MIN_HEADER > n) break;
          ok = fmt.plausible(d, n, r);
        }
      }
      const unsigned long long m = __ballot(ok);
      if (m) start = c0 + (uint32_t)__ffsll((long long)m) - 1;
    }
  }
  uint32_t r = start, k = 0, bad = 0;
  if (start != NONE) {
    uint32_t* out = rec_off + (size_t)s * seg_cap<F>();
    while (r < hi) {
      if ((uint64_t)r + F::LEN_BYTES > n) { bad = 2; break; }
      const uint32_t sz = (uint32_t)__builtin_amdgcn_readfirstlane((int)fmt.record_bytes(d, r));
      if (sz == 0) { bad = 1; break; }
      if ((uint64_t)r + sz > n) { bad = 2; break; }  // record cut off by the end of the slab: carried by the caller
      if (lane == 0) out[k] = r;
      ++k;
      r += sz;
    }
    // the next record's header is not fully inside the slab: whatever follows cannot be recognised by the segments
    // behind this one, and does not need to be -- it is the cut-off tail
    if (bad == 0 && (uint64_t)r + F::MIN_HEADER > n) bad = 2;
  }
  if (lane == 0) seg[s] = SegInfo{start, r, k, bad};
}

// buffers k_chain_check clears before the extract kernel ORs validity bits into them (round 3: these were three to five
// hipMemsetAsync calls per slab, each a tiny kernel queued behind whatever else runs on the chip)
struct ZeroList {
  uint32_t* p[2 + 16];
  int n;
  uint32_t words;  // 32-bit words to clear in each
};

// scalars: [0] rows, [1] undecided, [2] consumed bytes.
// The proof: the chain that enters segment s at `expected` must find start(s) == expected; a record longer than a segment makes
// the chain skip whole segments, which are then ignored (whatever their guess was); the segment whose chain met the cut-off
// record ends the slab.
//   * The common slab needs no walk: every segment's chain lands on the guessed start of the NEXT segment (no record spans a
//     whole segment, nothing malformed, only the last segment may meet the cut-off record) -- one comparison per segment, all at
//     once, straight from the segment records in HBM.
//   * Otherwise one thread walks the segments in order, eight records in flight (the chain nearly always advances by one segment,
//     so the loads do not depend on the walk) and marks the segments it jumped over.
// One workgroup of 256 threads and no dynamic LDS (round 4: a 1024-thread workgroup needs sixteen free wave slots on ONE CU): until then the kernel staged 16 bytes per segment in LDS -- 107 KB for a 6720-segment slab -- and its
// one workgroup could not start on any CU before enough inflate waves of the next slab had drained there: the "1.3-1.9 ms" rocprofv3
// showed for it in the BAM pipeline were that wait, a quarter of the parse stream.
template <int UNUSED = 0>  // a template only so that several translation units may include this header
__global__ __launch_bounds__(256) void k_chain_check(SegInfo* __restrict__ seg, uint32_t n_seg, uint32_t* __restrict__ base,
                                                      unsigned* __restrict__ scalars, const ZeroList zl) {
  __shared__ unsigned part[256];
  __shared__ unsigned s_last, s_err, s_plain;
  if (threadIdx.x == 0) s_plain = 1;
  for (int b = 0; b < zl.n; ++b)
    for (uint32_t i = threadIdx.x; i < zl.words; i += 256) zl.p[b][i] = 0;
  __syncthreads();
  const uint4* seg4 = reinterpret_cast<const uint4*>(seg);  // {start, landing, count, bad}
  // the segment whose chain met the cut-off record ends the slab (usually the last one, sometimes the one before: the record
  // that the slab cuts off may start there): L = the first such segment, the proof covers segments 0 .. L
  __shared__ unsigned s_cut;
  if (threadIdx.x == 0) s_cut = n_seg - 1;
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < n_seg; s += 256)
    if (seg[s].bad == 2) atomicMin(&s_cut, s);
  __syncthreads();
  {
    const uint32_t L = s_cut;
    bool ok = true;
    for (uint32_t s = threadIdx.x; s <= L; s += 256) {
      const uint4 cur = seg4[s];
      if (s == 0 && cur.x != 0) ok = false;
      if (cur.x == NONE) ok = false;
      if (s < L) {
        if (cur.w != 0 || seg[s + 1].start != cur.y) ok = false;
      } else if (cur.w == 1) {
        ok = false;
      }
    }
#ifdef EXON_CHAIN_SERIAL_CHECK  // A/B builds: always the serial proof
    ok = false;
#endif
    if (!ok) s_plain = 0;  // (plain store of the same value from any number of threads)
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_plain) {
    s_last = s_cut;
    s_err = 0;
  }
  if (threadIdx.x == 0 && !s_plain) {
    unsigned err = 0;
    uint32_t expected = 0, last = n_seg - 1;
    bool done = false;
    for (uint32_t s0 = 0; s0 < n_seg && !done; s0 += 8) {
      uint4 e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = s0 + k < n_seg ? seg4[s0 + k] : uint4{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t s = s0 + k;
        if (done || s >= n_seg) break;
        const uint32_t hi = (s + 1) * SEG;  // the last segment is shorter, but no chain position lies beyond the slab
        if (expected >= hi) {               // covered by a record that started earlier: not part of the chain
          seg[s].count = 0;
          continue;
        }
        if (e[k].x != expected || e[k].w == 1) {
          err = 1;
          done = true;
          break;
        }
        expected = e[k].y;
        last = s;
        if (e[k].w == 2) done = true;  // the cut-off record: everything behind it is its bytes
      }
    }
    s_last = last;
    s_err = err;
  }
  __syncthreads();
  const uint32_t last = s_last;
  const uint32_t per = (n_seg + 255) / 256;
  const uint32_t s0 = threadIdx.x * per, s1 = min(n_seg, s0 + per);
  unsigned sum = 0;
  for (uint32_t s = s0; s < s1; ++s) {
    if (s > last) seg[s].count = 0;
    else sum += seg[s].count;
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    unsigned v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (uint32_t s = s0; s < s1; ++s) {
    base[s] = run;
    run += seg[s].count;
  }
  if (threadIdx.x == 255) scalars[0] = part[255];
  if (threadIdx.x == 0) {
    scalars[2] = seg[last].landing;
    scalars[3] = s_plain;  // diagnostics: 1 = the parallel proof sufficed
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_err) atomicAdd(&scalars[1], 1u);
}

}  // namespace chain
