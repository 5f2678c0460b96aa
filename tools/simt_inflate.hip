// simt_inflate.hip -- RESEARCH SPIKE (round 2), not part of the product: BGZF inflate with one block per LANE (64 blocks per
// wavefront).  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iexon_amd/csrc tools/simt_inflate.hip -o
// tools/bin/libsimt_inflate.so ; driver: tools/time_simt_inflate.py.  Measured (profiles/r2_tuning.md): byte-identical to zlib on
// 2.2 GB of VCF text, but 25 GB/s against 92 GB/s for inflate.hip on the same 33 k blocks; without the match copies (tokenizer
// + literal stores only) 144 GB/s.  What it was meant to be:
//
// inflate.hip decodes one block per wavefront with wave-uniform (scalar) code: every symbol costs ~20-60 scalar instructions
// and a CU has ONE scalar unit, so the whole chip tops out near 80 GB/s of VCF text (profiles/r1_tuning.md) while the vector
// pipes idle.  This kernel turns the work sideways: every lane runs a plain serial inflate of its own block, so one vector
// instruction advances 64 blocks.  What makes that work where the round-1 spike (41 GB/s) did not:
//   * the first-level Huffman tables of every lane live in LDS (9-bit literal/length root + 6-bit distance root, u16 entries
//     = 1216 B per lane, 76 KiB per wavefront, two wavefronts per CU): a symbol is one ds_read, not a global load;
//   * codes longer than the root (rare symbols: ~0.5 % of a text block) are decoded bit-serially from per-length counts (LDS)
//     and the canonical symbol order (per-lane global scratch), in place;
//   * matches are copied 8 / 4 / 1 bytes per step by a per-lane state machine instead of a byte loop whose trip count is the
//     longest match of the wave;
//   * the input is prefetched one word ahead (the bit reservoir never waits for memory), table construction -- a long
//     divergent routine -- is batched: lanes at a block header wait until eight of them are there (or nothing else can run).
// Why it loses: each step of a wavefront holds a global load -> store round trip (some lane is always copying a match whose
// source is output it wrote earlier: ~2 us with two wavefronts per CU and nothing to hide it behind), and even the tokenizer
// alone costs ~1700 clocks per step (two LDS lookups, 64-bit shifts, a global-scratch lookup whenever ANY of 64 lanes meets a
// long code).
// The kernel is an ACCELERATOR, not an authority: a lane that meets anything unusual (an invalid or incomplete code, a
// distance before the block start, a size mismatch, input or output overrun, an iteration cap) stores INF_RETRY and leaves;
// exon_bgzf_inflate_launch then runs inflate.hip's kernel on exactly those blocks, which reports the definitive status.
// Output is therefore byte-identical to zlib by the same tests as before (tests/test_gpu_inflate.py runs every case through
// both kernels).  RFC 1951; replaces noodles' bgzf reader (exon-core/src/streaming_bgzf.rs:56-64 and the file openers).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "../include/exon_hip.h"

namespace {

struct Block {
  uint32_t comp_offset, comp_size, out_offset, out_size, crc32, reserved;
};
static_assert(sizeof(Block) == sizeof(exon_hip_bgzf_block), "block layouts must agree");

constexpr int INF_OK = 0, INF_RETRY = 100;
constexpr int LROOT = 9, DROOT = 6;
constexpr int LANE_LDS = (2 << LROOT) + (2 << DROOT) + 32 + 32;  // lit root, dist root, lit counts, dist counts (u16 each)
constexpr int LANE_SCRATCH = 1024;                                // global: lens[320] u8, sorted lit[288] u16, sorted dist[32] u16
constexpr unsigned MAX_STEPS = 600000;                            // > 8 steps per output byte of a 64 KiB block: a safety net

enum State : int { ST_HEADER = 0, ST_DECODE = 1, ST_COPY = 2, ST_STORED = 3, ST_DONE = 4 };

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

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ void st_u32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void st_u64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

// per-lane bit reservoir with one word of prefetch: `next` is the dword at comp[ip - 4 .. ip) already in a register
struct Bits {
  uint64_t buf;
  int cnt;
  uint32_t ip;    // byte offset of the next dword to prefetch
  uint32_t next;  // prefetched dword (the one the next refill appends)
  __device__ __forceinline__ void refill(const uint8_t* comp) {  // guarantees cnt >= 32 afterwards
    if (cnt <= 32) {
      buf |= (uint64_t)next << cnt;
      cnt += 32;
      next = ld_u32(comp + ip);  // its latency is hidden until the NEXT refill
      ip += 4;
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
  // byte offset in comp of the first byte not yet consumed (only meaningful when cnt is a multiple of 8)
  __device__ __forceinline__ uint32_t byte_pos() const { return ip - 4u - (uint32_t)(cnt >> 3); }
};

// Builds one canonical code from lens[0, n): root table (u16 entries: [3:0] code length, 15 = longer than the root, 0 = no such
// code; [15:4] symbol), per-length counts (LDS), symbols in canonical order (scratch).  false: over-subscribed or incomplete
// (the caller retries the block with the other kernel, which knows which incomplete codes RFC 1951 tolerates).
__device__ bool build_code(const uint8_t* lens, int n, int root, uint16_t* table, uint16_t* count, uint16_t* sorted) {
  for (int l = 0; l < 16; ++l) count[l] = 0;
  for (int s = 0; s < n; ++s) count[lens[s]]++;
  count[0] = 0;
  int left = 1;
  uint32_t next[16], offs[16];
  uint32_t code = 0, o = 0;
  for (int l = 1; l <= 15; ++l) {
    left = (left << 1) - (int)count[l];
    if (left < 0) return false;
    code = (code + (l > 1 ? count[l - 1] : 0)) << 1;
    next[l] = code;
    offs[l] = o;
    o += count[l];
  }
  if (left != 0) return false;  // incomplete code
  const int size = 1 << root;
  for (int i = 0; i < size; ++i) table[i] = 0;
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (l == 0) continue;
    const uint32_t c = next[l]++;
    sorted[offs[l]++] = (uint16_t)s;
    const uint32_t rev = __brev(c) >> (32 - l);
    if (l <= root) {
      const uint16_t e = (uint16_t)((s << 4) | l);
      for (uint32_t j = rev; j < (uint32_t)size; j += 1u << l) table[j] = e;
    } else {
      table[rev & (uint32_t)(size - 1)] = 15;  // slow marker
    }
  }
  return true;
}

// canonical bit-serial decode (codes of any length); -1: no such code
__device__ int decode_slow(Bits& b, const uint16_t* count, const uint16_t* sorted) {
  int code = 0, first = 0, index = 0;
  uint32_t v = (uint32_t)b.buf;
  for (int len = 1; len <= 15; ++len) {
    code |= (int)(v & 1u);
    v >>= 1;
    const int n = count[len];
    if (code - n < first) {
      b.drop(len);
      return sorted[index + (code - first)];
    }
    index += n;
    first += n;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

template <bool NOCOPY>
__global__ __launch_bounds__(64) void k_inflate_simt(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks, int n_blocks,
                                                     uint8_t* __restrict__ out, uint8_t* __restrict__ scratch, int* __restrict__ status) {
  extern __shared__ uint8_t lds[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  uint16_t* lit = reinterpret_cast<uint16_t*>(lds + lane * LANE_LDS);
  uint16_t* dst = lit + (1 << LROOT);
  uint16_t* lcount = dst + (1 << DROOT);
  uint16_t* dcount = lcount + 16;
  uint8_t* lens = scratch + (size_t)b * LANE_SCRATCH;
  uint16_t* lsorted = reinterpret_cast<uint16_t*>(lens + 320);
  uint16_t* dsorted = lsorted + 288;

  int st = ST_DONE, result = INF_OK;
  Block blk{0, 0, 0, 0, 0, 0};
  if (b < n_blocks) {
    blk = blocks[b];
    st = ST_HEADER;
  }
  const uint32_t comp_end = blk.comp_offset + blk.comp_size;
  uint8_t* obase = out + blk.out_offset;
  const uint32_t osize = blk.out_size;
  Bits br;
  br.buf = 0;
  br.cnt = 0;
  br.ip = blk.comp_offset;
  br.next = 0;
  if (st != ST_DONE) {
    br.next = ld_u32(comp + br.ip);
    br.ip += 4;
    br.refill(comp);
  }
  uint32_t pos = 0, rem = 0, dist = 0, sp = 0;
  bool last = false;
  auto fail = [&]() {
    result = INF_RETRY;
    st = ST_DONE;
  };

  for (unsigned step = 0; step < MAX_STEPS; ++step) {
    const unsigned long long active = __ballot(st != ST_DONE);
    if (!active) break;

    // ---- block headers + table construction: batched (long, divergent) ---------------------------------------------
    const unsigned long long at_header = __ballot(st == ST_HEADER);
    if (at_header && (__popcll(at_header) >= 8 || at_header == active)) {
      if (st == ST_HEADER) {
        br.refill(comp);
        last = br.take(1) != 0;
        const int type = (int)br.take(2);
        if (type == 0) {  // stored
          br.drop(br.cnt & 7);
          br.refill(comp);
          const uint32_t len = br.take(16);
          br.refill(comp);
          const uint32_t nlen = br.take(16);
          if ((len ^ nlen) != 0xFFFFu) fail();
          else {
            sp = br.byte_pos();
            rem = len;
            if (sp + len > comp_end || pos + len > osize) fail();
            else st = ST_STORED;
          }
        } else if (type == 1 || type == 2) {
          bool ok = true;
          int hlit = 288, hdist = 30;
          if (type == 1) {
            for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            for (int s = 0; s < 30; ++s) lens[288 + s] = 5;
            // the fixed distance code has 32 codes of 5 bits (30, 31 unused): complete it for the builder
            lens[288 + 30] = lens[288 + 31] = 5;
            hdist = 32;
          } else {
            br.refill(comp);
            hlit = (int)br.take(5) + 257;
            hdist = (int)br.take(5) + 1;
            const int hclen = (int)br.take(4) + 4;
            if (hlit > 286 || hdist > 30) ok = false;
            // code-length code: 19 x 3 bits, then a 7-bit direct table in the (not yet built) distance root area
            uint8_t cl[19];
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (int i = 0; i < hclen; ++i) {
              br.refill(comp);
              cl[order[i]] = (uint8_t)br.take(3);
            }
            uint8_t* cl_lut = reinterpret_cast<uint8_t*>(dst);  // 128 bytes: [7:3] symbol, [2:0] length (0 = invalid)
            {
              int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
              for (int i = 0; i < 19; ++i) cnt[cl[i]]++;
              cnt[0] = 0;
              int left = 1;
              uint32_t next[8], code = 0;
              for (int l = 1; l <= 7; ++l) {
                left = (left << 1) - cnt[l];
                code = (code + (l > 1 ? (uint32_t)cnt[l - 1] : 0u)) << 1;
                next[l] = code;
              }
              if (left != 0) ok = false;  // over-subscribed or incomplete: let the other kernel judge
              for (int i = 0; i < 128; ++i) cl_lut[i] = 0;
              if (ok)
                for (int s = 0; s < 19; ++s) {
                  const int l = cl[s];
                  if (!l) continue;
                  const uint32_t c = next[l]++;
                  const uint32_t rev = __brev(c) >> (32 - l);
                  for (uint32_t j = rev; j < 128u; j += 1u << l) cl_lut[j] = (uint8_t)((s << 3) | l);
                }
            }
            int i = 0;
            const int total = hlit + hdist;
            while (ok && i < total) {
              br.refill(comp);
              const uint8_t e = cl_lut[br.peek(7)];
              if ((e & 7) == 0) { ok = false; break; }
              br.drop(e & 7);
              const int s = e >> 3;
              if (s < 16) {
                lens[i++] = (uint8_t)s;
              } else {
                int rep, val = 0;
                if (s == 16) {
                  if (i == 0) { ok = false; break; }
                  val = lens[i - 1];
                  rep = 3 + (int)br.take(2);
                } else if (s == 17) {
                  rep = 3 + (int)br.take(3);
                } else {
                  rep = 11 + (int)br.take(7);
                }
                if (i + rep > total) { ok = false; break; }
                while (rep--) lens[i++] = (uint8_t)val;
              }
            }
            if (ok && lens[256] == 0) ok = false;  // no end-of-block code
            if (ok) {  // move the distance lengths behind slot 288 and clear the unused literal slots
              uint8_t dl[30];
              for (int k = 0; k < 30; ++k) dl[k] = k < hdist ? lens[hlit + k] : 0;
              for (int k = hlit; k < 288; ++k) lens[k] = 0;
              for (int k = 0; k < 30; ++k) lens[288 + k] = dl[k];
              hlit = 288;
              hdist = 30;
            }
          }
          if (ok) ok = build_code(lens + 288, hdist, DROOT, dst, dcount, dsorted);
          if (ok) ok = build_code(lens, hlit, LROOT, lit, lcount, lsorted);
          if (ok) st = ST_DECODE;
          else fail();
        } else {
          fail();
        }
        if (br.ip > comp_end + 24u) fail();
      }
    }

    // ---- one literal / length symbol per lane ----------------------------------------------------------------------
    if (st == ST_DECODE) {
      br.refill(comp);
      const uint32_t e = lit[br.peek(LROOT)];
      int sym;
      const int l = (int)(e & 15u);
      if (l != 15 && l != 0) {
        br.drop(l);
        sym = (int)(e >> 4);
      } else if (l == 15) {
        sym = decode_slow(br, lcount, lsorted);
      } else {
        sym = -1;
      }
      if (sym < 0) {
        fail();
      } else if (sym < 256) {
        if (pos < osize) obase[pos++] = (uint8_t)sym;
        else fail();
      } else if (sym == 256) {
        if (last) {
          if (pos != osize) result = INF_RETRY;
          st = ST_DONE;
        } else {
          st = ST_HEADER;
        }
      } else if (sym > 285) {
        fail();
      } else {
        uint32_t base;
        int extra;
        length_code(sym - 257, &base, &extra);
        const uint32_t mlen = base + br.take(extra);
        br.refill(comp);
        const uint32_t e2 = dst[br.peek(DROOT)];
        const int l2 = (int)(e2 & 15u);
        int ds;
        if (l2 != 15 && l2 != 0) {
          br.drop(l2);
          ds = (int)(e2 >> 4);
        } else if (l2 == 15) {
          ds = decode_slow(br, dcount, dsorted);
        } else {
          ds = -1;
        }
        if (ds < 0 || ds > 29) {
          fail();
        } else {
          distance_code(ds, &base, &extra);
          br.refill(comp);
          const uint32_t d = base + br.take(extra);
          if (d > pos || pos + mlen > osize) fail();
          else {
            dist = d;
            rem = mlen;
            st = ST_COPY;
          }
        }
      }
      if (br.ip > comp_end + 24u) fail();
    }

    // ---- match copies: 8 / 4 / 1 bytes per step (a match decoded above copies its first chunk right away) -----------
    if (NOCOPY && st == ST_COPY) {  // timing experiment: what the tokenizer alone costs (output is wrong)
      pos += rem;
      rem = 0;
      st = ST_DECODE;
    }
    if (st == ST_COPY) {
      uint8_t* d8 = obase + pos;
      const uint8_t* s8 = d8 - dist;
      if (dist >= 8 && rem >= 8) {
        st_u64(d8, ld_u64(s8));
        pos += 8;
        rem -= 8;
      } else if (dist >= 4 && rem >= 4) {
        st_u32(d8, ld_u32(s8));
        pos += 4;
        rem -= 4;
      } else {
        *d8 = *s8;
        pos += 1;
        rem -= 1;
      }
      if (rem == 0) st = ST_DECODE;
    }

    // ---- stored blocks: raw bytes from the input -------------------------------------------------------------------
    if (st == ST_STORED) {
      if (rem >= 8) {
        st_u64(obase + pos, ld_u64(comp + sp));
        pos += 8;
        sp += 8;
        rem -= 8;
      } else if (rem > 0) {
        obase[pos++] = comp[sp++];
        rem -= 1;
      }
      if (rem == 0) {
        if (last) {
          if (pos != osize) result = INF_RETRY;
          st = ST_DONE;
        } else {  // the next block header starts at the byte behind the stored data
          br.buf = 0;
          br.cnt = 0;
          br.ip = sp;
          br.next = ld_u32(comp + br.ip);
          br.ip += 4;
          br.refill(comp);
          st = ST_HEADER;
        }
      }
    }
  }
  if (b < n_blocks) status[b] = (st == ST_DONE) ? result : INF_RETRY;
}

}  // namespace

// driver entry points (tools/time_simt_inflate.py): scratch = n_blocks x 1 KiB; status[b] = 0 done, 100 = not handled
extern "C" int simt_scratch_bytes_per_block() { return LANE_SCRATCH; }
extern "C" int simt_inflate(const uint8_t* d_comp, const void* d_blocks, int n_blocks, uint8_t* d_out, uint8_t* d_scratch, int* d_status,
                            int nocopy, float* ms) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_simt<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * LANE_LDS);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_simt<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * LANE_LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  if (nocopy)
    hipLaunchKernelGGL(k_inflate_simt<true>, dim3((n_blocks + 63) / 64), dim3(64), 64 * LANE_LDS, 0, d_comp, (const Block*)d_blocks, n_blocks, d_out,
                       d_scratch, d_status);
  else
    hipLaunchKernelGGL(k_inflate_simt<false>, dim3((n_blocks + 63) / 64), dim3(64), 64 * LANE_LDS, 0, d_comp, (const Block*)d_blocks, n_blocks, d_out,
                       d_scratch, d_status);
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -1;
  hipEventElapsedTime(ms, e0, e1);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
