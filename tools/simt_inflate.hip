// simt_inflate.hip -- RESEARCH SPIKE, not part of the product: DEFLATE with one BGZF block per LANE (plain scalar
// inflate per thread, tables in per-thread global scratch), to see what the vector pipes + memory system can do against
// inflate.hip's one-wavefront-per-block scalar design.  build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC
// tools/simt_inflate.hip -o tools/bin/libsimt_inflate.so ; driver: tools/time_simt_inflate.py
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Block { uint32_t comp_offset, comp_size, out_offset, out_size, crc32, reserved; };

constexpr int LB = 9, DB = 7;               // first-level table bits
constexpr int LUT_WORDS = (1 << LB) + (1 << DB);

struct Bits {
  const uint8_t* p;
  uint64_t buf;
  int cnt;
  __device__ __forceinline__ void refill() {
    if (cnt <= 32) { uint32_t w; __builtin_memcpy(&w, p, 4); p += 4; buf |= (uint64_t)w << cnt; cnt += 32; }
  }
  __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = (uint32_t)buf & ((1u << n) - 1u); buf >>= n; cnt -= n; return v; }
};

__device__ const uint16_t LBASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__device__ const uint8_t LEXT[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__device__ const uint16_t DBASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
__device__ const uint8_t DEXT[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

// canonical code -> first-level table (entry = sym << 4 | len; 0 = longer than the table) + count/offs/sorted symbols
struct Code { uint16_t count[16]; uint16_t sym[288]; };
__device__ int build(const uint8_t* lens, int n, int bits, uint32_t* lut, Code& c) {
  for (int i = 0; i < 16; ++i) c.count[i] = 0;
  for (int s = 0; s < n; ++s) c.count[lens[s]]++;
  c.count[0] = 0;
  uint16_t offs[16]; offs[1] = 0;
  for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + c.count[l];
  for (int s = 0; s < n; ++s) if (lens[s]) c.sym[offs[lens[s]]++] = (uint16_t)s;
  for (int i = 0; i < (1 << bits); ++i) lut[i] = 0;
  int code = 0, idx = 0;
  for (int l = 1; l <= bits; ++l) {
    for (int k = 0; k < c.count[l]; ++k, ++code, ++idx) {
      const uint32_t rev = __brev((uint32_t)code) >> (32 - l);
      const uint32_t e = ((uint32_t)c.sym[idx] << 4) | (uint32_t)l;
      for (uint32_t j = rev; j < (1u << bits); j += 1u << l) lut[j] = e;
    }
    code <<= 1;
  }
  return 1;
}
__device__ __forceinline__ int decode(Bits& b, const uint32_t* lut, int bits, const Code& c) {
  const uint32_t e = lut[(uint32_t)b.buf & ((1u << bits) - 1u)];
  if (e) { b.buf >>= (e & 15); b.cnt -= (e & 15); return (int)(e >> 4); }
  int code = 0, first = 0, index = 0;  // bit-serial canonical decode
  uint32_t v = (uint32_t)b.buf;
  for (int len = 1; len <= 15; ++len) {
    code |= (int)(v & 1u); v >>= 1;
    const int n = c.count[len];
    if (code - n < first) { b.buf >>= len; b.cnt -= len; return c.sym[index + (code - first)]; }
    index += n; first += n; first <<= 1; code <<= 1;
  }
  return -1;
}

__global__ __launch_bounds__(64) void k_simt_inflate(const uint8_t* __restrict__ comp, const Block* __restrict__ blocks, int n_blocks,
                                                     uint8_t* __restrict__ out, uint32_t* __restrict__ scratch, int* __restrict__ status) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= n_blocks) return;
  const Block blk = blocks[b];
  uint32_t* lit = scratch + (size_t)b * LUT_WORDS;
  uint32_t* dst = lit + (1 << LB);
  Code lc, dc;
  uint8_t lens[320];
  Bits br{comp + blk.comp_offset, 0, 0};
  br.refill(); br.refill();
  uint8_t* o = out + blk.out_offset;
  uint32_t pos = 0;
  int err = 0;
  bool last = false;
  while (!last && !err) {
    br.refill();
    last = br.take(1);
    const int type = (int)br.take(2);
    if (type == 0) {
      br.take(br.cnt & 7); br.refill();
      const uint32_t len = br.take(16); br.refill(); br.take(16);
      const uint8_t* src = br.p - (br.cnt >> 3);
      for (uint32_t j = 0; j < len; ++j) o[pos + j] = src[j];
      pos += len;
      br.p = src + len; br.buf = 0; br.cnt = 0; br.refill(); br.refill();
      continue;
    }
    if (type == 1) {
      for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
      for (int s = 0; s < 30; ++s) lens[288 + s] = 5;
      build(lens, 288, LB, lit, lc); build(lens + 288, 30, DB, dst, dc);
    } else if (type == 2) {
      br.refill();
      const int hlit = (int)br.take(5) + 257, hdist = (int)br.take(5) + 1, hclen = (int)br.take(4) + 4;
      const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
      uint8_t cl[19];
      for (int i = 0; i < 19; ++i) cl[i] = 0;
      for (int i = 0; i < hclen; ++i) { br.refill(); cl[order[i]] = (uint8_t)br.take(3); }
      Code cc; uint32_t clut[128];
      build(cl, 19, 7, clut, cc);
      int i = 0;
      while (i < hlit + hdist) {
        br.refill();
        const int s = decode(br, clut, 7, cc);
        if (s < 16) lens[i++] = (uint8_t)s;
        else {
          int rep, val = 0;
          if (s == 16) { val = lens[i - 1]; rep = 3 + (int)br.take(2); }
          else if (s == 17) rep = 3 + (int)br.take(3);
          else rep = 11 + (int)br.take(7);
          while (rep-- && i < 320) lens[i++] = (uint8_t)val;
        }
      }
      uint8_t dl[32];
      for (int k = 0; k < 32; ++k) dl[k] = k < hdist ? lens[hlit + k] : 0;
      for (int k = hlit; k < 288; ++k) lens[k] = 0;
      build(lens, 288, LB, lit, lc); build(dl, 30, DB, dst, dc);
    } else { err = 1; break; }
    for (;;) {
      br.refill();
      int s = decode(br, lit, LB, lc);
      if (s < 256) { if (s < 0) { err = 2; break; } o[pos++] = (uint8_t)s; continue; }
      if (s == 256) break;
      s -= 257;
      if (s >= 29) { err = 3; break; }
      const uint32_t len = LBASE[s] + br.take(LEXT[s]);
      br.refill();
      const int ds = decode(br, dst, DB, dc);
      if (ds < 0 || ds >= 30) { err = 4; break; }
      br.refill();  // 13 extra bits need more than the 32 - 15 left
      const uint32_t d = DBASE[ds] + br.take(DEXT[ds]);
      if (d > pos) { err = 5; break; }
      const uint8_t* src = o + pos - d;
      for (uint32_t j = 0; j < len; ++j) o[pos + j] = src[j];
      pos += len;
    }
  }
  if (!err && pos != blk.out_size) err = 6;
  status[b] = err;
}

extern "C" int simt_inflate(const uint8_t* d_comp, const void* d_blocks, int n_blocks, uint8_t* d_out, uint32_t* d_scratch, int* d_status, float* ms) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_simt_inflate, dim3((n_blocks + 63) / 64), dim3(64), 0, 0, d_comp, (const Block*)d_blocks, n_blocks, d_out, d_scratch, d_status);
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -1;
  hipEventElapsedTime(ms, e0, e1);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int simt_lut_words() { return LUT_WORDS; }
