// tune_k5.hip -- developer harness (NOT part of the product): variants of the per-position quality histogram.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_k5.hip -o tools/bin/tune_k5 ; run: tune_k5 [n_reads] [L]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__host__ __device__ inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
__host__ __device__ inline uint64_t rnd(uint64_t seed, uint64_t col, uint64_t i) { return mix64(seed + col * 0xD1B54A32D192ED03ULL + (i + 1) * 0x9E3779B97F4A7C15ULL); }

__global__ void gen(uint64_t seed, int64_t n, int L, uint8_t* bytes) {
  const int64_t total = n * L; const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / L; const int p = (int)(e - r * L); const uint64_t h = rnd(seed, 0, (uint64_t)e);
    const int s = (int)(h & 0xFF) + (int)((h >> 8) & 0xFF) + (int)((h >> 16) & 0xFF) + (int)((h >> 24) & 0xFF);
    const int d = ((s * 3 + 4096 - 1530) >> 7) - 32; int q = 38 - (10 * p) / L + d; q = q < 0 ? 0 : (q > 41 ? 41 : q);
    bytes[e] = (uint8_t)(33 + q); }
}
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_t ldnt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p)); }

// ---- A: shipped v1: wave per read, byte loads, hist[p][257] -------------------------------------------
__global__ __launch_bounds__(1024) void vA(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];
  for (int i = threadIdx.x; i < L * 257; i += 1024) h[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63; const int64_t gw = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 16;
  for (int64_t r = gw; r < n; r += nw) for (int p = lane; p < L; p += 64) atomicAdd(&h[p * 257 + bytes[r * L + p]], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < L * 256; i += 1024) { unsigned v = h[(i >> 8) * 257 + (i & 255)]; if (v) atomicAdd(&out[i], (unsigned long long)v); }
}

// ---- R: read-only ceiling: 16 B/lane nt loads, J chunks in flight ----------------------------------------
template <int J>
__global__ __launch_bounds__(1024) void vR(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  const int64_t nch = n * L / 16; const int64_t S = (int64_t)gridDim.x * 1024; unsigned acc = 0;
  for (int64_t c = (int64_t)blockIdx.x * 1024 + threadIdx.x; c + (J - 1) * S < nch; c += J * S) {
    v4i_t v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) v[j] = ldnt(bytes + 16 * (c + j * S));
#pragma unroll
    for (int j = 0; j < J; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  if (acc == 0x1234567u) atomicAdd(&out[0], 1ull);
}

// ---- B: register-direct: each lane owns 16-byte chunks; bins [p][256] with a per-row rotation -------------
// ROT: 0 = stride 257 (bank = p + byte), 1 = row stride 256 and byte' = byte + 3*(p>>4) + p
template <int J, int ROT>
__global__ __launch_bounds__(1024) void vB(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];
  const int HS = ROT ? 256 : 257;
  for (int i = threadIdx.x; i < L * HS; i += 1024) h[i] = 0;
  __syncthreads();
  const int64_t nch = n * L / 16; const int64_t S = (int64_t)gridDim.x * 1024;
  const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  int p0 = (int)((16 * c0) % L); const int pS = (int)((16 * S) % L);
  auto add = [&](int p, unsigned b) {
    if (ROT) atomicAdd(&h[p * 256 + ((b + 3 * (p >> 4) + p) & 255)], 1u);
    else atomicAdd(&h[p * 257 + b], 1u); };
  auto chunk = [&](v4i_t v, int p) {
    const unsigned d[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) { int pk = p + k; pk = pk >= L ? pk - L : pk; add(pk, (d[k >> 2] >> (8 * (k & 3))) & 0xFF); } };
  int64_t c = c0;
  for (; c + (J - 1) * S < nch; c += J * S) {
    v4i_t v[J]; int pj[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { v[j] = ldnt(bytes + 16 * (c + j * S)); pj[j] = p0; p0 += pS; p0 = p0 >= L ? p0 - L : p0; }
#pragma unroll
    for (int j = 0; j < J; ++j) chunk(v[j], pj[j]);
  }
  for (; c < nch; c += S) { chunk(ldnt(bytes + 16 * c), p0); p0 += pS; p0 = p0 >= L ? p0 - L : p0; }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 256; i += 1024) { const int p = i >> 8, b = i & 255;
    unsigned v = ROT ? h[p * 256 + ((b + 3 * (p >> 4) + p) & 255)] : h[p * 257 + b]; if (v) atomicAdd(&out[i], (unsigned long long)v); }
}

// ---- C: like B but two private histogram copies (waves 0-7 / 8-15) when they fit, halving same-bin contention
template <int J>
__global__ __launch_bounds__(1024) void vC(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];
  // 16-bit packed pairs: bin (p, b) lives in half (b & 1) of word [p][b >> 1]; row = 128 words + 1 pad
  for (int i = threadIdx.x; i < L * 129; i += 1024) h[i] = 0;
  __syncthreads();
  const int64_t nch = n * L / 16; const int64_t S = (int64_t)gridDim.x * 1024;
  const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  int p0 = (int)((16 * c0) % L); const int pS = (int)((16 * S) % L);
  auto chunk = [&](v4i_t v, int p) {
    const unsigned d[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) { int pk = p + k; pk = pk >= L ? pk - L : pk; const unsigned b = (d[k >> 2] >> (8 * (k & 3))) & 0xFF;
      atomicAdd(&h[pk * 129 + (b >> 1)], 1u << (16 * (b & 1))); } };
  int it = 0;
  for (int64_t c = c0; c + (J - 1) * S < nch; c += J * S) {
    v4i_t v[J]; int pj[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { v[j] = ldnt(bytes + 16 * (c + j * S)); pj[j] = p0; p0 += pS; p0 = p0 >= L ? p0 - L : p0; }
#pragma unroll
    for (int j = 0; j < J; ++j) chunk(v[j], pj[j]);
    if (++it == 2) {  // 16-bit fields: flush well before 65535 (worst case all 1024 threads hit one bin: 16*J*2 per thread)
      it = 0; __syncthreads();
      for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i >> 7, w = i & 127; const unsigned x = h[p * 129 + w]; h[p * 129 + w] = 0;
        if (x & 0xFFFF) atomicAdd(&out[p * 256 + 2 * w], (unsigned long long)(x & 0xFFFF)); if (x >> 16) atomicAdd(&out[p * 256 + 2 * w + 1], (unsigned long long)(x >> 16)); }
      __syncthreads(); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i >> 7, w = i & 127; const unsigned x = h[p * 129 + w];
    if (x & 0xFFFF) atomicAdd(&out[p * 256 + 2 * w], (unsigned long long)(x & 0xFFFF)); if (x >> 16) atomicAdd(&out[p * 256 + 2 * w + 1], (unsigned long long)(x >> 16)); }
}

// ---- D: conflict-free layout.  One dword (4 consecutive positions) per lane per load; histogram is byte-major
// h[byte < 128][perm(p)], perm(p) = (p & 3) * Q + (p >> 2), Q = LP / 4 (multiple of 32): for a fixed byte-in-dword
// k, consecutive lanes hit consecutive banks whatever the data is.  A4: L % 4 == 0 (a dword never straddles reads).
template <int LP, int J, bool A4>
__global__ __launch_bounds__(1024) void vD(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];  // [128][LP]
  constexpr int Q = LP / 4;
  for (int i = threadIdx.x; i < 128 * LP; i += 1024) h[i] = 0;
  __syncthreads();
  const int64_t nd = n * L / 4; const int64_t S = (int64_t)gridDim.x * 1024;
  const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  int p0 = (int)((4 * c0) % L); const int pS = (int)((4 * S) % L);
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  auto one = [&](unsigned d, int p) {
    if (__builtin_expect((d & 0x80808080u) != 0, 0)) {  // non-ASCII byte: slow path straight to global memory
#pragma unroll
      for (int k = 0; k < 4; ++k) { int pk = p + k; pk = pk >= L ? pk - L : pk; atomicAdd(&out[pk * 256 + ((d >> (8 * k)) & 0xFF)], 1ull); }
      return; }
    if (A4) {
      unsigned* base = h + (p >> 2);
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(base + ((d >> (8 * k)) & 0xFF) * LP + k * Q, 1u);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { int pk = p + k; pk = pk >= L ? pk - L : pk;
        atomicAdd(h + ((d >> (8 * k)) & 0xFF) * LP + (pk & 3) * Q + (pk >> 2), 1u); }
    } };
  int64_t c = c0;
  for (; c + (J - 1) * S < nd; c += J * S) {
    unsigned v[J]; int pj[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { v[j] = __builtin_nontemporal_load(src + c + j * S); pj[j] = p0; p0 += pS; p0 = p0 >= L ? p0 - L : p0; }
#pragma unroll
    for (int j = 0; j < J; ++j) one(v[j], pj[j]);
  }
  for (; c < nd; c += S) { one(src[c], p0); p0 += pS; p0 = p0 >= L ? p0 - L : p0; }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i % L, b = i / L;
    const unsigned v = h[b * LP + (p & 3) * Q + (p >> 2)]; if (v) atomicAdd(&out[p * 256 + b], (unsigned long long)v); }
}

// ---- E: D for short reads.  With L/4 < 32 two lanes of one 32-lane LDS group can hold the same dword-of-read q
// (lanes 25 apart at L = 100) and collide on a bank (different byte) or an address (same byte).  Column = position of
// the lane inside its 32-lane group's run of the stream, u = q(first lane of the group) + (lane & 31) < L/4 + 31: the 32
// lanes of a group then always hit 32 consecutive banks.  Folded over u mod L/4 at the end.
template <int J>
__global__ __launch_bounds__(1024) void vE(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];  // [128][LP], LP = 4 * QP
  const int QL = L / 4, QP = (QL + 31 + 7) & ~7, LP = 4 * QP;
  for (int i = threadIdx.x; i < 128 * LP; i += 1024) h[i] = 0;
  __syncthreads();
  const int64_t nd = n * QL; const int64_t S = (int64_t)gridDim.x * 1024;
  const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const int l31 = threadIdx.x & 31;
  int tg = (int)((c0 - l31) % QL); const int tS = (int)(S % QL);
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  auto one = [&](unsigned d, int u) {
    if (__builtin_expect((d & 0x80808080u) != 0, 0)) {
      int q = u; while (q >= QL) q -= QL;
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&out[(4 * q + k) * 256 + ((d >> (8 * k)) & 0xFF)], 1ull);
      return; }
    unsigned* base = h + u;
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(base + ((d >> (8 * k)) & 0xFF) * LP + k * QP, 1u);
  };
  int64_t c = c0;
  for (; c + (J - 1) * S < nd; c += J * S) {
    unsigned v[J]; int uj[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { v[j] = __builtin_nontemporal_load(src + c + j * S); uj[j] = tg + l31; tg += tS; tg = tg >= QL ? tg - QL : tg; }
#pragma unroll
    for (int j = 0; j < J; ++j) one(v[j], uj[j]);
  }
  for (; c < nd; c += S) { one(src[c], tg + l31); tg += tS; tg = tg >= QL ? tg - QL : tg; }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i % L, b = i / L; unsigned v = 0;
    for (int u = p >> 2; u < QP; u += QL) v += h[b * LP + (p & 3) * QP + u];
    if (v) atomicAdd(&out[p * 256 + b], (unsigned long long)v); }
}


// ---- F: collision-free for EVERY lane of a wave instruction.  ds_add_u32 costs 4.2 LDS clocks per wave instruction when all
// 64 addresses differ, 7+ when two lanes hit the SAME address (tools/lds_atomic_rate.hip) -- and with D's layout lanes 25
// apart (L = 100) share a column, so nearly every instruction holds such a pair.  Here a wave always works on whole "rows"
// of 64 consecutive dwords whose row index is == r (mod Ld), Ld = L/4: lane l then sits at dword-of-read (64 r + l) % Ld in
// EVERY iteration; lanes that wrap around the read length (w = (d0 + l) / Ld = 0..3) use separate table copies, so no two
// lanes of an instruction ever share an address, whatever the data.  Table: [w][byte - 32 (96 printable rows)][k][Ld].
template <int J>
__global__ __launch_bounds__(1024) void vF(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];
  const int Ld = L / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = (Ld - 1 + 63) / Ld + 1;
  const int nwords = C * 96 * 4 * Ld;
  for (int i = threadIdx.x; i < nwords; i += 1024) h[i] = 0;
  __syncthreads();
  const int NW = gridDim.x * 16, g = blockIdx.x * 16 + wave;
  const int r = g % Ld, slot = g / Ld, nslots = NW / Ld;
  const int64_t nd = n * Ld, nrows = nd / 64;
  const int t = (64 * r) % Ld + lane, w = t / Ld, d = t - w * Ld;
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  // byte address of bin (byte, k) for this lane: byte * 16 Ld + base[k]
  const unsigned rowb = 16u * Ld;
  unsigned base[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) base[k] = 4u * (unsigned)(((w * 96 - 32) * 4 + k) * Ld + d);
  auto slow = [&](unsigned dw, int q) {  // q = dword-of-read
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b >= 32 && b < 128) atomicAdd(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(h) + b * rowb + base[k]), 1u);
      else atomicAdd(&out[(size_t)(4 * q + k) * 256 + b], 1ull);
    }
  };
  auto one = [&](unsigned dw) {
    const unsigned u = (dw & 0x7F7F7F7Fu) + 0x60606060u;          // bit 7 of each byte: (b & 127) >= 32
    if (__builtin_expect(((u & ~dw) & 0x80808080u) != 0x80808080u, 0)) { slow(dw, d); return; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned b = (dw >> (8 * k)) & 0xFF;
      atomicAdd(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(h) + b * rowb + base[k]), 1u);
    }
  };
  if (slot < nslots) {
    const int64_t qstep = nslots;
    int64_t q = slot;
    for (; (q + (J - 1) * qstep) * Ld + r < nrows; q += J * qstep) {
      unsigned v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = __builtin_nontemporal_load(src + ((q + j * qstep) * Ld + r) * 64 + lane);
#pragma unroll
      for (int j = 0; j < J; ++j) one(v[j]);
    }
    for (; q * Ld + r < nrows; q += qstep) one(src[(q * Ld + r) * 64 + lane]);
  }
  if (g == NW - 1) {  // the last partial row (< 64 dwords)
    const int64_t c = nrows * 64 + lane;
    if (c < nd) {
      const int q = (int)(c % Ld);
      const unsigned dw = src[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned b = (dw >> (8 * k)) & 0xFF;
        if (b >= 32 && b < 128) atomicAdd(&h[((0 * 96 + (int)b - 32) * 4 + k) * Ld + q], 1u);
        else atomicAdd(&out[(size_t)(4 * q + k) * 256 + b], 1ull);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 96; i += 1024) {
    const int p = i % L, b = i / L;  // b = byte - 32
    unsigned v = 0;
    for (int ww = 0; ww < C; ++ww) v += h[((ww * 96 + b) * 4 + (p & 3)) * Ld + (p >> 2)];
    if (v) atomicAdd(&out[p * 256 + b + 32], (unsigned long long)v);
  }
}

// ---- H: a lane GROUP (G = 16 / 32 / 64 lanes) per read, lane-in-group = dword-of-read, one table copy per group of the
// wave: [k][byte < 128][copy][G] x u32 -- every lane of a wave instruction owns its own word whatever the data is (no bank
// conflict, no same-address serialisation), the address is ONE v_perm_b32 (byte 1 of the address = the data byte; byte 0
// = per-lane constant; k in the instruction offset / byte 2).  L/4 of the G lanes are active.
template <int G, int J>
__global__ __launch_bounds__(1024) void vH(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];  // [4][128][64] words = 128 KB
  const int Ld = L / 4;
  for (int i = threadIdx.x; i < 4 * 128 * 64; i += 1024) h[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int d = lane & (G - 1);
  const bool active = d < Ld;
  constexpr int RPW = 64 / G;                       // reads per wave instruction
  const int64_t gw = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 16;
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  // address bytes: [0] = lane * 4 (copy * G * 4 + d * 4 = lane * 4 since copy = lane / G), [1] = data byte, [2] = k >> 1; k & 1 -> +32 KiB
  const unsigned c01 = (unsigned)lane * 4u, c23 = c01 | 0x10000u;
  char* hb = reinterpret_cast<char*>(h);
  auto one = [&](unsigned dw, int q) {
    if (__builtin_expect((dw & 0x80808080u) != 0, 0)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const unsigned b = (dw >> (8 * k)) & 0xFF;
        if (b < 128) atomicAdd(reinterpret_cast<unsigned*>(hb + (k >> 1) * 65536 + (k & 1) * 32768 + b * 256 + c01), 1u);
        else atomicAdd(&out[(size_t)(4 * q + k) * 256 + b], 1ull); }
      return; }
    // v_perm_b32 D = {S0, S1} bytes: selector byte i picks: 0-3 = S1 bytes, 4-7 = S0 bytes
    const unsigned a0 = __builtin_amdgcn_perm(dw, c01, 0x03020400u);  // byte1 <- dw.byte0
    const unsigned a1 = __builtin_amdgcn_perm(dw, c01, 0x03020500u);  // byte1 <- dw.byte1
    const unsigned a2 = __builtin_amdgcn_perm(dw, c23, 0x03020600u);  // byte1 <- dw.byte2, byte2 = 1
    const unsigned a3 = __builtin_amdgcn_perm(dw, c23, 0x03020700u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a0), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a1 + 32768), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a2), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a3 + 32768), 1u);
  };
  const int64_t nrw = n / RPW;  // whole wave-rows of RPW reads
  int64_t rw = gw;
  const int sub = lane / G;
  for (; rw + (J - 1) * nw < nrw; rw += J * nw) {
    unsigned v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t idx = ((rw + j * nw) * RPW + sub) * Ld + (active ? d : 0);
      v[j] = __builtin_nontemporal_load(src + idx);
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < J; ++j) one(v[j], d);
    }
  }
  for (; rw < nrw; rw += nw) { if (active) one(src[(rw * RPW + sub) * Ld + d], d); }
  if (gw == 0) for (int64_t r = nrw * RPW + sub; r < n; r += RPW) if (active) one(src[r * Ld + d], d);
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i % L, b = i / L; unsigned v = 0;
    for (int c = 0; c < RPW; ++c) v += h[((p & 3) >> 1) * 16384 + ((p & 3) & 1) * 8192 + b * 64 + c * G + (p >> 2)];
    if (v) atomicAdd(&out[p * 256 + b], (unsigned long long)v); }
}

// ---- I: F's work distribution (a wave owns the 64-dword rows whose index is == r mod Ld, so every lane keeps ONE
// dword-of-read d and wrap count w for the whole launch; all 64 lanes busy) + H's table and addressing ([k][byte][64 cols],
// one v_perm_b32 per byte).  Column = d + 32 (w & 1) for Ld <= 32, d otherwise: two lanes of a 32-lane half with the same d
// always differ by one in w, so a half never holds two equal addresses (equal addresses ACROSS halves cost nothing:
// tools/lds_atomic_rate.hip); bank = d, at worst a few 2-way conflicts (4.3 instead of 4.2 clocks).
template <int J>
__global__ __launch_bounds__(1024) void vI(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  extern __shared__ unsigned h[];  // [4][128][64] words = 128 KB
  const int Ld = L / 4;
  for (int i = threadIdx.x; i < 4 * 128 * 64; i += 1024) h[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NW = gridDim.x * 16, g = blockIdx.x * 16 + wave;
  const int r = g % Ld, slot = g / Ld, nslots = NW / Ld;
  const int64_t nd = n * Ld, nrows = nd / 64;
  const int t = (64 * r) % Ld + lane, w = t / Ld, d = t - w * Ld;
  const int col = Ld <= 32 ? d + 32 * (w & 1) : d;
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  const unsigned c01 = (unsigned)col * 4u, c23 = c01 | 0x10000u;
  char* hb = reinterpret_cast<char*>(h);
  auto slow = [&](unsigned dw, int q, unsigned cc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b < 128) atomicAdd(reinterpret_cast<unsigned*>(hb + (k >> 1) * 65536 + (k & 1) * 32768 + b * 256 + cc), 1u);
      else atomicAdd(&out[(size_t)(4 * q + k) * 256 + b], 1ull); }
  };
  auto one = [&](unsigned dw) {
    if (__builtin_expect((dw & 0x80808080u) != 0, 0)) { slow(dw, d, c01); return; }
    const unsigned a0 = __builtin_amdgcn_perm(dw, c01, 0x03020400u);
    const unsigned a1 = __builtin_amdgcn_perm(dw, c01, 0x03020500u);
    const unsigned a2 = __builtin_amdgcn_perm(dw, c23, 0x03020600u);
    const unsigned a3 = __builtin_amdgcn_perm(dw, c23, 0x03020700u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a0), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a1 + 32768), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a2), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a3 + 32768), 1u);
  };
  if (slot < nslots) {
    const int64_t qstep = nslots;
    int64_t q = slot;
    for (; (q + (J - 1) * qstep) * Ld + r < nrows; q += J * qstep) {
      unsigned v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = __builtin_nontemporal_load(src + ((q + j * qstep) * Ld + r) * 64 + lane);
#pragma unroll
      for (int j = 0; j < J; ++j) one(v[j]);
    }
    for (; q * Ld + r < nrows; q += qstep) one(src[(q * Ld + r) * 64 + lane]);
  }
  if (g == NW - 1) {  // the last partial row (< 64 dwords)
    const int64_t c = nrows * 64 + lane;
    if (c < nd) { const int q = (int)(c % Ld); slow(src[c], q, (unsigned)q * 4u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i % L, b = i / L, k = p & 3;
    unsigned v = h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2)];
    if (Ld <= 32) v += h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2) + 32];
    if (v) atomicAdd(&out[p * 256 + b], (unsigned long long)v); }
}

// ---- J: I with (1) a STATIC LDS table -- its address is the constant 0, so the bin address needs no add after the v_perm
// (with extern __shared__ the compiler keeps a `v_add_u32 v, <lds base>, v` per atomic: 4 of I's 10 VALU per dword) and
// (2) the non-ASCII test hoisted: the J dwords of an iteration are ORed together (v_or3) and tested once.
template <int J, int G>
__global__ __launch_bounds__(1024) void vJ(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  __shared__ unsigned h[4 * 128 * 64];  // [4][128][64] words = 128 KB
  const int Ld = L / 4;
  for (int i = threadIdx.x; i < 4 * 128 * 64; i += 1024) h[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NW = gridDim.x * 16, g = blockIdx.x * 16 + wave;
  const int r = g % Ld, slot = g / Ld, nslots = NW / Ld;
  const int64_t nd = n * Ld, nrows = nd / 64;
  const int t = (64 * r) % Ld + lane, w = t / Ld, d = t - w * Ld;
  const int col = Ld <= 32 ? d + 32 * (w & 1) : d;
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  const unsigned c01 = (unsigned)col * 4u, c23 = c01 | 0x10000u;
  char* hb = reinterpret_cast<char*>(h);
  auto slow = [&](unsigned dw, int q, unsigned cc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b < 128) atomicAdd(reinterpret_cast<unsigned*>(hb + (k >> 1) * 65536 + (k & 1) * 32768 + b * 256 + cc), 1u);
      else atomicAdd(&out[(size_t)(4 * q + k) * 256 + b], 1ull); }
  };
  auto fast = [&](unsigned dw) {
    const unsigned a0 = __builtin_amdgcn_perm(dw, c01, 0x03020400u);
    const unsigned a1 = __builtin_amdgcn_perm(dw, c01, 0x03020500u);
    const unsigned a2 = __builtin_amdgcn_perm(dw, c23, 0x03020600u);
    const unsigned a3 = __builtin_amdgcn_perm(dw, c23, 0x03020700u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a0), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a1 + 32768), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a2), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a3 + 32768), 1u);
  };
  auto one = [&](unsigned dw) {
    if (__builtin_expect((dw & 0x80808080u) != 0, 0)) { slow(dw, d, c01); return; }
    fast(dw);
  };
  if (slot < nslots) {
    const int64_t qstep = nslots;
    int64_t q = slot;
    for (; (q + (J - 1) * qstep) * Ld + r < nrows; q += J * qstep) {
      unsigned v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = __builtin_nontemporal_load(src + ((q + j * qstep) * Ld + r) * 64 + lane);
      if (G == 1) {
#pragma unroll
        for (int j = 0; j < J; ++j) one(v[j]);
      } else {
#pragma unroll
        for (int j0 = 0; j0 < J; j0 += G) {
          unsigned any = 0;
#pragma unroll
          for (int j = j0; j < j0 + G; ++j) any |= v[j];
          if (__builtin_expect(__any((any & 0x80808080u) != 0), 0)) {
#pragma unroll 1
            for (int j = j0; j < j0 + G; ++j) one(v[j]);
          } else {
#pragma unroll
            for (int j = j0; j < j0 + G; ++j) fast(v[j]);
          }
        }
      }
    }
    for (; q * Ld + r < nrows; q += qstep) one(src[(q * Ld + r) * 64 + lane]);
  }
  if (g == NW - 1) {  // the last partial row (< 64 dwords)
    const int64_t c = nrows * 64 + lane;
    if (c < nd) { const int q = (int)(c % Ld); slow(src[c], q, (unsigned)q * 4u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i % L, b = i / L, k = p & 3;
    unsigned v = h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2)];
    if (Ld <= 32) v += h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2) + 32];
    if (v) atomicAdd(&out[p * 256 + b], (unsigned long long)v); }
}

// ---- L: K (static LDS table) with (1) a ROLLING load window: after a dword is consumed its register is refilled with the row J
// ahead, so a wave keeps J - 1 loads in flight all the time (K issues J, then drains them to zero while it works through the
// atomics: 12 in flight on average) and (2) wave-uniform row addressing: the wave index goes through readfirstlane, so the row
// base is scalar and the load is `global_load_dword v, v_lane_off, s[base]` -- no 64-bit vector add per load.
template <int J>
__global__ __launch_bounds__(1024) void vL(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  __shared__ unsigned h[4 * 128 * 64];
  const int Ld = L / 4;
  for (int i = threadIdx.x; i < 4 * 128 * 64; i += 1024) h[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = gridDim.x * 16, g = blockIdx.x * 16 + wave;
  const int r = g % Ld, slot = g / Ld, nslots = NW / Ld;
  const int64_t nd = n * Ld, nrows = nd / 64;
  const int t = (64 * r) % Ld + lane, w = t / Ld, d = t - w * Ld;
  const int col = Ld <= 32 ? d + 32 * (w & 1) : d;
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  const unsigned c01 = (unsigned)col * 4u, c23 = c01 | 0x10000u;
  char* hb = reinterpret_cast<char*>(h);
  auto slow = [&](unsigned dw, int q, unsigned cc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b < 128) atomicAdd(reinterpret_cast<unsigned*>(hb + (k >> 1) * 65536 + (k & 1) * 32768 + b * 256 + cc), 1u);
      else atomicAdd(&out[(size_t)(4 * q + k) * 256 + b], 1ull); }
  };
  auto one = [&](unsigned dw) {
    if (__builtin_expect((dw & 0x80808080u) != 0, 0)) { slow(dw, d, c01); return; }
    const unsigned a0 = __builtin_amdgcn_perm(dw, c01, 0x03020400u);
    const unsigned a1 = __builtin_amdgcn_perm(dw, c01, 0x03020500u);
    const unsigned a2 = __builtin_amdgcn_perm(dw, c23, 0x03020600u);
    const unsigned a3 = __builtin_amdgcn_perm(dw, c23, 0x03020700u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a0), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a1 + 32768), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a2), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a3 + 32768), 1u);
  };
  if (slot < nslots) {
    const int64_t stride = (int64_t)nslots * Ld, first = (int64_t)slot * Ld + r;  // this wave's rows: first + i * stride
    const int64_t nmine = first < nrows ? (nrows - first + stride - 1) / stride : 0;
    const unsigned* p = src + first * 64;  // wave-uniform
    const int64_t pstep = stride * 64;
    int64_t i = 0;
    if (nmine >= J) {
      unsigned v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = __builtin_nontemporal_load(p + j * pstep + lane);
      p += J * pstep;
      for (i = J; i + J <= nmine; i += J) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          one(v[j]);
          v[j] = __builtin_nontemporal_load(p + j * pstep + lane);
        }
        p += J * pstep;
      }
#pragma unroll
      for (int j = 0; j < J; ++j) one(v[j]);
    }
    if (i < nmine) {  // the last < J rows, all loads at once
      unsigned v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = __builtin_nontemporal_load(p + (i + j < nmine ? j : 0) * pstep + lane);
#pragma unroll
      for (int j = 0; j < J; ++j)
        if (i + j < nmine) one(v[j]);
    }
  }
  if (g == NW - 1) {  // the last partial row (< 64 dwords)
    const int64_t c = nrows * 64 + lane;
    if (c < nd) { const int q = (int)(c % Ld); slow(src[c], q, (unsigned)q * 4u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i % L, b = i / L, k = p & 3;
    unsigned v = h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2)];
    if (Ld <= 32) v += h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2) + 32];
    if (v) atomicAdd(&out[p * 256 + b], (unsigned long long)v); }
}

// ---- M: L without a branch per dword.  The dword is masked to 7 bits per byte before the v_perm (a byte >= 128 lands in the bin
// of byte & 127 for the moment), the unmasked dwords of an iteration are ORed together, and ONE wave-uniform test per J rows
// sends an iteration that saw a high bit through a fix-up: its rows are read again, and every byte >= 128 is taken out of the bin
// it went to (ds_sub) and added to the global histogram.  Real quality strings never take it.  Per dword: v_and, v_or, 4 v_perm,
// 4 ds_add, 1 load with a scalar base -- no exec-mask juggling (L: v_and, v_cmp, 6 scalar instructions and 2 branches).
template <int J>
__global__ __launch_bounds__(1024) void vM(const uint8_t* bytes, int64_t n, int L, unsigned long long* out) {
  __shared__ unsigned h[4 * 128 * 64];
  const int Ld = L / 4;
  for (int i = threadIdx.x; i < 4 * 128 * 64; i += 1024) h[i] = 0;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = gridDim.x * 16, g = blockIdx.x * 16 + wave;
  const int r = g % Ld, slot = g / Ld, nslots = NW / Ld;
  const int64_t nd = n * Ld, nrows = nd / 64;
  const int t = (64 * r) % Ld + (int)lane, w = t / Ld, d = t - w * Ld;
  const int col = Ld <= 32 ? d + 32 * (w & 1) : d;
  const unsigned* src = reinterpret_cast<const unsigned*>(bytes);
  const unsigned c01 = (unsigned)col * 4u, c23 = c01 | 0x10000u;
  char* hb = reinterpret_cast<char*>(h);
  auto bin = [&](int k, unsigned b, unsigned cc) { return reinterpret_cast<unsigned*>(hb + (k >> 1) * 65536 + (k & 1) * 32768 + b * 256 + cc); };
  auto slow = [&](unsigned dw, int q, unsigned cc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b < 128) atomicAdd(bin(k, b, cc), 1u);
      else atomicAdd(&out[(size_t)(4 * q + k) * 256 + b], 1ull); }
  };
  auto fix = [&](unsigned dw) {  // dw went through fast(): move its bytes >= 128 to where they belong
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b >= 128) { atomicSub(bin(k, b & 127, c01), 1u); atomicAdd(&out[(size_t)(4 * d + k) * 256 + b], 1ull); } }
  };
  auto fast = [&](unsigned dw) {
    const unsigned m = dw & 0x7F7F7F7Fu;
    const unsigned a0 = __builtin_amdgcn_perm(m, c01, 0x03020400u);
    const unsigned a1 = __builtin_amdgcn_perm(m, c01, 0x03020500u);
    const unsigned a2 = __builtin_amdgcn_perm(m, c23, 0x03020600u);
    const unsigned a3 = __builtin_amdgcn_perm(m, c23, 0x03020700u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a0), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a1 + 32768), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a2), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a3 + 32768), 1u);
  };
  if (slot < nslots) {
    const int64_t stride = (int64_t)nslots * Ld, first = (int64_t)slot * Ld + r;  // this wave's rows: first + i * stride
    const int64_t nmine = first < nrows ? (nrows - first + stride - 1) / stride : 0;
    const unsigned* p = src + first * 64;  // wave-uniform
    const int64_t pstep = stride * 64;
    int64_t i = 0;
    auto fixup = [&](unsigned acc) {
      if (__builtin_expect(__any((acc & 0x80808080u) != 0), 0)) {
#pragma unroll 1
        for (int j = 0; j < J; ++j) fix(p[j * pstep + lane]);
      }
    };
    if (nmine >= J) {
      unsigned v[J];
      const unsigned* ld = p + lane;  // running pointer: loads go out in row order
#pragma unroll
      for (int j = 0; j < J; ++j) { v[j] = __builtin_nontemporal_load(ld); ld += pstep; __builtin_amdgcn_sched_barrier(0); }  // in row order, as the loop issues them
      for (i = J; i + J <= nmine; i += J) {
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          asm volatile("v_or_b32 %0, %0, %1" : "+v"(acc) : "v"(v[j]));  // as asm: a v_or3 of two rows would be placed after the
          fast(v[j]);                                                     // first one's reload and cost a register rotation
          v[j] = __builtin_nontemporal_load(ld); ld += pstep;
          __builtin_amdgcn_sched_barrier(0);  // keep the window rolling: without it the scheduler copies all J masked dwords
        }                                     // first (a wait for every load) and issues the J loads in one burst
        fixup(acc);
        p += J * pstep;
      }
      unsigned acc = 0;
#pragma unroll
      for (int j = 0; j < J; ++j) { acc |= v[j]; fast(v[j]); }
      fixup(acc);
      p += J * pstep;
    }
    if (i < nmine) {  // the last < J rows, all loads at once
      unsigned v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = __builtin_nontemporal_load(p + (i + j < nmine ? j : 0) * pstep + lane);
#pragma unroll
      for (int j = 0; j < J; ++j)
        if (i + j < nmine) slow(v[j], d, c01);
    }
  }
  if (g == NW - 1) {  // the last partial row (< 64 dwords)
    const int64_t c = nrows * 64 + lane;
    if (c < nd) { const int q = (int)(c % Ld); slow(src[c], q, (unsigned)q * 4u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L * 128; i += 1024) { const int p = i % L, b = i / L, k = p & 3;
    unsigned v = h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2)];
    if (Ld <= 32) v += h[(k >> 1) * 16384 + (k & 1) * 8192 + b * 64 + (p >> 2) + 32];
    if (v) atomicAdd(&out[p * 256 + b], (unsigned long long)v); }
}

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : (int64_t)2e8; const int L = argc > 2 ? atoi(argv[2]) : 100;
  n = n / 4096 * 4096;
  uint8_t* bytes; unsigned long long* out; CK(hipMalloc(&bytes, n * L + 64)); CK(hipMalloc(&out, (size_t)L * 256 * 8));
  hipLaunchKernelGGL(gen, dim3(65536), dim3(256), 0, 0, 5ull, n, L, bytes); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> ref, h((size_t)L * 256); bool have_ref = false;
  auto run = [&](const char* name, auto kern, size_t lds, int grid) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float tot = 0, best = 1e9; const int reps = 5;
    for (int r = 0; r < reps + 1; ++r) { CK(hipMemsetAsync(out, 0, (size_t)L * 256 * 8, 0)); CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, bytes, n, L, out); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 1) { tot += ms; if (ms < best) best = ms; } }
    CK(hipGetLastError());
    CK(hipMemcpy(h.data(), out, (size_t)L * 256 * 8, hipMemcpyDeviceToHost));
    const char* ok = "-";
    if (lds) { if (!have_ref) { ref = h; have_ref = true; ok = "ref"; } else ok = (h == ref) ? "match" : "MISMATCH"; }
    const double gbs = (double)n * (L + 4) / (tot / reps * 1e-3) / 1e9;
    printf("%-26s grid=%4d  avg %.3f ms best %.3f ms  %.0f GB/s (%.1f%% of 8TB/s) %s\n", name, grid, tot / reps, best, gbs, gbs / 80.0, ok);
  };
  printf("n_reads=%lld L=%d bytes=%.2f GB\n", (long long)n, L, (double)n * L / 1e9);
  if (getenv("TUNE_K5_ONLY_IJ")) {  // the shipped variant against its successors, on one box; the list runs three times
    const size_t il = 4 * 128 * 64 * 4;  // (the clocks of an idle GPU take ~100 ms of load to settle: early entries read low)
    for (int pass = 0; pass < 3; ++pass) {
      printf("pass %d\n", pass);
      run("R read-only J4", vR<4>, 0, 256);
      run("I rows+perm J24", vI<24>, il, 256);
      run("K I+static J24", vJ<24, 1>, 4, 256);
      run("L rolling J16", vL<16>, 4, 256);
      run("L rolling J24", vL<24>, 4, 256);
      run("L rolling J32", vL<32>, 4, 256);
      run("L rolling J48", vL<48>, 4, 256);
      run("M branch-free J16", vM<16>, 4, 256);
      run("M branch-free J24", vM<24>, 4, 256);
      run("M branch-free J32", vM<32>, 4, 256);
    }
    return 0;
  }
  run("A v1 wave/read", vA, (size_t)L * 257 * 4, 256);
  run("R read-only J1", vR<1>, 0, 256);
  run("R read-only J4", vR<4>, 0, 256);
  run("R read-only J4 x2", vR<4>, 0, 512);
  if (L % 4 == 0 && L >= 64 && L <= 256) { const size_t il = 4 * 128 * 64 * 4; run("I rows+perm J8", vI<8>, il, 256); run("I rows+perm J16", vI<16>, il, 256); run("I rows+perm J24", vI<24>, il, 256); }
  run("B reg-direct J1 s257", vB<1, 0>, (size_t)L * 257 * 4, 256);
  run("B reg-direct J4 s257", vB<4, 0>, (size_t)L * 257 * 4, 256);
  if (L <= 128) {
    if (L % 4 == 0) { run("D cf LP128 J4 A4", vD<128, 4, true>, 128 * 128 * 4, 256); run("D cf LP128 J8 A4", vD<128, 8, true>, 128 * 128 * 4, 256);
      run("D cf LP128 J16 A4", vD<128, 16, true>, 128 * 128 * 4, 256); run("D cf LP128 J8 A4 x2", vD<128, 8, true>, 128 * 128 * 4, 512); }
    run("D cf LP128 J8 gen", vD<128, 8, false>, 128 * 128 * 4, 256); run("D cf LP128 J16 gen", vD<128, 16, false>, 128 * 128 * 4, 256);
    if (L % 4 == 0) { const int Ld = L / 4, C = (Ld - 1 + 63) / Ld + 1; const size_t fl = (size_t)C * 96 * 4 * Ld * 4;
      if (fl <= 160 * 1024) { run("F collision-free J8", vF<8>, fl, 256); run("F collision-free J16", vF<16>, fl, 256); run("F collision-free J24", vF<24>, fl, 256); } }
    if (L % 4 == 0) { const size_t hl = 4 * 128 * 64 * 4;
      if (L <= 64) { run("H group16 J8", vH<16, 8>, hl, 256); run("H group16 J16", vH<16, 16>, hl, 256); }
      if (L <= 128) { run("H group32 J8", vH<32, 8>, hl, 256); run("H group32 J16", vH<32, 16>, hl, 256); run("H group32 J24", vH<32, 24>, hl, 256); } }
    if (L % 4 == 0) { const int QP = (L / 4 + 31 + 7) & ~7; run("E group-run J8", vE<8>, (size_t)128 * 4 * QP * 4, 256); run("E group-run J16", vE<16>, (size_t)128 * 4 * QP * 4, 256); }
  } else if (L <= 256) {
    if (L % 4 == 0) { const size_t hl = 4 * 128 * 64 * 4; run("H group64 J8", vH<64, 8>, hl, 256); run("H group64 J16", vH<64, 16>, hl, 256); }
    if (L % 4 == 0) { run("D cf LP256 J8 A4", vD<256, 8, true>, 128 * 256 * 4, 256); run("D cf LP256 J16 A4", vD<256, 16, true>, 128 * 256 * 4, 256); }
    run("D cf LP256 J8 gen", vD<256, 8, false>, 128 * 256 * 4, 256); run("D cf LP256 J16 gen", vD<256, 16, false>, 128 * 256 * 4, 256);
  }
  return 0;
}
