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
  run("A v1 wave/read", vA, (size_t)L * 257 * 4, 256);
  run("R read-only J1", vR<1>, 0, 256);
  run("R read-only J4", vR<4>, 0, 256);
  run("R read-only J4 x2", vR<4>, 0, 512);
  run("B reg-direct J1 s257", vB<1, 0>, (size_t)L * 257 * 4, 256);
  run("B reg-direct J4 s257", vB<4, 0>, (size_t)L * 257 * 4, 256);
  if (L <= 128) {
    if (L % 4 == 0) { run("D cf LP128 J4 A4", vD<128, 4, true>, 128 * 128 * 4, 256); run("D cf LP128 J8 A4", vD<128, 8, true>, 128 * 128 * 4, 256);
      run("D cf LP128 J16 A4", vD<128, 16, true>, 128 * 128 * 4, 256); run("D cf LP128 J8 A4 x2", vD<128, 8, true>, 128 * 128 * 4, 512); }
    run("D cf LP128 J8 gen", vD<128, 8, false>, 128 * 128 * 4, 256); run("D cf LP128 J16 gen", vD<128, 16, false>, 128 * 128 * 4, 256);
    if (L % 4 == 0) { const int QP = (L / 4 + 31 + 7) & ~7; run("E group-run J8", vE<8>, (size_t)128 * 4 * QP * 4, 256); run("E group-run J16", vE<16>, (size_t)128 * 4 * QP * 4, 256); }
  } else if (L <= 256) {
    if (L % 4 == 0) { run("D cf LP256 J8 A4", vD<256, 8, true>, 128 * 256 * 4, 256); run("D cf LP256 J16 A4", vD<256, 16, true>, 128 * 256 * 4, 256); }
    run("D cf LP256 J8 gen", vD<256, 8, false>, 128 * 256 * 4, 256); run("D cf LP256 J16 gen", vD<256, 16, false>, 128 * 256 * 4, 256);
  }
  return 0;
}
