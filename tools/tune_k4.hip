// tune_k4.hip -- developer harness (NOT part of the product): A/B variants of the config-4 kernel on one GPU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tune_k4.hip -o gpurun_out/tune_k4   run: ./tune_k4 [rows]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
__host__ __device__ inline uint64_t rnd(uint64_t seed, uint64_t col, uint64_t i) { return mix64(seed + col * 0xD1B54A32D192ED03ULL + (i + 1) * 0x9E3779B97F4A7C15ULL); }
static inline uint32_t pct_thr(int p) { return (uint32_t)((((uint64_t)p) << 32) / 100); }

__global__ void gen(uint64_t seed, int64_t n, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, float* af, uint8_t* av, float* qual, uint8_t* qv, int32_t* fid) {
  int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; int lane = threadIdx.x & 63; bool a = false, q = false;
  if (k < n) { uint64_t r0 = rnd(seed,0,k), r1 = rnd(seed,1,k), r2 = rnd(seed,2,k);
    uint32_t e = (uint32_t)((((r0 >> 23) & 0xFF) * 14) >> 8); uint32_t bits = ((126u - e) << 23) | (uint32_t)(r0 & 0x7FFFFF);
    float x = __uint_as_float(bits); if (((r0 >> 31) & 0x3FF) == 0) x = 0.01f; af[k] = x; a = (r0 >> 44) >= 10486;
    qual[k] = (float)((double)((uint32_t)r1 % 10000u) / 10.0); q = (r1 >> 44) >= 31457; uint32_t u = (uint32_t)(r2 >> 32);
    fid[k] = (u >= t0) + (u >= t1) + (u >= t2) + (u >= t3); }
  unsigned long long ma = __ballot(a), mq = __ballot(q); int64_t w0 = k - lane;
  if (lane < 8) { int64_t r = w0 + lane * 8; if (r < n) { av[r >> 3] = (uint8_t)(ma >> (lane * 8)); qv[r >> 3] = (uint8_t)(mq >> (lane * 8)); } }
}

constexpr int G = 5;
struct Out { unsigned long long cnn[G], crow[G]; double sum[G]; };

template <typename T> __device__ __forceinline__ T ld16(const void* p) { return *reinterpret_cast<const T*>(p); }
typedef int v4i_t __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ T ld16nt(const void* p) { v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p)); T r; __builtin_memcpy(&r, &v, 16); return r; }
__device__ __forceinline__ int32_t f32_key(float f) { int32_t b = __float_as_int(f); return b ^ ((b >> 31) & 0x7FFFFFFF); }
__device__ __forceinline__ unsigned long long wsum(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }
__device__ __forceinline__ double wsum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }

template <int THREADS>
__device__ __forceinline__ void block_out(const unsigned* cnn, const unsigned* crow, const double* sum, Out* out) {
  // atomics are fine for the harness (order-insensitive check uses tolerance)
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < G; ++k) { unsigned long long a = wsum((unsigned long long)cnn[k]), b = wsum((unsigned long long)crow[k]); double c = wsum(sum[k]);
    if (lane == 0) { atomicAdd(&out->cnn[k], a); atomicAdd(&out->crow[k], b); atomicAdd(&out->sum[k], c); } }
}

__device__ __forceinline__ unsigned valid8(const uint8_t* bm, int64_t wbase, int lane) {
  const uint8_t* p = bm + (wbase >> 3) + (lane >> 1); const unsigned sh = (lane & 1) * 4;
  return ((unsigned(p[0]) >> sh) & 0xFu) | (((unsigned(p[32]) >> sh) & 0xFu) << 4);
}

// ---- V0: the shipped round-1 kernel body ------------------------------------------------------------
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void v0(const float* x, const uint8_t* xv, const float* y, const uint8_t* yv, const int32_t* gid, int64_t n, int32_t klo, int32_t khi, Out* out) {
  constexpr int TILE = THREADS * 8; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double sum[G]; unsigned cnn[G], crow[G]; for (int k = 0; k < G; ++k) { sum[k] = 0; cnn[k] = 0; crow[k] = 0; }
  auto row = [&](float xf, float yf, int32_t g, unsigned xvb, unsigned yvb) {
    const int32_t kx = f32_key(xf); const unsigned pass = xvb & unsigned(kx >= klo) & unsigned(kx <= khi); const unsigned yq = pass & yvb; const double yd = (double)yf;
#pragma unroll
    for (int k = 0; k < G; ++k) { const unsigned m = unsigned(g == k); crow[k] += pass & m; cnn[k] += yq & m; sum[k] += (yq & m) ? yd : 0.0; } };
  const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * 512; const int64_t r0 = wbase + lane * 4, r1 = r0 + 256;
    float4 x0, x1, y0, y1; int4 g0, g1;
    if (NT) { x0 = ld16nt<float4>(x + r0); x1 = ld16nt<float4>(x + r1); y0 = ld16nt<float4>(y + r0); y1 = ld16nt<float4>(y + r1); g0 = ld16nt<int4>(gid + r0); g1 = ld16nt<int4>(gid + r1); }
    else { x0 = ld16<float4>(x + r0); x1 = ld16<float4>(x + r1); y0 = ld16<float4>(y + r0); y1 = ld16<float4>(y + r1); g0 = ld16<int4>(gid + r0); g1 = ld16<int4>(gid + r1); }
    const unsigned xm = valid8(xv, wbase, lane), ym = valid8(yv, wbase, lane);
    row(x0.x, y0.x, g0.x, xm >> 0 & 1, ym >> 0 & 1); row(x0.y, y0.y, g0.y, xm >> 1 & 1, ym >> 1 & 1); row(x0.z, y0.z, g0.z, xm >> 2 & 1, ym >> 2 & 1); row(x0.w, y0.w, g0.w, xm >> 3 & 1, ym >> 3 & 1);
    row(x1.x, y1.x, g1.x, xm >> 4 & 1, ym >> 4 & 1); row(x1.y, y1.y, g1.y, xm >> 5 & 1, ym >> 5 & 1); row(x1.z, y1.z, g1.z, xm >> 6 & 1, ym >> 6 & 1); row(x1.w, y1.w, g1.w, xm >> 7 & 1, ym >> 7 & 1);
  }
  block_out<THREADS>(cnn, crow, sum, out);
}

// ---- V1: read-only ceiling (same loads, trivial use) -------------------------------------------------
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void v1(const float* x, const uint8_t* xv, const float* y, const uint8_t* yv, const int32_t* gid, int64_t n, int32_t klo, int32_t khi, Out* out) {
  constexpr int TILE = THREADS * 8; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0; const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * 512; const int64_t r0 = wbase + lane * 4, r1 = r0 + 256;
    int4 a, b, c, d, e, f;
    if (NT) { a = ld16nt<int4>(x + r0); b = ld16nt<int4>(x + r1); c = ld16nt<int4>(y + r0); d = ld16nt<int4>(y + r1); e = ld16nt<int4>(gid + r0); f = ld16nt<int4>(gid + r1); }
    else { a = ld16<int4>(x + r0); b = ld16<int4>(x + r1); c = ld16<int4>(y + r0); d = ld16<int4>(y + r1); e = ld16<int4>(gid + r0); f = ld16<int4>(gid + r1); }
    const unsigned xm = valid8(xv, wbase, lane), ym = valid8(yv, wbase, lane);
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w ^ e.x ^ e.y ^ e.z ^ e.w ^ f.x ^ f.y ^ f.z ^ f.w ^ xm ^ ym;
  }
  if (acc == 0x12345678u) atomicAdd(&out->cnn[0], 1ull);
}

// ---- V2: reduced VALU: packed byte counters, group id folded with the predicate ------------------------
template <int THREADS, bool NT, bool PREFETCH>
__global__ __launch_bounds__(THREADS) void v2(const float* x, const uint8_t* xv, const float* y, const uint8_t* yv, const int32_t* gid, int64_t n, int32_t klo, int32_t khi, Out* out) {
  constexpr int TILE = THREADS * 8; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double sum[G]; unsigned cnn[G], crow[G]; for (int k = 0; k < G; ++k) { sum[k] = 0; cnn[k] = 0; crow[k] = 0; }
  unsigned long long prow = 0, pnn = 0;  // 8 x 8-bit fields, field 7.. unused; dummy group 7 absorbs failing rows
  auto row = [&](float xf, float yf, int32_t g, unsigned xvb, unsigned yvb) {
    const int32_t kx = f32_key(xf); const bool pass = xvb && (kx >= klo) && (kx <= khi); const bool yq = pass && yvb;
    const unsigned gp = pass ? (unsigned)g : 7u, gq = yq ? (unsigned)g : 7u;
    prow += 1ull << (gp * 8); pnn += 1ull << (gq * 8);
    const double yd = (double)yf;
#pragma unroll
    for (int k = 0; k < G; ++k) sum[k] += (gq == (unsigned)k) ? yd : 0.0; };
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < G; ++k) { crow[k] += (unsigned)(prow >> (8 * k)) & 0xFF; cnn[k] += (unsigned)(pnn >> (8 * k)) & 0xFF; } prow = 0; pnn = 0; };
  const int64_t ntiles = n / TILE; int it = 0;
  float4 x0, x1, y0, y1; int4 g0, g1; unsigned xm, ym;
  auto load = [&](int64_t tile) { const int64_t wbase = tile * TILE + (int64_t)wave * 512; const int64_t r0 = wbase + lane * 4, r1 = r0 + 256;
    if (NT) { x0 = ld16nt<float4>(x + r0); x1 = ld16nt<float4>(x + r1); y0 = ld16nt<float4>(y + r0); y1 = ld16nt<float4>(y + r1); g0 = ld16nt<int4>(gid + r0); g1 = ld16nt<int4>(gid + r1); }
    else { x0 = ld16<float4>(x + r0); x1 = ld16<float4>(x + r1); y0 = ld16<float4>(y + r0); y1 = ld16<float4>(y + r1); g0 = ld16<int4>(gid + r0); g1 = ld16<int4>(gid + r1); }
    xm = valid8(xv, wbase, lane); ym = valid8(yv, wbase, lane); };
  int64_t tile = blockIdx.x;
  if (PREFETCH) { if (tile < ntiles) load(tile); }
  for (; tile < ntiles; tile += gridDim.x) {
    if (!PREFETCH) load(tile);
    const float4 cx0 = x0, cx1 = x1, cy0 = y0, cy1 = y1; const int4 cg0 = g0, cg1 = g1; const unsigned cxm = xm, cym = ym;
    if (PREFETCH) { if (tile + gridDim.x < ntiles) load(tile + gridDim.x); }
    row(cx0.x, cy0.x, cg0.x, cxm >> 0 & 1, cym >> 0 & 1); row(cx0.y, cy0.y, cg0.y, cxm >> 1 & 1, cym >> 1 & 1); row(cx0.z, cy0.z, cg0.z, cxm >> 2 & 1, cym >> 2 & 1); row(cx0.w, cy0.w, cg0.w, cxm >> 3 & 1, cym >> 3 & 1);
    row(cx1.x, cy1.x, cg1.x, cxm >> 4 & 1, cym >> 4 & 1); row(cx1.y, cy1.y, cg1.y, cxm >> 5 & 1, cym >> 5 & 1); row(cx1.z, cy1.z, cg1.z, cxm >> 6 & 1, cym >> 6 & 1); row(cx1.w, cy1.w, cg1.w, cxm >> 7 & 1, cym >> 7 & 1);
    if (++it == 31) { flush(); it = 0; }
  }
  flush();
  block_out<THREADS>(cnn, crow, sum, out);
}

// ---- V3: generalised: J sub-tiles of 4 rows per lane, packed counters, optional nt / prefetch -----------
template <int THREADS, int J, bool NT, bool PREFETCH, bool CHUNKED>
__global__ __launch_bounds__(THREADS) void v3(const float* x, const uint8_t* xv, const float* y, const uint8_t* yv, const int32_t* gid, int64_t n, int32_t klo, int32_t khi, Out* out) {
  constexpr int WT = 256 * J; constexpr int TILE = (THREADS / 64) * WT; const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double sum[G]; unsigned cnn[G], crow[G]; for (int k = 0; k < G; ++k) { sum[k] = 0; cnn[k] = 0; crow[k] = 0; }
  unsigned long long prow = 0, pnn = 0;
  auto row = [&](float xf, float yf, int32_t g, unsigned xvb, unsigned yvb) {
    const int32_t kx = f32_key(xf); const bool pass = xvb && (kx >= klo) && (kx <= khi); const bool yq = pass && yvb;
    const unsigned gp = pass ? (unsigned)g : 7u, gq = yq ? (unsigned)g : 7u;
    prow += 1ull << (gp * 8); pnn += 1ull << (gq * 8);
    const double yd = (double)yf;
#pragma unroll
    for (int k = 0; k < G; ++k) sum[k] += (gq == (unsigned)k) ? yd : 0.0; };
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < G; ++k) { crow[k] += (unsigned)(prow >> (8 * k)) & 0xFF; cnn[k] += (unsigned)(pnn >> (8 * k)) & 0xFF; } prow = 0; pnn = 0; };
  const int64_t ntiles = n / TILE; int it = 0;
  float4 xs[J], ys[J]; int4 gs[J]; unsigned xm[J], ym[J];
  auto load = [&](int64_t tile) { const int64_t wbase = tile * TILE + (int64_t)wave * WT;
#pragma unroll
    for (int j = 0; j < J; ++j) { const int64_t r = wbase + j * 256 + lane * 4;
      if (NT) { xs[j] = ld16nt<float4>(x + r); ys[j] = ld16nt<float4>(y + r); gs[j] = ld16nt<int4>(gid + r); }
      else { xs[j] = ld16<float4>(x + r); ys[j] = ld16<float4>(y + r); gs[j] = ld16<int4>(gid + r); }
      const int64_t bo = (wbase >> 3) + j * 32 + (lane >> 1); const unsigned sh = (lane & 1) * 4;
      xm[j] = (unsigned(xv[bo]) >> sh) & 0xF; ym[j] = (unsigned(yv[bo]) >> sh) & 0xF; } };
  int64_t t0, t1, ts;
  if (CHUNKED) { const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x; t0 = blockIdx.x * per; t1 = t0 + per < ntiles ? t0 + per : ntiles; ts = 1; }
  else { t0 = blockIdx.x; t1 = ntiles; ts = gridDim.x; }
  int64_t tile = t0;
  if (PREFETCH) { if (tile < t1) load(tile); }
  for (; tile < t1; tile += ts) {
    if (!PREFETCH) load(tile);
    float4 cx[J], cy[J]; int4 cg[J]; unsigned cxm[J], cym[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { cx[j] = xs[j]; cy[j] = ys[j]; cg[j] = gs[j]; cxm[j] = xm[j]; cym[j] = ym[j]; }
    if (PREFETCH) { if (tile + ts < t1) load(tile + ts); }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      row(cx[j].x, cy[j].x, cg[j].x, cxm[j] >> 0 & 1, cym[j] >> 0 & 1); row(cx[j].y, cy[j].y, cg[j].y, cxm[j] >> 1 & 1, cym[j] >> 1 & 1);
      row(cx[j].z, cy[j].z, cg[j].z, cxm[j] >> 2 & 1, cym[j] >> 2 & 1); row(cx[j].w, cy[j].w, cg[j].w, cxm[j] >> 3 & 1, cym[j] >> 3 & 1); }
    it += J; if (it >= 60) { flush(); it = 0; }
  }
  flush();
  block_out<THREADS>(cnn, crow, sum, out);
}

static int32_t h_key(float f) { int32_t b; memcpy(&b, &f, 4); return b ^ ((b >> 31) & 0x7FFFFFFF); }

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? (int64_t)atof(argv[1]) : (int64_t)1e9; n = n / 8192 * 8192;
  float *af, *qual; int32_t* fid; uint8_t *av, *qv; Out* out;
  CK(hipMalloc(&af, n * 4)); CK(hipMalloc(&qual, n * 4)); CK(hipMalloc(&fid, n * 4)); CK(hipMalloc(&av, n / 8 + 64)); CK(hipMalloc(&qv, n / 8 + 64)); CK(hipMalloc(&out, sizeof(Out)));
  hipLaunchKernelGGL(gen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, 4ull, n, pct_thr(85), pct_thr(90), pct_thr(96), pct_thr(99), af, av, qual, qv, fid);
  CK(hipDeviceSynchronize());
  // f32 key range for "> 0.01": first f32 whose widening exceeds 0.01
  float t = 0.01f; if ((double)t <= 0.01) t = nextafterf(t, 1.0f);
  const int32_t klo = h_key(t), khi = INT32_MAX;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Out ref; bool have_ref = false;
  auto run = [&](const char* name, auto kern, int threads, int bpc) {
    int grid = 256 * bpc; float best = 1e9, tot = 0; const int reps = 8; Out h;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipMemsetAsync(out, 0, sizeof(Out), 0)); CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, af, av, qual, qv, fid, n, klo, khi, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    CK(hipMemcpy(&h, out, sizeof(Out), hipMemcpyDeviceToHost));
    const char* ok = "";
    if (!have_ref) { ref = h; have_ref = true; ok = "ref"; }
    else { bool same = true; for (int k = 0; k < G; ++k) same = same && h.cnn[k] == ref.cnn[k] && h.crow[k] == ref.crow[k] && fabs(h.sum[k] - ref.sum[k]) <= 1e-9 * fabs(ref.sum[k]); ok = same ? "match" : "MISMATCH"; }
    double gbs = n * 12.25 / (tot / reps * 1e-3) / 1e9;
    printf("%-28s thr=%4d bpc=%2d  avg %.3f ms  best %.3f ms  %.0f GB/s (%.1f%% of 8TB/s)  %s\n", name, threads, bpc, tot / reps, best, gbs, gbs / 80.0, ok);
  };
  run("v0 shipped", v0<256, false>, 256, 6);
  run("v0 shipped", v0<256, false>, 256, 8);
  run("v1 read-only nt 1024", v1<1024, true>, 1024, 2);
  run("v1 read-only nt 512", v1<512, true>, 512, 4);
  run("v3 256 J2", v3<256, 2, false, false, false>, 256, 6);
  run("v3 256 J2 nt", v3<256, 2, true, false, false>, 256, 6);
  run("v3 512 J2 nt", v3<512, 2, true, false, false>, 512, 3);
  run("v3 1024 J2 nt", v3<1024, 2, true, false, false>, 1024, 1);
  run("v3 1024 J2 nt", v3<1024, 2, true, false, false>, 1024, 2);
  run("v3 512 J2 nt pf", v3<512, 2, true, true, false>, 512, 2);
  run("v3 1024 J2 nt pf", v3<1024, 2, true, true, false>, 1024, 1);
  run("v3 256 J4 nt", v3<256, 4, true, false, false>, 256, 4);
  run("v3 512 J4 nt", v3<512, 4, true, false, false>, 512, 2);
  run("v3 1024 J4 nt", v3<1024, 4, true, false, false>, 1024, 1);
  run("v3 256 J1 nt", v3<256, 1, true, false, false>, 256, 8);
  run("v3 512 J1 nt", v3<512, 1, true, false, false>, 512, 4);
  run("v3 1024 J1 nt", v3<1024, 1, true, false, false>, 1024, 2);
  run("v3 512 J2 nt chunk", v3<512, 2, true, false, true>, 512, 3);
  run("v3 1024 J2 nt chunk", v3<1024, 2, true, false, true>, 1024, 2);
  run("v3 512 J2 nt pf chunk", v3<512, 2, true, true, true>, 512, 2);
  return 0;
}
