// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the scan -> filter -> aggregate hot path.
//
// All four operators are HBM-read bound (arithmetic intensity < 0.5 op/B), so there is no MFMA here.
// Shape shared by every main kernel:
//   * persistent grid (compute_units x blocks_per_cu workgroups of 256 threads = 4 waves), grid-stride
//     over 2048-row tiles; inside a tile each wave owns 512 rows and every global load instruction is
//     a fully coalesced 16 B/lane (1 KiB per wave) access;
//   * Arrow validity bitmaps are consumed as one byte per lane-pair (a nibble per lane per 4 rows);
//   * per-lane register accumulators -> wave shuffle reduction -> LDS across the 4 waves -> one
//     partial record per workgroup in the workspace (no global atomics);
//   * `finalize_partials` folds the per-workgroup records into the caller's running state in a fixed
//     order, so f64 sums are bit-reproducible for a given launch shape.
// Reference semantics restated by each kernel are cited at the kernel.
#include "kernels.h"

#include <algorithm>
#include <cstring>

namespace exon {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int ROWS_PER_LANE = 8;
constexpr int TILE = THREADS * ROWS_PER_LANE;  // 2048 rows per workgroup iteration
constexpr int WAVE_TILE = 64 * ROWS_PER_LANE;  // 512 rows per wave iteration

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T ld16(const void* p) {
  return *reinterpret_cast<const T*>(p);
}

// validity bits of this lane's 8 rows: rows [wbase + 4*lane, +4) -> bits 0..3, rows [wbase + 256 + 4*lane, +4)
// -> bits 4..7.  wbase is a multiple of 512, so the wave's 512 validity bits are 64 consecutive bytes.
__device__ __forceinline__ unsigned valid8(const uint8_t* __restrict__ bm, int64_t wbase, int lane) {
  if (bm == nullptr) return 0xFFu;
  const uint8_t* p = bm + (wbase >> 3) + (lane >> 1);
  const unsigned sh = (lane & 1) * 4;
  return ((unsigned(p[0]) >> sh) & 0xFu) | (((unsigned(p[32]) >> sh) & 0xFu) << 4);
}
__device__ __forceinline__ bool valid1(const uint8_t* __restrict__ bm, int64_t r) {
  return bm == nullptr ? true : ((bm[r >> 3] >> (r & 7)) & 1);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// total-order key of an f32 (IEEE 754 totalOrder, what arrow-rs `cmp` kernels use for floats):
// signed-int comparison of the keys == total_cmp of the floats.
__device__ __forceinline__ int32_t f32_key(float f) {
  int32_t b = __float_as_int(f);
  return b ^ ((b >> 31) & 0x7FFFFFFF);
}

// ------------------------------------------------------------------------------------------------
// finalize: state[v] += sum over workgroups b (fixed order) of partials[b][v]
// words [0, n_i64) are int64 counters, words [n_i64, V) are float64 sums.
// grid = ceil(V / 32), block = 256 = 32 values x 8 segments of the workgroup range.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_partials(const unsigned long long* __restrict__ partials, int nblocks,
                                                         int V, int n_i64, int64_t* __restrict__ st_i64,
                                                         double* __restrict__ st_f64) {
  __shared__ unsigned long long red[8][32];
  const int vi = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const int v = blockIdx.x * 32 + vi;
  const int per = (nblocks + 7) / 8;
  const int b0 = seg * per, b1 = min(nblocks, b0 + per);
  unsigned long long acc_i = 0;
  double acc_f = 0.0;
  if (v < V) {
    if (v < n_i64) {
      for (int b = b0; b < b1; ++b) acc_i += partials[(size_t)b * V + v];
    } else {
      for (int b = b0; b < b1; ++b) acc_f += __longlong_as_double((long long)partials[(size_t)b * V + v]);
    }
  }
  red[seg][vi] = (v < n_i64) ? acc_i : (unsigned long long)__double_as_longlong(acc_f);
  __syncthreads();
  if (seg == 0 && v < V) {
    if (v < n_i64) {
      unsigned long long t = 0;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[s][vi];
      st_i64[v] += (int64_t)t;
    } else {
      double t = 0.0;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += __longlong_as_double((long long)red[s][vi]);
      st_f64[v - n_i64] += t;
    }
  }
}

static hipError_t run_finalize(hipStream_t s, const Workspace& ws, int nblocks, int V, int n_i64, int64_t* st_i64,
                               double* st_f64) {
  const int grid = (V + 31) / 32;
  hipLaunchKernelGGL(finalize_partials, dim3(grid), dim3(256), 0, s, ws.partials, nblocks, V, n_i64, st_i64, st_f64);
  return hipGetLastError();
}

// resident workgroups per CU of kernel `f` (the persistent grid must not exceed what is co-resident,
// otherwise the surplus workgroups run as a second, badly balanced round)
template <typename F>
static int resident_blocks(F f, int threads, size_t lds) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, threads, lds) != hipSuccess || nb < 1) nb = 1;
  return nb;
}

static int grid_for(const LaunchCfg& cfg, int64_t n, int resident) {
  int64_t tiles = (n + TILE - 1) / TILE;
  int64_t g = (int64_t)cfg.compute_units * std::min(cfg.blocks_per_cu, resident);
  if (g > tiles) g = tiles;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------
// K2 region_count
//   chrom = lit AND pos >= a AND pos <= b, Kleene AND, FilterExec keeps TRUE, COUNT(*)
//   (exon-core/src/physical_plan/region_physical_expr.rs:220-240; interval test of
//    exon-vcf/src/indexed_async_batch_stream.rs:99-116).  12 B/row: i32 chrom id + i64 pos.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned k2_row(int32_t c, int64_t p, unsigned cv, unsigned pv, int32_t id, int64_t a,
                                           int64_t b) {
  return (cv & pv) & unsigned(c == id) & unsigned(p >= a) & unsigned(p <= b);
}

__global__ __launch_bounds__(THREADS) void k2_region_count_main(const int32_t* __restrict__ chrom,
                                                                const uint8_t* __restrict__ cvalid,
                                                                const int64_t* __restrict__ pos,
                                                                const uint8_t* __restrict__ pvalid, int64_t n,
                                                                int32_t id, int64_t a, int64_t b,
                                                                unsigned long long* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned cnt = 0;
  const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WAVE_TILE;
    const int64_t r0 = wbase + lane * 4, r1 = r0 + 256;
    const int4 c0 = ld16<int4>(chrom + r0), c1 = ld16<int4>(chrom + r1);
    const longlong2 p00 = ld16<longlong2>(pos + r0), p01 = ld16<longlong2>(pos + r0 + 2);
    const longlong2 p10 = ld16<longlong2>(pos + r1), p11 = ld16<longlong2>(pos + r1 + 2);
    const unsigned cm = valid8(cvalid, wbase, lane), pm = valid8(pvalid, wbase, lane);
    cnt += k2_row(c0.x, p00.x, cm >> 0 & 1, pm >> 0 & 1, id, a, b);
    cnt += k2_row(c0.y, p00.y, cm >> 1 & 1, pm >> 1 & 1, id, a, b);
    cnt += k2_row(c0.z, p01.x, cm >> 2 & 1, pm >> 2 & 1, id, a, b);
    cnt += k2_row(c0.w, p01.y, cm >> 3 & 1, pm >> 3 & 1, id, a, b);
    cnt += k2_row(c1.x, p10.x, cm >> 4 & 1, pm >> 4 & 1, id, a, b);
    cnt += k2_row(c1.y, p10.y, cm >> 5 & 1, pm >> 5 & 1, id, a, b);
    cnt += k2_row(c1.z, p11.x, cm >> 6 & 1, pm >> 6 & 1, id, a, b);
    cnt += k2_row(c1.w, p11.y, cm >> 7 & 1, pm >> 7 & 1, id, a, b);
  }
  for (int64_t r = ntiles * TILE + (int64_t)blockIdx.x * THREADS + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * THREADS)
    cnt += k2_row(chrom[r], pos[r], valid1(cvalid, r), valid1(pvalid, r), id, a, b);

  __shared__ unsigned long long red[WAVES];
  const unsigned long long w = wave_sum((unsigned long long)cnt);
  if (lane == 0) red[wave] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) t += red[i];
    partials[blockIdx.x] = t;
  }
}

size_t k2_partial_words(const LaunchCfg& cfg) { return (size_t)cfg.compute_units * cfg.blocks_per_cu; }

hipError_t launch_region_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* chrom,
                               const uint8_t* chrom_valid, const int64_t* pos, const uint8_t* pos_valid, int64_t n,
                               int32_t region_chrom, int64_t start, int64_t end, int64_t* d_count) {
  if (n <= 0) return hipSuccess;
  static const int resident = resident_blocks(k2_region_count_main, THREADS, 0);
  const int grid = grid_for(cfg, n, resident);
  hipLaunchKernelGGL(k2_region_count_main, dim3(grid), dim3(THREADS), 0, s, chrom, chrom_valid, pos, pos_valid, n,
                     region_chrom, start, end, ws.partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return run_finalize(s, ws, grid, 1, 1, d_count, nullptr);
}

// ------------------------------------------------------------------------------------------------
// K3 flag_mapq_group_count
//   WHERE (flag & M) = V AND CAST(mapping_quality AS INT) >= Q  GROUP BY reference  COUNT(*)
//   flag test = sam_flag_function (exon-core/src/udfs/sam/samflags.rs:26-47); mapq NULL when 255
//   (exon-bam/src/array_builder.rs:136-143) -> NULL predicate -> row dropped; NULL reference is its own
//   group (index n_refs).  9.25 B/row: i32 flag + u8 mapq + i32 ref id + 2 validity bits.
//   Group table: one u32[n_refs+1] table per wave in LDS, LDS atomics, folded per workgroup.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void k3_flag_mapq_group_count_main(
    const int32_t* __restrict__ flag, const uint8_t* __restrict__ fvalid, const uint8_t* __restrict__ mapq,
    const uint8_t* __restrict__ mvalid, const int32_t* __restrict__ ref, const uint8_t* __restrict__ rvalid,
    int64_t n, int32_t mask, int32_t value, int32_t qmin, int32_t R, unsigned long long* __restrict__ partials,
    int* __restrict__ status) {
  extern __shared__ unsigned k3_tbl[];  // [WAVES][R+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V = R + 1;
  for (int i = threadIdx.x; i < WAVES * V; i += THREADS) k3_tbl[i] = 0;
  __syncthreads();
  unsigned* mine = k3_tbl + wave * V;
  unsigned bad = 0;

  auto row = [&](int32_t f, unsigned q, int32_t r, unsigned fv, unsigned mv, unsigned rv) {
    const bool pass = fv && ((f & mask) == value) && mv && ((int32_t)q >= qmin);
    if (pass) {
      const unsigned key = rv ? (unsigned)r : (unsigned)R;
      if (key <= (unsigned)R) atomicAdd(&mine[key], 1u);
      else bad = 1;
    }
  };

  const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WAVE_TILE;
    const int64_t r0 = wbase + lane * 4, r1 = r0 + 256;
    const int4 f0 = ld16<int4>(flag + r0), f1 = ld16<int4>(flag + r1);
    const int4 g0 = ld16<int4>(ref + r0), g1 = ld16<int4>(ref + r1);
    const unsigned q0 = *reinterpret_cast<const unsigned*>(mapq + r0);
    const unsigned q1 = *reinterpret_cast<const unsigned*>(mapq + r1);
    const unsigned fm = valid8(fvalid, wbase, lane), mm = valid8(mvalid, wbase, lane),
                   rm = valid8(rvalid, wbase, lane);
    row(f0.x, q0 & 0xFF, g0.x, fm >> 0 & 1, mm >> 0 & 1, rm >> 0 & 1);
    row(f0.y, q0 >> 8 & 0xFF, g0.y, fm >> 1 & 1, mm >> 1 & 1, rm >> 1 & 1);
    row(f0.z, q0 >> 16 & 0xFF, g0.z, fm >> 2 & 1, mm >> 2 & 1, rm >> 2 & 1);
    row(f0.w, q0 >> 24, g0.w, fm >> 3 & 1, mm >> 3 & 1, rm >> 3 & 1);
    row(f1.x, q1 & 0xFF, g1.x, fm >> 4 & 1, mm >> 4 & 1, rm >> 4 & 1);
    row(f1.y, q1 >> 8 & 0xFF, g1.y, fm >> 5 & 1, mm >> 5 & 1, rm >> 5 & 1);
    row(f1.z, q1 >> 16 & 0xFF, g1.z, fm >> 6 & 1, mm >> 6 & 1, rm >> 6 & 1);
    row(f1.w, q1 >> 24, g1.w, fm >> 7 & 1, mm >> 7 & 1, rm >> 7 & 1);
  }
  for (int64_t r = ntiles * TILE + (int64_t)blockIdx.x * THREADS + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * THREADS)
    row(flag[r], mapq[r], ref[r], valid1(fvalid, r), valid1(mvalid, r), valid1(rvalid, r));

  if (bad) atomicOr(status, 2);
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += THREADS) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += k3_tbl[w * V + v];
    partials[(size_t)blockIdx.x * V + v] = t;
  }
}

size_t k3_partial_words(const LaunchCfg& cfg, int n_refs) {
  return (size_t)cfg.compute_units * cfg.blocks_per_cu * (size_t)(n_refs + 1);
}

hipError_t launch_flag_mapq_group_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* flag,
                                        const uint8_t* flag_valid, const uint8_t* mapq, const uint8_t* mapq_valid,
                                        const int32_t* ref_id, const uint8_t* ref_valid, int64_t n, int32_t flag_mask,
                                        int32_t flag_value, int32_t mapq_min, int32_t n_refs, int64_t* d_counts) {
  if (n <= 0) return hipSuccess;
  const int V = n_refs + 1;
  const size_t lds = (size_t)WAVES * V * sizeof(unsigned);
  const int grid = grid_for(cfg, n, resident_blocks(k3_flag_mapq_group_count_main, THREADS, lds));
  hipLaunchKernelGGL(k3_flag_mapq_group_count_main, dim3(grid), dim3(THREADS), lds, s, flag, flag_valid, mapq,
                     mapq_valid, ref_id, ref_valid, n, flag_mask, flag_value, mapq_min, n_refs, ws.partials,
                     ws.status);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return run_finalize(s, ws, grid, V, V, d_counts, nullptr);
}

// ------------------------------------------------------------------------------------------------
// K4 cmp_avg_by_group
//   WHERE CAST(x AS DOUBLE) <op> thr   SELECT g, AVG(y), COUNT(*) GROUP BY g      (x, y Float32)
//   DataFusion coerces Float32-vs-Float64-literal comparisons to Float64 and arrow-rs compares floats
//   in IEEE totalOrder.  f32 -> f64 widening is monotone under totalOrder, so the host folds
//   (<op>, thr) into an inclusive range [klo, khi] of f32 totalOrder keys (complemented for !=); the
//   kernel does two 32-bit integer compares per row and is exact (parity-tested against the oracle's
//   plain f64 comparison, including f32(0.01) which widens to 0.00999999977 < 0.01).
//   AVG state = f64 sum + count of non-null y (Float32 widened to Float64 before the add).
//   12.25 B/row: f32 x + f32 y + i32 group id + 2 validity bits.  Group ids < G <= 8 live in registers.
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(THREADS) void k4_cmp_avg_by_group_main(
    const float* __restrict__ x, const uint8_t* __restrict__ xvalid, const float* __restrict__ y,
    const uint8_t* __restrict__ yvalid, const int32_t* __restrict__ gid, int64_t n, int32_t klo, int32_t khi,
    int32_t negate, unsigned long long* __restrict__ partials, int* __restrict__ status) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double sum[G];
  unsigned cnn[G], crow[G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    sum[k] = 0.0;
    cnn[k] = 0;
    crow[k] = 0;
  }
  unsigned bad = 0;

  auto row = [&](float xf, float yf, int32_t g, unsigned xv, unsigned yv) {
    const int32_t kx = f32_key(xf);
    const unsigned inr = unsigned(kx >= klo) & unsigned(kx <= khi);
    const unsigned pass = xv & (inr ^ (unsigned)negate);
    const unsigned yq = pass & yv;
    const double yd = (double)yf;
    bad |= pass & unsigned((unsigned)g >= (unsigned)G);
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const unsigned m = unsigned(g == k);
      crow[k] += pass & m;
      cnn[k] += yq & m;
      sum[k] += (yq & m) ? yd : 0.0;
    }
  };

  const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WAVE_TILE;
    const int64_t r0 = wbase + lane * 4, r1 = r0 + 256;
    const float4 x0 = ld16<float4>(x + r0), x1 = ld16<float4>(x + r1);
    const float4 y0 = ld16<float4>(y + r0), y1 = ld16<float4>(y + r1);
    const int4 g0 = ld16<int4>(gid + r0), g1 = ld16<int4>(gid + r1);
    const unsigned xm = valid8(xvalid, wbase, lane), ym = valid8(yvalid, wbase, lane);
    row(x0.x, y0.x, g0.x, xm >> 0 & 1, ym >> 0 & 1);
    row(x0.y, y0.y, g0.y, xm >> 1 & 1, ym >> 1 & 1);
    row(x0.z, y0.z, g0.z, xm >> 2 & 1, ym >> 2 & 1);
    row(x0.w, y0.w, g0.w, xm >> 3 & 1, ym >> 3 & 1);
    row(x1.x, y1.x, g1.x, xm >> 4 & 1, ym >> 4 & 1);
    row(x1.y, y1.y, g1.y, xm >> 5 & 1, ym >> 5 & 1);
    row(x1.z, y1.z, g1.z, xm >> 6 & 1, ym >> 6 & 1);
    row(x1.w, y1.w, g1.w, xm >> 7 & 1, ym >> 7 & 1);
  }
  for (int64_t r = ntiles * TILE + (int64_t)blockIdx.x * THREADS + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * THREADS)
    row(x[r], y[r], gid[r], valid1(xvalid, r), valid1(yvalid, r));

  if (bad) atomicOr(status, 4);

  // per-workgroup record: [cnn[G]] [crow[G]] [sum[G]]  (fixed-order reductions)
  __shared__ unsigned long long red[WAVES][3 * G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const unsigned long long a = wave_sum((unsigned long long)cnn[k]);
    const unsigned long long b = wave_sum((unsigned long long)crow[k]);
    const double c = wave_sum(sum[k]);
    if (lane == 0) {
      red[wave][k] = a;
      red[wave][G + k] = b;
      red[wave][2 * G + k] = (unsigned long long)__double_as_longlong(c);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3 * G) {
    const int v = threadIdx.x;
    unsigned long long out;
    if (v < 2 * G) {
      out = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) out += red[w][v];
    } else {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) t += __longlong_as_double((long long)red[w][v]);
      out = (unsigned long long)__double_as_longlong(t);
    }
    partials[(size_t)blockIdx.x * (3 * G) + v] = out;
  }
}

// fold per-template-G records (padded to G) into the caller's n_groups-sized state
__global__ __launch_bounds__(256) void k4_finalize(const unsigned long long* __restrict__ partials, int nblocks, int G,
                                                   int n_groups, int64_t* __restrict__ counts,
                                                   double* __restrict__ sums) {
  // one wave per output word; lanes split the workgroup range, fixed-order shuffle tree
  const int lane = threadIdx.x & 63, word = blockIdx.x * (256 / 64) + (threadIdx.x >> 6);
  if (word >= 3 * n_groups) return;
  const int kind = word / n_groups, g = word % n_groups;  // 0: cnn, 1: crow, 2: sum
  const int v = kind * G + g;
  if (kind < 2) {
    unsigned long long acc = 0;
    for (int b = lane; b < nblocks; b += 64) acc += partials[(size_t)b * (3 * G) + v];
    acc = wave_sum(acc);
    if (lane == 0) counts[kind * n_groups + g] += (int64_t)acc;
  } else {
    double acc = 0.0;
    for (int b = lane; b < nblocks; b += 64) acc += __longlong_as_double((long long)partials[(size_t)b * (3 * G) + v]);
    acc = wave_sum(acc);
    if (lane == 0) sums[g] += acc;
  }
}

size_t k4_partial_words(const LaunchCfg& cfg, int n_groups) {
  (void)n_groups;
  return (size_t)cfg.compute_units * cfg.blocks_per_cu * 3 * 8;
}

template <int G>
static hipError_t k4_launch_g(hipStream_t s, const LaunchCfg& cfg, int* grid_out, const Workspace& ws, const float* x, const uint8_t* xv,
                              const float* y, const uint8_t* yv, const int32_t* gid, int64_t n, int32_t klo,
                              int32_t khi, int32_t negate) {
  static const int resident = resident_blocks(k4_cmp_avg_by_group_main<G>, THREADS, 0);
  const int grid = grid_for(cfg, n, resident);
  *grid_out = grid;
  hipLaunchKernelGGL(k4_cmp_avg_by_group_main<G>, dim3(grid), dim3(THREADS), 0, s, x, xv, y, yv, gid, n, klo, khi,
                     negate, ws.partials, ws.status);
  return hipGetLastError();
}

// ---- host: fold (<op>, thr) into an inclusive f32 totalOrder key range --------------------------
static inline int32_t h_f32_key(float f) {
  int32_t b;
  memcpy(&b, &f, 4);
  return b ^ ((b >> 31) & 0x7FFFFFFF);
}
static inline float h_key_f32(int32_t k) {
  int32_t b = k ^ ((k >> 31) & 0x7FFFFFFF);
  float f;
  memcpy(&f, &b, 4);
  return f;
}
static inline int64_t h_f64_key(double d) {
  int64_t b;
  memcpy(&b, &d, 8);
  return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}
// number of f32 keys k (as a count from INT32_MIN) with key64((double)f32(k)) < t  -> first key >= t
static int64_t first_key_not_less(int64_t t, bool strict_greater) {
  // smallest key k in [INT32_MIN, INT32_MAX+1] such that key64(widen(k)) >= t (or > t)
  int64_t lo = INT32_MIN, hi = (int64_t)INT32_MAX + 1;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    const int64_t km = h_f64_key((double)h_key_f32((int32_t)mid));
    const bool ok = strict_greater ? (km > t) : (km >= t);
    if (ok) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}
bool cmp_to_key_range(double thr, int cmp_op, int32_t* klo, int32_t* khi, int32_t* negate) {
  const int64_t t = h_f64_key(thr);
  const int64_t ge = first_key_not_less(t, false);  // first key with widen >= thr
  const int64_t gt = first_key_not_less(t, true);   // first key with widen >  thr
  int64_t lo, hi;
  *negate = 0;
  switch (cmp_op) {
    case 0: lo = gt; hi = INT32_MAX; break;                  // >
    case 1: lo = ge; hi = INT32_MAX; break;                  // >=
    case 2: lo = INT32_MIN; hi = ge - 1; break;              // <
    case 3: lo = INT32_MIN; hi = gt - 1; break;              // <=
    case 4: lo = ge; hi = gt - 1; break;                     // =
    case 5: lo = ge; hi = gt - 1; *negate = 1; break;        // !=
    default: return false;
  }
  if (lo > hi) {  // empty range: encode as an impossible interval
    *klo = INT32_MAX;
    *khi = INT32_MIN;
  } else {
    *klo = (int32_t)lo;
    *khi = (int32_t)hi;
  }
  return true;
}

hipError_t launch_cmp_avg_by_group(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const float* x,
                                   const uint8_t* x_valid, const float* y, const uint8_t* y_valid, const int32_t* gid,
                                   int64_t n, double thr, int cmp_op, int n_groups, int64_t* d_counts,
                                   double* d_sums) {
  if (n <= 0) return hipSuccess;
  if (n_groups < 1 || n_groups > 8) return hipErrorInvalidValue;
  int32_t klo, khi, negate;
  if (!cmp_to_key_range(thr, cmp_op, &klo, &khi, &negate)) return hipErrorInvalidValue;
  int grid = 1;
  hipError_t e;
  int G;
  switch (n_groups) {
#define EXON_K4_CASE(GG)                                                                       \
  case GG:                                                                                     \
    G = GG;                                                                                    \
    e = k4_launch_g<GG>(s, cfg, &grid, ws, x, x_valid, y, y_valid, gid, n, klo, khi, negate);        \
    break;
    EXON_K4_CASE(1)
    EXON_K4_CASE(2)
    EXON_K4_CASE(3)
    EXON_K4_CASE(4)
    EXON_K4_CASE(5)
    EXON_K4_CASE(6)
    EXON_K4_CASE(7)
    EXON_K4_CASE(8)
#undef EXON_K4_CASE
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  const int words = 3 * n_groups;
  hipLaunchKernelGGL(k4_finalize, dim3((words + 3) / 4), dim3(256), 0, s, ws.partials, grid, G, n_groups, d_counts,
                     d_sums);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K5 qual_pos_hist
//   quality_scores_to_list (char - 33; exon-core/src/udfs/sequence/quality_score_string_to_list.rs:83-86)
//   + unnest with ordinality + GROUP BY (position, score) COUNT(*), reported on the raw byte
//   (Phred = bin - 33).  104 B/read at L = 100: i32 offset + L quality bytes.
//   One 1024-thread workgroup per CU owns a u32[LT][257] LDS histogram (row padded by one bank so that
//   consecutive positions with similar bytes fall on different banks); positions >= LT fall back to
//   global atomics.  v1: one wave per read, lanes stride over positions.
// ------------------------------------------------------------------------------------------------
constexpr int K5_THREADS = 1024;
constexpr int K5_LT_MAX = 144;  // 144 * 257 * 4 = 148,032 B of the 160 KiB LDS
constexpr int K5_STRIDE = 257;

__global__ __launch_bounds__(K5_THREADS) void k5_qual_pos_hist_main(const int32_t* __restrict__ off,
                                                                    const uint8_t* __restrict__ bytes,
                                                                    int64_t n_reads, int lmax, int lt,
                                                                    unsigned long long* __restrict__ partials,
                                                                    unsigned long long* __restrict__ d_hist,
                                                                    int* __restrict__ status) {
  extern __shared__ unsigned k5_h[];  // [lt][257]
  for (int i = threadIdx.x; i < lt * K5_STRIDE; i += K5_THREADS) k5_h[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t gwave = (int64_t)blockIdx.x * (K5_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (K5_THREADS / 64);
  unsigned bad = 0;
  for (int64_t r = gwave; r < n_reads; r += nwaves) {
    const int32_t o0 = off[r], o1 = off[r + 1];
    int len = o1 - o0;
    if (len > lmax) {
      bad = 1;
      len = lmax;
    }
    for (int p = lane; p < len; p += 64) {
      const unsigned b = bytes[(int64_t)o0 + p];
      if (p < lt) atomicAdd(&k5_h[p * K5_STRIDE + b], 1u);
      else atomicAdd(&d_hist[(size_t)p * 256 + b], 1ull);
    }
  }
  if (bad) atomicOr(status, 8);
  __syncthreads();
  const int W = lt * 256;
  for (int i = threadIdx.x; i < W; i += K5_THREADS)
    partials[(size_t)blockIdx.x * W + i] = k5_h[(i >> 8) * K5_STRIDE + (i & 255)];
}

static int k5_lt(int lmax) { return lmax < K5_LT_MAX ? lmax : K5_LT_MAX; }
size_t k5_partial_words(const LaunchCfg& cfg, int lmax) { return (size_t)cfg.compute_units * k5_lt(lmax) * 256; }

hipError_t launch_qual_pos_hist(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* offsets,
                                const uint8_t* bytes, int64_t n_reads, int lmax, int64_t* d_hist) {
  if (n_reads <= 0) return hipSuccess;
  if (lmax < 1) return hipErrorInvalidValue;
  const int lt = k5_lt(lmax);
  int64_t g = cfg.compute_units;
  const int64_t need = (n_reads + (K5_THREADS / 64) - 1) / (K5_THREADS / 64);
  if (g > need) g = need;
  const int grid = (int)g;
  const size_t lds = (size_t)lt * K5_STRIDE * sizeof(unsigned);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k5_qual_pos_hist_main),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k5_qual_pos_hist_main, dim3(grid), dim3(K5_THREADS), lds, s, offsets, bytes, n_reads, lmax, lt,
                     ws.partials, reinterpret_cast<unsigned long long*>(d_hist), ws.status);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int W = lt * 256;
  return run_finalize(s, ws, grid, W, W, d_hist, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Synthetic inputs (DESIGN.md "Synthetic inputs"): counter-based, bit-identical to oracle/exon_oracle.c
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t rnd(uint64_t seed, uint64_t col, uint64_t i) {
  return mix64(seed + col * 0xD1B54A32D192ED03ULL + (i + 1) * 0x9E3779B97F4A7C15ULL);
}

static const int64_t GRCH37_LEN[25] = {249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663,
                                       146364022, 141213431, 135534747, 135006516, 133851895, 115169878, 107349540,
                                       102531392, 90354753,  81195210,  78077248,  59128983,  63025520,  48129895,
                                       51304566,  155270560, 59373566,  16569};

struct C2Table {
  int64_t starts[25];
  int64_t len[24];
};
struct C3Table {
  uint32_t fthr[11];
  int32_t flags[12];
  uint32_t rthr[24];
  uint32_t m8, m20, m40, m98;
};
static inline uint32_t pct_thr(int cum_pct) { return (uint32_t)((((uint64_t)cum_pct) << 32) / 100); }

// writes the validity byte(s) of the wave's 64 rows from a ballot; rows beyond n carry 0 bits
__device__ __forceinline__ void store_valid64(uint8_t* bm, int64_t wave_row0, int64_t n, bool v, int lane) {
  const unsigned long long m = __ballot(v);
  if (lane < 8) {
    const int64_t r = wave_row0 + lane * 8;
    if (r < n) bm[r >> 3] = (uint8_t)(m >> (lane * 8));
  }
}

__global__ __launch_bounds__(256) void gen_c2_kernel(uint64_t seed, int64_t lo, int64_t hi, C2Table t,
                                                     int32_t* __restrict__ chrom, int64_t* __restrict__ pos) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = lo + k;
  if (i >= hi) return;
  int c = 0;
  while (c < 23 && i >= t.starts[c + 1]) ++c;
  const int64_t j = i - t.starts[c], nc = t.starts[c + 1] - t.starts[c];
  int64_t w = t.len[c] / nc;
  if (w < 1) w = 1;
  int64_t p = 1 + j * w + (int64_t)(rnd(seed, 0, (uint64_t)i) % (uint64_t)w);
  if (p > t.len[c]) p = t.len[c];
  chrom[k] = c;
  pos[k] = p;
}

__global__ __launch_bounds__(256) void gen_c3_kernel(uint64_t seed, int64_t lo, int64_t hi, C3Table t,
                                                     int32_t* __restrict__ flag, uint8_t* __restrict__ mapq,
                                                     uint8_t* __restrict__ mvalid, int32_t* __restrict__ ref,
                                                     uint8_t* __restrict__ rvalid) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = hi - lo;
  const int lane = threadIdx.x & 63;
  bool qv = false, rv = false;
  if (k < n) {
    const uint64_t i = (uint64_t)(lo + k);
    const uint64_t r0 = rnd(seed, 0, i), r1 = rnd(seed, 1, i), r2 = rnd(seed, 2, i);
    const uint32_t u0 = (uint32_t)(r0 >> 32), u1 = (uint32_t)(r1 >> 32), v1 = (uint32_t)r1, u2 = (uint32_t)(r2 >> 32);
    int fi = 0;
#pragma unroll
    for (int q = 0; q < 11; ++q) fi += (u0 >= t.fthr[q]);
    const int32_t f = t.flags[fi];
    flag[k] = f;
    uint8_t q;
    qv = true;
    if (u1 < t.m8) q = 0;
    else if (u1 < t.m20) q = (uint8_t)(1 + v1 % 29);
    else if (u1 < t.m40) q = (uint8_t)(30 + v1 % 30);
    else if (u1 < t.m98) q = 60;
    else {
      q = 255;
      qv = false;
    }
    mapq[k] = q;
    int rc = 0;
#pragma unroll
    for (int c = 0; c < 24; ++c) rc += (u2 >= t.rthr[c]);
    rv = !(f & 4);
    ref[k] = rv ? rc : -1;
  }
  const int64_t wave_row0 = k - lane;
  store_valid64(mvalid, wave_row0, n, qv, lane);
  store_valid64(rvalid, wave_row0, n, rv, lane);
}

__global__ __launch_bounds__(256) void gen_c4_kernel(uint64_t seed, int64_t lo, int64_t hi, uint32_t t0, uint32_t t1,
                                                     uint32_t t2, uint32_t t3, float* __restrict__ af,
                                                     uint8_t* __restrict__ avalid, float* __restrict__ qual,
                                                     uint8_t* __restrict__ qvalid, int32_t* __restrict__ fid) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = hi - lo;
  const int lane = threadIdx.x & 63;
  bool av = false, qv = false;
  if (k < n) {
    const uint64_t i = (uint64_t)(lo + k);
    const uint64_t r0 = rnd(seed, 0, i), r1 = rnd(seed, 1, i), r2 = rnd(seed, 2, i);
    const uint32_t e = (uint32_t)((((r0 >> 23) & 0xFF) * 14) >> 8);
    const uint32_t bits = ((126u - e) << 23) | (uint32_t)(r0 & 0x7FFFFF);
    float a = __uint_as_float(bits);
    if (((r0 >> 31) & 0x3FF) == 0) a = 0.01f;
    af[k] = a;
    av = (r0 >> 44) >= 10486;
    const uint32_t kq = (uint32_t)(r1 & 0xFFFFFFFFu) % 10000u;
    qual[k] = (float)((double)kq / 10.0);
    qv = (r1 >> 44) >= 31457;
    const uint32_t u = (uint32_t)(r2 >> 32);
    fid[k] = (int32_t)((u >= t0) + (u >= t1) + (u >= t2) + (u >= t3));
  }
  const int64_t wave_row0 = k - lane;
  store_valid64(avalid, wave_row0, n, av, lane);
  store_valid64(qvalid, wave_row0, n, qv, lane);
}

__global__ __launch_bounds__(256) void gen_c5_kernel(uint64_t seed, int64_t lo, int64_t hi, int32_t L,
                                                     int32_t* __restrict__ off, uint8_t* __restrict__ bytes) {
  const int64_t total = (hi - lo) * (int64_t)L;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / L;
    const int p = (int)(e - r * L);
    const uint64_t h = rnd(seed, 0, (uint64_t)((lo + r) * L + p));
    const int s = (int)(h & 0xFF) + (int)((h >> 8) & 0xFF) + (int)((h >> 16) & 0xFF) + (int)((h >> 24) & 0xFF);
    const int d = ((s * 3 + 4096 - 1530) >> 7) - 32;
    int q = 38 - (10 * p) / L + d;
    q = q < 0 ? 0 : (q > 41 ? 41 : q);
    bytes[e] = (uint8_t)(33 + q);
  }
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= hi - lo; r += stride) off[r] = (int32_t)(r * L);
}

static unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_gen_c2(hipStream_t s, uint64_t seed, int64_t n_total, int64_t lo, int64_t hi, int32_t* chrom,
                         int64_t* pos) {
  if (hi <= lo) return hipSuccess;
  C2Table t;
  unsigned __int128 total = 0, cum = 0;
  for (int c = 0; c < 24; ++c) total += (unsigned __int128)GRCH37_LEN[c];
  t.starts[0] = 0;
  for (int c = 0; c < 24; ++c) {
    cum += (unsigned __int128)GRCH37_LEN[c];
    t.starts[c + 1] = (int64_t)(((unsigned __int128)n_total * cum) / total);
    t.len[c] = GRCH37_LEN[c];
  }
  t.starts[24] = n_total;
  hipLaunchKernelGGL(gen_c2_kernel, dim3(blocks_for(hi - lo)), dim3(256), 0, s, seed, lo, hi, t, chrom, pos);
  return hipGetLastError();
}

hipError_t launch_gen_c3(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, int32_t* flag, uint8_t* mapq,
                         uint8_t* mapq_valid, int32_t* ref_id, uint8_t* ref_valid) {
  if (hi <= lo) return hipSuccess;
  static const int32_t FLAGS[12] = {99, 147, 83, 163, 1123, 1171, 1187, 1107, 77, 141, 355, 65};
  static const int PCT[12] = {21, 21, 21, 21, 2, 2, 2, 2, 1, 1, 1, 5};
  C3Table t;
  int cum = 0;
  for (int k = 0; k < 11; ++k) {
    cum += PCT[k];
    t.fthr[k] = pct_thr(cum);
  }
  for (int k = 0; k < 12; ++k) t.flags[k] = FLAGS[k];
  unsigned __int128 total = 0, c128 = 0;
  for (int c = 0; c < 25; ++c) total += (unsigned __int128)GRCH37_LEN[c];
  for (int c = 0; c < 24; ++c) {
    c128 += (unsigned __int128)GRCH37_LEN[c];
    t.rthr[c] = (uint32_t)((c128 << 32) / total);
  }
  t.m8 = pct_thr(8);
  t.m20 = pct_thr(20);
  t.m40 = pct_thr(40);
  t.m98 = pct_thr(98);
  hipLaunchKernelGGL(gen_c3_kernel, dim3(blocks_for(hi - lo)), dim3(256), 0, s, seed, lo, hi, t, flag, mapq,
                     mapq_valid, ref_id, ref_valid);
  return hipGetLastError();
}

hipError_t launch_gen_c4(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, float* af, uint8_t* af_valid,
                         float* qual, uint8_t* qual_valid, int32_t* filter_id) {
  if (hi <= lo) return hipSuccess;
  hipLaunchKernelGGL(gen_c4_kernel, dim3(blocks_for(hi - lo)), dim3(256), 0, s, seed, lo, hi, pct_thr(85),
                     pct_thr(90), pct_thr(96), pct_thr(99), af, af_valid, qual, qual_valid, filter_id);
  return hipGetLastError();
}

hipError_t launch_gen_c5(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, int32_t read_len, int32_t* offsets,
                         uint8_t* bytes) {
  if (hi < lo) return hipErrorInvalidValue;
  int64_t total = (hi - lo) * (int64_t)read_len + 1;
  int64_t g = (total + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(gen_c5_kernel, dim3((unsigned)g), dim3(256), 0, s, seed, lo, hi, read_len, offsets, bytes);
  return hipGetLastError();
}

}  // namespace exon
