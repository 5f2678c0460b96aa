// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the scan -> filter -> aggregate hot path.
//
// All four operators are HBM-read bound (arithmetic intensity < 0.5 op/B), so there is no MFMA here.
// Shape shared by the streaming kernels (K2/K3/K4), settled by A/B runs on MI355X (tools/tune_k4.hip,
// profiles/r1_tuning.md):
//   * persistent grid, grid-stride over tiles.  Large inputs use ONE 1024-thread workgroup per CU
//     (16 waves, each owning 1024 rows of a 16384-row tile: J = 4 sub-tiles of 256 rows, 4 rows per
//     lane per sub-tile); small inputs use 256-thread workgroups with J = 2 so that every CU still
//     gets several tiles.  Big workgroups + non-temporal loads are what lift a read-only sweep of the
//     same columns from 5.7 to 6.7 TB/s on this chip;
//   * every column load is a fully coalesced, non-temporal 16 B/lane access (1 KiB per wave
//     instruction; the data is streamed once, so it should not displace L2/MALL lines);
//   * Arrow validity bitmaps are consumed as one byte per lane-pair (a nibble per lane per 4 rows);
//   * per-lane register accumulators (counters packed 8 x 8 bit into one u64 and spilled to u32
//     registers every <= 255 rows) -> wave shuffle reduction -> LDS across the waves -> one partial
//     record per workgroup in the workspace: no global atomics anywhere;
//   * `finalize_partials` folds the per-workgroup records into the caller's running state in a fixed
//     order, so f64 sums are bit-reproducible for a given launch shape.
// Reference semantics restated by each kernel are cited at the kernel.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace exon {

// launch shapes
struct ShapeBig {
  static constexpr int THREADS = 1024, J = 4;
};
struct ShapeSmall {
  static constexpr int THREADS = 256, J = 2;
};
struct ShapeBigJ2 {  // register-hungry variants (K4 with an LDS overflow table) that would spill at J = 4
  static constexpr int THREADS = 1024, J = 2;
};
template <typename S>
struct ShapeOf {
  static constexpr int WAVES = S::THREADS / 64;
  static constexpr int WAVE_TILE = 256 * S::J;        // rows per wave per iteration
  static constexpr int TILE = WAVES * WAVE_TILE;      // rows per workgroup per iteration
};

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
typedef int v4i_t __attribute__((ext_vector_type(4)));

// streaming (non-temporal) 16-byte load
template <typename T>
__device__ __forceinline__ T ld16(const void* p) {
  static_assert(sizeof(T) == 16, "16-byte vector expected");
#ifdef EXON_LD16_PLAIN  // A/B builds only (tools/build_variant.sh): every column load without the streaming hint
  const v4i_t v = *reinterpret_cast<const v4i_t*>(p);
#else
  const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p));
#endif
  T r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}
// streaming 8-byte load (tier-3 records: written once, read once)
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 ldnt8(const uint2* p) {
  const v2u_t v = __builtin_nontemporal_load(reinterpret_cast<const v2u_t*>(p));
  return uint2{v.x, v.y};
}
// The same without the streaming hint.  An i64 column gives every lane 32 consecutive bytes = two 16-byte loads that touch
// the SAME cache lines (lane stride 32 B): when both carry the streaming hint, some lines are fetched from HBM twice (PMC:
// 1.047 x the algorithmic bytes in K2, 1.049 x in K6); with a plain second load the line is still there: K2 1.877 -> 1.827 ms.
template <class T>
__device__ __forceinline__ T ld16_plain(const void* p) {
  static_assert(sizeof(T) == 16, "16-byte vector expected");
  const v4i_t v = *reinterpret_cast<const v4i_t*>(p);
  T r;
  __builtin_memcpy(&r, &v, 16);
  return r;
}

// validity nibble of this lane's 4 rows of sub-tile j: rows [wbase + 256 j + 4 lane, +4).
// wbase is a multiple of 256, so the sub-tile's 256 bits are 32 consecutive bytes.
// Branch-free: a NULL bitmap is replaced by a 64-byte all-ones buffer, so the load is unconditional.
// (With three optional bitmaps the compiler no longer unswitches the tile loop on their nullness and leaves
// `if (bm) { load; s_waitcnt vmcnt(0) }` inside it, which drains every load in flight -- measured on K3.)
__device__ __forceinline__ unsigned valid4_ones(const uint8_t* __restrict__ bm, const uint8_t* __restrict__ ones,
                                                int64_t wbase, int j, int lane) {
  const uint8_t* p = bm ? bm + (wbase >> 3) + j * 32 + (lane >> 1) : ones + (lane >> 1);
  return (unsigned(*p) >> ((lane & 1) * 4)) & 0xFu;
}
__device__ __forceinline__ bool valid1(const uint8_t* __restrict__ bm, int64_t r) {
  return bm == nullptr ? true : ((bm[r >> 3] >> (r & 7)) & 1);
}

// Which workgroup takes the rows behind the last whole tile.  Tiles are dealt b, b + grid, ...: the LAST workgroups are the
// ones with a tile less when the tiles do not divide evenly, so the remainder -- a dependent load behind the tile loop, one
// HBM latency -- goes to them and stays off the critical path (the first workgroups carried it until round 4; A/B build:
// -DEXON_REM_FRONT, profiles/r4_small_configs.log).
__device__ __forceinline__ unsigned rem_block() {
#ifdef EXON_REM_FRONT
  return blockIdx.x;
#else
  return gridDim.x - 1u - blockIdx.x;
#endif
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
// grid = ceil(V / 32), block = 256 = 32 values x 8 segments of the workgroup range; every thread keeps
// 4 independent loads in flight, the 8 segment sums are combined in a fixed order.
// ------------------------------------------------------------------------------------------------
// overwrite != 0: state[v] = sum (the state needs no zeroing pass before the launch that produces it).
// (Folding inside the main kernel by the last workgroup to finish -- the ticket pattern -- was tried and dropped: the
// agent-scope release/acquire fences it needs are L2 writeback / invalidate scans on this chip and cost more than this
// second launch, 37 vs 28 us on config 2's 10 M rows; profiles/r2_tuning.md.)
__global__ __launch_bounds__(256) void finalize_partials(const unsigned long long* __restrict__ partials, int nblocks,
                                                         int V, int n_i64, int64_t* __restrict__ st_i64,
                                                         double* __restrict__ st_f64, int overwrite) {
  __shared__ unsigned long long red[8][32];
  const int vi = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const int v = blockIdx.x * 32 + vi;
  const int per = (nblocks + 7) / 8;
  const int b0 = seg * per, b1 = min(nblocks, b0 + per);
  unsigned long long acc_i = 0;
  double acc_f = 0.0;
  if (v < V) {
    const unsigned long long* p = partials + v;
    if (v < n_i64) {
      int b = b0;
      for (; b + 4 <= b1; b += 4) {
        const unsigned long long a0 = p[(size_t)b * V], a1 = p[(size_t)(b + 1) * V], a2 = p[(size_t)(b + 2) * V],
                                 a3 = p[(size_t)(b + 3) * V];
        acc_i += a0 + a1 + a2 + a3;
      }
      for (; b < b1; ++b) acc_i += p[(size_t)b * V];
    } else {
      int b = b0;
      for (; b + 4 <= b1; b += 4) {
        const double a0 = __longlong_as_double((long long)p[(size_t)b * V]),
                     a1 = __longlong_as_double((long long)p[(size_t)(b + 1) * V]),
                     a2 = __longlong_as_double((long long)p[(size_t)(b + 2) * V]),
                     a3 = __longlong_as_double((long long)p[(size_t)(b + 3) * V]);
        acc_f += ((a0 + a1) + (a2 + a3));
      }
      for (; b < b1; ++b) acc_f += __longlong_as_double((long long)p[(size_t)b * V]);
    }
  }
  red[seg][vi] = (v < n_i64) ? acc_i : (unsigned long long)__double_as_longlong(acc_f);
  __syncthreads();
  if (seg == 0 && v < V) {
    if (v < n_i64) {
      unsigned long long t = 0;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += red[s][vi];
      st_i64[v] = (overwrite ? 0 : st_i64[v]) + (int64_t)t;
    } else {
      double t = 0.0;
#pragma unroll
      for (int s = 0; s < 8; ++s) t += __longlong_as_double((long long)red[s][vi]);
      st_f64[v - n_i64] = (overwrite ? 0.0 : st_f64[v - n_i64]) + t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Single-launch step (round 3): the LAST workgroup to finish folds the per-workgroup records, in the same fixed
// order as finalize_partials, so a step is ONE kernel -- no dependent launch (~3 us on this chip) and half the host
// launch cost, which is what configs 2 / 3 and the 125e6-row shard of an 8-GPU run are made of.
// Round 2's ticket pattern used agent-scope release / acquire FENCES, which on gfx950 are L2 writeback / invalidate
// scans (the 8 XCDs do not share an L2) and cost more than the second launch.  This one needs no fence:
//   * a workgroup's record words are written with agent-scope relaxed ATOMIC stores (global_store ... sc1: written
//     through to the memory side, never left dirty in this XCD's L2);
//   * s_waitcnt vmcnt(0) in the writing waves + the workgroup barrier: those stores are acknowledged before
//   * thread 0 takes a ticket (agent-scope relaxed fetch-add on ws.status[2]);
//   * the workgroup that draws the last ticket reads every record with agent-scope relaxed ATOMIC loads (sc1: served
//     from the memory side, not from a stale L2 line), folds them block 0, 1, ... (bit-reproducible f64 sums for a
//     given launch shape), writes the state and resets the ticket for the next launch on this stream.
// Only records of <= FUSE_MAX_V words take this route: bigger ones (LDS group tables) are a multi-workgroup job and
// keep the finalize_partials launch.  EXON_HIP_FUSE_FOLD=0 turns it off (A/B runs).
// ------------------------------------------------------------------------------------------------
constexpr int FUSE_MAX_V = 256;
struct FoldArgs {
  unsigned* ticket;  // nullptr: no fused fold (finalize_partials follows)
  int64_t* st_i64;
  double* st_f64;
  int V, n_i64, overwrite;
};
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Call with ALL threads of the workgroup after the record `partials[blockIdx.x * V ...]` was written with st_agent().
template <int THREADS>
__device__ __forceinline__ void fold_if_last(const FoldArgs& fa, const unsigned long long* __restrict__ partials) {
  __shared__ unsigned long long fold_red[THREADS];
  __shared__ int fold_last;
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): this wave's record stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(fa.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fold_last = (prev == gridDim.x - 1);
  }
  __syncthreads();
  if (!fold_last) return;
  // thread = (segment of the workgroup range) x (value): VP values side by side, THREADS / VP segments; a segment adds
  // its blocks in order (8 independent loads in flight), the segments are combined by a fixed binary tree.
  const int nblocks = (int)gridDim.x, V = fa.V, n_i64 = fa.n_i64;
  const int lg = V > 16 ? 5 : V > 8 ? 4 : V > 4 ? 3 : V > 2 ? 2 : V > 1 ? 1 : 0;
  const int VP = 1 << lg, SEGS = THREADS >> lg;
  const int vi = threadIdx.x & (VP - 1), seg = threadIdx.x >> lg;
  const int per = (nblocks + SEGS - 1) / SEGS;
  const int b0 = seg * per, b1 = min(nblocks, b0 + per);
  for (int v0 = 0; v0 < V; v0 += VP) {
    const int v = v0 + vi;
    const bool is_int = v < n_i64;
    unsigned long long acc_i = 0;
    double acc_f = 0.0;
    if (v < V) {
      const unsigned long long* p = partials + v;
      for (int b = b0; b < b1; b += 8) {
        unsigned long long a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (b + i < b1) ? ld_agent(p + (size_t)(b + i) * V) : 0ull;  // 0 bits = +0.0
        if (is_int) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc_i += a[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc_f += __longlong_as_double((long long)a[i]);
        }
      }
    }
    fold_red[threadIdx.x] = is_int ? acc_i : (unsigned long long)__double_as_longlong(acc_f);
    __syncthreads();
    for (int st = SEGS >> 1; st > 0; st >>= 1) {
      if (seg < st) {
        const unsigned long long x = fold_red[threadIdx.x], y = fold_red[threadIdx.x + (st << lg)];
        fold_red[threadIdx.x] =
            is_int ? x + y
                   : (unsigned long long)__double_as_longlong(__longlong_as_double((long long)x) + __longlong_as_double((long long)y));
      }
      __syncthreads();
    }
    if (seg == 0 && v < V) {
      if (is_int) {
        fa.st_i64[v] = (fa.overwrite ? 0 : fa.st_i64[v]) + (int64_t)fold_red[threadIdx.x];
      } else {
        fa.st_f64[v - n_i64] = (fa.overwrite ? 0.0 : fa.st_f64[v - n_i64]) + __longlong_as_double((long long)fold_red[threadIdx.x]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) __hip_atomic_store(fa.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static bool fuse_fold_enabled() {
  static const bool on = [] {
    const char* v = getenv("EXON_HIP_FUSE_FOLD");
    return !(v && v[0] == '0');
  }();
  return on;
}
static FoldArgs fold_args(const LaunchCfg& cfg, const Workspace& ws, int V, int n_i64, int64_t* st_i64, double* st_f64) {
  FoldArgs fa;
  fa.ticket = (fuse_fold_enabled() && V <= FUSE_MAX_V) ? reinterpret_cast<unsigned*>(ws.status + 2) : nullptr;
  fa.st_i64 = st_i64;
  fa.st_f64 = st_f64;
  fa.V = V;
  fa.n_i64 = n_i64;
  fa.overwrite = cfg.overwrite ? 1 : 0;
  return fa;
}

static hipError_t run_finalize(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, int nblocks, int V, int n_i64,
                               int64_t* st_i64, double* st_f64) {
  if (fuse_fold_enabled() && V <= FUSE_MAX_V) return hipSuccess;  // folded by the main kernel's last workgroup
  const int grid = (V + 31) / 32;
  hipLaunchKernelGGL(finalize_partials, dim3(grid), dim3(256), 0, s, ws.partials, nblocks, V, n_i64, st_i64, st_f64,
                     cfg.overwrite ? 1 : 0);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fold_states: AggregateExec(Final) across GPUs after ONE all-gather of the packed partial states.
// gathered = [world][V] words (rank-major; words [0, n_i64) int64, the rest float64); out[v] = sum over ranks in
// rank order 0, 1, ..., world-1: every rank computes the same bits whatever algorithm the collective used.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_states(const unsigned long long* __restrict__ gathered, int world, int V,
                                                   int n_i64, unsigned long long* __restrict__ out) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  if (v < n_i64) {
    unsigned long long t = 0;
    for (int r = 0; r < world; ++r) t += gathered[(size_t)r * V + v];
    out[v] = t;
  } else {
    double t = 0.0;
    for (int r = 0; r < world; ++r) t += __longlong_as_double((long long)gathered[(size_t)r * V + v]);
    out[v] = (unsigned long long)__double_as_longlong(t);
  }
}

hipError_t launch_fold_states(hipStream_t s, const void* gathered, int world, int64_t n_i64, int64_t n_f64, void* out) {
  const int64_t V = n_i64 + n_f64;
  if (V <= 0 || world < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fold_states, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, s,
                     static_cast<const unsigned long long*>(gathered), world, (int)V, (int)n_i64,
                     static_cast<unsigned long long*>(out));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// permute_add_state: bring a packed partial state keyed by one dictionary's ids into the id order of another.
// One thread per source key; targets are distinct unless a dictionary repeats a name (then the adds to the shared target
// are atomic: counts stay exact).  Layout: [planes_i64 x G int64][tail_i64 int64][planes_f64 x G float64].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void permute_add_state(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
                                                         const int32_t* __restrict__ map, int n_map, int G, int planes_i64, int tail_i64,
                                                         int planes_f64) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g < tail_i64) {
    const size_t w = (size_t)planes_i64 * G + g;
    if (src[w]) atomicAdd(&dst[w], src[w]);
  }
  if (g >= n_map || g >= G) return;
  const int t = map[g];
  if (t < 0 || t >= G) return;
  for (int p = 0; p < planes_i64; ++p) {
    const unsigned long long v = src[(size_t)p * G + g];
    if (v) atomicAdd(&dst[(size_t)p * G + t], v);
  }
  const size_t fbase = (size_t)planes_i64 * G + tail_i64;
  for (int p = 0; p < planes_f64; ++p) {
    const double v = __longlong_as_double((long long)src[fbase + (size_t)p * G + g]);
    if (v != 0.0) atomicAdd(reinterpret_cast<double*>(dst) + fbase + (size_t)p * G + t, v);
  }
}

hipError_t launch_permute_add_state(hipStream_t s, const void* src, void* dst, const int32_t* map, int n_map, int G, int planes_i64,
                                    int tail_i64, int planes_f64) {
  if (G < 0 || n_map < 0 || planes_i64 < 0 || planes_f64 < 0 || tail_i64 < 0) return hipErrorInvalidValue;
  const int threads = std::max(std::min(n_map, G), tail_i64);
  if (threads == 0) return hipSuccess;
  hipLaunchKernelGGL(permute_add_state, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, static_cast<const unsigned long long*>(src),
                     static_cast<unsigned long long*>(dst), map, n_map, G, planes_i64, tail_i64, planes_f64);
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

// The big shapes pay off as soon as every CU gets a tile (>= 2.1 M rows on 256 CUs: even ~2 tiles per CU beat the
// 256-thread shape, 10 M rows: 41 -> 31 us).  Between J = 4 (16384-row tiles) and J = 2 (8192-row tiles): round 3 took J = 2 only
// where the static tile -> workgroup deal made its critical path (ceil(tiles / CUs) tiles) more than 3 % shorter, ties going to
// the bigger tile.  Round 4 swept both shapes over 1e7 .. 1e9 rows on one box (profiles/r4_shape_sweep.log): for K2 and K3
// J = 2 is never slower and up to 9 % faster below 1e8 rows (K2 at 2e7 rows: 38.5 vs 42.0 us = 0.78 vs 0.72 of peak, at 5e7:
// 92.0 vs 96.8 us; K3 at 2e7: 31.7 vs 33.4 us), equal within the box-to-box noise from 2.5e8 rows up -- a tie on the critical
// path is not a tie: the smaller tile also ramps up and drains faster.  K2 / K3 / K6 take J = 2 whenever they take a big
// shape; K4 keeps J = 4 (its own rule below).
enum { SHAPE_SMALL = 0, SHAPE_BIG_J2 = 1, SHAPE_BIG_J4 = 2 };
static int pick_shape(const LaunchCfg& cfg, int64_t n) {
  static const int forced = [] {
    const char* v = getenv("EXON_HIP_SHAPE");  // A/B runs: 0 small, 1 big J=2, 2 big J=4
    return v && v[0] >= '0' && v[0] <= '2' ? v[0] - '0' : -1;
  }();
  if (forced >= 0) return forced;
  const int64_t cus = std::max(cfg.compute_units, 1);
  return n / ShapeOf<ShapeBigJ2>::TILE < cus ? SHAPE_SMALL : SHAPE_BIG_J2;
}
static bool use_big_shape(const LaunchCfg& cfg, int64_t n) { return pick_shape(cfg, n) != SHAPE_SMALL; }

template <typename S>
static int grid_for(const LaunchCfg& cfg, int64_t n, int resident) {
  const int64_t tiles = (n + ShapeOf<S>::TILE - 1) / ShapeOf<S>::TILE;
  const int per_cu = S::THREADS >= 1024 ? 1 : std::min(cfg.blocks_per_cu, resident);
  int64_t g = (int64_t)cfg.compute_units * per_cu;
  if (g > tiles) g = tiles;
  if (g < 1) g = 1;
  return (int)g;
}
static int max_grid(const LaunchCfg& cfg) { return cfg.compute_units * std::max(cfg.blocks_per_cu, 1); }

// ------------------------------------------------------------------------------------------------
// Read probe: what THIS box streams out of HBM through the access pattern of K2-K6 -- the same persistent grid (one
// 1024-thread workgroup per CU, J = 4 sub-tiles: ShapeBig), the same non-temporal 16 B/lane loads over `NB` buffers walked in
// lock-step like a plan's columns, and next to no arithmetic (four integer adds per load; one word per wave written at the
// end so that the loads are live).  bench.py runs it over the resident columns before and after the timed steps:
// roofline.box_read_ceiling_GBps.  A filter + aggregate kernel cannot beat it; how close it gets is what tuning can still move.
// ------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(ShapeBig::THREADS) void k_read_probe(const uint4* __restrict__ b0, const uint4* __restrict__ b1,
                                                                  const uint4* __restrict__ b2, const uint4* __restrict__ b3,
                                                                  int64_t nvec, unsigned* __restrict__ sink) {
  using S = ShapeBig;
  constexpr int J = S::J, WAVES = ShapeOf<S>::WAVES, WT = 64 * J, TILE = WAVES * WT;  // in 16-byte vectors
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4* const bufs[4] = {b0, b1, b2, b3};
  uint4 acc = uint4{0, 0, 0, 0};
  const int64_t ntiles = nvec / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WT;
    uint4 v[NB][J];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int b = 0; b < NB; ++b) v[b][j] = ld16<uint4>(bufs[b] + wbase + j * 64 + lane);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        acc.x += v[b][j].x;
        acc.y += v[b][j].y;
        acc.z += v[b][j].z;
        acc.w += v[b][j].w;
      }
  }
  unsigned long long t = (unsigned long long)(acc.x ^ acc.y) + (acc.z ^ acc.w);
  t = wave_sum(t);
  if (lane == 0) sink[blockIdx.x * WAVES + wave] = (unsigned)t;
}
// one pass over the first `bytes_each` bytes (rounded down to whole 64 KiB tiles) of each of n_buffers (1..4) buffers;
// sink: compute_units * 16 words
hipError_t launch_read_probe(hipStream_t s, const LaunchCfg& cfg, const void* const* buffers, int n_buffers, int64_t bytes_each,
                             unsigned* sink) {
  const int64_t nvec = bytes_each / 16;
  constexpr int64_t TILE = (int64_t)ShapeOf<ShapeBig>::WAVES * 64 * ShapeBig::J;
  const int64_t tiles = nvec / TILE;
  if (tiles < 1 || n_buffers < 1 || n_buffers > 4) return hipErrorInvalidValue;
  const int grid = (int)std::min<int64_t>(tiles, std::max(cfg.compute_units, 1));
  const uint4* b[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < n_buffers; ++i) b[i] = reinterpret_cast<const uint4*>(buffers[i]);
  switch (n_buffers) {
    case 1: hipLaunchKernelGGL(k_read_probe<1>, dim3(grid), dim3(ShapeBig::THREADS), 0, s, b[0], b[1], b[2], b[3], nvec, sink); break;
    case 2: hipLaunchKernelGGL(k_read_probe<2>, dim3(grid), dim3(ShapeBig::THREADS), 0, s, b[0], b[1], b[2], b[3], nvec, sink); break;
    case 3: hipLaunchKernelGGL(k_read_probe<3>, dim3(grid), dim3(ShapeBig::THREADS), 0, s, b[0], b[1], b[2], b[3], nvec, sink); break;
    default: hipLaunchKernelGGL(k_read_probe<4>, dim3(grid), dim3(ShapeBig::THREADS), 0, s, b[0], b[1], b[2], b[3], nvec, sink); break;
  }
  return hipGetLastError();
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

template <typename S>
__global__ __launch_bounds__(S::THREADS) void k2_region_count_main(const int32_t* __restrict__ chrom,
                                                                   const uint8_t* __restrict__ cvalid,
                                                                   const int64_t* __restrict__ pos,
                                                                   const uint8_t* __restrict__ pvalid, int64_t n,
                                                                   int32_t id, int64_t a, int64_t b,
                                                                   unsigned long long* __restrict__ partials,
                                                                   const uint8_t* __restrict__ ones, const FoldArgs fa) {
  constexpr int J = S::J, THREADS = S::THREADS, WAVES = ShapeOf<S>::WAVES, WT = ShapeOf<S>::WAVE_TILE,
                TILE = ShapeOf<S>::TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned cnt = 0;
  const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WT;
    int4 c[J];
    longlong2 p0[J], p1[J];
    unsigned cm[J], pm[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t r = wbase + j * 256 + lane * 4;
      c[j] = ld16<int4>(chrom + r);
      p0[j] = ld16<longlong2>(pos + r);
      p1[j] = ld16_plain<longlong2>(pos + r + 2);
      cm[j] = valid4_ones(cvalid, ones, wbase, j, lane);
      pm[j] = valid4_ones(pvalid, ones, wbase, j, lane);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      cnt += k2_row(c[j].x, p0[j].x, cm[j] >> 0 & 1, pm[j] >> 0 & 1, id, a, b);
      cnt += k2_row(c[j].y, p0[j].y, cm[j] >> 1 & 1, pm[j] >> 1 & 1, id, a, b);
      cnt += k2_row(c[j].z, p1[j].x, cm[j] >> 2 & 1, pm[j] >> 2 & 1, id, a, b);
      cnt += k2_row(c[j].w, p1[j].y, cm[j] >> 3 & 1, pm[j] >> 3 & 1, id, a, b);
    }
  }
  for (int64_t r = ntiles * TILE + (int64_t)rem_block() * THREADS + threadIdx.x; r < n;
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
    st_agent(&partials[blockIdx.x], t);
  }
  if (fa.ticket) fold_if_last<THREADS>(fa, partials);
}

size_t k2_partial_words(const LaunchCfg& cfg) { return (size_t)max_grid(cfg); }

template <typename S>
static hipError_t k2_launch(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* chrom,
                            const uint8_t* cv, const int64_t* pos, const uint8_t* pv, int64_t n, int32_t id,
                            int64_t a, int64_t b, int* grid_out, const FoldArgs& fa) {
  static const int resident = resident_blocks(k2_region_count_main<S>, S::THREADS, 0);
  const int grid = grid_for<S>(cfg, n, resident);
  *grid_out = grid;
  hipLaunchKernelGGL(k2_region_count_main<S>, dim3(grid), dim3(S::THREADS), 0, s, chrom, cv, pos, pv, n, id, a, b,
                     ws.partials, reinterpret_cast<const uint8_t*>(ws.status + 8), fa);
  return hipGetLastError();
}

hipError_t launch_region_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* chrom,
                               const uint8_t* chrom_valid, const int64_t* pos, const uint8_t* pos_valid, int64_t n,
                               int32_t region_chrom, int64_t start, int64_t end, int64_t* d_count) {
  if (n <= 0) return hipSuccess;
  int grid = 1;
  const FoldArgs fa = fold_args(cfg, ws, 1, 1, d_count, nullptr);
  const int shape = pick_shape(cfg, n);
  hipError_t e = shape == SHAPE_BIG_J4
                     ? k2_launch<ShapeBig>(s, cfg, ws, chrom, chrom_valid, pos, pos_valid, n, region_chrom, start, end, &grid, fa)
                 : shape == SHAPE_BIG_J2
                     ? k2_launch<ShapeBigJ2>(s, cfg, ws, chrom, chrom_valid, pos, pos_valid, n, region_chrom, start, end, &grid, fa)
                     : k2_launch<ShapeSmall>(s, cfg, ws, chrom, chrom_valid, pos, pos_valid, n, region_chrom, start, end, &grid, fa);
  if (e != hipSuccess) return e;
  return run_finalize(s, cfg, ws, grid, 1, 1, d_count, nullptr);
}

// ------------------------------------------------------------------------------------------------
// K6 overlap_count  (interval hit, range form)
//   WHERE bam_region_filter('<ref>:<a>-<b>', reference, start, end)  COUNT(*)
//   = SemiLazyRecord::intersects (exon-bam/src/indexed_async_batch_stream.rs:66-87): same reference id AND
//   aln_start <= region_end AND region_start <= aln_end, false when reference / start / end is missing.
//   20.375 B/row: i32 reference id + i64 start + i64 end + 3 validity bits.  Shape of K2.
// ------------------------------------------------------------------------------------------------
//   STRICT: the BED / GFF form `reference = lit AND start > a AND "end" < b` (StartEndIntervalPhysicalExpr evaluates its
//   inner BinaryExprs: exon-core/src/physical_plan/start_end_interval_physical_expr.rs:93-139, 186-191): the row's interval
//   lies strictly inside (a, b).  Same columns, same traffic.
template <bool STRICT>
__device__ __forceinline__ unsigned k6_row(int32_t r, int64_t s, int64_t e, unsigned rv, unsigned sv, unsigned ev, int32_t id,
                                           int64_t a, int64_t b) {
  if (STRICT) return (rv & sv & ev) & unsigned(r == id) & unsigned(s > a) & unsigned(e < b);
  return (rv & sv & ev) & unsigned(r == id) & unsigned(s <= b) & unsigned(e >= a);
}

template <typename S, bool STRICT>
__global__ __launch_bounds__(S::THREADS) void k6_overlap_count_main(const int32_t* __restrict__ ref,
                                                                    const uint8_t* __restrict__ rvalid,
                                                                    const int64_t* __restrict__ start,
                                                                    const uint8_t* __restrict__ svalid,
                                                                    const int64_t* __restrict__ end,
                                                                    const uint8_t* __restrict__ evalid, int64_t n,
                                                                    int32_t id, int64_t a, int64_t b,
                                                                    unsigned long long* __restrict__ partials,
                                                                    const uint8_t* __restrict__ ones, const FoldArgs fa) {
  constexpr int J = S::J, THREADS = S::THREADS, WAVES = ShapeOf<S>::WAVES, WT = ShapeOf<S>::WAVE_TILE,
                TILE = ShapeOf<S>::TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned cnt = 0;
  const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WT;
    int4 c[J];
    longlong2 s0[J], s1[J], e0[J], e1[J];
    unsigned rm[J], sm[J], em[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t r = wbase + j * 256 + lane * 4;
      c[j] = ld16<int4>(ref + r);
      s0[j] = ld16<longlong2>(start + r);
      s1[j] = ld16_plain<longlong2>(start + r + 2);
      e0[j] = ld16<longlong2>(end + r);
      e1[j] = ld16_plain<longlong2>(end + r + 2);
      rm[j] = valid4_ones(rvalid, ones, wbase, j, lane);
      sm[j] = valid4_ones(svalid, ones, wbase, j, lane);
      em[j] = valid4_ones(evalid, ones, wbase, j, lane);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      cnt += k6_row<STRICT>(c[j].x, s0[j].x, e0[j].x, rm[j] >> 0 & 1, sm[j] >> 0 & 1, em[j] >> 0 & 1, id, a, b);
      cnt += k6_row<STRICT>(c[j].y, s0[j].y, e0[j].y, rm[j] >> 1 & 1, sm[j] >> 1 & 1, em[j] >> 1 & 1, id, a, b);
      cnt += k6_row<STRICT>(c[j].z, s1[j].x, e1[j].x, rm[j] >> 2 & 1, sm[j] >> 2 & 1, em[j] >> 2 & 1, id, a, b);
      cnt += k6_row<STRICT>(c[j].w, s1[j].y, e1[j].y, rm[j] >> 3 & 1, sm[j] >> 3 & 1, em[j] >> 3 & 1, id, a, b);
    }
  }
  for (int64_t r = ntiles * TILE + (int64_t)rem_block() * THREADS + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * THREADS)
    cnt += k6_row<STRICT>(ref[r], start[r], end[r], valid1(rvalid, r), valid1(svalid, r), valid1(evalid, r), id, a, b);

  __shared__ unsigned long long red[WAVES];
  const unsigned long long w = wave_sum((unsigned long long)cnt);
  if (lane == 0) red[wave] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) t += red[i];
    st_agent(&partials[blockIdx.x], t);
  }
  if (fa.ticket) fold_if_last<THREADS>(fa, partials);
}

template <typename S, bool STRICT>
static hipError_t k6_launch(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* ref, const uint8_t* rv,
                            const int64_t* start, const uint8_t* sv, const int64_t* end, const uint8_t* ev, int64_t n,
                            int32_t id, int64_t a, int64_t b, int* grid_out, const FoldArgs& fa) {
  static const int resident = resident_blocks(k6_overlap_count_main<S, STRICT>, S::THREADS, 0);
  const int grid = grid_for<S>(cfg, n, resident);
  *grid_out = grid;
  hipLaunchKernelGGL((k6_overlap_count_main<S, STRICT>), dim3(grid), dim3(S::THREADS), 0, s, ref, rv, start, sv, end, ev, n, id, a, b,
                     ws.partials, reinterpret_cast<const uint8_t*>(ws.status + 8), fa);
  return hipGetLastError();
}

hipError_t launch_overlap_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* ref,
                                const uint8_t* ref_valid, const int64_t* start, const uint8_t* start_valid,
                                const int64_t* end, const uint8_t* end_valid, int64_t n, int32_t region_ref,
                                int64_t region_start, int64_t region_end, int64_t* d_count, bool strict) {
  if (n <= 0) return hipSuccess;
  int grid = 1;
  const bool big = use_big_shape(cfg, n);
  hipError_t e;
#define EXON_K6(SHAPE, STRICT) \
  k6_launch<SHAPE, STRICT>(s, cfg, ws, ref, ref_valid, start, start_valid, end, end_valid, n, region_ref, region_start, region_end, &grid, fa)
  const FoldArgs fa = fold_args(cfg, ws, 1, 1, d_count, nullptr);
  if (strict) e = big ? EXON_K6(ShapeBigJ2, true) : EXON_K6(ShapeSmall, true);
  else e = big ? EXON_K6(ShapeBigJ2, false) : EXON_K6(ShapeSmall, false);
#undef EXON_K6
  if (e != hipSuccess) return e;
  return run_finalize(s, cfg, ws, grid, 1, 1, d_count, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Pushed-down region filter as a row mask (vcf_region_filter / bam_region_filter evaluated by the scan itself):
//   point form  IndexedAsyncBatchStream::filter (exon-vcf/src/indexed_async_batch_stream.rs:99-116):
//               chrom == region.name AND start <= pos <= end; a record without pos never matches
//   range form  SemiLazyRecord::intersects (exon-bam/src/indexed_async_batch_stream.rs:66-87):
//               same reference AND aln_start <= region_end AND region_start <= aln_end; any of them missing: no match
// The GPU decode path keeps every parsed row in place and hands the plan's kernel a validity bitmap for its FIRST
// operand = (that operand's own validity) AND (row passes): all fused kernels drop rows whose first operand is NULL, so
// the filter costs one pass over 12-20 B/row of freshly parsed columns and no compaction.  *n_pass counts the rows kept
// (the scan reports rows it emitted, like the reference's filtered stream).
// ------------------------------------------------------------------------------------------------
template <bool RANGE>
__global__ __launch_bounds__(256) void k_region_mask(const int32_t* __restrict__ id_col, const uint8_t* __restrict__ id_valid,
                                                     const int64_t* __restrict__ start, const int64_t* __restrict__ end,
                                                     const uint8_t* __restrict__ pos_valid, const uint8_t* __restrict__ in_valid,
                                                     int64_t n, int32_t id, int64_t a, int64_t b, uint8_t* __restrict__ out_valid,
                                                     unsigned long long* __restrict__ n_pass) {
  const int lane = threadIdx.x & 63;
  unsigned long long kept = 0;
  const int64_t n64 = (n + 63) & ~(int64_t)63;  // whole waves: the ballot needs every lane
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n64; r += (int64_t)gridDim.x * 256) {
    bool pass = false;
    if (r < n) {
      const bool ok = valid1(id_valid, r) && valid1(pos_valid, r) && id_col[r] == id;
      if (RANGE) pass = ok && start[r] <= b && end[r] >= a;
      else pass = ok && start[r] >= a && start[r] <= b;
    }
    const unsigned long long hits = __ballot(pass);                           // rows the scan emits
    const unsigned long long m = __ballot(pass && valid1(in_valid, r < n ? r : 0));  // ... of which the plan's operand is valid
    if (lane < 8) {
      const int64_t r0 = r - lane + lane * 8;
      if (r0 < n) out_valid[r0 >> 3] = (uint8_t)(m >> (lane * 8));
    }
    if (lane == 0) kept += (unsigned long long)__popcll(hits);
  }
  // one atomic per workgroup
  __shared__ unsigned long long red[4];
  if (lane == 0) red[threadIdx.x >> 6] = kept;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(n_pass, t);
  }
}

hipError_t launch_region_mask(hipStream_t s, bool range_form, const int32_t* id_col, const uint8_t* id_valid, const int64_t* start,
                              const int64_t* end, const uint8_t* pos_valid, const uint8_t* in_valid, int64_t n, int32_t id,
                              int64_t a, int64_t b, uint8_t* out_valid, unsigned long long* n_pass) {
  if (n <= 0) return hipSuccess;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
  if (range_form)
    hipLaunchKernelGGL(k_region_mask<true>, dim3(grid), dim3(256), 0, s, id_col, id_valid, start, end, pos_valid, in_valid, n, id, a, b,
                       out_valid, n_pass);
  else
    hipLaunchKernelGGL(k_region_mask<false>, dim3(grid), dim3(256), 0, s, id_col, id_valid, start, end, pos_valid, in_valid, n, id, a, b,
                       out_valid, n_pass);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K3 flag_mapq_group_count
//   WHERE (flag & M) = V AND CAST(mapping_quality AS INT) >= Q  GROUP BY reference  COUNT(*)
//   flag test = sam_flag_function (exon-core/src/udfs/sam/samflags.rs:26-47); mapq NULL when 255
//   (exon-bam/src/array_builder.rs:136-143) -> NULL predicate -> row dropped; NULL reference is its own
//   group (index n_refs).  9.25 B/row: i32 flag + u8 mapq + i32 ref id + 2 validity bits.
//   Group table: one u32[n_refs+1] table per wave in LDS, LDS atomics, folded per workgroup.
// ------------------------------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(S::THREADS) void k3_flag_mapq_group_count_main(
    const int32_t* __restrict__ flag, const uint8_t* __restrict__ fvalid, const uint8_t* __restrict__ mapq,
    const uint8_t* __restrict__ mvalid, const int32_t* __restrict__ ref, const uint8_t* __restrict__ rvalid,
    int64_t n, int32_t mask, int32_t value, int32_t qmin, int32_t R, unsigned long long* __restrict__ partials,
    int* __restrict__ status, const uint8_t* __restrict__ ones, const FoldArgs fa) {
  constexpr int J = S::J, THREADS = S::THREADS, WAVES = ShapeOf<S>::WAVES, WT = ShapeOf<S>::WAVE_TILE,
                TILE = ShapeOf<S>::TILE;
  // per-wave table: [R+1] group counters + 64 per-lane dummy slots.  Rows that fail the predicate add to their
  // lane's dummy slot, so the row loop is branch-free straight-line code (an exec-masked `if (pass)` per row costs
  // 16 tiny basic blocks per iteration and serialises the LDS adds behind the mask updates).
  extern __shared__ unsigned k3_tbl[];  // [WAVES][R+1+64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int V = R + 1, VS = V + 64;
  for (int i = threadIdx.x; i < WAVES * VS; i += THREADS) k3_tbl[i] = 0;
  __syncthreads();
  unsigned* mine = k3_tbl + wave * VS;
  const unsigned dummy = (unsigned)(V + lane);
  unsigned kmax = 0;  // largest valid-row key seen (range check once, not per row)

  auto row = [&](int32_t f, unsigned q, int32_t r, unsigned fv, unsigned mv, unsigned rv) {
    const bool pass = fv && ((f & mask) == value) && mv && ((int32_t)q >= qmin);
    const unsigned key = rv ? (unsigned)r : (unsigned)R;
    kmax = max(kmax, pass ? key : 0u);
    atomicAdd(&mine[(pass && key <= (unsigned)R) ? key : dummy], 1u);
  };

  // The 4 rows of a lane x 64 lanes = 256 consecutive rows.  In a coordinate-sorted BAM (the normal case) they all carry
  // ONE reference: 64 lanes adding to the same LDS counter serialise (measured: 1.6 x the step time of the synthetic mix,
  // tools/time_skew.py), so when every passing row of the group has the same key one lane adds the group's count --
  // three wave-uniform tests per 256 rows.  Mixed keys take the branch-free per-row adds.
  int uni_skip = 0;  // wave-uniform
  auto rows4 = [&](int4 f4, unsigned q4, int4 r4, unsigned fv, unsigned mv, unsigned rv) {
    const bool p0 = (fv >> 0 & 1) && ((f4.x & mask) == value) && (mv >> 0 & 1) && ((int32_t)(q4 & 0xFF) >= qmin);
    const bool p1 = (fv >> 1 & 1) && ((f4.y & mask) == value) && (mv >> 1 & 1) && ((int32_t)(q4 >> 8 & 0xFF) >= qmin);
    const bool p2 = (fv >> 2 & 1) && ((f4.z & mask) == value) && (mv >> 2 & 1) && ((int32_t)(q4 >> 16 & 0xFF) >= qmin);
    const bool p3 = (fv >> 3 & 1) && ((f4.w & mask) == value) && (mv >> 3 & 1) && ((int32_t)(q4 >> 24) >= qmin);
    const unsigned k0 = (rv >> 0 & 1) ? (unsigned)r4.x : (unsigned)R, k1 = (rv >> 1 & 1) ? (unsigned)r4.y : (unsigned)R;
    const unsigned k2 = (rv >> 2 & 1) ? (unsigned)r4.z : (unsigned)R, k3 = (rv >> 3 & 1) ? (unsigned)r4.w : (unsigned)R;
    kmax = max(max(kmax, p0 ? k0 : 0u), max(max(p1 ? k1 : 0u, p2 ? k2 : 0u), p3 ? k3 : 0u));
    // operands of the four adds; ONE code path issues them -- the uniform-key case only rewrites the operands
    unsigned s0 = (p0 && k0 <= (unsigned)R) ? k0 : dummy, s1 = (p1 && k1 <= (unsigned)R) ? k1 : dummy;
    unsigned s2 = (p2 && k2 <= (unsigned)R) ? k2 : dummy, s3 = (p3 && k3 <= (unsigned)R) ? k3 : dummy;
    unsigned a0 = 1u;
    if (uni_skip > 0) {  // the last test found mixed keys: no test for the next 15 groups
      --uni_skip;
    } else {
      const unsigned kw = (unsigned)__builtin_amdgcn_readfirstlane((int)k0);
      const bool same = (k0 == kw) & (k1 == kw) & (k2 == kw) & (k3 == kw);
      if (__all(same) && kw <= (unsigned)R) {
        // all 256 rows carry ONE reference: lane 0 adds the group's count once, every other add goes to a dummy slot
        const unsigned total = (unsigned)(__popcll(__ballot(p0)) + __popcll(__ballot(p1)) + __popcll(__ballot(p2)) + __popcll(__ballot(p3)));
        s0 = lane == 0 ? kw : dummy;
        a0 = lane == 0 ? total : 0u;
        s1 = s2 = s3 = dummy;
      } else {
        uni_skip = 15;
      }
    }
    atomicAdd(&mine[s0], a0);
    atomicAdd(&mine[s1], 1u);
    atomicAdd(&mine[s2], 1u);
    atomicAdd(&mine[s3], 1u);
  };

  const int64_t ntiles = n / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WT;
    int4 f[J], g[J];
    unsigned q[J], fm[J], mm[J], rm[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t r = wbase + j * 256 + lane * 4;
      f[j] = ld16<int4>(flag + r);
      g[j] = ld16<int4>(ref + r);
      q[j] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(mapq + r));
      fm[j] = valid4_ones(fvalid, ones, wbase, j, lane);
      mm[j] = valid4_ones(mvalid, ones, wbase, j, lane);
      rm[j] = valid4_ones(rvalid, ones, wbase, j, lane);
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
      rows4(f[j], q[j], g[j], fm[j], mm[j], rm[j]);
  }
  for (int64_t r = ntiles * TILE + (int64_t)rem_block() * THREADS + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * THREADS)
    row(flag[r], mapq[r], ref[r], valid1(fvalid, r), valid1(mvalid, r), valid1(rvalid, r));

  if (kmax > (unsigned)R) atomicOr(status, 2);
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += THREADS) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += k3_tbl[w * VS + v];
    st_agent(&partials[(size_t)blockIdx.x * V + v], t);
  }
  if (fa.ticket) fold_if_last<THREADS>(fa, partials);
}

size_t k3_partial_words(const LaunchCfg& cfg, int n_refs) {
  if (n_refs + 1 > 4096) return 16;  // the global-atomic path writes the caller's counters directly
  return (size_t)max_grid(cfg) * (size_t)(n_refs + 1);
}

template <typename S>
static hipError_t k3_launch(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* flag,
                            const uint8_t* fv, const uint8_t* mapq, const uint8_t* mv, const int32_t* ref,
                            const uint8_t* rv, int64_t n, int32_t mask, int32_t value, int32_t qmin, int32_t R,
                            int* grid_out, const FoldArgs& fa) {
  const size_t lds = (size_t)ShapeOf<S>::WAVES * (R + 1 + 64) * sizeof(unsigned);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k3_flag_mapq_group_count_main<S>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const int grid = grid_for<S>(cfg, n, resident_blocks(k3_flag_mapq_group_count_main<S>, S::THREADS, lds));
  *grid_out = grid;
  hipLaunchKernelGGL(k3_flag_mapq_group_count_main<S>, dim3(grid), dim3(S::THREADS), lds, s, flag, fv, mapq, mv, ref,
                     rv, n, mask, value, qmin, R, ws.partials, ws.status, reinterpret_cast<const uint8_t*>(ws.status + 8), fa);
  return hipGetLastError();
}

// More references than an LDS table holds (assemblies with tens of thousands of scaffolds): one global atomic per passing
// row straight into the caller's counters.  Keys spread over many addresses, so the atomics do not serialise the way a
// handful of hot groups would; still several times slower than the LDS path and used only beyond EXON_HIP_MAX_GROUPS.
__global__ __launch_bounds__(256) void k3_flag_mapq_group_count_global(
    const int32_t* __restrict__ flag, const uint8_t* __restrict__ fvalid, const uint8_t* __restrict__ mapq,
    const uint8_t* __restrict__ mvalid, const int32_t* __restrict__ ref, const uint8_t* __restrict__ rvalid, int64_t n,
    int32_t mask, int32_t value, int32_t qmin, int32_t R, unsigned long long* __restrict__ counts, int* __restrict__ status) {
  bool bad = false;
  const int lane = threadIdx.x & 63;
  const int64_t n64 = (n + 63) & ~(int64_t)63;  // whole waves stay in the loop together: the ballots need every lane
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n64; r += (int64_t)gridDim.x * 256) {
    bool pass = false;
    unsigned key = 0;
    if (r < n) {
      pass = valid1(fvalid, r) && ((flag[r] & mask) == value) && valid1(mvalid, r) && ((int32_t)mapq[r] >= qmin);
      key = valid1(rvalid, r) ? (unsigned)ref[r] : (unsigned)R;
      if (pass && key > (unsigned)R) {
        bad = true;
        pass = false;
      }
    }
    // a coordinate-sorted file gives a wave 64 reads of ONE reference: one atomic for the wave instead of 64 on one address
    const unsigned long long pm = __ballot(pass);
    if (pm == 0) continue;  // wave-uniform
    const int leader = (int)__ffsll((long long)pm) - 1;
    const unsigned kw = (unsigned)__builtin_amdgcn_readlane((int)key, leader);
    if (__ballot(pass && key != kw) == 0) {
      if (lane == leader) atomicAdd(&counts[kw], (unsigned long long)__popcll(pm));
    } else if (pass) {
      atomicAdd(&counts[key], 1ull);
    }
  }
  if (bad) atomicOr(&status[0], 2);  // same status bit as the LDS path: reference id out of range
}

hipError_t launch_flag_mapq_group_count(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* flag,
                                        const uint8_t* flag_valid, const uint8_t* mapq, const uint8_t* mapq_valid,
                                        const int32_t* ref_id, const uint8_t* ref_valid, int64_t n, int32_t flag_mask,
                                        int32_t flag_value, int32_t mapq_min, int32_t n_refs, int64_t* d_counts) {
  if (n <= 0) return hipSuccess;
  if (n_refs + 1 > 4096) {
    if (cfg.overwrite) {  // the global-atomic path adds straight into the caller's counters
      hipError_t e0 = hipMemsetAsync(d_counts, 0, (size_t)(n_refs + 1) * 8, s);
      if (e0 != hipSuccess) return e0;
    }
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)cfg.compute_units * 32);
    hipLaunchKernelGGL(k3_flag_mapq_group_count_global, dim3(grid), dim3(256), 0, s, flag, flag_valid, mapq, mapq_valid, ref_id,
                       ref_valid, n, flag_mask, flag_value, mapq_min, n_refs, reinterpret_cast<unsigned long long*>(d_counts),
                       ws.status);
    return hipGetLastError();
  }
  int grid = 1;
  // the 16-wave shape needs 16 x (n_refs+1) x 4 B of LDS: fine up to ~2.5k references
  const bool big = use_big_shape(cfg, n) && (size_t)16 * (n_refs + 1 + 64) * 4 <= 160 * 1024;
  const int V = n_refs + 1;
  const FoldArgs fa = fold_args(cfg, ws, V, V, d_counts, nullptr);
  hipError_t e = big ? (pick_shape(cfg, n) == SHAPE_BIG_J2
                            ? k3_launch<ShapeBigJ2>(s, cfg, ws, flag, flag_valid, mapq, mapq_valid, ref_id, ref_valid, n, flag_mask,
                                                    flag_value, mapq_min, n_refs, &grid, fa)
                            : k3_launch<ShapeBig>(s, cfg, ws, flag, flag_valid, mapq, mapq_valid, ref_id, ref_valid, n,
                                                  flag_mask, flag_value, mapq_min, n_refs, &grid, fa))
                     : k3_launch<ShapeSmall>(s, cfg, ws, flag, flag_valid, mapq, mapq_valid, ref_id, ref_valid, n,
                                             flag_mask, flag_value, mapq_min, n_refs, &grid, fa);
  if (e != hipSuccess) return e;
  return run_finalize(s, cfg, ws, grid, V, V, d_counts, nullptr);
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
//   12.25 B/row: f32 x + f32 y + i32 group id + 2 validity bits.  Group ids < G <= 8 live in registers:
//   COUNT(*) / COUNT(y) in two u64 registers of 8 x 8-bit fields (a row adds 1 << 8g), spilled to
//   per-group u32 registers before a field can overflow; sums in G f64 registers.
// ------------------------------------------------------------------------------------------------
//   More than 8 groups (OVF), three tiers by dictionary id -- ids are handed out in order of first appearance, so the
//   frequent keys of a skewed column are the early ones:
//     ids 0..G-1          registers, as above (G = 4 in this variant);
//     ids G..NL-1         a per-workgroup LDS table (NL <= G + 4096) of 16-byte entries {f64 sum, u32 count(y), u32 count(*)}:
//                         ds_add_f64 + 2 x ds_add_u32 on ONE address register.  The block is entered only when some lane
//                         of the wave needs it (one scalar branch per 4 rows per lane) and is branch-free inside: a row
//                         that does not belong there adds to its lane's dummy entry (K3's finding: an exec-masked `if`
//                         per row costs a basic block each and serialises the adds behind the masks), a passing row
//                         whose y is NULL adds 0 / +0.0 to its group's count(y) / sum;
//     ids NL..NG-1        global atomics straight into the caller's state (only when NG > NL: high-cardinality keys), again
//                         behind a wave-uniform test, so a column whose hot keys are early hardly ever gets there.
//   Counts stay exact; the f64 sums of tiers 2 / 3 depend on the atomic order (~1e-16 relative run to run).
//   The variant is bound by vector-instruction issue, not by LDS conflicts or HBM (profiles/r3_groupby.md: the same time
//   for 64 and 4096 keys, uniform and skewed): a CU issues one wave64 vector instruction per clock, so every instruction
//   per row is 0.025 ms per 1e9 rows.  Hence 4 register groups instead of 8 (4 instead of 8 conditional f64 adds, 32-bit
//   packed counters), the predicate evaluated once per row for all tiers, one LDS address per row.
constexpr int K4_TAIL_MAX_RANGES = 2048;  // (2^24 - 4100) / 8192 ids
constexpr int K4_TAIL_MAX_GRID = 2048;    // workgroups of the main kernel the partitioned tier 3 has histogram rows for
constexpr int K4_TAIL_RANGE = 8192;  // ids per range of the partitioned tier 3 = entries of k4_tail_aggregate's LDS table (128 KiB)
// The main kernel counts its tier-3 records per id range in LDS.  With few ranges the 64 lanes of a wave instruction land on
// a handful of counters and same-address LDS atomics serialise, so every range gets 2^k counters picked by lane (summed when
// the workgroup stores its row of the histogram); k shrinks as the ranges grow: at most 3072 counters = 12 KiB behind the
// tier-2 table, which keeps two workgroups on a CU.
__host__ __device__ static inline int k4_hist_copies_log2(int n_ranges) {
  return n_ranges <= 192 ? 4 : n_ranges <= 384 ? 3 : n_ranges <= 768 ? 2 : n_ranges <= 1536 ? 1 : 0;
}
// Direct partition (round 6): the main kernel writes a tier-3 record straight into a CHUNK of its id range -- no compaction,
// no scatter pass.  A workgroup owns `cpw` chunks of K4_CHUNK records (its rows / K4_CHUNK + one per stream: what it can
// fill + one partly filled chunk per stream); a STREAM = (id range, copy) has an LDS cursor word  chunk << 13 | position:
// one returning LDS add per record hands out the slot, the lane that draws position K4_CHUNK opens the stream's next chunk
// (an LDS counter: the chunks are the workgroup's own, no global atomic) and publishes the new cursor; lanes that drew a
// position behind it wait for that.  Few ranges get 2-4 copies each (picked by lane) so that the 64 lanes of an add do not
// queue on a dozen addresses.  Behind the tile loop the workgroup appends its chunks to the per-range chunk lists
// (one global atomic per range and workgroup); k4_tail_aggregate_chunks walks the lists.
constexpr int K4_CHUNK = 2048;        // records per chunk (16 KiB): two per thread of k4_tail_aggregate_chunks
constexpr int K4_POS_BITS = 13;       // position field of a cursor: K4_CHUNK + 1024 lanes' failed draws < 8192
constexpr unsigned K4_POS_MASK = (1u << K4_POS_BITS) - 1u;
constexpr int K4_DIRECT_MAX_RANGES = 128;
static inline int k4_stream_copies_log2(int n_ranges) {
  static const int forced = [] {
    const char* v = getenv("EXON_HIP_K4_STREAM_COPIES_LOG2");  // A/B
    return v && v[0] >= '0' && v[0] <= '3' ? v[0] - '0' : -1;
  }();
  if (forced >= 0) return forced;
  return n_ranges == 1 ? 2 : 0;  // measured (profiles/r6_groupby_direct.md): the main kernel slows down with the streams a wave store spreads over
}
struct K4Tail {                 // tier 3 (unused when NG == NL)
  unsigned long long* counts;   // the caller's [cnn[NG]] [crow[NG]]
  double* sums;                 // the caller's [sum[NG]]
  // partitioned form (round 3, default): instead of three global atomics per row the main kernel COMPACTS the tier-3 rows
  // into a private region per workgroup -- records {id | y-valid << 31, y bits} -- and counts them per id range of
  // K4_TAIL_RANGE ids; k4_tail_scatter then groups the records by range and k4_tail_aggregate runs the LDS table over
  // each range.  rec == nullptr: the atomic form (EXON_HIP_K4_TAIL_ATOMICS=1, and the short tail loop of a launch).
  uint2* rec;                   // [grid][cap_wg]
  unsigned* wg_count;           // [grid] records each workgroup wrote
  unsigned* wg_hist;            // [grid][n_ranges] records per workgroup and id range
  int no_uniform_test;          // EXON_HIP_K4_UNIFORM=0 (A/B): tier 2 never tests for a uniform key
  // direct partition (lists != nullptr; rec = the chunk pool)
  uint2* lists;                 // [n_ranges][list_stride] {chunk, records in it}
  unsigned list_stride;
  unsigned* n_list;             // [n_ranges] entries of each list   (zeroed before the launch)
  unsigned* totals;             // [n_ranges] records of each range  (zeroed before the launch)
  int copies_log2;              // streams per range
};
struct K4Entry {  // tier-2 table entry
  double sum;
  unsigned cnn, crow;
};
template <int G, typename S, bool OVF, bool YI = false>
__global__ __launch_bounds__(S::THREADS) void k4_cmp_avg_by_group_main(
    const float* __restrict__ x, const uint8_t* __restrict__ xvalid, const float* __restrict__ y,
    const uint8_t* __restrict__ yvalid, const int32_t* __restrict__ gid, int64_t n, int32_t klo, int32_t khi,
    int32_t negate, int32_t keymask, int32_t yint, int32_t NG, int32_t NL, unsigned long long* __restrict__ partials,
    int* __restrict__ status, const uint8_t* __restrict__ ones, const FoldArgs fa, const K4Tail tail) {
  constexpr int J = S::J, THREADS = S::THREADS, WAVES = ShapeOf<S>::WAVES, WT = ShapeOf<S>::WAVE_TILE,
                TILE = ShapeOf<S>::TILE;
  // packed 8-bit row counters: 4 fields in a u32 when G <= 4, 8 fields in a u64 otherwise
  constexpr bool NARROW = G <= 4;
  using Packed = typename std::conditional<NARROW, unsigned, unsigned long long>::type;
  constexpr unsigned FMASK = NARROW ? 3u : 7u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double sum[G];
  unsigned cnn[G], crow[G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    sum[k] = 0.0;
    cnn[k] = 0;
    crow[k] = 0;
  }
  Packed prow = 0, pnn = 0;  // field g
  unsigned gmax = 0;
  // tier-2 table (OVF only): NO entries + 64 per-lane dummy entries
  extern __shared__ K4Entry k4_ovf[];
  const int NO = OVF ? NL - G : 0, NOD = NO + 64;
  if (OVF) {
    for (int i = threadIdx.x; i < NOD; i += THREADS) {
      k4_ovf[i].sum = 0.0;
      k4_ovf[i].cnn = 0;
      k4_ovf[i].crow = 0;
    }
    __syncthreads();
  }
  const unsigned dummy = (unsigned)(NO + lane);

  // keymask = 0x7FFFFFFF: x is Float32, compared through its totalOrder key; keymask = 0: x is Int32 (INFO Type=Integer,
  // schema_builder.rs:197-205), the bit pattern IS the key -- integer predicates never pass through f32
  auto passes = [&](float xf, unsigned xv) -> unsigned {
    const int32_t bx = __float_as_int(xf);
    const int32_t kx = bx ^ ((bx >> 31) & keymask);
    const unsigned inr = unsigned(kx >= klo) & unsigned(kx <= khi);
    return xv & (inr ^ (unsigned)negate);
  };
  // AVG's argument widened to f64 (DataFusion casts Float32 AND Int32 arguments of avg to Float64): yint != 0 when the
  // column holds Int32 values (an INFO field of Type=Integer)
  // (the instruction-bound > 8-group variant gets the choice at compile time -- YI -- instead of two conversions and a
  // 64-bit select per row: 3 of its ~50 vector instructions)
  auto ydbl = [&](float yf) -> double {
    if (OVF) return YI ? (double)__float_as_int(yf) : (double)yf;
    return yint ? (double)__float_as_int(yf) : (double)yf;
  };
  // tier 1: register groups (every row); `pass` = the row's predicate
  auto row = [&](unsigned pass, float yf, int32_t g, unsigned yv) {
    unsigned yq = pass & yv;
    gmax = max(gmax, (unsigned)g);  // ids are validated over ALL rows (cheaper than a per-row flag)
    if (OVF) {
      const unsigned in_regs = unsigned((unsigned)g < (unsigned)G);
      pass &= in_regs;
      yq &= in_regs;
    }
    const unsigned sh = ((unsigned)g & FMASK) * 8u;
    prow += (Packed)pass << sh;
    pnn += (Packed)yq << sh;
    const int32_t gq = yq ? g : -1;
    const double yd = ydbl(yf);
#pragma unroll
    for (int k = 0; k < G; ++k) sum[k] += (gq == k) ? yd : 0.0;
  };
  // tier 2: branch-free LDS adds (called for all 64 lanes once any lane of the wave has a row with id >= G)
  auto row_lds = [&](unsigned pass, float yf, int32_t g, unsigned yv) {
    const unsigned in2 = pass & unsigned((unsigned)g - (unsigned)G < (unsigned)NO);
    K4Entry* e = &k4_ovf[in2 ? (unsigned)g - (unsigned)G : dummy];
    atomicAdd(&e->crow, 1u);
    atomicAdd(&e->cnn, in2 & yv);
    atomicAdd(&e->sum, (in2 & yv) ? ydbl(yf) : 0.0);
  };
  // The 4 rows of a lane x 64 lanes = 256 consecutive rows.  When ALL of them carry the same key -- a file whose dominant
  // FILTER list got a dictionary id >= G, or rows sorted by key -- 64 lanes adding to one LDS entry serialise (measured:
  // 2.6 x the step time of the mixed-key case, tools/time_skew.py); then lane 0 adds the group's totals (counts from
  // ballots, the sum from a wave reduction).  The test (4 compares + a vote) is skipped for 15 groups after it found mixed
  // keys, and the mixed-key code is the plain per-row path above: this variant is instruction-bound.
  int uni_skip = tail.no_uniform_test ? 0x7FFFFFFF : 0;  // wave-uniform
  auto rows4_uniform = [&](int kw, unsigned p0, unsigned p1, unsigned p2, unsigned p3, float4 y4, unsigned yv) {
    if ((unsigned)kw - (unsigned)G >= (unsigned)NO) return;  // the one key lives in another tier
    const unsigned n0 = p0 & (yv >> 0 & 1), n1 = p1 & (yv >> 1 & 1), n2 = p2 & (yv >> 2 & 1), n3 = p3 & (yv >> 3 & 1);
    const unsigned rows = (unsigned)(__popcll(__ballot(p0 != 0)) + __popcll(__ballot(p1 != 0)) + __popcll(__ballot(p2 != 0)) + __popcll(__ballot(p3 != 0)));
    const unsigned nn = (unsigned)(__popcll(__ballot(n0 != 0)) + __popcll(__ballot(n1 != 0)) + __popcll(__ballot(n2 != 0)) + __popcll(__ballot(n3 != 0)));
    const double tot = wave_sum(((n0 ? ydbl(y4.x) : 0.0) + (n1 ? ydbl(y4.y) : 0.0)) + ((n2 ? ydbl(y4.z) : 0.0) + (n3 ? ydbl(y4.w) : 0.0)));
    if (lane == 0 && rows != 0) {
      K4Entry* e = &k4_ovf[(unsigned)kw - (unsigned)G];
      atomicAdd(&e->crow, rows);
      atomicAdd(&e->cnn, nn);
      atomicAdd(&e->sum, tot);
    }
  };
  // tier 3, partitioned form: append the row to this workgroup's region (one LDS atomic per wave instruction reserves the
  // slots of all its tier-3 lanes; the id ranges are counted in an LDS histogram, flushed once per workgroup)
  __shared__ unsigned tail_cursor;
  unsigned* tail_hist = reinterpret_cast<unsigned*>(k4_ovf + NOD);  // behind the tier-2 table (dynamic LDS)
  const int n_ranges = (OVF && tail.rec) ? (NG - NL + K4_TAIL_RANGE - 1) / K4_TAIL_RANGE : 0;
  const int hcl = k4_hist_copies_log2(n_ranges);
  const unsigned hcopy = (unsigned)lane & ((1u << hcl) - 1u);
  // rows a workgroup can meet in the tile loop = the size of its region (the host computes the same number)
  const unsigned cap_wg = (unsigned)(((n / TILE + gridDim.x - 1) / gridDim.x) * TILE);
  // direct partition: LDS block behind the tier-2 table = [cursor[S]] [entries[R]] [records[R]] [base[R]] [stream of chunk: u16[cpw]]
  // [list slot of chunk: u16[cpw]]
  const bool direct = OVF && tail.rec && tail.lists;
  const int ccl = tail.copies_log2, n_streams = n_ranges << ccl;
  const unsigned ccopy = (unsigned)lane & ((1u << ccl) - 1u);
  const unsigned cpw = cap_wg / K4_CHUNK + (unsigned)n_streams;  // chunks of this workgroup
  unsigned* const d_cur = tail_hist;
  unsigned* const d_ent = d_cur + n_streams + 64;  // (64 per-lane dummy cursors in between)
  unsigned* const d_rec = d_ent + n_ranges;
  unsigned* const d_base = d_rec + n_ranges;
  unsigned short* const d_cstream = reinterpret_cast<unsigned short*>(d_base + n_ranges);
  unsigned short* const d_cslot = d_cstream + cpw;
  uint2* const d_pool = tail.rec + (size_t)blockIdx.x * cpw * K4_CHUNK;
  if (direct) {
    if (threadIdx.x == 0) tail_cursor = (unsigned)n_streams;  // the next chunk to open; stream s starts in chunk s
    for (int i = threadIdx.x; i < n_streams; i += THREADS) {
      d_cur[i] = (unsigned)i << K4_POS_BITS;
      d_cstream[i] = (unsigned short)i;
    }
    for (int i = threadIdx.x; i < 2 * n_ranges; i += THREADS) d_ent[i] = 0;
    __syncthreads();
  } else if (OVF && tail.rec) {
    if (threadIdx.x == 0) tail_cursor = 0;
    for (int i = threadIdx.x; i < (n_ranges << hcl); i += THREADS) tail_hist[i] = 0;
    __syncthreads();
  }
  // Called by all 64 lanes.  Fast path: the 4 draws of a lane go out back to back (a lane without a tier-3 row draws from
  // its own dummy word: no branch around the LDS atomics), then the records are stored.  Slow path, once per K4_CHUNK
  // records of a stream: the lane that drew position K4_CHUNK opens the stream's next chunk and publishes the new cursor;
  // lanes that drew a position behind it draw again once they see it.  That loop is WAVE-UNIFORM (a vote) and its body
  // straight-line predicated code: a lane never spins on its own -- the lane that opens the chunk may sit in this very
  // wave, and a divergent spin loop would keep it from running.
  auto rows4_tail_direct = [&](unsigned p0, unsigned p1, unsigned p2, unsigned p3, float4 y4, int4 g4, unsigned yv) {
    const unsigned span = (unsigned)(NG - NL);
    const unsigned pp[4] = {p0, p1, p2, p3};
    const int32_t gg[4] = {g4.x, g4.y, g4.z, g4.w};
    const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
    unsigned pend = 0, so[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned d = (unsigned)gg[k] - (unsigned)NL;
      const bool t = pp[k] && d < span;
      pend |= (unsigned)t << k;
      so[k] = t ? ((d / K4_TAIL_RANGE) << ccl) + ccopy : (unsigned)n_streams + (unsigned)lane;
    }
    if (!__any(pend != 0)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = atomicAdd(&d_cur[so[k]], 1u);
    auto store = [&](int k, unsigned chunk, unsigned pos) {
      const unsigned byte_off = ((chunk << 11) | pos) << 3;  // < cpw * 16 KiB: 32 bits
      const v2u_t r = {(unsigned)gg[k] | ((yv >> k & 1u) << 31), (unsigned)__float_as_int(yy[k])};
      *reinterpret_cast<v2u_t*>(reinterpret_cast<char*>(d_pool) + byte_off) = r;  // (the streaming hint changes nothing: 902 / 904 us)
    };
    auto open_next = [&](int k) {
      const unsigned c2 = atomicAdd(&tail_cursor, 1u);
      if (c2 < cpw) {  // cannot fail: cpw covers every row of the tile loop + a partly filled chunk per stream
        d_cstream[c2] = (unsigned short)so[k];
        store(k, c2, 0u);
      }
      __threadfence_block();
      *reinterpret_cast<volatile unsigned*>(&d_cur[so[k]]) = (c2 << K4_POS_BITS) | 1u;
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if ((pend >> k & 1u) && (v[k] & K4_POS_MASK) < (unsigned)K4_CHUNK) {
        store(k, v[k] >> K4_POS_BITS, v[k] & K4_POS_MASK);
        pend &= ~(1u << k);
      }
    }
    if (!__any(pend != 0)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if ((pend >> k & 1u) && (v[k] & K4_POS_MASK) == (unsigned)K4_CHUNK) {
        open_next(k);
        pend &= ~(1u << k);
      }
    }
    while (__any(pend != 0)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!(pend >> k & 1u)) continue;
        if ((*reinterpret_cast<volatile unsigned*>(&d_cur[so[k]]) & K4_POS_MASK) >= (unsigned)K4_CHUNK) continue;  // not yet published
        const unsigned w = atomicAdd(&d_cur[so[k]], 1u);
        const unsigned pos = w & K4_POS_MASK;
        if (pos < (unsigned)K4_CHUNK) {
          store(k, w >> K4_POS_BITS, pos);
          pend &= ~(1u << k);
        } else if (pos == (unsigned)K4_CHUNK) {
          open_next(k);
          pend &= ~(1u << k);
        }
      }
    }
  };
  // (one reservation for the 4 rows of every lane: the returning LDS atomic and the readfirstlane behind it are a round trip
  //  the wave waits for)
  auto rows4_tail_append = [&](unsigned p0, unsigned p1, unsigned p2, unsigned p3, float4 y4, int4 g4, unsigned yv) {
    const unsigned span = (unsigned)(NG - NL);
    const unsigned d0 = (unsigned)g4.x - (unsigned)NL, d1 = (unsigned)g4.y - (unsigned)NL, d2 = (unsigned)g4.z - (unsigned)NL,
                   d3 = (unsigned)g4.w - (unsigned)NL;
    const bool t0 = p0 && d0 < span, t1 = p1 && d1 < span, t2 = p2 && d2 < span, t3 = p3 && d3 < span;
    const unsigned long long m0 = __ballot(t0), m1 = __ballot(t1), m2 = __ballot(t2), m3 = __ballot(t3);
    const unsigned c0 = (unsigned)__popcll(m0), c1 = (unsigned)__popcll(m1), c2 = (unsigned)__popcll(m2), c3 = (unsigned)__popcll(m3);
    if (c0 + c1 + c2 + c3 == 0) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&tail_cursor, c0 + c1 + c2 + c3);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    const unsigned long long below = (1ull << lane) - 1ull;
    uint2* dst = tail.rec + (size_t)blockIdx.x * cap_wg;
    auto put = [&](bool t, unsigned at, unsigned d, int32_t g, float yf, unsigned y1) {
      if (!t) return;
      if (at < cap_wg)  // cannot overflow: cap_wg covers every row this workgroup reads in the tile loop
        dst[at] = uint2{(unsigned)g | (y1 << 31), (unsigned)__float_as_int(yf)};
      atomicAdd(&tail_hist[((d / K4_TAIL_RANGE) << hcl) + hcopy], 1u);
    };
    put(t0, base + (unsigned)__popcll(m0 & below), d0, g4.x, y4.x, yv >> 0 & 1);
    put(t1, base + c0 + (unsigned)__popcll(m1 & below), d1, g4.y, y4.y, yv >> 1 & 1);
    put(t2, base + c0 + c1 + (unsigned)__popcll(m2 & below), d2, g4.z, y4.z, yv >> 2 & 1);
    put(t3, base + c0 + c1 + c2 + (unsigned)__popcll(m3 & below), d3, g4.w, y4.w, yv >> 3 & 1);
  };
  // tier 3, atomic form: global atomics straight into the state (divergent on purpose)
  auto row_tail = [&](unsigned pass, float yf, int32_t g, unsigned yv) {
    if (pass && (unsigned)g >= (unsigned)NL && (unsigned)g < (unsigned)NG) {
      atomicAdd(&tail.counts[NG + g], 1ull);
      if (yv) {
        atomicAdd(&tail.counts[g], 1ull);
        atomicAdd(&tail.sums[g], ydbl(yf));
      }
    }
  };
  auto spill = [&]() {
#pragma unroll
    for (int k = 0; k < G; ++k) {
      crow[k] += (unsigned)(prow >> (8 * k)) & 0xFFu;
      cnn[k] += (unsigned)(pnn >> (8 * k)) & 0xFFu;
    }
    prow = 0;
    pnn = 0;
  };

  const int64_t ntiles = n / TILE;
  int since = 0;  // rows added to the packed counters since the last spill
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wbase = tile * TILE + (int64_t)wave * WT;
    float4 xs[J], ys[J];
    int4 gs[J];
    unsigned xm[J], ym[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t r = wbase + j * 256 + lane * 4;
      xs[j] = ld16<float4>(x + r);
      ys[j] = ld16<float4>(y + r);
      gs[j] = ld16<int4>(gid + r);
      xm[j] = valid4_ones(xvalid, ones, wbase, j, lane);
      ym[j] = valid4_ones(yvalid, ones, wbase, j, lane);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const unsigned p0 = passes(xs[j].x, xm[j] >> 0 & 1), p1 = passes(xs[j].y, xm[j] >> 1 & 1),
                     p2 = passes(xs[j].z, xm[j] >> 2 & 1), p3 = passes(xs[j].w, xm[j] >> 3 & 1);
      // (high-cardinality keys: a wave whose 256 rows hold no register group skips tier 1 -- 4 conditional f64 adds per row)
      const unsigned gn = OVF ? min(min((unsigned)gs[j].x, (unsigned)gs[j].y), min((unsigned)gs[j].z, (unsigned)gs[j].w)) : 0u;
      if (!OVF || NG <= NL || __any(gn < (unsigned)G)) {
        row(p0, ys[j].x, gs[j].x, ym[j] >> 0 & 1);
        row(p1, ys[j].y, gs[j].y, ym[j] >> 1 & 1);
        row(p2, ys[j].z, gs[j].z, ym[j] >> 2 & 1);
        row(p3, ys[j].w, gs[j].w, ym[j] >> 3 & 1);
      }
      if (OVF) {
        // largest id among this lane's 4 rows (unsigned: a negative id is "huge" and is reported through gmax)
        const unsigned gm = max(max((unsigned)gs[j].x, (unsigned)gs[j].y), max((unsigned)gs[j].z, (unsigned)gs[j].w));
        gmax = max(gmax, gm);
        if (__any(gm >= (unsigned)G)) {
          bool uni = false;
          const int kw = __builtin_amdgcn_readfirstlane(gs[j].x);
          if (uni_skip > 0) {
            --uni_skip;
          } else {
            uni = __all((gs[j].x == kw) & (gs[j].y == kw) & (gs[j].z == kw) & (gs[j].w == kw));
            uni_skip = uni ? 0 : 15;
          }
          if (uni) {
            rows4_uniform(kw, p0, p1, p2, p3, ys[j], ym[j]);
          } else {
            row_lds(p0, ys[j].x, gs[j].x, ym[j] >> 0 & 1);
            row_lds(p1, ys[j].y, gs[j].y, ym[j] >> 1 & 1);
            row_lds(p2, ys[j].z, gs[j].z, ym[j] >> 2 & 1);
            row_lds(p3, ys[j].w, gs[j].w, ym[j] >> 3 & 1);
          }
          if (NG > NL && __any(gm >= (unsigned)NL)) {
            if (direct) {
              rows4_tail_direct(p0, p1, p2, p3, ys[j], gs[j], ym[j]);
            } else if (tail.rec) {
              rows4_tail_append(p0, p1, p2, p3, ys[j], gs[j], ym[j]);
            } else {
              row_tail(p0, ys[j].x, gs[j].x, ym[j] >> 0 & 1);
              row_tail(p1, ys[j].y, gs[j].y, ym[j] >> 1 & 1);
              row_tail(p2, ys[j].z, gs[j].z, ym[j] >> 2 & 1);
              row_tail(p3, ys[j].w, gs[j].w, ym[j] >> 3 & 1);
            }
          }
        }
      }
    }
    since += 4 * J;
    if (since > 255 - 4 * J) {
      spill();
      since = 0;
    }
  }
  spill();
  since = 0;
  for (int64_t r = ntiles * TILE + (int64_t)rem_block() * THREADS + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * THREADS) {
    const float yf = y[r];
    const int32_t g = gid[r];
    const unsigned yv = valid1(yvalid, r), pass = passes(x[r], valid1(xvalid, r));
    row(pass, yf, g, yv);
    if (OVF && (unsigned)g >= (unsigned)G) {  // the tail loop is short: plain divergent code
      if ((unsigned)g < (unsigned)NL) {
        if (pass) {
          K4Entry* e = &k4_ovf[g - G];
          atomicAdd(&e->crow, 1u);
          if (yv) {
            atomicAdd(&e->cnn, 1u);
            atomicAdd(&e->sum, ydbl(yf));
          }
        }
      } else if (NG > NL) {
        row_tail(pass, yf, g, yv);
      }
    }
    if (++since == 255) {
      spill();
      since = 0;
    }
  }
  spill();

  if (gmax >= (unsigned)(OVF ? NG : G)) atomicOr(status, 4);
  if (direct) {  // this workgroup's chunks -> the per-range chunk lists
    __syncthreads();
    const unsigned used = min(tail_cursor, cpw);
    for (unsigned j = threadIdx.x; j < used; j += THREADS) {
      const unsigned st = d_cstream[j], v = d_cur[st];
      const unsigned cnt = (v >> K4_POS_BITS) == j ? min(v & K4_POS_MASK, (unsigned)K4_CHUNK) : (unsigned)K4_CHUNK;
      if (cnt) {
        d_cslot[j] = (unsigned short)atomicAdd(&d_ent[st >> ccl], 1u);
        atomicAdd(&d_rec[st >> ccl], cnt);
      }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n_ranges; r += THREADS) {
      if (d_ent[r]) {
        d_base[r] = atomicAdd(&tail.n_list[r], d_ent[r]);
        atomicAdd(&tail.totals[r], d_rec[r]);
      }
    }
    __syncthreads();
    for (unsigned j = threadIdx.x; j < used; j += THREADS) {
      const unsigned st = d_cstream[j], v = d_cur[st], r = st >> ccl;
      const unsigned cnt = (v >> K4_POS_BITS) == j ? min(v & K4_POS_MASK, (unsigned)K4_CHUNK) : (unsigned)K4_CHUNK;
      if (cnt) tail.lists[(size_t)r * tail.list_stride + d_base[r] + d_cslot[j]] = uint2{blockIdx.x * cpw + j, cnt};
    }
  } else if (OVF && tail.rec) {  // what this workgroup compacted: its record count, and its share of the per-range histogram
    __syncthreads();
    if (threadIdx.x == 0) tail.wg_count[blockIdx.x] = min(tail_cursor, cap_wg);
    for (int i = threadIdx.x; i < n_ranges; i += THREADS) {
      unsigned t = 0;
      for (int c = 0; c < (1 << hcl); ++c) t += tail_hist[(i << hcl) + c];
      tail.wg_hist[(size_t)blockIdx.x * n_ranges + i] = t;
    }
  }

  // per-workgroup record: [cnn[RG]] [crow[RG]] [sum[RG]]  (fixed-order reductions for the register groups)
  const int RG = OVF ? NL : G;  // groups per record
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
    const int kind = v / G, g = v - kind * G;  // 0: cnn, 1: crow, 2: sum
    st_agent(&partials[(size_t)blockIdx.x * (3 * RG) + kind * RG + g], out);
  }
  if (OVF) {
    for (int i = threadIdx.x; i < NO; i += THREADS) {
      unsigned long long* rec = partials + (size_t)blockIdx.x * (3 * RG);
      st_agent(&rec[G + i], k4_ovf[i].cnn);
      st_agent(&rec[RG + G + i], k4_ovf[i].crow);
      st_agent(&rec[2 * RG + G + i], (unsigned long long)__double_as_longlong(k4_ovf[i].sum));
    }
  }
  if (fa.ticket) fold_if_last<THREADS>(fa, partials);
}

constexpr int K4_OVF_REGS = 4;                   // register groups of the > 8-group variant
// ids handled by registers + the LDS table
static int k4_lds_groups() {
  static const int ids = [] {
    const char* v = getenv("EXON_HIP_K4_LDS_IDS");  // A/B: entries of the tier-2 table
    const int k = v ? atoi(v) : 0;
    return k >= 64 && k <= 8192 ? k : 4096;
  }();
  return K4_OVF_REGS + ids;
}
#define K4_LDS_GROUPS k4_lds_groups()
static int k4_nl(int n_groups) { return n_groups < K4_LDS_GROUPS ? n_groups : K4_LDS_GROUPS; }

size_t k4_partial_words(const LaunchCfg& cfg, int n_groups) { return (size_t)max_grid(cfg) * 3 * (size_t)k4_nl(n_groups); }

template <int G, typename S, bool OVF, bool YI = false>
static hipError_t k4_launch(hipStream_t s, const LaunchCfg& cfg, int* grid_out, const Workspace& ws, const float* x,
                            const uint8_t* xv, const float* y, const uint8_t* yv, const int32_t* gid, int64_t n,
                            int32_t klo, int32_t khi, int32_t negate, int32_t keymask, int32_t yint, int32_t n_groups,
                            const FoldArgs& fa, const K4Tail& tail) {
  const int nl = k4_nl(n_groups);
  const size_t n_ranges = (OVF && tail.rec) ? (size_t)(n_groups - nl + K4_TAIL_RANGE - 1) / K4_TAIL_RANGE : 0;
  size_t lds = OVF ? (size_t)(nl - G + 64) * sizeof(K4Entry) + (n_ranges << k4_hist_copies_log2((int)n_ranges)) * sizeof(unsigned) : 0;
  if (OVF && tail.rec && tail.lists) {  // the direct partition's block instead of the histogram (an upper bound: the grid is not known yet)
    const int64_t tiles = n / ShapeOf<S>::TILE, g_min = std::max<int64_t>(1, std::min<int64_t>(tiles, cfg.compute_units));
    const size_t n_streams = n_ranges << k4_stream_copies_log2((int)n_ranges);
    const size_t cpw_ub = (size_t)((tiles + g_min - 1) / g_min) * ShapeOf<S>::TILE / K4_CHUNK + n_streams;
    lds = (size_t)(nl - G + 64) * sizeof(K4Entry) + (n_streams + 64 + 3 * n_ranges) * sizeof(unsigned) + 4 * cpw_ub + 8;
  }
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k4_cmp_avg_by_group_main<G, S, OVF, YI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const int grid = grid_for<S>(cfg, n, resident_blocks(k4_cmp_avg_by_group_main<G, S, OVF, YI>, S::THREADS, lds));
  *grid_out = grid;
  hipLaunchKernelGGL((k4_cmp_avg_by_group_main<G, S, OVF, YI>), dim3(grid), dim3(S::THREADS), lds, s, x, xv, y, yv, gid, n,
                     klo, khi, negate, keymask, yint, n_groups, nl, ws.partials, ws.status, reinterpret_cast<const uint8_t*>(ws.status + 8), fa,
                     tail);
  return hipGetLastError();
}

// ---- host: fold (<op>, thr) into an inclusive f32 totalOrder key range --------------------------
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
// smallest key k in [INT32_MIN, INT32_MAX + 1] whose column value compares >= t (or > t) with the literal.
//   Float32 column: the value of key k is the f32 with that totalOrder key, widened to f64, compared in totalOrder;
//   Int32 column (x_is_int): the value IS k.  DataFusion compares Int32 with an Int64 literal as integers and with a
//   Float64 literal after casting the column to Float64; every int32 is exact in f64, so one f64 comparison covers both
//   (an integer literal beyond 2^53 is beyond int32 anyway: the range saturates the same way).  A NaN or -0.0 literal
//   behaves as in arrow-rs' totalOrder float compare: NaN above every number, (f64)0 = +0.0 above -0.0.
static int64_t first_key_not_less(int64_t t, bool strict_greater, bool x_is_int) {
  int64_t lo = INT32_MIN, hi = (int64_t)INT32_MAX + 1;
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    const int64_t km = h_f64_key(x_is_int ? (double)(int32_t)mid : (double)h_key_f32((int32_t)mid));
    const bool ok = strict_greater ? (km > t) : (km >= t);
    if (ok) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}
bool cmp_to_key_range(double thr, int cmp_op, bool x_is_int, int32_t* klo, int32_t* khi, int32_t* negate) {
  const int64_t t = h_f64_key(thr);
  const int64_t ge = first_key_not_less(t, false, x_is_int);  // first key with value >= thr
  const int64_t gt = first_key_not_less(t, true, x_is_int);   // first key with value >  thr
  int64_t lo, hi;
  *negate = 0;
  switch (cmp_op) {
    case 0: lo = gt; hi = INT32_MAX; break;            // >
    case 1: lo = ge; hi = INT32_MAX; break;            // >=
    case 2: lo = INT32_MIN; hi = ge - 1; break;        // <
    case 3: lo = INT32_MIN; hi = gt - 1; break;        // <=
    case 4: lo = ge; hi = gt - 1; break;               // =
    case 5: lo = ge; hi = gt - 1; *negate = 1; break;  // !=
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

// Round 2's path for more groups than the LDS table holds: ONE global atomic triple per passing row, whatever its id.
// Kept behind EXON_HIP_K4_GLOBAL_ONLY=1 as the A/B baseline of the tiered kernel above (profiles/r3_groupby.md).
__global__ __launch_bounds__(256) void k4_cmp_avg_by_group_global(const float* __restrict__ x, const uint8_t* __restrict__ xvalid,
                                                                  const float* __restrict__ y, const uint8_t* __restrict__ yvalid,
                                                                  const int32_t* __restrict__ gid, int64_t n, int32_t klo, int32_t khi,
                                                                  int32_t negate, int32_t keymask, int32_t yint, int32_t NG,
                                                                  unsigned long long* __restrict__ counts, double* __restrict__ sums,
                                                                  int* __restrict__ status) {
  bool bad = false;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
    const unsigned g = (unsigned)gid[r];
    if (g >= (unsigned)NG) {  // ids are validated over ALL rows, like the LDS paths do
      bad = true;
      continue;
    }
    const int32_t bx = __float_as_int(x[r]);
    const int32_t kx = bx ^ ((bx >> 31) & keymask);
    const unsigned inr = unsigned(kx >= klo) & unsigned(kx <= khi);
    if (!(valid1(xvalid, r) && (inr ^ (unsigned)negate))) continue;
    atomicAdd(&counts[NG + g], 1ull);
    if (valid1(yvalid, r)) {
      atomicAdd(&counts[g], 1ull);
      atomicAdd(&sums[g], yint ? (double)__float_as_int(y[r]) : (double)y[r]);
    }
  }
  if (bad) atomicOr(&status[0], 4);
}

// records [blocks][3 x RG] (kinds cnn, crow, sum over the first RG ids) ADDED into a state of NG > RG groups
// ([cnn[NG]] [crow[NG]] | [sum[NG]]): the fold of the tiered kernel when tier 3 exists.  One thread per (kind, id),
// blocks summed in order.  The state was zeroed (overwrite) or holds earlier launches; tier 3 has added to it already.
__global__ __launch_bounds__(1024) void k4_finalize_head(const unsigned long long* __restrict__ partials, int nblocks, int RG, int NG,
                                                         unsigned long long* __restrict__ counts, double* __restrict__ sums) {
  // 32 values x 32 segments of the workgroup range per block, 8 loads in flight per thread: the records are 3 x RG x 8 bytes
  // apart (98 KB at RG = 4100), every load of a thread is a new page, and the fold is latency-bound (8 segments x 4 loads
  // took 0.10 ms for 512 records, a seventh of a 2^28-row launch's main kernel)
  __shared__ unsigned long long red[32][32];
  const int vi = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const int v = blockIdx.x * 32 + vi, V = 3 * RG;
  const int kind = v < V ? v / RG : 0, g = v - kind * RG;
  const int per = (nblocks + 31) / 32;
  const int b0 = seg * per, b1 = min(nblocks, b0 + per);
  unsigned long long acc_i = 0;
  double acc_f = 0.0;
  if (v < V) {
    const unsigned long long* p = partials + v;
    int b = b0;
    if (kind < 2) {
      for (; b + 8 <= b1; b += 8) {
        unsigned long long t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = p[(size_t)(b + k) * V];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc_i += t[k];
      }
      for (; b < b1; ++b) acc_i += p[(size_t)b * V];
    } else {
      for (; b + 8 <= b1; b += 8) {
        unsigned long long t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = p[(size_t)(b + k) * V];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc_f += __longlong_as_double((long long)t[k]);
      }
      for (; b < b1; ++b) acc_f += __longlong_as_double((long long)p[(size_t)b * V]);
    }
  }
  red[seg][vi] = kind < 2 ? acc_i : (unsigned long long)__double_as_longlong(acc_f);
  __syncthreads();
  if (seg == 0 && v < V) {
    if (kind < 2) {
      unsigned long long t = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += red[k][vi];
      if (t) counts[(size_t)kind * NG + g] += t;
    } else {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += __longlong_as_double((long long)red[k][vi]);
      sums[g] += t;
    }
  }
}

// ---- partitioned tier 3 (ids >= K4_LDS_GROUPS): scatter by id range, then the LDS table over each range ---------------------
// wg_hist[w][r] (records of workgroup w in range r) -> in place, the exclusive prefix over w; totals[r] = the range's records.
// One workgroup per range; grid_main <= 2048 (two rows per thread at most).
__global__ __launch_bounds__(1024) void k4_tail_wg_scan(unsigned* __restrict__ wg_hist, int grid_main, int n_ranges, unsigned* __restrict__ totals) {
  __shared__ unsigned part[1024];
  const int r = blockIdx.x;
  const int per = (grid_main + 1023) / 1024;
  const int w0 = threadIdx.x * per, w1 = min(grid_main, w0 + per);
  unsigned sum = 0;
  for (int w = w0; w < w1; ++w) sum += wg_hist[(size_t)w * n_ranges + r];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int w = w0; w < w1; ++w) {
    const unsigned c = wg_hist[(size_t)w * n_ranges + r];
    wg_hist[(size_t)w * n_ranges + r] = run;
    run += c;
  }
  if (threadIdx.x == 1023) totals[r] = part[1023];
}
// offsets[r] = exclusive prefix of totals[0 .. n_ranges), offsets[n_ranges] = all records; slice_start[r] = exclusive prefix
// of the workgroups k4_tail_aggregate gives range r: its share of `target` by records (at least one; none for an empty
// range), so that a skewed key distribution does not leave the chip to the few heavy ranges.  One workgroup; n_ranges <= 2048.
__global__ __launch_bounds__(1024) void k4_tail_offsets(const unsigned* __restrict__ totals, int n_ranges, unsigned* __restrict__ offsets,
                                                        int target, unsigned* __restrict__ slice_start) {
  __shared__ unsigned part[1024];
  const int per = (n_ranges + 1023) / 1024;
  const int r0 = threadIdx.x * per, r1 = min(n_ranges, r0 + per);
  unsigned sum = 0;
  for (int r = r0; r < r1; ++r) sum += totals[r];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const unsigned all = part[1023];
  unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int r = r0; r < r1; ++r) {
    offsets[r] = run;
    run += totals[r];
  }
  if (threadIdx.x == 1023) offsets[n_ranges] = all;
  __syncthreads();
  auto slices_of = [&](unsigned len) -> unsigned {
    if (len == 0) return 0u;
    const unsigned long long q = ((unsigned long long)len * (unsigned)target + all - 1) / all;
    return (unsigned)(q < 1 ? 1 : q);
  };
  sum = 0;
  for (int r = r0; r < r1; ++r) sum += slices_of(totals[r]);
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int r = r0; r < r1; ++r) {
    slice_start[r] = run;
    run += slices_of(totals[r]);
  }
  if (threadIdx.x == 1023) slice_start[n_ranges] = part[1023];
}
// One workgroup per source region (= per workgroup of the main kernel).  Where this workgroup's records of range r go is
// known before it starts -- offsets[r] + the records the workgroups before it hold of r (k4_tail_wg_scan) -- so an LDS
// cursor per range hands out the output slots: no global atomics (512 workgroups drawing from a dozen global cursors
// serialised: 12 ns per draw on one address), no barriers inside the loop, and the next 8 records of a thread are in flight
// while the current ones are placed.  8 B read + 8 B written per record.
constexpr int K4_SCATTER_PER = 8;
__global__ __launch_bounds__(1024) void k4_tail_scatter(const uint2* __restrict__ rec, const unsigned* __restrict__ wg_count, unsigned cap_wg,
                                                        int NL, int n_ranges, const unsigned* __restrict__ offsets,
                                                        const unsigned* __restrict__ wg_excl, uint2* __restrict__ out) {
  extern __shared__ unsigned sc_pos[];  // [n_ranges] next output slot of this workgroup in each range
  for (int i = threadIdx.x; i < n_ranges; i += 1024) sc_pos[i] = offsets[i] + wg_excl[(size_t)blockIdx.x * n_ranges + i];
  __syncthreads();
  const uint2* src = rec + (size_t)blockIdx.x * cap_wg;
  const unsigned total = wg_count[blockIdx.x];
  constexpr unsigned CH = 1024 * K4_SCATTER_PER;
  uint2 v[K4_SCATTER_PER], nx[K4_SCATTER_PER];
#pragma unroll
  for (int k = 0; k < K4_SCATTER_PER; ++k) {
    const unsigned i = (unsigned)k * 1024u + threadIdx.x;
    v[k] = i < total ? ldnt8(src + i) : uint2{0, 0};
  }
  for (unsigned c0 = 0; c0 < total; c0 += CH) {
#pragma unroll
    for (int k = 0; k < K4_SCATTER_PER; ++k) {
      const unsigned i = c0 + CH + (unsigned)k * 1024u + threadIdx.x;
      nx[k] = i < total ? ldnt8(src + i) : uint2{0, 0};
    }
#pragma unroll
    for (int k = 0; k < K4_SCATTER_PER; ++k) {
      const unsigned i = c0 + (unsigned)k * 1024u + threadIdx.x;
      if (i < total) out[atomicAdd(&sc_pos[((v[k].x & 0x7FFFFFFFu) - (unsigned)NL) / K4_TAIL_RANGE], 1u)] = v[k];
    }
#pragma unroll
    for (int k = 0; k < K4_SCATTER_PER; ++k) v[k] = nx[k];
  }
}
// A workgroup = one slice of one range's records (slice_start, k4_tail_offsets): the tier-2 LDS table over K4_TAIL_RANGE ids,
// flushed into the caller's state with one global atomic triple per id the slice touched.  One 128 KiB table = one workgroup
// per CU, so the loads have to come from inside the thread: 8 records in flight each.  count(*) and count(y) of an entry are
// one 64-bit LDS add (count(y) in the low word: a slice holds < 2^28 records, it cannot carry).
__global__ __launch_bounds__(1024) void k4_tail_aggregate(const uint2* __restrict__ recs, const unsigned* __restrict__ offsets,
                                                          const unsigned* __restrict__ slice_start, int n_ranges, int NL, int NG,
                                                          int yint, unsigned long long* __restrict__ counts, double* __restrict__ sums) {
  __shared__ K4Entry tab[K4_TAIL_RANGE];
  if (blockIdx.x >= slice_start[n_ranges]) return;
  int rlo = 0, rhi = n_ranges - 1;  // the last range whose first slice is <= blockIdx.x (empty ranges have no slices)
  while (rlo < rhi) {
    const int mid = (rlo + rhi + 1) >> 1;
    if (slice_start[mid] <= blockIdx.x) rlo = mid;
    else rhi = mid - 1;
  }
  const int range = rlo;
  const unsigned n_slices = slice_start[range + 1] - slice_start[range], slice = blockIdx.x - slice_start[range];
  const unsigned lo = offsets[range], hi = offsets[range + 1];
  const unsigned len = hi - lo, per = (len + n_slices - 1) / n_slices;
  const unsigned a = lo + slice * per, b = min(hi, a + per);
  if (a >= b) return;
  for (int i = threadIdx.x; i < K4_TAIL_RANGE; i += 1024) {
    tab[i].sum = 0.0;
    tab[i].cnn = 0;
    tab[i].crow = 0;
  }
  __syncthreads();
  const unsigned id0 = (unsigned)NL + (unsigned)range * (unsigned)K4_TAIL_RANGE;
  constexpr int PER = 8;
  for (unsigned i0 = a; i0 < b; i0 += PER * 1024u) {
    uint2 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const unsigned i = i0 + (unsigned)k * 1024u + threadIdx.x;
      v[k] = i < b ? ldnt8(recs + i) : uint2{0xFFFFFFFFu, 0};
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const unsigned i = i0 + (unsigned)k * 1024u + threadIdx.x;
      if (i >= b) continue;
      K4Entry* e = &tab[(v[k].x & 0x7FFFFFFFu) - id0];
      const unsigned yv = v[k].x >> 31;
      atomicAdd(reinterpret_cast<unsigned long long*>(&e->cnn), (1ull << 32) | yv);  // {cnn, crow} are adjacent, cnn first
      atomicAdd(&e->sum, yv ? (yint ? (double)(int32_t)v[k].y : (double)__uint_as_float(v[k].y)) : 0.0);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K4_TAIL_RANGE; i += 1024) {
    const unsigned g = id0 + (unsigned)i;
    if (g >= (unsigned)NG || tab[i].crow == 0) continue;
    atomicAdd(&counts[NG + g], (unsigned long long)tab[i].crow);
    if (tab[i].cnn) {
      atomicAdd(&counts[g], (unsigned long long)tab[i].cnn);
      atomicAdd(&sums[g], tab[i].sum);
    }
  }
}

// The direct partition's aggregate: the same table, fed from the chunk list of the workgroup's range (slices of a range
// take the list's entries in turns, 8 chunks = 16 records per thread in flight).
__global__ __launch_bounds__(1024) void k4_tail_aggregate_chunks(const uint2* __restrict__ pool, const uint2* __restrict__ lists, unsigned list_stride,
                                                                 const unsigned* __restrict__ n_list, const unsigned* __restrict__ slice_start,
                                                                 int n_ranges, int NL, int NG, int yint, unsigned long long* __restrict__ counts,
                                                                 double* __restrict__ sums) {
  static_assert(K4_CHUNK == 2048, "two records per thread and chunk");
  // sums and counts in separate arrays: a wave instruction's 64 adds then spread over all LDS banks (16-byte entries: over half)
  __shared__ double tab_sum[K4_TAIL_RANGE];
  __shared__ unsigned long long tab_cnt[K4_TAIL_RANGE];  // count(*) << 32 | count(y): a slice holds < 2^28 records
  if (blockIdx.x >= slice_start[n_ranges]) return;
  int rlo = 0, rhi = n_ranges - 1;
  while (rlo < rhi) {
    const int mid = (rlo + rhi + 1) >> 1;
    if (slice_start[mid] <= blockIdx.x) rlo = mid;
    else rhi = mid - 1;
  }
  const int range = rlo;
  const unsigned n_slices = slice_start[range + 1] - slice_start[range], slice = blockIdx.x - slice_start[range];
  const unsigned nl = n_list[range];
  constexpr unsigned Q = 8;  // chunks per trip: one 16-byte load (2 records) per thread and chunk, 128 B in flight per thread
  if (slice * Q >= nl) return;
  for (int i = threadIdx.x; i < K4_TAIL_RANGE; i += 1024) {
    tab_sum[i] = 0.0;
    tab_cnt[i] = 0;
  }
  __syncthreads();
  const unsigned id0 = (unsigned)NL + (unsigned)range * (unsigned)K4_TAIL_RANGE;
  const uint2* L = lists + (size_t)range * list_stride;
  auto add = [&](unsigned key, unsigned ybits) {
    const unsigned i = (key & 0x7FFFFFFFu) - id0, yv = key >> 31;
    atomicAdd(&tab_cnt[i], (1ull << 32) | yv);
    atomicAdd(&tab_sum[i], yv ? (yint ? (double)(int32_t)ybits : (double)__uint_as_float(ybits)) : 0.0);
  };
  for (unsigned k0 = slice * Q; k0 < nl; k0 += n_slices * Q) {
    uint4 v[Q];
    unsigned cnt[Q];
#pragma unroll
    for (unsigned q = 0; q < Q; ++q) {
      const uint2 e = k0 + q < nl ? L[k0 + q] : uint2{0, 0};
      cnt[q] = e.y;
      v[q] = 2u * threadIdx.x < e.y ? ld16<uint4>(pool + (size_t)e.x * K4_CHUNK + 2u * threadIdx.x) : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (unsigned q = 0; q < Q; ++q) {
      if (2u * threadIdx.x < cnt[q]) add(v[q].x, v[q].y);
      if (2u * threadIdx.x + 1u < cnt[q]) add(v[q].z, v[q].w);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K4_TAIL_RANGE; i += 1024) {
    const unsigned g = id0 + (unsigned)i;
    const unsigned long long c = tab_cnt[i];
    if (g >= (unsigned)NG || c == 0) continue;
    atomicAdd(&counts[NG + g], c >> 32);
    if (c & 0xFFFFFFFFull) {
      atomicAdd(&counts[g], c & 0xFFFFFFFFull);
      atomicAdd(&sums[g], tab_sum[i]);
    }
  }
}

// rows per launch of the partitioned tier 3: its scratch is 2 x 8 bytes per row of a launch, so a longer table is cut
// into launches of this many rows (a multiple of every tile size and of 8: column and bitmap pointers stay aligned)
constexpr int64_t K4_TAIL_CHUNK_ROWS = int64_t(1) << 28;
constexpr int64_t K4_TAIL_SLACK = 2048 * 2048;  // grid x tile of the largest launch shape: rounding of the per-workgroup regions
size_t k4_tail_records(int64_t n, int n_groups) {
  if (n_groups <= K4_LDS_GROUPS) return 0;
  return (size_t)(std::min<int64_t>(n, K4_TAIL_CHUNK_ROWS) + K4_TAIL_SLACK);
}
static bool k4_tail_atomics_forced() {
  static const bool on = [] {
    const char* v = getenv("EXON_HIP_K4_TAIL_ATOMICS");  // A/B: tier 3 as global atomics (round 3's first version)
    return v && v[0] == '1';
  }();
  return on;
}

static hipError_t k4_one_launch(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const float* x, const uint8_t* x_valid,
                                const float* y, const uint8_t* y_valid, const int32_t* gid, int64_t n, int32_t klo, int32_t khi,
                                int32_t negate, int32_t keymask, int32_t yint, int n_groups, int64_t* d_counts, double* d_sums) {
  const int nl = k4_nl(n_groups);
  const bool has_tail = n_groups > nl;
  // K4 keeps the 16384-row tile whenever every CU gets one: measured on MI355X at 1e7 rows J = 4 27.8 us, J = 2 31.5 us
  // (its per-tile bookkeeping -- counter spills, 15-value reductions -- outweighs the better tile balance that pays for K2)
  const bool big = n / ShapeOf<ShapeBig>::TILE >= (int64_t)cfg.compute_units && pick_shape(cfg, n) != SHAPE_SMALL;
  int grid = 1;
  hipError_t e;
  const FoldArgs fa = has_tail ? FoldArgs{nullptr, nullptr, nullptr, 0, 0, 0}
                               : fold_args(cfg, ws, 3 * n_groups, 2 * n_groups, d_counts, d_sums);
  static const int no_uni = [] {
    const char* v = getenv("EXON_HIP_K4_UNIFORM");
    return v && v[0] == '0' ? 1 : 0;
  }();
  K4Tail tail{reinterpret_cast<unsigned long long*>(d_counts), d_sums, nullptr, nullptr, nullptr, no_uni, nullptr, 0, nullptr, nullptr, 0};
  const int n_ranges = has_tail ? (n_groups - nl + K4_TAIL_RANGE - 1) / K4_TAIL_RANGE : 0;
  // partitioned tier 3 when its scratch is there (capi.cpp sizes it with k4_tail_records) and the launch has whole tiles
  const bool partition = has_tail && !k4_tail_atomics_forced() && ws.tail_rec_a && ws.tail_rec_b && ws.tail_u32 &&
                         ws.tail_capacity >= (size_t)(n + K4_TAIL_SLACK) && n_ranges <= K4_TAIL_MAX_RANGES &&
                         max_grid(cfg) <= K4_TAIL_MAX_GRID;
  // the direct partition (round 6) when the chunk pool + the chunk lists fit the two record buffers (one allocation)
  static const bool scatter_forced = [] {
    const char* v = getenv("EXON_HIP_K4_TAIL_SCATTER");  // A/B: round 3's compact -> scatter -> aggregate
    return v && v[0] == '1';
  }();
  const int64_t tile_rows = big ? ShapeOf<ShapeBigJ2>::TILE : ShapeOf<ShapeSmall>::TILE;
  const size_t n_streams = (size_t)n_ranges << k4_stream_copies_log2(n_ranges);
  // chunks of all workgroups, at most: their rows (rounded up to tiles per workgroup) + one per stream and workgroup
  const int64_t grid_ub = big ? cfg.compute_units : std::min<int64_t>(max_grid(cfg), n / tile_rows + 1);  // grid_for's bounds
  const size_t pool_chunks = (size_t)(n + grid_ub * tile_rows) / K4_CHUNK + (size_t)grid_ub * n_streams;
  const int64_t tiles_all = n / tile_rows, grid_min = std::max<int64_t>(1, std::min<int64_t>(tiles_all, cfg.compute_units));
  const size_t cpw_max = (size_t)((tiles_all + grid_min - 1) / grid_min) * tile_rows / K4_CHUNK + n_streams;
  const bool direct = partition && !scatter_forced && n_ranges <= K4_DIRECT_MAX_RANGES && ws.tail_rec_b == ws.tail_rec_a + ws.tail_capacity &&
                      pool_chunks * K4_CHUNK + (size_t)n_ranges * pool_chunks <= 2 * ws.tail_capacity &&
                      cpw_max < 65536;  // chunk numbers inside a workgroup are 16-bit
  unsigned *wg_count = nullptr, *totals = nullptr, *offsets = nullptr, *slice_start = nullptr, *wg_hist = nullptr;
  if (partition) {  // every word the kernels below read is written by the launch before it: nothing to clear
    wg_count = ws.tail_u32;                       // [K4_TAIL_MAX_GRID]
    totals = wg_count + K4_TAIL_MAX_GRID;         // [K4_TAIL_MAX_RANGES]
    offsets = totals + K4_TAIL_MAX_RANGES;        // [K4_TAIL_MAX_RANGES + 1]
    slice_start = offsets + K4_TAIL_MAX_RANGES + 1;      // [K4_TAIL_MAX_RANGES + 1]
    wg_hist = slice_start + K4_TAIL_MAX_RANGES + 1;      // [grid][n_ranges]
    tail.rec = ws.tail_rec_a;
    tail.wg_count = wg_count;
    tail.wg_hist = wg_hist;
    if (direct) {
      tail.lists = ws.tail_rec_a + pool_chunks * K4_CHUNK;
      tail.list_stride = (unsigned)pool_chunks;
      tail.n_list = wg_count;
      tail.totals = totals;
      tail.copies_log2 = k4_stream_copies_log2(n_ranges);
      hipError_t ez = hipMemsetAsync(wg_count, 0, (size_t)(K4_TAIL_MAX_GRID + K4_TAIL_MAX_RANGES) * 4, s);  // list lengths + range totals
      if (ez != hipSuccess) return ez;
    }
  }
  // (Round 4 also built the partition INSIDE the main kernel -- records grouped by id range per tile, no scatter pass -- bit-identical
  // and not faster: 5.84 -> 5.82 ms per 1e9 rows at 1e5 uniform keys, 3.84 -> 5.19 ms at zipf keys; the barrier per 8192-row tile
  // costs more than the scatter pass saves.  Removed in round 5; the measurements are in profiles/r4_groupby_binned.md.)
  switch (n_groups) {
#define EXON_K4_CASE(GG)                                                                                                        \
  case GG:                                                                                                                      \
    e = big ? k4_launch<GG, ShapeBig, false>(s, cfg, &grid, ws, x, x_valid, y, y_valid, gid, n, klo, khi, negate, keymask, yint, GG, fa, tail)   \
            : k4_launch<GG, ShapeSmall, false>(s, cfg, &grid, ws, x, x_valid, y, y_valid, gid, n, klo, khi, negate, keymask, yint, GG, fa, tail); \
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
    default:  // > 8 groups: 4 in registers + LDS table (+ tier 3 beyond K4_LDS_GROUPS ids)
      // (16384-row tiles fit too -- 124 VGPRs -- and measured the same: profiles/r3_groupby.md)
#define EXON_K4_OVF(SHAPE, YI) k4_launch<K4_OVF_REGS, SHAPE, true, YI>(s, cfg, &grid, ws, x, x_valid, y, y_valid, gid, n, klo, khi, negate, keymask, yint, n_groups, fa, tail)
      e = big ? (yint ? EXON_K4_OVF(ShapeBigJ2, true) : EXON_K4_OVF(ShapeBigJ2, false)) : (yint ? EXON_K4_OVF(ShapeSmall, true) : EXON_K4_OVF(ShapeSmall, false));
#undef EXON_K4_OVF
      break;
  }
  if (e != hipSuccess) return e;
  if (has_tail) {
    hipLaunchKernelGGL(k4_finalize_head, dim3((3 * nl + 31) / 32), dim3(1024), 0, s, ws.partials, grid, nl, n_groups,
                       reinterpret_cast<unsigned long long*>(d_counts), d_sums);
    static const int rounds_d = [] {
      const char* v = getenv("EXON_HIP_K4_TAIL_ROUNDS");
      const int k = v ? atoi(v) : 0;
      return k >= 1 && k <= 16 ? k : 1;
    }();
    if (direct) {
      const int target = std::max(1, rounds_d * cfg.compute_units - n_ranges);
      hipLaunchKernelGGL(k4_tail_offsets, dim3(1), dim3(1024), 0, s, totals, n_ranges, offsets, target, slice_start);
      hipLaunchKernelGGL(k4_tail_aggregate_chunks, dim3(target + n_ranges), dim3(1024), 0, s, ws.tail_rec_a, tail.lists, tail.list_stride, tail.n_list,
                         slice_start, n_ranges, nl, n_groups, yint, reinterpret_cast<unsigned long long*>(d_counts), d_sums);
    } else if (partition) {
      const int64_t tile = big ? ShapeOf<ShapeBigJ2>::TILE : ShapeOf<ShapeSmall>::TILE;
      const unsigned cap_wg = (unsigned)(((n / tile + grid - 1) / grid) * tile);  // the main kernel's formula
      hipLaunchKernelGGL(k4_tail_wg_scan, dim3(n_ranges), dim3(1024), 0, s, wg_hist, grid, n_ranges, totals);
      // k4_tail_aggregate holds one workgroup per CU (128 KiB table): `rounds` workgroups per CU in all, shared out over the
      // ranges by their records -- whole rounds, because a workgroup more than a multiple of the CUs is a round more.
      // One round measured best (1e5 zipf keys: 3.90 / 4.07 / 4.32 / 4.54 ms per 1e9 rows for 1 / 2 / 3 / 4: every
      // workgroup zeroes and flushes a whole table)
      static const int rounds = [] {
        const char* v = getenv("EXON_HIP_K4_TAIL_ROUNDS");  // A/B
        const int k = v ? atoi(v) : 0;
        return k >= 1 && k <= 16 ? k : 1;
      }();
      const int target = std::max(1, rounds * cfg.compute_units - n_ranges);  // + at most one per range from rounding up
      hipLaunchKernelGGL(k4_tail_offsets, dim3(1), dim3(1024), 0, s, totals, n_ranges, offsets, target, slice_start);
      hipLaunchKernelGGL(k4_tail_scatter, dim3(grid), dim3(1024), (size_t)n_ranges * 4, s, ws.tail_rec_a, wg_count, cap_wg, nl, n_ranges, offsets,
                         wg_hist, ws.tail_rec_b);
      hipLaunchKernelGGL(k4_tail_aggregate, dim3(target + n_ranges), dim3(1024), 0, s, ws.tail_rec_b, offsets, slice_start, n_ranges, nl, n_groups, yint,
                         reinterpret_cast<unsigned long long*>(d_counts), d_sums);
    }
    return hipGetLastError();
  }
  // per-workgroup records are [cnn[G]] [crow[G]] [sum[G]] = the state layout [counts[2G]] [sums[G]]
  return run_finalize(s, cfg, ws, grid, 3 * n_groups, 2 * n_groups, d_counts, d_sums);
}

hipError_t launch_cmp_avg_by_group(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const float* x,
                                   const uint8_t* x_valid, const float* y, const uint8_t* y_valid, const int32_t* gid,
                                   int64_t n, double thr, int cmp_op, int n_groups, int64_t* d_counts,
                                   double* d_sums) {
  if (n <= 0) return hipSuccess;
  if (n_groups < 1) return hipErrorInvalidValue;
  int32_t klo, khi, negate;
  if (!cmp_to_key_range(thr, cmp_op, cfg.x_is_int, &klo, &khi, &negate)) return hipErrorInvalidValue;
  const int32_t keymask = cfg.x_is_int ? 0 : 0x7FFFFFFF, yint = cfg.y_is_int ? 1 : 0;
  const bool has_tail = n_groups > k4_nl(n_groups);
  static const bool global_only = [] {
    const char* v = getenv("EXON_HIP_K4_GLOBAL_ONLY");
    return v && v[0] == '1';
  }();
  if (has_tail && cfg.overwrite) {  // tier 3 (and k4_finalize_head) ADD to the caller's arrays
    hipError_t e0 = hipMemsetAsync(d_counts, 0, (size_t)n_groups * 16, s);
    if (e0 == hipSuccess) e0 = hipMemsetAsync(d_sums, 0, (size_t)n_groups * 8, s);
    if (e0 != hipSuccess) return e0;
  }
  if (has_tail && global_only) {
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)cfg.compute_units * 32);
    hipLaunchKernelGGL(k4_cmp_avg_by_group_global, dim3(grid), dim3(256), 0, s, x, x_valid, y, y_valid, gid, n, klo, khi, negate, keymask, yint, n_groups,
                       reinterpret_cast<unsigned long long*>(d_counts), d_sums, ws.status);
    return hipGetLastError();
  }
  if (!has_tail || n <= K4_TAIL_CHUNK_ROWS)
    return k4_one_launch(s, cfg, ws, x, x_valid, y, y_valid, gid, n, klo, khi, negate, keymask, yint, n_groups, d_counts, d_sums);
  // a table longer than the tier-3 scratch covers: launches of K4_TAIL_CHUNK_ROWS rows, all accumulating (the state was
  // cleared above when the caller asked for overwrite)
  LaunchCfg acc = cfg;
  acc.overwrite = false;
  for (int64_t r0 = 0; r0 < n; r0 += K4_TAIL_CHUNK_ROWS) {
    const int64_t m = std::min<int64_t>(K4_TAIL_CHUNK_ROWS, n - r0);
    hipError_t e = k4_one_launch(s, acc, ws, x + r0, x_valid ? x_valid + (r0 >> 3) : nullptr, y + r0, y_valid ? y_valid + (r0 >> 3) : nullptr, gid + r0, m,
                                 klo, khi, negate, keymask, yint, n_groups, d_counts, d_sums);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// K5 qual_pos_hist
//   quality_scores_to_list (char - 33; exon-core/src/udfs/sequence/quality_score_string_to_list.rs:83-86)
//   + unnest with ordinality + GROUP BY (position, score) COUNT(*), reported on the raw byte
//   (Phred = bin - 33).  104 B/read at L = 100: i32 offset + L quality bytes.
//
//   What limits it (profiles/r2_tuning.md, tools/lds_atomic_rate.hip, tools/tune_k5.hip): one ds_add_u32 per byte is
//   unavoidable, and a wave instruction of them costs 4.2 LDS clocks when the 64 addresses differ (15 bytes/clock/CU =
//   9.3 TB/s on the chip, above what HBM delivers) -- but (a) two lanes of one 32-lane half on the SAME address serialise
//   (6-7 clocks with many such pairs; equal addresses in different halves are free), (b) data-dependent banks cost 6
//   clocks, and (c) the vector ALU competes: round 1's path A spent 22 VALU instructions per dword (position bookkeeping,
//   shift/mask/multiply-add per byte) and ran at 5.5 TB/s.  One 1024-thread workgroup per CU owns a u32 LDS histogram of
//   the ASCII half of the byte range (quality strings are Phred+33 <= 126; a byte >= 128 goes straight to a global
//   atomic).  Three paths inside ONE kernel, chosen ON THE DEVICE from `k5_scan_offsets` (no host round trip):
//     A  ANY uniform read length 9 <= L <= 310 whose position pattern repeats after P = L / gcd(L, 4) >= 8 dwords, any base
//        alignment (round 3; round 2: L % 4 == 0, 32 <= L <= 256, 4-byte aligned).  The bytes are read as 256-byte rows of
//        MEMORY; a wave owns the rows of one residue class, so every lane keeps ONE dword-of-period d (4 fixed positions) and
//        one wrap count w for the whole launch: no per-iteration position arithmetic.  Table [5 planes][byte < 128][64 columns]
//        u32 (160 KiB), position p in slot (p & 3) ceil(L / 4) + (p >> 2) (+ a copy chosen by w when P < 32: a 32-lane half
//        never holds two equal addresses); a bin's address is plane << 15 | byte << 8 | column << 2: ONE v_perm_b32 per byte.
//        No branch per dword, a rolling window of 24 loads: 7 VALU instructions per dword, all 64 lanes busy: 6.4-6.5 TB/s.
//        Details at the head of the path in k5_main.
//     B  what is left of the uniform lengths (P < 8): 16-byte chunk per lane, h[p][129] (+1 pad: bank = p + byte).
//     G  ragged reads: a half-wave per read, one (unaligned) dword per lane, 8 reads in flight; byte-major layout
//        h[byte][perm(p)], perm(p) = (p & 3) * LP/4 + (p >> 2), when lmax <= LP.
//   Paths A/B never touch the offsets buffer again (they use L), so HBM traffic is 4 + L bytes per read.
// ------------------------------------------------------------------------------------------------
constexpr int K5_THREADS = 1024;
constexpr int K5_PT_MAX = 310;  // positions kept in LDS by the [p][129] layout: 310 * 129 * 4 B = 159,960 B
constexpr int K5_JA = 24, K5_JB = 4;
constexpr int K5_A_WORDS = 5 * 128 * 64;  // path A table: [plane][byte][column], 320 slots
// The workgroup's histogram block, STATIC so that its LDS address is the constant 0: with `extern __shared__` the compiler
// keeps one `v_add_u32 v, <lds base>, v` per atomic after the v_perm (4 of path A's 10 vector instructions per dword;
// tools/tune_k5.hip variant K: +3 %).  Sized for the largest layout: [K5_PT_MAX][129] words.
constexpr int K5_LDS_WORDS = 40960;  // all 160 KiB of the CU
static_assert(K5_LDS_WORDS >= K5_PT_MAX * 129 && K5_LDS_WORDS >= K5_A_WORDS && K5_LDS_WORDS >= 128 * 256 + 64, "K5 LDS block");
__shared__ __attribute__((aligned(16))) unsigned k5_h[K5_LDS_WORDS];

// flags[0]: bit 0 = read lengths differ inside a chunk or between chunks, or reads are not back to back (-> path G);
// bit 1 / bit 2 = some chunk's first byte is not 4- / 16-byte aligned (path B needs the latter; path A takes any base); a read longer than lmax -> status bit 8.
// `off` / `ends`: start and end byte of every read.  An Arrow Utf8 column passes (offsets, offsets + 1); a view over
// raw FASTQ text passes two separate arrays (reads are then not contiguous, which forces path G).
// blockIdx.y = chunk.
// CONTIG: every chunk is an Arrow column (ends == off + 1), so read r's length is off[r+1] - off[r] and "back to back" holds
// by construction: ONE 16-byte load per four reads (the fifth offset comes from the next lane), two in flight per thread.
template <bool CONTIG>
__global__ __launch_bounds__(256) void k5_scan_offsets(const K5Chunks ch, int lmax, int* __restrict__ flags, int* __restrict__ status) {
  const int32_t* __restrict__ off = ch.off[blockIdx.y];
  const int32_t* __restrict__ ends = ch.ends[blockIdx.y];
  const int64_t n = ch.n[blockIdx.y];
  const int L = ch.ends[0][0] - ch.off[0][0];  // every chunk is held to the first chunk's read length
  bool ragged = false, too_long = false;
  // four reads per thread per step: 16-byte loads (4-byte aligned: the hardware takes unaligned dwordx4)
  const int64_t n4 = n >> 2, S = (int64_t)gridDim.x * 256;
  auto ld4 = [](const int32_t* p) {
    int4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
  };
  if (CONTIG) {
    const int lane = threadIdx.x & 63;
    // whole waves stay in the loop together (the shuffle needs the neighbour): c is clamped, `live` masks the result
    const int64_t c0 = (int64_t)blockIdx.x * 256 + threadIdx.x, w0 = c0 - lane;
    for (int64_t w = w0; w < n4; w += 2 * S) {
      const int64_t ca = w + lane, cb = ca + S;
      const bool la = ca < n4, lb = cb < n4;
      const int4 a = la ? ld4(off + (ca << 2)) : int4{0, 0, 0, 0};
      const int4 b = lb ? ld4(off + (cb << 2)) : int4{0, 0, 0, 0};
      // the last live lane of a wave cannot trust the shuffle (its neighbour is dead): give it the direct load
      {
        int nx = __shfl_down(a.x, 1);
        const bool direct = la && (lane == 63 || ca + 1 >= n4);
        if (direct) nx = off[(ca << 2) + 4];
        const int l0 = a.y - a.x, l1 = a.z - a.y, l2 = a.w - a.z, l3 = nx - a.w;
        if (la) {
          ragged |= (l0 != L) | (l1 != L) | (l2 != L) | (l3 != L);
          too_long |= (l0 > lmax) | (l1 > lmax) | (l2 > lmax) | (l3 > lmax);
        }
      }
      {
        int nx = __shfl_down(b.x, 1);
        const bool direct = lb && (lane == 63 || cb + 1 >= n4);
        if (direct) nx = off[(cb << 2) + 4];
        const int l0 = b.y - b.x, l1 = b.z - b.y, l2 = b.w - b.z, l3 = nx - b.w;
        if (lb) {
          ragged |= (l0 != L) | (l1 != L) | (l2 != L) | (l3 != L);
          too_long |= (l0 > lmax) | (l1 > lmax) | (l2 > lmax) | (l3 > lmax);
        }
      }
    }
  } else {
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n4; c += S) {
      const int64_t i = c << 2;
      const int4 a = ld4(off + i), b = ld4(ends + i);
      const int l0 = b.x - a.x, l1 = b.y - a.y, l2 = b.z - a.z, l3 = b.w - a.w;
      ragged |= (l0 != L) | (l1 != L) | (l2 != L) | (l3 != L) | (a.y != b.x) | (a.z != b.y) | (a.w != b.z);
      if (i + 4 < n) ragged |= (off[i + 4] != b.w);
      too_long |= (l0 > lmax) | (l1 > lmax) | (l2 > lmax) | (l3 > lmax);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // the last n % 4 reads
    const int64_t i = (n4 << 2) + threadIdx.x;
    const int len = ends[i] - off[i];
    ragged |= (len != L) || (i + 1 < n && off[i + 1] != ends[i]);
    too_long |= (len > lmax);
  }
  if (__any(ragged) && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 1);
  if (__any(too_long) && (threadIdx.x & 63) == 0) atomicOr(status, 8);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(ch.bytes[blockIdx.y] + off[0]);
    const int f = ((base & 3) ? 2 : 0) | ((base & 15) ? 4 : 0);
    if (f) atomicOr(&flags[0], f);
  }
}

enum { K5_PATH_A = 0, K5_PATH_B = 2, K5_PATH_G = 3 };
// path A: period of the (dword -> positions) map in dwords, and the number of table copies that keep two lanes of one 32-lane
// half off the same address (see k5_main)
__host__ __device__ __forceinline__ int k5_period(int L) { return (L & 3) == 0 ? L >> 2 : (L & 1) == 0 ? L >> 1 : L; }
__host__ __device__ __forceinline__ int k5_copies(int P) { return P >= 32 ? 1 : P >= 16 ? 2 : 4; }
__device__ __forceinline__ int k5_pick_path(const K5Chunks& ch, int lmax, const int* flags) {
  const int f = flags[0];
  if (f & 1) return K5_PATH_G;
  const int L = ch.ends[0][0] - ch.off[0][0];
  if (L < 1 || L > lmax) return K5_PATH_G;
  if (k5_period(L) >= 8 && L <= K5_PT_MAX) return K5_PATH_A;  // any base alignment; slots: copies x 4 ceil(L / 4) <= 320
  if (L <= K5_PT_MAX && !(f & 4)) return K5_PATH_B;
  return K5_PATH_G;
}

// partial record of a workgroup: u32 [pt][128] (position-major, ASCII half; a launch gives a workgroup < 2^32 reads); bytes >= 128 never reach it
__device__ void k5_path_b(const K5Chunks& ch, int lmax, int pt, unsigned long long* __restrict__ partials,
                          unsigned long long* __restrict__ d_hist);
template <int LP>
__device__ void k5_path_ragged(const K5Chunks& ch, int lmax, int pt, unsigned long long* __restrict__ partials,
                               unsigned long long* __restrict__ d_hist);

// One launch for all chunks: path A in place, paths B / G through their functions (same workgroup shape, same LDS block).
// The LDS histogram is zeroed once, fed by every chunk, and flushed once into the workgroup's partial record.
template <int LP>
__global__ __launch_bounds__(K5_THREADS) void k5_main(const K5Chunks ch, int lmax, int pt, const int* __restrict__ flags,
                                                      unsigned long long* __restrict__ partials,
                                                      unsigned long long* __restrict__ d_hist) {
  const int path = k5_pick_path(ch, lmax, flags);
  if (path == K5_PATH_B) {
    k5_path_b(ch, lmax, pt, partials, d_hist);
    return;
  }
  if (path == K5_PATH_G) {
    k5_path_ragged<LP>(ch, lmax, pt, partials, d_hist);
    return;
  }
  // ---- path A: any uniform read length (round 3; was L % 4 == 0 only, and 101 / 150 / 151 / 250-byte reads ran path B at 2.2 TB/s) ----
  // The byte stream is read as dwords.  Dword D of a chunk holds the bytes of positions (4 D + k) mod L; that repeats with the
  // period P = L / gcd(L, 4) dwords.  A wave owns the 64-dword rows whose index is == r (mod P), so a lane keeps ONE
  // dword-of-period d -- four fixed positions p_k -- for the whole launch.  Position p lives in slot s(p) = (p & 3) C + (p >> 2),
  // C = ceil(L / 4): the lanes of a row step p by 4, i.e. s by 1, so their banks differ (s mod 64 = the column = the bank).  Lanes
  // that share d (P < 64) sit P lanes apart; when that is inside one 32-lane half (P < 32) they use `copies` = 2 (P >= 16) or 4
  // (P >= 8) copies of the table, chosen by the wrap count w, so a half never holds two equal addresses.  Table: 5 planes of
  // [128 bytes][64 columns] u32 = 160 KiB = 320 slots.  Bin address = plane << 15 | byte << 8 | column << 2: bits 16-17 and the
  // column come from the lane's constant c_k, bit 15 (the plane's low bit) is ORed into the masked data byte's bit 7 by the same
  // v_and_or_b32 that masks the dword, and ONE v_perm_b32 per byte drops the data byte into bits 8-15.
  constexpr int J = K5_JA;
  for (int i = threadIdx.x; i < K5_A_WORDS; i += K5_THREADS) k5_h[i] = 0;
  __syncthreads();
  const int L = ch.ends[0][0] - ch.off[0][0];
  const int P = k5_period(L), C = (L + 3) >> 2, copies = k5_copies(P);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the row bases below are wave-uniform
  const int NW = (int)gridDim.x * (K5_THREADS / 64), g = (int)blockIdx.x * (K5_THREADS / 64) + wave;
  // The wave's rows: index == r (mod Pr), every nslots-th of them.  A lane's dword-of-period in row R is (64 R + lane) mod P, which
  // depends on R mod Pr only, Pr = P / gcd(P, 64) (Pr = 1 when P divides 64: rows are then dealt one by one).  A wave's consecutive
  // rows must NOT sit a multiple of 4 KiB apart (16 rows: its 24 loads in flight then queue on the same HBM channel; L = 125 /
  // 249 / 251, where 4096 / P waves per residue is a power of two, ran at 4.5-5.1 TB/s instead of 6.1, L = 64 at 5.7): a slot is
  // given up until the stride is not (at most 6 % of the waves).
  const int Pr = P / min(P & -P, 64);
  int nslots = NW / Pr;
  if ((Pr & 15) != 0)
    while (nslots > 1 && ((nslots * Pr) & 15) == 0) --nslots;
  const int r = g % Pr, slot = g / Pr;
  const int t = (64 * r) % P + lane, w = t / P, d = t - w * P;
  char* hb = reinterpret_cast<char*>(k5_h);
  // byte address of the bin of (dword-of-period dd, wrap count ww, byte-in-dword k) without the byte value, and its position
  // Rows are 256-byte pieces of MEMORY (aligned loads whatever the chunk's first byte is): a chunk whose bytes start `sh` bytes
  // into its first row has dword D = 64 R + lane of row R holding the positions (4 D + k - sh) mod L.  shl = (-sh) mod L, per chunk.
  int shl = 0;
  auto pos_of = [&](int dd, int k) { return (4 * dd + k + shl) % L; };
  auto bin_base = [&](int dd, int ww, int k) {
    const int p = pos_of(dd, k), sl = (p & 3) * C + (p >> 2) + (ww & (copies - 1)) * 4 * C;
    return (unsigned)((sl >> 6) << 15 | (sl & 63) << 2);
  };
  unsigned ck[4], pb = 0;  // the lane's constants (per chunk: they follow shl): address bits 16-17 + column per k; the planes' low bits at bit 7 of byte k
  auto lane_constants = [&]() {
    pb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned a = bin_base(d, w, k);
      ck[k] = a & ~0x8000u;
      pb |= ((a >> 15) & 1u) << (8 * k + 7);
    }
  };
  auto slow = [&](unsigned dw, int dd, int ww) {  // a byte >= 128 somewhere, or the last, partial row (dd is not the lane's d)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b < 128) atomicAdd(reinterpret_cast<unsigned*>(hb + bin_base(dd, ww, k) + b * 256), 1u);
      else atomicAdd(&d_hist[(size_t)pos_of(dd, k) * 256 + b], 1ull);
    }
  };
  // The steady state has no branch per dword (tools/tune_k5.hip variant M: 6.52 -> 6.77 TB/s at L = 100, 6.44 -> 6.76 at
  // L = 148, 6.00 -> 6.42 at L = 64).  A dword is masked to 7 bits per byte before the v_perm, so a byte >= 128 lands in the bin
  // of byte & 127 for the moment; the unmasked dwords of J rows are ORed together and ONE wave-uniform test per J rows sends an
  // iteration that saw a high bit through `fix`: its rows are read again and every byte >= 128 is taken out of the bin it went
  // to (ds_sub) and added to the global histogram.  Quality strings never take it.  With the test per dword (v_and, v_cmp, six
  // scalar instructions and two branches between the load and its atomics) a wave's dependent chain per dword was longer than
  // the LDS pipe needs for it, and four waves per SIMD did not cover that.
  auto fast = [&](unsigned dw) {
    const unsigned m = (dw & 0x7F7F7F7Fu) | pb;  // v_and_or_b32
    // v_perm_b32: result byte 1 <- data byte k (bit 7 = the plane's low bit), bytes 0 / 2 / 3 <- the lane's constant
    const unsigned a0 = __builtin_amdgcn_perm(m, ck[0], 0x03020400u);
    const unsigned a1 = __builtin_amdgcn_perm(m, ck[1], 0x03020500u);
    const unsigned a2 = __builtin_amdgcn_perm(m, ck[2], 0x03020600u);
    const unsigned a3 = __builtin_amdgcn_perm(m, ck[3], 0x03020700u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a0), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a1), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a2), 1u);
    atomicAdd(reinterpret_cast<unsigned*>(hb + a3), 1u);
  };
  auto one = [&](unsigned dw) {  // the rows outside the steady state
    if (__builtin_expect((dw & 0x80808080u) != 0, 0)) slow(dw, d, w);
    else fast(dw);
  };
  auto fix = [&](unsigned dw) {  // dw went through fast(): move its bytes >= 128 to where they belong
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned b = (dw >> (8 * k)) & 0xFF;
      if (b >= 128) {
        atomicSub(reinterpret_cast<unsigned*>(hb + bin_base(d, w, k) + (b & 127) * 256), 1u);
        atomicAdd(&d_hist[(size_t)pos_of(d, k) * 256 + b], 1ull);
      }
    }
  };
  const int64_t stride = (int64_t)nslots * Pr, first = (int64_t)slot * Pr + r;  // this wave's rows: first + i * stride
  for (int ci = 0; ci < ch.count; ++ci) {  // the wave keeps its rows, d and w; the positions behind them shift with the chunk's base
    const uint8_t* base = ch.bytes[ci] + ch.off[ci][0];
    const int sh = (int)(reinterpret_cast<uintptr_t>(base) & 255);
    const unsigned* src = reinterpret_cast<const unsigned*>(base - sh);  // 256-byte aligned
    const int64_t nbytes = ch.n[ci] * (int64_t)L, span = sh + nbytes;    // the chunk's bytes are [sh, span) of the row grid
    const int64_t R0 = sh ? 1 : 0, R1 = span >> 8;                       // rows [R0, R1) lie entirely inside the chunk
    shl = (L - sh % L) % L;
    lane_constants();
    if (slot < nslots) {
      const int64_t firstc = first < R0 ? first + stride : first;
      const int64_t nmine = firstc < R1 ? (R1 - firstc + stride - 1) / stride : 0;
      const unsigned* p = src + firstc * 64;  // wave-uniform
      const int64_t pstep = stride * 64;
      int64_t i = 0;
      auto fixup = [&](unsigned acc) {
        if (__builtin_expect(__any((acc & 0x80808080u) != 0), 0)) {
#pragma unroll 1
          for (int j = 0; j < J; ++j) fix(p[j * pstep + lane]);
        }
      };
      if (nmine >= J) {
        // A ROLLING window of J loads: a register is refilled with the row J ahead as soon as its dword is consumed
        // (s_waitcnt vmcnt(J - 1) throughout).  The scheduling barriers keep the compiler from gathering the J masked copies
        // first (= a wait for every load) and issuing the loads in one burst.
        unsigned v[J];
        const unsigned* ld = p + lane;  // the next row to load: loads go out in row order, so ONE running pointer serves them
#pragma unroll                          // (J scalar row offsets would not fit the scalar registers next to the chunk table)
        for (int j = 0; j < J; ++j) {
          v[j] = __builtin_nontemporal_load(ld);
          ld += pstep;
          __builtin_amdgcn_sched_barrier(0);  // in row order, as the loop issues them: the loop's wait counts hold from the first pass
        }
        for (i = J; i + J <= nmine; i += J) {
          unsigned acc = 0;
#pragma unroll
          for (int j = 0; j < J; ++j) {
            // as asm: the compiler would merge two rows' ORs into a v_or3 placed after the first row's reload, which costs
            // a new register per load and a rotation of register copies at the loop head (and the wait that goes with it)
            asm volatile("v_or_b32 %0, %0, %1" : "+v"(acc) : "v"(v[j]));
            fast(v[j]);
            v[j] = __builtin_nontemporal_load(ld);
            ld += pstep;
            __builtin_amdgcn_sched_barrier(0);
          }
          fixup(acc);
          p += J * pstep;
        }
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          acc |= v[j];
          fast(v[j]);
        }
        fixup(acc);
        p += J * pstep;
      }
      if (i < nmine) {
        // the last < J rows of this wave, all loads in flight at once (one at a time they cost a memory round trip each:
        // 40 us of a 345 us launch at 20 M reads).  A row's validity is wave-uniform; invalid slots re-read the first row.
        unsigned v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) v[j] = __builtin_nontemporal_load(p + (i + j < nmine ? j : 0) * pstep + lane);
#pragma unroll
        for (int j = 0; j < J; ++j)
          if (i + j < nmine) one(v[j]);
      }
    }
    if (g == NW - 1 && nbytes > 0) {  // the rows the chunk covers only in part: its first (sh > 0) and its last, byte by byte
      auto edge_row = [&](int64_t R) {
        const int64_t e0 = (R * 64 + lane) * 4;  // byte offset of this lane's dword in the row grid
        if (e0 + 4 <= sh || e0 >= span) return;
        const unsigned dw = src[R * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t e = e0 + k;
          if (e < sh || e >= span) continue;
          const unsigned b = (dw >> (8 * k)) & 0xFF;
          const int pe = (int)((e - sh) % L), sl = (pe & 3) * C + (pe >> 2);
          if (b < 128) atomicAdd(reinterpret_cast<unsigned*>(hb + (unsigned)((sl >> 6) << 15 | (sl & 63) << 2) + b * 256), 1u);
          else atomicAdd(&d_hist[(size_t)pe * 256 + b], 1ull);
        }
      };
      if (R0) edge_row(0);
      if ((span & 255) != 0 && !(R0 && R1 == 0)) edge_row(R1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < pt * 128; i += K5_THREADS) {
    const int p = i >> 7, b = i & 127;
    unsigned v = 0;
    if (p < L) {
      const int sl = (p & 3) * C + (p >> 2);
      for (int c = 0; c < copies; ++c) {
        const int sc = sl + c * 4 * C;
        v += k5_h[(sc >> 6) * 8192 + b * 64 + (sc & 63)];
      }
    }
    reinterpret_cast<unsigned*>(partials)[(size_t)blockIdx.x * pt * 128 + i] = v;
  }
}

__device__ void k5_path_b(const K5Chunks& ch, int lmax, int pt, unsigned long long* __restrict__ partials,
                          unsigned long long* __restrict__ d_hist) {
  // k5_h as [pt][129]
  for (int i = threadIdx.x; i < pt * 129; i += K5_THREADS) k5_h[i] = 0;
  __syncthreads();
  auto add = [&](int p, unsigned b) {
    if (p < pt && b < 128) atomicAdd(&k5_h[p * 129 + b], 1u);
    else atomicAdd(&d_hist[(size_t)p * 256 + b], 1ull);
  };
  const int L = ch.ends[0][0] - ch.off[0][0];
  for (int ci = 0; ci < ch.count; ++ci) {  // uniform read length: 16-byte chunk per lane
    constexpr int J = K5_JB;
    const uint8_t* src = ch.bytes[ci] + ch.off[ci][0];
    const int64_t total = ch.n[ci] * (int64_t)L, nch = total / 16, S = (int64_t)gridDim.x * K5_THREADS;
    const int64_t c0 = (int64_t)blockIdx.x * K5_THREADS + threadIdx.x;
    int p0 = (int)((16 * c0) % L);
    const int pS = (int)((16 * S) % L);
    auto chunk = [&](v4i_t v, int p) {
      const unsigned d[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        int pk = p + k;
        while (pk >= L) pk -= L;  // L may be < 16
        add(pk, (d[k >> 2] >> (8 * (k & 3))) & 0xFF);
      }
    };
    int64_t c = c0;
    for (; c + (J - 1) * S < nch; c += J * S) {
      v4i_t v[J];
      int pj[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(src + 16 * (c + j * S)));
        pj[j] = p0;
        p0 += pS;
        p0 = p0 >= L ? p0 - L : p0;
      }
#pragma unroll
      for (int j = 0; j < J; ++j) chunk(v[j], pj[j]);
    }
    for (; c < nch; c += S) {
      chunk(*reinterpret_cast<const v4i_t*>(src + 16 * c), p0);
      p0 += pS;
      p0 = p0 >= L ? p0 - L : p0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)  // < 16 trailing bytes
      for (int64_t e = nch * 16; e < total; ++e) add((int)(e % L), src[e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < pt * 128; i += K5_THREADS)
    reinterpret_cast<unsigned*>(partials)[(size_t)blockIdx.x * pt * 128 + i] = k5_h[(i >> 7) * 129 + (i & 127)];
}

// Path G: ragged reads.  A half-wave (32 lanes) owns a read; lane l loads the (possibly unaligned) dword holding
// positions 4l..4l+3, so positions are known without any search, a wave instruction still covers two ~100-byte
// reads (like path A's dword loads), and J reads are in flight per half-wave.  With lmax <= LP (BM) the histogram
// uses path A's conflict-free byte-major layout h[byte][perm(p)] (the two half-waves of a wave are separate LDS
// lane groups, so their equal positions never conflict) and the inner loop is branch-free: bytes past the end of
// a read add to a per-lane dummy word, every load is unconditional.  Longer reads fall back to h[p][129].
template <int LP, bool BM>
__device__ void k5_path_ragged_impl(const K5Chunks& ch, int lmax, int pt, unsigned long long* __restrict__ partials,
                                    unsigned long long* __restrict__ d_hist) {
  constexpr int Q = LP / 4, J = 8;
  const int nwords = BM ? 128 * LP + 64 : pt * 129;
  for (int i = threadIdx.x; i < nwords; i += K5_THREADS) k5_h[i] = 0;
  __syncthreads();
  unsigned* const dummy = k5_h + 128 * LP + (threadIdx.x & 63);  // BM only
  auto add_slow = [&](int p, unsigned b) {
    if (b < 128 && (BM || p < pt)) atomicAdd(&k5_h[BM ? (int)b * LP + (p & 3) * Q + (p >> 2) : p * 129 + (int)b], 1u);
    else atomicAdd(&d_hist[(size_t)p * 256 + b], 1ull);
  };
  auto load_u32 = [&](const uint8_t* p) {
    unsigned v;
    __builtin_memcpy(&v, p, 4);  // unaligned dword load
    return v;
  };
  const int hl = threadIdx.x & 31;
  const int64_t ghw = (int64_t)blockIdx.x * (K5_THREADS / 32) + (threadIdx.x >> 5);
  const int64_t nhw = (int64_t)gridDim.x * (K5_THREADS / 32);
  const int q = hl * 4;
  constexpr bool TWO = LP > 128;  // reads can reach past position 128: fetch a second dword per lane
  for (int ci = 0; ci < ch.count; ++ci) {
  const int32_t* __restrict__ off = ch.off[ci];
  const int32_t* __restrict__ ends = ch.ends[ci];
  const uint8_t* __restrict__ bytes = ch.bytes[ci];
  const int64_t n = ch.n[ci];
  // The dword of positions [q, q+4) of a read of `len` >= 4 bytes at byte offset o0.  A partial last dword is
  // fetched as the dword that ENDS at the read's end (never touches bytes past the read) and shifted down.  The load
  // is UNCONDITIONAL (inactive lanes read the first dword of the buffer): a load inside a branch makes the compiler
  // drain all loads in flight (s_waitcnt vmcnt(0)) at every such branch.  nb = valid bytes (0..4) in the low bytes.
  auto fetch = [&](int64_t o0, int len, int q, int* nb) -> unsigned {
    const int rem = len - q;
    const bool active = rem > 0 && len >= 4;
    *nb = active ? (rem > 4 ? 4 : rem) : 0;
    const int64_t a = active ? o0 + (rem >= 4 ? q : len - 4) : 0;
    const unsigned d = load_u32(bytes + a);
    return (active && rem < 4) ? d >> (8 * (4 - rem)) : d;
  };
  auto emit = [&](unsigned d, int nb, int q) {
    if (BM) {
      if (__builtin_expect((d & 0x80808080u) != 0 && nb > 0, 0)) {  // non-ASCII byte somewhere: slow path
        for (int k = 0; k < nb; ++k) add_slow(q + k, (d >> (8 * k)) & 0xFF);
        return;
      }
      unsigned* base = k5_h + (q >> 2);  // q is a multiple of 4: perm(q + k) = k * Q + (q >> 2)
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(k < nb ? base + ((d >> (8 * k)) & 0x7F) * LP + k * Q : dummy, 1u);
    } else {
      for (int k = 0; k < nb; ++k) add_slow(q + k, (d >> (8 * k)) & 0xFF);
    }
  };
  for (int64_t r0 = ghw; r0 < n; r0 += nhw * J) {
    int64_t o0[J];
    int len[J], nb0[J], nb1[J];
    unsigned d0[J], d1[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {  // offsets: unconditional loads (index clamped), length zeroed past the end
      const int64_t rr = r0 + (int64_t)j * nhw;
      const int64_t rc = rr < n ? rr : n - 1;
      const int32_t a = off[rc], b = ends[rc];
      o0[j] = a;
      len[j] = rr < n ? min(b - a, lmax) : 0;  // reads longer than lmax are flagged by k5_scan_offsets
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {  // all data loads of the iteration are issued before any use
      d0[j] = fetch(o0[j], len[j], q, &nb0[j]);
      if (TWO) d1[j] = fetch(o0[j], len[j], q + 128, &nb1[j]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      emit(d0[j], nb0[j], q);
      if (TWO) emit(d1[j], nb1[j], q + 128);
      for (int q2 = q + (TWO ? 256 : 128); q2 < len[j]; q2 += 128) {  // longer reads (only when !BM)
        int nb;
        const unsigned d = fetch(o0[j], len[j], q2, &nb);
        emit(d, nb, q2);
      }
      if (len[j] > 0 && len[j] < 4 && hl == 0)  // reads shorter than a dword: bytewise
        for (int k = 0; k < len[j]; ++k) add_slow(k, bytes[o0[j] + k]);
    }
  }
  }  // chunks
  __syncthreads();
  for (int i = threadIdx.x; i < pt * 128; i += K5_THREADS) {
    const int p = i >> 7, b = i & 127;
    reinterpret_cast<unsigned*>(partials)[(size_t)blockIdx.x * pt * 128 + i] = BM ? k5_h[b * LP + (p & 3) * Q + (p >> 2)] : k5_h[p * 129 + b];
  }
}

template <int LP>
__device__ void k5_path_ragged(const K5Chunks& ch, int lmax, int pt, unsigned long long* __restrict__ partials,
                               unsigned long long* __restrict__ d_hist) {
  if (lmax <= LP) k5_path_ragged_impl<LP, true>(ch, lmax, pt, partials, d_hist);
  else k5_path_ragged_impl<LP, false>(ch, lmax, pt, partials, d_hist);
}

// d_hist[p][b] += sum over workgroups of partial[wg][p][b]  (p < pt, b < 128).  blockIdx.y splits the workgroup
// range 8 ways (integer adds commute, so the segment sums are merged with atomics: still bit-exact).
__global__ __launch_bounds__(256) void k5_finalize(const unsigned* __restrict__ partials, int nblocks, int pt,
                                                   unsigned long long* __restrict__ d_hist, int* __restrict__ flags) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) flags[0] = 0;  // ready for the next batch's offsets scan
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= pt * 128) return;
  const size_t W = (size_t)pt * 128;
  const int per = (nblocks + (int)gridDim.y - 1) / (int)gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  unsigned long long acc = 0;
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
    unsigned t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = partials[(size_t)(b + k) * W + i];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += t[k];
  }
  for (; b < b1; ++b) acc += partials[(size_t)b * W + i];
  if (acc) atomicAdd(&d_hist[(size_t)(i >> 7) * 256 + (i & 127)], acc);
}

static int k5_pt(int lmax) { return lmax < K5_PT_MAX ? lmax : K5_PT_MAX; }
size_t k5_partial_words(const LaunchCfg& cfg, int lmax) { return ((size_t)cfg.compute_units * k5_pt(lmax) * 128 + 1) / 2; }  // u32 records

hipError_t launch_qual_pos_hist(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* offsets,
                                const uint8_t* bytes, int64_t n_reads, int lmax, int64_t* d_hist) {
  return launch_qual_pos_hist_views(s, cfg, ws, offsets, offsets + 1, bytes, n_reads, lmax, d_hist);
}

hipError_t launch_qual_pos_hist_views(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const int32_t* offsets,
                                      const int32_t* ends, const uint8_t* bytes, int64_t n_reads, int lmax,
                                      int64_t* d_hist) {
  if (n_reads <= 0) return hipSuccess;
  K5Chunks ch;
  ch.off[0] = offsets;
  ch.ends[0] = ends;
  ch.bytes[0] = bytes;
  ch.n[0] = n_reads;
  ch.count = 1;
  return launch_qual_pos_hist_chunks(s, cfg, ws, ch, lmax, d_hist);
}

hipError_t launch_qual_pos_hist_chunks(hipStream_t s, const LaunchCfg& cfg, const Workspace& ws, const K5Chunks& ch, int lmax,
                                       int64_t* d_hist) {
  if (ch.count < 1 || ch.count > K5_MAX_CHUNKS || lmax < 1) return hipErrorInvalidValue;
  int64_t n_max = 0;
  for (int c = 0; c < ch.count; ++c) {
    if (ch.n[c] <= 0) return hipErrorInvalidValue;
    n_max = std::max(n_max, ch.n[c]);
  }
  const int pt = k5_pt(lmax);
  const int grid = cfg.compute_units;
  int* flags = ws.status + 1;
  unsigned long long* hist = reinterpret_cast<unsigned long long*>(d_hist);
  // flags[0] is zero here: the workspace starts zeroed and k5_finalize, the last kernel of every launch, resets it
  // (a memset per launch would be a fourth kernel; a launch that fails midway can leave it set, which only costs speed:
  // the ragged path is correct for uniform reads too)
  hipError_t e;
  if (cfg.overwrite && (e = hipMemsetAsync(d_hist, 0, (size_t)lmax * 256 * 8, s)) != hipSuccess) return e;  // k5_finalize adds
  int sgrid = (int)std::min<int64_t>((n_max / 4 + 255) / 256 + 1, (int64_t)cfg.compute_units * 8);
  if (ch.count > 1) sgrid = std::max(1, std::min(sgrid, cfg.compute_units * 16 / ch.count));
  bool contig = true;
  for (int c = 0; c < ch.count; ++c) contig &= (ch.ends[c] == ch.off[c] + 1);
  if (contig) hipLaunchKernelGGL(k5_scan_offsets<true>, dim3(sgrid, ch.count), dim3(256), 0, s, ch, lmax, flags, ws.status);
  else hipLaunchKernelGGL(k5_scan_offsets<false>, dim3(sgrid, ch.count), dim3(256), 0, s, ch, lmax, flags, ws.status);
  const bool lp256 = lmax > 128;  // LDS: the static k5_h block holds every layout ([128][LP] + dummies, [pt][129], path A's table)
  if (lp256) hipLaunchKernelGGL(k5_main<256>, dim3(grid), dim3(K5_THREADS), 0, s, ch, lmax, pt, flags, ws.partials, hist);
  else hipLaunchKernelGGL(k5_main<128>, dim3(grid), dim3(K5_THREADS), 0, s, ch, lmax, pt, flags, ws.partials, hist);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k5_finalize, dim3((pt * 128 + 255) / 256, 8), dim3(256), 0, s, reinterpret_cast<const unsigned*>(ws.partials), grid, pt, hist, flags);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
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

// C6 (seed 6): alignments for the interval-overlap count.  reference id over 25 references ~ length (same thresholds as
// C3), NULL for 2 % of the rows together with start/end (an unmapped read has neither); start uniform in [1, 249e6],
// end = start + (rnd % 20000).
__global__ __launch_bounds__(256) void gen_c6_kernel(uint64_t seed, int64_t lo, int64_t hi, C3Table t, int32_t* __restrict__ ref,
                                                     uint8_t* __restrict__ rvalid, int64_t* __restrict__ start,
                                                     int64_t* __restrict__ end, uint8_t* __restrict__ pvalid) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = hi - lo;
  const int lane = threadIdx.x & 63;
  bool mapped = false;
  if (k < n) {
    const uint64_t i = (uint64_t)(lo + k);
    const uint64_t r0 = rnd(seed, 0, i), r1 = rnd(seed, 1, i);
    const uint32_t u0 = (uint32_t)(r0 >> 32);
    int rc = 0;
#pragma unroll
    for (int c = 0; c < 24; ++c) rc += (u0 >= t.rthr[c]);
    mapped = ((uint32_t)r0 % 50u) != 0u;
    ref[k] = mapped ? rc : -1;
    const int64_t st = 1 + (int64_t)(r1 % 249000000ull);
    start[k] = mapped ? st : 0;
    end[k] = mapped ? st + (int64_t)((r1 >> 40) % 20000ull) : 0;
  }
  const int64_t wave_row0 = k - lane;
  store_valid64(rvalid, wave_row0, n, mapped, lane);
  store_valid64(pvalid, wave_row0, n, mapped, lane);
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
// One dispatch holds fewer than 2^32 work-items (a larger grid is cut off without an error: 4.6e9 rows came back as the
// 3e8 rows of the remainder, found by tests/test_gpu_fullsize.py's > 2^32-row case).  The one-thread-per-row generators
// therefore write tables in pieces of 2^30 rows; a piece starts on a multiple of 64 rows, so its validity bits start on a
// byte and the kernels' wave-wide validity stores stay aligned.
constexpr int64_t GEN_PIECE = int64_t(1) << 30;

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
  for (int64_t o = 0; o < hi - lo; o += GEN_PIECE) {
    const int64_t m = std::min(GEN_PIECE, hi - lo - o);
    hipLaunchKernelGGL(gen_c2_kernel, dim3(blocks_for(m)), dim3(256), 0, s, seed, lo + o, lo + o + m, t, chrom + o, pos + o);
  }
  return hipGetLastError();
}

static C3Table make_c3_table() {
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
  return t;
}

hipError_t launch_gen_c3(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, int32_t* flag, uint8_t* mapq,
                         uint8_t* mapq_valid, int32_t* ref_id, uint8_t* ref_valid) {
  if (hi <= lo) return hipSuccess;
  const C3Table t = make_c3_table();
  for (int64_t o = 0; o < hi - lo; o += GEN_PIECE) {
    const int64_t m = std::min(GEN_PIECE, hi - lo - o);
    hipLaunchKernelGGL(gen_c3_kernel, dim3(blocks_for(m)), dim3(256), 0, s, seed, lo + o, lo + o + m, t, flag + o, mapq + o,
                       mapq_valid + o / 8, ref_id + o, ref_valid + o / 8);
  }
  return hipGetLastError();
}

hipError_t launch_gen_c6(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, int32_t* ref_id, uint8_t* ref_valid,
                         int64_t* start, int64_t* end, uint8_t* pos_valid) {
  if (hi <= lo) return hipSuccess;
  for (int64_t o = 0; o < hi - lo; o += GEN_PIECE) {
    const int64_t m = std::min(GEN_PIECE, hi - lo - o);
    hipLaunchKernelGGL(gen_c6_kernel, dim3(blocks_for(m)), dim3(256), 0, s, seed, lo + o, lo + o + m, make_c3_table(), ref_id + o,
                       ref_valid + o / 8, start + o, end + o, pos_valid + o / 8);
  }
  return hipGetLastError();
}

hipError_t launch_gen_c4(hipStream_t s, uint64_t seed, int64_t lo, int64_t hi, float* af, uint8_t* af_valid,
                         float* qual, uint8_t* qual_valid, int32_t* filter_id) {
  if (hi <= lo) return hipSuccess;
  for (int64_t o = 0; o < hi - lo; o += GEN_PIECE) {
    const int64_t m = std::min(GEN_PIECE, hi - lo - o);
    hipLaunchKernelGGL(gen_c4_kernel, dim3(blocks_for(m)), dim3(256), 0, s, seed, lo + o, lo + o + m, pct_thr(85), pct_thr(90),
                       pct_thr(96), pct_thr(99), af + o, af_valid + o / 8, qual + o, qual_valid + o / 8, filter_id + o);
  }
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
