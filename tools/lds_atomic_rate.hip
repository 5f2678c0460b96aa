// lds_atomic_rate.hip -- developer harness (NOT part of the product): what one CU's LDS pipe sustains for returnless
// ds_add_u32 wave-instructions under the access patterns K5 can produce.  Reports LDS-pipe cycles per wave-instruction
// (all 16 waves of a 1024-thread workgroup issuing back to back, one workgroup per CU).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_atomic_rate.hip -o tools/bin/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// MODE 0: address = lane (32 banks x 2 halves: conflict-free, all distinct)
// MODE 1: lanes l and l+25 share a bank, different address (K5 path A at L = 100, different bytes)
// MODE 2: lanes l and l+25 share the ADDRESS (same byte at the same column)
// MODE 3: random address per lane per op (LCG), 16 K words
// MODE 4: address = lane, 64-bit add (ds_add_u64)
// MODE 5: plain ds_write_b32 to address = lane (no atomic)
// MODE 6: address = lane, atomic WITH return (ds_add_rtn_u32)
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rate(int iters, unsigned* sink, unsigned long long* cycles) {
  __shared__ unsigned h[32768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768; i += WAVES * 64) h[i] = 0;
  __syncthreads();
  unsigned a0;
  if (MODE == 1) a0 = (lane % 25) + 32 * (lane / 25) * 7;          // same bank (mod 32), different row
  else if (MODE == 2) a0 = lane % 25;                               // same address
  else if (MODE == 7) a0 = lane & 31;                               // lanes l and l+32 share the address (cross-half only)
  else if (MODE == 8) a0 = (lane & 15) + 32 * (lane >> 5);          // lanes l and l+16 share the address (same half)
  else if (MODE == 9) a0 = lane == 63 ? 0 : lane;                   // ONE colliding pair, across halves (0, 63)
  else if (MODE == 10) a0 = lane == 31 ? 0 : lane;                  // ONE colliding pair inside a half (0, 31)
  else if (MODE == 11) a0 = lane == 1 ? 0 : lane;                   // ONE colliding pair, adjacent lanes (0, 1)
  else a0 = lane;
  a0 += wave * 1024;
  unsigned x = threadIdx.x * 2654435761u + 12345u, acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      unsigned a = a0 + k * 64;
      if (MODE == 3) { x = x * 1664525u + 1013904223u; a = (x >> 8) & 16383; }
      if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(h) + a, 1ull);
      else if (MODE == 5) __builtin_nontemporal_store(i + k, h + a);
      else if (MODE == 6) acc += atomicAdd(h + a, 1u);
      else __hip_atomic_fetch_add(h + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  unsigned s = acc;
  for (int i = threadIdx.x; i < 32768; i += WAVES * 64) s += h[i];
  if (s == 0xdeadbeef) sink[0] = s;
}

template <int MODE, int WAVES>
static void run(const char* name, int grid) {
  unsigned* sink; unsigned long long* cyc; CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, grid * 8));
  const int iters = 4096;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rate<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, 16, sink, cyc);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((rate<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, iters, sink, cyc);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  const double wave_ops_per_cu = (double)iters * 8 * WAVES;
  const double cycles = ms * 1e-3 * clk_khz * 1e3;
  printf("%-44s waves/CU %2d: %.2f clk per wave-instruction (%.3f ms, %.1f lane-ops/clk/CU)\n", name, WAVES, cycles / wave_ops_per_cu, ms,
         64.0 * wave_ops_per_cu / cycles);
  CK(hipFree(sink)); CK(hipFree(cyc));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int grid = p.multiProcessorCount;
  printf("%s, %d CUs, %d kHz\n", p.name, grid, p.clockRate);
  run<0, 16>("ds_add_u32 addr = lane (conflict-free)", grid);
  run<0, 8>("ds_add_u32 addr = lane (conflict-free)", grid);
  run<0, 4>("ds_add_u32 addr = lane (conflict-free)", grid);
  run<1, 16>("ds_add_u32 7 of 32 lanes share a bank", grid);
  run<2, 16>("ds_add_u32 lanes 25 apart share the address", grid);
  run<3, 16>("ds_add_u32 random address in 16 K words", grid);
  run<4, 16>("ds_add_u64 addr = lane", grid);
  run<5, 16>("ds_write_b32 addr = lane", grid);
  run<6, 16>("ds_add_rtn_u32 addr = lane", grid);
  run<7, 16>("ds_add_u32 lanes l, l+32 share the address", grid);
  run<8, 16>("ds_add_u32 lanes l, l+16 share the address", grid);
  run<9, 16>("ds_add_u32 one colliding pair (0, 63)", grid);
  run<10, 16>("ds_add_u32 one colliding pair (0, 31)", grid);
  run<11, 16>("ds_add_u32 one colliding pair (0, 1)", grid);
  return 0;
}
