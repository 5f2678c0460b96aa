// How many single-wave workgroups does a gfx950 CU really hold, by VGPR count / SGPR count / scratch use?  Every workgroup spins
// for a fixed number of clocks; a launch of CUs x W workgroups takes one spin when all are resident and two when they are not.
// (Round 4: the serial inflate kernel -- 70 VGPRs, 106 SGPRs, 16-48 B of scratch -- holds exactly 24 per CU although
// hipOccupancyMaxActiveBlocksPerMultiprocessor says 27: which resource is it?)
//   hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o tools/bin/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void spin(unsigned long long clocks) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < clocks) __builtin_amdgcn_s_sleep(8);
}
template <int V>  // V = highest VGPR touched
__global__ __launch_bounds__(64) void k_vgpr(unsigned long long clocks, unsigned* out) {
  if (V == 63) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (V == 64) asm volatile("v_mov_b32 v64, 0" ::: "v64");
  if (V == 69) asm volatile("v_mov_b32 v69, 0" ::: "v69");
  if (V == 71) asm volatile("v_mov_b32 v71, 0" ::: "v71");
  if (V == 72) asm volatile("v_mov_b32 v72, 0" ::: "v72");
  if (V == 79) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  spin(clocks);
  if (clocks == 1) out[0] = 1;
}
__global__ __launch_bounds__(64) void k_sgpr(unsigned long long clocks, unsigned* out) {  // 63 VGPRs, s101 touched
  asm volatile("v_mov_b32 v63, 0\n s_mov_b32 s101, 0" ::: "v63", "s101");
  spin(clocks);
  if (clocks == 1) out[0] = 1;
}
template <int S>  // 63 VGPRs, highest SGPR touched = S
__global__ __launch_bounds__(64) void k_sgpr_n(unsigned long long clocks, unsigned* out) {
  asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (S == 63) asm volatile("s_mov_b32 s63, 0" ::: "s63");
  if (S == 71) asm volatile("s_mov_b32 s71, 0" ::: "s71");
  if (S == 79) asm volatile("s_mov_b32 s79, 0" ::: "s79");
  if (S == 87) asm volatile("s_mov_b32 s87, 0" ::: "s87");
  if (S == 89) asm volatile("s_mov_b32 s89, 0" ::: "s89");
  if (S == 95) asm volatile("s_mov_b32 s95, 0" ::: "s95");
  if (S == 97) asm volatile("s_mov_b32 s97, 0" ::: "s97");
  spin(clocks);
  if (clocks == 1) out[0] = 1;
}
__global__ __launch_bounds__(64) void k_scratch(unsigned long long clocks, unsigned* out, int idx) {  // 63 VGPRs + a private array
  asm volatile("v_mov_b32 v63, 0" ::: "v63");
  volatile unsigned priv[8];
  for (int i = 0; i < 8; ++i) priv[i] = i + idx;
  spin(clocks);
  if (clocks == 1) out[0] = priv[idx & 7];
}

template <class F>
static void sweep(const char* name, F launch, int cus) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-44s", name);
  for (int W : {24, 25, 26, 27, 28, 29, 32, 33}) {
    launch(cus * W, 1000ull);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(cus * W, 200000ull);  // 2 ms at 100 MHz counter ... whatever the unit, one spin vs two
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  W=%d: %.2f", W, ms);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned* out;
  CHECK(hipMalloc(&out, 64));
  sweep("v63 (64 VGPRs)", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_vgpr<63>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("v64 (65 VGPRs)", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_vgpr<64>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("v69 (70 VGPRs)", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_vgpr<69>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("v71 (72 VGPRs)", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_vgpr<71>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("v72 (73 VGPRs)", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_vgpr<72>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("v79 (80 VGPRs)", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_vgpr<79>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s101 (102 SGPRs)", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s63", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr_n<63>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s71", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr_n<71>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s79", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr_n<79>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s87", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr_n<87>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s89", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr_n<89>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s95", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr_n<95>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + s97", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_sgpr_n<97>, dim3(g), dim3(64), 0, 0, c, out); }, cus);
  sweep("64 VGPRs + 32 B of scratch", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_scratch, dim3(g), dim3(64), 0, 0, c, out, 3); }, cus);
  sweep("64 VGPRs + 6 KB of LDS", [&](int g, unsigned long long c) { hipLaunchKernelGGL(k_vgpr<63>, dim3(g), dim3(64), 5888, 0, c, out); }, cus);
  return 0;
}
