// Issue rates a serial (wave-uniform) decoder can count on, measured: dependent chains of scalar-ALU or vector-ALU
// instructions, W single-wave workgroups per CU.  Question behind it (round 4, inflate): the symbol loop of inflate.hip is
// bound by the CU's ONE scalar unit; would the same uniform arithmetic issue faster as VALU instructions (4 SIMD-32 units per
// CU, a wave64 instruction every 2 clocks each), and do scalar-flavoured and vector-flavoured waves overlap on one CU?
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

extern __shared__ unsigned char dyn_lds[];

// mode 0: scalar chain; 1: vector chain (uniform values); 2: blockIdx parity picks; 3: per-iteration mix 12 S + 5 V (the
// literal loop's mix); 4: scalar chain with a taken branch per 6 instructions; 5: vector chain with v_cmp + s_cbranch_vccnz
__global__ __launch_bounds__(64) void k_chain(int mode, int iters, unsigned* out) {
  unsigned s = (unsigned)iters, v = threadIdx.x;
  int m = mode == 2 ? (int)((blockIdx.x >> 8) & 1) : mode;
  if (m == 0) {
    asm volatile(
        "s_mov_b32 s40, %[it]\n s_mov_b32 s41, 1\n s_mov_b32 s42, 3\n"
        "L_s%=:\n"
        "s_add_u32 s41, s41, s42\n s_xor_b32 s41, s41, s42\n s_add_u32 s41, s41, s42\n s_lshr_b32 s43, s41, 3\n"
        "s_add_u32 s41, s41, s43\n s_xor_b32 s41, s41, s42\n s_add_u32 s41, s41, s42\n s_lshr_b32 s43, s41, 3\n"
        "s_add_u32 s41, s41, s43\n s_xor_b32 s41, s41, s42\n s_add_u32 s41, s41, s42\n s_lshr_b32 s43, s41, 3\n"
        "s_add_u32 s41, s41, s43\n s_xor_b32 s41, s41, s42\n s_add_u32 s41, s41, s42\n s_and_b32 s41, s41, 0xffff\n"
        "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_s%=\n"
        "v_mov_b32 %[v], s41\n"
        : [v] "=v"(v) : [it] "s"(s) : "s40", "s41", "s42", "s43", "scc");
  } else if (m == 1) {
    asm volatile(
        "s_mov_b32 s40, %[it]\n v_mov_b32 v41, 1\n v_mov_b32 v42, 3\n"
        "L_v%=:\n"
        "v_add_u32 v41, v41, v42\n v_xor_b32 v41, v41, v42\n v_add_u32 v41, v41, v42\n v_lshrrev_b32 v43, 3, v41\n"
        "v_add_u32 v41, v41, v43\n v_xor_b32 v41, v41, v42\n v_add_u32 v41, v41, v42\n v_lshrrev_b32 v43, 3, v41\n"
        "v_add_u32 v41, v41, v43\n v_xor_b32 v41, v41, v42\n v_add_u32 v41, v41, v42\n v_lshrrev_b32 v43, 3, v41\n"
        "v_add_u32 v41, v41, v43\n v_xor_b32 v41, v41, v42\n v_add_u32 v41, v41, v42\n v_and_b32 v41, 0xffff, v41\n"
        "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_v%=\n"
        "v_mov_b32 %[v], v41\n"
        : [v] "=v"(v) : [it] "s"(s) : "s40", "v41", "v42", "v43", "scc");
  } else if (m == 3) {
    asm volatile(
        "s_mov_b32 s40, %[it]\n s_mov_b32 s41, 1\n s_mov_b32 s42, 3\n v_mov_b32 v41, 1\n v_mov_b32 v42, 3\n"
        "L_m%=:\n"
        "s_add_u32 s41, s41, s42\n s_xor_b32 s41, s41, s42\n v_add_u32 v41, s41, v42\n s_add_u32 s41, s41, s42\n s_lshr_b32 s43, s41, 3\n"
        "v_xor_b32 v41, v41, v42\n s_add_u32 s41, s41, s43\n s_xor_b32 s41, s41, s42\n v_add_u32 v41, v41, v42\n s_add_u32 s41, s41, s42\n"
        "s_lshr_b32 s43, s41, 3\n v_lshrrev_b32 v43, 3, v41\n s_add_u32 s41, s41, s43\n s_xor_b32 s41, s41, s42\n v_add_u32 v41, v41, v43\n"
        "s_add_u32 s41, s41, s42\n s_and_b32 s41, s41, 0xffff\n"
        "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_m%=\n"
        "v_add_u32 %[v], s41, v41\n"
        : [v] "=v"(v) : [it] "s"(s) : "s40", "s41", "s42", "s43", "v41", "v42", "v43", "scc");
  } else if (m == 4) {
    asm volatile(
        "s_mov_b32 s40, %[it]\n s_mov_b32 s41, 1\n s_mov_b32 s42, 3\n"
        "L_b%=:\n"
        "s_add_u32 s41, s41, s42\n s_xor_b32 s41, s41, s42\n s_add_u32 s41, s41, s42\n s_lshr_b32 s43, s41, 3\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_b1%=\n s_nop 0\n"
        "L_b1%=:\n"
        "s_add_u32 s41, s41, s43\n s_xor_b32 s41, s41, s42\n s_add_u32 s41, s41, s42\n s_lshr_b32 s43, s41, 3\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_b2%=\n s_nop 0\n"
        "L_b2%=:\n"
        "s_add_u32 s41, s41, s43\n s_xor_b32 s41, s41, s42\n s_add_u32 s41, s41, s42\n s_and_b32 s41, s41, 0xffff\n"
        "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_b%=\n"
        "v_mov_b32 %[v], s41\n"
        : [v] "=v"(v) : [it] "s"(s) : "s40", "s41", "s42", "s43", "scc");
  } else if (m >= 6) {
    // LDS: what the literal store of the inflate loop costs -- all 64 lanes writing the SAME byte address (6), one lane
    // doing it (7), 64 lanes writing 64 consecutive bytes (8); the table lookup: all lanes reading the same dword, 16 in
    // flight (9), one at a time with a wait (10)
    unsigned addr = m == 8 ? threadIdx.x : 128u;
    unsigned long long keep = 0;
    if (m == 7) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 1\n" : "=s"(keep));
    if (m <= 8) {
      asm volatile(
          "s_mov_b32 s40, %[it]\n"
          "L_w%=:\n"
          "ds_write_b8 %[a], %[a]\n ds_write_b8 %[a], %[a] offset:64\n ds_write_b8 %[a], %[a] offset:128\n ds_write_b8 %[a], %[a] offset:192\n"
          "ds_write_b8 %[a], %[a] offset:256\n ds_write_b8 %[a], %[a] offset:320\n ds_write_b8 %[a], %[a] offset:384\n ds_write_b8 %[a], %[a] offset:448\n"
          "ds_write_b8 %[a], %[a] offset:512\n ds_write_b8 %[a], %[a] offset:576\n ds_write_b8 %[a], %[a] offset:640\n ds_write_b8 %[a], %[a] offset:704\n"
          "ds_write_b8 %[a], %[a] offset:768\n ds_write_b8 %[a], %[a] offset:832\n ds_write_b8 %[a], %[a] offset:896\n ds_write_b8 %[a], %[a] offset:960\n"
          "s_waitcnt lgkmcnt(0)\n"
          "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_w%=\n"
          : : [it] "s"(s), [a] "v"(addr) : "s40", "scc", "memory");
    } else if (m == 9) {
      asm volatile(
          "s_mov_b32 s40, %[it]\n"
          "L_r%=:\n"
          "ds_read_b32 v41, %[a]\n ds_read_b32 v42, %[a] offset:64\n ds_read_b32 v43, %[a] offset:128\n ds_read_b32 v44, %[a] offset:192\n"
          "ds_read_b32 v41, %[a] offset:256\n ds_read_b32 v42, %[a] offset:320\n ds_read_b32 v43, %[a] offset:384\n ds_read_b32 v44, %[a] offset:448\n"
          "ds_read_b32 v41, %[a] offset:512\n ds_read_b32 v42, %[a] offset:576\n ds_read_b32 v43, %[a] offset:640\n ds_read_b32 v44, %[a] offset:704\n"
          "ds_read_b32 v41, %[a] offset:768\n ds_read_b32 v42, %[a] offset:832\n ds_read_b32 v43, %[a] offset:896\n ds_read_b32 v44, %[a] offset:960\n"
          "s_waitcnt lgkmcnt(0)\n"
          "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_r%=\n"
          : : [it] "s"(s), [a] "v"(addr) : "s40", "v41", "v42", "v43", "v44", "scc", "memory");
    } else {
      asm volatile(
          "s_mov_b32 s40, %[it]\n v_mov_b32 v41, %[a]\n"
          "L_q%=:\n"
          "ds_read_b32 v42, v41\n s_waitcnt lgkmcnt(0)\n v_and_b32 v41, 0x7c, v42\n ds_read_b32 v42, v41\n s_waitcnt lgkmcnt(0)\n v_and_b32 v41, 0x7c, v42\n"
          "ds_read_b32 v42, v41\n s_waitcnt lgkmcnt(0)\n v_and_b32 v41, 0x7c, v42\n ds_read_b32 v42, v41\n s_waitcnt lgkmcnt(0)\n v_and_b32 v41, 0x7c, v42\n"
          "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_q%=\n"
          : : [it] "s"(s), [a] "v"(addr) : "s40", "v41", "v42", "scc", "memory");
    }
    if (m == 7) asm volatile("s_mov_b64 exec, %0\n" : : "s"(keep));
  } else {
    asm volatile(
        "s_mov_b32 s40, %[it]\n v_mov_b32 v41, 1\n v_mov_b32 v42, 3\n v_mov_b32 v44, 1\n"
        "L_c%=:\n"
        "v_add_u32 v41, v41, v42\n v_xor_b32 v41, v41, v42\n v_add_u32 v41, v41, v42\n v_lshrrev_b32 v43, 3, v41\n v_cmp_ne_u32 vcc, 0, v44\n s_cbranch_vccnz L_c1%=\n s_nop 0\n"
        "L_c1%=:\n"
        "v_add_u32 v41, v41, v43\n v_xor_b32 v41, v41, v42\n v_add_u32 v41, v41, v42\n v_lshrrev_b32 v43, 3, v41\n v_cmp_ne_u32 vcc, 0, v44\n s_cbranch_vccnz L_c2%=\n s_nop 0\n"
        "L_c2%=:\n"
        "v_add_u32 v41, v41, v43\n v_xor_b32 v41, v41, v42\n v_add_u32 v41, v41, v42\n v_and_b32 v41, 0xffff, v41\n"
        "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 L_c%=\n"
        "v_mov_b32 %[v], v41\n"
        : [v] "=v"(v) : [it] "s"(s) : "s40", "v41", "v42", "v43", "v44", "vcc", "scc");
  }
  if (v == 0xdeadbeefu) out[0] = v + dyn_lds[threadIdx.x];
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate / 1e6;
  printf("%s: %d CUs, %.2f GHz (nominal)\n", p.name, cus, ghz);
  unsigned* out;
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const char* names[] = {"scalar chain (16 S + loop 3)", "vector chain (16 V + loop 3 S)", "half the waves each (blockIdx parity)",
                         "12 S + 5 V per iteration + loop 3", "scalar chain, taken branch per 4 (14 S + 2 cmp + 3 B + loop)",
                         "vector chain, v_cmp + taken s_cbranch_vccnz per 4",
                         "LDS: 16 ds_write_b8, 64 lanes on ONE address", "LDS: 16 ds_write_b8, one lane", "LDS: 16 ds_write_b8, 64 consecutive bytes",
                         "LDS: 16 ds_read_b32 of one address in flight", "LDS: 4 dependent ds_read_b32 (wait + v_and each)"};
  for (int mode = 0; mode < 11; ++mode) {
    const int iters = mode >= 6 ? 20000 : 200000;
    for (int W : {1, 4, 8, 16, 24, 32}) {
      const size_t lds = W >= 32 ? 4096 : (size_t)(160 * 1024 / W) - 512;  // caps the resident single-wave workgroups per CU at W
      const size_t use = lds > 64 * 1024 ? 64 * 1024 : lds;
      const int grid = cus * W;
      hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), use, 0, mode, 1000, out);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), use, 0, mode, iters, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      // instructions per iteration as written (loop overhead included)
      const double per_iter = mode == 3 ? 20.0 : mode == 10 ? 15.0 : mode >= 6 ? 20.0 : mode >= 4 ? 23.0 : 19.0;
      const double clk = ms * 1e-3 * ghz * 1e9;
      printf("mode %d %-58s W=%2d (lds %6zu): %8.3f ms  %6.1f clk/iteration/wave  %5.2f instr/clk/CU\n", mode, names[mode], W, use, ms,
             clk / iters, per_iter * iters * W / clk);
    }
  }
  return 0;
}
