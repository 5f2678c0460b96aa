// list_kernels.h -- shared by the VCF text parser (gpu_parse.hip) and the BCF parser (bcf_parse.hip): the bookkeeping of
// list-valued INFO fields in the Arrow List layout.  A format's extract kernel leaves, per row, the number of items (0 for a
// NULL list) and where they are; here the counts become int32 offsets (block sums -> scan -> the format's fill kernel adds
// the in-block prefix) and the per-item flags become the child validity bitmap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {  // one copy per translation unit

constexpr int LIST_TPB = 256;
// per-workgroup sums of cnt[0 .. n_rows) (n_rows read from the device: the line count of this slab)
__global__ __launch_bounds__(LIST_TPB) void k_list_block_sums(const uint32_t* __restrict__ cnt, const unsigned* __restrict__ n_rows_p,
                                                              unsigned cap, unsigned* __restrict__ block_sums) {
  __shared__ unsigned red[LIST_TPB / 64];
  const unsigned n_rows = min(*n_rows_p, cap);
  const unsigned row = blockIdx.x * LIST_TPB + threadIdx.x;
  unsigned c = row < n_rows ? cnt[row] : 0u;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of `nb` workgroup totals in place; total -> *total_out
// (One workgroup of 256 threads, not 1024: a workgroup starts only when ONE CU has wave slots for all of it, and beside the
//  inflate of the next slab -- 30 of a CU's 32 slots -- sixteen free slots took 0.5-1.3 ms to appear: round 4.)
__global__ __launch_bounds__(256) void k_list_scan_blocks(unsigned* __restrict__ counts, int nb, unsigned* __restrict__ total_out) {
  __shared__ unsigned part[256];
  const int per = (nb + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
  unsigned s = 0;
  for (int b = b0; b < b1; ++b) s += counts[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int b = b0; b < b1; ++b) {
    const unsigned c = counts[b];
    counts[b] = run;
    run += c;
  }
  if (threadIdx.x == 255) *total_out = part[255];
}
// row -> first item index: block_offsets[block] + the prefix of the counts inside the block (call with all LIST_TPB threads)
__device__ __forceinline__ unsigned list_first_item(unsigned c, const unsigned* __restrict__ block_offsets) {
  __shared__ unsigned wave_tot[LIST_TPB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned base = block_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  return base + incl - c;
}
// byte-per-item flags -> Arrow validity bitmap (8 items per thread); n = offsets[n_rows], never more than the `cap_items`
// the flag / bitmap buffers were sized for (a slab whose items do not fit is handed back to the host decoder by the fill
// kernel's exception count: nothing past the buffers may be touched on the way there)
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t* __restrict__ flags, const int32_t* __restrict__ offsets,
                                                   const unsigned* __restrict__ n_rows_p, unsigned cap, unsigned cap_items,
                                                   uint8_t* __restrict__ bitmap) {
  const unsigned n = min((unsigned)offsets[min(*n_rows_p, cap)], cap_items);
  for (unsigned b = blockIdx.x * 256 + threadIdx.x; b * 8 < n; b += gridDim.x * 256) {
    unsigned v = 0;
    for (unsigned k = 0; k < 8 && b * 8 + k < n; ++k) v |= (unsigned)(flags[b * 8 + k] & 1) << k;
    bitmap[b] = (uint8_t)v;
  }
}

}  // namespace
