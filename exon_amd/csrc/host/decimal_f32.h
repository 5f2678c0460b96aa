// decimal_f32.h -- correctly rounded decimal -> binary32 without libc, usable in device code.
//
// Rust's `str::parse::<f32>` (what noodles uses for QUAL and Float INFO values) is correctly rounded; the host
// decoders get that from a Clinger fast path + strtof (formats.h).  A GPU-side parser has no strtof, so this is the
// Eisel-Lemire algorithm (D. Lemire, "Number parsing at a gigabyte per second", 2021) specialised to binary32:
// a <= 19-digit decimal significand w and a power of ten q are turned into the nearest float with one or two
// 64x64->128 multiplications by a truncated 128-bit power of five.  Inputs it cannot decide (more than 19
// significant digits, exponents outside the table) are reported so the caller can route the row to the host.
// Validated bit-for-bit against strtof on tens of millions of random strings (tests/test_decimal_f32.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EXON_HD __host__ __device__ __forceinline__
#else
#define EXON_HD inline
#endif

namespace exon {
namespace dec {

constexpr int kSmallestPow10 = -65, kLargestPow10 = 38;

// 128-bit truncated powers of five, 5^q for q in [-65, 38] ({high, low}); negative powers are rounded up
// (generation recipe of the fast_float reference table, reproduced with exact integer arithmetic).
#if defined(__HIP_DEVICE_COMPILE__)
__device__
#endif
static const uint64_t kPow5[104][2] = {
    {0x86ccbb52ea94baeaULL, 0x98e947129fc2b4e9ULL},
    {0xa87fea27a539e9a5ULL, 0x3f2398d747b36224ULL},
    {0xd29fe4b18e88640eULL, 0x8eec7f0d19a03aadULL},
    {0x83a3eeeef9153e89ULL, 0x1953cf68300424acULL},
    {0xa48ceaaab75a8e2bULL, 0x5fa8c3423c052dd7ULL},
    {0xcdb02555653131b6ULL, 0x3792f412cb06794dULL},
    {0x808e17555f3ebf11ULL, 0xe2bbd88bbee40bd0ULL},
    {0xa0b19d2ab70e6ed6ULL, 0x5b6aceaeae9d0ec4ULL},
    {0xc8de047564d20a8bULL, 0xf245825a5a445275ULL},
    {0xfb158592be068d2eULL, 0xeed6e2f0f0d56712ULL},
    {0x9ced737bb6c4183dULL, 0x55464dd69685606bULL},
    {0xc428d05aa4751e4cULL, 0xaa97e14c3c26b886ULL},
    {0xf53304714d9265dfULL, 0xd53dd99f4b3066a8ULL},
    {0x993fe2c6d07b7fabULL, 0xe546a8038efe4029ULL},
    {0xbf8fdb78849a5f96ULL, 0xde98520472bdd033ULL},
    {0xef73d256a5c0f77cULL, 0x963e66858f6d4440ULL},
    {0x95a8637627989aadULL, 0xdde7001379a44aa8ULL},
    {0xbb127c53b17ec159ULL, 0x5560c018580d5d52ULL},
    {0xe9d71b689dde71afULL, 0xaab8f01e6e10b4a6ULL},
    {0x9226712162ab070dULL, 0xcab3961304ca70e8ULL},
    {0xb6b00d69bb55c8d1ULL, 0x3d607b97c5fd0d22ULL},
    {0xe45c10c42a2b3b05ULL, 0x8cb89a7db77c506aULL},
    {0x8eb98a7a9a5b04e3ULL, 0x77f3608e92adb242ULL},
    {0xb267ed1940f1c61cULL, 0x55f038b237591ed3ULL},
    {0xdf01e85f912e37a3ULL, 0x6b6c46dec52f6688ULL},
    {0x8b61313bbabce2c6ULL, 0x2323ac4b3b3da015ULL},
    {0xae397d8aa96c1b77ULL, 0xabec975e0a0d081aULL},
    {0xd9c7dced53c72255ULL, 0x96e7bd358c904a21ULL},
    {0x881cea14545c7575ULL, 0x7e50d64177da2e54ULL},
    {0xaa242499697392d2ULL, 0xdde50bd1d5d0b9e9ULL},
    {0xd4ad2dbfc3d07787ULL, 0x955e4ec64b44e864ULL},
    {0x84ec3c97da624ab4ULL, 0xbd5af13bef0b113eULL},
    {0xa6274bbdd0fadd61ULL, 0xecb1ad8aeacdd58eULL},
    {0xcfb11ead453994baULL, 0x67de18eda5814af2ULL},
    {0x81ceb32c4b43fcf4ULL, 0x80eacf948770ced7ULL},
    {0xa2425ff75e14fc31ULL, 0xa1258379a94d028dULL},
    {0xcad2f7f5359a3b3eULL, 0x096ee45813a04330ULL},
    {0xfd87b5f28300ca0dULL, 0x8bca9d6e188853fcULL},
    {0x9e74d1b791e07e48ULL, 0x775ea264cf55347eULL},
    {0xc612062576589ddaULL, 0x95364afe032a819eULL},
    {0xf79687aed3eec551ULL, 0x3a83ddbd83f52205ULL},
    {0x9abe14cd44753b52ULL, 0xc4926a9672793543ULL},
    {0xc16d9a0095928a27ULL, 0x75b7053c0f178294ULL},
    {0xf1c90080baf72cb1ULL, 0x5324c68b12dd6339ULL},
    {0x971da05074da7beeULL, 0xd3f6fc16ebca5e04ULL},
    {0xbce5086492111aeaULL, 0x88f4bb1ca6bcf585ULL},
    {0xec1e4a7db69561a5ULL, 0x2b31e9e3d06c32e6ULL},
    {0x9392ee8e921d5d07ULL, 0x3aff322e62439fd0ULL},
    {0xb877aa3236a4b449ULL, 0x09befeb9fad487c3ULL},
    {0xe69594bec44de15bULL, 0x4c2ebe687989a9b4ULL},
    {0x901d7cf73ab0acd9ULL, 0x0f9d37014bf60a11ULL},
    {0xb424dc35095cd80fULL, 0x538484c19ef38c95ULL},
    {0xe12e13424bb40e13ULL, 0x2865a5f206b06fbaULL},
    {0x8cbccc096f5088cbULL, 0xf93f87b7442e45d4ULL},
    {0xafebff0bcb24aafeULL, 0xf78f69a51539d749ULL},
    {0xdbe6fecebdedd5beULL, 0xb573440e5a884d1cULL},
    {0x89705f4136b4a597ULL, 0x31680a88f8953031ULL},
    {0xabcc77118461cefcULL, 0xfdc20d2b36ba7c3eULL},
    {0xd6bf94d5e57a42bcULL, 0x3d32907604691b4dULL},
    {0x8637bd05af6c69b5ULL, 0xa63f9a49c2c1b110ULL},
    {0xa7c5ac471b478423ULL, 0x0fcf80dc33721d54ULL},
    {0xd1b71758e219652bULL, 0xd3c36113404ea4a9ULL},
    {0x83126e978d4fdf3bULL, 0x645a1cac083126eaULL},
    {0xa3d70a3d70a3d70aULL, 0x3d70a3d70a3d70a4ULL},
    {0xccccccccccccccccULL, 0xcccccccccccccccdULL},
    {0x8000000000000000ULL, 0x0000000000000000ULL},
    {0xa000000000000000ULL, 0x0000000000000000ULL},
    {0xc800000000000000ULL, 0x0000000000000000ULL},
    {0xfa00000000000000ULL, 0x0000000000000000ULL},
    {0x9c40000000000000ULL, 0x0000000000000000ULL},
    {0xc350000000000000ULL, 0x0000000000000000ULL},
    {0xf424000000000000ULL, 0x0000000000000000ULL},
    {0x9896800000000000ULL, 0x0000000000000000ULL},
    {0xbebc200000000000ULL, 0x0000000000000000ULL},
    {0xee6b280000000000ULL, 0x0000000000000000ULL},
    {0x9502f90000000000ULL, 0x0000000000000000ULL},
    {0xba43b74000000000ULL, 0x0000000000000000ULL},
    {0xe8d4a51000000000ULL, 0x0000000000000000ULL},
    {0x9184e72a00000000ULL, 0x0000000000000000ULL},
    {0xb5e620f480000000ULL, 0x0000000000000000ULL},
    {0xe35fa931a0000000ULL, 0x0000000000000000ULL},
    {0x8e1bc9bf04000000ULL, 0x0000000000000000ULL},
    {0xb1a2bc2ec5000000ULL, 0x0000000000000000ULL},
    {0xde0b6b3a76400000ULL, 0x0000000000000000ULL},
    {0x8ac7230489e80000ULL, 0x0000000000000000ULL},
    {0xad78ebc5ac620000ULL, 0x0000000000000000ULL},
    {0xd8d726b7177a8000ULL, 0x0000000000000000ULL},
    {0x878678326eac9000ULL, 0x0000000000000000ULL},
    {0xa968163f0a57b400ULL, 0x0000000000000000ULL},
    {0xd3c21bcecceda100ULL, 0x0000000000000000ULL},
    {0x84595161401484a0ULL, 0x0000000000000000ULL},
    {0xa56fa5b99019a5c8ULL, 0x0000000000000000ULL},
    {0xcecb8f27f4200f3aULL, 0x0000000000000000ULL},
    {0x813f3978f8940984ULL, 0x4000000000000000ULL},
    {0xa18f07d736b90be5ULL, 0x5000000000000000ULL},
    {0xc9f2c9cd04674edeULL, 0xa400000000000000ULL},
    {0xfc6f7c4045812296ULL, 0x4d00000000000000ULL},
    {0x9dc5ada82b70b59dULL, 0xf020000000000000ULL},
    {0xc5371912364ce305ULL, 0x6c28000000000000ULL},
    {0xf684df56c3e01bc6ULL, 0xc732000000000000ULL},
    {0x9a130b963a6c115cULL, 0x3c7f400000000000ULL},
    {0xc097ce7bc90715b3ULL, 0x4b9f100000000000ULL},
    {0xf0bdc21abb48db20ULL, 0x1e86d40000000000ULL},
    {0x96769950b50d88f4ULL, 0x1314448000000000ULL},
};

EXON_HD void mul64(uint64_t a, uint64_t b, uint64_t* hi, uint64_t* lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  *lo = a * b;
  *hi = __umul64hi(a, b);
#else
  const unsigned __int128 p = (unsigned __int128)a * b;
  *lo = (uint64_t)p;
  *hi = (uint64_t)(p >> 64);
#endif
}
EXON_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}

// nearest binary32 (bit pattern, sign excluded) to w * 10^q; w != 0.
EXON_HD uint32_t eisel_lemire_f32(uint64_t w, int q) {
  if (q < kSmallestPow10) return 0u;
  if (q > kLargestPow10) return 0x7F800000u;
  const int lz = clz64(w);
  w <<= lz;
  uint64_t hi, lo;
  mul64(w, kPow5[q - kSmallestPow10][0], &hi, &lo);
  const uint64_t precision_mask = 0xFFFFFFFFFFFFFFFFULL >> 26;  // mantissa bits (23) + 3
  if ((hi & precision_mask) == precision_mask) {
    uint64_t hi2, lo2;
    mul64(w, kPow5[q - kSmallestPow10][1], &hi2, &lo2);
    lo += hi2;
    if (hi2 > lo) ++hi;
  }
  const int upperbit = (int)(hi >> 63);
  const int shift = upperbit + 64 - 23 - 3;
  uint64_t mantissa = hi >> shift;
  int power2 = (int)((((152170 + 65536) * (int64_t)q) >> 16) + 63) + upperbit - lz + 127;
  if (power2 <= 0) {  // subnormal
    if (-power2 + 1 >= 64) return 0u;
    mantissa >>= -power2 + 1;
    mantissa += (mantissa & 1);
    mantissa >>= 1;
    power2 = (mantissa < (1ULL << 23)) ? 0 : 1;
    return ((uint32_t)power2 << 23) | (uint32_t)(mantissa & 0x7FFFFF);
  }
  if (lo <= 1 && q >= -17 && q <= 10 && (mantissa & 3) == 1) {  // exactly halfway: round to even
    if ((mantissa << shift) == hi) mantissa &= ~1ULL;
  }
  mantissa += (mantissa & 1);
  mantissa >>= 1;
  if (mantissa >= (2ULL << 23)) {
    mantissa = 1ULL << 23;
    ++power2;
  }
  mantissa &= ~(1ULL << 23);
  if (power2 >= 0xFF) return 0x7F800000u;
  return ((uint32_t)power2 << 23) | (uint32_t)mantissa;
}

// Parses [p, p+n) as a decimal float (sign, digits, optional fraction, optional exponent).
// Returns 1 and the value's bits on success, 0 if the text is not a plain decimal or needs the slow path
// (more than 19 significant digits, inf/nan spellings, hex floats ...).
EXON_HD int parse_f32(const char* p, int n, uint32_t* bits) {
  int i = 0;
  uint32_t sign = 0;
  if (i < n && (p[i] == '-' || p[i] == '+')) sign = (p[i++] == '-') ? 0x80000000u : 0u;
  // (one exit per digit and nothing else to branch on: leading zeros leave w at 0 by themselves, a digit is significant from the
  //  first one that is not '0' on, and "more than 19" is asked once at the end -- on the GPU every `if` inside a loop that lanes leave at different
  //  times is exec-mask bookkeeping on the CU's one scalar unit, which is what bounds k_parse_lines)
  uint64_t w = 0;
  unsigned nz = 0;  // not 0 from the first digit that is not '0' on
  int digits = 0, q = 0;
  const int i0 = i;
  for (; i < n; ++i) {
    const unsigned d = (unsigned)(unsigned char)p[i] - (unsigned)'0';
    if (d > 9u) break;
    w = w * 10 + d;
    nz |= d;
    digits += nz != 0;
  }
  int nd = i - i0;  // digits met, significant or not
  if (i < n && p[i] == '.') {
    const int f0 = ++i;
    for (; i < n; ++i) {
      const unsigned d = (unsigned)(unsigned char)p[i] - (unsigned)'0';
      if (d > 9u) break;
      w = w * 10 + d;
      nz |= d;
      digits += nz != 0;
    }
    q = f0 - i;
    nd += i - f0;
  }
  if (nd == 0) return 0;
  if (digits > 19) return 0;  // (w has wrapped by then: not used)
  if (i < n && (p[i] == 'e' || p[i] == 'E')) {
    ++i;
    bool eneg = false;
    if (i < n && (p[i] == '-' || p[i] == '+')) eneg = p[i++] == '-';
    int e = 0, ed = 0;
    for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i, ++ed)
      if (e < 100000) e = e * 10 + (p[i] - '0');
    if (ed == 0) return 0;
    q += eneg ? -e : e;
  }
  if (i != n) return 0;
  if (w == 0) {
    *bits = sign;
    return 1;
  }
  // Clinger's exact case first: a significand below 2^24 and a power of ten up to 10^10 are both binary32 values, so ONE correctly
  // rounded division (IEEE; hipcc's default for fp32 too) is the correctly rounded result -- what QUAL and most INFO numbers are
  // (381.1, 0.0123), at a tenth of the instructions of the general path below.
  if (w < (1ull << 24) && q <= 0 && q >= -10) {
    const float p10[11] = {1.f, 10.f, 100.f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
    const float r = (float)(uint32_t)w / p10[-q];
    uint32_t rb;
    __builtin_memcpy(&rb, &r, 4);
    *bits = sign | rb;
    return 1;
  }
  *bits = sign | eisel_lemire_f32(w, q < -100000 ? -100000 : (q > 100000 ? 100000 : q));
  return 1;
}

}  // namespace dec
}  // namespace exon
