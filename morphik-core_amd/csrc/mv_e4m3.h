// mv_e4m3.h -- the e4m3fn (OCP FP8) codec of the device code (mirrors oracle/mv_oracle.c: orc_e4m3_encode / _decode / orc_pow2_scale_exp).
// Shared by mv_fp8.hip (page slab, query split) and mv_fde_batch.hip (the query terms of the e4m3 FDE pass).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mv {
namespace {

// v finite, |v| arbitrary: round to nearest even e4m3fn, saturating at +-448.
__device__ __forceinline__ uint32_t e4m3_encode(float v) {
  const uint32_t u = __float_as_uint(v);
  const uint32_t sign = (u >> 24) & 0x80u;
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return sign | 0x7fu;  // NaN
  if (a >= 0x43e80000u) return sign | 0x7eu;  // |v| >= 464 rounds past 448 -> saturate (also inf)
  int e = (int)(a >> 23) - 127;
  if (e < -6) e = -6;
  const float q = rintf(__uint_as_float(a) * __uint_as_float((uint32_t)(127 + 3 - e) << 23));  // exact scaling, RNE
  int qi = (int)q;
  if (qi == 16) { qi = 8; e += 1; }
  uint32_t code = (e == -6 && qi < 8) ? (uint32_t)qi : (uint32_t)(((e + 7) << 3) | (qi - 8));
  if (code > 0x7eu) code = 0x7eu;
  return sign | code;
}
__device__ __forceinline__ float e4m3_decode(uint32_t c) {
  const uint32_t E = (c >> 3) & 15u, M = c & 7u;
  const float mag = E == 0 ? (float)M * 0.001953125f /* 2^-9 */ : (float)(8u + M) * __uint_as_float((E + 127u - 10u) << 23);
  return (c & 0x80u) ? -mag : mag;
}
// floor(log2(448 / amax)) for amax > 0 given as fp32 bits; 0 for amax == 0.  Clamped to +-100.
__device__ __forceinline__ int pow2_scale_exp(uint32_t amax_bits) {
  if ((amax_bits & 0x7fffffffu) == 0u) return 0;
  const int ea = (int)((amax_bits >> 23) & 0xffu) - 127;
  const uint32_t mant = amax_bits & 0x7fffffu;
  int e = 8 - ea - (mant > 0x600000u ? 1 : 0);  // 448 = 1.75 * 2^8
  if (e > 100) e = 100;
  if (e < -100) e = -100;
  return e;
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }

// ---- FP4 (e2m1) codec of the FDE slab's 4-bit copy (mv_fde4.hip, the F4 form of mv_fde_batch.hip; oracle: orc_fp4_encode / _decode):
// bit 3 sign, bits 2..0 -> {0, 0.5, 1, 1.5, 2, 3, 4, 6}; round to nearest, ties to the even code, saturating at 6.
__device__ __forceinline__ uint32_t fp4_encode(float y) {
  const float a = fabsf(y);
  uint32_t c;
  if (a <= 0.25f) c = 0;
  else if (a < 0.75f) c = 1;
  else if (a <= 1.25f) c = 2;
  else if (a < 1.75f) c = 3;
  else if (a <= 2.5f) c = 4;
  else if (a < 3.5f) c = 5;
  else if (a <= 5.0f) c = 6;
  else c = 7;
  return c | ((__float_as_uint(y) >> 31) << 3);
}
__device__ __forceinline__ float fp4_decode(uint32_t c) {
  const uint32_t m = c & 7u;
  const float mag = m < 4u ? 0.5f * (float)m : (m == 4u ? 2.0f : (m == 5u ? 3.0f : (m == 6u ? 4.0f : 6.0f)));
  return (c & 8u) ? -mag : mag;
}

}  // namespace
}  // namespace mv
