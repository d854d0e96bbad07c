// Shared pieces of the EXPERIMENTAL 3xBF16 kernels (conv3d_b3.hip, conv3d_b3_wgrad.hip): the exact 3-way bf16 split of
// an fp32 value and the bf16 MFMA wrapper.
//
// x = hi + mid + lo with hi = RNE_bf16(x), mid = RNE_bf16(x - hi), lo = RNE_bf16(x - hi - mid): the two residuals are
// exact fp32 subtractions and the sum reproduces x to 2^-27 |x|.  A product x*w is taken as the six cross terms of
// weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi), each exact in fp32 (8 x 8 significand bits),
// accumulated in fp32 by the MFMA, smallest first; the three dropped terms are <= 2^-25 |x w|.
#pragma once
#include "common.h"

typedef float b3_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned b3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned b3_u32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline unsigned b3_bits(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
__host__ __device__ inline float b3_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
__host__ __device__ inline unsigned b3_bf16_rne(float x) {      // fp32 -> bf16 bits, round to nearest even
  unsigned u = b3_bits(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__host__ __device__ inline void b3_split(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = b3_bf16_rne(x);
  const float r1 = x - b3_float(hi << 16);
  mid = b3_bf16_rne(r1);
  const float r2 = r1 - b3_float(mid << 16);
  lo = b3_bf16_rne(r2);
}

// two fp32 -> three words of packed bf16 pairs (element 0 in the low half); v_cvt_pk_bf16_f32 on the GPU
__device__ __forceinline__ void b3_split_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef CFUN_HIP_EMULATION
  unsigned h0, m0, l0, h1, m1, l1;
  b3_split(x0, h0, m0, l0);
  b3_split(x1, h1, m1, l1);
  hi = h0 | (h1 << 16); mid = m0 | (m1 << 16); lo = l0 | (l1 << 16);
#else
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r1 = v - f32x2{b3_float(hi << 16), b3_float(hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 r2 = r1 - f32x2{b3_float(mid << 16), b3_float(mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
#endif
}

// D = A (16 rows x 32 k) * B (32 k x 16 columns) + C on v_mfma_f32_16x16x32_bf16; a / b carry 8 bf16 as raw bits
#ifdef CFUN_HIP_EMULATION
inline b3_f32x4 b3_mfma(b3_u32x4 a, b3_u32x4 b, b3_f32x4 c) { return hipemu_mfma_16x16x32_bf16(a, b, c); }
#else
__device__ __forceinline__ b3_f32x4 b3_mfma(b3_u32x4 a, b3_u32x4 b, b3_f32x4 c) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#endif
