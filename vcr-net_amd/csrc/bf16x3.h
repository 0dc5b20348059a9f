// Shared pieces of the opt-in exact-split kernels on the bf16 matrix pipe (attention_bf16x3.hip, edgeconv_bf16x3.hip):
// an fp32 value is split EXACTLY into three bf16 pieces x = x1 + x2 + x3 (8 + 8 + 8 significand bits, both
// subtractions exact in fp32) and a product a.b is evaluated as the six partial products of weight >= 2^-16 of the
// leading one, a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2), on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
#pragma once
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: a -> low half, RNE
  const bf16x2_t v = __builtin_convertvector(f32x2{a, b}, bf16x2_t);
  return __builtin_bit_cast(unsigned, v);
}
// exact 3-way split of two floats, packed pairs out (both subtractions are exact in fp32)
__device__ __forceinline__ void split3x2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pk_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
  l = pk_bf16(sa, sb);
}
__device__ __forceinline__ void split3x8(const float* x, bf16x8& h, bf16x8& m, bf16x8& l) {
  u32x4 H, M, L;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned hh, mm, ll;
    split3x2(x[2 * i], x[2 * i + 1], hh, mm, ll);
    H[i] = hh; M[i] = mm; L[i] = ll;
  }
  h = __builtin_bit_cast(bf16x8, H); m = __builtin_bit_cast(bf16x8, M); l = __builtin_bit_cast(bf16x8, L);
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// c += a . b with a = a0 + a1 + a2, b = b0 + b1 + b2 (plane 0 = leading piece); smallest terms first
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&a)[3], bf16x8 b0, bf16x8 b1, bf16x8 b2, f32x16 c) {
  c = mfma_bf16(a[1], b1, c);
  c = mfma_bf16(a[0], b2, c);
  c = mfma_bf16(a[2], b0, c);
  c = mfma_bf16(a[0], b1, c);
  c = mfma_bf16(a[1], b0, c);
  c = mfma_bf16(a[0], b0, c);
  return c;
}
