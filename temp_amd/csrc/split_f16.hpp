// fp32 products on the f16 matrix pipe: the scaled two-way split (round 6).
//
// gemm_bx.hpp runs an fp32 product as SIX bf16 MFMA products of an exact three-way split (8 + 8 + 8 bits).  The f16 pipe of gfx950
// issues at the same rate (v_mfma_f32_32x32x16_f16: 32 768 flop per 32 cycles per SIMD) and an f16 carries 11 significant bits, so
// TWO pieces hold 22-23 bits and THREE products do:
//
//     x' = x . s                     s = a power of two chosen per ROW of the activations-side operand / per COLUMN of the
//                                    weights-side operand, so that the largest |x'| of the row lies in [2^14, 2^15) (f16 max 65 504)
//     h  = rn16(x'),  l = rn16(x' - h)       (x' - h is exact in fp32; |x' - h - l| <= 2^-23 |x'|: one fp32 rounding)
//     a.b = [ ah.bh + (ah.bl + al.bh) ] / (sa . sb)   +   [ al.bl : dropped, <= 2^-22 |a.b| ]
//
// Every f16 x f16 product is exact in the fp32 accumulator of the MFMA (22 bits), the scales are powers of two and factor out of
// the sum exactly, so the result differs from the exact fp32 product by the residual of the two cuts and the dropped term -- the
// error class of the three-way bf16 split (measured against fp64: tools/f16x2_probe.hip, profiles/r06_f16x2_probe.txt) at HALF the
// MFMA count and two operand planes instead of three.
//
// Range.  f16 has 5 exponent bits: below 2^-14 its spacing is a constant 2^-24.  With the row maximum scaled to [2^14, 2^15) the
// piece h is normal for every element within 2^-28 of the row maximum and l for every element within 2^-17 of it; smaller
// elements keep an ABSOLUTE error of 2^-25 / s <= 2^-39 of the row maximum (they are cut on the subnormal grid, which the MFMA
// honours: checked by the probe) -- against the 2^-24 RELATIVE error fp32 itself gives the row maximum that is noise.
// Non-finite operands: inf . s = inf, l = inf - inf = NaN: the outputs that depend on a non-finite element are non-finite (NaN),
// nothing else changes -- the contract of gemm_bx.hpp; a row / column maximum of inf or NaN only fixes that row's / column's scale.
#pragma once
#include <hip/hip_runtime.h>

namespace temp {

typedef _Float16 hx_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hx_f16x2 __attribute__((ext_vector_type(2)));
typedef float hx_f2 __attribute__((ext_vector_type(2)));
typedef unsigned int hx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int hx_u32x2 __attribute__((ext_vector_type(2)));

// |x| as an unsigned key: monotone in |x| for finite values, NaN > inf > finite (integer max = order-independent, deterministic)
__device__ __forceinline__ unsigned hx_abs_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
__device__ __forceinline__ unsigned hx_abs_bits4(float4 v) {
  return max(max(hx_abs_bits(v.x), hx_abs_bits(v.y)), max(hx_abs_bits(v.z), hx_abs_bits(v.w)));
}
// the power of two s with  absmax . s  in [2^14, 2^15), from the key of the row / column maximum; and 1 / s.  The exponent is
// clamped so that both are normal numbers: an all-zero row gets s = 2^125 (0 . s = 0), a non-finite maximum s = 2^-113.
__host__ __device__ __forceinline__ unsigned hx_clamp_exp(unsigned abs_bits) {
  unsigned e = abs_bits >> 23;
  e = e < 16u ? 16u : e;
  return e > 254u ? 254u : e;
}
__device__ __forceinline__ float hx_scale(unsigned abs_bits) { return __uint_as_float((268u - hx_clamp_exp(abs_bits)) << 23); }
__device__ __forceinline__ float hx_inv_scale(unsigned abs_bits) { return __uint_as_float((hx_clamp_exp(abs_bits) - 14u) << 23); }

// two consecutive elements, scaled by s -> one dword per plane (element 0 in the low half): 6 VALU instructions
// (v_pk_mul_f32, v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32); the bf16 three-way split takes 9
__device__ __forceinline__ void hx_split_pair(float x0, float x1, float s, unsigned& H, unsigned& L) {
  const hx_f2 y = {x0 * s, x1 * s};
  const hx_f16x2 h = __builtin_convertvector(y, hx_f16x2);
  const hx_f2 r = {y[0] - (float)h[0], y[1] - (float)h[1]};
  const hx_f16x2 l = __builtin_convertvector(r, hx_f16x2);
  H = __builtin_bit_cast(unsigned, h);
  L = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void hx_split4(const float4 a, float s, hx_u32x2& H, hx_u32x2& L) {
  unsigned h, l;
  hx_split_pair(a.x, a.y, s, h, l); H[0] = h; L[0] = l;
  hx_split_pair(a.z, a.w, s, h, l); H[1] = h; L[1] = l;
}
__device__ __forceinline__ void hx_split8(const float4 a, const float4 b, float s, hx_u32x4& H, hx_u32x4& L) {
  unsigned h, l;
  hx_split_pair(a.x, a.y, s, h, l); H[0] = h; L[0] = l;
  hx_split_pair(a.z, a.w, s, h, l); H[1] = h; L[1] = l;
  hx_split_pair(b.x, b.y, s, h, l); H[2] = h; L[2] = l;
  hx_split_pair(b.z, b.w, s, h, l); H[3] = h; L[3] = l;
}
__device__ __forceinline__ hx_f16x8 hx_frag(const hx_u32x4 v) { return __builtin_bit_cast(hx_f16x8, v); }

// acc += w . a with the three significant products, small terms first (w = weights-side fragment, a = activations-side)
#define HX_MMA(acc, wh, wl, ah, al)                                                            \
  do {                                                                                         \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, acc, 0, 0, 0);                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc, 0, 0, 0);                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc, 0, 0, 0);                        \
  } while (0)

}  // namespace temp
