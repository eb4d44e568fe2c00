// bf16x3 split helpers shared by the split-bf16 kernels (conv_igemm_bf3.hip, conv_wgrad_bf3.hip).
#pragma once
#include "viai_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// split two floats into three packed bf16 pairs (round-to-nearest at every level)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(r0, r1);
    float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = cvt_pk_bf16(s0, s1);
}

constexpr int BF3_BK = 32;                          // reduction elements per staged chunk (two 16-deep MFMA k-steps)
constexpr int BF3_PITCH = 80;                       // bytes per LDS row: 32 bf16 + 8 pad (conflict-free ds_read_b128)

// ---- f16x2 split (forward convolutions): a * S = h1 + h2 in two fp16 terms (2 x 11 = 22 significand bits), three partial
// products h1w1 + h1w2 + h2w1 on v_mfma_f32_32x32x16_f16 -- half the MFMA work of bf16x3.  fp16 has a narrow exponent
// range, so the operands are pre-scaled by exact powers of two (activations F16_ASCALE, weights F16_WSCALE in the pack
// kernel) and the accumulator is scaled back in the epilogue; measured error vs fp64 equals the fp32 kernels' for O(1)
// activations (BatchNorm outputs, mel inputs).  Gradients span too many decades for static scales: the backward
// kernels stay on bf16x3.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16_ASCALE = 16.0f, F16_WSCALE = 256.0f;

__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float f16_lo(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float f16_hi(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }

// two floats -> two packed fp16 pairs (p1 = the leading fp16 terms of x0*S, x1*S; p2 = the fp16 remainders), S a power of two
// held in a scalar register, L = 65504 / S.  Six instructions: the clamp (values beyond the fp16 range saturate instead of
// turning into inf / NaN: with the static activation scale that is |x| > ~4094, far outside anything a normalised network
// produces), then v_fma_mixlo/hi_f16 does scale + round, and scale + subtract-leading-term + round, in one fma each (the
// scaling is exact, so the remainder x*S - h1 is formed without rounding before the single conversion to fp16).
__device__ __forceinline__ void split2_pair(float x0, float x1, float S, float L, unsigned& p1, unsigned& p2) {
    x0 = __builtin_amdgcn_fmed3f(x0, -L, L);
    x1 = __builtin_amdgcn_fmed3f(x1, -L, L);
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(p1) : "v"(x0), "s"(S));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(p1) : "v"(x1), "s"(S));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(p2) : "v"(x0), "s"(S), "v"(p1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(p2) : "v"(x1), "s"(S), "v"(p1));
}
__device__ __forceinline__ float f16_clamp_for_scale(float S) { return 65504.f / S; }

// dynamic f16x2 operand scale from a device-side abs-max: the power of two that puts max |x| in [2^13, 2^14)
__device__ __forceinline__ float f16_scale_from_amax(const float* amax) {
    const unsigned bits = __float_as_uint(*amax);
    if (bits == 0u) return 1.f;
    int field = 127 + 14 - ((int)((bits >> 23) & 255u) - 126);          // 2^(14 - e), amax < 2^e
    field = field < 1 ? 1 : (field > 254 ? 254 : field);
    return __uint_as_float((unsigned)field << 23);
}
