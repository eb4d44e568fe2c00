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

// ---- P16: activation / gradient tensors stored PRE-SPLIT (round 4).  Same bytes as the fp32 NHWC tensor, C % 32 == 0: per pixel, per
// 32-channel group g, [32 leading fp16 terms (64 B)][32 remainder terms (64 B)] at byte offset g * 128 of the pixel's C * 4 bytes -- i.e.
// exactly what split2_pair would make of the group's fp32 values with the tensor's scale.  A 16-byte piece q (0 .. 7) of a group holds
// plane q >> 2, channels 8 (q & 3) .. + 7; a consumer that staged fp32 channel quads (16 B per lane, split, two 8-byte LDS stores) stages
// pieces instead (16 B per lane, one 16-byte LDS store into the piece's plane): same addresses, same byte counts, no conversion.  The
// scale is the power of two f16_scale_from_amax derives from the tensor's magnitude slot, which the PRODUCER fills with an a-priori bound
// (BatchNorm: |gamma| sqrt(M - 1) + |beta| by Samuelson's inequality; its backward: from per-channel max |d pre|), so it is known before
// the pass that writes the planes.
__device__ __forceinline__ float f16_scale_from_amax_value(float amax) {
    const unsigned bits = __float_as_uint(amax);
    if (bits == 0u) return 1.f;
    int field = 127 + 14 - ((int)((bits >> 23) & 255u) - 126);
    field = field < 1 ? 1 : (field > 254 ? 254 : field);
    return __uint_as_float((unsigned)field << 23);
}
// eight consecutive channels (two fp32 quads) -> their leading / remainder pieces
__device__ __forceinline__ void p16_split8(const f32x4& a, const f32x4& b, float S, float L, u32x4& hi, u32x4& lo) {
    unsigned h, l;
    split2_pair(a[0], a[1], S, L, h, l); hi[0] = h; lo[0] = l;
    split2_pair(a[2], a[3], S, L, h, l); hi[1] = h; lo[1] = l;
    split2_pair(b[0], b[1], S, L, h, l); hi[2] = h; lo[2] = l;
    split2_pair(b[2], b[3], S, L, h, l); hi[3] = h; lo[3] = l;
}
// One staged 16-byte item of a 32-channel chunk row going to LDS.  fp32 source: channel quad q (0 .. 7), split into both planes (8 bytes
// each at q * 8); P16 source: piece q, copied into its plane.  `row` = the row's address in plane 0, `plane` = bytes between the planes.
template <bool P16>
__device__ __forceinline__ void stage_put32(unsigned char* row, int plane, int q, const u32x4& raw, float S, float L) {
    if constexpr (P16) {
        *reinterpret_cast<u32x4*>(row + (q >> 2) * plane + (q & 3) * 16) = raw;
    } else {
        const f32x4 v = __builtin_bit_cast(f32x4, raw);
        unsigned a1, a2, b1, b2;
        split2_pair(v[0], v[1], S, L, a1, a2);
        split2_pair(v[2], v[3], S, L, b1, b2);
        const u32x2 p1 = {a1, b1}, p2 = {a2, b2};
        *reinterpret_cast<u32x2*>(row + q * 8) = p1;
        *reinterpret_cast<u32x2*>(row + plane + q * 8) = p2;
    }
}

// max of v over the block, returned to every thread (blocks of 64 .. 1024 threads; every thread must call it)
__device__ __forceinline__ float block_max_all(float v) {
    __shared__ float viai_bmx[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) viai_bmx[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = viai_bmx[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, viai_bmx[w]);
    return m;
}

// the tensor-wide bound |gamma| rad + |beta| over C channels (forward P16 producers; rad = sqrt(M - 1) of the statistics' population), identical in
// every block; block 0 publishes it in *amax.  Returns the scale.
__device__ __forceinline__ float p16_fwd_scale(const float* __restrict__ gamma, const float* __restrict__ beta, int C, float rad, float* __restrict__ amax) {
    float b = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) b = fmaxf(b, fabsf(gamma ? gamma[c] : 1.f) * rad + fabsf(beta ? beta[c] : 0.f));
    const float bound = block_max_all(b) * 1.001f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *amax = bound;
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(f16_scale_from_amax_value(bound))));
}
// one channel quad (channels 4 q4 .. 4 q4 + 3 of the pixel at `pix`, a float pointer to the pixel's first channel) into the P16 planes: two 8-byte stores
__device__ __forceinline__ void p16_store_quad(float* pix, int q4, const f32x4& v, float S, float L) {
    unsigned h0, l0, h1, l1;
    split2_pair(v[0], v[1], S, L, h0, l0);
    split2_pair(v[2], v[3], S, L, h1, l1);
    unsigned char* d = reinterpret_cast<unsigned char*>(pix) + (q4 >> 3) * 128 + (q4 & 7) * 8;
    const u32x2 hi = {h0, h1}, lo = {l0, l1};
    *reinterpret_cast<u32x2*>(d) = hi;
    *reinterpret_cast<u32x2*>(d + 64) = lo;
}
