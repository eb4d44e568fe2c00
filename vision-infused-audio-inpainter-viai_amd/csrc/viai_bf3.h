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
