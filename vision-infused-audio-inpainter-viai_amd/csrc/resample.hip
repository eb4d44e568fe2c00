// Bilinear (align_corners=True) resize and the bottleneck (k,1) average pool on
// NHWC fp32 (gfx950); pure HBM streaming, 16-byte channel vectors per lane.
// Reference call sites: New_Inpainting_Networks.py:78,83 (F.interpolate),
// Inpainting_Networks.py:65,77 (nn.AvgPool2d((3,1)), floor mode).
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"

namespace {

// torch's area_pixel_compute_scale<float>(in, out, align_corners=true)
__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

__device__ __forceinline__ void ac_src(float scale, int dst, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * (float)dst;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// the interpolation with its roundings spelled out (three fused multiply-adds over three products), so that every kernel that resizes
// -- alone or inside the BatchNorm apply pass -- produces the same bits whatever the compiler would contract
__device__ __forceinline__ f32x4 ac_lerp(const f32x4& v00, const f32x4& v01, const f32x4& v10, const f32x4& v11, float lx0, float lx1, float ly0, float ly1) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t0 = __builtin_fmaf(lx1, v01[e], lx0 * v00[e]);
        const float t1 = __builtin_fmaf(lx1, v11[e], lx0 * v10[e]);
        o[e] = __builtin_fmaf(ly1, t1, ly0 * t0);
    }
    return o;
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int N, int IH, int IW, int OH, int OW, int C) {
    const int c4n = C / 4;
    const long total = (long)N * OH * OW * c4n;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c4 = (int)(i % c4n); long r = i / c4n;
        int ox = (int)(r % OW); r /= OW;
        int oy = (int)(r % OH); int n = (int)(r / OH);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        ac_src(sh, oy, IH, y0, y1, ly0, ly1);
        ac_src(sw, ox, IW, x0, x1, lx0, lx1);
        const float* b = x + (size_t)n * IH * IW * C + c4 * 4;
        f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * IW + x0) * C);
        f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * IW + x1) * C);
        f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * IW + x0) * C);
        f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * IW + x1) * C);
        f32x4 o = ac_lerp(v00, v01, v10, v11, lx0, lx1, ly0, ly1);
        *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = o;
    }
}

// gather form of the backward: every input pixel sums the output pixels that
// read it (deterministic, no atomics).  Candidates are bounded conservatively
// and accepted by re-evaluating the forward index computation bit-for-bit.
__device__ __forceinline__ void cand_range(float scale, int i, int out, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 1.f) * inv) - 1;
    hi = (int)ceilf(((float)i + 1.f) * inv) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           int N, int IH, int IW, int OH, int OW, int C) {
    const int c4n = C / 4;
    const long total = (long)N * IH * IW * c4n;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c4 = (int)(i % c4n); long r = i / c4n;
        int ix = (int)(r % IW); r /= IW;
        int iy = (int)(r % IH); int n = (int)(r / IH);
        int ylo, yhi, xlo, xhi;
        cand_range(sh, iy, OH, ylo, yhi);
        cand_range(sw, ix, OW, xlo, xhi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* b = dy + (size_t)n * OH * OW * C + c4 * 4;
        for (int oy = ylo; oy <= yhi; ++oy) {
            int y0, y1; float l0, l1;
            ac_src(sh, oy, IH, y0, y1, l0, l1);
            float wy = (y0 == iy ? l0 : 0.f) + (y1 == iy ? l1 : 0.f);
            if (y0 != iy && y1 != iy) continue;
            f32x4 rowacc = {0.f, 0.f, 0.f, 0.f};
            for (int ox = xlo; ox <= xhi; ++ox) {
                int x0, x1; float m0, m1;
                ac_src(sw, ox, IW, x0, x1, m0, m1);
                if (x0 != ix && x1 != ix) continue;
                float wx = (x0 == ix ? m0 : 0.f) + (x1 == ix ? m1 : 0.f);
                rowacc += wx * *reinterpret_cast<const f32x4*>(b + ((size_t)oy * OW + ox) * C);
            }
            acc += wy * rowacc;
        }
        *reinterpret_cast<f32x4*>(dx + (size_t)i * 4) = acc;
    }
}

// ---- the same two passes with the index arithmetic out of the way.  The kernels above decode (n, y, x, channel quad) of every float4 with three
// 64-bit divisions (~100 instructions each), and the backward walks a conservatively padded candidate window (up to 9 x 9 output pixels) recomputing the
// forward's source index for every (row, column) pair although at most ~5 x 5 of them touch the input pixel: 62 us to gather 134 MB, 46 us to write it
// -- instruction-bound streaming passes.  Here a thread keeps ONE channel quad (256 % (C / 4) == 0) and decodes a pixel index < 2^24 with two
// float-reciprocal divisions; the backward shrinks each axis' candidate range to the outputs that really touch the pixel (they are contiguous),
// evaluates the <= NW column weights once, and then only loads and multiplies.  Same terms in the same order as the kernels above.
__device__ __forceinline__ void rs_divmod(int q, int d, float inv_d, int& quo, int& rem) {       // 0 <= q < 2^24
    quo = (int)((float)q * inv_d);
    rem = q - quo * d;
    if (rem < 0) { --quo; rem += d; } else if (rem >= d) { ++quo; rem -= d; }
}

__global__ __launch_bounds__(256) void bilinear_fwd_px_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int N, int IH, int IW, int OH, int OW, int C, int c4sh) {
    const int c4n = 1 << c4sh, c4 = threadIdx.x & (c4n - 1), pl = threadIdx.x >> c4sh, ppb = 256 >> c4sh;
    const int npix = N * OH * OW;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    const float inv_ow = 1.0f / (float)OW, inv_oh = 1.0f / (float)OH;
    for (int p = blockIdx.x * ppb + pl; p < npix; p += gridDim.x * ppb) {
        int r_, ox, n, oy;
        rs_divmod(p, OW, inv_ow, r_, ox);
        rs_divmod(r_, OH, inv_oh, n, oy);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        ac_src(sh, oy, IH, y0, y1, ly0, ly1);
        ac_src(sw, ox, IW, x0, x1, lx0, lx1);
        const float* b = x + (size_t)n * IH * IW * C + c4 * 4;
        f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * IW + x0) * C);
        f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * IW + x1) * C);
        f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * IW + x0) * C);
        f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * IW + x1) * C);
        f32x4 o = ac_lerp(v00, v01, v10, v11, lx0, lx1, ly0, ly1);
        *reinterpret_cast<f32x4*>(y + ((size_t)p * c4n + c4) * 4) = o;
    }
}

// BatchNorm apply + activation + bilinear resize in ONE pass: out = interpolate(act(scale * y + shift)) without the tensor in between --
// the last layer of every decoder block is followed by F.interpolate (New_Inpainting_Networks.py:76-83), and the post-activation map z
// was written by the apply pass only to be read back by the resize.  The four taps are normalised on load (an output pixel's taps are
// shared with its neighbours through L1); max |z| over the taps is max |z| of the whole map when the resize does not shrink it (every
// input pixel is then some output pixel's tap), which is all the consumers' operand scale needs, and the resize never exceeds it.
// One 1024-thread block per CU: the pass ends in the abs-max atomics (viai_common.h block_absmax_to).
// P16: the resized tensor is written pre-split (viai_bf3.h; C % 32 == 0) with the scale of the bound |gamma| rad + |beta| -- an interpolation never
// exceeds its taps -- which block 0 stores in *amax (no abs-max reduction, no atomics)
template <int ACT, bool P16 = false>
__global__ __launch_bounds__(1024) void bn_act_bilinear_fwd_kernel(const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   float* __restrict__ out, int N, int IH, int IW, int OH, int OW, int C, int c4sh,
                                                                   float slope, float* __restrict__ amax, const float* __restrict__ gamma = nullptr,
                                                                   const float* __restrict__ beta = nullptr, float rad = 0.f) {
    float pS = 1.f, pL = 0.f;
    if constexpr (P16) { pS = p16_fwd_scale(gamma, beta, C, rad, amax); pL = f16_clamp_for_scale(pS); }
    const int c4n = 1 << c4sh, c4 = threadIdx.x & (c4n - 1), pl = threadIdx.x >> c4sh, ppb = 1024 >> c4sh;
    const int npix = N * OH * OW;
    const float sh_ = ac_scale(IH, OH), sw_ = ac_scale(IW, OW);
    const float inv_ow = 1.0f / (float)OW, inv_oh = 1.0f / (float)OH;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4), sf = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
    float mx = 0.f;
    auto z = [&](const f32x4& v) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = viai_act(v[e] * sc[e] + sf[e], ACT, slope); mx = fmaxf(mx, fabsf(o[e])); }
        return o;
    };
    for (int p = blockIdx.x * ppb + pl; p < npix; p += gridDim.x * ppb) {
        int r_, ox, n, oy;
        rs_divmod(p, OW, inv_ow, r_, ox);
        rs_divmod(r_, OH, inv_oh, n, oy);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        ac_src(sh_, oy, IH, y0, y1, ly0, ly1);
        ac_src(sw_, ox, IW, x0, x1, lx0, lx1);
        const float* b = y + (size_t)n * IH * IW * C + c4 * 4;
        const f32x4 r00 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * IW + x0) * C);
        const f32x4 r01 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * IW + x1) * C);
        const f32x4 r10 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * IW + x0) * C);
        const f32x4 r11 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * IW + x1) * C);
        const f32x4 v00 = z(r00), v01 = z(r01), v10 = z(r10), v11 = z(r11);
        const f32x4 o = ac_lerp(v00, v01, v10, v11, lx0, lx1, ly0, ly1);       // the expression of bilinear_fwd_kernel
        if constexpr (P16) p16_store_quad(out + (size_t)p * c4n * 4, c4, o, pS, pL);
        else *reinterpret_cast<f32x4*>(out + ((size_t)p * c4n + c4) * 4) = o;
    }
    if constexpr (!P16) { if (amax != nullptr) block_absmax_to(amax, mx); }
}

// weight with which output index o reads input index i along one axis (0 and false when it does not)
__device__ __forceinline__ bool ac_touch(float scale, int o, int in, int i, float& w) {
    int i0, i1; float l0, l1;
    ac_src(scale, o, in, i0, i1, l0, l1);
    w = (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
    return i0 == i || i1 == i;
}

template <int NW>
__global__ __launch_bounds__(256) void bilinear_bwd_px_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                              int N, int IH, int IW, int OH, int OW, int C, int c4sh) {
    const int c4n = 1 << c4sh, c4 = threadIdx.x & (c4n - 1), pl = threadIdx.x >> c4sh, ppb = 256 >> c4sh;
    const int npix = N * IH * IW;
    const float sh = ac_scale(IH, OH), sw = ac_scale(IW, OW);
    const float inv_iw = 1.0f / (float)IW, inv_ih = 1.0f / (float)IH;
    // (a block's pixels are a strip of one row.  8 x 4 tiles in XCD-contiguous order -- so that the two input rows an output row feeds are
    // read through the same L2 -- were SLOWER: 68.7 vs 57 us on the 134 MB gather, same box)
    for (int p = blockIdx.x * ppb + pl; p < npix; p += gridDim.x * ppb) {
        int r_, ix, n, iy;
        rs_divmod(p, IW, inv_iw, r_, ix);
        rs_divmod(r_, IH, inv_ih, n, iy);
        int ylo, yhi, xlo, xhi;
        float w_;
        cand_range(sh, iy, OH, ylo, yhi);
        cand_range(sw, ix, OW, xlo, xhi);
        while (ylo <= yhi && !ac_touch(sh, ylo, IH, iy, w_)) ++ylo;
        while (yhi >= ylo && !ac_touch(sh, yhi, IH, iy, w_)) --yhi;
        while (xlo <= xhi && !ac_touch(sw, xlo, IW, ix, w_)) ++xlo;
        while (xhi >= xlo && !ac_touch(sw, xhi, IW, ix, w_)) --xhi;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* b = dy + (size_t)n * OH * OW * C + c4 * 4;
        if (xhi - xlo < NW) {
            float wx[NW]; bool on[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                wx[k] = 0.f;
                on[k] = (xlo + k <= xhi) && ac_touch(sw, xlo + k, IW, ix, wx[k]);
            }
            for (int oy = ylo; oy <= yhi; ++oy) {
                float wy;
                if (!ac_touch(sh, oy, IH, iy, wy)) continue;
                const float* row = b + ((size_t)oy * OW + xlo) * C;
                f32x4 v[NW];
#pragma unroll
                for (int k = 0; k < NW; ++k) v[k] = on[k] ? *reinterpret_cast<const f32x4*>(row + (size_t)k * C) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 rowacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NW; ++k) if (on[k]) rowacc += wx[k] * v[k];
                acc += wy * rowacc;
            }
        } else {                                            // strong down-sampling: many outputs per input pixel
            for (int oy = ylo; oy <= yhi; ++oy) {
                float wy;
                if (!ac_touch(sh, oy, IH, iy, wy)) continue;
                f32x4 rowacc = {0.f, 0.f, 0.f, 0.f};
                for (int ox = xlo; ox <= xhi; ++ox) {
                    float wx;
                    if (!ac_touch(sw, ox, IW, ix, wx)) continue;
                    rowacc += wx * *reinterpret_cast<const f32x4*>(b + ((size_t)oy * OW + ox) * C);
                }
                acc += wy * rowacc;
            }
        }
        *reinterpret_cast<f32x4*>(dx + ((size_t)p * c4n + c4) * 4) = acc;
    }
}

// log2(C / 4) when the per-pixel kernels apply (C / 4 a power of two that divides 256, pixel count below 2^24), else -1
inline int px_shift(int C, long npix) {
    const int c4n = C / 4;
    if (c4n < 1 || c4n > 256 || (c4n & (c4n - 1)) != 0 || npix >= (1L << 24)) return -1;
    int s = 0;
    while ((1 << s) < c4n) ++s;
    return s;
}

__global__ __launch_bounds__(256) void avgpool_h_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int N, int IH, int W, int C, int k) {
    const int OH = IH / k, c4n = C / 4;
    const long total = (long)N * OH * W * c4n;
    const float inv = 1.f / (float)k;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long rowlen = (long)W * c4n;
        long r = i / rowlen, off = i % rowlen;
        int oh = (int)(r % OH); int n = (int)(r / OH);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < k; ++j)
            s += *reinterpret_cast<const f32x4*>(x + (((size_t)n * IH + oh * k + j) * rowlen + off) * 4);
        *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = s * inv;
    }
}

__global__ __launch_bounds__(256) void avgpool_h_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            int N, int IH, int W, int C, int k) {
    const int OH = IH / k, c4n = C / 4;
    const long total = (long)N * IH * W * c4n;
    const float inv = 1.f / (float)k;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long rowlen = (long)W * c4n;
        long r = i / rowlen, off = i % rowlen;
        int ih = (int)(r % IH); int n = (int)(r / IH);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ih < OH * k) v = inv * *reinterpret_cast<const f32x4*>(dy + (((size_t)n * OH + ih / k) * rowlen + off) * 4);
        *reinterpret_cast<f32x4*>(dx + (size_t)i * 4) = v;
    }
}

// nn.MaxPool2d(k, s, p) on NHWC with the window argmax kept as one byte per output element (first maximum in
// row-major window order, like torch) so that the backward is an exact gather.  (networks/Image_Embedding.py:21)
// BN: the pooled tensor is act(scale * x + shift) (BatchNorm apply + ReLU of the ResNet stem, networks/Image_Embedding.py:20-23) formed
// while the window is read -- the post-activation tensor is never written; amax receives max |pooled| (= max of the un-pooled tensor for
// ReLU outputs: every pixel lies in a window)
template <bool BN>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                                          int N, int IH, int IW, int OH, int OW, int C, int k, int st, int pd,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, int act, float slope, float* __restrict__ amax) {
    const int c4n = C / 4;
    const long total = (long)N * OH * OW * c4n;
    float mx = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c4 = (int)(i % c4n); long r = i / c4n;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BN) { sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4); sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4); }
        int ox = (int)(r % OW); r /= OW;
        int oy = (int)(r % OH); int n = (int)(r / OH);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bi[4] = {0, 0, 0, 0};
        for (int a = 0; a < k; ++a) {
            int iy = oy * st - pd + a;
            if ((unsigned)iy >= (unsigned)IH) continue;
            for (int b = 0; b < k; ++b) {
                int ix = ox * st - pd + b;
                if ((unsigned)ix >= (unsigned)IW) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * IH + iy) * IW + ix) * C + c4 * 4);
                if constexpr (BN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = viai_act(v[e] * sc[e] + sh[e], act, slope);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[e] > best[e]) { best[e] = v[e]; bi[e] = a * k + b; }
            }
        }
        *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = best;
        *reinterpret_cast<unsigned*>(idx + (size_t)i * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        if constexpr (BN) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(best[0]), fabsf(best[1]))), fmaxf(fabsf(best[2]), fabsf(best[3])));
    }
    if constexpr (BN) { if (amax != nullptr) block_absmax_to(amax, mx); }
}

// The stem's pool (3 x 3, stride 2, pad 1: networks/Image_Embedding.py:23) with compile-time extents: the nine window loads of an output quad are issued
// together (clamped addresses + a validity bit each; the generic kernel above walks the window with two run-time loops and a branch per tap: one dependent
// load at a time), 32-bit index arithmetic (fewer than 2^31 output quads).  Same candidates in the same order: the result and the argmax bytes are the generic
// kernel's bit for bit.  TWIN: the pooled tensor is also written pre-split (P16) for the first BasicBlock's conv1; |pooled| <= |gamma| rad + |beta|.
template <bool TWIN>
__global__ __launch_bounds__(256) void bn_act_maxpool3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                                                  int N, int IH, int IW, int OH, int OW, int C, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int act, float slope, float* __restrict__ amax,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float rad,
                                                                  float* __restrict__ yp, float* __restrict__ p_amax) {
    float S = 1.f, L = 0.f;
    if constexpr (TWIN) { S = p16_fwd_scale(gamma, beta, C, rad, p_amax); L = f16_clamp_for_scale(S); }
    const unsigned c4n = (unsigned)(C / 4);
    const unsigned total = (unsigned)N * OH * OW * c4n;
    float mx = 0.f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned c4 = i % c4n, pix = i / c4n;
        const unsigned ox = pix % (unsigned)OW, r = pix / (unsigned)OW;
        const unsigned oy = r % (unsigned)OH, n = r / (unsigned)OH;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4), sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        f32x4 v[9]; bool ok[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int iy = 2 * (int)oy - 1 + a, ix = 2 * (int)ox - 1 + b;
                ok[a * 3 + b] = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
                const int iyc = min(max(iy, 0), IH - 1), ixc = min(max(ix, 0), IW - 1);
                v[a * 3 + b] = *reinterpret_cast<const f32x4*>(x + (((size_t)n * IH + iyc) * IW + ixc) * C + c4 * 4);
            }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = viai_act(v[t][e] * sc[e] + sh[e], act, slope);
                if (ok[t] && u > best[e]) { best[e] = u; bi[e] = t; }
            }
        *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = best;
        *reinterpret_cast<unsigned*>(idx + (size_t)i * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        if constexpr (TWIN) p16_store_quad(yp + (size_t)pix * C, (int)c4, best, S, L);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(best[0]), fabsf(best[1]))), fmaxf(fabsf(best[2]), fabsf(best[3])));
    }
    if (amax != nullptr) block_absmax_to(amax, mx);
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx,
                                                          int N, int IH, int IW, int OH, int OW, int C, int k, int st, int pd) {
    const int c4n = C / 4;
    const long total = (long)N * IH * IW * c4n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c4 = (int)(i % c4n); long r = i / c4n;
        int ix = (int)(r % IW); r /= IW;
        int iy = (int)(r % IH); int n = (int)(r / IH);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // windows (oy, ox) that contain (iy, ix): oy*st - pd <= iy <= oy*st - pd + k - 1
        int oy_lo = (iy + pd - k + 1 + st - 1) / st; if (iy + pd - k + 1 < 0) oy_lo = 0;
        int ox_lo = (ix + pd - k + 1 + st - 1) / st; if (ix + pd - k + 1 < 0) ox_lo = 0;
        int oy_hi = min((iy + pd) / st, OH - 1), ox_hi = min((ix + pd) / st, OW - 1);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                unsigned want = (unsigned)((iy - (oy * st - pd)) * k + (ix - (ox * st - pd)));
                size_t o = (((size_t)n * OH + oy) * OW + ox) * c4n + c4;
                unsigned pk = *reinterpret_cast<const unsigned*>(idx + o * 4);
                f32x4 g = *reinterpret_cast<const f32x4*>(dy + o * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (((pk >> (8 * e)) & 0xffu) == want) acc[e] += g[e];
            }
        *reinterpret_cast<f32x4*>(dx + (size_t)i * 4) = acc;
    }
}

// nn.AvgPool2d(7) on a 7x7 map == mean over HW (networks/Image_Embedding.py:27,66): y[n][c] = mean_p x[n][p][c]
__global__ __launch_bounds__(256) void avgpool_hw_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int P, int C) {
    const long total = (long)N * C;
    const float inv = 1.f / (float)P;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c = (int)(i % C); int n = (int)(i / C);
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += x[((size_t)n * P + p) * C + c];
        y[i] = s * inv;
    }
}

__global__ __launch_bounds__(256) void avgpool_hw_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int P, int C) {
    const long total = (long)N * P * C;
    const float inv = 1.f / (float)P;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c = (int)(i % C); int n = (int)(i / ((long)P * C));
        dx[i] = dy[(size_t)n * C + c] * inv;
    }
}

// residual join of BasicBlock: out = relu(a + b)  (networks/ResNet.py:51-52); backward: da = db = dout * (out > 0)
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ o, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        f32x4 v = a[i] + b[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        o[i] = v;
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ o, f32x4* __restrict__ d, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        f32x4 gv = g[i], ov = o[i], r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = ov[e] > 0.f ? gv[e] : 0.f;
        d[i] = r;
    }
}

// frames (N, C, H, W) NCHW -> (N, H, W, 4) NHWC with zero-padded channels (C <= 4): the layout conv1 of the
// ResNets consumes (Image_Embedding.py:188-189 view(-1, 3|2, 224, 224))
// amax (optional, one zero-initialised float): receives max |x| -- the operand scale of the stem conv's f16x2 kernels (a pass of its own over the 0.8 GB of
// 1024 frames otherwise: viai_absmax)
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, f32x4* __restrict__ y, long N, int C, long HW, float* __restrict__ amax) {
    const long total = N * HW;
    float mx = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long n = i / HW, p = i % HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) { v[c] = x[(n * C + c) * HW + p]; mx = fmaxf(mx, fabsf(v[c])); }
        y[i] = v;
    }
    if (amax != nullptr) block_absmax_to(amax, mx);
}

// F.avg_pool2d(x, k, s, p, count_include_pad=False) on NHWC: the pix2pixHD input pyramid of a multi-scale
// discriminator (BASELINE cfg 4; composition of reference MelDiscriminators)
__global__ __launch_bounds__(256) void avgpool2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int N, int IH, int IW, int OH, int OW, int C, int k, int st, int pd) {
    const long total = (long)N * OH * OW * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c = (int)(i % C); long r = i / C;
        int ox = (int)(r % OW); r /= OW;
        int oy = (int)(r % OH); int n = (int)(r / OH);
        float s = 0.f; int cnt = 0;
        for (int a = 0; a < k; ++a) {
            int iy = oy * st - pd + a;
            if ((unsigned)iy >= (unsigned)IH) continue;
            for (int b = 0; b < k; ++b) {
                int ix = ox * st - pd + b;
                if ((unsigned)ix >= (unsigned)IW) continue;
                s += x[(((size_t)n * IH + iy) * IW + ix) * C + c]; ++cnt;
            }
        }
        y[i] = s / (float)cnt;
    }
}

__global__ __launch_bounds__(256) void avgpool2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            int N, int IH, int IW, int OH, int OW, int C, int k, int st, int pd) {
    const long total = (long)N * IH * IW * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int c = (int)(i % C); long r = i / C;
        int ix = (int)(r % IW); r /= IW;
        int iy = (int)(r % IH); int n = (int)(r / IH);
        float acc = 0.f;
        for (int oy = max(0, (iy + pd - k + st) / st); oy <= min(OH - 1, (iy + pd) / st); ++oy)
            for (int ox = max(0, (ix + pd - k + st) / st); ox <= min(OW - 1, (ix + pd) / st); ++ox) {
                int y0 = max(oy * st - pd, 0), y1 = min(oy * st - pd + k, IH), x0 = max(ox * st - pd, 0), x1 = min(ox * st - pd + k, IW);
                acc += dy[(((size_t)n * OH + oy) * OW + ox) * C + c] / (float)((y1 - y0) * (x1 - x0));
            }
        dx[i] = acc;
    }
}

inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int viai_bilinear_ac_fwd(const float* x, float* y, int N, int IH, int IW, int OH, int OW, int C, void* stream) {
    if (C % 4 != 0 || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return (int)hipErrorInvalidValue;
    long total = (long)N * OH * OW * (C / 4);
    const int c4sh = px_shift(C, (long)N * OH * OW);
    if (c4sh >= 0) VIAI_LAUNCH(bilinear_fwd_px_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, IH, IW, OH, OW, C, c4sh);
    else VIAI_LAUNCH(bilinear_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, IH, IW, OH, OW, C);
    return viai_launch_status();
}

// out (N, OH, OW, C) = bilinear(act(scale * y + shift)), y (N, IH, IW, C); *z_amax (optional, zero-initialised) receives max |act(..)| over
// the taps read (= over the whole map when OH >= IH and OW >= IW).  C / 4 a power of two <= 256, N * OH * OW < 2^24 (else invalid value)
extern "C" int viai_bn_act_bilinear_fwd_amax(const float* y, const float* scale, const float* shift, float* out, int N, int IH, int IW,
                                             int OH, int OW, int C, int act, float slope, float* z_amax, void* stream) {
    if (C % 4 != 0 || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return (int)hipErrorInvalidValue;
    const int c4sh = px_shift(C, (long)N * OH * OW);
    if (c4sh < 0) return (int)hipErrorInvalidValue;
    const long ppb = 1024 >> c4sh;
    long blocks = ((long)N * OH * OW + ppb - 1) / ppb;
    if (blocks > 256) blocks = 256;
    const dim3 g((unsigned)blocks), b(1024);
    hipStream_t st = (hipStream_t)stream;
    switch (act) {
    case VIAI_ACT_RELU: VIAI_LAUNCH(bn_act_bilinear_fwd_kernel<VIAI_ACT_RELU>, g, b, 0, st, y, scale, shift, out, N, IH, IW, OH, OW, C, c4sh, slope, z_amax, (const float*)nullptr, (const float*)nullptr, 0.f); break;
    case VIAI_ACT_LRELU: VIAI_LAUNCH(bn_act_bilinear_fwd_kernel<VIAI_ACT_LRELU>, g, b, 0, st, y, scale, shift, out, N, IH, IW, OH, OW, C, c4sh, slope, z_amax, (const float*)nullptr, (const float*)nullptr, 0.f); break;
    case VIAI_ACT_NONE: VIAI_LAUNCH(bn_act_bilinear_fwd_kernel<VIAI_ACT_NONE>, g, b, 0, st, y, scale, shift, out, N, IH, IW, OH, OW, C, c4sh, slope, z_amax, (const float*)nullptr, (const float*)nullptr, 0.f); break;
    default: return (int)hipErrorInvalidValue;
    }
    return viai_launch_status();
}

// (ABI 13) the same pass writing the resized tensor pre-split (P16); gamma / beta / m_stat give the bound stored in *z_amax.  C % 32 == 0
extern "C" int viai_bn_act_bilinear_fwd_p16(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                            float* out, int N, int IH, int IW, int OH, int OW, int C, int act, float slope, float* z_amax, void* stream) {
    if (C % 32 != 0 || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || z_amax == nullptr || m_stat < 1) return (int)hipErrorInvalidValue;
    const int c4sh = px_shift(C, (long)N * OH * OW);
    if (c4sh < 0) return (int)hipErrorInvalidValue;
    const long ppb = 1024 >> c4sh;
    long blocks = ((long)N * OH * OW + ppb - 1) / ppb;
    if (blocks > 512) blocks = 512;
    const dim3 g((unsigned)blocks), b(1024);
    hipStream_t st = (hipStream_t)stream;
    const float rad = sqrtf((float)(m_stat > 1 ? m_stat - 1 : 1));
    switch (act) {
    case VIAI_ACT_RELU: VIAI_LAUNCH((bn_act_bilinear_fwd_kernel<VIAI_ACT_RELU, true>), g, b, 0, st, y, scale, shift, out, N, IH, IW, OH, OW, C, c4sh, slope, z_amax, gamma, beta, rad); break;
    case VIAI_ACT_LRELU: VIAI_LAUNCH((bn_act_bilinear_fwd_kernel<VIAI_ACT_LRELU, true>), g, b, 0, st, y, scale, shift, out, N, IH, IW, OH, OW, C, c4sh, slope, z_amax, gamma, beta, rad); break;
    case VIAI_ACT_NONE: VIAI_LAUNCH((bn_act_bilinear_fwd_kernel<VIAI_ACT_NONE, true>), g, b, 0, st, y, scale, shift, out, N, IH, IW, OH, OW, C, c4sh, slope, z_amax, gamma, beta, rad); break;
    default: return (int)hipErrorInvalidValue;
    }
    return viai_launch_status();
}

extern "C" int viai_bilinear_ac_bwd(const float* dy, float* dx, int N, int IH, int IW, int OH, int OW, int C, void* stream) {
    if (C % 4 != 0 || N <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return (int)hipErrorInvalidValue;
    long total = (long)N * IH * IW * (C / 4);
    const int c4sh = px_shift(C, (long)N * IH * IW);
    if (c4sh >= 0) VIAI_LAUNCH(bilinear_bwd_px_kernel<6>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, IH, IW, OH, OW, C, c4sh);
    else VIAI_LAUNCH(bilinear_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, IH, IW, OH, OW, C);
    return viai_launch_status();
}

extern "C" int viai_avgpool_h_fwd(const float* x, float* y, int N, int IH, int W, int C, int k, void* stream) {
    if (C % 4 != 0 || k <= 0 || IH / k <= 0) return (int)hipErrorInvalidValue;
    long total = (long)N * (IH / k) * W * (C / 4);
    VIAI_LAUNCH(avgpool_h_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, IH, W, C, k);
    return viai_launch_status();
}

extern "C" int viai_avgpool_h_bwd(const float* dy, float* dx, int N, int IH, int W, int C, int k, void* stream) {
    if (C % 4 != 0 || k <= 0 || IH / k <= 0) return (int)hipErrorInvalidValue;
    long total = (long)N * IH * W * (C / 4);
    VIAI_LAUNCH(avgpool_h_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, IH, W, C, k);
    return viai_launch_status();
}

extern "C" int viai_maxpool_fwd(const float* x, float* y, unsigned char* idx, int N, int IH, int IW, int C, int k, int s, int p, void* stream) {
    if (C % 4 != 0 || k * k > 255) return (int)hipErrorInvalidValue;
    int OH = (IH + 2 * p - k) / s + 1, OW = (IW + 2 * p - k) / s + 1;
    long total = (long)N * OH * OW * (C / 4);
    VIAI_LAUNCH(maxpool_fwd_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, idx, N, IH, IW, OH, OW, C, k, s, p,
                (const float*)nullptr, (const float*)nullptr, 0, 0.f, (float*)nullptr);
    return viai_launch_status();
}

// out = maxpool(act(scale * y + shift)) with the argmax bytes; act = ReLU or none (max |out| -> out_amax, optional)
extern "C" int viai_bn_act_maxpool_fwd(const float* y, const float* scale, const float* shift, float* out, unsigned char* idx,
                                       int N, int IH, int IW, int C, int k, int s, int p, int act, float slope, float* out_amax, void* stream) {
    if (C % 4 != 0 || k * k > 255 || (act != VIAI_ACT_RELU && act != VIAI_ACT_NONE)) return (int)hipErrorInvalidValue;
    int OH = (IH + 2 * p - k) / s + 1, OW = (IW + 2 * p - k) / s + 1;
    long total = (long)N * OH * OW * (C / 4);
    if (k == 3 && s == 2 && p == 1 && total < (1L << 31) - (1L << 24)) {
        VIAI_LAUNCH(bn_act_maxpool3_fwd_kernel<false>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y, out, idx, N, IH, IW, OH, OW, C, scale, shift, act, slope,
                    out_amax, (const float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr);
        return viai_launch_status();
    }
    VIAI_LAUNCH(maxpool_fwd_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y, out, idx, N, IH, IW, OH, OW, C, k, s, p,
                scale, shift, act, slope, out_amax);
    return viai_launch_status();
}

// viai_bn_act_maxpool_fwd for the 3 x 3 / stride 2 / pad 1 pool with a second, pre-split (P16) copy of the pooled tensor (C % 32 == 0) whose scale comes from
// the bound |gamma| sqrt(m_stat - 1) + |beta| (written to *p_amax): the first BasicBlock's conv1 stages its pieces, the fp32 copy stays the first join's residual.
extern "C" int viai_bn_act_maxpool_fwd_twin(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                            float* out, float* out_p16, unsigned char* idx, int N, int IH, int IW, int C, int k, int s, int p, int act, float slope,
                                            float* out_amax, float* p_amax, void* stream) {
    if (C % 32 != 0 || k != 3 || s != 2 || p != 1 || (act != VIAI_ACT_RELU && act != VIAI_ACT_NONE) || gamma == nullptr || beta == nullptr || out_p16 == nullptr ||
        p_amax == nullptr)
        return (int)hipErrorInvalidValue;
    int OH = (IH + 2 * p - k) / s + 1, OW = (IW + 2 * p - k) / s + 1;
    long total = (long)N * OH * OW * (C / 4);
    if (total >= (1L << 31) - (1L << 24)) return (int)hipErrorInvalidValue;
    const float rad = sqrtf((float)(m_stat > 1 ? m_stat - 1 : 1));
    VIAI_LAUNCH(bn_act_maxpool3_fwd_kernel<true>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y, out, idx, N, IH, IW, OH, OW, C, scale, shift, act, slope,
                out_amax, gamma, beta, rad, out_p16, p_amax);
    return viai_launch_status();
}

extern "C" int viai_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int N, int IH, int IW, int C, int k, int s, int p, void* stream) {
    if (C % 4 != 0) return (int)hipErrorInvalidValue;
    int OH = (IH + 2 * p - k) / s + 1, OW = (IW + 2 * p - k) / s + 1;
    long total = (long)N * IH * IW * (C / 4);
    VIAI_LAUNCH(maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, N, IH, IW, OH, OW, C, k, s, p);
    return viai_launch_status();
}

extern "C" int viai_avgpool_hw_fwd(const float* x, float* y, int N, int P, int C, void* stream) {
    VIAI_LAUNCH(avgpool_hw_fwd_kernel, dim3(ew_blocks((long)N * C)), dim3(256), 0, (hipStream_t)stream, x, y, N, P, C);
    return viai_launch_status();
}

extern "C" int viai_avgpool_hw_bwd(const float* dy, float* dx, int N, int P, int C, void* stream) {
    VIAI_LAUNCH(avgpool_hw_bwd_kernel, dim3(ew_blocks((long)N * P * C)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, P, C);
    return viai_launch_status();
}

extern "C" int viai_add_relu_fwd(const float* a, const float* b, float* out, long n, void* stream) {
    if (n % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(add_relu_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(a),
                reinterpret_cast<const f32x4*>(b), reinterpret_cast<f32x4*>(out), n / 4);
    return viai_launch_status();
}

extern "C" int viai_relu_bwd(const float* g, const float* out, float* d, long n, void* stream) {
    if (n % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(relu_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(g),
                reinterpret_cast<const f32x4*>(out), reinterpret_cast<f32x4*>(d), n / 4);
    return viai_launch_status();
}

extern "C" int viai_nchw_to_nhwc4(const float* x, float* y, long N, int C, long HW, void* stream) {
    if (C < 1 || C > 4) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(nchw_to_nhwc4_kernel, dim3(ew_blocks(N * HW)), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<f32x4*>(y), N, C, HW, (float*)nullptr);
    return viai_launch_status();
}
// (ABI 15) ... and max |x| into *amax (zero-initialised)
extern "C" int viai_nchw_to_nhwc4_amax(const float* x, float* y, long N, int C, long HW, float* amax, void* stream) {
    if (C < 1 || C > 4 || amax == nullptr) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(nchw_to_nhwc4_kernel, dim3(ew_blocks(N * HW)), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<f32x4*>(y), N, C, HW, amax);
    return viai_launch_status();
}

extern "C" int viai_avgpool2d_fwd(const float* x, float* y, int N, int IH, int IW, int C, int k, int s, int p, void* stream) {
    int OH = (IH + 2 * p - k) / s + 1, OW = (IW + 2 * p - k) / s + 1;
    VIAI_LAUNCH(avgpool2d_fwd_kernel, dim3(ew_blocks((long)N * OH * OW * C)), dim3(256), 0, (hipStream_t)stream, x, y, N, IH, IW, OH, OW, C, k, s, p);
    return viai_launch_status();
}

extern "C" int viai_avgpool2d_bwd(const float* dy, float* dx, int N, int IH, int IW, int C, int k, int s, int p, void* stream) {
    int OH = (IH + 2 * p - k) / s + 1, OW = (IW + 2 * p - k) / s + 1;
    VIAI_LAUNCH(avgpool2d_bwd_kernel, dim3(ew_blocks((long)N * IH * IW * C)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, IH, IW, OH, OW, C, k, s, p);
    return viai_launch_status();
}
