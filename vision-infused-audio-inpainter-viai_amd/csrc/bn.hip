// Training-mode BatchNorm2d + activation on NHWC fp32 (gfx950), HBM-bound.
// Reference call sites: Inpainting_Networks.py:72-76, New_Inpainting_Networks.py:33-36,72-75,86,
// Discriminator_Networks.py:39-46 (nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1,
// biased variance for normalisation, unbiased for running_var).
//
// Forward statistics arrive as block-local (mean_b, M2_b) pairs from the conv
// epilogue and are merged here with Chan's parallel formula in fp64, so a
// 1M-element channel never sees the E[x^2] - E[x]^2 cancellation.
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// NT threads per channel: 256, or 1024 where a channel has thousands of partial blocks (the ResNet branch on 1024 frames: 25 088 tiles per
// channel of layer1 -- one 256-thread block walked them in 98 dependent rounds, 46 us per layer and 73 such launches per step)
template <int NT>
__global__ __launch_bounds__(NT) void bn_finalize_kernel(
    const float* __restrict__ part, int nblk, int rows, long M, int C,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
    float* mean_o, float* invstd_o, float* scale_o, float* shift_o, int th, int tw, int OH, int OW) {
    __shared__ double red[NT / 64];
    const int c = blockIdx.x;
    const float* pm = part + (size_t)c * nblk;
    const float* p2 = part + (size_t)(C + c) * nblk;
    const long last_n = M - (long)(nblk - 1) * rows;
    // ONE pass over the partials: Chan's merge written around a pivot k = the first block's mean instead of the (not yet known) global
    // mean:  S = sum n_b (m_b - k),  Q = sum [M2_b + n_b (m_b - k)^2]  =>  mean = k + S / M,  M2 = Q - S^2 / M.
    // In fp64 with |m_b - k| of the order of the spread of the block means the subtraction Q - S^2/M loses nothing that matters
    // (tests/test_kernels_gpu.py::test_bn_large_mean_is_stable); the two-pass form cost two more block reductions (four barriers).
    // what the epilogue needs is fetched now, not after the reduction (these kernels are chains of memory round trips: 6 - 7 us each,
    // 33 of them on the step's critical path)
    float g = 1.f, bta = 0.f, rm = 0.f, rv = 0.f;
    if (threadIdx.x == 0) {
        if (gamma) g = gamma[c];
        if (beta) bta = beta[c];
        if (running_mean) rm = running_mean[c];
        if (running_var) rv = running_var[c];
    }
    const double k = (double)pm[0];
    double s = 0.0, q = 0.0;
    // th > 0: block b is the th x tw output tile (n, ty, tx) of an OH x OW map, clipped at the map's edge (viai_bn_finalize_tiles)
    const int tiles_x = th > 0 ? (OW + tw - 1) / tw : 1, tiles_y = th > 0 ? (OH + th - 1) / th : 1;
    auto count = [&](int b) -> double {
        if (th < 0) {                                       // partials merged per persistent block of the linear-tile conv kernel (viai_bn_finalize_lin): part b =
            const int blk = b / tw;                         // (block b / PW, pixel sub-block b % PW) holds 128 pixels of every item the block walked: items blk, blk + G, ...
            return 128.0 * (double)((OH - blk + OW - 1) / OW);          // (tw = PW, OH = items, OW = G = blocks)
        }
        if (th > 0) {
            const int tx = b % tiles_x, ty = (b / tiles_x) % tiles_y;
            return (double)(min(th, OH - ty * th) * min(tw, OW - tx * tw));
        }
        return (b == nblk - 1) ? (double)last_n : (double)rows;
    };
    int b = threadIdx.x;
    for (; b + 3 * NT < nblk; b += 4 * NT) {                // eight independent loads in flight per lane; the sums keep the order of the plain walk
        float m[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { m[u] = pm[b + u * NT]; v[u] = p2[b + u * NT]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double nb = count(b + u * NT), d = (double)m[u] - k;
            s += nb * d;
            q += (double)v[u] + nb * d * d;
        }
    }
    for (; b < nblk; b += NT) {
        const double nb = count(b), d = (double)pm[b] - k;
        s += nb * d;
        q += (double)p2[b] + nb * d * d;
    }
    // both sums through one pair of barriers
    s = wave_sum_d_dpp(s); q = wave_sum_d_dpp(q);
    __shared__ double red2[NT / 64];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red2[threadIdx.x >> 6] = q; }
    __syncthreads();
    if constexpr (NT == 256) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        q = (red2[0] + red2[1]) + (red2[2] + red2[3]);
    } else {
        s = 0.0; q = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { s += red[w]; q += red2[w]; }
    }
    const double mean = k + s / (double)M;
    const double m2 = fmax(q - s * s / (double)M, 0.0);
    if (threadIdx.x == 0) {
        const double var = m2 / (double)M;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        mean_o[c] = (float)mean;
        invstd_o[c] = invstd;
        scale_o[c] = g * invstd;
        shift_o[c] = bta - (float)mean * g * invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * rm + momentum * (float)mean;
        if (running_var) {
            double unb = (M > 1) ? m2 / (double)(M - 1) : var;
            running_var[c] = (1.f - momentum) * rv + momentum * (float)unb;
        }
        if (nbt && c == 0) *nbt += 1;
    }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* mean_o, float* invstd_o, float* scale_o, float* shift_o) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.f / sqrtf(rv[c] + eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean_o[c] = rm[c]; invstd_o[c] = invstd; scale_o[c] = g * invstd; shift_o[c] = b - rm[c] * g * invstd;
}

// z = act(scale*y + shift).  FIXED: 256 % (C/4) == 0 -> one channel quad per thread, coefficients loaded once (see
// bn_bwd_apply_kernel); four independent 16-byte loads in flight per thread.
// RES: z = act(scale*y + shift + res) -- the residual join of a ResNet block (networks/ResNet.py:49-53) in the same pass.
// NT threads per block: 256, or 1024 when the pass ends in the abs-max atomics (one fat block per CU: 8x fewer same-address atomics
// queued behind the kernel's last store, viai_common.h block_absmax_to)
template <int ACT, bool FIXED, bool RES = false, int NT = 256>
__global__ __launch_bounds__(NT) void bn_act_fwd_kernel(const f32x4* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, f32x4* __restrict__ z,
                                                         long n4, int C, float slope, float* __restrict__ amax, const f32x4* __restrict__ res = nullptr) {
    const unsigned c4n = (unsigned)(C / 4);
    const long stride = (long)gridDim.x * NT;
    long i = blockIdx.x * (long)NT + threadIdx.x;
    unsigned cq = (unsigned)((unsigned long)i % c4n);
    const unsigned cstep = (unsigned)((unsigned long)stride % c4n);
    f32x4 sc = *reinterpret_cast<const f32x4*>(scale + cq * 4), sh = *reinterpret_cast<const f32x4*>(shift + cq * 4);
    float mx = 0.f;
    auto one = [&](const f32x4& v, const f32x4& r = f32x4{0.f, 0.f, 0.f, 0.f}) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = viai_act(RES ? (v[e] * sc[e] + sh[e]) + r[e] : v[e] * sc[e] + sh[e], ACT, slope); mx = fmaxf(mx, fabsf(o[e])); }
        return o;
    };
    auto next = [&]() {
        if constexpr (!FIXED) {
            cq += cstep; if (cq >= c4n) cq -= c4n;
            sc = *reinterpret_cast<const f32x4*>(scale + cq * 4); sh = *reinterpret_cast<const f32x4*>(shift + cq * 4);
        }
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = y[i + u * stride]; if constexpr (RES) r[u] = res[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { z[i + u * stride] = RES ? one(v[u], r[u]) : one(v[u]); next(); }
    }
    for (; i < n4; i += stride) { z[i] = RES ? one(y[i], res[i]) : one(y[i]); next(); }
    // max |z|: the f16x2 operand scale of the kernels that consume z (forward conv of the next layer, this tensor's weight gradient)
    if (amax != nullptr) block_absmax_to(amax, mx);
}

// grid of a streaming pass over n4 float4 items: 256-thread blocks, at most 2048 of them -- or, when the pass ends in the abs-max
// atomics, 1024-thread blocks, at most one per CU
inline unsigned stream_grid(long n4, int nt) {
    long blocks = (n4 + nt - 1) / nt;
    const long cap = nt == 1024 ? 256 : 2048;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <int ACT, bool FIXED, bool RES>
int launch_bn_act_fwd_t(const float* y, const float* scale, const float* shift, const float* res, float* z, long n4, int C, float slope, float* amax, hipStream_t st) {
    auto a0 = reinterpret_cast<const f32x4*>(y); auto a1 = reinterpret_cast<f32x4*>(z); auto a2 = reinterpret_cast<const f32x4*>(res);
    if (amax != nullptr) VIAI_LAUNCH((bn_act_fwd_kernel<ACT, FIXED, RES, 1024>), dim3(stream_grid(n4, 1024)), dim3(1024), 0, st, a0, scale, shift, a1, n4, C, slope, amax, a2);
    else VIAI_LAUNCH((bn_act_fwd_kernel<ACT, FIXED, RES, 256>), dim3(stream_grid(n4, 256)), dim3(256), 0, st, a0, scale, shift, a1, n4, C, slope, amax, a2);
    return viai_launch_status();
}

template <bool FIXED>
int launch_bn_act_fwd(const float* y, const float* scale, const float* shift, float* z, long n4, int C, int act, float slope, float* amax, hipStream_t st) {
    switch (act) {
    case VIAI_ACT_RELU: return launch_bn_act_fwd_t<VIAI_ACT_RELU, FIXED, false>(y, scale, shift, nullptr, z, n4, C, slope, amax, st);
    case VIAI_ACT_LRELU: return launch_bn_act_fwd_t<VIAI_ACT_LRELU, FIXED, false>(y, scale, shift, nullptr, z, n4, C, slope, amax, st);
    case VIAI_ACT_SIGMOID: return launch_bn_act_fwd_t<VIAI_ACT_SIGMOID, FIXED, false>(y, scale, shift, nullptr, z, n4, C, slope, amax, st);
    default: return launch_bn_act_fwd_t<VIAI_ACT_NONE, FIXED, false>(y, scale, shift, nullptr, z, n4, C, slope, amax, st);
    }
}

__device__ __forceinline__ float act_grad(float pre, int act, float slope) {
    if (act == VIAI_ACT_SIGMOID) { float s = 1.f / (1.f + __expf(-pre)); return s * (1.f - s); }
    return viai_act_grad_pl(pre, act, slope);
}

// BatchNorm + activation followed by nn.MaxPool2d (ResNet stem, networks/Image_Embedding.py:20-23): the gradient of the post-activation
// tensor is a gather from the pooled gradient and the window argmax bytes (exactly maxpool_bwd_kernel's sum), formed where the BatchNorm
// backward loads it -- the 3.3 GB tensor per 1024 frames is neither written nor read back twice.
struct PoolGather {
    const float* dpool; const unsigned char* idx;
    const float* dpool2;          // second addend of the pooled gradient, or null (ops.fork2: the gradient of the stem's output reaches it as two addends)
    int IH, IW, OH, OW, k, st, pd;
};
// pixel (n, iy, ix) of the un-pooled map, channels c .. c + 3.  The kernels below walk the pixels with a constant stride and keep the
// three coordinates incrementally (PoolPos): the two 64-bit divisions per element of the first version made the pass instruction-bound
// (4.8 ms on 3.3 GB where the memory time is ~1 ms).
__device__ __forceinline__ f32x4 pool_dz(const PoolGather& p, int n, int iy, int ix, int c, int C) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = iy + p.pd - p.k + 1 < 0 ? 0 : (iy + p.pd - p.k + 1 + p.st - 1) / p.st;
    const int ox_lo = ix + p.pd - p.k + 1 < 0 ? 0 : (ix + p.pd - p.k + 1 + p.st - 1) / p.st;
    const int oy_hi = min((iy + p.pd) / p.st, p.OH - 1), ox_hi = min((ix + p.pd) / p.st, p.OW - 1);
    if (p.k <= 2 * p.st) {
        // at most 2 x 2 windows hold a pixel (k <= 2 s: the 3 x 3 / stride 2 pool of the stem): all four candidates are fetched up front
        // (clamped addresses, a validity bit each) so that eight loads are in flight per element instead of a chain of dependent pairs
        unsigned pk[4]; f32x4 g[4]; unsigned want[4]; bool ok[4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int oy = oy_lo + a, ox = ox_lo + b;
                ok[a * 2 + b] = oy <= oy_hi && ox <= ox_hi;
                const int oyc = min(oy, oy_hi), oxc = min(ox, ox_hi);
                want[a * 2 + b] = (unsigned)((iy - (oy * p.st - p.pd)) * p.k + (ix - (ox * p.st - p.pd)));
                const size_t o = (((size_t)n * p.OH + oyc) * p.OW + oxc) * C + c;
                pk[a * 2 + b] = *reinterpret_cast<const unsigned*>(p.idx + o);
                g[a * 2 + b] = *reinterpret_cast<const f32x4*>(p.dpool + o);
                if (p.dpool2 != nullptr) g[a * 2 + b] += *reinterpret_cast<const f32x4*>(p.dpool2 + o);
            }
#pragma unroll
        for (int w = 0; w < 4; ++w)                 // (same order as the loop below: oy outer, ox inner)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (ok[w] && ((pk[w] >> (8 * e)) & 0xffu) == want[w]) acc[e] += g[w][e];
        return acc;
    }
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const unsigned want = (unsigned)((iy - (oy * p.st - p.pd)) * p.k + (ix - (ox * p.st - p.pd)));
            const size_t o = (((size_t)n * p.OH + oy) * p.OW + ox) * C + c;
            const unsigned pk = *reinterpret_cast<const unsigned*>(p.idx + o);
            f32x4 g = *reinterpret_cast<const f32x4*>(p.dpool + o);
            if (p.dpool2 != nullptr) g += *reinterpret_cast<const f32x4*>(p.dpool2 + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (((pk >> (8 * e)) & 0xffu) == want) acc[e] += g[e];
        }
    return acc;
}
struct PoolPos {
    int n, iy, ix;
    __device__ __forceinline__ void set(const PoolGather& p, long row) {
        ix = (int)(row % p.IW); const long r = row / p.IW;
        iy = (int)(r % p.IH); n = (int)(r / p.IH);
    }
    // advance by (dn images, dy rows, dx pixels), dx < IW, dy < IH
    __device__ __forceinline__ void step(const PoolGather& p, int dn, int dy, int dx) {
        ix += dx; if (ix >= p.IW) { ix -= p.IW; ++iy; }
        iy += dy; if (iy >= p.IH) { iy -= p.IH; ++n; }
        n += dn;
    }
};

// partial sums over a row range: part[blk][0][c] = sum dpre, part[blk][1][c] = sum dpre * xhat
// MAXDP: a third partial per channel, max |dpre| over the block's rows (part[blk][2][c]): what the pre-split (P16) apply pass bounds |dy| with
// JOIN (round 5): the BatchNorm in front of a residual join (networks/ResNet.py:46-53).  The gradient of the join's output arrives as one or two addends
// (dz, dz2 or null: ops.fork2) and goes through the join's ReLU (mask from its saved output zj) before it reaches this BatchNorm, whose own activation is
// none: the masked sum is formed on load, written out once (dres: it is also the residual branch's gradient, and the apply pass reads it) and reduced in
// the same pass -- the elementwise pass that made it and the re-read of it are gone.
template <bool POOL = false, bool MAXDP = false, bool JOIN = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dz, const float* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ part, long M, int C, long rows_per_blk, int act, float slope, const PoolGather pg_ = PoolGather{},
    const float* __restrict__ dz2 = nullptr, const float* __restrict__ zj = nullptr, float* __restrict__ dres = nullptr) {
    __shared__ f32x4 r1[256], r2[256];
    __shared__ f32x4 r3[MAXDP ? 256 : 1];
    constexpr int PS = MAXDP ? 3 : 2;
    const int tid = threadIdx.x;
    const int CG = C / 4;
    const long row0 = blockIdx.x * rows_per_blk;
    long row1 = row0 + rows_per_blk; if (row1 > M) row1 = M;
    for (int g0 = 0; g0 < CG; g0 += 256) {
        const int cgw = min(256, CG - g0);
        const int pg = 256 / cgw;
        const int cg = tid % cgw, pl = tid / cgw;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
        if (pl < pg) {
            const int c = (g0 + cg) * 4;
            f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
            f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
            PoolPos pp{0, 0, 0};
            int sy = 0, sx = 0;                          // the row stride pg as (rows, pixels) of the un-pooled map
            if constexpr (POOL) { pp.set(pg_, row0 + pl); sy = (pg / pg_.IW) % pg_.IH; sx = pg % pg_.IW; }
            const int sn = POOL ? pg / (pg_.IW * pg_.IH) : 0;
            for (long r = row0 + pl; r < row1; r += 4L * pg) {
                f32x4 g[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {           // 8 independent 16-byte loads in flight per lane
                    long rr = r + (long)u * pg;
                    bool ok = rr < row1;
                    if constexpr (POOL) { g[u] = ok ? pool_dz(pg_, pp.n, pp.iy, pp.ix, c, C) : f32x4{0.f, 0.f, 0.f, 0.f}; pp.step(pg_, sn, sy, sx); }
                    else g[u] = ok ? *reinterpret_cast<const f32x4*>(dz + rr * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    v[u] = ok ? *reinterpret_cast<const f32x4*>(y + rr * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (JOIN) {
                        if (ok) {
                            const f32x4 zz = *reinterpret_cast<const f32x4*>(zj + rr * C + c);
                            if (dz2 != nullptr) g[u] += *reinterpret_cast<const f32x4*>(dz2 + rr * C + c);      // the same single fp32 addition autograd would have made
#pragma unroll
                            for (int e = 0; e < 4; ++e) g[u][e] = g[u][e] * (zz[e] > 0.f ? 1.f : 0.f);           // act_bwd_out_kernel's product, ReLU
                            *reinterpret_cast<f32x4*>(dres + rr * C + c) = g[u];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float dp = g[u][e] * act_grad(v[u][e] * sc[e] + sh[e], act, slope);
                        s1[e] += dp;
                        s2[e] += dp * (v[u][e] - mu[e]) * is[e];
                        if constexpr (MAXDP) s3[e] = fmaxf(s3[e], fabsf(dp));
                    }
            }
        }
        r1[tid] = s1; r2[tid] = s2;
        if constexpr (MAXDP) r3[tid] = s3;
        __syncthreads();
        if (tid < cgw) {
            f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f}, t3 = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < pg; ++k) {
                t1 += r1[k * cgw + tid]; t2 += r2[k * cgw + tid];
                if constexpr (MAXDP) { const f32x4 m = r3[k * cgw + tid]; for (int e = 0; e < 4; ++e) t3[e] = fmaxf(t3[e], m[e]); }
            }
            const int c = (g0 + tid) * 4;
            *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * PS + 0) * C + c) = t1;
            *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * PS + 1) * C + c) = t2;
            if constexpr (MAXDP) *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * PS + 2) * C + c) = t3;
        }
        __syncthreads();
    }
}

// The pool-fused reduce from the POOLED side (round 5).  The two sums are linear in the un-pooled gradient, and that gradient is a scatter of the pooled
// one: sum over pixels of dz(pixel) f(pixel) = sum over pooled elements of dpool f(its argmax pixel).  So the pass walks the pooled tensor (a quarter of
// the pixels for the stem's 3 x 3 / stride-2 pool) and fetches y at the argmax position of each element -- one 4-byte read per element -- instead of
// walking the un-pooled map and testing up to four windows per pixel (16 GB of L1 / L2 traffic and the compares on the ResNet stem's 3.3 GB map:
// 2.27 ms where the memory time is 0.8).  Same terms as bn_bwd_reduce_kernel<true>, summed in another order (an un-pooled pixel that is the argmax of
// two windows contributes twice instead of once with the summed gradient): results agree to rounding, not bit for bit.  k = 3 only (the stem).
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(
    const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ part, long MP, int C, long rows_per_blk, int act, float slope, const PoolGather p) {
    __shared__ f32x4 r1[256], r2[256];
    const int tid = threadIdx.x;
    const int CG = C / 4;
    const long row0 = blockIdx.x * rows_per_blk;                 // pooled pixels of this block
    long row1 = row0 + rows_per_blk; if (row1 > MP) row1 = MP;
    for (int g0 = 0; g0 < CG; g0 += 256) {
        const int cgw = min(256, CG - g0);
        const int pg = 256 / cgw;
        const int cg = tid % cgw, pl = tid / cgw;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (pl < pg) {
            const int c = (g0 + cg) * 4;
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
            // pooled pixel r = (n, oy, ox), advanced by pg pixels per step (pg <= 256 < OW * OH is not assumed: general carry)
            long r = row0 + pl;
            int ox = (int)(r % p.OW); long t = r / p.OW; int oy = (int)(t % p.OH); int n = (int)(t / p.OH);
            const int sx = pg % p.OW, sy = (pg / p.OW) % p.OH, sn = pg / (p.OW * p.OH);
            for (; r < row1; r += 2L * pg) {
                unsigned pk[2]; f32x4 g[2]; float v[2][4]; bool ok[2];
                int bo[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {                    // two pooled elements x (1 + 1 + 4) loads in flight per lane
                    ok[u] = r + (long)u * pg < row1;
                    const size_t o = ok[u] ? (size_t)(r + (long)u * pg) * C + c : (size_t)row0 * C + c;
                    pk[u] = *reinterpret_cast<const unsigned*>(p.idx + o);
                    g[u] = *reinterpret_cast<const f32x4*>(p.dpool + o);
                    if (p.dpool2 != nullptr) g[u] += *reinterpret_cast<const f32x4*>(p.dpool2 + o);
                    // window origin (may lie in the padding: the argmax never does); a lane past the block's range reads element row0's window position of pixel 0
                    bo[u] = ok[u] ? ((n * p.IH + oy * p.st - p.pd) * p.IW + ox * p.st - p.pd) * C + c : c;
                    ox += sx; if (ox >= p.OW) { ox -= p.OW; ++oy; }
                    oy += sy; if (oy >= p.OH) { oy -= p.OH; ++n; }
                    n += sn;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int b = (int)((pk[u] >> (8 * e)) & 0xffu);
                        const int wy = (b * 11) >> 5, wx = b - 3 * wy;   // b / 3, b % 3 (b < 9)
                        v[u][e] = y[(long)bo[u] + (long)(wy * p.IW + wx) * C + e];
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dp = ok[u] ? g[u][e] * act_grad(v[u][e] * sc[e] + sh[e], act, slope) : 0.f;
                        s1[e] += dp;
                        s2[e] += dp * (v[u][e] - mu[e]) * is[e];
                    }
            }
        }
        r1[tid] = s1; r2[tid] = s2;
        __syncthreads();
        if (tid < cgw) {
            f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < pg; ++k) { t1 += r1[k * cgw + tid]; t2 += r2[k * cgw + tid]; }
            const int c = (g0 + tid) * 4;
            *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + c) = t1;
            *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + c) = t2;
        }
        __syncthreads();
    }
}

// sums over the row-block partials (fp64, fixed order), 4 channels x 64 partial lanes per block.  Emits
//   sums[0][c] = k0, sums[1][c] = k1  with  dy = scale*dpre + k1*(y-mean) + k0   (training-mode BN backward:
//   dy = scale*(dpre - s1/M - xhat*s2/M), xhat = (y-mean)*invstd), plus dgamma = s2, dbeta = s1.
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ part, int nblk, int C, long M,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, int training,
                                                           float* sums, float* dgamma, float* dbeta, int accumulate, int ps) {
    __shared__ double r1[4][4], r2[4][4];
    const int cl = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    double s1 = 0.0, s2 = 0.0;
    float mxdp = 0.f;
    float p_sc = 0.f, p_is = 0.f, p_db = 0.f, p_dg = 0.f;        // the epilogue's operands, requested before the reduction
    if (pl == 0 && c < C) {
        if (training || ps == 3) { p_sc = scale[c]; p_is = invstd[c]; }
        if (accumulate) { if (dbeta) p_db = dbeta[c]; if (dgamma) p_dg = dgamma[c]; }
    }
    if (c < C) {
        int b = pl;
        for (; b + 192 < nblk; b += 256) {               // 8 independent loads in flight per lane
            float a0 = part[((size_t)b * ps + 0) * C + c], b0 = part[((size_t)b * ps + 1) * C + c];
            float a1 = part[((size_t)(b + 64) * ps + 0) * C + c], b1 = part[((size_t)(b + 64) * ps + 1) * C + c];
            float a2 = part[((size_t)(b + 128) * ps + 0) * C + c], b2 = part[((size_t)(b + 128) * ps + 1) * C + c];
            float a3 = part[((size_t)(b + 192) * ps + 0) * C + c], b3 = part[((size_t)(b + 192) * ps + 1) * C + c];
            s1 += (double)a0; s1 += (double)a1; s1 += (double)a2; s1 += (double)a3;
            s2 += (double)b0; s2 += (double)b1; s2 += (double)b2; s2 += (double)b3;
            if (ps == 3) for (int u = 0; u < 4; ++u) mxdp = fmaxf(mxdp, part[((size_t)(b + 64 * u) * 3 + 2) * C + c]);
        }
        for (; b < nblk; b += 64) {
            s1 += (double)part[((size_t)b * ps + 0) * C + c];
            s2 += (double)part[((size_t)b * ps + 1) * C + c];
            if (ps == 3) mxdp = fmaxf(mxdp, part[((size_t)b * 3 + 2) * C + c]);
        }
    }
    // fixed-order tree: across the 16 partial lanes of a wave by shuffles, then across the four waves through LDS
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    __shared__ float r3[4][4];
    if (ps == 3) { for (int o = 4; o < 64; o <<= 1) mxdp = fmaxf(mxdp, __shfl_xor(mxdp, o, 64)); }
    if ((threadIdx.x & 63) < 4) { r1[threadIdx.x >> 6][cl] = s1; r2[threadIdx.x >> 6][cl] = s2; r3[threadIdx.x >> 6][cl] = mxdp; }
    __syncthreads();
    if (pl == 0 && c < C) {
        s1 = (r1[0][cl] + r1[1][cl]) + (r1[2][cl] + r1[3][cl]);
        s2 = (r2[0][cl] + r2[1][cl]) + (r2[2][cl] + r2[3][cl]);
        if (ps == 3) {
            // |dy| <= |scale| max|dpre| + |k1| max|y - mean| + |k0| with max |y - mean| <= sqrt(M - 1) / invstd (Samuelson: no sample lies more than
            // sqrt(M - 1) standard deviations from the mean; invstd^-2 = var + eps >= var): the magnitude the P16 apply pass scales dy by
            mxdp = fmaxf(fmaxf(r3[0][cl], r3[1][cl]), fmaxf(r3[2][cl], r3[3][cl]));
            const double sc_ = fabs((double)p_sc);
            double bnd = sc_ * (double)mxdp;
            if (training) bnd += sc_ * fabs(s2) * sqrt((double)(M > 1 ? M - 1 : 1)) / (double)M + sc_ * fabs(s1) / (double)M;
            sums[2 * C + c] = (float)(bnd * 1.001);
        }
        if (dbeta) dbeta[c] = accumulate ? p_db + (float)s1 : (float)s1;
        if (dgamma) dgamma[c] = accumulate ? p_dg + (float)s2 : (float)s2;
        if (training) {
            double sc = (double)p_sc;
            sums[C + c] = (float)(-sc * s2 * (double)p_is / (double)M);     // k1
            sums[c] = (float)(-sc * s1 / (double)M);                              // k0
        } else {
            sums[c] = 0.f; sums[C + c] = 0.f;
        }
    }
}

// dy = scale*dpre + k1*(y-mean) + k0, and max |dy| for the f16x2 consumers.  FIXED: 256 % (C/4) == 0, so a thread keeps one
// channel quad for its whole grid-stride walk and the five coefficient vectors are loaded once; otherwise the quad index
// advances by (stride mod C/4) with one conditional subtract.  Four independent element pairs are in flight per thread (the
// first version recomputed a 64-bit modulo and reloaded 80 bytes of coefficients per 32 bytes of data: 2.0 TB/s).
template <int ACT, bool FIXED, bool POOL = false, int NT = 256>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(
    const f32x4* __restrict__ dz, const f32x4* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ sums, f32x4* __restrict__ dy,
    long n4, int C, float slope, float* __restrict__ amax, const PoolGather pg_ = PoolGather{}) {
    const unsigned c4n = (unsigned)(C / 4);
    const long stride = (long)gridDim.x * NT;
    long i = blockIdx.x * (long)NT + threadIdx.x;
    unsigned cq = (unsigned)((unsigned long)i % c4n);
    const unsigned cstep = (unsigned)((unsigned long)stride % c4n);
    f32x4 sc, sh, k0, k1, mu;
    auto coeffs = [&](unsigned q) {
        const int c = (int)q * 4;
        sc = *reinterpret_cast<const f32x4*>(scale + c); sh = *reinterpret_cast<const f32x4*>(shift + c);
        k0 = *reinterpret_cast<const f32x4*>(sums + c); k1 = *reinterpret_cast<const f32x4*>(sums + C + c);
        mu = *reinterpret_cast<const f32x4*>(mean + c);
    };
    coeffs(cq);
    float mx = 0.f;
    auto one = [&](const f32x4& g, const f32x4& v) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float dp = g[e] * act_grad(v[e] * sc[e] + sh[e], ACT, slope);
            // (y - mean) first: keeps the fp32 rounding error relative to the centred value
            o[e] = sc[e] * dp + (k1[e] * (v[e] - mu[e]) + k0[e]);
            mx = fmaxf(mx, fabsf(o[e]));
        }
        return o;
    };
    auto next = [&]() {
        if constexpr (!FIXED) { cq += cstep; if (cq >= c4n) cq -= c4n; coeffs(cq); }
    };
    // POOL: the pixel of float4 index i, kept incrementally; the walk advances by `stride` float4s = (stride / c4n) pixels + (stride % c4n) quads
    PoolPos pp{0, 0, 0};
    int pq = 0, sn = 0, sy = 0, sx = 0, sq = 0;
    if constexpr (POOL) {
        pp.set(pg_, i / c4n); pq = (int)(i % c4n);
        const long srow = stride / c4n; sq = (int)(stride % c4n);
        sx = (int)(srow % pg_.IW); sy = (int)((srow / pg_.IW) % pg_.IH); sn = (int)(srow / ((long)pg_.IW * pg_.IH));
    }
    auto grad = [&](long q) -> f32x4 {                 // gradient of the post-activation tensor at float4 index q (POOL: q is the walk's current index)
        if constexpr (POOL) {
            const f32x4 gq = pool_dz(pg_, pp.n, pp.iy, pp.ix, pq * 4, C);
            pq += sq;
            int carry = 0;
            if (pq >= (int)c4n) { pq -= c4n; carry = 1; }
            pp.step(pg_, sn, sy, sx);
            if (carry) pp.step(pg_, 0, 0, 1);
            return gq;
        } else return dz[q];
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 g[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { g[u] = grad(i + u * stride); v[u] = y[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { dy[i + u * stride] = one(g[u], v[u]); next(); }
    }
    for (; i < n4; i += stride) { dy[i] = one(grad(i), y[i]); next(); }
    // max |dy| of the tensor: the consumers' f16x2 operand scale (order-independent, deterministic)
    if (amax != nullptr) block_absmax_to(amax, mx);
}

// The pool-fused apply pass for the stem's 3 x 3 / stride-2 / pad-1 pool on an even map, by 2 x 2 PIXEL BLOCKS (round 5).  The four pixels (2a + dy, 2b + dx)
// lie in the windows (a, b), (a, b + 1), (a + 1, b), (a + 1, b + 1) only -- 1, 2, 2 and 4 of them -- so a thread that owns the block (and a channel quad)
// loads those four pooled elements ONCE and hands each to the pixels it may belong to; the per-pixel kernel fetched four candidates for every pixel
// (16 pooled loads per block instead of 4).  Same sums in the same order (windows by rows, then columns), same expression: bit-identical.
template <int ACT>
__global__ __launch_bounds__(256) void bn_pool2x2_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ sums, float* __restrict__ dy,
                                                                   long nblk4, int C, float slope, float* __restrict__ amax, const PoolGather p) {
    const int c4n = C / 4, BW = p.IW / 2, BH = p.IH / 2;
    float mx = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nblk4; i += (long)gridDim.x * 256) {
        const int cq = (int)(i % c4n); long t = i / c4n;
        const int b = (int)(t % BW); t /= BW;
        const int a = (int)(t % BH); const int n = (int)(t / BH);
        const int c = cq * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(sums + c), k1 = *reinterpret_cast<const f32x4*>(sums + C + c), mu = *reinterpret_cast<const f32x4*>(mean + c);
        // the four windows (wa, wb) = (a + r, b + s): argmax bytes and gradients (a window past the pooled map: no contribution)
        unsigned pk[4]; f32x4 g[4]; bool ok[4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int wa = a + r, wb = b + q;
                ok[r * 2 + q] = wa < p.OH && wb < p.OW;
                const size_t o = (((size_t)n * p.OH + (ok[r * 2 + q] ? wa : a)) * p.OW + (ok[r * 2 + q] ? wb : b)) * C + c;
                pk[r * 2 + q] = *reinterpret_cast<const unsigned*>(p.idx + o);
                g[r * 2 + q] = *reinterpret_cast<const f32x4*>(p.dpool + o);
                if (p.dpool2 != nullptr) g[r * 2 + q] += *reinterpret_cast<const f32x4*>(p.dpool2 + o);
            }
        f32x4 v[4];
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
            for (int dx_ = 0; dx_ < 2; ++dx_) v[dy_ * 2 + dx_] = *reinterpret_cast<const f32x4*>(y + (((size_t)n * p.IH + 2 * a + dy_) * p.IW + 2 * b + dx_) * C + c);
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
            for (int dx_ = 0; dx_ < 2; ++dx_) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // windows (a + r, b + q) that hold pixel (2a + dy_, 2b + dx_): r <= dy_, q <= dx_; its position in that window: row dy_ + 1 - 2 r, column dx_ + 1 - 2 q
#pragma unroll
                for (int r = 0; r <= dy_; ++r)
#pragma unroll
                    for (int q = 0; q <= dx_; ++q) {
                        const unsigned want = (unsigned)((dy_ + 1 - 2 * r) * 3 + (dx_ + 1 - 2 * q));
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (ok[r * 2 + q] && ((pk[r * 2 + q] >> (8 * e)) & 0xffu) == want) acc[e] += g[r * 2 + q][e];
                    }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float vv = v[dy_ * 2 + dx_][e];
                    const float dp = acc[e] * act_grad(vv * sc[e] + sh[e], ACT, slope);
                    o[e] = sc[e] * dp + (k1[e] * (vv - mu[e]) + k0[e]);
                    mx = fmaxf(mx, fabsf(o[e]));
                }
                *reinterpret_cast<f32x4*>(dy + (((size_t)n * p.IH + 2 * a + dy_) * p.IW + 2 * b + dx_) * C + c) = o;
            }
    }
    if (amax != nullptr) block_absmax_to(amax, mx);
}

template <int ACT, bool FIXED, bool POOL>
int launch_bn_bwd_apply_t(const float* dz, const float* y, const float* mean, const float* scale, const float* shift, const float* sums,
                          float* dy, long n4, int C, float slope, float* amax, hipStream_t st, const PoolGather& pg) {
    auto a0 = reinterpret_cast<const f32x4*>(dz); auto a1 = reinterpret_cast<const f32x4*>(y); auto a2 = reinterpret_cast<f32x4*>(dy);
    if (amax != nullptr) VIAI_LAUNCH((bn_bwd_apply_kernel<ACT, FIXED, POOL, 1024>), dim3(stream_grid(n4, 1024)), dim3(1024), 0, st, a0, a1, mean, scale, shift, sums, a2, n4, C, slope, amax, pg);
    else VIAI_LAUNCH((bn_bwd_apply_kernel<ACT, FIXED, POOL, 256>), dim3(stream_grid(n4, 256)), dim3(256), 0, st, a0, a1, mean, scale, shift, sums, a2, n4, C, slope, amax, pg);
    return viai_launch_status();
}

template <bool FIXED>
int launch_bn_bwd_apply(const float* dz, const float* y, const float* mean, const float* scale, const float* shift, const float* sums,
                        float* dy, long n4, int C, int act, float slope, float* amax, hipStream_t st) {
    const PoolGather pg{};
    switch (act) {
    case VIAI_ACT_RELU: return launch_bn_bwd_apply_t<VIAI_ACT_RELU, FIXED, false>(dz, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
    case VIAI_ACT_LRELU: return launch_bn_bwd_apply_t<VIAI_ACT_LRELU, FIXED, false>(dz, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
    case VIAI_ACT_SIGMOID: return launch_bn_bwd_apply_t<VIAI_ACT_SIGMOID, FIXED, false>(dz, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
    default: return launch_bn_bwd_apply_t<VIAI_ACT_NONE, FIXED, false>(dz, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
    }
}

// ---- pre-split (P16) producers: the same two passes writing their output as fp16 planes (viai_bf3.h) -- what the f16x2 consumers would
// otherwise make of the fp32 values, once per consumer and per staging.  A thread owns a channel OCTET (two 16-byte loads, one 16-byte store
// per plane).  The scale must be known before the pass, so it comes from a bound instead of the measured maximum:
//   forward  |z| = |act(gamma xhat + beta)| <= |gamma| sqrt(M - 1) + |beta|   (training-mode statistics over M samples: Samuelson)
//   backward |dy| <= sums[2C + c]  (bn_bwd_final_kernel, ps = 3)
// every block reduces the per-channel bounds to the tensor's (max is exact: all blocks agree), block 0 stores it for the consumers.
// z (P16) = act(scale y + shift); `rad` = sqrt(M - 1) of the statistics' population.  FIXED: C / 8 is a power of two <= 256, so a thread keeps
// ONE octet for its whole grid-stride walk (coefficients loaded once, no division per item -- see bn_bwd_apply_kernel).
template <bool FIXED>
__global__ __launch_bounds__(256) void bn_act_fwd_p16_kernel(const f32x4* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float rad,
                                                             u32x4* __restrict__ z, long n8, int C, int act, float slope, float* __restrict__ amax) {
    const float S = p16_fwd_scale(gamma, beta, C, rad, amax), L = f16_clamp_for_scale(S);
    const unsigned c8n = (unsigned)(C / 8);
    const int lg = 31 - __builtin_clz(c8n);
    const long stride = (long)gridDim.x * 256;
    unsigned o = FIXED ? (threadIdx.x & (c8n - 1)) : 0u;
    f32x4 sc0, sc1, sh0, sh1;
    auto coeffs = [&]() {
        sc0 = *reinterpret_cast<const f32x4*>(scale + o * 8); sc1 = *reinterpret_cast<const f32x4*>(scale + o * 8 + 4);
        sh0 = *reinterpret_cast<const f32x4*>(shift + o * 8); sh1 = *reinterpret_cast<const f32x4*>(shift + o * 8 + 4);
    };
    if constexpr (FIXED) coeffs();
    auto item = [&](long i, const f32x4& v0, const f32x4& v1) {
        long pix;
        if constexpr (FIXED) pix = i >> lg;
        else { o = (unsigned)((unsigned long)i % c8n); pix = i / c8n; coeffs(); }
        f32x4 a0, a1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] = viai_act(v0[e] * sc0[e] + sh0[e], act, slope); a1[e] = viai_act(v1[e] * sc1[e] + sh1[e], act, slope); }
        u32x4 hi, lo;
        p16_split8(a0, a1, S, L, hi, lo);
        u32x4* dst = z + pix * (C / 4) + (o >> 2) * 8 + (o & 3);          // 16-byte units: group (o >> 2) starts at 8, piece o & 3
        dst[0] = hi; dst[4] = lo;
    };
    long i = blockIdx.x * 256L + threadIdx.x;
    for (; i + stride < n8; i += 2 * stride) {                            // four independent 16-byte loads in flight per thread
        const f32x4 v0 = y[2 * i], v1 = y[2 * i + 1], w0 = y[2 * (i + stride)], w1 = y[2 * (i + stride) + 1];
        item(i, v0, v1); item(i + stride, w0, w1);
    }
    for (; i < n8; i += stride) item(i, y[2 * i], y[2 * i + 1]);
}

// The residual join of a ResNet block with a P16 TWIN: z = act(scale y + shift + res) as fp32 (the next join's residual, the mask of this join's backward, the
// 1 x 1 downsample conv) AND as the two fp16 planes zp (the next block's conv1 forward and weight gradient stage 16-byte pieces instead of splitting 822 MB of
// fp32 per pass).  Scale of the planes: |z| <= |gamma| rad + |beta| + max |res| (res_amax: the exact maximum its producer published), in *p_amax; z's own exact
// maximum still goes to *z_amax (fp32 consumers keep their scale bit for bit).  C / 4 a power of two <= 256, C % 32 == 0.
template <int ACT>
__global__ __launch_bounds__(1024) void bn_add_act_twin_kernel(const f32x4* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float rad,
                                                               const f32x4* __restrict__ res, const float* __restrict__ res_amax,
                                                               f32x4* __restrict__ z, float* __restrict__ zp, long n4, int C, float slope,
                                                               float* __restrict__ z_amax, float* __restrict__ p_amax) {
    float b = 0.f;
    for (int c = threadIdx.x; c < C; c += 1024) b = fmaxf(b, fabsf(gamma[c]) * rad + fabsf(beta[c]));
    const float bound = (block_max_all(b) + res_amax[0]) * 1.001f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *p_amax = bound;
    const float S = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(f16_scale_from_amax_value(bound)))), L = f16_clamp_for_scale(S);
    const unsigned c4n = (unsigned)(C / 4);
    const int lg = 31 - __builtin_clz(c4n);
    const long stride = (long)gridDim.x * 1024;
    long i = blockIdx.x * 1024L + threadIdx.x;
    const unsigned cq = threadIdx.x & (c4n - 1);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + cq * 4), sh = *reinterpret_cast<const f32x4*>(shift + cq * 4);
    float mx = 0.f;
    auto one = [&](long k, const f32x4& v, const f32x4& r) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = viai_act((v[e] * sc[e] + sh[e]) + r[e], ACT, slope); mx = fmaxf(mx, fabsf(o[e])); }
        z[k] = o;
        p16_store_quad(zp + (k >> lg) * C, (int)cq, o, S, L);
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = y[i + u * stride]; r[u] = res[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(i + u * stride, v[u], r[u]);
    }
    for (; i < n4; i += stride) one(i, y[i], res[i]);
    block_absmax_to(z_amax, mx);
}

// dy (P16) = scale dpre + k1 (y - mean) + k0
template <bool FIXED>
__global__ __launch_bounds__(256) void bn_bwd_apply_p16_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ y, const float* __restrict__ mean,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ sums,
                                                               u32x4* __restrict__ dy, long n8, int C, int act, float slope, float* __restrict__ amax,
                                                               f32x4* __restrict__ dy32 = nullptr) {
    // dy32 (viai_bn_act_bwd_p16_twin): the fp32 values as well, for a data-gradient kernel without a P16 loader beside a weight-gradient kernel with one
    float b = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) b = fmaxf(b, sums[2 * C + c]);
    const float bound = block_max_all(b);
    if (blockIdx.x == 0 && threadIdx.x == 0) *amax = bound;
    const float S = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(f16_scale_from_amax_value(bound)))), L = f16_clamp_for_scale(S);
    const unsigned c8n = (unsigned)(C / 8);
    const int lg = 31 - __builtin_clz(c8n);
    const long stride = (long)gridDim.x * 256;
    unsigned o = FIXED ? (threadIdx.x & (c8n - 1)) : 0u;
    f32x4 sc[2], sh[2], k0[2], k1[2], mu[2];
    auto coeffs = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = (int)o * 8 + 4 * h;
            sc[h] = *reinterpret_cast<const f32x4*>(scale + c); sh[h] = *reinterpret_cast<const f32x4*>(shift + c);
            k0[h] = *reinterpret_cast<const f32x4*>(sums + c); k1[h] = *reinterpret_cast<const f32x4*>(sums + C + c);
            mu[h] = *reinterpret_cast<const f32x4*>(mean + c);
        }
    };
    if constexpr (FIXED) coeffs();
    auto item = [&](long i, const f32x4 (&g)[2], const f32x4 (&v)[2]) {
        long pix;
        if constexpr (FIXED) pix = i >> lg;
        else { o = (unsigned)((unsigned long)i % c8n); pix = i / c8n; coeffs(); }
        f32x4 r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dp = g[h][e] * act_grad(v[h][e] * sc[h][e] + sh[h][e], act, slope);
                r[h][e] = sc[h][e] * dp + (k1[h][e] * (v[h][e] - mu[h][e]) + k0[h][e]);          // the expression of bn_bwd_apply_kernel, bit for bit
            }
        u32x4 hi, lo;
        p16_split8(r[0], r[1], S, L, hi, lo);
        u32x4* dst = dy + pix * (C / 4) + (o >> 2) * 8 + (o & 3);
        dst[0] = hi; dst[4] = lo;
        if (dy32 != nullptr) { dy32[2 * i] = r[0]; dy32[2 * i + 1] = r[1]; }
    };
    long i = blockIdx.x * 256L + threadIdx.x;
    for (; i + stride < n8; i += 2 * stride) {                            // eight independent 16-byte loads in flight per thread
        const f32x4 g0[2] = {dz[2 * i], dz[2 * i + 1]}, v0[2] = {y[2 * i], y[2 * i + 1]};
        const f32x4 g1[2] = {dz[2 * (i + stride)], dz[2 * (i + stride) + 1]}, v1[2] = {y[2 * (i + stride)], y[2 * (i + stride) + 1]};
        item(i, g0, v0); item(i + stride, g1, v1);
    }
    for (; i < n8; i += stride) { const f32x4 g0[2] = {dz[2 * i], dz[2 * i + 1]}, v0[2] = {y[2 * i], y[2 * i + 1]}; item(i, g0, v0); }
}

__global__ void act_bwd_out_kernel(const float* __restrict__ dz, const float* __restrict__ z, float* __restrict__ dx,
                                   long n, int act, float slope) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float zz = z[i], g = dz[i], d = 1.f;
        if (act == VIAI_ACT_SIGMOID) d = zz * (1.f - zz);
        else if (act == VIAI_ACT_RELU) d = zz > 0.f ? 1.f : 0.f;
        else if (act == VIAI_ACT_LRELU) d = zz > 0.f ? 1.f : slope;
        dx[i] = g * d;
    }
}

inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int viai_bn_finalize(const float* stat_part, int nblk, int rows_per_blk, long M, int C,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                int64_t* nbt, float momentum, float eps,
                                float* mean, float* invstd, float* scale, float* shift, void* stream) {
    if (nblk <= 0 || C <= 0 || M <= 0) return (int)hipErrorInvalidValue;
    if (nblk > 4096) VIAI_LAUNCH(bn_finalize_kernel<1024>, dim3(C), dim3(1024), 0, (hipStream_t)stream, stat_part, nblk, rows_per_blk, M, C,
                                 gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift, 0, 0, 0, 0);
    else VIAI_LAUNCH(bn_finalize_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, stat_part, nblk, rows_per_blk, M, C,
                     gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift, 0, 0, 0, 0);
    return viai_launch_status();
}

extern "C" int viai_bn_finalize_tiles(const float* stat_part, int N, int OH, int OW, int tile_h, int tile_w, int C,
                                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                                      int64_t* nbt, float momentum, float eps,
                                      float* mean, float* invstd, float* scale, float* shift, void* stream) {
    if (N <= 0 || OH <= 0 || OW <= 0 || tile_h <= 0 || tile_w <= 0 || C <= 0) return (int)hipErrorInvalidValue;
    const int nblk = N * ((OH + tile_h - 1) / tile_h) * ((OW + tile_w - 1) / tile_w);
    if (nblk > 4096) VIAI_LAUNCH(bn_finalize_kernel<1024>, dim3(C), dim3(1024), 0, (hipStream_t)stream, stat_part, nblk, tile_h * tile_w, (long)N * OH * OW, C,
                                 gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift, tile_h, tile_w, OH, OW);
    else VIAI_LAUNCH(bn_finalize_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, stat_part, nblk, tile_h * tile_w, (long)N * OH * OW, C,
                     gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift, tile_h, tile_w, OH, OW);
    return viai_launch_status();
}

// (ABI 17) the finalize behind conv_lin_dma_kernel (a pre-split forward whose viai_conv2d_p16_ok mask has VIAI_P16_OK_FWD_LIN): the kernel's partials are per 128
// consecutive pixels, or -- layers with one channel block, round 6 -- merged per persistent block; the library knows which (same predicate as the launch).
extern "C" int viai_bn_finalize_lin(const float* stat_part, long M, int C,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    int64_t* nbt, float momentum, float eps,
                                    float* mean, float* invstd, float* scale, float* shift, void* stream) {
    if (C <= 0 || M <= 0 || M % 128 != 0) return (int)hipErrorInvalidValue;
    int G = 0, PW = 0, items = 0;
    const int parts = viai_lin_dma_stat_merge(M, C, &G, &PW, &items);
    if (parts <= 0) return viai_bn_finalize(stat_part, (int)(M / 128), 128, M, C, gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift, stream);
    VIAI_LAUNCH(bn_finalize_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, stat_part, parts, 128, M, C,
                gamma, beta, running_mean, running_var, nbt, momentum, eps, mean, invstd, scale, shift, -1, PW, items, G);
    return viai_launch_status();
}

extern "C" int viai_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                   float eps, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    VIAI_LAUNCH(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta, rm, rv, eps,
                       mean, invstd, scale, shift);
    return viai_launch_status();
}

extern "C" int viai_bn_act_fwd_amax(const float* y, const float* scale, const float* shift, float* z,
                                    long M, int C, int act, float slope, float* z_amax, void* stream) {
    if (C % 4 != 0) return (int)hipErrorInvalidValue;
    long n4 = M * C / 4;
    const int c4n = C / 4;
    if (256 % c4n == 0) return launch_bn_act_fwd<true>(y, scale, shift, z, n4, C, act, slope, z_amax, (hipStream_t)stream);
    return launch_bn_act_fwd<false>(y, scale, shift, z, n4, C, act, slope, z_amax, (hipStream_t)stream);
}
// z = act(scale*y + shift + res): BatchNorm apply + residual add + ReLU of a ResNet block in one pass (ReLU or no activation)
extern "C" int viai_bn_add_act_fwd_amax(const float* y, const float* scale, const float* shift, const float* res, float* z,
                                        long M, int C, int act, float slope, float* z_amax, void* stream) {
    if (C % 4 != 0 || res == nullptr || (act != VIAI_ACT_RELU && act != VIAI_ACT_NONE)) return (int)hipErrorInvalidValue;
    const long n4 = M * C / 4;
    hipStream_t st = (hipStream_t)stream;
    const bool fixed = 256 % (C / 4) == 0;
    if (act == VIAI_ACT_RELU) {
        if (fixed) return launch_bn_act_fwd_t<VIAI_ACT_RELU, true, true>(y, scale, shift, res, z, n4, C, slope, z_amax, st);
        return launch_bn_act_fwd_t<VIAI_ACT_RELU, false, true>(y, scale, shift, res, z, n4, C, slope, z_amax, st);
    }
    if (fixed) return launch_bn_act_fwd_t<VIAI_ACT_NONE, true, true>(y, scale, shift, res, z, n4, C, slope, z_amax, st);
    return launch_bn_act_fwd_t<VIAI_ACT_NONE, false, true>(y, scale, shift, res, z, n4, C, slope, z_amax, st);
}

// viai_bn_add_act_fwd_amax with a second, pre-split (P16) copy of z for consumers that stage fp16 pieces (viai_conv2d_p16_ok): see bn_add_act_twin_kernel.
// m_stat: the population of the BatchNorm statistics; res_amax: max |res| (device, required); p_amax receives the planes' magnitude bound.
extern "C" int viai_bn_add_act_fwd_twin(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                        const float* res, const float* res_amax, float* z, float* z_p16, long M, int C, int act, float slope,
                                        float* z_amax, float* p_amax, void* stream) {
    const int c4 = C / 4;
    if (C % 32 != 0 || c4 > 256 || (c4 & (c4 - 1)) != 0 || res == nullptr || res_amax == nullptr || gamma == nullptr || beta == nullptr || z_amax == nullptr ||
        p_amax == nullptr || z_p16 == nullptr || (act != VIAI_ACT_RELU && act != VIAI_ACT_NONE))
        return (int)hipErrorInvalidValue;
    const long n4 = M * C / 4;
    hipStream_t st = (hipStream_t)stream;
    const float rad = sqrtf((float)(m_stat > 1 ? m_stat - 1 : 1));
    auto a0 = reinterpret_cast<const f32x4*>(y); auto a1 = reinterpret_cast<f32x4*>(z); auto a2 = reinterpret_cast<const f32x4*>(res);
    if (act == VIAI_ACT_RELU) VIAI_LAUNCH(bn_add_act_twin_kernel<VIAI_ACT_RELU>, dim3(stream_grid(n4, 1024)), dim3(1024), 0, st, a0, scale, shift, gamma, beta, rad, a2, res_amax, a1, z_p16, n4, C, slope, z_amax, p_amax);
    else VIAI_LAUNCH(bn_add_act_twin_kernel<VIAI_ACT_NONE>, dim3(stream_grid(n4, 1024)), dim3(1024), 0, st, a0, scale, shift, gamma, beta, rad, a2, res_amax, a1, z_p16, n4, C, slope, z_amax, p_amax);
    return viai_launch_status();
}

extern "C" int viai_bn_act_fwd(const float* y, const float* scale, const float* shift, float* z,
                               long M, int C, int act, float slope, void* stream) {
    return viai_bn_act_fwd_amax(y, scale, shift, z, M, C, act, slope, nullptr, stream);
}

// amax = max(amax, max |x|): the f16x2 operand scale of a tensor no kernel of this library produced (one streaming pass)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ amax) {
    float mx = 0.f;
    const long n4 = n / 4, stride = (long)gridDim.x * 256L;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = x4[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) mx = fmaxf(mx, fabsf(x[n4 * 4 + threadIdx.x]));
    block_absmax_to(amax, mx);
}
extern "C" int viai_absmax(const float* x, long n, float* amax, void* stream) {
    if (x == nullptr || amax == nullptr || n < 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    VIAI_LAUNCH(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, amax);
    return viai_launch_status();
}

extern "C" int viai_bn_bwd_blocks(long M, int C) {
    // ~512 row blocks for large tensors (two per CU, eight 16-byte loads in flight per lane: enough to cover HBM latency; more
    // blocks only lengthen the partial array the final kernel walks), but never fewer rows per block than one unrolled pass of
    // the reduce kernel covers (256 threads = C/4 channel quads x pixel lanes, 4 rows in flight per lane)
    constexpr long target = 512;
    long rows_min = 4096 / (C > 0 ? C : 1);
    if (rows_min < 4) rows_min = 4;
    long rows = (M + target - 1) / target;
    if (rows < rows_min) rows = rows_min;
    long b = (M + rows - 1) / rows;
    if (b < 1) b = 1;
    return (int)b;
}

// amax (optional, one zero-initialised float): receives max |dy| -- the operand scale of the f16x2 backward kernels
extern "C" int viai_bn_act_bwd_amax(const float* dz, const float* y, const float* mean, const float* invstd,
                               const float* scale, const float* shift, float* part, float* sums,
                               float* dgamma, float* dbeta, float* dy, long M, int C, int act, float slope,
                               int training, float* amax, void* stream) {
    if (C % 4 != 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = viai_bn_bwd_blocks(M, C);
    const long rpb = (M + nblk - 1) / nblk;
    VIAI_LAUNCH(bn_bwd_reduce_kernel<false>, dim3(nblk), dim3(256), 0, st, dz, y, mean, invstd, scale, shift, part, M, C, rpb, act, slope, PoolGather{}, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    VIAI_LAUNCH(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nblk, C, M, mean, invstd, scale, training & 1, sums, dgamma, dbeta, (training >> 1) & 1, 2);
    if (dy != nullptr) {
        long n4 = M * C / 4;
        const int c4n = C / 4;
        if (c4n > 0 && 256 % c4n == 0) return launch_bn_bwd_apply<true>(dz, y, mean, scale, shift, sums, dy, n4, C, act, slope, amax, st);
        return launch_bn_bwd_apply<false>(dz, y, mean, scale, shift, sums, dy, n4, C, act, slope, amax, st);
    }
    return viai_launch_status();
}

// The same backward where nn.MaxPool2d(k, s, p) follows the activation (ReLU or none): dpool (N, OH, OW, C) and the argmax bytes of
// viai_bn_act_maxpool_fwd stand in for dz, which is gathered on load in both passes.  y, dy: (N, IH, IW, C).
static int bn_act_pool_bwd_impl(const float* dpool, const float* dpool2, const unsigned char* idx, int N, int IH, int IW, int k, int s, int p,
                                         const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                                         float* part, float* sums, float* dgamma, float* dbeta, float* dy, int C, int act, float slope,
                                         int training, float* amax, void* stream) {
    if (C % 4 != 0 || dy == nullptr || k * k > 255 || (act != VIAI_ACT_RELU && act != VIAI_ACT_NONE)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)N * IH * IW;
    PoolGather pg{dpool, idx, dpool2, IH, IW, (IH + 2 * p - k) / s + 1, (IW + 2 * p - k) / s + 1, k, s, p};
    const int nblk = viai_bn_bwd_blocks(M, C);
    const long rpb = (M + nblk - 1) / nblk;
    const long MP = (long)N * pg.OH * pg.OW;
    int nb = nblk;
    if (k == 3 && M * C < (1l << 31)) {                  // the sums from the pooled side (int offsets into y): same partial layout, blocks over pooled pixels
        nb = viai_bn_bwd_blocks(MP, C);
        if (nb > nblk) nb = nblk;                        // (`part` is sized for viai_bn_bwd_blocks(M, C))
        const long rpp = (MP + nb - 1) / nb;
        VIAI_LAUNCH(bn_pool_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, y, mean, invstd, scale, shift, part, MP, C, rpp, act, slope, pg);
    } else
        VIAI_LAUNCH(bn_bwd_reduce_kernel<true>, dim3(nblk), dim3(256), 0, st, (const float*)nullptr, y, mean, invstd, scale, shift, part, M, C, rpb, act, slope, pg, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    VIAI_LAUNCH(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nb, C, M, mean, invstd, scale, training & 1, sums, dgamma, dbeta, (training >> 1) & 1, 2);
    const long n4 = M * C / 4;
    const bool fixed = 256 % (C / 4) == 0;
    if (k == 3 && s == 2 && p == 1 && IH % 2 == 0 && IW % 2 == 0) {       // the stem's pool: by 2 x 2 pixel blocks
        const long nb4 = n4 / 4;
        if (act == VIAI_ACT_RELU) VIAI_LAUNCH(bn_pool2x2_bwd_apply_kernel<VIAI_ACT_RELU>, dim3(stream_grid(nb4, 256)), dim3(256), 0, st, y, mean, scale, shift, sums, dy, nb4, C, slope, amax, pg);
        else VIAI_LAUNCH(bn_pool2x2_bwd_apply_kernel<VIAI_ACT_NONE>, dim3(stream_grid(nb4, 256)), dim3(256), 0, st, y, mean, scale, shift, sums, dy, nb4, C, slope, amax, pg);
        return viai_launch_status();
    }
    if (act == VIAI_ACT_RELU) {
        if (fixed) return launch_bn_bwd_apply_t<VIAI_ACT_RELU, true, true>(nullptr, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
        return launch_bn_bwd_apply_t<VIAI_ACT_RELU, false, true>(nullptr, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
    }
    if (fixed) return launch_bn_bwd_apply_t<VIAI_ACT_NONE, true, true>(nullptr, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
    return launch_bn_bwd_apply_t<VIAI_ACT_NONE, false, true>(nullptr, y, mean, scale, shift, sums, dy, n4, C, slope, amax, st, pg);
}

extern "C" int viai_bn_act_pool_bwd_amax(const float* dpool, const unsigned char* idx, int N, int IH, int IW, int k, int s, int p,
                                         const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                                         float* part, float* sums, float* dgamma, float* dbeta, float* dy, int C, int act, float slope,
                                         int training, float* amax, void* stream) {
    return bn_act_pool_bwd_impl(dpool, nullptr, idx, N, IH, IW, k, s, p, y, mean, invstd, scale, shift, part, sums, dgamma, dbeta, dy, C, act, slope, training, amax, stream);
}
// (ABI 15) the pooled gradient as two addends (dpool + dpool2, summed where it is loaded: the one fp32 addition autograd would have made in a pass of its own)
extern "C" int viai_bn_act_pool_bwd_amax2(const float* dpool, const float* dpool2, const unsigned char* idx, int N, int IH, int IW, int k, int s, int p,
                                          const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                                          float* part, float* sums, float* dgamma, float* dbeta, float* dy, int C, int act, float slope,
                                          int training, float* amax, void* stream) {
    return bn_act_pool_bwd_impl(dpool, dpool2, idx, N, IH, IW, k, s, p, y, mean, invstd, scale, shift, part, sums, dgamma, dbeta, dy, C, act, slope, training, amax, stream);
}

// the final pass alone (conv_direct.hip: the fused Cin = 1 layer produces the partials itself)
int viai_bn_bwd_final_launch(const float* part, int nblk, int C, long M, const float* mean, const float* invstd, const float* scale,
                             int training, float* sums, float* dgamma, float* dbeta, int accumulate, hipStream_t st, int ps) {
    VIAI_LAUNCH(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nblk, C, M, mean, invstd, scale, training, sums, dgamma, dbeta, accumulate, ps);
    return viai_launch_status();
}

extern "C" int viai_bn_act_bwd(const float* dz, const float* y, const float* mean, const float* invstd,
                               const float* scale, const float* shift, float* part, float* sums,
                               float* dgamma, float* dbeta, float* dy, long M, int C, int act, float slope,
                               int training, void* stream) {
    return viai_bn_act_bwd_amax(dz, y, mean, invstd, scale, shift, part, sums, dgamma, dbeta, dy, M, C, act, slope, training, nullptr, stream);
}

// dx = (dz + dz2) * act'(.): the gradient of a residual join's output arrives as its two addends (the next block's conv1 data gradient and the
// next join's residual gradient, networks/ResNet.py:46-53) and is summed where the activation's mask is applied, instead of in a pass of its own
__global__ void add_act_bwd_out_kernel(const f32x4* __restrict__ dz, const f32x4* __restrict__ dz2, const f32x4* __restrict__ z, f32x4* __restrict__ dx,
                                       long n4, int act, float slope) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 zz = z[i], g = dz[i] + dz2[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = 1.f;
            if (act == VIAI_ACT_SIGMOID) d = zz[e] * (1.f - zz[e]);
            else if (act == VIAI_ACT_RELU) d = zz[e] > 0.f ? 1.f : 0.f;
            else if (act == VIAI_ACT_LRELU) d = zz[e] > 0.f ? 1.f : slope;
            o[e] = g[e] * d;
        }
        dx[i] = o;
    }
}
extern "C" int viai_add_act_bwd_from_output(const float* dz, const float* dz2, const float* z, float* dx, long n, int act, float slope, void* stream) {
    if (n % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(add_act_bwd_out_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)dz, (const f32x4*)dz2, (const f32x4*)z, (f32x4*)dx,
                n / 4, act, slope);
    return viai_launch_status();
}

extern "C" int viai_act_bwd_from_output(const float* dz, const float* z, float* dx, long n, int act, float slope, void* stream) {
    VIAI_LAUNCH(act_bwd_out_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dz, z, dx, n, act, slope);
    return viai_launch_status();
}

// ---- (ABI 13) pre-split outputs.  z / dy are written as P16 planes (csrc/viai_bf3.h: per pixel and 32-channel group, 32 leading fp16 terms then
// 32 remainder terms of value * scale), the layout the f16x2 conv kernels stage without converting; *amax receives the magnitude BOUND the
// scale was derived from (consumers derive the same scale from it).  C % 32 == 0.
extern "C" int viai_bn_act_fwd_p16(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                   float* z, long M, int C, int act, float slope, float* z_amax, void* stream) {
    if (C % 32 != 0 || z_amax == nullptr || m_stat < 1 || act == VIAI_ACT_SIGMOID) return (int)hipErrorInvalidValue;
    const long n8 = M * C / 8;
    const float rad = sqrtf((float)(m_stat > 1 ? m_stat - 1 : 1));
    const int c8n = C / 8;
    if ((c8n & (c8n - 1)) == 0 && c8n <= 256)
        VIAI_LAUNCH(bn_act_fwd_p16_kernel<true>, dim3(stream_grid(n8, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(y), scale, shift, gamma, beta, rad,
                    reinterpret_cast<u32x4*>(z), n8, C, act, slope, z_amax);
    else
        VIAI_LAUNCH(bn_act_fwd_p16_kernel<false>, dim3(stream_grid(n8, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(y), scale, shift, gamma, beta, rad,
                    reinterpret_cast<u32x4*>(z), n8, C, act, slope, z_amax);
    return viai_launch_status();
}

// viai_bn_act_bwd_amax with dy pre-split; part: 3 * C * viai_bn_bwd_blocks(M, C) floats, sums: 3 * C floats
// (ABI 15) BatchNorm backward behind a residual join with a ReLU: dres = (dz + dz2) * [zj > 0] (dz2 may be NULL) is written and takes dz's place in
// viai_bn_act_bwd_p16 with no activation of the BatchNorm's own -- bit for bit viai_add_act_bwd_from_output followed by viai_bn_act_bwd_p16
extern "C" int viai_bn_join_bwd_p16(const float* dz, const float* dz2, const float* zj, float* dres, const float* y, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, float* part, float* sums, float* dgamma, float* dbeta, float* dy,
                                    long M, int C, int training, float* amax, void* stream) {
    if (C % 32 != 0 || amax == nullptr || dy == nullptr || dres == nullptr || zj == nullptr) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = viai_bn_bwd_blocks(M, C);
    const long rpb = (M + nblk - 1) / nblk;
    VIAI_LAUNCH((bn_bwd_reduce_kernel<false, true, true>), dim3(nblk), dim3(256), 0, st, dz, y, mean, invstd, scale, shift, part, M, C, rpb, VIAI_ACT_NONE, 0.f, PoolGather{},
                dz2, zj, dres);
    VIAI_LAUNCH(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nblk, C, M, mean, invstd, scale, training & 1, sums, dgamma, dbeta, (training >> 1) & 1, 3);
    const long n8 = M * C / 8;
    const int c8n = C / 8;
    if ((c8n & (c8n - 1)) == 0 && c8n <= 256)
        VIAI_LAUNCH(bn_bwd_apply_p16_kernel<true>, dim3(stream_grid(n8, 256)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dres), reinterpret_cast<const f32x4*>(y), mean, scale, shift, sums,
                    reinterpret_cast<u32x4*>(dy), n8, C, VIAI_ACT_NONE, 0.f, amax, (f32x4*)nullptr);
    else
        VIAI_LAUNCH(bn_bwd_apply_p16_kernel<false>, dim3(stream_grid(n8, 256)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dres), reinterpret_cast<const f32x4*>(y), mean, scale, shift, sums,
                    reinterpret_cast<u32x4*>(dy), n8, C, VIAI_ACT_NONE, 0.f, amax, (f32x4*)nullptr);
    return viai_launch_status();
}

static int bn_act_bwd_p16_impl(const float* dz, const float* y, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, float* part, float* sums,
                                   float* dgamma, float* dbeta, float* dy, long M, int C, int act, float slope,
                                   int training, float* amax, float* dy32, void* stream) {
    if (C % 32 != 0 || amax == nullptr || dy == nullptr || act == VIAI_ACT_SIGMOID) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = viai_bn_bwd_blocks(M, C);
    const long rpb = (M + nblk - 1) / nblk;
    VIAI_LAUNCH((bn_bwd_reduce_kernel<false, true>), dim3(nblk), dim3(256), 0, st, dz, y, mean, invstd, scale, shift, part, M, C, rpb, act, slope, PoolGather{}, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    VIAI_LAUNCH(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nblk, C, M, mean, invstd, scale, training & 1, sums, dgamma, dbeta, (training >> 1) & 1, 3);
    const long n8 = M * C / 8;
    const int c8n = C / 8;
    if ((c8n & (c8n - 1)) == 0 && c8n <= 256)
        VIAI_LAUNCH(bn_bwd_apply_p16_kernel<true>, dim3(stream_grid(n8, 256)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dz), reinterpret_cast<const f32x4*>(y), mean, scale, shift, sums,
                    reinterpret_cast<u32x4*>(dy), n8, C, act, slope, amax, reinterpret_cast<f32x4*>(dy32));
    else
        VIAI_LAUNCH(bn_bwd_apply_p16_kernel<false>, dim3(stream_grid(n8, 256)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(dz), reinterpret_cast<const f32x4*>(y), mean, scale, shift, sums,
                    reinterpret_cast<u32x4*>(dy), n8, C, act, slope, amax, reinterpret_cast<f32x4*>(dy32));
    return viai_launch_status();
}
extern "C" int viai_bn_act_bwd_p16(const float* dz, const float* y, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, float* part, float* sums,
                                   float* dgamma, float* dbeta, float* dy, long M, int C, int act, float slope,
                                   int training, float* amax, void* stream) {
    return bn_act_bwd_p16_impl(dz, y, mean, invstd, scale, shift, part, sums, dgamma, dbeta, dy, M, C, act, slope, training, amax, nullptr, stream);
}
// (ABI 15) ... and the fp32 tensor beside the planes (dy32: M x C floats, the values viai_bn_act_bwd_amax writes): a layer whose data-gradient kernel
// has no P16 loader while its weight-gradient kernel has one (the stride-2 3 x 3 convs of ResNet-18 on 28 / 14 / 7-pixel maps, networks/ResNet.py:100-112)
extern "C" int viai_bn_act_bwd_p16_twin(const float* dz, const float* y, const float* mean, const float* invstd,
                                        const float* scale, const float* shift, float* part, float* sums,
                                        float* dgamma, float* dbeta, float* dy, float* dy32, long M, int C, int act, float slope,
                                        int training, float* amax, void* stream) {
    if (dy32 == nullptr) return (int)hipErrorInvalidValue;
    return bn_act_bwd_p16_impl(dz, y, mean, invstd, scale, shift, part, sums, dgamma, dbeta, dy, M, C, act, slope, training, amax, dy32, stream);
}

// fp32 view of a pre-split tensor: x[i] = (leading + remainder) / scale(*amax).  For tests and for consumers without a P16 loader.
__global__ __launch_bounds__(256) void p16_decode_kernel(const u32x4* __restrict__ p, f32x4* __restrict__ x, long n8, int C, const float* __restrict__ amax) {
    const float inv = 1.f / f16_scale_from_amax(amax);
    const unsigned c8n = (unsigned)(C / 8);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const unsigned o = (unsigned)((unsigned long)i % c8n);
        const long pix = i / c8n;
        const u32x4* src = p + pix * (C / 4) + (o >> 2) * 8 + (o & 3);
        const u32x4 hi = src[0], lo = src[4];
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            a[2 * e] = (f16_lo(hi[e]) + f16_lo(lo[e])) * inv; a[2 * e + 1] = (f16_hi(hi[e]) + f16_hi(lo[e])) * inv;
            b[2 * e] = (f16_lo(hi[2 + e]) + f16_lo(lo[2 + e])) * inv; b[2 * e + 1] = (f16_hi(hi[2 + e]) + f16_hi(lo[2 + e])) * inv;
        }
        x[2 * i] = a; x[2 * i + 1] = b;
    }
}
extern "C" int viai_p16_decode(const float* p16, float* x, long M, int C, const float* amax, void* stream) {
    if (C % 32 != 0 || amax == nullptr) return (int)hipErrorInvalidValue;
    const long n8 = M * C / 8;
    VIAI_LAUNCH(p16_decode_kernel, dim3(stream_grid(n8, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const u32x4*>(p16), reinterpret_cast<f32x4*>(x), n8, C, amax);
    return viai_launch_status();
}
