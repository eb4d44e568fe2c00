// WaveNet vocoder pieces that are not convolutions (gfx950): weight normalisation, the gated activation,
// residual / skip scaling, the scalar-input first layer, the conditioning up-sampler, the discretised
// mixture-of-logistics loss and sampler.  The dilated / 1x1 Conv1d layers themselves run on the implicit-GEMM
// MFMA kernels (conv_igemm.hip, H = 1).  Reference: wavenet_vocoder/{wavenet,modules,mixture}.py.
// Activations are (B, 1, T, C) NHWC, i.e. (B*T) rows of C channels.
#include "viai_common.h"
#include "viai_internal.h"

namespace {

inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ float block_sum(float v, float* red) {      // 256 threads
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------- weight norm (modules.py:39, torch weight_norm dim=0)
// w[r][:] = g[r] * v[r][:] / ||v[r]||      one block per row r
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ w, float* __restrict__ norm, int L) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) { float t = v[(size_t)r * L + i]; s += t * t; }
    const float n = sqrtf(block_sum(s, red));
    const float sc = g[r] / n;
    for (int i = threadIdx.x; i < L; i += 256) w[(size_t)r * L + i] = v[(size_t)r * L + i] * sc;
    if (threadIdx.x == 0) norm[r] = n;
}

// dg[r] = <dw, v>/n ;  dv = g/n * (dw - v * <dw, v>/n^2)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                              const float* __restrict__ g, const float* __restrict__ norm,
                                                              float* __restrict__ dv, float* __restrict__ dg, int L, int accumulate) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) s += dw[(size_t)r * L + i] * v[(size_t)r * L + i];
    const float dot = block_sum(s, red);
    const float n = norm[r], gs = g[r] / n, k = dot / (n * n);
    for (int i = threadIdx.x; i < L; i += 256) {
        float t = gs * (dw[(size_t)r * L + i] - v[(size_t)r * L + i] * k);
        dv[(size_t)r * L + i] = accumulate ? dv[(size_t)r * L + i] + t : t;
    }
    if (threadIdx.x == 0) dg[r] = accumulate ? dg[r] + dot / n : dot / n;
}

// ---------------------------------------------------------------- gated activation (modules.py:183-201)
// y, yc: rows of 2*H channels [a | b];  z[row][h] = tanh(a + ca) * sigmoid(b + cb)
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ yc,
                                                      float* __restrict__ z, long rows, int H) {
    const int h4n = H / 4;
    const long total = rows * h4n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long r = i / h4n; int h = (int)(i % h4n) * 4;
        f32x4 a = *reinterpret_cast<const f32x4*>(y + r * 2 * H + h), b = *reinterpret_cast<const f32x4*>(y + r * 2 * H + H + h);
        if (yc) { a += *reinterpret_cast<const f32x4*>(yc + r * 2 * H + h); b += *reinterpret_cast<const f32x4*>(yc + r * 2 * H + H + h); }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tanhf(a[e]) * (1.f / (1.f + expf(-b[e])));
        *reinterpret_cast<f32x4*>(z + r * H + h) = o;
    }
}

// dy[row] = [dz*(1-tanh^2)*sig | dz*tanh*sig*(1-sig)]   (the same tensor is the gradient of y AND of yc)
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y, const float* __restrict__ yc,
                                                      float* __restrict__ dy, long rows, int H) {
    const int h4n = H / 4;
    const long total = rows * h4n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long r = i / h4n; int h = (int)(i % h4n) * 4;
        f32x4 a = *reinterpret_cast<const f32x4*>(y + r * 2 * H + h), b = *reinterpret_cast<const f32x4*>(y + r * 2 * H + H + h);
        if (yc) { a += *reinterpret_cast<const f32x4*>(yc + r * 2 * H + h); b += *reinterpret_cast<const f32x4*>(yc + r * 2 * H + H + h); }
        f32x4 g = *reinterpret_cast<const f32x4*>(dz + r * H + h), da, db;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = tanhf(a[e]), s = 1.f / (1.f + expf(-b[e]));
            da[e] = g[e] * (1.f - t * t) * s;
            db[e] = g[e] * t * s * (1.f - s);
        }
        *reinterpret_cast<f32x4*>(dy + r * 2 * H + h) = da;
        *reinterpret_cast<f32x4*>(dy + r * 2 * H + H + h) = db;
    }
}

// out = (a + b) * s   (b may be null: out = a * s)
__global__ __launch_bounds__(256) void add_scale_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ o, float s, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) o[i] = b ? (a[i] + b[i]) * s : a[i] * s;
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ o, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        f32x4 v = a[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        o[i] = v;
    }
}

// ---------------------------------------------------------------- scalar-input first conv (wavenet.py:118: Conv1d1x1(1, C))
// y[p][c] = x[p] * w[c] + b[c]
__global__ __launch_bounds__(256) void outer_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, long rows, int C) {
    const int c4n = C / 4;
    const long total = rows * c4n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long r = i / c4n; int c = (int)(i % c4n) * 4;
        f32x4 wv = *reinterpret_cast<const f32x4*>(w + c), bv = *reinterpret_cast<const f32x4*>(b + c);
        *reinterpret_cast<f32x4*>(y + r * C + c) = x[r] * wv + bv;
    }
}

// part[blk][0][c] = sum_p dy[p][c] * x[p];  part[blk][1][c] = sum_p dy[p][c]
__global__ __launch_bounds__(256) void outer_bwd_part_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                             long rows, int C, long rows_per_blk) {
    __shared__ f32x4 r1[256], r2[256];
    const int tid = threadIdx.x, CG = C / 4;
    const long row0 = blockIdx.x * rows_per_blk;
    long row1 = row0 + rows_per_blk; if (row1 > rows) row1 = rows;
    for (int g0 = 0; g0 < CG; g0 += 256) {
        const int cgw = min(256, CG - g0), pg = 256 / cgw, cg = tid % cgw, pl = tid / cgw;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (pl < pg)
            for (long r = row0 + pl; r < row1; r += pg) {
                f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * C + (g0 + cg) * 4);
                s1 += g * x[r]; s2 += g;
            }
        r1[tid] = s1; r2[tid] = s2;
        __syncthreads();
        if (tid < cgw) {
            f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < pg; ++k) { t1 += r1[k * cgw + tid]; t2 += r2[k * cgw + tid]; }
            *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + (g0 + tid) * 4) = t1;
            *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + (g0 + tid) * 4) = t2;
        }
        __syncthreads();
    }
}

__global__ void outer_bwd_final_kernel(const float* __restrict__ part, int nblk, int C, float* dw, float* db, int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) { s1 += (double)part[((size_t)b * 2) * C + c]; s2 += (double)part[((size_t)b * 2 + 1) * C + c]; }
    dw[c] = accumulate ? dw[c] + (float)s1 : (float)s1;
    db[c] = accumulate ? db[c] + (float)s2 : (float)s2;
}

// ---------------------------------------------------------------- conditioning up-sampler (wavenet.py:153-164)
// ConvTranspose2d(1, 1, (KH, S), stride (1, S), padding ((KH-1)/2, 0)) + ReLU on (B, F, T):
//   out[b][f][t*S + j] = relu(bias + sum_r in[b][f + P - r][t] * w[r][j])
template <int KH>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, long BF, int F, int T, int S) {
    constexpr int P = (KH - 1) / 2;
    const long total = BF * T * S;
    const float b0 = bias[0];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long bf = i / ((long)T * S); int to = (int)(i % ((long)T * S));
        int t = to / S, j = to % S, f = (int)(bf % F);
        float acc = b0;
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            int fi = f + P - r;
            if ((unsigned)fi < (unsigned)F) acc += x[(bf - f + fi) * T + t] * w[r * S + j];
        }
        y[i] = acc > 0.f ? acc : 0.f;
    }
}

// dx[b][f][t] = sum_{r,j} dpre[b][f - P + r][t*S + j] * w[r][j],  dpre = dy * (y > 0)
template <int KH>
__global__ __launch_bounds__(256) void upsample_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ w,
                                                             float* __restrict__ dx, long BF, int F, int T, int S) {
    constexpr int P = (KH - 1) / 2;
    const long total = BF * T;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long bf = i / T; int t = (int)(i % T), f = (int)(bf % F);
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            int fo = f - P + r;
            if ((unsigned)fo >= (unsigned)F) continue;
            const long base = ((bf - f + fo) * T + t) * S;
            for (int j = 0; j < S; ++j) { float g = y[base + j] > 0.f ? dy[base + j] : 0.f; acc += g * w[r * S + j]; }
        }
        dx[i] = acc;
    }
}

// part[blk][r*S + j] = sum dpre[b][f][t*S+j] * x[b][f + P - r][t];  part[blk][KH*S] = sum dpre     (S <= 16)
template <int KH>
__global__ __launch_bounds__(256) void upsample_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                                             float* __restrict__ part, long BF, int F, int T, int S) {
    constexpr int P = (KH - 1) / 2;
    __shared__ float red[4];
    float acc[KH * 16 + 1];
#pragma unroll
    for (int k = 0; k < KH * 16 + 1; ++k) acc[k] = 0.f;
    const long total = BF * T;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        long bf = i / T; int t = (int)(i % T), f = (int)(bf % F);
        float xin[KH];
#pragma unroll
        for (int r = 0; r < KH; ++r) { int fi = f + P - r; xin[r] = ((unsigned)fi < (unsigned)F) ? x[(bf - f + fi) * T + t] : 0.f; }
        const long base = (bf * T + t) * S;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < S) {
                float g = y[base + j] > 0.f ? dy[base + j] : 0.f;
                acc[KH * 16] += g;
#pragma unroll
                for (int r = 0; r < KH; ++r) acc[r * 16 + j] += g * xin[r];
            }
    }
    for (int k = 0; k < KH * 16 + 1; ++k) {
        float s = block_sum(acc[k], red);
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * (KH * 16 + 1) + k] = s;
    }
}

__global__ void upsample_bwd_final_kernel(const float* __restrict__ part, int nblk, int KH, int S, float* dw, float* db, int accumulate) {
    int k = threadIdx.x;
    const int W = KH * 16 + 1;
    if (k >= W) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[(size_t)b * W + k];
    if (k == KH * 16) { db[0] = accumulate ? db[0] + (float)s : (float)s; return; }
    int r = k / 16, j = k % 16;
    if (j < S) dw[r * S + j] = accumulate ? dw[r * S + j] + (float)s : (float)s;
}

// ---------------------------------------------------------------- discretised mixture of logistics (mixture.py:25-105)
__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch softplus (threshold 20)
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// yhat: rows of `pitch` floats, first 3*K are [logit | mean | log_scale]; target y[row]; out loss[row] = -logsumexp_k(...)
// dyh (optional, same pitch): d(sum_rows wrow[row] * loss[row]) / d yhat
template <int K>
__global__ __launch_bounds__(256) void mol_loss_kernel(const float* __restrict__ yhat, const float* __restrict__ y, const float* __restrict__ wrow,
                                                       float* __restrict__ loss, float* __restrict__ dyh, long rows, int pitch,
                                                       float num_classes, float log_scale_min) {
    const float h = 1.f / (num_classes - 1.f), logh2 = logf((num_classes - 1.f) * 0.5f);
    for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256L) {
        const float* p = yhat + r * pitch;
        const float yy = y[r];
        float logit[K], lp[K], dm[K], dls[K];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) { logit[k] = p[k]; mx = fmaxf(mx, logit[k]); }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) se += expf(logit[k] - mx);
        const float lse_logit = mx + logf(se);
        float best = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float m = p[K + k], lsr = p[2 * K + k];
            const bool clamped = lsr < log_scale_min;
            const float ls = clamped ? log_scale_min : lsr;
            const float inv = expf(-ls), c = yy - m;
            const float pin = inv * (c + h), min_ = inv * (c - h), mid = inv * c;
            float v, gm, gs;                              // value, d/dmean, d/dlog_scale
            if (yy < -0.999f) {
                v = pin - softplusf(pin);
                float d = sigmoidf(-pin);
                gm = d * (-inv); gs = d * (-pin);
            } else if (yy > 0.999f) {
                v = -softplusf(min_);
                float d = -sigmoidf(min_);
                gm = d * (-inv); gs = d * (-min_);
            } else {
                const float sp = sigmoidf(pin), sm = sigmoidf(min_), cd = sp - sm;
                if (cd > 1e-5f) {
                    v = logf(fmaxf(cd, 1e-12f));
                    const float dp = sp * (1.f - sp) / cd, dn = -sm * (1.f - sm) / cd;
                    gm = (dp + dn) * (-inv); gs = dp * (-pin) + dn * (-min_);
                } else {
                    v = mid - ls - 2.f * softplusf(mid) - logh2;
                    const float d = 1.f - 2.f * sigmoidf(mid);
                    gm = d * (-inv); gs = d * (-mid) - 1.f;
                }
            }
            if (clamped) gs = 0.f;
            lp[k] = v + (logit[k] - lse_logit);
            dm[k] = gm; dls[k] = gs;
            best = fmaxf(best, lp[k]);
        }
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) s2 += expf(lp[k] - best);
        const float lse = best + logf(s2);
        loss[r] = -lse;
        if (dyh) {
            const float wr = wrow ? wrow[r] : 1.f;
            float* q = dyh + r * pitch;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float pk = expf(lp[k] - lse);                 // posterior responsibility
                const float sk = expf(logit[k] - lse_logit);        // prior softmax
                q[k] = wr * (sk - pk);
                q[K + k] = wr * (-pk * dm[k]);
                q[2 * K + k] = wr * (-pk * dls[k]);
            }
            for (int k = 3 * K; k < pitch; ++k) q[k] = 0.f;
        }
    }
}

// sample (mixture.py:117-153) with INJECTED uniforms u1[row][K], u2[row] in (1e-5, 1-1e-5)
template <int K>
__global__ __launch_bounds__(256) void mol_sample_kernel(const float* __restrict__ yhat, const float* __restrict__ u1, const float* __restrict__ u2,
                                                         float* __restrict__ out, long rows, int pitch, float log_scale_min) {
    for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256L) {
        const float* p = yhat + r * pitch;
        float best = -INFINITY; int arg = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float t = p[k] - logf(-logf(u1[r * K + k]));
            if (t > best) { best = t; arg = k; }
        }
        const float m = p[K + arg], ls = fmaxf(p[2 * K + arg], log_scale_min), u = u2[r];
        float x = m + expf(ls) * (logf(u) - logf(1.f - u));
        out[r] = fminf(fmaxf(x, -1.f), 1.f);
    }
}

// masked mean: out = sum(loss * mask) / sum(mask)     (loss_functions.py:43-62)
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* __restrict__ loss, const float* __restrict__ mask, long n, float* out, float* wrow) {
    __shared__ double r1[4], r2[4];
    double a = 0.0, b = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) { float m = mask ? mask[i] : 1.f; a += (double)loss[i] * m; b += m; }
    a = wave_sum_d(a); b = wave_sum_d(b);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = a; r2[threadIdx.x >> 6] = b; }
    __syncthreads();
    const double sa = r1[0] + r1[1] + r1[2] + r1[3], sb = r2[0] + r2[1] + r2[2] + r2[3];
    if (threadIdx.x == 0) *out = (float)(sa / sb);
    if (wrow) for (long i = threadIdx.x; i < n; i += 256) wrow[i] = (mask ? mask[i] : 1.f) / (float)sb;
}

__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ d, const float* __restrict__ gscale, long n) {
    const float g = *gscale;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) d[i] *= g;
}

}  // namespace

extern "C" int viai_weight_norm_fwd(const float* v, const float* g, float* w, float* norm, int rows, int L, void* stream) {
    VIAI_LAUNCH(weight_norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, w, norm, L);
    return viai_launch_status();
}

extern "C" int viai_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm, float* dv, float* dg,
                                    int rows, int L, int accumulate, void* stream) {
    VIAI_LAUNCH(weight_norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dw, v, g, norm, dv, dg, L, accumulate);
    return viai_launch_status();
}

extern "C" int viai_glu_fwd(const float* y, const float* yc, float* z, long rows, int H, void* stream) {
    if (H % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(glu_fwd_kernel, dim3(ew_blocks(rows * (H / 4))), dim3(256), 0, (hipStream_t)stream, y, yc, z, rows, H);
    return viai_launch_status();
}

extern "C" int viai_glu_bwd(const float* dz, const float* y, const float* yc, float* dy, long rows, int H, void* stream) {
    if (H % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(glu_bwd_kernel, dim3(ew_blocks(rows * (H / 4))), dim3(256), 0, (hipStream_t)stream, dz, y, yc, dy, rows, H);
    return viai_launch_status();
}

extern "C" int viai_add_scale(const float* a, const float* b, float* out, float s, long n, void* stream) {
    if (n % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(add_scale_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(a),
                reinterpret_cast<const f32x4*>(b), reinterpret_cast<f32x4*>(out), s, n / 4);
    return viai_launch_status();
}

extern "C" int viai_relu_fwd(const float* a, float* out, long n, void* stream) {
    if (n % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(relu_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(a), reinterpret_cast<f32x4*>(out), n / 4);
    return viai_launch_status();
}

extern "C" int viai_outer_fwd(const float* x, const float* w, const float* b, float* y, long rows, int C, void* stream) {
    if (C % 4 != 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(outer_fwd_kernel, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, w, b, y, rows, C);
    return viai_launch_status();
}

extern "C" int viai_outer_bwd_blocks(long rows) { long b = (rows + 1023) / 1024; if (b > 1024) b = 1024; if (b < 1) b = 1; return (int)b; }

extern "C" int viai_outer_bwd(const float* dy, const float* x, float* part, float* dw, float* db, long rows, int C, int accumulate, void* stream) {
    if (C % 4 != 0) return (int)hipErrorInvalidValue;
    int nb = viai_outer_bwd_blocks(rows);
    long rpb = (rows + nb - 1) / nb;
    VIAI_LAUNCH(outer_bwd_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, part, rows, C, rpb);
    VIAI_LAUNCH(outer_bwd_final_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, nb, C, dw, db, accumulate);
    return viai_launch_status();
}

extern "C" int viai_upsample_fwd(const float* x, const float* w, const float* bias, float* y, int B, int F, int T, int KH, int S, void* stream) {
    if (KH != 3 && KH != 1) return (int)hipErrorInvalidValue;
    long BF = (long)B * F;
    int blocks = ew_blocks(BF * T * S);
    if (KH == 3) VIAI_LAUNCH(upsample_fwd_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, BF, F, T, S);
    else VIAI_LAUNCH(upsample_fwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, BF, F, T, S);
    return viai_launch_status();
}

extern "C" int viai_upsample_bwd_blocks(void) { return 512; }

extern "C" int viai_upsample_bwd(const float* dy, const float* y, const float* x, const float* w, float* part, float* dx, float* dw, float* db,
                                 int B, int F, int T, int KH, int S, int accumulate, void* stream) {
    if ((KH != 3 && KH != 1) || S > 16) return (int)hipErrorInvalidValue;
    long BF = (long)B * F;
    hipStream_t st = (hipStream_t)stream;
    const int nb = 512;
    if (KH == 3) {
        if (dx) VIAI_LAUNCH(upsample_bwd_x_kernel<3>, dim3(ew_blocks(BF * T)), dim3(256), 0, st, dy, y, w, dx, BF, F, T, S);
        VIAI_LAUNCH(upsample_bwd_w_kernel<3>, dim3(nb), dim3(256), 0, st, dy, y, x, part, BF, F, T, S);
    } else {
        if (dx) VIAI_LAUNCH(upsample_bwd_x_kernel<1>, dim3(ew_blocks(BF * T)), dim3(256), 0, st, dy, y, w, dx, BF, F, T, S);
        VIAI_LAUNCH(upsample_bwd_w_kernel<1>, dim3(nb), dim3(256), 0, st, dy, y, x, part, BF, F, T, S);
    }
    VIAI_LAUNCH(upsample_bwd_final_kernel, dim3(1), dim3(64), 0, st, part, nb, KH, S, dw, db, accumulate);
    return viai_launch_status();
}

extern "C" int viai_mol_loss(const float* yhat, const float* y, const float* mask, float* loss_rows, float* wrow, float* loss,
                             float* dyhat, long rows, int pitch, int nr_mix, float num_classes, float log_scale_min, void* stream) {
    if (nr_mix != 10 || pitch < 30) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    // pass 1: per-row losses; masked mean + normalised row weights; pass 2 (if dyhat): gradients with those weights
    VIAI_LAUNCH(mol_loss_kernel<10>, dim3(ew_blocks(rows)), dim3(256), 0, st, yhat, y, (const float*)nullptr, loss_rows, (float*)nullptr, rows, pitch, num_classes, log_scale_min);
    VIAI_LAUNCH(masked_mean_kernel, dim3(1), dim3(256), 0, st, loss_rows, mask, rows, loss, wrow);
    if (dyhat) VIAI_LAUNCH(mol_loss_kernel<10>, dim3(ew_blocks(rows)), dim3(256), 0, st, yhat, y, wrow, loss_rows, dyhat, rows, pitch, num_classes, log_scale_min);
    return viai_launch_status();
}

extern "C" int viai_scale_by_scalar(float* d, const float* gscale, long n, void* stream) {
    VIAI_LAUNCH(scale_rows_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, d, gscale, n);
    return viai_launch_status();
}

extern "C" int viai_mol_sample(const float* yhat, const float* u1, const float* u2, float* out, long rows, int pitch, int nr_mix,
                               float log_scale_min, void* stream) {
    if (nr_mix != 10) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(mol_sample_kernel<10>, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, yhat, u1, u2, out, rows, pitch, log_scale_min);
    return viai_launch_status();
}

// =====================================================================================================
// Incremental synthesis (wavenet.py:237-364, conv.py:17-46).  One time step = first conv + per layer two
// GEMV-batch kernels (gate, then out/skip) + head (two 1x1 layers + MoL sample).  The time index lives in
// DEVICE memory (`step`), ring-buffer slots are derived from it inside the kernels, so every launch has static
// arguments and one step can be captured once into a hipGraph and replayed T times.
// Linearised weights: Wlin[g][j*C + ci] = w[g][ci][j] (conv.py:53-57), taps j = 0..2 read x[t-(2-j)*d].
namespace {

constexpr int WN_MAXB = 8;

// x0[b][:] = cur[b] * w_first + b_first -> ring0[slot(t)];  cur = test_inputs[t] | out[t-1] | 0
// ONE block.  The time index comes either by value (t_arg >= 0: viai_wavenet_synth_run, the host loop knows it) or from device memory
// (t_arg < 0: viai_wavenet_synth_step, replayable from a hipGraph): there this kernel also advances it once all of its threads have
// read it -- `*step` counts the first-conv launches, the other kernels of the time step read t = *step - 1 (a separate one-thread
// tick launch cost 4.3 us per time step).  A load of the index is a full memory round trip (~1.5 us at these tiny grids) in FRONT of
// every address computation of every kernel, which is why the by-value path exists.
__global__ __launch_bounds__(256) void wn_first_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ test_inputs,
                                                       int n_test, const float* __restrict__ out, float* __restrict__ ring, int ring_len,
                                                       int* __restrict__ step, int t_arg, int B, int C, int T) {
    const int t = t_arg >= 0 ? t_arg : *step;
    const int slot = t % ring_len;
    for (int i = threadIdx.x; i < B * C; i += 256) {
        int b = i / C, c = i % C;
        float cur = (t < n_test) ? test_inputs[(size_t)b * n_test + t] : (t > 0 ? out[(size_t)b * T + t - 1] : 0.f);
        ring[((size_t)b * ring_len + slot) * C + c] = cur * w[c] + bias[c];
    }
    if (t_arg < 0) {
        __syncthreads();
        if (threadIdx.x == 0) *step = t + 1;
    }
}

// gate: z[b][h] = tanh(A) * sigmoid(Bv),  A/Bv = rows h / h+H of (Wlin . [x(t-2d); x(t-d); x(t)] + b + Wc . c_t + bc)
// One BLOCK per output pair (h, h+H): the 3C + cin reduction is cut into 16-byte chunks over the 256 threads (reference size:
// 384 + 20 chunks, at most two per thread, all loads of a thread in flight at once), the 2 x NB partial sums meet in LDS.
// (The first version gave a WAVE the whole row: six dependent trips of 10 loads per lane on 64 blocks, 10.8 us per launch.)
template <int NB>
__global__ __launch_bounds__(256) void wn_gate_kernel(const float* __restrict__ ring, int ring_len, int dil, const float* __restrict__ wlin,
                                                      const float* __restrict__ bconv, const float* __restrict__ wc, const float* __restrict__ bc,
                                                      const float* __restrict__ cond, const float* __restrict__ gadd, float* __restrict__ z,
                                                      const int* __restrict__ step, int t_arg, int C, int H, int cin, int T) {
    __shared__ float red[2 * NB][260];
    const int t = t_arg >= 0 ? t_arg : *step - 1;
    const int h = blockIdx.x, tid = threadIdx.x;
    // the epilogue's biases are fetched now, not after the reduction (one memory round trip less on the critical path)
    float ba = 0.f, bg = 0.f;
    if ((tid & 31) == 0 && tid < 32 * NB) {
        ba = bconv[h] + (bc ? bc[h] : 0.f); bg = bconv[h + H] + (bc ? bc[h + H] : 0.f);
        if (gadd != nullptr) { ba += gadd[(size_t)(tid >> 5) * 2 * H + h]; bg += gadd[(size_t)(tid >> 5) * 2 * H + h + H]; }
    }
    const int K = 3 * C;
    const int nq = K / 4, nqc = wc != nullptr ? cin / 4 : 0;
    float a[NB], g[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { a[b] = 0.f; g[b] = 0.f; }
    const float* wa = wlin + (size_t)h * K;
    const float* wg = wlin + (size_t)(h + H) * K;
    for (int q = tid; q < nq + nqc; q += 256) {
        f32x4 va, vg;
        const float* xb;
        size_t xstride;
        bool live = true;
        if (q < nq) {
            const int k = q * 4;
            const int j = k / C, ci = k - j * C;
            const int tt = t - (2 - j) * dil;
            live = tt >= 0;
            va = *reinterpret_cast<const f32x4*>(wa + k); vg = *reinterpret_cast<const f32x4*>(wg + k);
            xb = ring + (size_t)(live ? tt % ring_len : 0) * C + ci;
            xstride = (size_t)ring_len * C;
        } else {
            const int k = (q - nq) * 4;
            va = *reinterpret_cast<const f32x4*>(wc + (size_t)h * cin + k); vg = *reinterpret_cast<const f32x4*>(wc + (size_t)(h + H) * cin + k);
            xb = cond + (size_t)t * cin + k;
            xstride = (size_t)T * cin;
        }
        if (live) {
            f32x4 x[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) x[b] = *reinterpret_cast<const f32x4*>(xb + b * xstride);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                a[b] += x[b][0] * va[0] + x[b][1] * va[1] + x[b][2] * va[2] + x[b][3] * va[3];
                g[b] += x[b][0] * vg[0] + x[b][1] * vg[1] + x[b][2] * vg[2] + x[b][3] * vg[3];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) { red[2 * b][tid] = a[b]; red[2 * b + 1][tid] = g[b]; }
    __syncthreads();
    // 16 lanes per value: lane `part` adds 16 of the 256 partials, four butterfly steps finish the sum (fixed order)
    if (tid < 32 * NB) {
        const int v = tid >> 4, part = tid & 15;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += red[v][part * 16 + ((j + part) & 15)];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float other = __shfl_down(sum, 16, 64);          // the gate half (value 2b + 1) sits 16 lanes up
        if ((tid & 31) == 0) {
            const int b = tid >> 5;
            const float sa = sum + ba, sg = other + bg;
            z[(size_t)b * H + h] = tanhf(sa) * (1.f / (1.f + expf(-sg)));
        }
    }
}

// out/skip: o < C: x_next[b][o] = (Wout[o].z + bout[o] + x_t[b][o]) * sqrt(.5) -> next ring;  o >= C: skip accumulate
template <int NB>
__global__ __launch_bounds__(256) void wn_out_kernel(const float* __restrict__ z, const float* __restrict__ wout, const float* __restrict__ bout,
                                                     const float* __restrict__ wskip, const float* __restrict__ bskip,
                                                     const float* __restrict__ ring, int ring_len, float* __restrict__ next_ring, int next_len,
                                                     float* __restrict__ skips, int first, const int* __restrict__ step, int t_arg, int C, int H, int S) {
    const int t = t_arg >= 0 ? t_arg : *step - 1;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= C + S) return;
    const float* w = o < C ? wout + (size_t)o * H : wskip + (size_t)(o - C) * H;
    // what the epilogue adds (residual x_t / running skip sum, bias) is fetched NOW by lane b for stream b, beside the dot
    // products, instead of after the reduction
    float pre = 0.f;
    if (lane < NB) pre = o < C ? ring[((size_t)lane * ring_len + (t % ring_len)) * C + o] : (first ? 0.f : skips[(size_t)lane * S + (o - C)]);
    const float bias = o < C ? bout[o] : bskip[o - C];
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    for (int k = lane * 4; k < H; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(z + (size_t)b * H + k);
            acc[b] += x[0] * wv[0] + x[1] * wv[1] + x[2] * wv[2] + x[3] * wv[3];
        }
    }
    const float r5 = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float v = wave_sum_dpp(acc[b]);
        const float p = __shfl(pre, b, 64);
        if (lane == 0) {
            if (o < C) {
                if (next_ring) next_ring[((size_t)b * next_len + (t % next_len)) * C + o] = (v + bias + p) * r5;
            } else {
                const float sv = v + bias;
                skips[(size_t)b * S + (o - C)] = first ? sv : (p + sv) * r5;
            }
        }
    }
}

// head: relu -> W1 -> relu -> W2 -> MoL sample with injected uniforms -> out[b][t]; one block of 16 waves per stream.
// At one block per stream the kernel is a chain of load latencies, so everything that does not depend on the previous stage is
// fetched up front (uniforms, biases) and every stage has all of its rows in flight at once: a wave takes 16 rows, each read
// coalesced (16 bytes per lane).  (A row per THREAD, 256 strided loads in sequence: 17.7 us; four rows per trip: 49 us.)
__global__ __launch_bounds__(1024) void wn_head_kernel(const float* __restrict__ skips, const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ u1,
                                                       const float* __restrict__ u2, float* __restrict__ out, float* __restrict__ yhat_dbg,
                                                       const int* __restrict__ step, int t_arg, int S, int OC, int T, float log_scale_min) {
    extern __shared__ __attribute__((aligned(16))) float sm[];          // [S] relu(skips), [S] hidden, [OC] logits, [OC/3 + 1] uniforms
    float* xin = sm; float* hid = sm + S; float* yo = sm + 2 * S; float* us = yo + ((OC + 3) & ~3);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int RB = 16;
    const int K = OC / 3;
    // Stage order = what can be in flight together: W1's rows depend on nothing, so they (and the biases) are requested before the
    // time index or the skip sums arrive; W2's rows are requested as soon as W1's registers are free, i.e. before W1's reductions
    // and the barrier.  A wave holds ONE 16-row batch at a time (16 waves at 128 registers each: holding both layers spilt to scratch,
    // 38 us per launch).  Needs S <= 256 and out_ch <= 256 (the reference: 256 / 30); wn_head_generic_kernel takes anything else.
    const int k0 = lane * 4;
    const int r1 = wave * RB, r2 = wave * RB;
    const float bv1 = b1[min(r1 + (lane & (RB - 1)), S - 1)], bv2 = b2[min(r2 + (lane & (RB - 1)), OC - 1)];
    f32x4 p[RB];
    if (r1 < S && k0 < S) {
#pragma unroll
        for (int r = 0; r < RB; ++r) p[r] = *reinterpret_cast<const f32x4*>(w1 + (size_t)min(r1 + r, S - 1) * S + k0);
    }
    const int t = t_arg >= 0 ? t_arg : *step - 1;
    for (int k = tid; k < S; k += 1024) { float v = skips[(size_t)b * S + k]; xin[k] = v > 0.f ? v : 0.f; }
    if (tid < K) us[tid] = u1[((size_t)b * T + t) * K + tid];
    if (tid == K) us[K] = u2[(size_t)b * T + t];
    __syncthreads();
    // one 16-row batch: acc[r] = row (r0 + r) . x with the row slices already in registers (S <= 256: one 16-byte slice per lane)
    auto batch = [&](const float* __restrict__, const f32x4 (&pre)[RB], const float* x, int, float (&acc)[RB]) {
        const f32x4 xv = k0 < S ? *reinterpret_cast<const f32x4*>(x + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = k0 < S ? pre[r][0] * xv[0] + pre[r][1] * xv[1] + pre[r][2] * xv[2] + pre[r][3] * xv[3] : 0.f;
    };
    auto finish = [&](const float (&acc)[RB], float bv, int h0, int n_out, auto&& emit) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float v = wave_sum_dpp(acc[r]) + __shfl(bv, r, 64);
            if (lane == 0 && h0 + r < n_out) emit(h0 + r, v);
        }
    };
    auto hid_out = [&](int h, float v) { hid[h] = v > 0.f ? v : 0.f; };
    auto logit_out = [&](int o, float v) { yo[o] = v; if (yhat_dbg) yhat_dbg[((size_t)b * T + t) * OC + o] = v; };
    float acc[RB];
    if (r1 < S) batch(w1, p, xin, S, acc);
    if (r2 < OC && k0 < S) {                                   // W2's first batch: in flight across W1's reductions and the barrier
#pragma unroll
        for (int r = 0; r < RB; ++r) p[r] = *reinterpret_cast<const f32x4*>(w2 + (size_t)min(r2 + r, OC - 1) * S + k0);
    }
    if (r1 < S) finish(acc, bv1, r1, S, hid_out);
    __syncthreads();
    if (r2 < OC) { batch(w2, p, hid, OC, acc); finish(acc, bv2, r2, OC, logit_out); }
    __syncthreads();
    if (tid == 0) {
        float best = -INFINITY; int arg = 0;
        for (int k = 0; k < K; ++k) {
            float v = yo[k] - logf(-logf(us[k]));
            if (v > best) { best = v; arg = k; }
        }
        const float m = yo[K + arg], ls = fmaxf(yo[2 * K + arg], log_scale_min), u = us[K];
        float x = m + expf(ls) * (logf(u) - logf(1.f - u));
        out[(size_t)b * T + t] = fminf(fmaxf(x, -1.f), 1.f);
    }
}

// any S / out_ch: a row per thread (the first version of the head)
__global__ __launch_bounds__(256) void wn_head_generic_kernel(const float* __restrict__ skips, const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ u1,
                                                              const float* __restrict__ u2, float* __restrict__ out, float* __restrict__ yhat_dbg,
                                                              const int* __restrict__ step, int t_arg, int S, int OC, int T, float log_scale_min) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xin = sm; float* hid = sm + S; float* yo = sm + 2 * S;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int t = t_arg >= 0 ? t_arg : *step - 1;
    for (int k = tid; k < S; k += 256) { float v = skips[(size_t)b * S + k]; xin[k] = v > 0.f ? v : 0.f; }
    __syncthreads();
    for (int h = tid; h < S; h += 256) {
        float acc = b1[h];
        for (int k = 0; k < S; ++k) acc += w1[(size_t)h * S + k] * xin[k];
        hid[h] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    for (int o = tid; o < OC; o += 256) {
        float acc = b2[o];
        for (int k = 0; k < S; ++k) acc += w2[(size_t)o * S + k] * hid[k];
        yo[o] = acc;
        if (yhat_dbg) yhat_dbg[((size_t)b * T + t) * OC + o] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        const int K = OC / 3;
        float best = -INFINITY; int arg = 0;
        for (int k = 0; k < K; ++k) {
            float v = yo[k] - logf(-logf(u1[((size_t)b * T + t) * K + k]));
            if (v > best) { best = v; arg = k; }
        }
        const float m = yo[K + arg], ls = fmaxf(yo[2 * K + arg], log_scale_min), u = u2[(size_t)b * T + t];
        float x = m + expf(ls) * (logf(u) - logf(1.f - u));
        out[(size_t)b * T + t] = fminf(fmaxf(x, -1.f), 1.f);
    }
}

// ---- fused stages (ABI 7): ONE dependent launch per layer instead of two -----------------------------------------------------------
// A time step of the plain form is a chain of 2 L + 2 dependent launches: gate_l needs all of x_l(t), which out_{l-1} produces from all
// of z_{l-1}.  But x_l(t) = (Wo_{l-1} z_{l-1} + bo_{l-1} + x_{l-1}(t)) sqrt(.5) is LINEAR in z_{l-1}, and gate_l uses it only through
// the current tap Wc_l^2 x_l(t), so
//     Wc_l^2 x_l(t) = [sqrt(.5) Wc_l^2 Wo_{l-1}] z_{l-1} + [sqrt(.5) Wc_l^2] x_{l-1}(t) + sqrt(.5) Wc_l^2 bo_{l-1}
// and gate_l can start from what stage l - 1 left behind: z_{l-1} and x_{l-1}(t).  The host builds the extended rows
// [Wc^0 | Wc^1 | sqrt(.5) Wc^2 | sqrt(.5) Wc^2 Wo_{l-1}] (K = 3 C + H) and the folded bias once; stage l then runs, side by side in
// one launch, (A) gate_l -> z_l from the two past taps (ring_l), x_{l-1}(t) (ring_{l-1}) and z_{l-1}, and (B) the out / skip rows of
// layer l - 1 (x_l(t) -> ring_l for the FUTURE time steps' past taps, the running skip sum).  Stage 0 forms x_0(t) = first conv inline.
// The last layer's skip row product moves into the head.  L + 1 dependent launches per time step; z is double-buffered (stage l
// writes z_l while its B blocks still read z_{l-1}).  Reassociation only: sums agree with the plain form to fp32 rounding.
struct WnStage {
    const float* ring; int ring_len, dil;          // x_l at the past time steps
    const float* ring_prev; int prev_len;          // x_{l-1}, current slot written by stage l - 1 (l > 0)
    const float* w; const float* bias;             // [G][K] extended rows, [G] folded bias
    const float* wc; const float* cond; const float* gadd;
    const float* z_in; float* z_out;
    const float* w_first; const float* b_first; const float* test_inputs; int n_test; const float* out;   // stage 0
    const float* w_out; const float* b_out; const float* w_skip; const float* b_skip;                      // layer l - 1 (B blocks)
    float* ring_w; float* skips; int first_skip;
    const int* step; int t_arg, l, C, H, S, cin, T, B;
};

__global__ void wn_tick_kernel(int* step) { *step += 1; }

template <int NB>
__global__ __launch_bounds__(256) void wn_stage_kernel(const WnStage a) {
    __shared__ float red[2 * NB][260];
    const int t = a.t_arg >= 0 ? a.t_arg : *a.step - 1;
    const int tid = threadIdx.x, C = a.C, H = a.H;
    if ((int)blockIdx.x >= H) {
        // ---------------------------------------------------------------- B: what layer l - 1 still owes (or the first conv)
        const int bid = blockIdx.x - H;
        if (a.l == 0) {
            for (int i = tid; i < NB * C; i += 256) {
                const int b = i / C, c = i - b * C;
                const float cur = (t < a.n_test) ? a.test_inputs[(size_t)b * a.n_test + t] : (t > 0 ? a.out[(size_t)b * a.T + t - 1] : 0.f);
                a.ring_w[((size_t)b * a.ring_len + (t % a.ring_len)) * C + c] = cur * a.w_first[c] + a.b_first[c];
            }
            return;
        }
        const int o = bid * 4 + (tid >> 6), lane = tid & 63, S = a.S;
        if (o >= C + S) return;
        const float* w = o < C ? a.w_out + (size_t)o * H : a.w_skip + (size_t)(o - C) * H;
        float pre = 0.f;
        if (lane < NB) pre = o < C ? a.ring_prev[((size_t)lane * a.prev_len + (t % a.prev_len)) * C + o] : (a.first_skip ? 0.f : a.skips[(size_t)lane * S + (o - C)]);
        const float bias = o < C ? a.b_out[o] : a.b_skip[o - C];
        float acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = 0.f;
        for (int k = lane * 4; k < H; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(a.z_in + (size_t)b * H + k);
                acc[b] += x[0] * wv[0] + x[1] * wv[1] + x[2] * wv[2] + x[3] * wv[3];
            }
        }
        const float r5 = 0.70710678118654752f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float v = wave_sum_dpp(acc[b]);
            const float p = __shfl(pre, b, 64);
            if (lane == 0) {
                if (o < C) a.ring_w[((size_t)b * a.ring_len + (t % a.ring_len)) * C + o] = (v + bias + p) * r5;
                else { const float sv = v + bias; a.skips[(size_t)b * S + (o - C)] = a.first_skip ? sv : (p + sv) * r5; }
            }
        }
        return;
    }
    // -------------------------------------------------------------------- A: gate pair (h, h + H) of layer l
    const int h = blockIdx.x;
    float ba = 0.f, bg = 0.f;
    if ((tid & 31) == 0 && tid < 32 * NB) {
        ba = a.bias[h]; bg = a.bias[h + H];
        if (a.gadd != nullptr) { ba += a.gadd[(size_t)(tid >> 5) * 2 * H + h]; bg += a.gadd[(size_t)(tid >> 5) * 2 * H + h + H]; }
    }
    const int K = 3 * C + (a.l > 0 ? H : 0);
    const int nq = K / 4, nqc = a.wc != nullptr ? a.cin / 4 : 0;
    float sa[NB], sg[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { sa[b] = 0.f; sg[b] = 0.f; }
    const float* wa = a.w + (size_t)h * K;
    const float* wg = a.w + (size_t)(h + H) * K;
    for (int q = tid; q < nq + nqc; q += 256) {
        f32x4 va, vg, x[NB];
        bool live = true;
        if (q < nq) {
            const int k = q * 4;
            va = *reinterpret_cast<const f32x4*>(wa + k); vg = *reinterpret_cast<const f32x4*>(wg + k);
            if (k < 2 * C) {                                        // past taps: x_l(t - 2 d), x_l(t - d)
                const int j = k / C, ci = k - j * C;
                const int tt = t - (2 - j) * a.dil;
                live = tt >= 0;
                if (live) {
                    const float* xb = a.ring + (size_t)(tt % a.ring_len) * C + ci;
#pragma unroll
                    for (int b = 0; b < NB; ++b) x[b] = *reinterpret_cast<const f32x4*>(xb + (size_t)b * a.ring_len * C);
                }
            } else if (k < 3 * C) {                                 // current tap
                const int ci = k - 2 * C;
                if (a.l == 0) {                                     // x_0(t): the first conv, formed here
                    const f32x4 wf = *reinterpret_cast<const f32x4*>(a.w_first + ci), bf = *reinterpret_cast<const f32x4*>(a.b_first + ci);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float cur = (t < a.n_test) ? a.test_inputs[(size_t)b * a.n_test + t] : (t > 0 ? a.out[(size_t)b * a.T + t - 1] : 0.f);
                        x[b] = cur * wf + bf;
                    }
                } else {                                            // x_{l-1}(t), against sqrt(.5) Wc^2
                    const float* xb = a.ring_prev + (size_t)(t % a.prev_len) * C + ci;
#pragma unroll
                    for (int b = 0; b < NB; ++b) x[b] = *reinterpret_cast<const f32x4*>(xb + (size_t)b * a.prev_len * C);
                }
            } else {                                                // z_{l-1}, against sqrt(.5) Wc^2 Wo_{l-1}
                const int kz = k - 3 * C;
#pragma unroll
                for (int b = 0; b < NB; ++b) x[b] = *reinterpret_cast<const f32x4*>(a.z_in + (size_t)b * H + kz);
            }
        } else {
            const int k = (q - nq) * 4;
            va = *reinterpret_cast<const f32x4*>(a.wc + (size_t)h * a.cin + k); vg = *reinterpret_cast<const f32x4*>(a.wc + (size_t)(h + H) * a.cin + k);
            const float* xb = a.cond + (size_t)t * a.cin + k;
#pragma unroll
            for (int b = 0; b < NB; ++b) x[b] = *reinterpret_cast<const f32x4*>(xb + (size_t)b * a.T * a.cin);
        }
        if (live) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                sa[b] += x[b][0] * va[0] + x[b][1] * va[1] + x[b][2] * va[2] + x[b][3] * va[3];
                sg[b] += x[b][0] * vg[0] + x[b][1] * vg[1] + x[b][2] * vg[2] + x[b][3] * vg[3];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) { red[2 * b][tid] = sa[b]; red[2 * b + 1][tid] = sg[b]; }
    __syncthreads();
    if (tid < 32 * NB) {
        const int v = tid >> 4, part = tid & 15;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += red[v][part * 16 + ((j + part) & 15)];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float other = __shfl_down(sum, 16, 64);
        if ((tid & 31) == 0) {
            const int b = tid >> 5;
            const float va_ = sum + ba, vg_ = other + bg;
            a.z_out[(size_t)b * H + h] = tanhf(va_) * (1.f / (1.f + expf(-vg_)));
        }
    }
}

// head of the fused form: first the LAST layer's skip rows (skips <- (skips + Ws z + bs) sqrt(.5), relu'd into LDS), then wn_head_kernel's
// two layers and the sampler.  16 waves x 16 rows, S <= 256, H <= 256, out_ch <= 256.
__global__ __launch_bounds__(1024) void wn_head_fused_kernel(const float* __restrict__ skips, const float* __restrict__ z, const float* __restrict__ wsk,
                                                             const float* __restrict__ bsk, int first_skip, int H,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ u1,
                                                             const float* __restrict__ u2, float* __restrict__ out, float* __restrict__ yhat_dbg,
                                                             const int* __restrict__ step, int t_arg, int S, int OC, int T, float log_scale_min) {
    extern __shared__ __attribute__((aligned(16))) float sm[];          // [S] relu(skips), [S] hidden, [OC] logits, [OC/3 + 1] uniforms
    float* xin = sm; float* hid = sm + S; float* yo = sm + 2 * S; float* us = yo + ((OC + 3) & ~3);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int RB = 16;
    const int K = OC / 3;
    const int k0 = lane * 4;
    const int r0 = wave * RB;
    f32x4 p[RB];
    if (r0 < S && k0 < H) {
#pragma unroll
        for (int r = 0; r < RB; ++r) p[r] = *reinterpret_cast<const f32x4*>(wsk + (size_t)min(r0 + r, S - 1) * H + k0);
    }
    const float bvs = bsk[min(r0 + (lane & (RB - 1)), S - 1)];
    const float bv1 = b1[min(r0 + (lane & (RB - 1)), S - 1)], bv2 = b2[min(r0 + (lane & (RB - 1)), OC - 1)];
    const float prev = (!first_skip && r0 < S) ? skips[(size_t)b * S + min(r0 + (lane & (RB - 1)), S - 1)] : 0.f;
    const int t = t_arg >= 0 ? t_arg : *step - 1;
    if (tid < K) us[tid] = u1[((size_t)b * T + t) * K + tid];
    if (tid == K) us[K] = u2[(size_t)b * T + t];
    float acc[RB];
    const float r5 = 0.70710678118654752f;
    if (r0 < S) {
        const f32x4 zv = k0 < H ? *reinterpret_cast<const f32x4*>(z + (size_t)b * H + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = k0 < H ? p[r][0] * zv[0] + p[r][1] * zv[1] + p[r][2] * zv[2] + p[r][3] * zv[3] : 0.f;
    }
    if (r0 < S && k0 < S) {                                    // W1's rows: in flight across the skip reductions
#pragma unroll
        for (int r = 0; r < RB; ++r) p[r] = *reinterpret_cast<const f32x4*>(w1 + (size_t)min(r0 + r, S - 1) * S + k0);
    }
    if (r0 < S) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float sv = wave_sum_dpp(acc[r]) + __shfl(bvs, r, 64);
            const float pv = __shfl(prev, r, 64);
            if (lane == 0 && r0 + r < S) { const float v = first_skip ? sv : (pv + sv) * r5; xin[r0 + r] = v > 0.f ? v : 0.f; }
        }
    }
    __syncthreads();
    if (r0 < S) {
        const f32x4 xv = k0 < S ? *reinterpret_cast<const f32x4*>(xin + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = k0 < S ? p[r][0] * xv[0] + p[r][1] * xv[1] + p[r][2] * xv[2] + p[r][3] * xv[3] : 0.f;
    }
    if (r0 < OC && k0 < S) {
#pragma unroll
        for (int r = 0; r < RB; ++r) p[r] = *reinterpret_cast<const f32x4*>(w2 + (size_t)min(r0 + r, OC - 1) * S + k0);
    }
    if (r0 < S) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float v = wave_sum_dpp(acc[r]) + __shfl(bv1, r, 64);
            if (lane == 0 && r0 + r < S) hid[r0 + r] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    if (r0 < OC) {
        const f32x4 xv = k0 < S ? *reinterpret_cast<const f32x4*>(hid + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = k0 < S ? p[r][0] * xv[0] + p[r][1] * xv[1] + p[r][2] * xv[2] + p[r][3] * xv[3] : 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float v = wave_sum_dpp(acc[r]) + __shfl(bv2, r, 64);
            if (lane == 0 && r0 + r < OC) { yo[r0 + r] = v; if (yhat_dbg) yhat_dbg[((size_t)b * T + t) * OC + r0 + r] = v; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        float best = -INFINITY; int arg = 0;
        for (int k = 0; k < K; ++k) {
            float v = yo[k] - logf(-logf(us[k]));
            if (v > best) { best = v; arg = k; }
        }
        const float m = yo[K + arg], ls = fmaxf(yo[2 * K + arg], log_scale_min), u = us[K];
        float x = m + expf(ls) * (logf(u) - logf(1.f - u));
        out[(size_t)b * T + t] = fminf(fmaxf(x, -1.f), 1.f);
    }
}

// ---- the head as three row-parallel launches -------------------------------------------------------------------------------------
// wn_head_fused_kernel runs one block per stream: each of the 8 blocks pulls the 0.5 MB of head weights through ONE compute unit (25 us per
// time step, rocprofv3).  Spread over the rows instead -- a wave per row, every weight row read once for all streams, as the stages do:
//   rows kernel, mode 0: skips <- relu((skips + Ws z + bs) sqrt(.5))   (the last layer's skip rows; in place)
//   rows kernel, mode 1: hid   <- relu(W1 skips + b1)                   (hid lives in the z buffer the last stage did not write)
//   wn_head_sample_kernel: logits of stream b = W2 hid_b + b2 and the mixture-of-logistics sample (mixture.py:125-153, injected uniforms)
template <int NB>
__global__ __launch_bounds__(256) void wn_head_rows_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ x, int K,
                                                           float* __restrict__ out, int nrows, int mode, int first_skip) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= nrows) return;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    float pre = 0.f;
    if (mode == 0 && !first_skip && lane < NB) pre = out[(size_t)lane * nrows + o];
    const float bv = bias[o];
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)o * K + k);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + k);
            acc[b] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
    }
    const float r5 = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float v = wave_sum_dpp(acc[b]) + bv;
        const float p = __shfl(pre, b, 64);
        if (lane == 0) {
            const float y = mode == 0 ? (first_skip ? v : (p + v) * r5) : v;
            out[(size_t)b * nrows + o] = y > 0.f ? y : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void wn_head_sample_kernel(const float* __restrict__ hid, const float* __restrict__ w2, const float* __restrict__ b2,
                                                             const float* __restrict__ u1, const float* __restrict__ u2, float* __restrict__ out,
                                                             float* __restrict__ yhat_dbg, const int* __restrict__ step, int t_arg, int S, int OC, int T,
                                                             float log_scale_min) {
    __shared__ float yo[256];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = t_arg >= 0 ? t_arg : *step - 1;
    const int K3 = OC / 3;
    for (int o = wave; o < OC; o += 4) {
        float acc = 0.f;
        for (int k = lane * 4; k < S; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w2 + (size_t)o * S + k);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(hid + (size_t)b * S + k);
            acc += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
        const float v = wave_sum_dpp(acc) + b2[o];
        if (lane == 0) { yo[o] = v; if (yhat_dbg) yhat_dbg[((size_t)b * T + t) * OC + o] = v; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float best = -INFINITY; int arg = 0;
        for (int k = 0; k < K3; ++k) {
            const float v = yo[k] - logf(-logf(u1[((size_t)b * T + t) * K3 + k]));
            if (v > best) { best = v; arg = k; }
        }
        const float m = yo[K3 + arg], ls = fmaxf(yo[2 * K3 + arg], log_scale_min), u = u2[(size_t)b * T + t];
        const float x = m + expf(ls) * (logf(u) - logf(1.f - u));
        out[(size_t)b * T + t] = fminf(fmaxf(x, -1.f), 1.f);
    }
}

bool wn_fused_ok(const viai_wn_synth* s) {
    if (!s->fused || s->z2 == nullptr) return false;
    if (s->S > 256 || s->out_ch > 256 || s->G / 2 > 256) return false;
    for (int l = 0; l < s->n_layers; ++l) if (s->layers[l].w_stage == nullptr || s->layers[l].b_stage == nullptr) return false;
    return true;
}

template <int NB>
int wn_step_fused_impl(const viai_wn_synth* s, int t_arg, hipStream_t st) {
    const int C = s->C, H = s->G / 2, S = s->S, n = s->n_layers;
    const viai_wn_layer* L = s->layers;
    if (t_arg < 0) VIAI_LAUNCH(wn_tick_kernel, dim3(1), dim3(1), 0, st, s->step);
    float* zb[2] = {s->z, s->z2};
    for (int l = 0; l < n; ++l) {
        WnStage a{};
        a.ring = L[l].ring; a.ring_len = L[l].ring_len; a.dil = L[l].dilation;
        a.w = L[l].w_stage; a.bias = L[l].b_stage; a.wc = L[l].w_c; a.cond = s->cond; a.gadd = L[l].g_add;
        a.z_out = zb[l & 1]; a.z_in = zb[(l & 1) ^ 1];
        a.w_first = s->w_first; a.b_first = s->b_first; a.test_inputs = s->test_inputs; a.n_test = s->n_test; a.out = s->out;
        a.ring_w = L[l].ring; a.skips = s->skips;
        if (l > 0) {
            a.ring_prev = L[l - 1].ring; a.prev_len = L[l - 1].ring_len;
            a.w_out = L[l - 1].w_out; a.b_out = L[l - 1].b_out; a.w_skip = L[l - 1].w_skip; a.b_skip = L[l - 1].b_skip;
            a.first_skip = (l == 1) ? 1 : 0;
        }
        a.step = s->step; a.t_arg = t_arg; a.l = l; a.C = C; a.H = H; a.S = S; a.cin = s->cin; a.T = s->T; a.B = s->B;
        const int nb = H + (l == 0 ? 1 : (C + S + 3) / 4);
        VIAI_LAUNCH(wn_stage_kernel<NB>, dim3(nb), dim3(256), 0, st, a);
    }
    if (S <= H && s->out_ch <= 256) {
        float* zlast = zb[(n - 1) & 1];
        float* hid = zb[n & 1];                    // (B, S) fits the (B, H) buffer; the next time step's stage 0 overwrites it
        VIAI_LAUNCH(wn_head_rows_kernel<NB>, dim3((S + 3) / 4), dim3(256), 0, st, L[n - 1].w_skip, L[n - 1].b_skip, (const float*)zlast, H, s->skips, S, 0, n == 1 ? 1 : 0);
        VIAI_LAUNCH(wn_head_rows_kernel<NB>, dim3((S + 3) / 4), dim3(256), 0, st, s->w_l1, s->b_l1, (const float*)s->skips, S, hid, S, 1, 0);
        VIAI_LAUNCH(wn_head_sample_kernel, dim3(s->B), dim3(256), 0, st, (const float*)hid, s->w_l2, s->b_l2, s->u1, s->u2, s->out, s->yhat_dbg, s->step, t_arg, S,
                    s->out_ch, s->T, s->log_scale_min);
        return viai_launch_status();
    }
    VIAI_LAUNCH(wn_head_fused_kernel, dim3(s->B), dim3(1024), (2 * S + ((s->out_ch + 3) & ~3) + s->out_ch / 3 + 1) * sizeof(float), st,
                s->skips, zb[(n - 1) & 1], L[n - 1].w_skip, L[n - 1].b_skip, n == 1 ? 1 : 0, H, s->w_l1, s->b_l1, s->w_l2, s->b_l2, s->u1, s->u2,
                s->out, s->yhat_dbg, s->step, t_arg, S, s->out_ch, s->T, s->log_scale_min);
    return viai_launch_status();
}

template <int NB>
int wn_step_impl(const viai_wn_synth* s, int t_arg, hipStream_t st) {
    const int C = s->C, H = s->G / 2, S = s->S;
    const viai_wn_layer* L = s->layers;
    VIAI_LAUNCH(wn_first_kernel, dim3(1), dim3(256), 0, st, s->w_first, s->b_first, s->test_inputs, s->n_test, s->out,
                L[0].ring, L[0].ring_len, s->step, t_arg, s->B, C, s->T);
    for (int l = 0; l < s->n_layers; ++l) {
        VIAI_LAUNCH(wn_gate_kernel<NB>, dim3(H), dim3(256), 0, st, L[l].ring, L[l].ring_len, L[l].dilation, L[l].w_conv, L[l].b_conv,
                    L[l].w_c, L[l].b_c, s->cond, L[l].g_add, s->z, s->step, t_arg, C, H, s->cin, s->T);
        const bool last = (l == s->n_layers - 1);
        VIAI_LAUNCH(wn_out_kernel<NB>, dim3((C + S + 3) / 4), dim3(256), 0, st, s->z, L[l].w_out, L[l].b_out, L[l].w_skip, L[l].b_skip,
                    L[l].ring, L[l].ring_len, last ? (float*)nullptr : L[l + 1].ring, last ? 1 : L[l + 1].ring_len, s->skips, l == 0 ? 1 : 0,
                    s->step, t_arg, C, H, S);
    }
    if (S <= 256 && s->out_ch <= 256)
        VIAI_LAUNCH(wn_head_kernel, dim3(s->B), dim3(1024), (2 * S + ((s->out_ch + 3) & ~3) + s->out_ch / 3 + 1) * sizeof(float), st, s->skips, s->w_l1, s->b_l1, s->w_l2, s->b_l2,
                    s->u1, s->u2, s->out, s->yhat_dbg, s->step, t_arg, S, s->out_ch, s->T, s->log_scale_min);
    else
        VIAI_LAUNCH(wn_head_generic_kernel, dim3(s->B), dim3(256), (2 * S + s->out_ch) * sizeof(float), st, s->skips, s->w_l1, s->b_l1, s->w_l2, s->b_l2,
                    s->u1, s->u2, s->out, s->yhat_dbg, s->step, t_arg, S, s->out_ch, s->T, s->log_scale_min);
    return viai_launch_status();
}

bool wn_valid(const viai_wn_synth* s) {
    return s && s->B >= 1 && s->B <= WN_MAXB && s->C % 4 == 0 && (s->G / 2) % 4 == 0 && s->cin % 4 == 0 && s->S % 4 == 0 && s->out_ch % 3 == 0 && s->n_layers >= 1;
}

int wn_step(const viai_wn_synth* s, int t_arg, hipStream_t st) {
    if (wn_fused_ok(s)) {
        switch (s->B) {
        case 1: return wn_step_fused_impl<1>(s, t_arg, st);
        case 2: return wn_step_fused_impl<2>(s, t_arg, st);
        case 4: return wn_step_fused_impl<4>(s, t_arg, st);
        case 8: return wn_step_fused_impl<8>(s, t_arg, st);
        default: return (int)hipErrorInvalidValue;
        }
    }
    switch (s->B) {
    case 1: return wn_step_impl<1>(s, t_arg, st);
    case 2: return wn_step_impl<2>(s, t_arg, st);
    case 4: return wn_step_impl<4>(s, t_arg, st);
    case 8: return wn_step_impl<8>(s, t_arg, st);
    default: return (int)hipErrorInvalidValue;
    }
}

}  // namespace

extern "C" int viai_wavenet_synth_step(const viai_wn_synth* s, void* stream) {
    if (!wn_valid(s)) return (int)hipErrorInvalidValue;
    return wn_step(s, -1, (hipStream_t)stream);
}

extern "C" int viai_wavenet_synth_run(const viai_wn_synth* s, int t0, int n_steps, void* stream) {
    if (!wn_valid(s) || t0 < 0 || n_steps < 0 || t0 + n_steps > s->T) return (int)hipErrorInvalidValue;
    for (int t = t0; t < t0 + n_steps; ++t) {
        const int e = wn_step(s, t, (hipStream_t)stream);
        if (e != 0) return e;
    }
    return 0;
}
