// Scalar losses (mean reduction), the inpainting mask, and the flat-arena Adam
// step for gfx950.  All HBM streaming; reductions are two-stage and ordered
// (deterministic).  Reference: loss_functions.py:79-104 (GANLoss = BCELoss /
// MSELoss vs an expanded scalar label), train_whole_sync.py:111 (L1 metric),
// utils/util.py:149-150 (optimizer_G / optimizer_D = torch.optim.Adam).
#include "viai_common.h"
#include "viai_internal.h"

namespace {

enum { L_BCE = 0, L_MSE = 1, L_L1 = 2 };

__device__ __forceinline__ float loss_elem(int kind, float a, float b_or_t) {
    if (kind == L_BCE) {           // torch BCELoss: logs clamped at -100
        float lp = fmaxf(logf(a), -100.f), l1p = fmaxf(logf(1.f - a), -100.f);
        return -(b_or_t * lp + (1.f - b_or_t) * l1p);
    }
    if (kind == L_MSE) { float d = a - b_or_t; return d * d; }
    return fabsf(a - b_or_t);
}

template <int KIND>
__global__ __launch_bounds__(256) void loss_part_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float target, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L)
        s += loss_elem(KIND, a[i], b ? b[i] : target);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ part, int nb, long n, float* loss) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += (double)part[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
}

template <int KIND>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float target, long n, const float* __restrict__ gscale,
                                                       float* __restrict__ da) {
    const float gs = (gscale ? *gscale : 1.f) / (float)n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        float x = a[i], t = b ? b[i] : target, g;
        if (KIND == L_BCE) g = (x - t) / fmaxf((1.f - x) * x, 1e-12f);      // torch binary_cross_entropy_backward
        else if (KIND == L_MSE) g = 2.f * (x - t);
        else g = (x > t) ? 1.f : ((x < t) ? -1.f : 0.f);
        da[i] = g * gs;
    }
}

__global__ void mask_mul_kernel(const float* __restrict__ s, const float* __restrict__ mask, float* __restrict__ out,
                                int N, int F, int T) {
    const long total = (long)N * F * T;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int t = (int)(i % T); int n = (int)(i / ((long)F * T));
        out[i] = s[i] * mask[(size_t)n * T + t];
    }
}

__global__ void adam_tick_kernel(double* state, double beta1, double beta2) {
    state[0] += 1.0;
    state[2] *= beta1;
    state[3] *= beta2;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const double* __restrict__ state,
                                                   float beta1, float beta2, float eps, float gscale) {
    const double bc1 = 1.0 - state[2], bc2 = 1.0 - state[3];
    const float step_size = (float)(state[1] / bc1);
    const float bc2s = (float)sqrt(bc2);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        float gi = g[i] * gscale;
        float mi = m[i] * beta1 + (1.f - beta1) * gi;        // exp_avg.lerp_(grad, 1-beta1)
        float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;   // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2s + eps;
        p[i] -= step_size * (mi / denom);
    }
}

__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}

inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

template <int KIND>
int loss_fwd(const float* a, const float* b, float target, long n, float* part, float* loss, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    int nb = viai_reduce_blocks(n);
    VIAI_LAUNCH(loss_part_kernel<KIND>, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, target, n, part);
    VIAI_LAUNCH(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, n, loss);
    return viai_launch_status();
}

template <int KIND>
int loss_bwd(const float* a, const float* b, float target, long n, const float* gscale, float* da, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(loss_bwd_kernel<KIND>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, target, n, gscale, da);
    return viai_launch_status();
}

}  // namespace

extern "C" int viai_reduce_blocks(long n) {
    long b = (n + 4095) / 4096;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int viai_bce_fwd(const float* p, float t, long n, float* part, float* loss, void* s) { return loss_fwd<L_BCE>(p, nullptr, t, n, part, loss, s); }
extern "C" int viai_bce_bwd(const float* p, float t, long n, const float* gs, float* dp, void* s) { return loss_bwd<L_BCE>(p, nullptr, t, n, gs, dp, s); }
extern "C" int viai_mse_fwd(const float* p, float t, long n, float* part, float* loss, void* s) { return loss_fwd<L_MSE>(p, nullptr, t, n, part, loss, s); }
extern "C" int viai_mse_bwd(const float* p, float t, long n, const float* gs, float* dp, void* s) { return loss_bwd<L_MSE>(p, nullptr, t, n, gs, dp, s); }
extern "C" int viai_l1_fwd(const float* a, const float* b, long n, float* part, float* loss, void* s) { return loss_fwd<L_L1>(a, b, 0.f, n, part, loss, s); }
extern "C" int viai_l1_bwd(const float* a, const float* b, long n, const float* gs, float* da, void* s) { return loss_bwd<L_L1>(a, b, 0.f, n, gs, da, s); }

// the six loss scalars of a G+D step in ONE launch (they were ~10 one-element torch kernels at the two ends of the step's critical path):
//   out[0] = 0.5 (d_fake + d_real)   out[1] = g_gan + lambda_l1 l1 (+ lambda_c contrast)   out[2] = g_gan   out[3] = l1   out[4] = d_real
//   out[5] = contrast (untouched when contrast == NULL)
__global__ void step_scalars_kernel(const float* d_real, const float* d_fake, const float* g_gan, const float* l1, const float* contrast,
                                    float lambda_l1, float lambda_c, float* out) {
    const float dr = *d_real, df = *d_fake, gg = *g_gan, l = *l1;
    float lg = gg + lambda_l1 * l;
    if (contrast != nullptr) { const float c = *contrast; lg = lg + lambda_c * c; out[5] = c; }
    out[0] = 0.5f * (df + dr); out[1] = lg; out[2] = gg; out[3] = l; out[4] = dr;
}
extern "C" int viai_step_scalars(const float* d_real, const float* d_fake, const float* g_gan, const float* l1, const float* contrast,
                                 float lambda_l1, float lambda_c, float* out, void* stream) {
    if (!d_real || !d_fake || !g_gan || !l1 || !out) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(step_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_real, d_fake, g_gan, l1, contrast, lambda_l1, lambda_c, out);
    return viai_launch_status();
}

extern "C" int viai_mask_mul(const float* s, const float* mask, float* out, int N, int F, int T, void* stream) {
    long total = (long)N * F * T;
    VIAI_LAUNCH(mask_mul_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, s, mask, out, N, F, T);
    return viai_launch_status();
}

extern "C" int viai_adam_step(float* p, const float* g, float* m, float* v, long n, double* state,
                              double beta1, double beta2, double eps, float grad_scale, void* stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
    VIAI_LAUNCH(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, state, (float)beta1, (float)beta2, (float)eps, grad_scale);
    return viai_launch_status();
}

// debug aid for the f16x2 conv path: counts[0] += #{ |x_i| > limit }, counts[1] = max(counts[1], bits(max |x_i|)), counts[2] += #{non-finite}
namespace {
__global__ __launch_bounds__(256) void range_count_kernel(const float* __restrict__ x, long n, float limit, unsigned* __restrict__ counts) {
    unsigned over = 0, bad = 0;
    float mx = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const float a = fabsf(x[i]);
        if (!(a <= 3.4e38f)) ++bad; else { mx = fmaxf(mx, a); if (a > limit) ++over; }
    }
    over = (unsigned)wave_sum((float)over); bad = (unsigned)wave_sum((float)bad);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) {
        if (over) atomicAdd(&counts[0], over);
        if (mx > 0.f) atomicMax(&counts[1], __float_as_uint(mx));
        if (bad) atomicAdd(&counts[2], bad);
    }
}
}  // namespace

extern "C" int viai_range_count(const float* x, long n, float limit, unsigned* counts, void* stream) {
    if (n <= 0) return 0;
    long b = (n + 255) / 256; if (b > 1024) b = 1024;
    VIAI_LAUNCH(range_count_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, n, limit, counts);
    return viai_launch_status();
}

extern "C" int viai_axpy(float a, const float* x, float* y, long n, void* stream) {
    VIAI_LAUNCH(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, x, y, n);
    return viai_launch_status();
}
