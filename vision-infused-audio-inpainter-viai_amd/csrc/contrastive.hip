// L2ContrastiveLoss (reference loss_functions.py:107-148) on gfx950: pairwise L2 distance matrix between two
// sets of n embeddings (d = 256 in VIAI), hinge^2 on the off-diagonal, diag^2 on the diagonal, / (2n).
// n is a few hundred: one wave per (a, b) pair for the distances, one block for the scalar, one block per
// row / column for the gradients (deterministic, no atomics).
#include "viai_common.h"
#include "viai_internal.h"

namespace {

// scores[a][b] = || f1[a] - f2[b] ||_2
__global__ __launch_bounds__(256) void l2c_scores_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         float* __restrict__ scores, int n, int d) {
    const int a = blockIdx.x;
    const int b = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= n) return;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) { float t = f1[(size_t)a * d + k] - f2[(size_t)b * d + k]; s += t * t; }
    s = wave_sum(s);
    if (lane == 0) scores[(size_t)a * n + b] = sqrtf(s);
}

__global__ __launch_bounds__(256) void l2c_loss_kernel(const float* __restrict__ scores, int n, float margin,
                                                       int max_violation, float* loss) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int a = threadIdx.x; a < n; a += 256) {
        float best = 0.f; double row = 0.0;
        for (int b = 0; b < n; ++b) {
            float sc = scores[(size_t)a * n + b];
            if (b == a) { acc += (double)sc * sc; continue; }
            float c = fmaxf(margin - sc, 0.f);
            if (max_violation) best = fmaxf(best, c); else row += (double)c * c;
        }
        acc += max_violation ? (double)best * best : row;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)((red[0] + red[1] + red[2] + red[3]) / (2.0 * n));
}

// coefficient d loss / d scores[a][b] divided by scores[a][b]  (0 where the distance is 0)
__device__ __forceinline__ float l2c_w(const float* scores, int n, int a, int b, float margin, int max_violation, int argmax_a) {
    float sc = scores[(size_t)a * n + b];
    if (sc <= 0.f) return 0.f;
    if (a == b) return 1.f / (float)n;                         // d(diag^2/(2n))/ds / s = (s/n)/s
    float c = fmaxf(margin - sc, 0.f);
    if (c <= 0.f) return 0.f;
    if (max_violation && b != argmax_a) return 0.f;
    return -c / ((float)n * sc);
}

__global__ __launch_bounds__(256) void l2c_argmax_kernel(const float* __restrict__ scores, int n, float margin, int* __restrict__ arg) {
    int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    float best = -1.f; int bi = -1;
    for (int b = 0; b < n; ++b) {
        if (b == a) continue;
        float c = fmaxf(margin - scores[(size_t)a * n + b], 0.f);
        if (c > best) { best = c; bi = b; }                    // first maximum, like torch.max
    }
    arg[a] = bi;
}

// which = 0: df1[a][k] = g * sum_b w_ab (f1[a][k] - f2[b][k]);  which = 1: df2[b][k] = -g * sum_a w_ab (f1[a][k] - f2[b][k])
__global__ __launch_bounds__(256) void l2c_bwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                      const float* __restrict__ scores, const int* __restrict__ arg,
                                                      int n, int d, float margin, int max_violation,
                                                      const float* __restrict__ gscale, float* __restrict__ out, int which) {
    const int r = blockIdx.x;
    const float g = gscale ? *gscale : 1.f;
    for (int k = threadIdx.x; k < d; k += 256) {
        float acc = 0.f;
        for (int o = 0; o < n; ++o) {
            const int a = which == 0 ? r : o, b = which == 0 ? o : r;
            float w = l2c_w(scores, n, a, b, margin, max_violation, max_violation ? arg[a] : -1);
            if (w != 0.f) acc += w * (f1[(size_t)a * d + k] - f2[(size_t)b * d + k]);
        }
        out[(size_t)r * d + k] = (which == 0 ? g : -g) * acc;
    }
}

}  // namespace

extern "C" int viai_l2c_fwd(const float* f1, const float* f2, int n, int d, float margin, int max_violation,
                            float* scores, float* loss, void* stream) {
    if (n <= 0 || d <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    VIAI_LAUNCH(l2c_scores_kernel, dim3(n, (n + 3) / 4), dim3(256), 0, st, f1, f2, scores, n, d);
    VIAI_LAUNCH(l2c_loss_kernel, dim3(1), dim3(256), 0, st, scores, n, margin, max_violation, loss);
    return viai_launch_status();
}

extern "C" int viai_l2c_bwd(const float* f1, const float* f2, const float* scores, int n, int d, float margin,
                            int max_violation, const float* gscale, int* argmax_ws, float* df1, float* df2, void* stream) {
    if (n <= 0 || d <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (max_violation) VIAI_LAUNCH(l2c_argmax_kernel, dim3((n + 255) / 256), dim3(256), 0, st, scores, n, margin, argmax_ws);
    if (df1) VIAI_LAUNCH(l2c_bwd_kernel, dim3(n), dim3(256), 0, st, f1, f2, scores, argmax_ws, n, d, margin, max_violation, gscale, df1, 0);
    if (df2) VIAI_LAUNCH(l2c_bwd_kernel, dim3(n), dim3(256), 0, st, f1, f2, scores, argmax_ws, n, d, margin, max_violation, gscale, df2, 1);
    return viai_launch_status();
}
