// The steps immediately before and after the train-step path, all HBM-bound streaming work:
//   batch assembly   Data_loaders/audio_loader.py:185-245 (frame normalise / flip / crop), :471-475,:508,:523 (mel + audio
//                    window of each clip, (T, D) -> (D, T) transpose)
//   EMA              loss_functions.py:65-76 ExponentialMovingAverage.update
//   inverse mel      utils/audio.py:135-144 _denormalize + _db_to_amp (inpainted mel -> amplitudes for the vocoder)
//   retrieval        utils/util.py:99-121 L2retrieval: rank of the matching clip under pairwise L2 distance
#include "viai_common.h"
#include "viai_internal.h"

namespace {

inline int ew_blocks_p(long n) {
    long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

// frames: uint8 [n][S][S][C] (RGB or (flow_x, flow_y)), already resized to S x S.
// out:    float [n][size][size][4] NHWC4 (channel C.. zero) -- the layout ResNet conv1 consumes -- with
//         out[y][x] = (px(crop_x + y, flip ? S-1-(crop_y + x) : crop_y + x) - 127) / 128
// (the reference flips the S x S image left-right, then crops rows [crop_x, +size) and columns [crop_y, +size))
__global__ __launch_bounds__(256) void frames_prep_kernel(const unsigned char* __restrict__ frames, f32x4* __restrict__ out,
                                                          long n, int S, int C, int size, int crop_x, int crop_y, int flip) {
    const long total = n * size * size;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int x = (int)(i % size); long r = i / size; int y = (int)(r % size); long f = r / size;
        int sy = crop_x + y, sx = crop_y + x;
        if (flip) sx = S - 1 - sx;
        const unsigned char* p = frames + ((f * S + sy) * S + sx) * C;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) v[c] = ((float)p[c] - 127.f) / 128.f;
        out[i] = v;
    }
}

// c: [T_total][D] mel frames of one utterance, x: [samples]; clip b takes mel frames [m0, m0 + L), m0 = 3 + 4 * start[b],
// and samples [m0 * hop, (m0 + L) * hop).  Outputs: c_out [B][D][L] (channel first), x_out [B][L * hop].
__global__ __launch_bounds__(256) void slice_clips_kernel(const float* __restrict__ c, const float* __restrict__ x,
                                                          const int* __restrict__ start, float* __restrict__ c_out,
                                                          float* __restrict__ x_out, int B, int D, int L, int hop,
                                                          long T_total, long samples) {
    const long nc = (long)B * D * L, nx = (long)B * L * hop;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nc + nx; i += (long)gridDim.x * 256L) {
        if (i < nc) {
            int t = (int)(i % L); long r = i / L; int d = (int)(r % D); int b = (int)(r / D);
            long m = 3 + 4L * start[b] + t;
            c_out[i] = (m < T_total) ? c[m * D + d] : 0.f;         // _pad_2d: zero padding past the end
        } else {
            long k = i - nc;
            long s = k % ((long)L * hop); int b = (int)(k / ((long)L * hop));
            long src = (3 + 4L * start[b]) * hop + s;
            x_out[k] = (src < samples) ? x[src] : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ shadow, const float* __restrict__ x, long n, float one_minus_decay) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        float s = shadow[i];
        float delta = s - x[i];                       // loss_functions.py:74-75, same operation order
        shadow[i] = s - one_minus_decay * delta;
    }
}

__global__ __launch_bounds__(256) void mel_denorm_amp_kernel(const float* __restrict__ S, float* __restrict__ out, long n, float min_level_db) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        float v = fminf(fmaxf(S[i], 0.f), 1.f) * -min_level_db + min_level_db;     // _denormalize
        out[i] = exp10f(v * 0.05f);                                                   // _db_to_amp
    }
}

// one block per caption i: d2(i, j) = |captions[i] - clips[j]|^2 for all j; rank = #{j : d2(i,j) < d2(i,i)} with
// index order breaking ties (position of clip i in the ascending sort of row i), top1 = argmin_j.
__global__ __launch_bounds__(256) void l2_ranks_kernel(const float* __restrict__ clips, const float* __restrict__ captions,
                                                       int n_clips, int dim, int* __restrict__ ranks, int* __restrict__ top1,
                                                       float* __restrict__ dist) {
    extern __shared__ float cap[];                 // [dim]
    __shared__ float red_v[256];
    __shared__ int red_i[256], red_c[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < dim; k += 256) cap[k] = captions[(size_t)i * dim + k];
    __syncthreads();
    auto d2 = [&](int j) {
        const float* c = clips + (size_t)j * dim;
        float s = 0.f;
        for (int k = 0; k < dim; ++k) { float d = cap[k] - c[k]; s += d * d; }
        return s;
    };
    const float dii = d2(i);                        // every thread evaluates the same value in the same order
    int cnt = 0, best_j = 0x7fffffff;
    float best = 3.4e38f;
    for (int j = tid; j < n_clips; j += 256) {
        float v = d2(j);
        if (dist) dist[(size_t)i * n_clips + j] = sqrtf(v);
        if (v < dii || (v == dii && j < i)) ++cnt;
        if (v < best || (v == best && j < best_j)) { best = v; best_j = j; }
    }
    red_v[tid] = best; red_i[tid] = best_j; red_c[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red_c[tid] += red_c[tid + s];
            if (red_v[tid + s] < red_v[tid] || (red_v[tid + s] == red_v[tid] && red_i[tid + s] < red_i[tid])) {
                red_v[tid] = red_v[tid + s]; red_i[tid] = red_i[tid + s];
            }
        }
        __syncthreads();
    }
    if (tid == 0) { ranks[i] = red_c[0]; top1[i] = red_i[0]; }
}

}  // namespace

extern "C" int viai_frames_prep(const unsigned char* frames, float* out, long n, int S, int C, int size,
                                int crop_x, int crop_y, int flip, void* stream) {
    if (C < 1 || C > 4 || size < 1 || crop_x < 0 || crop_y < 0 || crop_x + size > S || crop_y + size > S) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(frames_prep_kernel, dim3(ew_blocks_p(n * size * size)), dim3(256), 0, (hipStream_t)stream, frames,
                reinterpret_cast<f32x4*>(out), n, S, C, size, crop_x, crop_y, flip);
    return viai_launch_status();
}

extern "C" int viai_slice_clips(const float* c, const float* x, const int* start, float* c_out, float* x_out,
                                int B, int D, int L, int hop, long T_total, long samples, void* stream) {
    if (B < 1 || D < 1 || L < 1 || hop < 1) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(slice_clips_kernel, dim3(ew_blocks_p((long)B * L * (D + hop))), dim3(256), 0, (hipStream_t)stream, c, x, start,
                c_out, x_out, B, D, L, hop, T_total, samples);
    return viai_launch_status();
}

extern "C" int viai_ema_update(float* shadow, const float* x, long n, double decay, void* stream) {
    // (1 - decay) in double, then rounded once: 1.0f - 0.9999f would carry a 1.7e-4 relative cancellation error
    VIAI_LAUNCH(ema_kernel, dim3(ew_blocks_p(n)), dim3(256), 0, (hipStream_t)stream, shadow, x, n, (float)(1.0 - decay));
    return viai_launch_status();
}

extern "C" int viai_mel_denorm_amp(const float* S, float* out, long n, float min_level_db, void* stream) {
    VIAI_LAUNCH(mel_denorm_amp_kernel, dim3(ew_blocks_p(n)), dim3(256), 0, (hipStream_t)stream, S, out, n, min_level_db);
    return viai_launch_status();
}

extern "C" int viai_l2_ranks(const float* clips, const float* captions, int n_clips, int n_captions, int dim,
                             int* ranks, int* top1, float* dist, void* stream) {
    if (n_captions < 1 || n_captions > n_clips || dim < 1 || dim > 8192) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(l2_ranks_kernel, dim3(n_captions), dim3(256), dim * sizeof(float), (hipStream_t)stream, clips, captions,
                n_clips, dim, ranks, top1, dist);
    return viai_launch_status();
}
