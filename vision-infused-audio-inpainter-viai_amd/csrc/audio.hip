// Fused STFT -> |.| -> mel -> dB -> [0,1] (-> inpainting mask) front end, gfx950.
// Follows melspectrogram() of the reference's utils/audio.py:70-75:
//   D = lws.stft(y)            frames of `fft` samples, hop `hop`, zero padding of
//                              (fft - hop) samples on both sides (:90-108), analysis window
//   S = 20*log10(max(10^(min_db/20), mel_basis @ |D|)) - ref_level_db   (:116-132)
//   out = clip((S - min_db) / -min_db, 0, 1)                             (:139-140)
// The window and the mel basis are INPUTS (host builds them; lws / librosa are
// not available to pin them — "parity unpinned").
//
// One workgroup per (clip, frame): 1024-point Stockham radix-2 FFT in LDS
// (fp32, sincospif twiddles), magnitudes kept in LDS, then each lane owns mel
// bins and streams the transposed basis ([bin][mel], coalesced, L2-resident).
// HBM traffic is the waveform in and the mel out: bandwidth-bound by design.
#include "viai_common.h"
#include "viai_internal.h"

namespace {

template <int NFFT>
__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                       const float* __restrict__ basis_t, const float* __restrict__ mask,
                                                       float* __restrict__ mel, int n_samples, int hop, int n_mels,
                                                       int frames, float min_db, float ref_db) {
    __shared__ float2 bufA[NFFT];
    __shared__ float2 bufB[NFFT];
    const int frame = blockIdx.x, clip = blockIdx.y, tid = threadIdx.x;
    const float* y = wav + (size_t)clip * n_samples;
    const int start = frame * hop - (NFFT - hop);       // lws pads (fft - hop) zeros on the left
    for (int j = tid; j < NFFT; j += 256) {
        int i = start + j;
        float v = (i >= 0 && i < n_samples) ? y[i] * window[j] : 0.f;
        bufA[j] = make_float2(v, 0.f);
    }
    __syncthreads();
    float2* in = bufA;
    float2* out = bufB;
    for (int ns = 1; ns < NFFT; ns <<= 1) {
        for (int j = tid; j < NFFT / 2; j += 256) {
            int k = j & (ns - 1);
            float s, c;
            sincospif(-(float)k / (float)ns, &s, &c);
            float2 a = in[j], b = in[j + NFFT / 2];
            float2 bw = make_float2(b.x * c - b.y * s, b.x * s + b.y * c);
            int idx = ((j - k) << 1) + k;
            out[idx] = make_float2(a.x + bw.x, a.y + bw.y);
            out[idx + ns] = make_float2(a.x - bw.x, a.y - bw.y);
        }
        __syncthreads();
        float2* t = in; in = out; out = t;
    }
    // magnitudes of the one-sided spectrum -> reuse `out` storage as float[NFFT/2+1]
    float* mag = reinterpret_cast<float*>(out);
    for (int k = tid; k <= NFFT / 2; k += 256) {
        float2 v = in[k];
        mag[k] = sqrtf(v.x * v.x + v.y * v.y);
    }
    __syncthreads();
    const float min_level = exp10f(min_db / 20.f);
    const float mk = mask ? mask[(size_t)clip * frames + frame] : 1.f;
    for (int m = tid; m < n_mels; m += 256) {
        float acc = 0.f;
        for (int k = 0; k <= NFFT / 2; ++k) acc += basis_t[(size_t)k * n_mels + m] * mag[k];
        float S = 20.f * log10f(fmaxf(min_level, acc)) - ref_db;
        float nrm = (S - min_db) / (-min_db);
        nrm = fminf(fmaxf(nrm, 0.f), 1.f);
        mel[((size_t)clip * n_mels + m) * frames + frame] = nrm * mk;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Frame-batched kernel (fft = 1024): a block owns EIGHT consecutive frames of one clip.
//  * two real frames ride one complex FFT (z = a + i b;  A[k] = (Z[k] + conj Z[N-k]) / 2,  B[k] = (Z[k] - conj Z[N-k]) / 2i), so the
//    eight frames are four 1024-point FFTs, computed side by side: radix-4 Stockham (five passes, one butterfly of each FFT per thread
//    and pass), twiddles from a 768-entry table built once per block -- the same three twiddles serve all four FFTs;
//  * the mel basis is banded (a triangular filter touches a handful of bins): band m reads bins [lo[m], lo[m] + cnt[m]) only
//    (lo / cnt are computed on the host from the basis handed in; a dense basis still works, it is just slower);
//  * a thread owns mel band m for the eight frames and writes them as two 16-byte stores per band (32-byte runs of the [clip][mel][frame]
//    output instead of one float per lane at stride `frames`).
// LDS: 2 x 4 FFT buffers (64 KB) + twiddles (6 KB) -> two blocks per CU; grid = ceil(frames / 8) x clips.
constexpr int SB_F = 8;                       // frames per block
constexpr int SB_N = 1024;

__global__ __launch_bounds__(256) void stft_mel_banded_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                              const float* __restrict__ basis_t, const int* __restrict__ band_lo,
                                                              const int* __restrict__ band_cnt, const float* __restrict__ mask,
                                                              float* __restrict__ mel, int n_samples, int hop, int n_mels,
                                                              int frames, float min_db, float ref_db) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a[];
    float2* bufA = reinterpret_cast<float2*>(smem_a);                         // [4][1024]
    float2* bufB = bufA + 4 * SB_N;                                           // [4][1024]
    float2* tw = bufB + 4 * SB_N;                                             // [768]  exp(-2 pi i n / 1024)
    const int tid = threadIdx.x, clip = blockIdx.y, f0 = blockIdx.x * SB_F;
    const float* y = wav + (size_t)clip * n_samples;
    for (int n = tid; n < 768; n += 256) {
        float s, c;
        sincospif(-(float)n * (1.0f / 512.0f), &s, &c);
        tw[n] = make_float2(c, s);
    }
    // windowed frames: FFT q holds frame f0 + 2 q (real part) and f0 + 2 q + 1 (imaginary part); frames past the end are zero
    for (int j = tid; j < SB_N; j += 256) {
        const float w = window[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ia = (f0 + 2 * q) * hop - (SB_N - hop) + j, ib = ia + hop;        // lws pads (fft - hop) zeros on the left
            const float a = (ia >= 0 && ia < n_samples && f0 + 2 * q < frames) ? y[ia] * w : 0.f;
            const float b = (ib >= 0 && ib < n_samples && f0 + 2 * q + 1 < frames) ? y[ib] * w : 0.f;
            bufA[q * SB_N + j] = make_float2(a, b);
        }
    }
    __syncthreads();
    float2* in = bufA;
    float2* out = bufB;
    auto cmul = [](float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); };
#pragma unroll 1
    for (int p = 1; p < SB_N; p <<= 2) {                                       // p = 1, 4, 16, 64, 256
        const int k = tid & (p - 1), j = ((tid - k) << 2) + k;
        const int n1 = k * (256 / p);
        const float2 w1 = tw[n1], w2 = tw[2 * n1], w3 = tw[3 * n1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2* x = in + q * SB_N;
            float2* o = out + q * SB_N;
            const float2 u0 = x[tid], u1 = cmul(x[tid + 256], w1), u2 = cmul(x[tid + 512], w2), u3 = cmul(x[tid + 768], w3);
            const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y), v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
            const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y), d = make_float2(u1.x - u3.x, u1.y - u3.y);
            const float2 v3 = make_float2(d.y, -d.x);                          // (u1 - u3) * (-i)
            o[j] = make_float2(v0.x + v2.x, v0.y + v2.y);
            o[j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
            o[j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
            o[j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
        }
        __syncthreads();
        float2* t = in; in = out; out = t;
    }
    // magnitudes of the two real frames of every FFT: mag[frame][k], k = 0 .. 512, into the free buffer (stride 516 floats)
    constexpr int MS = 516;
    float* mag = reinterpret_cast<float*>(out);
    for (int k = tid; k <= SB_N / 2; k += 256) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 z = in[q * SB_N + k], zc = in[q * SB_N + ((SB_N - k) & (SB_N - 1))];
            const float ar = 0.5f * (z.x + zc.x), ai = 0.5f * (z.y - zc.y);             // A[k] = (Z[k] + conj Z[N-k]) / 2
            const float br = 0.5f * (z.y + zc.y), bi = 0.5f * (zc.x - z.x);             // B[k] = (Z[k] - conj Z[N-k]) / 2i
            mag[(2 * q) * MS + k] = sqrtf(ar * ar + ai * ai);
            mag[(2 * q + 1) * MS + k] = sqrtf(br * br + bi * bi);
        }
    }
    __syncthreads();
    const float min_level = exp10f(min_db / 20.f);
    const bool vec_ok = (frames % 4 == 0) && (f0 + SB_F <= frames);             // 16-byte aligned rows, all eight frames exist
    for (int m = tid; m < n_mels; m += 256) {
        const int lo = band_lo[m], cnt = band_cnt[m];
        float acc[SB_F];
#pragma unroll
        for (int f = 0; f < SB_F; ++f) acc[f] = 0.f;
        for (int i = 0; i < cnt; ++i) {
            const float w = basis_t[(size_t)(lo + i) * n_mels + m];
#pragma unroll
            for (int f = 0; f < SB_F; ++f) acc[f] += w * mag[f * MS + lo + i];
        }
        float r[SB_F];
#pragma unroll
        for (int f = 0; f < SB_F; ++f) {
            const float S = 20.f * log10f(fmaxf(min_level, acc[f])) - ref_db;
            float nrm = (S - min_db) / (-min_db);
            nrm = fminf(fmaxf(nrm, 0.f), 1.f);
            const float mk = (mask && f0 + f < frames) ? mask[(size_t)clip * frames + f0 + f] : 1.f;
            r[f] = nrm * mk;
        }
        float* dst = mel + ((size_t)clip * n_mels + m) * frames + f0;
        if (vec_ok) {
            *reinterpret_cast<f32x4*>(dst) = f32x4{r[0], r[1], r[2], r[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{r[4], r[5], r[6], r[7]};
        } else {
#pragma unroll
            for (int f = 0; f < SB_F; ++f)
                if (f0 + f < frames) dst[f] = r[f];
        }
    }
}

}  // namespace

// as viai_stft_mel for fft = 1024 with the support of every mel band given: band m is nonzero on bins [band_lo[m], band_lo[m] + band_cnt[m])
extern "C" int viai_stft_mel_banded(const float* wav, const float* window, const float* basis_t, const int* band_lo, const int* band_cnt,
                                    const float* mask, float* mel, int B, int n_samples, int fft, int hop, int n_mels, int frames,
                                    float min_level_db, float ref_level_db, void* stream) {
    if (B <= 0 || frames <= 0 || hop <= 0 || hop > fft || n_mels <= 0 || fft != SB_N || band_lo == nullptr || band_cnt == nullptr) return (int)hipErrorInvalidValue;
    constexpr int lds = (8 * SB_N + 768) * (int)sizeof(float2);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mel_banded_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
    dim3 grid((frames + SB_F - 1) / SB_F, B);
    VIAI_LAUNCH(stft_mel_banded_kernel, grid, dim3(256), lds, (hipStream_t)stream, wav, window, basis_t, band_lo, band_cnt, mask, mel, n_samples, hop, n_mels,
                frames, min_level_db, ref_level_db);
    return viai_launch_status();
}

// basis_t: [fft/2+1][n_mels] (transposed mel basis)
extern "C" int viai_stft_mel(const float* wav, const float* window, const float* basis_t, const float* mask,
                             float* mel, int B, int n_samples, int fft, int hop, int n_mels, int frames,
                             float min_level_db, float ref_level_db, void* stream) {
    if (B <= 0 || frames <= 0 || hop <= 0 || hop > fft || n_mels <= 0) return (int)hipErrorInvalidValue;
    dim3 grid(frames, B);
    hipStream_t st = (hipStream_t)stream;
    if (fft == 1024) VIAI_LAUNCH(stft_mel_kernel<1024>, grid, dim3(256), 0, st, wav, window, basis_t, mask, mel, n_samples, hop, n_mels, frames, min_level_db, ref_level_db);
    else if (fft == 512) VIAI_LAUNCH(stft_mel_kernel<512>, grid, dim3(256), 0, st, wav, window, basis_t, mask, mel, n_samples, hop, n_mels, frames, min_level_db, ref_level_db);
    else if (fft == 2048) VIAI_LAUNCH(stft_mel_kernel<2048>, grid, dim3(256), 0, st, wav, window, basis_t, mask, mel, n_samples, hop, n_mels, frames, min_level_db, ref_level_db);
    else return (int)hipErrorInvalidValue;
    return viai_launch_status();
}
