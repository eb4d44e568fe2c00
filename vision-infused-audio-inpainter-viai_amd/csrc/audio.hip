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

}  // namespace

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
