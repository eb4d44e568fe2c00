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
#include <cstdlib>

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


// ---------------------------------------------------------------------------------------------------------------------------
// Wave-per-FFT kernel (round 4; fft = 1024).  The frame-batched kernel above moves every FFT through LDS five times (radix-4 passes, 41 KB of LDS
// traffic per frame) and walks the mel bands with one dependent global load per bin: 1024 clips in 1.25 ms = 0.43 TB/s of its 0.54 GB.  Here
//  * a WAVE owns one 1024-point complex FFT = two real frames (z = a + i b), 16 points per lane, as 1024 = 16 x 16 x 4:
//      radix-16 in registers over n1 (x[n2 + 64 n1], lane = n2: the loads are coalesced and need no staging) -> twiddle W1024^(k1 n2) ->
//      ONE transpose through LDS ([k1][n2], rows padded to 136 dwords: conflict-free both ways) ->
//      radix-16 in registers over b (lane = (k1, a), element a + 4 b) -> twiddle W64^(a kb) ->
//      radix-4 over a ACROSS the four lanes of a quad with DPP quad_perm moves (no LDS);
//    output X[k1 + 16 (kb + 16 ka)] in lane (k1, q), register kb, ka = bit-reversed q;
//  * the two real spectra need X[k] and X[N - k]: the wave parks X in LDS once and the lanes holding k < 512 fetch their partners;
//    magnitudes go to a [bin][8 frames] table shared by the block's four waves (8 frames per block, as before: 32-byte output runs);
//  * the mel weights (banded: ~1000 non-zeros for Slaney triangles) are copied to LDS once per PERSISTENT block, a thread owns a band for the
//    eight frames and reads each bin's eight magnitudes with two 16-byte LDS loads;
//  * twiddles and the window live in registers for the block's lifetime (sincospif once per thread, not per frame group).
// LDS per frame: one transpose (8 KB out + 8 KB in per two frames) + the partner exchange + magnitudes + mel reads ~ 22 KB (was 41).
constexpr int SW_EROW = 68;                    // float2 per row of the transpose buffer (64 + 4 pad = 136 dwords)
constexpr int SW_WREG = 16 * SW_EROW;          // float2 per wave region (8704 bytes; the partner table, 1024 + 16 float2, fits in it)
constexpr int SW_MROW = 8;                     // floats per bin of the magnitude table (8 frames)

typedef float c32 __attribute__((ext_vector_type(2)));            // (re, im) as a register PAIR: the arithmetic below compiles to v_pk_* without shuffling moves
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return a + b; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return a - b; }
__device__ __forceinline__ c32 cmulc(c32 a, c32 w) { return a.xx * w + a.yy * c32{-w.y, w.x}; }
// forward 4-point DFT (W4 = -i), in place
__device__ __forceinline__ void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
    const c32 s0 = a0 + a2, d0 = a0 - a2, s1 = a1 + a3, d1 = a1 - a3;
    const c32 r = {d1.y, -d1.x};                // -i d1
    a0 = s0 + s1; a2 = s0 - s1;
    a1 = d0 + r;
    a3 = d0 - r;
}
// forward 16-point DFT of v[0..15], in place: X[m + 4 q] = sum_j W4^(j q) W16^(j m) sum_p v[j + 4 p] W4^(p m)
__device__ __forceinline__ void dft16(c32 (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dft4(v[j], v[j + 4], v[j + 8], v[j + 12]);           // v[j + 4 m] = T[j][m]
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    // W16^(j m), j, m = 1 .. 3
    v[1 + 4] = cmulc(v[1 + 4], c32{C1, -S1}); v[1 + 8] = cmulc(v[1 + 8], c32{H, -H}); v[1 + 12] = cmulc(v[1 + 12], c32{S1, -C1});
    v[2 + 4] = cmulc(v[2 + 4], c32{H, -H});   v[2 + 8] = c32{v[2 + 8].y, -v[2 + 8].x}; v[2 + 12] = cmulc(v[2 + 12], c32{-H, -H});
    v[3 + 4] = cmulc(v[3 + 4], c32{S1, -C1}); v[3 + 8] = cmulc(v[3 + 8], c32{-H, -H}); v[3 + 12] = cmulc(v[3 + 12], c32{-C1, S1});
#pragma unroll
    for (int m = 0; m < 4; ++m) dft4(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]);   // over j: v[4 m + q] = X[m + 4 q]
    // reorder to natural order k = m + 4 q
    c32 t[16];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) t[m + 4 * q] = v[4 * m + q];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = t[k];
}
// orders a wave's LDS stores before its own later LDS loads of other lanes' data (the LDS queue of a wave is in order: only the compiler must not move them)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int wave_max_i(int v) {           // wave-uniform maximum, in an SGPR
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false)); }

// Block = 8 waves = 16 frames per iteration; two iterations fill a [band][32 frames] output tile in LDS that is then written as 128-byte
// runs: with frame the fastest output index, a thread storing its band's 8 frames (32 bytes at a stride of `frames` floats, unaligned for odd
// frame counts) made every store instruction touch 64 different lines -- THAT, not the FFTs, held both earlier kernels at 0.4 TB/s.
constexpr int SW_OMEL = 256;
// NW waves per block (2 NW frames per iteration), TF frames per output tile, MP floats per bin of the magnitude table (a multiple of 4; padding turns 8-way store
// conflicts into 2-way), WCAP mel weights held in LDS (a denser basis is read from global memory), MMAX the largest n_mels.
//   <8, 32, 20, 2048, 1024>: one 8-wave block per CU (154 KB of LDS);  <4, 16, 8, 1024, 256>: TWO independent 4-wave blocks per CU (73 KB each) -- a block's waves
//   meet at two barriers per iteration, so the two waves of a SIMD walk the phases of an iteration in lockstep; two blocks drift apart.  It did NOT help (551 vs
//   523 us on 1024 clips): the bound is the per-wave VALU issue rate (`tools/probes/valu_rate.hip`: one wave issues a v_fma_f32 every ~7 cycles, a v_pk_fma_f32
//   every ~10; the SIMD takes one per 2 -- it needs 4+ waves to fill, and 240 registers leave room for two: VALU 48 % busy, `profiles/r04_d_pmc_stft.json`).
template <int NW, int TF, int MP, int WCAP, int MMAX>
struct SwCfg {
    static constexpr int FPI = 2 * NW, OT = TF + 1, NT = 64 * NW, FH = FPI / 8;
    static constexpr int lds = NW * SW_WREG * 8 + 516 * MP * 4 + WCAP * 4 + (MMAX + 8) * 4 + 64 * 8 + SW_OMEL * OT * 4;
};
template <int SW_NW, int SW_TF, int SW_MP, int SW_WCAP, int SW_MMAX>
__global__ __launch_bounds__(64 * SW_NW) __attribute__((amdgpu_waves_per_eu(2))) void stft_mel_wave_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                                                            const float* __restrict__ basis_t, const int* __restrict__ band_lo,
                                                            const int* __restrict__ band_cnt, const float* __restrict__ mask,
                                                            float* __restrict__ mel, int n_clips, int n_samples, int hop, int n_mels,
                                                            int frames, float min_db, float ref_db) {
    constexpr int SW_FPI = 2 * SW_NW, SW_OT = SW_TF + 1, SW_FH = SW_FPI / 8;      // SW_FH: eight-frame halves of an iteration (a mel item = (band, half))
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    constexpr int NT = 64 * SW_NW;
    c32* ebuf = reinterpret_cast<c32*>(smem_w);                                  // [8 waves][SW_WREG]
    float* mag = reinterpret_cast<float*>(ebuf + SW_NW * SW_WREG);                 // [513][16]
    float* wts = mag + 516 * SW_MP;                                                // [SW_WCAP] band weights, band after band
    int* boff = reinterpret_cast<int*>(wts + SW_WCAP);                             // [n_mels + 1] (n_mels <= 1024), boff[n_mels] = total
    c32* t64 = reinterpret_cast<c32*>(boff + SW_MMAX + 8);                                // [64] W64^n
    float* otile = reinterpret_cast<float*>(t64 + 64);                             // [min(n_mels, 256)][33]: 32 frames of every band (n_mels <= 256)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: the buffer descriptors below must be scalar, or every load becomes a waterfall loop)
    const bool tiled = n_mels <= SW_OMEL;
    // ---- once per block: band weights into LDS (if they fit)
    if (tid == 0) {
        int o = 0;
        for (int m = 0; m < n_mels; ++m) { boff[m] = o; o += band_cnt[m]; }
        boff[n_mels] = o;
    }
    __syncthreads();
    const bool wl = boff[n_mels] <= SW_WCAP;
    if (wl)
        for (int m = tid; m < n_mels; m += NT) {
            const int lo = band_lo[m], cnt = band_cnt[m], o = boff[m];
            for (int i = 0; i < cnt; ++i) wts[o + i] = basis_t[(size_t)(lo + i) * n_mels + m];
        }
    // ---- once per thread: window samples and twiddles of its lane
    const int n2 = lane, k1r = lane >> 2, qa = lane & 3;
    float win[16];
    c32 tw1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        win[i] = window[n2 + 64 * i];
        float sn, cs;
        sincospif(-(float)(i * n2) * (1.0f / 512.0f), &sn, &cs); tw1[i] = c32{cs, sn};       // W1024^(k1 n2), k1 = i
    }
    if (tid < 64) { float sn, cs; sincospif(-(float)tid * (1.0f / 32.0f), &sn, &cs); t64[tid] = c32{cs, sn}; }      // W64^(a kb) is looked up (a kb < 48)
    const int ka = ((qa & 1) << 1) | (qa >> 1);                                            // output index of the quad's radix-4 held by this lane
    const float sg1 = (qa & 2) ? -1.f : 1.f, sg2 = (qa & 1) ? -1.f : 1.f;
    const float rc = qa == 3 ? 0.f : 1.f, rd = qa == 3 ? 1.f : 0.f;                        // lane 3 multiplies by -i between the two steps
    const float min_level = exp10f(min_db / 20.f), inv_mdb = -1.f / min_db;
    c32* E = ebuf + w * SW_WREG;
    // the first (for n_mels <= 256: the only) mel item of this thread: band tid >> 1, frames 8 (tid & 1) .. + 8 of the iteration
    const bool own_0 = tid < SW_FH * n_mels;
    const int m_0 = own_0 ? tid / SW_FH : 0, hf_0 = tid % SW_FH;
    const int lo_0 = own_0 ? band_lo[m_0] : 0, cnt_0 = own_0 ? band_cnt[m_0] : 0, o_0 = boff[m_0];
    const int cmax_0 = wave_max_i(cnt_0);
    const int gpc = (frames + SW_TF - 1) / SW_TF, total = n_clips * gpc;                   // super-groups of 32 frames
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int g_end = min(total, (int)(blockIdx.x + 1) * per);
    __syncthreads();
    // the samples of an iteration are requested one iteration ahead (after the previous one's last use of v[]): a block's eight waves meet at two barriers per
    // iteration, so nothing else covers the HBM round trip.  Buffer loads: a sample index outside [0, n_samples) -- the lws zero padding on either side, a frame past
    // the end -- is out of the descriptor's range and reads as zero (guarded plain loads compile to one exec-masked branch + wait PER load: 32 dependent round trips
    // per iteration, which -- not the FFT -- was what both earlier kernels spent their 1.3 ms on); a negative byte offset is a huge unsigned one, so the left
    // padding needs no test either; a frame past the end (or an iteration past the block's range) gets an empty descriptor.
    float ra[16], rb[16];
    auto request = [&](int gi, int sub) {
        const int clip = gi / gpc, f0 = (gi - clip * gpc) * SW_TF + sub * SW_FPI;
        const float* y = wav + (size_t)clip * n_samples;
        const int fa = f0 + 2 * w;
        const int ia = fa * hop - (1024 - hop), ib = ia + hop;
        const bool live = gi < g_end, va = live && fa < frames, vb = live && fa + 1 < frames;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, va ? n_samples * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, vb ? n_samples * 4 : 0, 0x00020000);
        const int oa = (ia + n2) * 4, ob = (ib + n2) * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ra[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, oa + 256 * i, 0, 0));
            rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, ob + 256 * i, 0, 0));
        }
    };
    if ((int)(blockIdx.x * per) < g_end) request(blockIdx.x * per, 0);
    for (int gi = blockIdx.x * per; gi < g_end; ++gi) {
        const int clip = gi / gpc, F0 = (gi - clip * gpc) * SW_TF;
        for (int sub = 0; sub < SW_TF / SW_FPI; ++sub) {
            const int f0 = F0 + sub * SW_FPI;
            if (f0 >= frames) break;                                                       // (uniform)
            c32 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = c32{ra[i], rb[i]} * win[i];
            dft16(v);                                                                      // over n1 -> k1
#pragma unroll
            for (int i = 1; i < 16; ++i) v[i] = cmulc(v[i], tw1[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) E[i * SW_EROW + n2] = v[i];
            wave_lds_sync();                                                               // E is this wave's own region: no block barrier
#pragma unroll
            for (int b = 0; b < 16; ++b) v[b] = E[k1r * SW_EROW + qa + 4 * b];
            wave_lds_sync();                                                               // the region is re-used for the partner table below
            dft16(v);                                                                      // over b -> kb
#pragma unroll
            for (int i = 1; i < 16; ++i) v[i] = cmulc(v[i], t64[i * qa]);
            // radix-4 over a across the quad
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                c32 o = c32{dpp_f<0x4E>(v[i].x), dpp_f<0x4E>(v[i].y)};                     // lane ^ 2
                c32 f = sg1 * v[i] + o;
                f = rc * f + rd * c32{f.y, -f.x};
                o = c32{dpp_f<0xB1>(f.x), dpp_f<0xB1>(f.y)};                               // lane ^ 1
                v[i] = sg2 * f + o;
            }
            // v[kb] = X[k], k = k1r + 16 (kb + 16 ka).  Park X: entry k at float2 index k + 4 (k >> 8) (32 bytes of padding per 256 entries)
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                const int k = k1r + 16 * kb + 256 * ka;
                E[k + 4 * (k >> 8)] = v[kb];
            }
            wave_lds_sync();
            if (ka < 2) {
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    const int k = k1r + 16 * kb + 256 * ka, kp = (1024 - k) & 1023;
                    const c32 p = E[kp + 4 * (kp >> 8)], x = v[kb];
                    const float ar = x.x + p.x, ai = x.y - p.y, br = x.x - p.x, bi = x.y + p.y;
                    *reinterpret_cast<float2*>(mag + k * SW_MP + 2 * w) = make_float2(0.5f * __builtin_amdgcn_sqrtf(ar * ar + ai * ai), 0.5f * __builtin_amdgcn_sqrtf(br * br + bi * bi));
                }
            } else if (ka == 2 && k1r == 0) {
                *reinterpret_cast<float2*>(mag + 512 * SW_MP + 2 * w) = make_float2(fabsf(v[0].x), fabsf(v[0].y));  // k = 512: its own partner
            }
            {   // the next iteration's samples (v[] is dead from here on)
                const bool more = sub + 1 < SW_TF / SW_FPI && f0 + SW_FPI < frames;
                request(more ? gi : gi + 1, more ? sub + 1 : 0);
            }
            __syncthreads();
            // ---- mel bands: a thread owns band m for eight of the sixteen frames.  The band's extent is a per-thread constant for the first 512 items (fetched
            // once per block, above); the walk over its bins has a WAVE-uniform trip count (the widest band of the wave: the bands are sorted by width, lanes past
            // their own count multiply by zero) so that it is a scalar loop the compiler can unroll with its LDS reads in flight -- the per-lane `cnt` loop with
            // a select between an LDS and a global weight pointer was one exec-masked round trip (flat load + wait) per bin, 11 of them for the last wave.
            for (int mh = tid, pass = 0; mh - tid < SW_FH * n_mels; mh += NT, ++pass) {        // (uniform bound: every lane walks the same passes)
                const bool own = mh < SW_FH * n_mels;
                int m = m_0, hf = hf_0, lo = lo_0, cnt = cnt_0, o = o_0, cmax = cmax_0;
                if (pass > 0) {
                    m = own ? mh / SW_FH : 0; hf = mh % SW_FH;
                    lo = own ? band_lo[m] : 0; cnt = own ? band_cnt[m] : 0; o = boff[m];
                    cmax = wave_max_i(cnt);
                }
                const int fb = f0 + hf * 8;
                float mk[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) mk[f] = 1.f;
                if (mask != nullptr) {                                                            // (uniform) requested before the walk; frames past the end read as zero and are not stored
                    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)(mask + (size_t)clip * frames), 0, frames * 4, 0x00020000);
#pragma unroll
                    for (int f = 0; f < 8; ++f) mk[f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_m, (fb + f) * 4, 0, 0));
                }
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
                const float* mg = mag + lo * SW_MP + hf * 8;
                if (wl) {
#pragma unroll 4
                    for (int i = 0; i < cmax; ++i) {
                        const bool in = i < cnt;
                        const int ii = in ? i : 0;
                        const float wt = in ? wts[o + ii] : 0.f;
                        const f32x4 m0 = *reinterpret_cast<const f32x4*>(mg + ii * SW_MP), m1 = *reinterpret_cast<const f32x4*>(mg + ii * SW_MP + 4);
                        a0 += wt * m0; a1 += wt * m1;
                    }
                } else {
                    for (int i = 0; i < cmax; ++i) {
                        const bool in = i < cnt;
                        const int ii = in ? i : 0;
                        const float wt = in ? basis_t[(size_t)(lo + ii) * n_mels + m] : 0.f;
                        const f32x4 m0 = *reinterpret_cast<const f32x4*>(mg + ii * SW_MP), m1 = *reinterpret_cast<const f32x4*>(mg + ii * SW_MP + 4);
                        a0 += wt * m0; a1 += wt * m1;
                    }
                }
                if (own) {
#pragma unroll
                    for (int f = 0; f < 8; ++f) {
                        const float acc = f < 4 ? a0[f] : a1[f - 4];
                        const float S = 6.02059991327962f * __builtin_amdgcn_logf(fmaxf(min_level, acc)) - ref_db;      // 20 log10(x) = 6.0206 log2(x); v_log_f32: 1 ulp
                        float nrm = (S - min_db) * inv_mdb;
                        nrm = fminf(fmaxf(nrm, 0.f), 1.f);
                        if (tiled) otile[m * SW_OT + (fb - F0) + f] = nrm * mk[f];
                        else if (fb + f < frames) mel[((size_t)clip * n_mels + m) * frames + fb + f] = nrm * mk[f];
                    }
                }
            }
            __syncthreads();                                                               // the magnitude table is rewritten by the next iteration
        }
        if (tiled) {
            // the tile's rows out as runs of up to 32 consecutive frames: lane = frame, two bands per wave instruction
            const int nf = min(SW_TF, frames - F0);
            for (int idx = tid; idx < n_mels * SW_TF; idx += NT) {
                const int m = idx / SW_TF, f = idx % SW_TF;
                if (f < nf) mel[((size_t)clip * n_mels + m) * frames + F0 + f] = otile[m * SW_OT + f];
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One real frame per wave as a 512-point complex FFT (late round 4).  The kernel above holds 16 complex points per lane (240 registers: two waves per SIMD) and
// sits at the per-WAVE issue limit of the vector pipe (`tools/probes/valu_rate.hip`: one wave issues a VALU instruction every ~5 - 7 cycles, the SIMD accepts one
// per 2).  Here z[n] = x[2n] + i x[2n+1], n < 512 (the even / odd samples of ONE windowed frame), 8 points per lane, <= 128 registers: sixteen waves per CU.
//   512 = 8 x 8 x 8:  radix-8 in registers over n1 (n = n2 + 64 n1, lane = n2: one coalesced 8-byte load per point) -> twiddle W512^(k1 n2) ->
//   transpose through LDS ([k1][n2], rows of 72) -> lane (k1, a): radix-8 over b (n2 = a + 8 b) -> twiddle W64^(a kb) -> transpose ([k1][kb][a], rows of 10) ->
//   lane k1 + 8 kb: radix-8 over a -> Z[k], k = lane + 64 ka in register ka (natural order);
//   real spectrum X[k] = (Z[k] + conj Z[512 - k]) / 2 - i W1024^k (Z[k] - conj Z[512 - k]) / 2: the partner sits in lane 64 - lane, register 7 - ka (lane 0:
//   its own register 8 - ka) and is fetched with ds_bpermute (no LDS storage); |X| into the block's [bin][16 frames] table; band walk and output tile as above
//   with a thread per (band, four frames).
constexpr int R5_WCAP = 1024, R5_MMAX = 256;
constexpr int R5_ROW1 = 72, R5_WREG = 640;                         // c32 per wave region: max(8 x 72, 64 x 10).  R5_ROW2 (template): rows of the second transpose (10 c32 = 80 bytes: 16-byte reads; row index kb 8 + k1, conflict-free both ways)
// conflicts; 9 = conflict-free stores, 8-byte reads: measured SLOWER (491 - 510 vs 463 - 492 us, same box)            // c32 per wave region: max(8 x 72, 64 x 9).  Rows of 9: the (k1, kb) rows of one store instruction fall 16 banks apart (two passes, the minimum for 512 bytes; rows of 10 were 4-way)
// <16, 32, 20>: one 1024-thread block per CU (159 KB of LDS);  <8, 16, 8>: TWO 512-thread blocks per CU (78 KB each; the band offsets share the output tile's
// space: they are only read before the first iteration) -- one block's loads / barriers / write-out overlap the other's arithmetic
template <int NW, int TF, int MP> constexpr int r5_lds() { return NW * R5_WREG * 8 + 516 * MP * 4 + R5_WCAP * 4 + (NW == 16 ? (R5_MMAX + 8) * 4 : 0) + SW_OMEL * (TF + 1) * 4 + 64 * 8; }
// forward 8-point DFT of v[0..7] in place, natural order in and out: X[2m] = DFT4(v[n] + v[n + 4])[m], X[2m + 1] = DFT4((v[n] - v[n + 4]) W8^n)[m]
__device__ __forceinline__ void dft8(c32 (&v)[8]) {
    constexpr float H = 0.70710678118654752f;
    c32 e0 = v[0] + v[4], o0 = v[0] - v[4], e1 = v[1] + v[5], o1 = v[1] - v[5], e2 = v[2] + v[6], o2 = v[2] - v[6], e3 = v[3] + v[7], o3 = v[3] - v[7];
    o1 = c32{(o1.x + o1.y) * H, (o1.y - o1.x) * H};            // * (1 - i) / sqrt 2
    o2 = c32{o2.y, -o2.x};                                      // * -i
    o3 = c32{(o3.y - o3.x) * H, -(o3.x + o3.y) * H};           // * -(1 + i) / sqrt 2
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    v[0] = e0; v[2] = e1; v[4] = e2; v[6] = e3;
    v[1] = o0; v[3] = o1; v[5] = o2; v[7] = o3;
}
__device__ __forceinline__ float lane_from(float x, int src_lane) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane * 4, __builtin_bit_cast(int, x))); }

template <int R5_NW, int R5_TF, int R5_MP, int R5_ROW2>
__global__ __launch_bounds__(64 * R5_NW) __attribute__((amdgpu_waves_per_eu(4))) void stft_mel_r512_kernel(const float* __restrict__ wav, const float* __restrict__ window, const float* __restrict__ basis_t,
                                                                    const int* __restrict__ band_lo, const int* __restrict__ band_cnt, const float* __restrict__ mask,
                                                                    float* __restrict__ mel, int n_clips, int n_samples, int hop, int n_mels, int frames,
                                                                    float min_db, float ref_db) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    constexpr int NT = 64 * R5_NW, R5_OT = R5_TF + 1, QN = R5_NW / 4;                // QN: four-frame quarters of an iteration (a mel item = (band, quarter))
    c32* ebuf = reinterpret_cast<c32*>(smem_r);                                   // [16 waves][R5_WREG]
    float* mag = reinterpret_cast<float*>(ebuf + R5_NW * R5_WREG);                 // [513][16 frames (+4)]
    float* wts = mag + 516 * R5_MP;                                                // band weights, band after band
    int* boff = reinterpret_cast<int*>(wts + R5_WCAP);                             // [n_mels + 1]; the two-block shape overlays it on the output tile
    float* otile = R5_NW == 16 ? reinterpret_cast<float*>(boff + R5_MMAX + 8) : reinterpret_cast<float*>(boff);       // [n_mels][TF + 1]
    c32* t64 = reinterpret_cast<c32*>(otile + SW_OMEL * R5_OT);                    // [a][kb] W64^(a kb)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) {
        int o = 0;
        for (int m = 0; m < n_mels; ++m) { boff[m] = o; o += band_cnt[m]; }
        boff[n_mels] = o;
    }
    if (tid < 64) { float sn, cs; sincospif(-(float)((tid >> 3) * (tid & 7)) * (1.0f / 32.0f), &sn, &cs); t64[tid] = c32{cs, sn}; }
    __syncthreads();
    const bool wl = boff[n_mels] <= R5_WCAP;
    if (wl)
        for (int m = tid; m < n_mels; m += NT) {
            const int lo = band_lo[m], cnt = band_cnt[m], o = boff[m];
            for (int i = 0; i < cnt; ++i) wts[o + i] = basis_t[(size_t)(lo + i) * n_mels + m];
        }
    // ---- once per thread: window pairs, the three twiddle sets of its lane
    const int n2 = lane, k1r = lane >> 3, a = lane & 7;
    c32 win[8], tw1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        win[i] = c32{window[2 * (n2 + 64 * i)], window[2 * (n2 + 64 * i) + 1]};
        float sn, cs;
        sincospif(-(float)(i * n2) * (1.0f / 256.0f), &sn, &cs); tw1[i] = c32{cs, sn};       // W512^(k1 n2), k1 = i
    }
    float sn0, cs0;
    sincospif(-(float)lane * (1.0f / 512.0f), &sn0, &cs0);
    const c32 wk0 = {cs0, sn0};                                                            // W1024^lane; W1024^(lane + 64 ka) = wk0 W16^ka
    const float min_level = exp10f(min_db / 20.f), inv_mdb = -1.f / min_db;
    c32* E = ebuf + w * R5_WREG;
    const int gpc = (frames + R5_TF - 1) / R5_TF, total = n_clips * gpc;
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int g_end = min(total, (int)(blockIdx.x + 1) * per);
    // a thread's mel item: band tid / QN, frames 4 (tid % QN) .. + 4 of the iteration's
    const bool own = tid < QN * n_mels;
    const int m_ = own ? tid / QN : 0, qf = tid % QN;
    const int lo_ = own ? band_lo[m_] : 0, cnt_ = own ? band_cnt[m_] : 0, o_ = boff[m_];
    const int cmax = wave_max_i(cnt_);
    const int partner = (64 - lane) & 63;
    __syncthreads();
    for (int gi = blockIdx.x * per; gi < g_end; ++gi) {
        const int clip = gi / gpc, F0 = (gi - clip * gpc) * R5_TF;
        const float* y = wav + (size_t)clip * n_samples;
        for (int sub = 0; sub < R5_TF / R5_NW; ++sub) {
            const int f0 = F0 + sub * R5_NW;
            if (f0 >= frames) break;                                                       // (uniform)
            const int fa = f0 + w;
            const int ia = fa * hop - (1024 - hop);
            // (range-checked buffer loads: the zero padding on either side and frames past the end read as zero; hop and n_samples are even, so a pair never straddles)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, fa < frames ? n_samples * 4 : 0, 0x00020000);
            const int ob = (ia + 2 * n2) * 4;
            c32 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = __builtin_bit_cast(c32, __builtin_amdgcn_raw_buffer_load_b64(rs, ob + 512 * i, 0, 0)) * win[i];      // (indexing the builtin's result element-wise compiled to a ONE-dword load)
            }
            dft8(v);                                                                       // over n1 -> k1
#pragma unroll
            for (int i = 1; i < 8; ++i) v[i] = cmulc(v[i], tw1[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) E[i * R5_ROW1 + n2] = v[i];
            wave_lds_sync();
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = E[k1r * R5_ROW1 + a + 8 * b];
            wave_lds_sync();
            dft8(v);                                                                       // over b -> kb
#pragma unroll
            for (int i = 1; i < 8; ++i) v[i] = cmulc(v[i], t64[a * 8 + i]);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) E[(kb * 8 + k1r) * R5_ROW2 + a] = v[kb];          // row kb 8 + k1: the reader's rows are then lane-consecutive (see below)
            wave_lds_sync();
            {   // lane = k1 + 8 kb reads its eight a's: 64 contiguous bytes of row kb 8 + k1 = lane.  (Round 6: the rows were k1 8 + kb, i.e. row 8 (lane & 7) + (lane >> 3):
                // with 80-byte rows the start banks of lanes 0, 2, 4, 6 (mod 8) coincided -- a 4-way conflict on every ds_read_b128, 44 % of the kernel's LDS cycles,
                // profiles/r04_d_pmc_stft_r512.json.  Lane-consecutive rows start at banks 20 lane mod 64: sixteen distinct multiples of 4 per read group.)
                const c32* src = E + lane * R5_ROW2;
                if constexpr (R5_ROW2 % 2 == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const f32x4 t = reinterpret_cast<const f32x4*>(src)[j]; v[2 * j] = c32{t[0], t[1]}; v[2 * j + 1] = c32{t[2], t[3]}; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = src[j];
                }
            }
            wave_lds_sync();                                                               // (the region is rewritten by the next iteration's first transpose)
            dft8(v);                                                                       // over a -> ka: v[ka] = Z[lane + 64 ka]
            // ---- the real spectrum of the frame: partner Z[512 - k]
            c32 p[8];
#pragma unroll
            for (int ka = 0; ka < 8; ++ka) {
                const c32 q = {lane_from(v[7 - ka].x, partner), lane_from(v[7 - ka].y, partner)};
                const c32 self = ka == 0 ? v[0] : v[8 - ka];                                // lane 0: 512 - 64 ka = 64 (8 - ka), Z[512] = Z[0]
                p[ka] = lane == 0 ? self : q;
            }
            constexpr float W16C[8] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
            constexpr float W16S[8] = {0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
#pragma unroll
            for (int ka = 0; ka < 8; ++ka) {
                const int k = lane + 64 * ka;
                const c32 z = v[ka], q = p[ka];
                const float er = z.x + q.x, ei = z.y - q.y, dr = z.x - q.x, di = z.y + q.y;        // E = Z + conj P, D = Z - conj P
                const c32 wk = cmulc(wk0, c32{W16C[ka], W16S[ka]});
                const float tr = dr * wk.x - di * wk.y, ti = dr * wk.y + di * wk.x;                // T = W D
                const float xr = er + ti, xi = ei - tr;                                             // 2 X = E - i T
                mag[k * R5_MP + w] = 0.5f * __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
            }
            if (lane == 0) mag[512 * R5_MP + w] = fabsf(v[0].x - v[0].y);                           // X[512] = Re Z[0] - Im Z[0]
            __syncthreads();
            // ---- mel bands: a thread owns (band, four frames); wave-uniform trip count (see the kernel above)
            {
                const int fb = f0 + qf * 4;
                float mk[4] = {1.f, 1.f, 1.f, 1.f};
                if (mask != nullptr) {
                    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)(mask + (size_t)clip * frames), 0, frames * 4, 0x00020000);
#pragma unroll
                    for (int f = 0; f < 4; ++f) mk[f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_m, (fb + f) * 4, 0, 0));
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* mg = mag + lo_ * R5_MP + qf * 4;
                if (wl) {
#pragma unroll 4
                    for (int i = 0; i < cmax; ++i) {
                        const bool in = i < cnt_;
                        const int ii = in ? i : 0;
                        const float wt = in ? wts[o_ + ii] : 0.f;
                        acc += wt * *reinterpret_cast<const f32x4*>(mg + ii * R5_MP);
                    }
                } else {
                    for (int i = 0; i < cmax; ++i) {
                        const bool in = i < cnt_;
                        const int ii = in ? i : 0;
                        const float wt = in ? basis_t[(size_t)(lo_ + ii) * n_mels + m_] : 0.f;
                        acc += wt * *reinterpret_cast<const f32x4*>(mg + ii * R5_MP);
                    }
                }
                if (own) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const float S = 6.02059991327962f * __builtin_amdgcn_logf(fmaxf(min_level, acc[f])) - ref_db;
                        float nrm = (S - min_db) * inv_mdb;
                        nrm = fminf(fmaxf(nrm, 0.f), 1.f);
                        otile[m_ * R5_OT + (fb - F0) + f] = nrm * mk[f];
                    }
                }
            }
            __syncthreads();                                                               // the magnitude table is rewritten by the next iteration
        }
        const int nf = min(R5_TF, frames - F0);
        for (int idx = tid; idx < n_mels * R5_TF; idx += NT) {
            const int m = idx / R5_TF, f = idx % R5_TF;
            if (f < nf) mel[((size_t)clip * n_mels + m) * frames + F0 + f] = otile[m * R5_OT + f];
        }
        __syncthreads();
    }
}

template <int NW, int TF, int MP, int ROW2>
int launch_stft_r512(int per_cu, const float* wav, const float* window, const float* basis_t, const int* band_lo, const int* band_cnt, const float* mask, float* mel,
                     int B, int n_samples, int hop, int n_mels, int frames, float min_db, float ref_db, hipStream_t st) {
    constexpr int lds = r5_lds<NW, TF, MP>();
    auto kern = stft_mel_r512_kernel<NW, TF, MP, ROW2>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    const long groups = (long)B * ((frames + TF - 1) / TF);
    const int grid = (int)(groups < 256 * per_cu ? groups : 256 * per_cu);
    VIAI_LAUNCH(kern, dim3(grid), dim3(64 * NW), lds, st, wav, window, basis_t, band_lo, band_cnt, mask, mel, B, n_samples, hop, n_mels, frames, min_db, ref_db);
    return viai_launch_status();
}

template <int NW, int TF, int MP, int WCAP, int MMAX>
int launch_stft_wave(int per_cu, const float* wav, const float* window, const float* basis_t, const int* band_lo, const int* band_cnt, const float* mask, float* mel,
                     int B, int n_samples, int hop, int n_mels, int frames, float min_db, float ref_db, hipStream_t st) {
    using Cfg = SwCfg<NW, TF, MP, WCAP, MMAX>;
    auto kern = stft_mel_wave_kernel<NW, TF, MP, WCAP, MMAX>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::lds); attr = true; }
    const long groups = (long)B * ((frames + TF - 1) / TF);
    const int grid = (int)(groups < 256 * per_cu ? groups : 256 * per_cu);      // persistent blocks
    VIAI_LAUNCH(kern, dim3(grid), dim3(Cfg::NT), Cfg::lds, st, wav, window, basis_t, band_lo, band_cnt, mask, mel, B, n_samples, hop, n_mels, frames, min_db, ref_db);
    return viai_launch_status();
}

}  // namespace

// as viai_stft_mel for fft = 1024 with the support of every mel band given: band m is nonzero on bins [band_lo[m], band_lo[m] + band_cnt[m])
extern "C" int viai_stft_mel_banded(const float* wav, const float* window, const float* basis_t, const int* band_lo, const int* band_cnt,
                                    const float* mask, float* mel, int B, int n_samples, int fft, int hop, int n_mels, int frames,
                                    float min_level_db, float ref_level_db, void* stream) {
    if (B <= 0 || frames <= 0 || hop <= 0 || hop > fft || n_mels <= 0 || fft != SB_N || band_lo == nullptr || band_cnt == nullptr) return (int)hipErrorInvalidValue;
    // kernel choice by shape (the A/B variants of round 4 -- two 4-wave blocks per CU: 551 us on 1024 clips, the r512 kernel as two 8-wave blocks: 479 - 491 -- were
    // measured, lost and are gone): a frame per wave as a 512-point FFT (455 - 480 us) where the shape allows, else one 8-wave block per CU (523), else the
    // frame-batched kernel
    if (n_mels <= R5_MMAX && hop % 2 == 0 && n_samples % 2 == 0 && (reinterpret_cast<uintptr_t>(wav) & 7) == 0)
        return launch_stft_r512<16, 32, 20, 10>(1, wav, window, basis_t, band_lo, band_cnt, mask, mel, B, n_samples, hop, n_mels, frames, min_level_db, ref_level_db, (hipStream_t)stream);
    if (n_mels <= 1024)
        return launch_stft_wave<8, 32, 20, 2048, 1024>(1, wav, window, basis_t, band_lo, band_cnt, mask, mel, B, n_samples, hop, n_mels, frames, min_level_db, ref_level_db, (hipStream_t)stream);
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
