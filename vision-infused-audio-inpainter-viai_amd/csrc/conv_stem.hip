// The image-input 7 x 7 stride-2 convolution of the ResNet-18 visual branch (networks/Image_Embedding.py:20 `conv1 = nn.Conv2d(3 | 2, 64,
// kernel_size=7, stride=2, padding=3, bias=False)`, run on 2 x 1024 frames of 224 x 224 per step): forward and weight gradient on the
// fp16 matrix cores through the f16x2 operand split (gfx950).
//
// The frames are stored NHWC with four channels per pixel (the fourth, and for optical flow the third too, are zeros), so the seven
// taps of a kernel ROW and their channels are ONE contiguous run of memory: output pixel (oy, ox) and kernel row r read the 8 input
// pixels ix = 2 ox - 3 .. 2 ox + 4 of input row 2 oy - 3 + r, 32 consecutive floats (the eighth pixel meets zero weights).  That makes
// the layer a GEMM with K = 7 rows x 32:
//
//   y[p][co]      = sum_r sum_k  xrun(p, r)[k] * W[co][r][k]                 M = pixels, N = 64, K = 224
//   dW[r][co][k]  = sum_p        dy[p][co]     * xrun(p, r)[k]               M = 64, N = 7 x 32, K = pixels
//
// and the runs of neighbouring output pixels overlap: an 8 x 16 output tile reads a 21 x 38 pixel patch (12.8 KB), staged ONCE per tile
// as two fp16 planes [row][pixel][4 channels]; the run of (ox, r) starts 16 bytes after the run of (ox - 1, r), so an MFMA operand
// fragment is one 16-byte LDS read (forward) or one transposing read with a 16-byte "row" pitch (weight gradient: a Toeplitz operand).
//
// The exact-fp32 gather kernel these layers ran on (conv_igemm_kernel in row-run mode, wgrad_mfma_kernel) re-read every input pixel once
// per kernel row and ran at the fp32 MFMA rate: forward 4.6 ms, weight gradient 5.5 - 7.0 ms per network on 1024 frames; the output
// tensor alone (3.3 GB) is 0.6 ms of HBM time, which is the floor these kernels are built against.
//
// * stem_fwd_f16_kernel: persistent blocks of four waves; the whole filter (64 x 7 x 32 f16x2 = 112 VGPRs per wave for its 32 output
//   channels) lives in registers for the life of the block; per tile 84 MFMAs per wave; epilogue = NHWC stores + BatchNorm partials
//   (mean, M2 of the tile's 128 pixels per channel, the layout viai_bn_finalize merges).
// * stem_wgrad_f16_kernel: persistent blocks of four waves (2 along Cout x 2 that split the tile rows); a wave keeps one 32 x 32
//   accumulator per kernel row (112 AGPRs) over ALL the tiles of its block and writes one slab at the end (deterministic slab reduce).
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int ST_TH = 8, ST_TW = 16;             // output tile
constexpr int ST_K = 7;                          // kernel rows (= taps of the GEMM)
constexpr int ST_PW = 2 * ST_TW + 6;             // 38 input pixels per patch row
constexpr int ST_PITCH = ST_PW * 8;              // bytes per patch row and plane (4 channels x fp16 per pixel)
constexpr int ST_COUT = 64;
constexpr int OOB = 0x7fffffff;

// ------------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_fwd_f16_kernel(const ConvArgs a, int ntiles) {
    constexpr int PH = 2 * ST_TH + 5;             // 21 patch rows
    constexpr int NPIX = PH * ST_PW;              // 798 pixels of 16 bytes
    constexpr int NL = (NPIX + 255) / 256;
    constexpr int PLANE = PH * ST_PITCH;
    constexpr int STAGE = 2 * PLANE;
    constexpr int BLO = ST_K * 2 * 64 * 16;       // the remainder-term fragments of one channel tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];     // [2 stages][2 planes] patch, [2][2][64] floats, [2 channel tiles] remainder fragments
    float* red = reinterpret_cast<float*>(smem_s + 2 * STAGE);
    unsigned char* blo = smem_s + 2 * STAGE + 1024;

    const ConvGeom& g = a.g;
    const float ascale = a.amax != nullptr ? f16_scale_from_amax(a.amax) : F16_ASCALE;
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;      // tile rows 4 wm .. 4 wm + 3, output channels 32 wn .. 32 wn + 31
    const int tiles_x = g.OW / ST_TW, tiles_y = g.OH / ST_TH;
    const int img_bytes = g.IH * g.IW * 16;

    // the filter: fragment-major f16x2 image [plane][channel tile][row][k-step][lane][8] (pack_run_f16_kernel).  Leading terms in
    // registers, remainder terms in LDS (one 16-byte read per k-step): the registers go to a second pair of accumulators, see below
    u32x4 B[ST_K][2];
    {
        const unsigned short* wp = reinterpret_cast<const unsigned short*>(a.wp);
        constexpr int plane = 2 * ST_K * 2 * 512;
#pragma unroll
        for (int r = 0; r < ST_K; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                B[r][s] = *reinterpret_cast<const u32x4*>(wp + ((wn * ST_K + r) * 2 + s) * 512 + lane * 8);
                if (wm == 0)
                    *reinterpret_cast<u32x4*>(blo + wn * BLO + ((r * 2 + s) * 64 + lane) * 16) = *reinterpret_cast<const u32x4*>(wp + plane + ((wn * ST_K + r) * 2 + s) * 512 + lane * 8);
            }
    }
    const unsigned char* blo_w = blo + wn * BLO + lane * 16;

    // patch staging: thread -> patch pixels tid + 256 j
    int ppos[NL];                                 // (row << 8) | column
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int it = tid + 256 * j;
        const int pr = it / ST_PW;
        ppos[j] = (pr << 8) | (it - pr * ST_PW);
    }
    u32x4 raw[NL];
    auto gload = [&](int tile_) {
        const int tile = __builtin_amdgcn_readfirstlane(tile_);
        const int tx = tile % tiles_x; const int r_ = tile / tiles_x; const int ty = r_ % tiles_y, n = r_ / tiles_y;
        const int iy0 = 2 * ty * ST_TH - 3, ix0 = 2 * tx * ST_TW - 3;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * g.IH * g.IW * 4), 0, img_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int iy = iy0 + (ppos[j] >> 8), ix = ix0 + (ppos[j] & 255);
            const int dead = (((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | (NPIX - 1 - (tid + 256 * j))) >> 31) & OOB;
            raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((iy * g.IW + ix) * 16) | dead, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (256 * (j + 1) <= NPIX || tid + 256 * j < NPIX) {
                const f32x4 v = __builtin_bit_cast(f32x4, raw[j]);
                unsigned a1, a2, b1, b2;
                split2_pair(v[0], v[1], ascale, alim, a1, a2);
                split2_pair(v[2], v[3], ascale, alim, b1, b2);
                const u32x2 p1 = {a1, b1}, p2 = {a2, b2};
                unsigned char* d = smem_s + buf * STAGE + (ppos[j] >> 8) * ST_PITCH + (ppos[j] & 255) * 8;
                *reinterpret_cast<u32x2*>(d) = p1;
                *reinterpret_cast<u32x2*>(d + PLANE) = p2;
            }
    };

    // MFMA row m = lane & 31 of M-tile i of this wave -> tile pixel (4 wm + 2 i + (m >> 4), m & 15); k-half h = lane >> 5 -> run pixels 4 s + 2 h, + 1
    const int aoff = (8 * wm + 2 * ((lane & 31) >> 4)) * ST_PITCH + (2 * (lane & 15) + 2 * (lane >> 5)) * 8;
    const float inv = 1.0f / (ascale * F16_WSCALE);
    const int half = lane >> 5, col = lane & 31;
    const int co = wn * 32 + col;

    int tile = blockIdx.x;
    const int stride = gridDim.x;
    if (tile < ntiles) { gload(tile); lstore(0); }
    __syncthreads();
    for (int k = 0; tile < ntiles; tile += stride, ++k) {
        const int nxt = tile + stride;
        if (nxt < ntiles) gload(nxt);
        const unsigned char* Sb = smem_s + (k & 1) * STAGE + aoff;
        f32x16 acc[2], accx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][e] = 0.f; accx[i][e] = 0.f; }
#pragma unroll
        for (int r = 0; r < ST_K; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 af[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p) af[i][p] = *reinterpret_cast<const u32x4*>(Sb + p * PLANE + (4 * i + r) * ST_PITCH + s * 32);
                const u32x4 bl = *reinterpret_cast<const u32x4*>(blo_w + (r * 2 + s) * 1024);
                // The cross terms (remainder x leading) are 2^-11 of the leading products.  Added to the SAME accumulator, the fp16 MFMA aligns
                // them to the running sum and drops their low bits downwards: a negative bias of ~0.2 accumulator ulp per output, the same
                // sign everywhere -- harmless per element (rms error 1.7e-7), but it adds up coherently in the cancelling sums of the layers
                // behind (the 4-frame reference golden: first-layer gradients 1e-2 off instead of 7e-4).  In their own accumulators the cross
                // terms are summed among equals and join the leading sum once: bias / 15, rms error 1.2e-7 (tools/stem_bias.py).
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][1]), __builtin_bit_cast(f16x8, B[r][s]), accx[i], 0, 0, 0);
                    accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, bl), accx[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, B[r][s]), acc[i], 0, 0, 0);
                }
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] += accx[i][e];
        // ---- epilogue of the tile
        const int tx = tile % tiles_x; const int r_ = tile / tiles_x; const int ty = r_ % tiles_y, n = r_ / tiles_y;
        const int oy0 = ty * ST_TH + 4 * wm, ox0 = tx * ST_TW;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                const int oy = oy0 + 2 * i + (row >> 4), ox = ox0 + (row & 15);
                float v = acc[i][e] * inv;
                if (a.bias != nullptr) v += a.bias[co];
                if (a.stat == nullptr) v = viai_act(v, a.act, a.slope);
                acc[i][e] = v;
                a.out[(((size_t)n * g.OH + oy) * g.OW + ox) * ST_COUT + co] = v;
            }
        if (a.stat != nullptr) {                  // (mean, M2) of the tile's 128 pixels per channel: two-pass per wave (64 pixels), Chan merge of the two
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][e];
            t += __shfl_xor(t, 32, 64);
            const float mw = t * (1.0f / 64.f);
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = acc[i][e] - mw; m2 += d * d; }
            m2 += __shfl_xor(m2, 32, 64);
            if (half == 0) { red[(wm * 2 + 0) * 64 + co] = mw; red[(wm * 2 + 1) * 64 + co] = m2; }
            __syncthreads();
            if (wm == 0 && half == 0) {
                const float m0 = red[co], m1 = red[2 * 64 + co];
                const float mean = 0.5f * (m0 + m1);
                const float d0 = m0 - mean, d1 = m1 - mean;
                a.stat[(size_t)co * ntiles + tile] = mean;
                a.stat[(size_t)(ST_COUT + co) * ntiles + tile] = red[64 + co] + red[3 * 64 + co] + 64.f * (d0 * d0 + d1 * d1);
            }
        }
        if (nxt < ntiles) lstore((k & 1) ^ 1);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------- weight gradient
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_wgrad_f16_kernel(const WgradArgs a, int ntiles, int tiles_per_block) {
    constexpr int HR = 4;                         // output rows per stage (half a tile)
    constexpr int PH = 2 * HR + 5;                // 13 patch rows
    constexpr int XPIX = PH * ST_PW;              // 494
    constexpr int NX = (XPIX + 255) / 256;        // 2
    constexpr int XPLANE = PH * ST_PITCH;
    constexpr int DROW = ST_COUT * 2;             // dy LDS row: 64 channels = 128 B, the two 64-byte groups swapped by bit 1 of the pixel
    constexpr int DPIX = HR * ST_TW;              // 64
    constexpr int DPLANE = DPIX * DROW;
    constexpr int ND = DPIX * 16 / 256;           // 4 float4 items per thread
    constexpr int STAGE = 2 * DPLANE + 2 * XPLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];     // [2 stages]{dy plane 0, dy plane 1, x plane 0, x plane 1}

    const ConvGeom& g = a.g;
    const float dscale = f16_scale_from_amax(a.amax), dlim = f16_clamp_for_scale(dscale);
    const float xscale = a.xmax != nullptr ? f16_scale_from_amax(a.xmax) : F16_ASCALE, xlim = f16_clamp_for_scale(xscale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cm = wave & 1, kh = wave >> 1;      // output channels 32 cm .., tile rows kh, kh + 2 of a stage
    const int tiles_x = g.OW / ST_TW, tiles_y = g.OH / ST_TH;
    const int img_bytes = g.IH * g.IW * 16;
    const int tile0 = blockIdx.x * tiles_per_block;
    int tile1 = tile0 + tiles_per_block; if (tile1 > ntiles) tile1 = ntiles;
    const int nst = tile1 > tile0 ? (tile1 - tile0) * 2 : 0;

    // staging maps
    const int dq = tid & 15, dp0 = tid >> 4;      // dy item j: pixel (row j, column dp0) of the stage, channel quad dq
    const int d_goff = (dp0 * ST_COUT + dq * 4) * 4;
    const int d_lds = dp0 * DROW + (((dq >> 3) ^ ((dp0 >> 1) & 1)) * 64) + (dq & 7) * 8;
    int xpos[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int it = tid + 256 * j;
        const int pr = it / ST_PW;
        xpos[j] = (pr << 8) | (it - pr * ST_PW);
    }
    u32x4 draw[ND], xraw[NX];
    auto gload = [&](int s_) {
        const int s = __builtin_amdgcn_readfirstlane(s_);
        const int tile = tile0 + (s >> 1), h = s & 1;
        const int tx = tile % tiles_x; const int r_ = tile / tiles_x; const int ty = r_ % tiles_y, n = r_ / tiles_y;
        const int oy0 = ty * ST_TH + h * HR, ox0 = tx * ST_TW;
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (((size_t)n * g.OH + oy0) * g.OW + ox0) * ST_COUT), 0,
                                                                             ((HR - 1) * g.OW + ST_TW) * ST_COUT * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < ND; ++j) draw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, d_goff + j * g.OW * ST_COUT * 4, 0, 0);
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * g.IH * g.IW * 4), 0, img_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int iy = iy0 + (xpos[j] >> 8), ix = ix0 + (xpos[j] & 255);
            const int dead = (((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | (XPIX - 1 - (tid + 256 * j))) >> 31) & OOB;
            xraw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ((iy * g.IW + ix) * 16) | dead, 0, 0);
        }
    };
    auto put = [&](unsigned char* d, int plane, const u32x4& raw, float Sc, float L) {
        const f32x4 v = __builtin_bit_cast(f32x4, raw);
        unsigned a1, a2, b1, b2;
        split2_pair(v[0], v[1], Sc, L, a1, a2);
        split2_pair(v[2], v[3], Sc, L, b1, b2);
        const u32x2 p1 = {a1, b1}, p2 = {a2, b2};
        *reinterpret_cast<u32x2*>(d) = p1;
        *reinterpret_cast<u32x2*>(d + plane) = p2;
    };
    auto lstore = [&](int buf) {
        unsigned char* base = smem_w + buf * STAGE;
#pragma unroll
        for (int j = 0; j < ND; ++j) put(base + d_lds + j * ST_TW * DROW, DPLANE, draw[j], dscale, dlim);
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if (256 * (j + 1) <= XPIX || tid + 256 * j < XPIX)
                put(base + 2 * DPLANE + (xpos[j] >> 8) * ST_PITCH + (xpos[j] & 255) * 8, XPLANE, xraw[j], xscale, xlim);
    };

    f32x16 acc[ST_K];
#pragma unroll
    for (int r = 0; r < ST_K; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // fragment addresses.  16-lane group grp, lane li in it: row block m0 = 16 (grp & 1) of the operand's 32 rows, pixels kb + 4 rd + (li >> 2)
    const int grp = lane >> 4, li = lane & 15;
    const int m0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const int a_lane = (kb + (li >> 2)) * DROW + ((cm ^ ((li >> 3) & 1)) * 64) + m0 * 2 + (li & 3) * 8;
    // Toeplitz operand: element (pixel ox, run element e) of a patch row sits at half-word 8 ox + e
    const int b_lane = 2 * DPLANE + (kb + (li >> 2)) * 16 + m0 * 2 + (li & 3) * 8;
    auto frag = [&](const unsigned char* p, int rowpitch) -> f16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * rowpitch));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto kstep = [&](const unsigned char* Sb, int row) {          // tile row `row` of the stage: 16 pixels, seven kernel rows
        f16x8 af[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) af[p] = frag(Sb + a_lane + p * DPLANE + row * ST_TW * DROW, DROW);
#pragma unroll
        for (int r = 0; r < ST_K; ++r) {
            f16x8 bq[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) bq[p] = frag(Sb + b_lane + p * XPLANE + (2 * row + r) * ST_PITCH, 16);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bq[1], acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bq[0], acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bq[0], acc[r], 0, 0, 0);
        }
    };

    if (nst > 0) { gload(0); lstore(0); }
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        if (s + 1 < nst) gload(s + 1);
        const unsigned char* Sb = smem_w + (s & 1) * STAGE;
        kstep(Sb, kh);
        kstep(Sb, kh + 2);
        if (s + 1 < nst) lstore((s & 1) ^ 1);
        __syncthreads();
    }

    // ---- the two waves of a channel tile hold partial sums over different tile rows: add them through LDS (28 KB per pair, one pair at a time)
    float* red = reinterpret_cast<float*>(smem_w);
    for (int c = 0; c < 2; ++c) {
        if (cm == c && kh == 1) {
#pragma unroll
            for (int r = 0; r < ST_K; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(r * 16 + e) * 64 + lane] = acc[r][e];
        }
        __syncthreads();
        if (cm == c && kh == 0) {
#pragma unroll
            for (int r = 0; r < ST_K; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][e] += red[(r * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (kh != 0) return;
    // ---- slab [block][r][co][32]
    const float inv = 1.0f / (dscale * xscale);
    const int half = lane >> 5, col = lane & 31;
    float* dst = a.ws + (size_t)blockIdx.x * ST_K * ST_COUT * 32;
#pragma unroll
    for (int r = 0; r < ST_K; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            dst[((size_t)r * ST_COUT + cm * 32 + row) * 32 + col] = acc[r][e] * inv;
        }
}

// dw[co][ch][r][s] (+)= sum_z ws[z][r][co][s * 4 + ch]: one block per (r, co), eight partial sums per element added in a fixed order
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nz, int Cin, int accumulate) {
    __shared__ float part[8][32];
    const int r = blockIdx.x / ST_COUT, co = blockIdx.x % ST_COUT;
    const int zp = threadIdx.x >> 5, k = threadIdx.x & 31;
    const size_t slab = (size_t)ST_K * ST_COUT * 32;
    const float* src = ws + ((size_t)r * ST_COUT + co) * 32 + k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = zp;
    for (; z + 24 < nz; z += 32) {
        s0 += src[(size_t)z * slab]; s1 += src[(size_t)(z + 8) * slab]; s2 += src[(size_t)(z + 16) * slab]; s3 += src[(size_t)(z + 24) * slab];
    }
    for (; z < nz; z += 8) s0 += src[(size_t)z * slab];
    part[zp][k] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zp == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += part[i][k];
        const int sx = k >> 2, ch = k & 3;
        if (sx < ST_K && ch < Cin) {
            float* d = dw + (((size_t)co * Cin + ch) * ST_K + r) * ST_K + sx;
            *d = accumulate ? *d + t : t;
        }
    }
}

// fragment-major f16x2 image of the run weights: W[co][r][k = 4 s + ch] = w[co][ch][r][s] (zero for s = 7 and ch >= Cin)
__global__ void pack_run_f16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin) {
    constexpr int plane = 2 * ST_K * 2 * 512;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const int e = i & 7, lane = (i >> 3) & 63;
        int rr = i >> 9;
        const int kq = rr & 1; rr >>= 1;
        const int r = rr % ST_K, nt = rr / ST_K;
        const int co = nt * 32 + (lane & 31), k = kq * 16 + (lane >> 5) * 8 + e;
        const int sx = k >> 2, ch = k & 3;
        float x = 0.f;
        if (sx < ST_K && ch < Cin) x = __builtin_amdgcn_fmed3f(w[(((size_t)co * Cin + ch) * ST_K + r) * ST_K + sx] * F16_WSCALE, -65504.f, 65504.f);
        const _Float16 h1 = (_Float16)x;
        const _Float16 h2 = (_Float16)(x - (float)h1);
        wp[i] = __builtin_bit_cast(unsigned short, h1);
        wp[plane + i] = __builtin_bit_cast(unsigned short, h2);
    }
}

constexpr int STEM_WGRAD_BLOCKS = 512;

}  // namespace

// 7 x 7, stride 2, padding 3, 2 - 4 input channels stored four per pixel, 64 output channels, output extent a whole number of 8 x 16 tiles
bool viai_conv_stem_ok(const ConvGeom& g, int Cin, int Cout, int kh, int kw, int sh, int sw, int ph, int pw) {
    if (Cin < 2 || Cin > 4 || Cout != ST_COUT || kh != ST_K || kw != ST_K || sh != 2 || sw != 2 || ph != 3 || pw != 3) return false;
    if (g.OH % ST_TH != 0 || g.OW % ST_TW != 0 || g.IH != 2 * g.OH || g.IW != 2 * g.OW) return false;
    return (long)g.IH * g.IW * 16 < (1l << 31) && (long)g.N * (g.OH / ST_TH) * (g.OW / ST_TW) < (1l << 30);
}

int viai_conv_stem_fwd_launch(ConvArgs& a, hipStream_t st) {
    const ConvGeom& g = a.g;
    const int ntiles = g.N * (g.OH / ST_TH) * (g.OW / ST_TW);
    constexpr int lds = 2 * 2 * (2 * ST_TH + 5) * ST_PITCH + 2 * 2 * 64 * 4 + 2 * ST_K * 2 * 64 * 16;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
    a.nblk_m = ntiles; a.nblk_n = 1;
    int grid = 512;
    if (grid > ntiles) grid = ntiles;
    viai_tag_kernel("stem_f16x2");
    VIAI_LAUNCH(stem_fwd_f16_kernel, dim3(grid), dim3(256), lds, st, a, ntiles);
    return viai_launch_status();
}

int viai_conv_stem_wgrad_slabs(const ConvGeom& g) {
    const int ntiles = g.N * (g.OH / ST_TH) * (g.OW / ST_TW);
    int blocks = STEM_WGRAD_BLOCKS;
    if (blocks > ntiles) blocks = ntiles;
    const int per = (ntiles + blocks - 1) / blocks;
    return (ntiles + per - 1) / per;                  // no empty block
}

// slabs -> a.ws ([slab][7][64][32]), then dw[co][ch][7][7] (+)= their sum
int viai_conv_stem_wgrad_launch(WgradArgs& a, int Cin, float* dw, int accumulate, hipStream_t st) {
    const ConvGeom& g = a.g;
    const int ntiles = g.N * (g.OH / ST_TH) * (g.OW / ST_TW);
    const int blocks = viai_conv_stem_wgrad_slabs(g);
    const int per = (ntiles + blocks - 1) / blocks;
    constexpr int lds = 2 * (2 * 64 * 128 + 2 * 13 * ST_PITCH);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
    viai_tag_kernel("wgrad_stem_f16x2");
    VIAI_LAUNCH(stem_wgrad_f16_kernel, dim3(blocks), dim3(256), lds, st, a, ntiles, per);
    VIAI_LAUNCH(stem_wgrad_reduce_kernel, dim3(ST_K * ST_COUT), dim3(256), 0, st, (const float*)a.ws, dw, blocks, Cin, accumulate);
    return viai_launch_status();
}

int viai_conv_stem_pack(const float* w, float* wp, int Cin, hipStream_t st) {
    VIAI_LAUNCH(pack_run_f16_kernel, dim3(56), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wp), Cin);
    return viai_launch_status();
}
