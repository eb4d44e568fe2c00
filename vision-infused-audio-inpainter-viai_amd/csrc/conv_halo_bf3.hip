// Stride-1 small-channel convolution with an LDS-resident input tile ("halo" kernel), bf16x3 split MFMA, gfx950.
//
// The gather-GEMM kernels re-read (and re-split into bf16 terms) every input pixel once per tap.  For the 32/64
// channel 3x3 layers at 256x256 / 128x128 that repetition -- not the MFMAs -- is the bound.  Here a block owns an
// 8 x 16 output tile: it loads the (8+2) x (16+2) input patch ONCE (coalesced float4 rows, hardware zero fill for
// the padding), splits it into three bf16 planes in LDS and then walks the taps: the A operand of tap (dy, dx) is
// the same LDS tile read at a per-lane row offset.  Weights are fetched fragment-major straight from global
// memory (see conv_igemm_bf3.hip).  Same epilogue contract as the other conv kernels (bias/act or BN partials).
// Reference call sites: the 32/64-channel 3x3 stride-1 Conv2d / ConvTranspose2d layers of
// Inpainting_Networks.py:60-110 (MelEncoder/MelDecoder) and their autograd data gradients.
#include "viai_common.h"
#include "viai_internal.h"
#include <string>
#include "viai_bf3.h"
#include <type_traits>

namespace {

constexpr int HT_H = 8, HT_W = 16;                 // output tile (128 pixels = 4 waves x 32 MFMA rows)

constexpr int HT_HH = HT_H + 2, HT_HW = HT_W + 2, HT_HP = HT_HH * HT_HW;   // staged patch: 10 x 18 pixels (taps span <= 3 x 3)
constexpr int HALO_WIDE_ROW_BYTES = 1536;       // conv_halo_wide_f16_kernel, stride 1: LDS bytes per patch row (18 pixels x 80 bytes, padded to a multiple of 256)

// NP = 3: bf16x3; NP = 2: f16x2 (two fp16 planes, three partial products; operand scale static or from a.amax)
// P16 (NP = 2 only): the gathered tensor arrives pre-split (viai_bf3.h): pieces are copied into their plane instead of quads being split
// MY (round 6): vertical gather stride.  MY = 2 is the forward of a stride-(2, 1) conv (MelEncoder.conv2, Inpainting_Networks.py:58): the patch of an 8 x 16 output tile is
// 17 x 18 input pixels and output row r reads patch rows 2 r + dy.  The data gradient of such a layer runs as its two row-parity classes (unit gather stride, scatter
// stride ly = 2): the epilogue writes output pixel (oy ly + ay, ox lx + ax) of the full tensor, tiles are counted on the class's sub-lattice (SH x SW).
template <int CIN, int TN, int NP = 3, bool P16 = false, int MY = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CIN == 32 && TN == 1) ? 3 : 2))) void conv_halo_bf3_kernel(const ConvArgs a, int hy0, int hx0, int ntiles) {
    static_assert(!P16 || NP == 2, "P16 is an f16x2 format");
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int PITCH = CIN * 2 + 16;             // bytes per LDS pixel row (80 / 144: conflict-free ds_read_b128)
    constexpr int Q = CIN / 4;                      // float4 per pixel
    constexpr int HT_HH = (HT_H - 1) * MY + 3, HT_HP = HT_HH * HT_HW;      // staged patch (shadows the stride-1 constants of the file scope)
    constexpr int NL = (HT_HP * Q + 255) / 256;     // float4 per thread
    constexpr int KS = CIN / 16;                    // 16-deep k-steps per tap
    constexpr int BN = 32 * TN;
    constexpr int PLANE = HT_HP * PITCH;            // bytes per bf16 plane: an immediate offset of the ds_reads
    constexpr int KH = 2, U = KS / KH;              // weight prefetch unit = KH k-steps of one tap

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [3][HP][PITCH]

    const ConvGeom& g = a.g;
    const float ascale = (NP == 2 && a.amax != nullptr) ? f16_scale_from_amax(a.amax) : F16_ASCALE;
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = g.SW / HT_W, tiles_y = g.SH / HT_H;          // (SH x SW = OH x OW except for a data-gradient parity class)

    constexpr int OOB = 0x7fffffff;
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)(in_pixels * CIN * 4), 0x00020000);
    const int NT = (a.Cout + 31) / 32;
    const int frag_plane = NT * g.wtaps * KS * 1024;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, NP * frag_plane, 0x00020000);

    // patch staging: thread -> channel quad q of patch pixels h0 + (256 / Q) * j
    const int h0 = tid / Q, q = tid % Q;
    int poff[NL], prc[NL];                          // in-image byte offset relative to the patch origin; (row << 8) | col
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int h = h0 + (256 / Q) * j;
        const int hr = h / HT_HW, hc = h - hr * HT_HW;
        poff[j] = (hr * g.IW + hc) * CIN * 4 + q * 16;
        prc[j] = h < HT_HP ? (hr << 8) | hc : -1;
    }
    u32x4 reg[NL];
    auto load_patch = [&](int tile) {               // tile is wave-uniform; loads land in reg[] while the MFMAs run
        const int tx = tile % tiles_x; int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int py = ty * HT_H * MY + hy0, px = tx * HT_W + hx0;
        const int pbase = ((n * g.IH + py) * g.IW + px) * CIN * 4;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int iy = py + (prc[j] >> 8), ix = px + (prc[j] & 255);
            const bool ok = prc[j] >= 0 && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
            reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? pbase + poff[j] : OOB, 0, 0);
        }
    };
    auto store_patch = [&]() {                      // split into three bf16 planes on the way into LDS
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if (prc[j] >= 0) {
                const f32x4 v = __builtin_bit_cast(f32x4, reg[j]);
                unsigned char* d = smem_h + (h0 + (256 / Q) * j) * PITCH + q * 8;
                if constexpr (NP == 3) {
                    unsigned a1, a2, a3, b1, b2, b3;
                    split3_pair(v[0], v[1], a1, a2, a3);
                    split3_pair(v[2], v[3], b1, b2, b3);
                    const u32x2 p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3};
                    *reinterpret_cast<u32x2*>(d) = p1;
                    *reinterpret_cast<u32x2*>(d + PLANE) = p2;
                    *reinterpret_cast<u32x2*>(d + 2 * PLANE) = p3;
                } else {
                    stage_put32<P16>(smem_h + (h0 + (256 / Q) * j) * PITCH + (q >> 3) * 64, PLANE, q & 7, reg[j], ascale, alim);
                }
            }
        }
    };

    // weights: fragment-major planes [p][nt][tap][kq][lane] x 16 B; the lane part is the only per-lane offset
    int bvoff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bvoff[j] = (j < NT) ? j * g.wtaps * KS * 1024 + lane * 16 : OOB;
    const int nunits = g.ntaps * U;
    auto gloadB = [&](u32x4 (&bf)[KH][TN][NP], int u_) {
        const int u = __builtin_amdgcn_readfirstlane(u_);
        if (u < nunits) {
            const int t = u / U, k0 = (u % U) * KH;
            const int soff = (g.ws[t] * KS + k0) * 1024;
#pragma unroll
            for (int ks = 0; ks < KH; ++ks)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        bf[ks][j][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bvoff[j], soff + ks * 1024 + p * frag_plane, 0);
        }
    };

    // MFMA row i = lane & 31 -> tile pixel (2 * wave + (i >> 4), i & 15)
    const int pr = 2 * wave + ((lane & 31) >> 4), pc = lane & 15;
    const unsigned char* abase = smem_h + ((pr * MY - hy0) * HT_HW + (pc - hx0)) * PITCH + 16 * (lane >> 5);
    constexpr int PA[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0}, PB[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
    const int half = lane >> 5, col = lane & 31;
    float bv[TN];
    int co[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        co[j] = j * 32 + col;
        bv[j] = (a.bias != nullptr && co[j] < a.Cout) ? a.bias[co[j]] : 0.f;
    }
    const int oc2 = a.Cout - a.OC1;

    // persistent over tiles: block b takes logical tiles b, b + grid, ... (XCD-contiguous after the remap)
    constexpr bool PREF = !(CIN == 64 && TN == 2);      // next patch prefetched into registers during the MFMAs (register budget)
    int it = blockIdx.x;
    if (PREF && it < ntiles) load_patch(xcd_remap(it, ntiles));
    for (; it < ntiles; it += gridDim.x) {
        const int tile = xcd_remap(it, ntiles);
        if (!PREF) load_patch(tile);
        store_patch();
        __syncthreads();
        u32x4 bfa[KH][TN][NP], bfb[KH][TN][NP];         // two units of weight fragments in flight
        gloadB(bfa, 0);
        gloadB(bfb, 1);
        if (PREF && it + (int)gridDim.x < ntiles) load_patch(xcd_remap(it + gridDim.x, ntiles));

        // NC independent accumulation chains per output tile: a dependent MFMA cannot issue until its predecessor
        // retires, so a single chain would run the matrix pipe at half rate
        constexpr int NC = 2;
        f32x16 accc[TN][NC];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) accc[j][c][e] = 0.f;
        auto mma = [&](u32x4 (&bf)[KH][TN][NP], int u_) {
            const int u = __builtin_amdgcn_readfirstlane(u_);
            const int t = u / U, k0 = (u % U) * KH;
            const unsigned char* As = abase + (g.dy[t] * HT_HW + g.dx[t]) * PITCH + k0 * 32;
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) {
                u32x4 af[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) af[p] = *reinterpret_cast<const u32x4*>(As + p * PLANE + ks * 32);
#pragma unroll
                for (int k = 0; k < NPROD; ++k)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (NP == 3) accc[j][k % NC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[PA[k]]), __builtin_bit_cast(bf16x8, bf[ks][j][PB[k]]), accc[j][k % NC], 0, 0, 0);
                        else accc[j][k % NC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[PA[k]]), __builtin_bit_cast(f16x8, bf[ks][j][PB[k]]), accc[j][k % NC], 0, 0, 0);
                    }
            }
        };
#pragma unroll 1
        for (int u = 0; u < nunits; u += 2) {
            mma(bfa, u);
            gloadB(bfa, u + 2);
            if (u + 1 < nunits) {
                mma(bfb, u + 1);
                gloadB(bfb, u + 3);
            }
        }
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[j] = accc[j][0];
#pragma unroll
            for (int c = 1; c < NC; ++c) acc[j] += accc[j][c];
            if constexpr (NP == 2) acc[j] *= 1.0f / (ascale * F16_WSCALE);       // undo the operand scales (exact powers of two)
        }

        // ------------------------------------------------------------ epilogue of this tile
        const int tx = tile % tiles_x; int r_ = tile / tiles_x;
        const int ty = r_ % tiles_y, n = r_ / tiles_y;
        const int oy0 = ty * HT_H, ox0 = tx * HT_W;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int oy = (oy0 + 2 * wave + (row >> 4)) * g.ly + g.ay, ox = (ox0 + (row & 15)) * g.lx + g.ax;
            const size_t opix = ((size_t)n * g.OH + oy) * g.OW + ox;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[j][e] + bv[j];
                if (a.stat == nullptr) v = viai_act(v, a.act, a.slope);
                acc[j][e] = v;
                if (co[j] < a.Cout) {
                    if (co[j] < a.OC1) a.out[opix * a.OC1 + co[j]] = v;
                    else a.out2[opix * oc2 + (co[j] - a.OC1)] = v;
                }
            }
        }
        __syncthreads();                  // every wave is done reading the patch
        if (a.stat != nullptr) {          // block-local (mean, M2) over the 128 pixels of this tile
            float* red = reinterpret_cast<float*>(smem_h);          // [4 waves][BN]
            float s[TN], mean[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[j][e];
                t += __shfl_xor(t, 32, 64);
                s[j] = t;
            }
            if (half == 0)
#pragma unroll
                for (int j = 0; j < TN; ++j) red[wave * BN + co[j]] = s[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TN; ++j) mean[j] = (red[co[j]] + red[BN + co[j]] + red[2 * BN + co[j]] + red[3 * BN + co[j]]) * (1.f / 128.f);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float t = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { float d = acc[j][e] - mean[j]; t += d * d; }
                t += __shfl_xor(t, 32, 64);
                s[j] = t;
            }
            if (half == 0)
#pragma unroll
                for (int j = 0; j < TN; ++j) red[wave * BN + co[j]] = s[j];
            __syncthreads();
            if (wave == 0 && half == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (co[j] < a.Cout) {
                        float t = red[co[j]] + red[BN + co[j]] + red[2 * BN + co[j]] + red[3 * BN + co[j]];
                        a.stat[(size_t)co[j] * a.nblk_m + tile] = mean[j];
                        a.stat[(size_t)(a.Cout + co[j]) * a.nblk_m + tile] = t;
                    }
            }
            __syncthreads();              // red[] is overwritten by the next patch
        }
    }
}

// f16x2 variant for 32 -> (<= 32) channel 3 x 3 layers, weights resident in REGISTERS.  The bf16x3 kernel above streams every
// weight fragment from L2 once per tile and wave (54 KB per 128 pixels: at 2048+ tiles that stream, not the MFMAs or HBM, is
// the bound).  With two fp16 planes the whole filter is 9 taps x 2 k-steps x 2 planes = 36 fragments = 144 VGPRs, fetched once
// per persistent block; a tile is then: patch (10 x 18 x 32 fp32, prefetched during the previous tile) -> split into two fp16
// planes in LDS -> 54 MFMAs per wave whose A operands are ds_read_b128 at compile-time offsets -> epilogue.
struct HaloSlots { int s[9]; };                    // weight slot of window position (row * 3 + col)

template <bool P16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_halo_f16_c32_kernel(const ConvArgs a, int hy0, int hx0, int ntiles, HaloSlots slots) {
    constexpr int CIN = 32, PITCH = 80, Q = 8, KS = 2;
    constexpr int NL = (HT_HP * Q + 255) / 256;     // 6 float4 per thread
    constexpr int PLANE = HT_HP * PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2][HP][PITCH]

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = g.OW / HT_W, tiles_y = g.OH / HT_H;
    const float ascale = a.amax != nullptr ? f16_scale_from_amax(a.amax) : F16_ASCALE;
    const float alim = f16_clamp_for_scale(ascale);

    constexpr int OOB = 0x7fffffff;
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)(in_pixels * CIN * 4), 0x00020000);
    const int frag_plane = g.wtaps * KS * 1024;      // one 32-channel output tile
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 2 * frag_plane, 0x00020000);

    // the whole filter, once per block: wf[position][k-step][plane]
    u32x4 wf[9][KS][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                wf[t][ks][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, (slots.s[t] * KS + ks) * 1024 + p * frag_plane, 0);

    const int h0 = tid / Q, q = tid % Q;
    int poff[NL], prc[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int h = h0 + (256 / Q) * j;
        const int hr = h / HT_HW, hc = h - hr * HT_HW;
        poff[j] = (hr * g.IW + hc) * CIN * 4 + q * 16;
        prc[j] = h < HT_HP ? (hr << 8) | hc : -1;
    }
    u32x4 reg[NL];
    auto load_patch = [&](int tile) {
        const int tx = tile % tiles_x; int r = tile / tiles_x;
        const int ty = r % tiles_y, n = r / tiles_y;
        const int py = ty * HT_H + hy0, px = tx * HT_W + hx0;
        const int pbase = ((n * g.IH + py) * g.IW + px) * CIN * 4;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int iy = py + (prc[j] >> 8), ix = px + (prc[j] & 255);
            const int dead = (((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | prc[j]) >> 31) & OOB;
            reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (pbase + poff[j]) | dead, 0, 0);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if (prc[j] >= 0) stage_put32<P16>(smem_h + (h0 + (256 / Q) * j) * PITCH, PLANE, q, reg[j], ascale, alim);
        }
    };

    // MFMA row i = lane & 31 -> tile pixel (2 * wave + (i >> 4), i & 15); the patch origin is the window's top-left tap
    const int pr = 2 * wave + ((lane & 31) >> 4), pc = lane & 15;
    const unsigned char* abase = smem_h + (pr * HT_HW + pc) * PITCH + 16 * (lane >> 5);
    const int half = lane >> 5, col = lane & 31;
    const float bv = (a.bias != nullptr && col < a.Cout) ? a.bias[col] : 0.f;
    const int oc2 = a.Cout - a.OC1;
    const float inv = 1.0f / (ascale * F16_WSCALE);

    int it = blockIdx.x;
    if (it < ntiles) load_patch(xcd_remap(it, ntiles));
    for (; it < ntiles; it += gridDim.x) {
        const int tile = xcd_remap(it, ntiles);
        store_patch();
        __syncthreads();
        if (it + (int)gridDim.x < ntiles) load_patch(xcd_remap(it + gridDim.x, ntiles));

        // two accumulation chains, alternating: a dependent MFMA waits for its predecessor to retire
        f32x16 c0, c1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned char* As = abase + ((t / 3) * HT_HW + (t % 3)) * PITCH + ks * 32;
                const f16x8 a1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(As));
                const f16x8 a2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(As + PLANE));
                const f16x8 b1 = __builtin_bit_cast(f16x8, wf[t][ks][0]), b2 = __builtin_bit_cast(f16x8, wf[t][ks][1]);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, c0, 0, 0, 0);      // small terms on one chain
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, c0, 0, 0, 0);
            }
        f32x16 acc = (c0 + c1) * inv;

        const int tx = tile % tiles_x; int r_ = tile / tiles_x;
        const int ty = r_ % tiles_y, n = r_ / tiles_y;
        const int oy0 = ty * HT_H, ox0 = tx * HT_W;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int oy = oy0 + 2 * wave + (row >> 4), ox = ox0 + (row & 15);
            const size_t opix = ((size_t)n * g.OH + oy) * g.OW + ox;
            float v = acc[e] + bv;
            if (a.stat == nullptr) v = viai_act(v, a.act, a.slope);
            acc[e] = v;
            if (col < a.Cout) {
                if (col < a.OC1) a.out[opix * a.OC1 + col] = v;
                else a.out2[opix * oc2 + (col - a.OC1)] = v;
            }
        }
        __syncthreads();                  // every wave is done reading the patch
        if (a.stat != nullptr) {          // block-local (mean, M2) over the 128 pixels of this tile
            // each wave: exact two-pass (mean, M2) of its 32 rows in registers; one LDS exchange; wave 0 merges the four with
            // Chan's formula (equal counts).  One barrier instead of four per tile.
            float* red = reinterpret_cast<float*>(smem_h);          // [4 waves][2][32]
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[e];
            t += __shfl_xor(t, 32, 64);
            const float mw = t * (1.f / 32.f);
            float m2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = acc[e] - mw; m2 += d * d; }
            m2 += __shfl_xor(m2, 32, 64);
            if (half == 0) { red[(wave * 2 + 0) * 32 + col] = mw; red[(wave * 2 + 1) * 32 + col] = m2; }
            __syncthreads();
            if (wave == 0 && half == 0 && col < a.Cout) {
                const float m0 = red[0 * 32 + col], m1 = red[2 * 32 + col], m2_ = red[4 * 32 + col], m3 = red[6 * 32 + col];
                const float mean = 0.25f * ((m0 + m1) + (m2_ + m3));
                float M2 = (red[1 * 32 + col] + red[3 * 32 + col]) + (red[5 * 32 + col] + red[7 * 32 + col]);
                M2 += 32.f * (((m0 - mean) * (m0 - mean) + (m1 - mean) * (m1 - mean)) + ((m2_ - mean) * (m2_ - mean) + (m3 - mean) * (m3 - mean)));
                a.stat[(size_t)col * a.nblk_m + tile] = mean;
                a.stat[(size_t)(a.Cout + col) * a.nblk_m + tile] = M2;
            }
            __syncthreads();              // red[] is overwritten by the next patch
        }
    }
}

// Wide-layer halo kernel (f16x2): the wide gather-GEMM kernel (conv_igemm_bf3.hip) loads and splits every input pixel once per
// TAP; for a stride-1 3 x 3 layer that is nine times per output-channel block.  Here a block owns an 8 x 16 output tile and 64 * WN
// output channels and walks K as (32-channel chunk, tap, k-step): per chunk the 10 x 18 pixel patch is loaded and split ONCE into
// a two-stage LDS buffer and all nine taps read it at compile-time offsets -- 6x fewer activation loads / splits per MFMA, the
// weight-fragment stream (fragment-major, straight from global, one (tap, k-step pair) ahead) unchanged.
struct HaloWideSlots { int s[9]; };
#ifndef VIAI_HALO_WIDE_NSET3
#define VIAI_HALO_WIDE_NSET3 1
#endif
#ifndef VIAI_HALO_WIDE_FENCED
#define VIAI_HALO_WIDE_FENCED 1
#endif

// WM x TM = 4 (the tile's eight rows = WM waves x TM row pairs); BN = 32 * TN * WN output channels per block:
//   <2,2,2,2> / <2,4,2,2>: 128 / 256 channels (wide layers);  <2,2,2,1>: 64 channels;  <4,1,1,1>: 32 channels
// S = 2: 3 x 3 STRIDE-2 forward conv.  The (17 x 33) input patch of the tile is stored as four parity sub-patches
//   P[py][px][r][c] = patch(2 r + py, 2 c + px), so that window position (ty, tx) of output pixel (r, c) is sub-patch (ty & 1, tx & 1) at
//   (r + (ty >> 1), c + (tx >> 1)): consecutive output pixels read consecutive 80-byte rows, as in the stride-1 case.  One LDS stage (98 KB).
template <int WM, int WN, int TM, int TN, int S = 1, bool P16 = false>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2))) void conv_halo_wide_f16_kernel(const ConvArgs a, int hy0, int hx0, HaloWideSlots slots) {
    constexpr int TH = 2 * WM * TM;                // tile rows (each 32-row MFMA tile = two tile rows of 16 pixels): 8, or 4 for the 64-pixel stride-2 tiles
    static_assert(TH == 8 || (TH == 4 && S == 2), "config");
    constexpr int NP = 2, BN = 32 * TN * WN, NTHR = 64 * WM * WN;
    constexpr int PH = S == 2 ? 2 * TH + 1 : TH + 2, PW = S == 2 ? 33 : HT_HW, NPIX = PH * PW;            // staged patch (input pixels)
    // Bytes between patch rows.  ds_read_b128 is serviced in four fixed lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... --
    // MI355X_MICROARCH.md, LDS): a group mixes columns {0-3, 12-15} of one tile row with columns {4-11} of the next, so the fragment
    // read is conflict-free exactly when the row pitch is a multiple of the 256-byte bank row.  18 x 80 = 1440 bytes was not: every group
    // ran two-way conflicted (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, profiles/r03_i_pmc_dconv3.json).  Stride 1: rows padded
    // to 1536 bytes.
    // Stride 2 (round 5): the same holds for the sub-patch rows -- 17 x 80 = 1360 bytes ran the fragment reads two-way conflicted in one of every
    // two lane groups (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.41, profiles/r04_c_pmc_dconv2_1_p16.json): rows padded to 1536 bytes too.
    constexpr int PITCH = 80, RPB = HALO_WIDE_ROW_BYTES;
    constexpr int SUBB = (TH + 1) * RPB;                                                               // S = 2: bytes of one parity sub-patch
    constexpr int PLANE = S == 2 ? 4 * SUBB : PH * RPB, STAGE = NP * PLANE, NSTAGE = S == 2 ? 1 : 2;
    constexpr int RPP = NTHR / 8;                              // patch pixels staged per pass (8 lanes = 8 channel quads per pixel)
    constexpr int NL = (NPIX + RPP - 1) / RPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2 stages][2 planes][180][80]

    const ConvGeom& g = a.g;
    const float ascale = a.amax != nullptr ? f16_scale_from_amax(a.amax) : F16_ASCALE;
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % a.nblk_n, tile = bid / a.nblk_n;
    // tiles cover the output map; where it is not a multiple of 8 x 16 (stride 1 only: the 56 / 28 / 14-pixel maps of the ResNet branch, the
    // 80 x 104 maps of the reference's native 80 x 208 clips) the tile pixels outside are computed on zero-padded input and never stored
    const int tiles_x = (g.OW + HT_W - 1) / HT_W, tiles_y = (g.OH + TH - 1) / TH;
    const int tx = tile % tiles_x; const int r_ = tile / tiles_x; const int ty = r_ % tiles_y, n = r_ / tiles_y;
    const int py = ty * TH * S + hy0, px = tx * HT_W * S + hx0;
    const int Cin = a.C1 + a.C2, k16 = Cin / 16, nch = Cin / 32, nch1 = a.C1 / 32;        // chunks [0, nch1) read a.in, the rest a.in2 (virtual concat)
    constexpr int OOB = 0x7fffffff;
    const int NT = (a.Cout + 31) / 32;
    const int frag_plane = NT * g.wtaps * k16 * 1024;
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, NP * frag_plane, 0x00020000);

    // patch staging: thread -> channel quad q of patch pixels h0 + RPP * j (pixel validity does not depend on the chunk)
    const int h0 = tid >> 3, q = tid & 7;
    int poff[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int h = h0 + RPP * j;
        const int hr = h / PW, hc = h - hr * PW;
        const int iy = py + hr, ix = px + hc;
        const int dead = (((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | (NPIX - 1 - h)) >> 31) & OOB;
        poff[j] = ((n * g.IH + iy) * g.IW + ix) | dead;                          // pixel index (or out of range)
    }
    u32x4 raw[NL];
    auto gloadA = [&](int chunk_) {
        const int chunk = __builtin_amdgcn_readfirstlane(chunk_);
        const int dead = chunk < nch ? 0 : OOB;
        const bool first = chunk < nch1;
        const int cs = first ? a.C1 : a.C2;                                      // channel stride of the source this chunk reads
        const int c0 = first ? chunk : chunk - nch1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? a.in : a.in2), 0, (int)(in_pixels * cs * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int pd = poff[j] >> 31 | ((poff[j] == OOB) ? -1 : 0);           // all ones for an out-of-range pixel
            raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((unsigned)poff[j] * (unsigned)cs + (unsigned)(q * 4)) * 4u) | (pd & OOB) | dead, dead ? 0 : c0 * 128, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int h = h0 + RPP * j;
            if (h < NPIX) {
                const int hr = h / PW, hc = h - hr * PW;
                const int soff = S == 2 ? ((hr & 1) * 2 + (hc & 1)) * SUBB + (hr >> 1) * RPB + (hc >> 1) * PITCH : hr * RPB + hc * PITCH;
                stage_put32<P16>(smem_h + buf * STAGE + soff, PLANE, q, raw[j], ascale, alim);
            }
        }
    };
    int bbase[TN], sl[9];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nt = bn * (BN / 32) + wn * TN + j;
        bbase[j] = (nt < NT) ? nt * g.wtaps * k16 * 1024 + lane * 16 : OOB;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) sl[t] = __builtin_amdgcn_readfirstlane(slots.s[t]);
    // weight fragments of (position t, chunk cc): both k-steps
    auto gloadB = [&](u32x4 (&b0)[TN][NP], u32x4 (&b1)[TN][NP], int t, int cc_) {
        const int cc = __builtin_amdgcn_readfirstlane(cc_);
        const int dead = cc < nch ? 0 : OOB;
        const int soff = dead ? 0 : (sl[t] * k16 + cc * 2) * 1024;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                b0[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bbase[j] | dead, soff + p * frag_plane, 0);
                b1[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bbase[j] | dead, soff + 1024 + p * frag_plane, 0);
            }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // MFMA row r = lane & 31 of M-tile i -> tile pixel (2 (wm TM + i) + (r >> 4), r & 15); the patch origin is the window's top-left tap
    const int aoff = (wm * TM * 2 + ((lane & 31) >> 4)) * RPB + (lane & 15) * PITCH + 16 * (lane >> 5);
    // fragment sets (k-step 0 / 1 each), rotating per tap.  Stride 1: THREE sets, the fragments of tap t + 2 are fetched while tap t is
    // multiplied (the weight stream was the largest exposed wait of this kernel: one tap = 12 .. 24 MFMAs does not cover an L2 round trip);
    // nine taps = 3 x 3, so the set of a tap is the same in every chunk.  Stride 2 (226 registers already): two sets, one tap ahead.
    constexpr int NSET = (S == 1 && VIAI_HALO_WIDE_NSET3) ? 3 : 2;
    u32x4 B0[NSET][TN][NP], B1[NSET][TN][NP];
    gloadA(0);
    gloadB(B0[0], B1[0], 0, 0);
    if constexpr (NSET == 3) gloadB(B0[1], B1[1], 1, 0);
    lstore(0);
    gloadA(1);
    __syncthreads();

    auto loadA = [&](const unsigned char* As, u32x4 (&af)[TM][NP]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) af[i][p] = *reinterpret_cast<const u32x4*>(As + p * PLANE + i * 2 * RPB);
    };
    auto mfmas = [&](const u32x4 (&af)[TM][NP], const u32x4 (&b)[TN][NP]) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};        // smallest partial products first
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][PA[pr]]), __builtin_bit_cast(f16x8, b[j][PB[pr]]), acc[i][j], 0, 0, 0);
    };
    auto tap_off = [&](int t) -> int {
        return S == 2 ? (((t / 3) & 1) * 2 + ((t % 3) & 1)) * SUBB + ((t / 3) >> 1) * RPB + ((t % 3) >> 1) * PITCH : (t / 3) * RPB + (t % 3) * PITCH;
    };
    // one chunk: nine taps x two k-steps = 18 groups of 3 TM TN MFMAs.  The operands of a group are requested one group (A fragments,
    // LDS) or NSET - 1 taps (weight fragments, global) ahead, and a scheduling barrier on either side keeps each group a solid block of
    // MFMAs: left to itself the compiler threads the reads, loads and a partial s_waitcnt per operand through the MFMA stream, and
    // every issue slot between two MFMAs costs matrix-pipe time (MI355X_MICROARCH.md: one extra slot between MFMAs = 6 .. 43 cycles).
    // P = fragment set of tap 0 when there are two sets (nine is odd, so the parity flips from chunk to chunk)
    auto chunk = [&](int cc, auto P) {
        constexpr int p0 = decltype(P)::value;
        const unsigned char* Sb = smem_h + (NSTAGE == 2 ? (cc & 1) * STAGE : 0) + aoff;
        u32x4 afq[2][TM][NP];
        loadA(Sb + tap_off(0), afq[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int cur = NSET == 3 ? t % 3 : (p0 + t) & 1;
            constexpr int AHEAD = NSET - 1;
            const int ta = t + AHEAD, nxt = NSET == 3 ? ta % 3 : cur ^ 1;
            if (ta < 9) gloadB(B0[nxt], B1[nxt], ta, cc);
            else gloadB(B0[nxt], B1[nxt], ta - 9, cc + 1);
            loadA(Sb + tap_off(t) + 32, afq[1]);                               // k-step 1 of this tap
            if (VIAI_HALO_WIDE_FENCED) __builtin_amdgcn_sched_barrier(0);
            mfmas(afq[0], B0[cur]);
            if (VIAI_HALO_WIDE_FENCED) __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) loadA(Sb + tap_off(t + 1), afq[0]);                 // k-step 0 of the next tap
            if (VIAI_HALO_WIDE_FENCED) __builtin_amdgcn_sched_barrier(0);
            mfmas(afq[1], B1[cur]);
            if (VIAI_HALO_WIDE_FENCED) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NSTAGE == 1) __syncthreads();            // one stage: every wave is done reading chunk cc
        lstore(NSTAGE == 2 ? (cc & 1) ^ 1 : 0);                // chunk cc + 1 (loaded during this chunk) -> the other stage
        gloadA(cc + 2);
        __syncthreads();
    };
    for (int cc = 0; cc + 1 < nch; cc += 2) {
        chunk(cc, std::integral_constant<int, 0>{});
        chunk(cc + 1, std::integral_constant<int, 1>{});
    }
    if (nch & 1) chunk(nch - 1, std::integral_constant<int, 0>{});

    // ---------------------------------------------------------------- epilogue
    const float inv = 1.0f / (ascale * F16_WSCALE);
    const int half = lane >> 5, col = lane & 31;
    int co[TN]; float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        co[j] = bn * BN + (wn * TN + j) * 32 + col;
        bv[j] = (a.bias != nullptr && co[j] < a.Cout) ? a.bias[co[j]] : 0.f;
    }
    const int oy0 = ty * TH, ox0 = tx * HT_W;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int oy = oy0 + 2 * (wm * TM + i) + (row >> 4), ox = ox0 + (row & 15);
            const size_t opix = ((size_t)n * g.OH + oy) * g.OW + ox;
            const bool inside = oy < g.OH && ox < g.OW;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[i][j][e] * inv + bv[j];
                if (a.stat == nullptr) v = viai_act(v, a.act, a.slope);
                acc[i][j][e] = inside ? v : 0.f;                  // (zeros outside: the statistics below skip them by count)
                if (co[j] < a.Cout && inside) {
                    if (co[j] < a.OC1) a.out[opix * a.OC1 + co[j]] = v;
                    else a.out2[opix * (a.Cout - a.OC1) + (co[j] - a.OC1)] = v;
                }
            }
        }
    if (a.stat != nullptr) {          // (mean, M2) of the tile's VALID pixels per channel: per-wave two-pass over its rows, Chan merge of the WM waves
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_h);          // [WM][2][BN]
        // valid pixels of this wave's rows: (tile rows 2 wm TM .. 2 (wm + 1) TM - 1 that lie inside the map) x (tile columns inside)
        const int vc = min(HT_W, g.OW - ox0), vr_all = min(TH, g.OH - oy0);
        auto rows_of = [&](int w) { const int lo = 2 * w * TM; return max(0, min(vr_all - lo, 2 * TM)); };
        const float nw = (float)(rows_of(wm) * vc);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];           // pixels outside were set to zero above
            t += __shfl_xor(t, 32, 64);
            const float mw = nw > 0.f ? t / nw : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                    const bool inside = oy0 + 2 * (wm * TM + i) + (row >> 4) < g.OH && ox0 + (row & 15) < g.OW;
                    const float d = acc[i][j][e] - mw;
                    m2 += inside ? d * d : 0.f;
                }
            m2 += __shfl_xor(m2, 32, 64);
            if (half == 0) {
                red[(wm * 2 + 0) * BN + (wn * TN + j) * 32 + col] = mw;
                red[(wm * 2 + 1) * BN + (wn * TN + j) * 32 + col] = m2;
            }
        }
        __syncthreads();
        if (wm == 0 && half == 0)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (co[j] < a.Cout) {
                    const int c = (wn * TN + j) * 32 + col;
                    const float ntot = (float)(vr_all * vc);
                    float mean = 0.f, M2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) { mean += (float)(rows_of(w) * vc) * red[(w * 2) * BN + c]; M2 += red[(w * 2 + 1) * BN + c]; }
                    mean /= ntot;
#pragma unroll
                    for (int w = 0; w < WM; ++w) { const float d = red[(w * 2) * BN + c] - mean; M2 += (float)(rows_of(w) * vc) * d * d; }
                    a.stat[(size_t)co[j] * a.nblk_m + tile] = mean;
                    a.stat[(size_t)(a.Cout + co[j]) * a.nblk_m + tile] = M2;
                }
    }
}

template <int CIN, int TN, int NP = 3, int MY = 1>
int launch_halo(ConvArgs& a, int hy0, int hx0, hipStream_t st) {
    constexpr int PITCH = CIN * 2 + 16;
    constexpr int HT_HP = ((HT_H - 1) * MY + 3) * HT_HW;
    size_t lds = (size_t)NP * HT_HP * PITCH;
    if (lds < (size_t)4 * 32 * TN * sizeof(float)) lds = (size_t)4 * 32 * TN * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_bf3_kernel<CIN, TN, NP, false, MY>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_done = true;
    }
    a.nblk_m = a.M / 128;
    a.nblk_n = 1;
    int per_cu = (int)((160 * 1024) / lds);                  // resident blocks per CU (LDS-limited), persistent over the tiles
    if (per_cu > 4) per_cu = 4;
    int grid = 256 * per_cu;
    if (grid > a.nblk_m) grid = a.nblk_m;
    viai_tag_kernel(NP == 2 ? "halo_f16x2" : "halo_bf16x3");
    if constexpr (NP == 2) {
        if (a.in_p16) {
            static bool attr_p = false;
            if (!attr_p) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_bf3_kernel<CIN, TN, NP, true, MY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_p = true; }
            VIAI_LAUNCH((conv_halo_bf3_kernel<CIN, TN, NP, true, MY>), dim3(grid), dim3(256), lds, st, a, hy0, hx0, a.nblk_m);
            return viai_launch_status();
        }
    }
    if (a.in_p16) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH((conv_halo_bf3_kernel<CIN, TN, NP, false, MY>), dim3(grid), dim3(256), lds, st, a, hy0, hx0, a.nblk_m);
    return viai_launch_status();
}

}  // namespace

// Shapes the halo kernel takes: one source of 32 or 64 channels, <= 64 output channels, unit stride in both the
// gather and the scatter, taps within a 3 x 3 window, output extent a multiple of the 8 x 16 tile.
// Round 6: also the stride-(2, 1) layer of the encoder -- forward (gather stride my = 2, 32 -> 64 channels) and the two row-parity classes of its data gradient
// (scatter stride ly = 2 on a sub-lattice of SH x SW pixels).
bool viai_conv_halo_ok(const ConvGeom& g, int C1, int C2, int Cout) {
    if (C2 != 0 || (C1 != 32 && C1 != 64) || Cout > 64 || Cout < 1) return false;
    if (g.run || g.lx != 1 || g.mx != 1 || g.ax != 0 || g.SW != g.OW) return false;
    const bool plain = g.ly == 1 && g.my == 1 && g.ay == 0 && g.SH == g.OH;
    const bool fwd_s21 = g.ly == 1 && g.my == 2 && g.ay == 0 && g.SH == g.OH && C1 == 32;
    const bool dgrad_s21 = g.ly == 2 && g.my == 1 && (g.ay == 0 || g.ay == 1) && C1 == 64 && Cout == 32;
    if (!plain && !fwd_s21 && !dgrad_s21) return false;
    if (g.SH % HT_H != 0 || g.SW % HT_W != 0 || g.ntaps < 1) return false;
    int y0 = g.dy[0], y1 = g.dy[0], x0 = g.dx[0], x1 = g.dx[0];
    for (int t = 1; t < g.ntaps; ++t) {
        y0 = g.dy[t] < y0 ? g.dy[t] : y0; y1 = g.dy[t] > y1 ? g.dy[t] : y1;
        x0 = g.dx[t] < x0 ? g.dx[t] : x0; x1 = g.dx[t] > x1 ? g.dx[t] : x1;
    }
    return (y1 - y0) <= 2 && (x1 - x0) <= 2;
}

// f16x2 register-resident-filter variant: 32 input channels, <= 32 output channels, all nine positions of a 3 x 3 window
bool viai_conv_halo16_ok(const ConvGeom& g, int C1, int C2, int Cout) {
    constexpr int on = 1;
    if (!on || !viai_conv_halo_ok(g, C1, C2, Cout) || C1 != 32 || Cout > 32 || g.ntaps != 9 || g.my != 1 || g.ly != 1) return false;
    int y0 = g.dy[0], x0 = g.dx[0];
    for (int t = 1; t < 9; ++t) { y0 = g.dy[t] < y0 ? g.dy[t] : y0; x0 = g.dx[t] < x0 ? g.dx[t] : x0; }
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) seen |= 1u << ((g.dy[t] - y0) * 3 + (g.dx[t] - x0));
    return seen == 0x1ffu;
}

int viai_conv_halo_bf3_launch(ConvArgs& a, hipStream_t st) {
    const ConvGeom& g = a.g;
    if (!viai_conv_halo_ok(g, a.C1, a.C2, a.Cout)) return (int)hipErrorInvalidValue;
    if (a.OC1 % 32 != 0 && a.OC1 != a.Cout) return (int)hipErrorInvalidValue;
    if (a.wfrag == 3 && !viai_conv_halo16_ok(g, a.C1, a.C2, a.Cout)) {      // f16x2, filter streamed from L2 (64-channel / wide layers)
        int y0 = g.dy[0], x0 = g.dx[0];
        for (int t = 1; t < g.ntaps; ++t) { y0 = g.dy[t] < y0 ? g.dy[t] : y0; x0 = g.dx[t] < x0 ? g.dx[t] : x0; }
        const bool wide = a.Cout > 32;
        if (g.my == 2) return wide ? launch_halo<32, 2, 2, 2>(a, y0, x0, st) : launch_halo<32, 1, 2, 2>(a, y0, x0, st);
        if (a.C1 == 32) return wide ? launch_halo<32, 2, 2>(a, y0, x0, st) : launch_halo<32, 1, 2>(a, y0, x0, st);
        return wide ? launch_halo<64, 2, 2>(a, y0, x0, st) : launch_halo<64, 1, 2>(a, y0, x0, st);
    }
    if (a.wfrag == 3) {                                        // f16x2 fragment-major weights, filter in registers
        int y0 = g.dy[0], x0 = g.dx[0];
        for (int t = 1; t < 9; ++t) { y0 = g.dy[t] < y0 ? g.dy[t] : y0; x0 = g.dx[t] < x0 ? g.dx[t] : x0; }
        HaloSlots sl;
        for (int t = 0; t < 9; ++t) sl.s[(g.dy[t] - y0) * 3 + (g.dx[t] - x0)] = g.ws[t];
        const size_t lds = (size_t)2 * HT_HP * 80;
        a.nblk_m = a.M / 128;
        a.nblk_n = 1;
        int grid = 256 * 2;
        if (grid > a.nblk_m) grid = a.nblk_m;
        viai_tag_kernel("halo_c32_f16x2");
        if (viai_conv_halo_c32_dma_ok(a)) return viai_conv_halo_c32_dma_launch(a, y0, x0, sl.s, st);      // round 5: patch by LDS-DMA, three tiles deep
        if (a.in_p16) VIAI_LAUNCH(conv_halo_f16_c32_kernel<true>, dim3(grid), dim3(256), lds, st, a, y0, x0, a.nblk_m, sl);
        else VIAI_LAUNCH(conv_halo_f16_c32_kernel<false>, dim3(grid), dim3(256), lds, st, a, y0, x0, a.nblk_m, sl);
        return viai_launch_status();
    }
    int y0 = g.dy[0], y1 = g.dy[0], x0 = g.dx[0], x1 = g.dx[0];
    for (int t = 1; t < g.ntaps; ++t) {
        y0 = g.dy[t] < y0 ? g.dy[t] : y0; y1 = g.dy[t] > y1 ? g.dy[t] : y1;
        x0 = g.dx[t] < x0 ? g.dx[t] : x0; x1 = g.dx[t] > x1 ? g.dx[t] : x1;
    }
    const bool wide = a.Cout > 32;
    if (g.my == 2) return wide ? launch_halo<32, 2, 3, 2>(a, y0, x0, st) : launch_halo<32, 1, 3, 2>(a, y0, x0, st);
    if (a.C1 == 32) return wide ? launch_halo<32, 2>(a, y0, x0, st) : launch_halo<32, 1>(a, y0, x0, st);
    return wide ? launch_halo<64, 2>(a, y0, x0, st) : launch_halo<64, 1>(a, y0, x0, st);
}

// Stride-1 3 x 3 layers for the kernel above: one source with Cin a multiple of 32, one destination with 32, 64 or a multiple of 128
// channels, the full window, output extent a multiple of the 8 x 16 tile, and enough tiles to occupy the chip (smaller layers stay
// on the split-K / 64 x 64 kernels).  The fragment-major f16x2 weight image is a consequence of this predicate: conv_api.hip asks it
// when it packs, viai_conv_igemm_bf3_launch when it launches.
int viai_halo_tiles_y(const ConvGeom& g) { return (g.OH + HT_H - 1) / HT_H; }
// tile rows of the stride-2 forward instance: 4 (64-pixel tiles, two blocks per CU) when the map is a whole number of them and large enough
// to fill the chip that way, else 8
int viai_halo_s2_rows(const ConvGeom& g) {
    if (g.my != 2 || g.OH % 4 != 0) return 8;
    // (round 5) the loader / consumer kernel (conv_halo_dma.hip) writes one partial block per 4 x 16 pixels whatever the map size; the block geometry is a
    // property of the LAYER (viai_conv2d_stat_geom does not know which kernel will run), so the register-staged kernel follows on every map that kernel takes
    if (viai_halo_dma_on() && g.OH % 8 == 0 && g.OW % 16 == 0 && g.IH == 2 * g.OH && g.IW == 2 * g.OW) return 4;
    return (long)g.N * (g.OH / 4) * viai_halo_tiles_x(g) >= 512 ? 4 : 8;
}
int viai_halo_tiles_x(const ConvGeom& g) { return (g.OW + HT_W - 1) / HT_W; }

bool viai_conv_halo_wide_ok(const ConvArgs& a) {
    const ConvGeom& g = a.g;
    if (a.C1 % 32 != 0 || a.C2 % 32 != 0 || a.C1 < 32 || (a.OC1 != a.Cout && a.OC1 % 32 != 0)) return false;
    if (!(a.Cout == 32 || a.Cout == 64 || a.Cout % 128 == 0)) return false;
    const bool whole = g.OH % HT_H == 0 && g.OW % HT_W == 0;
    if (whole && a.Cout <= 64 && a.C1 + a.C2 <= 64 && a.C2 == 0) return false;     // the small-channel halo kernels take these (whole tiles only)
    if (g.run || g.ly != 1 || g.lx != 1 || g.my != g.mx || (g.my != 1 && g.my != 2) || g.SH != g.OH || g.SW != g.OW || g.ntaps != 9) return false;
    if (!whole) {
        // partial tiles (stride 1): worth it while the tiles are >= 2/3 full -- 56 x 56: 87 %, 28 x 28: 77 %, 14 x 14: 77 %, 20 x 26 (D.conv3 on the
        // reference's native 80 x 208 clips): 68 %, 7 x 7: 38 % (stays on the gather kernel)
        if (g.my != 1 || (long)g.OH * g.OW * 3 < (long)viai_halo_tiles_y(g) * viai_halo_tiles_x(g) * HT_H * HT_W * 2) return false;
    }
    if (g.my == 2) {                                               // stride-2 forward: eight-wave instances only, one source
        constexpr int s2 = 1;
        if (!s2 || a.C2 != 0 || a.Cout % 128 != 0 || a.OC1 != a.Cout) return false;
    }
    const long tiles = (long)g.N * viai_halo_tiles_y(g) * viai_halo_tiles_x(g);
    // small maps: 64-channel blocks double the block count (the 16 x 32 maps of G.convblock2: 64 tiles -> 128 / 256 blocks, each
    // with half the K-loop work of a 128-channel block) -- still better than the split-K kernel those layers ran on
    constexpr long min64 = 96;
    if (tiles * (a.Cout >= 128 ? a.Cout / 128 : 1) < 192 && !(a.Cout >= 128 && tiles * (a.Cout / 64) >= min64)) return false;
    int y0 = g.dy[0], x0 = g.dx[0];
    for (int t = 1; t < 9; ++t) { y0 = g.dy[t] < y0 ? g.dy[t] : y0; x0 = g.dx[t] < x0 ? g.dx[t] : x0; }
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int r = g.dy[t] - y0, c = g.dx[t] - x0;
        if (r > 2 || c > 2) return false;
        seen |= 1u << (r * 3 + c);
    }
    return seen == 0x1ffu;
}

template <int WM, int WN, int TM, int TN, int S = 1>
static int launch_halo_wide(ConvArgs& a, int y0, int x0, const HaloWideSlots& sl, hipStream_t st) {
    constexpr int TH = 2 * WM * TM;
    constexpr int lds = S == 2 ? 2 * 4 * (TH + 1) * HALO_WIDE_ROW_BYTES : 2 * 2 * HT_HH * HALO_WIDE_ROW_BYTES;
    a.nblk_m = a.g.N * ((a.g.OH + TH - 1) / TH) * viai_halo_tiles_x(a.g);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_wide_f16_kernel<WM, WN, TM, TN, S, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_wide_f16_kernel<WM, WN, TM, TN, S, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    a.nblk_n = (a.Cout + 32 * TN * WN - 1) / (32 * TN * WN);
    static const std::string fam = S == 2 ? std::string("halo_wide_s2_f16x2") : "halo_wide" + std::to_string(32 * TN * WN) + "_f16x2";
    viai_tag_kernel(fam.c_str());
    if (a.in_p16 && (a.C2 != 0 || a.amax == nullptr)) return (int)hipErrorInvalidValue;          // one pre-split source, with its scale
    if (a.in_p16) VIAI_LAUNCH((conv_halo_wide_f16_kernel<WM, WN, TM, TN, S, true>), dim3(a.nblk_m * a.nblk_n), dim3(64 * WM * WN), lds, st, a, y0, x0, sl);
    else VIAI_LAUNCH((conv_halo_wide_f16_kernel<WM, WN, TM, TN, S, false>), dim3(a.nblk_m * a.nblk_n), dim3(64 * WM * WN), lds, st, a, y0, x0, sl);
    return viai_launch_status();
}

int viai_conv_halo_wide_launch(ConvArgs& a, hipStream_t st) {
    const ConvGeom& g = a.g;
    if (!viai_conv_halo_wide_ok(a)) return (int)hipErrorInvalidValue;
    int y0 = g.dy[0], x0 = g.dx[0];
    for (int t = 1; t < 9; ++t) { y0 = g.dy[t] < y0 ? g.dy[t] : y0; x0 = g.dx[t] < x0 ? g.dx[t] : x0; }
    HaloWideSlots sl;
    for (int t = 0; t < 9; ++t) sl.s[(g.dy[t] - y0) * 3 + (g.dx[t] - x0)] = g.ws[t];
    a.nblk_m = g.N * viai_halo_tiles_y(g) * viai_halo_tiles_x(g);
    // (stride 2 keeps the 64-pixel wave tiles: <1,8,4,1,2> 70.7 vs 72.1 us on D.conv2_2 but with spills, <1,4,4,1,2> 269 vs 101 us on D.conv2_1)
    // round 4: 64-pixel tiles (4 x 16) for the stride-2 forward where the map allows: half the LDS (49 KB), four waves per block, so TWO blocks
    // share a CU and one block's patch load / epilogue runs under the other's MFMAs -- the 128-pixel block is alone on its CU (98 KB, one LDS
    // stage) and its K loop is only 2 .. 4 chunks long, so nothing covered its prologue and epilogue
    if (g.my == 2 && viai_conv_s2_dma_ok(a)) { viai_tag_kernel("halo_wide_s2_f16x2"); return viai_conv_s2_dma_launch(a, st); }
    if (g.my == 1 && viai_conv_s1_dma_ok(a)) { viai_tag_kernel("halo_wide256_f16x2"); return viai_conv_s1_dma_launch(a, st); }
    if (g.my == 2 && viai_halo_s2_rows(g) == 4) return (a.Cout % 256 == 0) ? launch_halo_wide<1, 4, 2, 2, 2>(a, y0, x0, sl, st) : launch_halo_wide<1, 4, 2, 1, 2>(a, y0, x0, sl, st);
    if (g.my == 2) return (a.Cout % 256 == 0) ? launch_halo_wide<2, 4, 2, 2, 2>(a, y0, x0, sl, st) : launch_halo_wide<2, 4, 2, 1, 2>(a, y0, x0, sl, st);
    if (a.Cout == 32) return launch_halo_wide<4, 1, 1, 1>(a, y0, x0, sl, st);
    if (a.Cout == 64 || (long)a.nblk_m * (a.Cout / 128) < 192) return launch_halo_wide<2, 2, 2, 1>(a, y0, x0, sl, st);
    constexpr int wn4 = 1;
    // 256-channel blocks: eight waves, each ALL 128 pixels x 32 channels (<1,8,4,1>) rather than 64 x 64 (<2,4,2,2>): a weight fragment
    // then feeds four M tiles, so the fragment stream through L1 halves (2 KB per 12 MFMAs) while the patch reads from LDS double (8 KB) --
    // LDS has twice L1's bandwidth, and the registers drop 231 -> 215.  D.conv3 203.5 -> 198 us, step 7.455 -> 7.423 ms (same box A/B).
    constexpr int tm4 = 1;
    if (tm4 && a.Cout % 256 == 0 && (long)a.nblk_m * (a.Cout / 256) >= 256) return launch_halo_wide<1, 8, 4, 1>(a, y0, x0, sl, st);
    if (wn4 && a.Cout % 256 == 0 && (long)a.nblk_m * (a.Cout / 256) >= 256) return launch_halo_wide<2, 4, 2, 2>(a, y0, x0, sl, st);
    return launch_halo_wide<1, 4, 4, 1>(a, y0, x0, sl, st);           // (128-channel blocks, same reasoning: 899 -> 855 us on 1024 x 28 x 28 x 128, 122.5 -> 117 us on 16 x 64 x 128 x 128)
}
