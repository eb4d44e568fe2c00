// Data gradient of the 3x3 / stride 2 / pad 1 convolutions (D.conv2_1, D.conv2_2, E.conv3-5) with the four output
// parity classes fused in one block, bf16x3 split MFMA, gfx950.
//
// dx(2by+a, 2bx+b) gathers dy at the "base" pixel (by, bx) shifted by (sy, sx) in {0,1}^2:
//     a = 0: (sy = 0, r = 1)            a = 1: (sy = 1, r = 0), (sy = 0, r = 2)        (same for b / sx / s)
// so nine (shift, class) pairs carry the nine taps.  Launched one class at a time (conv_igemm_bf3.hip) the short K
// loops (1, 2, 2, 4 taps) leave the prologue / epilogue and nine separate loads + bf16 splits of the dy tile
// dominant: 72-97 TFLOP/s.  Here a block owns 128 base pixels x 64 input channels x ALL FOUR classes (8 accumulator
// tiles per wave): each of the four shifted dy tiles is loaded and split once per K chunk and multiplied into every
// class that uses it, weights come fragment-major straight from global memory, and the epilogue writes 2 x 2 pixel
// quads.  Reference call sites: autograd of nn.Conv2d(.., 3, stride=2, padding=1) in Discriminator_Networks.py:23-31
// and Inpainting_Networks.py:58-63.
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"
#include <type_traits>
// timing ablation: bit 0 no weight-fragment loads, bit 1 no dy loads, bit 2 no split / LDS stores, bit 3 no output stores
#ifndef VIAI_ABL
#define VIAI_ABL 0
#endif

namespace {

// NP = 3: bf16x3 (six partial products); NP = 2: f16x2 (three, dy scaled by a power of two derived from a.amax)
// P16 (round 5, NP = 2 only): dy arrives pre-split -- a thread's 16-byte item of a pixel's 32-channel chunk is then a PIECE (piece q & 3 of plane q >> 2) at the
// same global address, stored as it is (the ResNet branch's stride-2 layers on 28 / 14 / 7-pixel maps, which the patch kernel below does not tile)
template <int NP, bool P16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_dgrad_s2_bf3_kernel(const ConvArgs a) {
    constexpr int BK = BF3_BK, BM = 128, TM = 2, NA = BM / 32;
    constexpr int APLANE = BM * BF3_PITCH, STAGE = NP * APLANE;
    constexpr int NPROD = NP == 3 ? 6 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // [2][3][128][80]

    const ConvGeom& g = a.g;                      // N, IH/IW = dy extent, OH/OW = dx extent
    const float ascale = (NP == 2) ? f16_scale_from_amax(a.amax) : 1.f;
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % a.nblk_n, bm = bid / a.nblk_n;
    const int m0 = bm * BM;
    const int K = a.C1;                           // dy channels (conv Cout)
    const int BH = g.OH / 2, BW = g.OW / 2;       // base lattice

    const int q = tid & 7, r0 = tid >> 3;
    int pbase[NA], by_[NA], bx_[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int m = m0 + r0 + 32 * j;
        if (m < a.M) {
            int bx = m % BW; int t = m / BW; int by = t % BH; int n = t / BH;
            by_[j] = by; bx_[j] = bx;
            pbase[j] = (n * g.IH + by) * g.IW + bx;
        } else { by_[j] = 1 << 20; bx_[j] = 1 << 20; pbase[j] = 0; }
    }
    constexpr int OOB = 0x7fffffff;
    const int k16 = K / 16;
    const int NT = (a.Cout + 31) / 32;
    const int frag_plane = NT * 9 * k16 * 1024;
    const int nt = bn * 2 + wn;                                      // this wave's 32-channel tile (of every class)
    const int bvoff = (nt < NT) ? nt * 9 * k16 * 1024 + lane * 16 : OOB;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((long)g.N * g.IH * g.IW * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, NP * frag_plane, 0x00020000);

    f32x16 acc[4][TM];                            // [class a*2+b][row tile]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][i][e] = 0.f;

    const int kchunks = K / BK;
    const int nchunks = 4 * kchunks;              // chunk = (shift, 32 channels)

    u32x4 raw[NA];
    auto gloadA = [&](int chunk_) {
        const int chunk = __builtin_amdgcn_readfirstlane(chunk_);
        const bool live = chunk < nchunks;
        const int sh = live ? chunk / kchunks : 0, c0 = (chunk - sh * kchunks) * BK;
        const int sy = sh >> 1, sx = sh & 1;
        const int toff = sy * g.IW + sx;
        const int dead_s = live ? 0 : OOB;
#pragma unroll
        for (int j = 0; j < NA; ++j) {               // offset masks instead of selects: no branches in the K loop (conv_igemm_bf3.hip)
            const int dead = (((g.IH - 1 - (by_[j] + sy)) | (g.IW - 1 - (bx_[j] + sx))) >> 31) & OOB;
#if VIAI_ABL & 2
            raw[j] = u32x4{(unsigned)(pbase[j] + toff + dead + dead_s), 0u, 0u, 0u};
#else
            raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (((pbase[j] + toff) * K + c0 + q * 4) * 4) | dead | dead_s, 0, 0);
#endif
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* As = smem_b + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if constexpr (P16) {
                *reinterpret_cast<u32x4*>(As + (q >> 2) * APLANE + (r0 + 32 * j) * BF3_PITCH + (q & 3) * 16) = raw[j];
                continue;
            }
            const f32x4 v = __builtin_bit_cast(f32x4, raw[j]);
            unsigned char* d = As + (r0 + 32 * j) * BF3_PITCH + q * 8;
            if constexpr (NP == 3) {
                unsigned a1, a2, a3, b1, b2, b3;
                split3_pair(v[0], v[1], a1, a2, a3);
                split3_pair(v[2], v[3], b1, b2, b3);
                const u32x2 p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3};
                *reinterpret_cast<u32x2*>(d) = p1;
                *reinterpret_cast<u32x2*>(d + APLANE) = p2;
                *reinterpret_cast<u32x2*>(d + 2 * APLANE) = p3;
            } else {
                unsigned a1, a2, b1, b2;
                split2_pair(v[0], v[1], ascale, alim, a1, a2);
                split2_pair(v[2], v[3], ascale, alim, b1, b2);
                const u32x2 p1 = {a1, b1}, p2 = {a2, b2};
                *reinterpret_cast<u32x2*>(d) = p1;
                *reinterpret_cast<u32x2*>(d + APLANE) = p2;
            }
        }
    };
    // weights of (tap, 16-deep k-step kq): three planes of this wave's channel tile
    auto gloadB = [&](u32x4 (&bf)[NP], int tap, int kq) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#if VIAI_ABL & 1
            bf[p] = u32x4{(unsigned)lane, (unsigned)(tap + kq), 0x3c003c00u, 0x3c003c00u};
#else
            bf[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bvoff, (tap * k16 + kq) * 1024 + p * frag_plane, 0);
#endif
    };

    gloadA(0);
    lstore(0);
    gloadA(1);
    __syncthreads();

    const int aoff = (wm * TM * 32 + (lane & 31)) * BF3_PITCH + 16 * (lane >> 5);
    constexpr int PA[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0}, PB[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};

    // one (class, k-step) unit: B fragments were fetched into `bf`; the next unit's are requested before the MFMAs
    auto mma = [&](f32x16 (&ac)[TM], const u32x4 (&af)[TM][NP], const u32x4 (&bf)[NP]) {
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (NP == 3) ac[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][PA[pr]]), __builtin_bit_cast(bf16x8, bf[PB[pr]]), ac[i], 0, 0, 0);
                else ac[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][PA[pr]]), __builtin_bit_cast(f16x8, bf[PB[pr]]), ac[i], 0, 0, 0);
            }
    };

    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = kc & 1;
        const int sh = __builtin_amdgcn_readfirstlane(kc / kchunks), c0q = (kc - sh * kchunks) * (BK / 16);
        const int sy = sh >> 1, sx = sh & 1;
        const unsigned char* As = smem_b + cur * STAGE + aoff;
        // classes served by this shift: a in {sy ? 1 : 0, 1}, b likewise; tap row r = (a == 0) ? 1 : (sy ? 0 : 2)
        const int na = sy ? 1 : 2, nb = sx ? 1 : 2;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            u32x4 af[TM][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[i][p] = *reinterpret_cast<const u32x4*>(As + p * APLANE + i * 32 * BF3_PITCH + ks * 32);
            u32x4 bfA[NP], bfB[NP];
            // unit order: (a, b) = (1,1), then (1,0) / (0,1), then (0,0) as far as the shift allows
            {
                const int r = sy ? 0 : 2, s = sx ? 0 : 2;
                gloadB(bfA, r * 3 + s, c0q + ks);                       // class (1,1): every shift
                if (nb == 2) gloadB(bfB, r * 3 + 1, c0q + ks);          // class (1,0): sx == 0
                mma(acc[3], af, bfA);
                if (na == 2) gloadB(bfA, 1 * 3 + s, c0q + ks);          // class (0,1): sy == 0
                if (nb == 2) mma(acc[2], af, bfB);
                if (na == 2 && nb == 2) gloadB(bfB, 1 * 3 + 1, c0q + ks);   // class (0,0): shift (0,0) only
                if (na == 2) mma(acc[1], af, bfA);
                if (na == 2 && nb == 2) mma(acc[0], af, bfB);
            }
        }
        // the other stage was last read in the previous iteration, which ended with a barrier: no barrier before the stores
#if !(VIAI_ABL & 4)
        lstore(cur ^ 1);
#endif
        gloadA(kc + 2);
        __syncthreads();
    }

    const float inv_bw = 1.0f / (float)BW, inv_bh = 1.0f / (float)BH;
    const float inv = (NP == 2) ? 1.0f / (ascale * F16_WSCALE) : 1.0f;      // applied at the store: scaling the 128 accumulators in place costs a second register set
    // ---------------------------------------------------------------- epilogue: four classes, 2 x 2 pixel quads
    const int half = lane >> 5, col = lane & 31;
    const int co = nt * 32 + col;
    const int oc2 = a.Cout - a.OC1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int m = m0 + (wm * TM + i) * 32 + row;
            if (m < a.M && co < a.Cout) {
                // m < 2^22 (viai_dgrad_s2_ok): float reciprocal + one correction step instead of four integer divisions per
                // row -- with K = 128 the divisions of this epilogue were 70 % of the kernel's VALU instructions
                int t = (int)((float)m * inv_bw); int bx = m - t * BW;
                if (bx < 0) { --t; bx += BW; } else if (bx >= BW) { ++t; bx -= BW; }
                int n = (int)((float)t * inv_bh); int by = t - n * BH;
                if (by < 0) { --n; by += BH; } else if (by >= BH) { ++n; by -= BH; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const size_t opix = ((size_t)n * g.OH + 2 * by + (c >> 1)) * g.OW + 2 * bx + (c & 1);
#if VIAI_ABL & 8
                    if (acc[c][i][e] == 1.2345f) a.out[opix] = 0.f;
#else
                    if (co < a.OC1) a.out[opix * a.OC1 + co] = acc[c][i][e] * inv;
                    else a.out2[opix * oc2 + (co - a.OC1)] = acc[c][i][e] * inv;
#endif
                }
            }
        }
}

// Patch-staged f16x2 variant (base lattice a multiple of 8 x 16): the kernel above loads and splits the dy tile once per SHIFT (four
// times per 32-channel chunk, one barrier each, a wave-uniform branch per class).  Here a block owns an 8 x 16 tile of the base
// lattice: per chunk the (8+1) x (16+1) dy patch is staged ONCE, the four shifts are four compile-time window offsets, and the nine
// (shift, class, tap) units of a chunk are a static list -- one barrier per chunk, no branches in the K loop.
template <bool P16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_dgrad_s2_patch_kernel(const ConvArgs a) {
    // RPB: bytes between patch rows.  17 x 80 = 1360 ran the fragment reads two-way conflicted in every other lane group (SQ_LDS_BANK_CONFLICT /
    // SQ_LDS_IDX_ACTIVE = 0.50, profiles/r04_c_pmc_dconv2_1_p16.json): a `ds_read_b128` group mixes columns of one tile row with columns of the next, so the
    // pitch must be a multiple of the 256-byte bank row (conv_halo_bf3.hip has the derivation)
    constexpr int NP = 2, TM = 2, PITCH = 80, PH = 9, PW = 17, HP = PH * PW, RPB = 1536, PLANE = PH * RPB, STAGE = NP * PLANE;
    constexpr int NL = (HP * 8 + 255) / 256;         // float4 per thread per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // [2 stages][2 planes][153][80]
    const ConvGeom& g = a.g;                         // N, IH/IW = dy extent = base lattice, OH/OW = dx extent
    const float ascale = f16_scale_from_amax(a.amax);
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % a.nblk_n, tile = bid / a.nblk_n;
    const int tiles_x = g.IW / 16, tiles_y = g.IH / 8;
    const int tx = tile % tiles_x; const int r_ = tile / tiles_x; const int ty = r_ % tiles_y, n = r_ / tiles_y;
    const int by0 = ty * 8, bx0 = tx * 16;
    const int K = a.C1, k16 = K / 16, nch = K / 32;
    constexpr int OOB = 0x7fffffff;
    const int NT = (a.Cout + 31) / 32;
    const int frag_plane = NT * 9 * k16 * 1024;
    const int nt = bn * 2 + wn;
    const int bvoff = (nt < NT) ? nt * 9 * k16 * 1024 + lane * 16 : OOB;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((long)g.N * g.IH * g.IW * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, NP * frag_plane, 0x00020000);

    const int h0 = tid >> 3, q = tid & 7;
    int poff[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int h = h0 + 32 * j;
        const int hr = h / PW, hc = h - hr * PW;
        const int iy = by0 + hr, ix = bx0 + hc;
        const int dead = (((g.IH - 1 - iy) | (g.IW - 1 - ix) | (HP - 1 - h)) >> 31) & OOB;
        poff[j] = ((((n * g.IH + iy) * g.IW + ix) * K + q * 4) * 4) | dead;
    }
    const int hc0 = h0 % PW, ls0 = (h0 / PW) * RPB + hc0 * PITCH;           // LDS position of this thread's first patch pixel
    u32x4 raw[NL];
    auto gloadA = [&](int chunk_) {
        const int chunk = __builtin_amdgcn_readfirstlane(chunk_);
        const int dead = chunk < nch ? 0 : OOB;
#pragma unroll
        for (int j = 0; j < NL; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, poff[j] | dead, dead ? 0 : chunk * 128, 0);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int h = h0 + 32 * j;
            // pixel h = h0 + 32 j = (row, column) of the 17-wide patch: 32 = one row + 15 columns, so with w = floor(15 j / 17) column wraps for column 0
            // the pixel has wrapped once more iff hc0 >= 17 (w + 1) - 15 j: a compare per item instead of a division by 17
            const int wj = (15 * j) / 17, thr = 17 * (wj + 1) - 15 * j;
            const int lo = ls0 + (j + wj) * RPB + (15 * j - 17 * wj) * PITCH + (hc0 >= thr ? RPB - PW * PITCH : 0);
            if (h < HP) stage_put32<P16>(smem_b + buf * STAGE + lo, PLANE, q, raw[j], ascale, alim);
        }
    };
    // the nine units of a chunk: (shift sy, sx) -> window offset; class (a, b) -> accumulator a * 2 + b; tap r * 3 + s
    //   shift (0,0): (1,1) t8, (1,0) t7, (0,1) t5, (0,0) t4;  shift (0,1): (1,1) t6, (0,1) t3;  shift (1,0): (1,1) t2, (1,0) t1;  shift (1,1): (1,1) t0
    constexpr int U_SH[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3}, U_CL[9] = {3, 2, 1, 0, 3, 1, 3, 2, 3}, U_TAP[9] = {8, 7, 5, 4, 6, 3, 2, 1, 0};
    auto gloadB = [&](u32x4 (&b0)[NP], u32x4 (&b1)[NP], int u, int cc_) {
        const int cc = __builtin_amdgcn_readfirstlane(cc_);
        const int dead = cc < nch ? 0 : OOB;
        const int soff = dead ? 0 : (U_TAP[u] * k16 + cc * 2) * 1024;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            b0[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bvoff | dead, soff + p * frag_plane, 0);
            b1[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bvoff | dead, soff + 1024 + p * frag_plane, 0);
        }
    };
    f32x16 acc[4][TM];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][i][e] = 0.f;

    // MFMA row r = lane & 31 of M-tile i -> base pixel (2 (wm * TM + i) + (r >> 4), r & 15) of the tile
    const int aoff = (wm * TM * 2 + ((lane & 31) >> 4)) * RPB + (lane & 15) * PITCH + 16 * (lane >> 5);
    // weight fragments two units ahead (three register sets; nine units = 3 x 3, so the set of a unit is the same in every chunk), A
    // fragments one group ahead, and every group of six MFMAs fenced by scheduling barriers so that nothing is issued between two MFMAs
    // on the same accumulator (conv_halo_bf3.hip has the measurements: D.conv3 212 -> 205 us, step 7.90 -> 7.75 ms from the same change)
    u32x4 B0[3][NP], B1[3][NP];
    gloadA(0);
    gloadB(B0[0], B1[0], 0, 0);
    gloadB(B0[1], B1[1], 1, 0);
    lstore(0);
    gloadA(1);
    __syncthreads();
    auto loadA = [&](const unsigned char* As, u32x4 (&af)[TM][NP]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) af[i][p] = *reinterpret_cast<const u32x4*>(As + p * PLANE + i * 2 * RPB);
    };
    auto mfmas = [&](f32x16 (&ac)[TM], const u32x4 (&af)[TM][NP], const u32x4 (&b)[NP]) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                ac[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][PA[pr]]), __builtin_bit_cast(f16x8, b[PB[pr]]), ac[i], 0, 0, 0);
    };
    auto chunk = [&](int cc, auto P) {
        const unsigned char* S = smem_b + (cc & 1) * STAGE + aoff;
        u32x4 afq[2][TM][NP];
        loadA(S + (U_SH[0] >> 1) * RPB + (U_SH[0] & 1) * PITCH, afq[0]);
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int cur = u % 3, ua = u + 2, nxt = ua % 3;
            if (ua < 9) gloadB(B0[nxt], B1[nxt], ua, cc);
            else gloadB(B0[nxt], B1[nxt], ua - 9, cc + 1);
            const unsigned char* As = S + (U_SH[u] >> 1) * RPB + (U_SH[u] & 1) * PITCH;
            loadA(As + 32, afq[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(acc[U_CL[u]], afq[0], B0[cur]);
            __builtin_amdgcn_sched_barrier(0);
            if (u + 1 < 9) loadA(S + (U_SH[u + 1] >> 1) * RPB + (U_SH[u + 1] & 1) * PITCH, afq[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(acc[U_CL[u]], afq[1], B1[cur]);
            __builtin_amdgcn_sched_barrier(0);
        }
        lstore((cc & 1) ^ 1);
        gloadA(cc + 2);
        __syncthreads();
    };
    for (int cc = 0; cc + 1 < nch; cc += 2) {
        chunk(cc, std::integral_constant<int, 0>{});
        chunk(cc + 1, std::integral_constant<int, 1>{});
    }
    if (nch & 1) chunk(nch - 1, std::integral_constant<int, 0>{});

    const float inv = 1.0f / (ascale * F16_WSCALE);
    const int half = lane >> 5, col = lane & 31;
    const int co = nt * 32 + col;
    const int oc2 = a.Cout - a.OC1;
    if (co < a.Cout) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                const int by = by0 + 2 * (wm * TM + i) + (row >> 4), bx = bx0 + (row & 15);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const size_t opix = ((size_t)n * g.OH + 2 * by + (c >> 1)) * g.OW + 2 * bx + (c & 1);
                    if (co < a.OC1) a.out[opix * a.OC1 + co] = acc[c][i][e] * inv;
                    else a.out2[opix * oc2 + (co - a.OC1)] = acc[c][i][e] * inv;
                }
            }
    }
}

}  // namespace

// conv: 3x3, stride (2,2), pad 1, no dilation, even input extent; channel counts that tile (K % 32, Cin % 64)
bool viai_dgrad_s2_ok(const viai_conv2d* c) {
    if (c->transposed || c->kh != 3 || c->kw != 3 || c->sh != 2 || c->sw != 2 || c->ph != 1 || c->pw != 1) return false;
    if ((c->dh > 1) || (c->dw > 1) || (c->ph2 >= 0 && c->ph2 != 1) || (c->pw2 >= 0 && c->pw2 != 1)) return false;
    if ((c->IH & 1) || (c->IW & 1) || c->Cout % 32 != 0 || (c->C1 + c->C2) % 64 != 0) return false;
    if (c->C2 > 0 && c->C1 % 32 != 0) return false;
    // enough 128-pixel x 64-channel tiles to occupy the chip; smaller layers go class by class through the split-K kernel
    if ((long)c->N * (c->IH / 2) * (c->IW / 2) >= (1L << 22)) return false;      // epilogue index arithmetic in fp32
    long blocks = (((long)c->N * (c->IH / 2) * (c->IW / 2) + 127) / 128) * ((c->C1 + c->C2) / 64);
    // where the base lattice tiles in 8 x 16 (patch-staged kernel) even a fraction of a round beats four class-by-class launches of
    // the split-K kernel (E.conv4 / E.conv5: 4 x ~30 us -> one ~25 us launch)
    const bool tiles = ((c->IH / 2) % 8 == 0) && ((c->IW / 2) % 16 == 0);
    constexpr long small = 32;
    return blocks >= (tiles ? small : 256);
}

int viai_conv_dgrad_s2_bf3_launch(ConvArgs& a, hipStream_t st) {
    // a.g: N, IH/IW = dy extent, OH/OW = dx extent; a.C1 = conv Cout (K), a.Cout = conv Cin, a.M = N * (OH/2) * (OW/2)
    a.nblk_m = (a.M + 127) / 128;
    a.nblk_n = (a.Cout + 63) / 64;
    constexpr int lds = 2 * 3 * 128 * BF3_PITCH;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_bf3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_bf3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_bf3_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    constexpr int patch = 1;
    if (a.amax != nullptr && patch && a.g.IH % 8 == 0 && a.g.IW % 16 == 0 && a.g.OH == 2 * a.g.IH && a.g.OW == 2 * a.g.IW && a.C1 % 32 == 0) {
        constexpr int lds_p = 2 * 2 * 9 * 1536;
        static bool attr_p = false;
        if (!attr_p) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_patch_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_patch_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p);
            attr_p = true;
        }
        viai_tag_kernel("dgrad_s2_patch_f16x2");
        if (a.in_p16) VIAI_LAUNCH(conv_dgrad_s2_patch_kernel<true>, dim3(a.nblk_m * a.nblk_n), dim3(256), lds_p, st, a);
        else VIAI_LAUNCH(conv_dgrad_s2_patch_kernel<false>, dim3(a.nblk_m * a.nblk_n), dim3(256), lds_p, st, a);
        return viai_launch_status();
    }
    if (a.in_p16 && (a.amax == nullptr || a.C1 % 32 != 0)) return (int)hipErrorInvalidValue;
    viai_tag_kernel(a.amax != nullptr ? "dgrad_s2_f16x2" : "dgrad_s2_bf16x3");
    if (a.in_p16) VIAI_LAUNCH((conv_dgrad_s2_bf3_kernel<2, true>), dim3(a.nblk_m * a.nblk_n), dim3(256), 2 * 2 * 128 * BF3_PITCH, st, a);
    else if (a.amax != nullptr) VIAI_LAUNCH(conv_dgrad_s2_bf3_kernel<2>, dim3(a.nblk_m * a.nblk_n), dim3(256), 2 * 2 * 128 * BF3_PITCH, st, a);   // f16x2
    else VIAI_LAUNCH(conv_dgrad_s2_bf3_kernel<3>, dim3(a.nblk_m * a.nblk_n), dim3(256), lds, st, a);
    return viai_launch_status();
}
