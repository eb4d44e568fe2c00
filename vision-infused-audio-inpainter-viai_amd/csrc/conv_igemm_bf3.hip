// Implicit-GEMM convolution on the bf16 matrix cores with fp32-grade accuracy ("bf16x3" split), gfx950.
//
// Every fp32 operand is split into three bf16 terms  a = a1 + a2 + a3  (a1 = bf16(a), a2 = bf16(a - a1),
// a3 = bf16(a - a1 - a2); 3 x 8 = 24 significand bits, i.e. the whole fp32 value) and the product a*b is
// accumulated in fp32 from the six partial products whose weight is >= 2^-16 relative:
//     a1b1 + a1b2 + a2b1 + a1b3 + a3b1 + a2b2          (dropped: a2b3, a3b2, a3b3 <= 2^-23 |ab|)
// on v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense): 6 bf16 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64
// cycles per 32x32x16 block, a 2.67x higher MFMA ceiling (416 TFLOP/s fp32-equivalent) at the rounding-error
// level of the exact-fp32 kernel (tests/test_kernels_gpu.py measures both against fp64).
//
// Same gather-GEMM structure as conv_igemm.hip (tap tables, buffer loads with hardware zero fill, XCD remap,
// BN partial statistics in the epilogue).  Differences: weights arrive pre-split as three bf16 planes
// (viai_conv2d_pack_* does it once per use), activations are split while being staged into LDS, LDS holds bf16
// planes [plane][row][32 + 8 pad] (80-byte rows: conflict-free ds_read_b128 operand fetches).
#include "viai_common.h"
#include "viai_internal.h"
#include <string>
#include "viai_bf3.h"
// timing ablation of the wide kernel: bit 0 no weight-fragment loads, bit 1 no activation loads, bit 2 no split / LDS stores
#ifndef VIAI_ABL
#define VIAI_ABL 0
#endif
#include <cstdlib>

namespace {


// Shared epilogue: bias + activation (or raw output + per-block BatchNorm partial statistics), NHWC stores with the
// dgrad parity-class scatter and the two-destination (virtual concat) split.  Same contract as conv_igemm.hip.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void bf3_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], unsigned char* smem_b, int lane, int wm, int wn,
                                             int m0, int n0, int bm) {
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    const ConvGeom& g = a.g;

    const int half = lane >> 5, col = lane & 31;
    float bv[TN];
    int co[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        co[j] = n0 + (wn * TN + j) * 32 + col;
        bv[j] = (a.bias != nullptr && co[j] < a.Cout) ? a.bias[co[j]] : 0.f;
    }
    const bool ident = (g.ly == 1 && g.lx == 1 && g.SH == g.OH && g.SW == g.OW);
    const int oc2 = a.Cout - a.OC1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int m = m0 + (wm * TM + i) * 32 + row;
            if (m < a.M) {
                size_t opix;
                if (ident) opix = (size_t)m;
                else {
                    int ox = m % g.SW; int t = m / g.SW; int oy = t % g.SH; int n = t / g.SH;
                    opix = ((size_t)n * g.OH + (oy * g.ly + g.ay)) * g.OW + (ox * g.lx + g.ax);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = acc[i][j][e] + bv[j];
                    if (a.stat == nullptr) v = viai_act(v, a.act, a.slope);
                    acc[i][j][e] = v;
                    if (co[j] < a.Cout) {
                        if (co[j] < a.OC1) a.out[opix * a.OC1 + co[j]] = v;
                        else a.out2[opix * oc2 + (co[j] - a.OC1)] = v;
                    }
                }
            }
        }
    }
    if (a.stat != nullptr) {
        float* red = reinterpret_cast<float*>(smem_b);
        const int cnt = min(BM, a.M - m0);
        float s[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                    int m = m0 + (wm * TM + i) * 32 + row;
                    t += (m < a.M) ? acc[i][j][e] : 0.f;
                }
            t += __shfl_xor(t, 32, 64);
            s[j] = t;
        }
        if (half == 0)
#pragma unroll
            for (int j = 0; j < TN; ++j) red[wm * BN + (wn * TN + j) * 32 + col] = s[j];
        __syncthreads();
        float mean[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[w * BN + (wn * TN + j) * 32 + col];
            mean[j] = t / (float)cnt;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                    int m = m0 + (wm * TM + i) * 32 + row;
                    float d = acc[i][j][e] - mean[j];
                    t += (m < a.M) ? d * d : 0.f;
                }
            t += __shfl_xor(t, 32, 64);
            s[j] = t;
        }
        if (half == 0)
#pragma unroll
            for (int j = 0; j < TN; ++j) red[wm * BN + (wn * TN + j) * 32 + col] = s[j];
        __syncthreads();
        if (wm == 0 && half == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (co[j] < a.Cout) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) t += red[w * BN + (wn * TN + j) * 32 + col];
                    a.stat[(size_t)co[j] * a.nblk_m + bm] = mean[j];
                    a.stat[(size_t)(a.Cout + co[j]) * a.nblk_m + bm] = t;
                }
        }
    }
}

// Weights never touch LDS: viai_conv2d_pack_* stores them "fragment-major" -- for every (plane, 32-channel output
// tile, tap, 16-deep k-step) the 64 lanes' 16-byte MFMA B fragments are contiguous (1 KiB), so a wave fetches a
// B operand with ONE fully coalesced buffer_load_dwordx4 (L1/L2-resident, shared by every block) one k-step
// ahead of its use.  LDS holds only the activation planes, double-buffered: one barrier per 32-deep chunk and
// the split + LDS stores of chunk k+1 overlap the MFMAs of chunk k.

// NP = 3: bf16x3 split (six partial products); NP = 2: f16x2 split (three partial products, forward launches, viai_bf3.h)
template <int NP, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_bf3_frag_kernel(const ConvArgs a) {
    constexpr int BK = BF3_BK;
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int APLANE = BM * BF3_PITCH;
    constexpr int STAGE = NP * APLANE;
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int SPLIT_VALU_PER_MFMA = 4;
    constexpr int RP = 8 * WM * WN;                     // A rows staged per pass (8 threads = 8 channel quads per row)
    constexpr int NA = BM / RP;                         // A float4 per thread per chunk
    static_assert(BM % RP == 0, "config");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // [2][3][BM][80]

    const ConvGeom& g = a.g;
    // f16x2 operand scale: static for forward activations, derived from the producer's abs-max for gradients
    const float ascale = (NP == 2 && a.amax != nullptr) ? f16_scale_from_amax(a.amax) : F16_ASCALE;
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % a.nblk_n, bm = bid / a.nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const int Cin = a.C1 + a.C2;

    const int q = tid & 7, r0 = tid >> 3;
    int pixbase[NA], iy0[NA], ix0[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int m = m0 + r0 + RP * j;
        if (m < a.M) {
            int ox = m % g.SW; int t = m / g.SW; int oy = t % g.SH; int n = t / g.SH;
            iy0[j] = oy * g.my; ix0[j] = ox * g.mx;
            pixbase[j] = (n * g.IH + iy0[j]) * g.IW + ix0[j];
        } else { iy0[j] = -100000; ix0[j] = -100000; pixbase[j] = 0; }
    }
    constexpr int OOB = 0x7fffffff;
    const int k16 = Cin / 16;                                        // 16-deep k-steps per tap
    const int NT = (a.Cout + 31) / 32;                               // 32-channel output tiles in the packed weights
    const int frag_plane = NT * g.wtaps * k16 * 1024;                // bytes per bf16 plane (fragment-major)
    int bbase[TN];                                                   // byte offset of this wave's n-tiles (+ lane)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int nt = (n0 >> 5) + wn * TN + j;
        bbase[j] = (nt < NT) ? nt * g.wtaps * k16 * 1024 + lane * 16 : OOB;
    }
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, NP * frag_plane, 0x00020000);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int cchunks = (Cin + BK - 1) / BK;
    const int nchunks = g.ntaps * cchunks;

    // per-row tap validity (bit t set: tap t reads inside the image), so the K loop only tests a bit
    unsigned tapok[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        unsigned m = 0;
        for (int t = 0; t < g.ntaps; ++t) {
            int iy = iy0[j] + g.dy[t], ix = ix0[j] + g.dx[t];
            m |= ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) ? (1u << t) : 0u;
        }
        tapok[j] = m;
    }

    // Software pipeline, three stages deep: while the MFMAs of chunk k run, the raw fp32 rows of chunk k+1 (loaded
    // during chunk k-1) are split into bf16 planes and stored to the other LDS stage, and the loads of chunk k+2 are
    // in flight.  The split is ~90 VALU instructions per chunk; issued between the MFMAs they cost nothing.
    auto gloadA = [&](u32x4 (&raw)[NA], int t_, int c0_) {
        const int t = __builtin_amdgcn_readfirstlane(t_);
        const int c0 = __builtin_amdgcn_readfirstlane(c0_);
        const bool live = t < g.ntaps;
        const int tt = live ? t : 0;
        const int toff = g.dy[tt] * g.IW + g.dx[tt];
        const bool first = c0 < a.C1;
        const int cs = first ? a.C1 : a.C2;
        const int coff = (first ? c0 : c0 - a.C1) + q * 4;
        // no selects and no && below: the compiler turns them into branches (and sinks the address multiply into them), which
        // splits the K loop's basic block, forces s_waitcnt vmcnt(0) at the joins and breaks the MFMA / VALU interleave.
        // An element outside the image / beyond Cin / past the last tap gets offset bits 0x7fffffff OR-ed in: out of range
        // for the descriptor, so the load returns zeros.
        const int dead_s = live ? 0 : OOB;                                        // scalar
        const int dead_l = ((Cin - 1 - (c0 + q * 4)) >> 31) & OOB;               // per lane: channel quad beyond Cin
        const float* src = first ? a.in : a.in2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(in_pixels * cs * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int dead_t = (int)(((tapok[j] >> tt) & 1u) - 1u) & OOB;        // tap outside the image for this row
            const int off = (((pixbase[j] + toff) * cs + coff) * 4) | dead_t | dead_l | dead_s;
#if VIAI_ABL & 2
            raw[j] = u32x4{(unsigned)off, 0u, 0u, 0u};
#else
            raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
#endif
        }
    };
    // B fragments of one 16-deep k-step: (tap t, channel offset c) -> 3 planes x TN tiles
    auto gloadB = [&](u32x4 (&bf)[TN][NP], int t_, int c_) {
        const int t = __builtin_amdgcn_readfirstlane(t_);
        const int c = __builtin_amdgcn_readfirstlane(c_);
        const int dead = ((t < g.ntaps) & (c < Cin)) ? 0 : OOB;      // scalar; OR-ed into the offset: zeros past the end
        const int tt = t < g.ntaps ? t : 0;                          // unconditional table read: no branch in the K loop
        const int koff = (g.ws[tt] * k16 + (c >> 4)) * 1024;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#if VIAI_ABL & 1
                bf[j][p] = u32x4{(unsigned)lane, (unsigned)koff, 0x3c003c00u, 0x3c003c00u};
#else
                bf[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bbase[j] | dead, koff + p * frag_plane, 0);
#endif
    };
    auto lstore = [&](const u32x4 (&raw)[NA], int buf) {
        unsigned char* As = smem_b + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const f32x4 v = __builtin_bit_cast(f32x4, raw[j]);
            unsigned char* d = As + (r0 + RP * j) * BF3_PITCH + q * 8;
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
    auto advance = [&](int& t, int& c) { c += BK; if (c >= Cin) { c = 0; ++t; } };

    int t_cur = 0, c_cur = 0;                 // chunk k (being multiplied)
    int t_n1 = 0, c_n1 = 0;                   // chunk k+1 (being split)
    advance(t_n1, c_n1);
    int t_n2 = t_n1, c_n2 = c_n1;             // chunk k+2 (being loaded)
    advance(t_n2, c_n2);
    u32x4 rawA[NA], rawB[NA];
    // weight fragments: bf16x3 fetches one k-step (24 MFMAs) ahead; f16x2 has half the MFMAs per k-step, too few to cover an
    // L2 round trip, so it fetches one whole chunk ahead into a second pair of fragment sets
    constexpr bool BDEEP = NP == 2;
    u32x4 bf0[TN][NP], bf1[TN][NP], bg0[BDEEP ? TN : 1][NP], bg1[BDEEP ? TN : 1][NP];
    gloadA(rawA, 0, 0);
    gloadB(bf0, 0, 0);
    if constexpr (BDEEP) gloadB(bf1, 0, 16);
    gloadA(rawB, t_n1, c_n1);
    lstore(rawA, 0);
    __syncthreads();

    const int aoff = (wm * TM * 32 + (lane & 31)) * BF3_PITCH + 16 * (lane >> 5);
    // partial products, smallest first: bf16x3 a2b2 a3b1 a1b3 a2b1 a1b2 a1b1; f16x2 a2b1 a1b2 a1b1
    constexpr int PA[6] = {NP == 3 ? 1 : 1, NP == 3 ? 2 : 0, 0, 1, 0, 0}, PB[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
    auto mfma = [](const u32x4& x, const u32x4& y, const f32x16& c) {
        if constexpr (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    };

    auto step = [&](u32x4 (&rload)[NA], const u32x4 (&rconv)[NA], int cur, u32x4 (&bf0)[TN][NP], u32x4 (&bf1)[TN][NP],
                    u32x4 (&bn0)[BDEEP ? TN : 1][NP], u32x4 (&bn1)[BDEEP ? TN : 1][NP]) {
        gloadA(rload, t_n2, c_n2);
        const unsigned char* As = smem_b + cur * STAGE + aoff;
        if constexpr (BDEEP) { gloadB(bn0, t_n1, c_n1); gloadB(bn1, t_n1, c_n1 + 16); }
        else gloadB(bf1, t_cur, c_cur + 16);
        {
            u32x4 af[TM][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[i][p] = *reinterpret_cast<const u32x4*>(As + p * APLANE + i * 32 * BF3_PITCH);
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma(af[i][PA[pr]], bf0[j][PB[pr]], acc[i][j]);
        }
        if constexpr (!BDEEP) gloadB(bf0, t_n1, c_n1);
        {
            u32x4 af[TM][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[i][p] = *reinterpret_cast<const u32x4*>(As + p * APLANE + i * 32 * BF3_PITCH + 32);
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mfma(af[i][PA[pr]], bf1[j][PB[pr]], acc[i][j]);
        }
#if !(VIAI_ABL & 4)
        lstore(rconv, cur ^ 1);
#endif
        // interleave: one MFMA, then a few of the split's VALU instructions, so the split runs in the MFMA shadow
#pragma unroll
        for (int i = 0; i < 2 * NPROD * TM * TN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, SPLIT_VALU_PER_MFMA, 0);
        }
        __syncthreads();
        t_cur = t_n1; c_cur = c_n1;
        t_n1 = t_n2; c_n1 = c_n2;
        advance(t_n2, c_n2);
    };
    // two chunks per trip (the register sets swap roles), the odd last chunk peeled: no branch inside the loop
    for (int kc = 0; kc + 1 < nchunks; kc += 2) {
        step(rawA, rawB, 0, bf0, bf1, bg0, bg1);
        if constexpr (BDEEP) step(rawB, rawA, 1, bg0, bg1, bf0, bf1);
        else step(rawB, rawA, 1, bf0, bf1, bg0, bg1);
    }
    if (nchunks & 1) step(rawA, rawB, 0, bf0, bf1, bg0, bg1);

    if constexpr (NP == 2) {                     // undo the operand scales (exact powers of two)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= 1.0f / (ascale * F16_WSCALE);
    }
    bf3_epilogue<TM, TN, WM, WN>(a, acc, smem_b, lane, wm, wn, m0, n0, bm);
}

// Variant for narrow / small tiles: weights as three row-major bf16 planes [plane][co][tap][ci], staged through LDS
// next to the activation planes (one LDS stage, two barriers per chunk).
// NP = 3: bf16x3; NP = 2: f16x2 (planar fp16 planes, operand scales as in the fragment-major kernel)
template <int TM, int TN, int WM, int WN, int NP = 3>
__global__ __launch_bounds__(256) void conv_igemm_bf3_lds_kernel(const ConvArgs a) {
    constexpr int BK = BF3_BK;
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int NA = BM / 32;                         // A float4 per thread per chunk (8 quads per row)
    constexpr int NBQ = (NP * BN * 4 + 255) / 256;      // B 16-byte pieces per thread per chunk
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int APLANE = BM * BF3_PITCH, BPLANE = BN * BF3_PITCH;
    static_assert(WM * WN == 4, "config");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;                         // [3][BM][80]
    unsigned char* Bs = smem_b + NP * APLANE;           // [NP][BN][80]

    const ConvGeom& g = a.g;
    const float ascale = (NP == 2 && a.amax != nullptr) ? f16_scale_from_amax(a.amax) : F16_ASCALE;
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % a.nblk_n, bm = bid / a.nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const int Cin = a.C1 + a.C2;

    const int q = tid & 7, r0 = tid >> 3;
    int pixbase[NA], iy0[NA], ix0[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int m = m0 + r0 + 32 * j;
        if (m < a.M) {
            int ox = m % g.SW; int t = m / g.SW; int oy = t % g.SH; int n = t / g.SH;
            iy0[j] = oy * g.my; ix0[j] = ox * g.mx;
            pixbase[j] = (n * g.IH + iy0[j]) * g.IW + ix0[j];
        } else { iy0[j] = -100000; ix0[j] = -100000; pixbase[j] = 0; }
    }
    constexpr int OOB = 0x7fffffff;
    // B pieces: idx = tid + 256*j -> (plane, row, 16-byte quad)
    const long wplane = (long)a.Cout * g.wtaps * Cin * 2;            // bytes per bf16 plane
    int bsrc[NBQ], bdst[NBQ];
#pragma unroll
    for (int j = 0; j < NBQ; ++j) {
        int idx = tid + 256 * j;
        int plane = idx / (BN * 4), rem = idx % (BN * 4), row = rem >> 2, q16 = rem & 3;
        int co = n0 + row;
        bsrc[j] = (idx < NP * BN * 4 && co < a.Cout) ? (int)(plane * wplane + ((long)co * g.wtaps * Cin) * 2 + q16 * 16) : OOB;
        bdst[j] = (idx < NP * BN * 4) ? plane * BPLANE + row * BF3_PITCH + q16 * 16 : -1;
    }
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (int)(NP * wplane), 0x00020000);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int cchunks = (Cin + BK - 1) / BK;
    const int nchunks = g.ntaps * cchunks;

    u32x4 areg[NA], breg[NBQ];
    auto gload = [&](int t_, int c0_) {
        const int t = __builtin_amdgcn_readfirstlane(t_);
        const int c0 = __builtin_amdgcn_readfirstlane(c0_);
        const int dyt = g.dy[t], dxt = g.dx[t];
        const int toff = dyt * g.IW + dxt;
        const bool first = c0 < a.C1;
        const int cs = first ? a.C1 : a.C2;
        const int coff = (first ? c0 : c0 - a.C1) + q * 4;
        // masks, not selects / &&: see conv_igemm_bf3_frag_kernel (an invalid element gets 0x7fffffff OR-ed into its offset)
        const int dead_l = ((Cin - 1 - (c0 + q * 4)) >> 31) & OOB;
        const float* src = first ? a.in : a.in2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(in_pixels * cs * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int iy = iy0[j] + dyt, ix = ix0[j] + dxt;
            const int dead_t = (((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix) >> 31) & OOB;     // any of the four negative: outside
            const int off = (((pixbase[j] + toff) * cs + coff) * 4) | dead_t | dead_l;
            areg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
        const int woff = (g.ws[t] * Cin + c0) * 2;
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
            int idx = tid + 256 * j;
            int q16 = idx & 3;
            const int dead_b = ((Cin - 1 - (c0 + q16 * 8)) >> 31) & OOB;
            breg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)((unsigned)bsrc[j] + (unsigned)woff) | dead_b | (bsrc[j] == OOB ? OOB : 0), 0, 0);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const f32x4 v = __builtin_bit_cast(f32x4, areg[j]);
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
#pragma unroll
        for (int j = 0; j < NBQ; ++j)
            if ((NP * BN * 4) % 256 == 0 || bdst[j] >= 0) *reinterpret_cast<u32x4*>(Bs + bdst[j]) = breg[j];
    };

    int t_next = 0, c_next = 0;
    gload(0, 0);
    lstore();
    c_next = BK;
    if (c_next >= Cin) { c_next = 0; t_next = 1; }
    __syncthreads();

    const int aoff = (wm * TM * 32 + (lane & 31)) * BF3_PITCH + 16 * (lane >> 5);
    const int boff = (wn * TN * 32 + (lane & 31)) * BF3_PITCH + 16 * (lane >> 5);

    for (int kc = 0; kc < nchunks; ++kc) {
        const bool more = (kc + 1 < nchunks);
        if (more) {
            gload(t_next, c_next);
            c_next += BK;
            if (c_next >= Cin) { c_next = 0; ++t_next; }
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            u32x4 af[TM][NP], bf[TN][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[i][p] = *reinterpret_cast<const u32x4*>(As + p * APLANE + aoff + i * 32 * BF3_PITCH + ks * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bf[j][p] = *reinterpret_cast<const u32x4*>(Bs + p * BPLANE + boff + j * 32 * BF3_PITCH + ks * 32);
            // smallest terms first: bf16x3 a2b2 a3b1 a1b3 a2b1 a1b2 a1b1; f16x2 a2b1 a1b2 a1b1
            constexpr int PA[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0}, PB[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pr = 0; pr < NPROD; ++pr) {
                        if constexpr (NP == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][PA[pr]]), __builtin_bit_cast(bf16x8, bf[j][PB[pr]]), acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][PA[pr]]), __builtin_bit_cast(f16x8, bf[j][PB[pr]]), acc[i][j], 0, 0, 0);
                    }
        }
        __syncthreads();                 // every wave is done reading this chunk
        if (more) lstore();
        __syncthreads();
    }

    if constexpr (NP == 2) {                     // undo the operand scales (exact powers of two)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= 1.0f / (ascale * F16_WSCALE);
    }
    bf3_epilogue<TM, TN, WM, WN>(a, acc, smem_b, lane, wm, wn, m0, n0, bm);
}

// Variant for layers with few output tiles (small M): 32 x 32 tile, 64-deep chunks, and the block's four waves split
// K -- wave w multiplies k-step w of every chunk -- so a layer with M = 512 ... 32768 pixels still fills the chip
// (16x more blocks than 64 x 64 tiles x 4 waves would give per wave-tile) and the epilogue (bias / activation /
// BatchNorm partials) stays fused: the four partial accumulators are summed through LDS and wave 0 finishes the tile.
// Planar weights [plane][co][tap][ci] staged through LDS, one stage, register-prefetched global loads.
template <int NP>
__global__ __launch_bounds__(256) void conv_igemm_bf3_sk_kernel(const ConvArgs a) {
    constexpr int BM = 32, BN = 32, BK = 64;
    constexpr int PITCH = BK * 2 + 16;                  // 144-byte rows: conflict-free ds_read_b128
    constexpr int APLANE = BM * PITCH, BPLANE = BN * PITCH;
    constexpr int NA = 2, NBQ = NP;                     // A float4 / B 16-byte pieces per thread per chunk

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;                         // [3][32][144]
    unsigned char* Bs = smem_b + NP * APLANE;           // [NP][32][144]

    const ConvGeom& g = a.g;
    const float ascale = (NP == 2 && a.amax != nullptr) ? f16_scale_from_amax(a.amax) : F16_ASCALE;
    const float alim = f16_clamp_for_scale(ascale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % a.nblk_n, bm = bid / a.nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const int Cin = a.C1 + a.C2;

    const int q = tid & 15, r0 = tid >> 4;              // A: row r0 + 16 j, channel quad q (16 quads = 64 channels)
    int pixbase[NA];
    unsigned tapok[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int m = m0 + r0 + 16 * j;
        unsigned msk = 0;
        pixbase[j] = 0;
        if (m < a.M) {
            int ox = m % g.SW; int t = m / g.SW; int oy = t % g.SH; int n = t / g.SH;
            int iy0 = oy * g.my, ix0 = ox * g.mx;
            pixbase[j] = (n * g.IH + iy0) * g.IW + ix0;
            for (int tt = 0; tt < g.ntaps; ++tt) {
                int iy = iy0 + g.dy[tt], ix = ix0 + g.dx[tt];
                msk |= ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) ? (1u << tt) : 0u;
            }
        }
        tapok[j] = msk;
    }
    constexpr int OOB = 0x7fffffff;
    const long wplane = (long)a.Cout * g.wtaps * Cin * 2;            // bytes per bf16 plane
    int bsrc[NBQ], bdst[NBQ];
#pragma unroll
    for (int j = 0; j < NBQ; ++j) {                                  // idx -> (plane, row, 16-byte piece of the 128-byte k row)
        int idx = tid + 256 * j;
        int plane = idx >> 8, rem = idx & 255, row = rem >> 3, q16 = rem & 7;
        int co = n0 + row;
        bsrc[j] = (co < a.Cout) ? (int)(plane * wplane + ((long)co * g.wtaps * Cin) * 2 + q16 * 16) : OOB;
        bdst[j] = plane * BPLANE + row * PITCH + q16 * 16;
    }
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (int)(NP * wplane), 0x00020000);

    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }

    const int cchunks = (Cin + BK - 1) / BK;
    const int nchunks = g.ntaps * cchunks;

    // global loads run one (bf16x3) or two (f16x2, see below) chunks ahead of the MFMAs; a three-deep ring of the larger
    // bf16x3 sets measured slower: the occupancy it costs the mid-sized layers outweighs the latency it hides on the smallest
    struct Regs { u32x4 a[NA]; u32x4 b[NBQ]; };
    int t_ld = 0, c_ld = 0;                         // next chunk to load
    auto gload = [&](Regs& R) {
        const int t = __builtin_amdgcn_readfirstlane(t_ld);
        const int c0 = __builtin_amdgcn_readfirstlane(c_ld);
        const bool live = t < g.ntaps;
        const int tt = live ? t : 0;
        const int toff = g.dy[tt] * g.IW + g.dx[tt];
        const bool first = c0 < a.C1;
        const int cs = first ? a.C1 : a.C2;
        const int coff = (first ? c0 : c0 - a.C1) + q * 4;
        const int dead_s = live ? 0 : OOB;
        const int dead_l = ((Cin - 1 - (c0 + q * 4)) >> 31) & OOB;
        const float* src = first ? a.in : a.in2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(in_pixels * cs * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int dead_t = (int)(((tapok[j] >> tt) & 1u) - 1u) & OOB;
            R.a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (((pixbase[j] + toff) * cs + coff) * 4) | dead_t | dead_l | dead_s, 0, 0);
        }
        const int woff = (g.ws[tt] * Cin + c0) * 2;
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
            const int q16 = (tid + 256 * j) & 7;
            const int dead_b = (((Cin - 1 - (c0 + q16 * 8)) >> 31) & OOB) | dead_s;
            R.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)((unsigned)bsrc[j] + (unsigned)woff) | dead_b | (bsrc[j] == OOB ? OOB : 0), 0, 0);
        }
        c_ld += BK;
        if (c_ld >= Cin) { c_ld = 0; ++t_ld; }
    };
    auto lstore = [&](const Regs& R) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const f32x4 v = __builtin_bit_cast(f32x4, R.a[j]);
            unsigned char* d = As + (r0 + 16 * j) * PITCH + q * 8;
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
#pragma unroll
        for (int j = 0; j < NBQ; ++j) *reinterpret_cast<u32x4*>(Bs + bdst[j]) = R.b[j];
    };

    // NP = 2: two register sets, so the loads run TWO chunks ahead (a chunk lasts ~0.45 us, about one L2 round trip: with one
    // set every chunk waited for its rows); bf16x3 keeps one set (its larger sets cost the mid-sized layers their occupancy)
    constexpr bool TWO = NP == 2;
    Regs R, R2;
    gload(R);
    lstore(R);
    gload(R);
    if constexpr (TWO) gload(R2);
    __syncthreads();

    const int foff = (lane & 31) * PITCH + wave * 32 + 16 * (lane >> 5);      // this wave's k-step of the chunk
    auto step = [&](Regs& Rs) {                             // Rs holds chunk k+1; it is stored after the MFMAs of chunk k, then refilled
        u32x4 afr[NP], bfr[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            afr[p] = *reinterpret_cast<const u32x4*>(As + p * APLANE + foff);
            bfr[p] = *reinterpret_cast<const u32x4*>(Bs + p * BPLANE + foff);
        }
        if constexpr (NP == 3) {
            bf16x8 af[3], bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) { af[p] = __builtin_bit_cast(bf16x8, afr[p]); bf[p] = __builtin_bit_cast(bf16x8, bfr[p]); }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc0, 0, 0, 0);     // smallest partial products first
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc1, 0, 0, 0);
        } else {
            f16x8 af[2], bf[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) { af[p] = __builtin_bit_cast(f16x8, afr[p]); bf[p] = __builtin_bit_cast(f16x8, bfr[p]); }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[0], acc0, 0, 0, 0);
        }
        __syncthreads();
        lstore(Rs);
        gload(Rs);                                          // chunk k+3 (two sets) or k+2 (zero-filled past the last one)
        __syncthreads();
    };
    if constexpr (TWO) {
        for (int kc = 0; kc + 1 < nchunks; kc += 2) { step(R); step(R2); }
        if (nchunks & 1) step(R);
    } else {
        for (int kc = 0; kc < nchunks; ++kc) step(R);
    }

    // ---- sum the four waves' partial tiles through LDS (fixed order), wave 0 finishes
    f32x16 acc = acc0 + acc1;
    if constexpr (NP == 2) acc *= 1.0f / (ascale * F16_WSCALE);      // undo the operand scales (exact powers of two)
    float* red = reinterpret_cast<float*>(smem_b);              // [3][16][64]
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((wave - 1) * 16 + e) * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += red[(w * 16 + e) * 64 + lane];

    const int half = lane >> 5, col = lane & 31;
    const int co = n0 + col;
    const float bv = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
    const bool ident = (g.ly == 1 && g.lx == 1 && g.SH == g.OH && g.SW == g.OW);
    const int oc2 = a.Cout - a.OC1;
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
        const int m = m0 + row;
        float v = acc[e] + bv;
        if (a.stat == nullptr) v = viai_act(v, a.act, a.slope);
        acc[e] = v;
        if (m < a.M) {
            size_t opix;
            if (ident) opix = (size_t)m;
            else {
                int ox = m % g.SW; int t = m / g.SW; int oy = t % g.SH; int n = t / g.SH;
                opix = ((size_t)n * g.OH + (oy * g.ly + g.ay)) * g.OW + (ox * g.lx + g.ax);
            }
            sum += v;
            if (co < a.Cout) {
                if (co < a.OC1) a.out[opix * a.OC1 + co] = v;
                else a.out2[opix * oc2 + (co - a.OC1)] = v;
            }
        }
    }
    if (a.stat != nullptr) {                 // block-local (mean, M2) over the tile's rows, inside one wave
        const int cnt = min(BM, a.M - m0);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum / (float)cnt;
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            float d = acc[e] - mean;
            m2 += (m0 + row < a.M) ? d * d : 0.f;
        }
        m2 += __shfl_xor(m2, 32, 64);
        if (half == 0 && co < a.Cout) {
            a.stat[(size_t)co * a.nblk_m + bm] = mean;
            a.stat[(size_t)(a.Cout + co) * a.nblk_m + bm] = m2;
        }
    }
}

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// fragment-major bf16 planes from w[no*s_no + ki*s_ki + t]:
//   dst[p][nt][t][kq][lane][e],  lane = kg*32 + j, e < 8:  value(no = nt*32 + j, k = kq*16 + kg*8 + e), zero for no >= n_out
// (blk / nblk: this block's index and the number of blocks working on the image)
__device__ __forceinline__ void pack_frag_body(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                               long s_no, long s_ki, int blk, int nblk) {
    const int NT = (n_out + 31) / 32, k16 = k_in / 16;
    const long plane = (long)NT * taps * k16 * 512;                  // bf16 elements per plane
    for (long i = blk * (long)blockDim.x + threadIdx.x; i < plane; i += (long)nblk * blockDim.x) {
        int e = (int)(i & 7); int lane = (int)((i >> 3) & 63); long r = i >> 9;
        int kq = (int)(r % k16); r /= k16; int t = (int)(r % taps); int nt = (int)(r / taps);
        int no = nt * 32 + (lane & 31), ki = kq * 16 + (lane >> 5) * 8 + e;
        float x = (no < n_out) ? w[no * s_no + ki * s_ki + t] : 0.f;
        unsigned short h1 = bf16_rne(x);
        float r1 = x - __uint_as_float((unsigned)h1 << 16);
        unsigned short h2 = bf16_rne(r1);
        float r2 = r1 - __uint_as_float((unsigned)h2 << 16);
        wp[i] = h1; wp[plane + i] = h2; wp[2 * plane + i] = bf16_rne(r2);
    }
}

// planes[p][no][t][ki] (bf16) from w[no*s_no + ki*s_ki + t]; same RNE split as the kernel's activations
__device__ __forceinline__ void pack_planar_body(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                                 long s_no, long s_ki, int blk, int nblk) {
    const long total = (long)n_out * taps * k_in;
    for (long i = blk * (long)blockDim.x + threadIdx.x; i < total; i += (long)nblk * blockDim.x) {
        int ki = (int)(i % k_in); long r = i / k_in; int t = (int)(r % taps); int no = (int)(r / taps);
        float x = w[no * s_no + ki * s_ki + t];
        unsigned short h1 = bf16_rne(x);
        float r1 = x - __uint_as_float((unsigned)h1 << 16);
        unsigned short h2 = bf16_rne(r1);
        float r2 = r1 - __uint_as_float((unsigned)h2 << 16);
        wp[i] = h1; wp[total + i] = h2; wp[2 * total + i] = bf16_rne(r2);
    }
}

// planes[p][no][t][ki] (fp16, two terms of w * F16_WSCALE): the planar image of the f16x2 LDS-weight / split-K kernels
__device__ __forceinline__ void pack_planar16_body(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                                   long s_no, long s_ki, int blk, int nblk) {
    const long total = (long)n_out * taps * k_in;
    for (long i = blk * (long)blockDim.x + threadIdx.x; i < total; i += (long)nblk * blockDim.x) {
        int ki = (int)(i % k_in); long r = i / k_in; int t = (int)(r % taps); int no = (int)(r / taps);
        // weights beyond the fp16 range of the scaled operand (|w| > 65504 / 256) saturate, like the activations, instead of becoming inf - inf
        const float x = __builtin_amdgcn_fmed3f(w[no * s_no + ki * s_ki + t] * F16_WSCALE, -65504.f, 65504.f);
        const _Float16 h1 = (_Float16)x;
        const _Float16 h2 = (_Float16)(x - (float)h1);
        wp[i] = __builtin_bit_cast(unsigned short, h1); wp[total + i] = __builtin_bit_cast(unsigned short, h2);
    }
}

// fragment-major f16x2 planes (forward f16x2 kernels): same geometry as pack_frag_body, two fp16 terms of w * F16_WSCALE
__device__ __forceinline__ void pack_frag16_body(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                                 long s_no, long s_ki, int blk, int nblk) {
    const int NT = (n_out + 31) / 32, k16 = k_in / 16;
    const long plane = (long)NT * taps * k16 * 512;
    for (long i = blk * (long)blockDim.x + threadIdx.x; i < plane; i += (long)nblk * blockDim.x) {
        int e = (int)(i & 7); int lane = (int)((i >> 3) & 63); long r = i >> 9;
        int kq = (int)(r % k16); r /= k16; int t = (int)(r % taps); int nt = (int)(r / taps);
        int no = nt * 32 + (lane & 31), ki = kq * 16 + (lane >> 5) * 8 + e;
        float x = (no < n_out) ? __builtin_amdgcn_fmed3f(w[no * s_no + ki * s_ki + t] * F16_WSCALE, -65504.f, 65504.f) : 0.f;   // saturate (see above)
        _Float16 h1 = (_Float16)x;
        _Float16 h2 = (_Float16)(x - (float)h1);
        wp[i] = __builtin_bit_cast(unsigned short, h1); wp[plane + i] = __builtin_bit_cast(unsigned short, h2);
    }
}

__global__ void pack_weight_bf3_frag_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                            long s_no, long s_ki) {
    pack_frag_body(w, wp, n_out, k_in, taps, s_no, s_ki, blockIdx.x, gridDim.x);
}
__global__ void pack_weight_f16_frag_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                            long s_no, long s_ki) {
    pack_frag16_body(w, wp, n_out, k_in, taps, s_no, s_ki, blockIdx.x, gridDim.x);
}
__global__ void pack_weight_f16_planar_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                              long s_no, long s_ki) {
    pack_planar16_body(w, wp, n_out, k_in, taps, s_no, s_ki, blockIdx.x, gridDim.x);
}
__global__ void pack_weight_bf3_planar_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int n_out, int k_in, int taps,
                                              long s_no, long s_ki) {
    pack_planar_body(w, wp, n_out, k_in, taps, s_no, s_ki, blockIdx.x, gridDim.x);
}

// every bf16x3 weight image of a model in ONE launch: a table of jobs, block -> job by its block range
__global__ void pack_jobs_kernel(const viai_pack_job* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                                      // last job whose first block is <= b (wave-uniform)
        int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    const viai_pack_job j = jobs[lo];
    if (j.frag == 2) {                                     // fp32 [no][t][ki] image of the streaming / exact-fp32 kernels
        const long total = (long)j.n_out * j.taps * j.k_in;
        const float* w = (const float*)j.w;
        float* wp = (float*)j.wp;
        for (long i = (b - j.blk0) * (long)blockDim.x + threadIdx.x; i < total; i += (long)j.nblk * blockDim.x) {
            int ki = (int)(i % j.k_in); long r = i / j.k_in; int t = (int)(r % j.taps); int no = (int)(r / j.taps);
            wp[i] = w[no * j.s_no + ki * j.s_ki + t];
        }
    } else if (j.frag == 4) pack_planar16_body((const float*)j.w, (unsigned short*)j.wp, j.n_out, j.k_in, j.taps, j.s_no, j.s_ki, b - j.blk0, j.nblk);
    else if (j.frag == 3) pack_frag16_body((const float*)j.w, (unsigned short*)j.wp, j.n_out, j.k_in, j.taps, j.s_no, j.s_ki, b - j.blk0, j.nblk);
    else if (j.frag) pack_frag_body((const float*)j.w, (unsigned short*)j.wp, j.n_out, j.k_in, j.taps, j.s_no, j.s_ki, b - j.blk0, j.nblk);
    else pack_planar_body((const float*)j.w, (unsigned short*)j.wp, j.n_out, j.k_in, j.taps, j.s_no, j.s_ki, b - j.blk0, j.nblk);
}

}  // namespace

template <bool FRAG, int TM, int TN, int WM, int WN, int NP = 3>
static int launch_bf3(ConvArgs& a, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    a.nblk_m = (a.M + BM - 1) / BM;
    a.nblk_n = (a.Cout + BN - 1) / BN;
    size_t lds = FRAG ? (size_t)2 * NP * BM * BF3_PITCH : (size_t)3 * (BM + BN) * BF3_PITCH;
    if (lds < (size_t)WM * BN * sizeof(float)) lds = (size_t)WM * BN * sizeof(float);
    void (*kern)(const ConvArgs);
    if constexpr (FRAG) kern = conv_igemm_bf3_frag_kernel<NP, TM, TN, WM, WN>;
    else kern = conv_igemm_bf3_lds_kernel<TM, TN, WM, WN, NP>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    static const std::string fam = "igemm" + std::to_string(BM) + "x" + std::to_string(BN) + (NP == 2 ? "_f16x2" : "_bf16x3");
    viai_tag_kernel(fam.c_str());
    VIAI_LAUNCH(kern, dim3(a.nblk_m * a.nblk_n), dim3(64 * WM * WN), lds, st, a);
    return viai_launch_status();
}

// Weight layout rule (must agree between pack and launch): wide tiles read fragment-major weights straight from
// global memory, narrow / small tiles (where all four waves would fetch the same fragments) stage planar weights in LDS.
bool viai_bf3_frag_layout(long M, int n_out) { return viai_igemm_tile_m(M, n_out) == 128 && n_out > 64; }

// The split-K kernel takes the layers whose 64 x 64 tiling would leave most CUs idle.
bool viai_bf3_sk_ok(long M, int n_out, int C1, int C2) {
    if ((C1 + C2) % 16 != 0 || (C2 > 0 && C1 % 64 != 0)) return false;
    constexpr int mode = 1;
    if (!mode) return false;
    long b64 = ((M + 63) / 64) * ((n_out + 63) / 64);
    return n_out > 32 && b64 < 512;
}

static int launch_bf3_sk(ConvArgs& a, hipStream_t st) {
    a.nblk_m = (a.M + 31) / 32;
    a.nblk_n = (a.Cout + 31) / 32;
    constexpr int lds = 2 * 3 * 32 * (64 * 2 + 16);
    viai_tag_kernel(a.wfrag == 4 ? "igemm_sk32x32_f16x2" : "igemm_sk32x32_bf16x3");
    if (a.wfrag == 4) VIAI_LAUNCH(conv_igemm_bf3_sk_kernel<2>, dim3(a.nblk_m * a.nblk_n), dim3(256), lds, st, a);       // planar f16x2 weights
    else VIAI_LAUNCH(conv_igemm_bf3_sk_kernel<3>, dim3(a.nblk_m * a.nblk_n), dim3(256), lds, st, a);
    return viai_launch_status();
}

int viai_conv_igemm_bf3_launch(ConvArgs& a, hipStream_t st) {
    const int Cin = a.C1 + a.C2;
    if (Cin % 16 != 0 || (a.C2 > 0 && a.C1 % 32 != 0)) return (int)hipErrorInvalidValue;
    if (a.OC1 % 32 != 0 && a.OC1 != a.Cout) return (int)hipErrorInvalidValue;
    if (a.in_p16 && !(a.wfrag == 3 && !a.sk && (viai_conv_halo_wide_ok(a) || viai_conv_lin_dma_ok(a)))) return (int)hipErrorInvalidValue;     // only the patch-staged kernels stage P16 pieces
    if (a.sk) return launch_bf3_sk(a, st);
    const int bm = viai_igemm_tile_m(a.M, a.Cout);           // same tile rule as the fp32 kernels (BN partial geometry)
    if (a.wfrag == 3) {                                                        // f16x2 weights
        if (viai_conv_lin_dma_ok(a)) { viai_tag_kernel("lin_dma_f16x2"); return viai_conv_lin_dma_launch(a, st); }   // pre-split input, maps that are not whole 8 x 16 tiles
        if (viai_conv_halo_wide_ok(a)) return viai_conv_halo_wide_launch(a, st);   // stride-1 3 x 3: patch staged once per chunk, not once per tap
        // 128 x 256 tile (eight waves) where the layer is wide and tall enough: every staged activation row then feeds 256
        // output channels, halving the load / split / LDS-store work per MFMA
        constexpr int wn4 = 1;
        if (wn4 && a.Cout % 256 == 0 && ((a.M + 127) / 128) * (a.Cout / 256) >= 256) {
            return launch_bf3<true, 2, 2, 2, 4, 2>(a, st);
        }
        return launch_bf3<true, 2, 2, 2, 2, 2>(a, st);
    }
    if (a.wfrag && a.wfrag != 4) return launch_bf3<true, 2, 2, 2, 2>(a, st);
    if (a.wfrag == 4) {                                                        // planar f16x2 weights
        if (bm == 64) return launch_bf3<false, 1, 1, 2, 2, 2>(a, st);
        if (a.Cout > 64) return launch_bf3<false, 2, 2, 2, 2, 2>(a, st);
        if (a.Cout > 32) return launch_bf3<false, 2, 1, 2, 2, 2>(a, st);
        return launch_bf3<false, 1, 1, 4, 1, 2>(a, st);
    }
    if (bm == 64) return launch_bf3<false, 1, 1, 2, 2>(a, st);
    if (a.Cout > 64) return launch_bf3<false, 2, 2, 2, 2>(a, st);
    if (a.Cout > 32) return launch_bf3<false, 2, 1, 2, 2>(a, st);
    return launch_bf3<false, 1, 1, 4, 1>(a, st);
}

size_t viai_bf3_packed_floats(int n_out, int k_in, int taps) {
    size_t elems = (size_t)((n_out + 31) / 32) * 32 * taps * k_in;       // bf16 elements per plane (either layout fits)
    return (3 * elems + 1) / 2;
}

int viai_pack_job_bf3(const float* w, void* wp, int n_out, int k_in, int taps, long s_no, long s_ki, int frag, viai_pack_job* job) {
    if (frag != 2 && k_in % 16 != 0) return (int)hipErrorInvalidValue;
    long total = (long)((frag == 1 || frag == 3) ? (n_out + 31) / 32 * 32 : n_out) * taps * k_in;
    long blocks = (total + 1023) / 1024;                   // four elements per thread
    // (a cap of 64 blocks per image left D.conv3's two 1.2 M-element images to 16 k threads walking 72 elements each, a chain of gather
    // latencies: the batched pack of D's arena -- on the critical path between Adam(D) and the G step's D forward -- took 36 us)
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    job->w = w; job->wp = wp; job->n_out = n_out; job->k_in = k_in; job->taps = taps; job->frag = frag;
    job->s_no = s_no; job->s_ki = s_ki; job->blk0 = 0; job->nblk = (int)blocks;
    return 0;
}

extern "C" int viai_pack_jobs_run(const viai_pack_job* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (njobs < 1 || total_blocks < 1 || jobs_dev == nullptr) return (int)hipErrorInvalidValue;
    VIAI_LAUNCH(pack_jobs_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    return viai_launch_status();
}

int viai_pack_weight_bf3(const float* w, void* wp, int n_out, int k_in, int taps, long s_no, long s_ki, int frag, hipStream_t st) {
    if (k_in % 16 != 0) return (int)hipErrorInvalidValue;
    long total = (long)((frag == 1 || frag == 3) ? (n_out + 31) / 32 * 32 : n_out) * taps * k_in;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (frag == 4) VIAI_LAUNCH(pack_weight_f16_planar_kernel, dim3(blocks), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wp), n_out, k_in, taps, s_no, s_ki);
    else if (frag == 3) VIAI_LAUNCH(pack_weight_f16_frag_kernel, dim3(blocks), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wp), n_out, k_in, taps, s_no, s_ki);
    else if (frag) VIAI_LAUNCH(pack_weight_bf3_frag_kernel, dim3(blocks), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wp), n_out, k_in, taps, s_no, s_ki);
    else VIAI_LAUNCH(pack_weight_bf3_planar_kernel, dim3(blocks), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(wp), n_out, k_in, taps, s_no, s_ki);
    return viai_launch_status();
}
