// Patch-staged weight gradient for the 3 x 3 layers, stride 1 and stride 2 (f16x2 split MFMA, gfx950).
//
//   G[t][co][ci] = sum over output pixels p of  dy[p][co] * x[p + t][ci]          (same contract as conv_wgrad_bf3.hip)
//
// wgrad_bf3_kernel runs one block per (tap, Cout tile, Cin tile): every x and dy tile is loaded, split and TRANSPOSED (both
// operands are channel-contiguous, the MFMA wants 8 consecutive pixels per lane) once per tap -- 9 non-MFMA VALU instructions per
// MFMA and 2.2x the algorithmic traffic on D.conv3.  This kernel removes both:
//
// * all nine taps in one block: a block owns 128 output channels x 64 input channels and walks the pixels in 8 x 16 tiles, HR tile
//   rows per stage.  Per stage the dy rows (HR x 16 pixels x 128 channels) and the x patch ((HR + 2) x 18 pixels x 64 channels) are
//   loaded and split ONCE; the nine taps are nine windows of the same patch.  Eight waves (4 along Cout x 2 along Cin); a wave keeps
//   a 32 x 32 tile of all nine filter positions in registers (9 accumulators = 144 AGPRs, two waves per SIMD), so a dy fragment is
//   read once per nine taps.  (A 64 x 32 wave tile would halve the LDS reads per MFMA but needs 288 accumulator registers; the
//   compiler keeps MFMA accumulators in the 256 AGPRs only and spills the rest.)
// * no transposition: the tiles go to LDS as they are in memory, [pixel][channel] fp16 planes, and the fragments come out through
//   gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block, every lane receives
//   the four pixels of ITS channel) -- two reads per 32 x 16 MFMA operand.
//
// Instances <S, BM, BN, HR, WK>  (BM x BN = Cout x Cin tile, HR tile rows per stage, WK = waves that split the k-steps of a stage):
//   <1, 128, 64, 4, 1>  stride 1: 8 waves (4 along Cout x 2 along Cin), two LDS stages of 74 KB, 1 block / CU
//   <1,  32, 32, 4, 4>  stride 1, narrow layers (the 32 / 64-channel layers of the generator): 4 waves that share ONE 32 x 32 tile and take
//                       one tile row each and add their accumulators through LDS at the end, 44 KB of LDS
//   <2, 128, 32, 2, 1>  stride 2: a stride-2 layer reads a (2 HR + 1) x 33 input patch per HR x 16 output pixels (4.4x the pixels of the stride-1
//                  case), so the Cin tile is 32 and a stage 2 tile rows; the patch is stored as four parity sub-patches
//                  P[py][px][r][c] = patch(2 r + py, 2 c + px), so that the 16 pixels of a k-step are again 16 consecutive LDS rows for every
//                  tap; 4 waves, 75 KB of LDS, 2 blocks / CU.
// Split-K over tile ranges, deterministic slab reduce (viai_wgrad_reduce) as before.
// Reference call sites: the autograd backward of nn.Conv2d(…, 3, 1, 1) / nn.ConvTranspose2d(…, 3, 1, 1)
// (networks/Discriminator_Networks.py:30 conv3, networks/New_Inpainting_Networks.py:24 TransConvBlock).
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"

namespace {

constexpr int WP_TW = 16, WP_TH = 8;                // output pixel tile

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgPatchSlots { int s[9]; };
#ifdef VIAI_PROF
__device__ unsigned long long* wg_prof_buf;      // tools/probes/wgrad_patch_bench.hip: [block][32 stages][4 stamps] shader-clock stamps of wave 0
#define WG_STAMP(s, i) do { if (tid == 0 && (s) < 32) wg_prof_buf[((size_t)blockIdx.x * 32 + (s)) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_STAMP(s, i) do { } while (0)
#endif

// two images per tile (stride 1, maps of at most 8 x 7 pixels, same extent in and out)
__host__ __device__ inline bool wgrad_patch_pairs(const ConvGeom& g) {
    if (!(g.my == 1 && g.mx == 1 && g.OH <= WP_TH && g.OW <= 7 && g.IH == g.OH && g.IW == g.OW && g.N >= 2 && g.ntaps == 9)) return false;
    int lo = g.dx[0], hi = g.dx[0];                      // the window must reach exactly one column to either side (pad 1)
    for (int t = 1; t < 9; ++t) { lo = g.dx[t] < lo ? g.dx[t] : lo; hi = g.dx[t] > hi ? g.dx[t] : hi; }
    return lo == -1 && hi == 1;
}
__host__ inline long wgrad_patch_tiles(const ConvGeom& g) {
    if (wgrad_patch_pairs(g)) return (g.N + 1) / 2;
    return (long)g.N * ((g.OH + WP_TH - 1) / WP_TH) * ((g.OW + WP_TW - 1) / WP_TW);
}

template <int S, int BM, int BN, int HR, int WK>
struct WgCfg {
    static constexpr int WMC = BM / 32, WN = BN / 32;               // waves along Cout / Cin
    static constexpr int NW = WMC * WN * WK;
    static constexpr int NTHR = 64 * NW;
    static constexpr int KSW = HR / WK;                              // k-steps (tile rows) per wave and stage
    static constexpr int PW = S == 1 ? WP_TW + 2 : 2 * WP_TW + 1;    // x patch columns (input pixels)
    static constexpr int PH = S == 1 ? HR + 2 : 2 * HR + 1;          // x patch rows
    static constexpr int XPIX = PH * PW;
    static constexpr int SUBW = 17;                                  // S = 2: columns of a parity sub-patch
    static constexpr int ROWSLOTS = S == 1 ? PW : SUBW;              // LDS slots between the windows of consecutive tile rows
    // S = 2 sub-patch bases (tight): (py, px) = (0,0): (HR + 1) rows, (0,1): (HR + 1), (1,0): HR, (1,1): HR
    static constexpr int XSLOTS = S == 1 ? XPIX : (4 * HR + 2) * SUBW;
    static constexpr int XPITCH = BN == 64 ? 192 : 64;               // 64 ch: +64 B pad; 32 ch: four 64-B rows are exactly one 256-B bank row
    static constexpr int DROW = BM * 2;                              // dy LDS row: 128 ch = 256 B with the 64-byte groups XOR-swizzled by (pixel & 3); 64 ch = 128 B, the two
                                                                     // groups swapped by bit 1 of the pixel (rows p, p+1 are the two halves of a 256-B bank row, p+2, p+3
                                                                     // take the other group: four distinct 64-B segments per four rows); 32 ch = 64 B
    static constexpr int DPIX = HR * WP_TW;
    // plane strides: + 64 bytes where the raw size is a multiple of 128 -- a P16 staging store (`ds_write_b128`: groups of 8 lanes, bank = (a / 4) mod 32,
    // MI355X_MICROARCH.md section LDS) writes the four leading and the four remainder pieces of a pixel's 32-channel group from the 8 lanes of one group, and
    // with the planes a multiple of 128 bytes apart the two halves hit the same banks (SQ_LDS_BANK_CONFLICT 0.135 on ResNet layer2's weight gradient)
    static constexpr int DPLANE = DPIX * DROW + ((DPIX * DROW) % 128 == 0 ? 64 : 0), XPLANE = XSLOTS * XPITCH + ((XSLOTS * XPITCH) % 128 == 0 ? 64 : 0);
    static constexpr int STAGE = 2 * DPLANE + 2 * XPLANE;
    static constexpr int LDS = 2 * STAGE;
    static constexpr int DQ = BM / 4;                                // channel quads per dy pixel
    static constexpr int PJ = NTHR / DQ;                             // dy pixels per staging pass
    static constexpr int ND = DPIX / PJ;                             // dy float4 items per thread and stage
    static constexpr int XQ = BN / 4;                                // channel quads per x pixel
    static constexpr int XPP = NTHR / XQ;                            // x pixels per staging pass
    static constexpr int NX = (XPIX + XPP - 1) / XPP;                // x float4 items per thread and stage (last one partial)
    static constexpr int UPK = 2 * (ND + NX) / KSW;                  // staging units (half items) per k-step of a wave
    static_assert((BM == 128 || BM == 64 || BM == 32) && (BN == 64 || BN == 32) && HR % WK == 0 && DPIX % PJ == 0 && (PJ == 8 || PJ % 16 == 0)
                  && (2 * (ND + NX)) % KSW == 0, "config");
    __host__ __device__ static constexpr int sub_base(int py, int px) {
        return S == 1 ? 0 : (py == 0 ? px * (HR + 1) * SUBW : 2 * (HR + 1) * SUBW + px * HR * SUBW);
    }
    // LDS slot (row of the x planes) of patch pixel (pr, pc)
    __host__ __device__ static constexpr int slot(int pr, int pc) {
        return S == 1 ? pr * PW + pc : sub_base(pr & 1, pc & 1) + (pr >> 1) * SUBW + (pc >> 1);
    }
    // first slot of the 16 consecutive pixels tile row r reads at window position (ty, tx)
    __host__ __device__ static constexpr int tap_slot(int r, int ty, int tx) {
        return S == 1 ? (r + ty) * PW + tx : sub_base(ty & 1, tx & 1) + (r + (ty >> 1)) * SUBW + (tx >> 1);
    }
};

// PD / PX: dy / x arrive pre-split (P16 planes, viai_bf3.h): the staged 16-byte item is a PIECE (8 channels of one plane) instead of a
// channel quad -- same global address, no split, one 16-byte LDS store into the piece's plane
template <int S, int BM, int BN, int HR, int WK, bool PD = false, bool PX = false>
__global__ __launch_bounds__(64 * (BM / 32) * (BN / 32) * WK) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_patch_f16_kernel(const WgradArgs a, int y0, int x0, WgPatchSlots slots) {
    using C = WgCfg<S, BM, BN, HR, WK>;
    constexpr int WP_DROW = C::DROW, WP_BM = BM;
    constexpr int DPLANE = C::DPLANE, XPLANE = C::XPLANE, STAGE = C::STAGE, ND = C::ND, NX = C::NX, XPITCH = C::XPITCH;
    constexpr int SPT = WP_TH / HR;                          // stages per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];      // [2 stages]{dy plane 0, dy plane 1, x plane 0, x plane 1}

    const ConvGeom& g = a.g;
    const float dscale = f16_scale_from_amax(a.amax), dlim = f16_clamp_for_scale(dscale);
    const float xscale = a.xmax != nullptr ? f16_scale_from_amax(a.xmax) : F16_ASCALE, xlim = f16_clamp_for_scale(xscale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (C::WMC * C::WN), wrem = wave % (C::WMC * C::WN);
    const int wm = wrem / C::WN, wn = wrem % C::WN;          // 32 output channels x 32 input channels per wave; wk: which k-steps of a stage

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_slab = a.nblk_ci * a.nblk_co;
    const int z = b / per_slab; b -= z * per_slab;
    const int bci = b % a.nblk_ci, bco = b / a.nblk_ci;
    const int co0 = bco * WP_BM, ci0 = bci * BN;
    const int Cin = a.C1 + a.C2;
    const bool first = ci0 < a.C1;
    const int xcs = first ? a.C1 : a.C2;
    const int xoff = first ? ci0 : ci0 - a.C1;
    constexpr int OOB = 0x7fffffff;
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? a.x : a.x2), 0, (int)(in_pixels * xcs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)a.M * a.Cout * 4), 0x00020000);

    // tiles cover the output map; where the map is not a multiple of 8 x 16 (the 56 / 28 / 14 / 7-pixel maps of the ResNet branch) the
    // pixels of a tile that fall outside are staged as zeros: dy = 0 contributes nothing to any tap
    // Maps of at most 8 x 7 pixels (the 7 x 7 maps of ResNet layer4) would fill 38 % of a tile: TWO images share one, side by side --
    // image n0 in tile columns 0 .. 7, image n0 + 1 in columns 8 .. 15.  Patch column 8 is the right padding of the first image and the
    // left padding of the second (both zero), tile column 7 is outside the first image (dy staged as zero), so every tap window of a
    // pixel still lies in its own image and the MFMA side of the kernel does not know the difference.
    const bool pair = S == 1 && wgrad_patch_pairs(g);
    const int tiles_x = pair ? 1 : (g.OW + WP_TW - 1) / WP_TW, tiles_y = (g.OH + WP_TH - 1) / WP_TH;
    const int ntiles = pair ? (g.N + 1) / 2 : g.N * tiles_y * tiles_x;
    const int tile0 = z * a.chunks_per_split;                 // chunks_per_split = tiles per K slab here
    int tile1 = tile0 + a.chunks_per_split;
    if (tile1 > ntiles) tile1 = ntiles;
    const int nst = (tile1 - tile0) * SPT;

    // ---- staging maps (thread -> items; everything but the stage origin is fixed per thread)
    // dy: item j = stage pixel tid / DQ + PJ j, channel quad q = tid % DQ
    constexpr int PJ = C::PJ;
    const int dq = tid % C::DQ, dp0 = tid / C::DQ;
    const int d_goff = ((((dp0 >> 4) * g.OW + (dp0 & 15)) * a.Cout) + co0 + dq * 4) * 4;       // pixel dp0 = (row dp0 >> 4, column dp0 & 15) of the stage
    // (P16: item dq is piece dq & 7 of 32-channel group dq >> 3: plane (dq >> 2) & 1, bytes 16 (dq & 3) .. of the group's 64-byte row)
    const int d_in = PD ? ((dq >> 2) & 1) * DPLANE + (dq & 3) * 16 : (dq & 7) * 8;
    const int d_lds = BM == 128 ? dp0 * WP_DROW + (((dq >> 3) ^ (dp0 & 3)) * 64) + d_in       // + j * PJ * WP_DROW   (PJ % 4 == 0: same swizzle)
                    : BM == 64  ? dp0 * WP_DROW + (((dq >> 3) ^ ((dp0 >> 1) & 1)) * 64) + d_in
                                : dp0 * WP_DROW + d_in;
    // x: item j = patch pixel (tid / XQ) + XPP j, channel quad q = tid % XQ
    const int xq = tid % C::XQ, xp0 = tid / C::XQ;

    u32x4 draw[ND], xraw[NX];
    // origin of stage s (uniform): image n, first output row / column, byte offset of the dy rows; a stage past the end is "dead"
    // (every load of it is pushed out of range and returns zeros that nobody stores)
    int g_n = 0, g_iy0 = 0, g_ix0 = 0, g_dbase = 0, g_dead = 0, g_oy0 = 0, g_ox0 = 0;
    auto gstage = [&](int s_) {
        const int s = __builtin_amdgcn_readfirstlane(s_);
        g_dead = s < nst ? 0 : OOB;
        const int sc = s < nst ? s : 0;
        const int tile = tile0 + sc / SPT, h = sc % SPT;
        const int tx = tile % tiles_x; const int r_ = tile / tiles_x; const int ty = r_ % tiles_y;
        g_n = pair ? 2 * (r_ / tiles_y) : r_ / tiles_y;
        const int oy0 = ty * WP_TH + h * HR, ox0 = tx * WP_TW;
        g_dbase = ((g_n * g.OH + oy0) * g.OW + ox0) * a.Cout * 4;
        g_oy0 = oy0; g_ox0 = ox0;
        g_iy0 = oy0 * S + y0; g_ix0 = ox0 * S + x0;
    };
    // item `it` (dy items first, then x items; compile-time index) of the stage set by gstage()
    auto gload_item = [&](int it) {
#pragma unroll
        for (int j = 0; j < ND; ++j)
            if (it == j) {
                const int row = (PJ * j) >> 4, dcol = (PJ * j) & 15;                 // pixel dp0 + PJ j: PJ = 8 (dp0 < 8) or a multiple of 16, so no carry
                const int oy = g_oy0 + row + (dp0 >> 4), ox = g_ox0 + dcol + (dp0 & 15);
                if (pair) {
                    const int second = ox >> 3, oxi = ox & 7;                                // which image of the pair, column in it
                    const int outside = (((g.OH - 1 - oy) | (g.OW - 1 - oxi) | (g.N - 1 - g_n - second)) >> 31) & OOB;
                    const int off = (((second * g.OH + row + (dp0 >> 4)) * g.OW + oxi) * a.Cout + co0 + dq * 4) * 4;
                    draw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off | g_dead | outside, g_dead ? 0 : g_dbase, 0);
                    continue;
                }
                const int outside = (((g.OH - 1 - oy) | (g.OW - 1 - ox)) >> 31) & OOB;       // partial tile at the bottom / right edge
                draw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (d_goff + (row * g.OW + dcol) * a.Cout * 4) | g_dead | outside, g_dead ? 0 : g_dbase, 0);
            }
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if (it == ND + j) {
                const int pp = xp0 + C::XPP * j;
                const int ppr = S == 1 ? (pp * 3641) >> 16 : (pp * 1986) >> 16;      // pp / PW (PW = 18: pp < 128; PW = 33: pp < 200)
                const int pc = pp - ppr * C::PW;
                const int second = (pair && pc >= 8) ? 1 : 0;                                 // patch columns 8 .. 17: the second image of a pair
                const int iy = g_iy0 + ppr, ix = g_ix0 + pc - 8 * second;
                const int dead = ((((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | (C::XPIX - 1 - pp) | (g.N - 1 - g_n - second)) >> 31) & OOB) | g_dead;
                const int off = (((((g_n + second) * g.IH + iy) * g.IW + ix) * xcs + xoff + xq * 4) * 4) | dead;
                xraw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            }
    };
    auto gload = [&](int s_) {
        gstage(s_);
#pragma unroll
        for (int it = 0; it < ND + NX; ++it) gload_item(it);
    };
    // LDS byte offset (within an x plane) of this thread's x item j.  Stride 1: affine in j.  Stride 2: recomputed at the store (a few
    // integer instructions per item and stage) rather than held in NX registers next to 144 accumulators.
    const int x_in = PX ? (xq >> 3) * 64 + ((xq >> 2) & 1) * XPLANE + (xq & 3) * 16 : xq * 8;
    const int x_lds0 = xp0 * XPITCH + x_in;
    auto x_lds = [&](int j) -> int {
        if constexpr (S == 1) return x_lds0 + j * C::XPP * XPITCH;
        const int pp = xp0 + C::XPP * j;
        const int ppr = (pp * 1986) >> 16, ppc = pp - ppr * C::PW;
        const int sl = ((ppr & 1) ? 2 * (HR + 1) * C::SUBW + (ppc & 1) * HR * C::SUBW : (ppc & 1) * (HR + 1) * C::SUBW) + (ppr >> 1) * C::SUBW + (ppc >> 1);
        return sl * XPITCH + x_in;
    };
    // split + store of one staged float4 in two halves (unit 0: channels 0, 1; unit 1: channels 2, 3 + the two 8-byte LDS stores), so the
    // work can be dealt out between the taps of a k-step, a few instructions at a time
    unsigned hp1, hp2;
    auto put_unit = [&](unsigned char* d, int plane, const u32x4& raw, float Sc, float L, int unit, auto P16) {
        if constexpr (decltype(P16)::value) {                 // a piece: copied as it is (d already points into its plane)
            if (unit == 1) *reinterpret_cast<u32x4*>(d) = raw;
            return;
        }
        const f32x4 v = __builtin_bit_cast(f32x4, raw);
        if (unit == 0) {
            split2_pair(v[0], v[1], Sc, L, hp1, hp2);
        } else {
            unsigned b1, b2;
            split2_pair(v[2], v[3], Sc, L, b1, b2);
            const u32x2 p1 = {hp1, b1}, p2 = {hp2, b2};
            *reinterpret_cast<u32x2*>(d) = p1;
            *reinterpret_cast<u32x2*>(d + plane) = p2;
        }
    };
    // staging unit U of a stage (compile-time): item U >> 1 (dy items first, then x items), half U & 1
    auto lstore_unit = [&](int buf, int U) {
        unsigned char* base = smem_p + buf * STAGE;
        const int it = U >> 1, half = U & 1;
#pragma unroll
        for (int j = 0; j < ND; ++j)
            if (it == j) put_unit(base + d_lds + j * PJ * WP_DROW, DPLANE, draw[j], dscale, dlim, half, std::integral_constant<bool, PD>{});
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if (it == ND + j && (C::XPP * (j + 1) <= C::XPIX || xp0 + C::XPP * j < C::XPIX)) put_unit(base + 2 * DPLANE + (half ? x_lds(j) : 0), XPLANE, xraw[j], xscale, xlim, half, std::integral_constant<bool, PX>{});
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- fragment addresses (lane parts).  16-lane group grp, lane i in it: channel block m0 = 16 (grp & 1), pixels kb + 4 rd + (i >> 2)
    const int grp = lane >> 4, li = lane & 15;
    const int m0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
    // (+ the rows of this wave's k-steps when the waves of a block split them)
    const int a_lane = (kb + (li >> 2) + wk * 16) * WP_DROW + (BM == 128 ? ((wm ^ (li >> 2)) * 64) : BM == 64 ? ((wm ^ ((li >> 3) & 1)) * 64) : 0) + m0 * 2 + (li & 3) * 8;
    const int b_lane = 2 * DPLANE + (kb + (li >> 2) + wk * C::ROWSLOTS) * XPITCH + wn * 64 + m0 * 2 + (li & 3) * 8;

    auto frag = [&](const unsigned char* p, int rowpitch) -> f16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * rowpitch));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // one k-step (16 pixels = one tile row) of all nine taps.  The B fragments of tap t + 1 are fetched before the MFMAs of tap t
    // (explicit two-deep pipeline).
    auto bfrag = [&](const unsigned char* Sb, int r, int t, f16x8 (&bb)[2]) {
        const unsigned char* bp = Sb + b_lane + C::tap_slot(r, t / 3, t % 3) * XPITCH;
        bb[0] = frag(bp, XPITCH);
        bb[1] = frag(bp + XPLANE, XPITCH);
    };
    // A wave's three MFMAs of a tap accumulate into the same registers: they must issue back to back (the accumulator is forwarded;
    // ONE instruction between them costs ~43 cycles, MI355X_MICROARCH.md), while instructions between MFMAs on different accumulators
    // hide behind the running MFMA.  So the fragment reads of the next tap and a slice of the staging work (`fill`) sit BETWEEN the
    // taps, fenced by scheduling barriers.
    auto kstep = [&](const unsigned char* Sb, int r, auto&& fill) {
        f16x8 af[2], bq[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) af[p] = frag(Sb + a_lane + p * DPLANE + r * 16 * WP_DROW, WP_DROW);
        bfrag(Sb, r, 0, bq[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) bfrag(Sb, r, t + 1, bq[(t + 1) & 1]);
            fill(t);
            __builtin_amdgcn_sched_barrier(0);
            // small partial products first (dy1 x2, dy2 x1, then dy1 x1).  The FIRST MFMA takes the fragment that was read LAST
            // (LDS returns in order), so the single wait sits in front of the triple, not inside it
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bq[t & 1][1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bq[t & 1][0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bq[t & 1][0], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Pipeline.  LDS holds two stages: stage s is multiplied while stage s + 1 is split / stored into the other buffer, one staging unit per
    // gap between taps.  The registers that carried an item of stage s + 1 are re-loaded with the same item of stage s + 2 THE MOMENT it
    // has been stored, so every global load has a full stage (and a barrier) to land before its turn comes round again -- issuing all
    // loads of a stage together at its end left them one barrier of slack and cost a global-memory latency per stage.
    if (nst > 0) {
        gload(0);
#pragma unroll
        for (int U = 0; U < 2 * (ND + NX); ++U) lstore_unit(0, U);
        gload(1);
    }
    __syncthreads();
    for (int s = 0; s + 1 < nst; ++s) {
        const unsigned char* Sb = smem_p + (s & 1) * STAGE;
        WG_STAMP(s, 0);
        gstage(s + 2);
#pragma unroll
        for (int r = 0; r < C::KSW; ++r)
            kstep(Sb, r * WK, [&](int t) {
                // the UPK staging units of this k-step, dealt over the nine gaps in order; a finished item is re-loaded at once
#pragma unroll
                for (int u = 0; u < C::UPK; ++u)
                    if ((u * 9) / C::UPK == t) {
                        const int U = r * C::UPK + u;
#if !(defined(VIAI_WG_ABL) && (VIAI_WG_ABL & 1))
                        lstore_unit((s & 1) ^ 1, U);
#endif
#if !(defined(VIAI_WG_ABL) && (VIAI_WG_ABL & 2))
                        if (U & 1) gload_item(U >> 1);
#endif
                    }
            });
        WG_STAMP(s, 1);
        __syncthreads();
        WG_STAMP(s, 2);
    }
    if (nst > 0) {                                  // last stage: nothing left to stage
        const unsigned char* Sb = smem_p + ((nst - 1) & 1) * STAGE;
#pragma unroll
        for (int r = 0; r < C::KSW; ++r) kstep(Sb, r * WK, [&](int) {});
    }

    // ---- waves that split the k-steps of a stage hold partial sums of the SAME tile: add them up through LDS in a fixed order (wave 1,
    // 2, ... into wave 0) so the block writes ONE slab (the slab reduce of the narrow instance read 4x the bytes otherwise)
    if constexpr (WK > 1) {
        static_assert(C::WMC * C::WN == 1 && 9 * 16 * 64 * 4 <= C::LDS, "one tile per block, accumulators fit the stage buffers");
        float* red = reinterpret_cast<float*>(smem_p);
        __syncthreads();                                   // every wave is done with the last stage
        for (int w = 1; w < WK; ++w) {
            if (wk == w) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[(t * 16 + e) * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] += red[(t * 16 + e) * 64 + lane];
            }
            __syncthreads();
        }
        if (wk != 0) return;
    }
    // ---- epilogue: G slab [z][slot][co][ci]
    const float inv = 1.0f / (dscale * xscale);
    const int half = lane >> 5, col = lane & 31;
    const int ci = ci0 + wn * 32 + col;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* dst = a.ws + ((size_t)z * g.wtaps + slots.s[t]) * a.Cout * Cin;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int co = co0 + wm * 32 + row;
            dst[(size_t)co * Cin + ci] = acc[t][e] * inv;
        }
    }
}

static bool window9(const ConvGeom& g, int* y0, int* x0, WgPatchSlots* sl) {
    if (g.ntaps != 9) return false;
    int yy = g.dy[0], xx = g.dx[0];
    for (int t = 1; t < 9; ++t) { yy = g.dy[t] < yy ? g.dy[t] : yy; xx = g.dx[t] < xx ? g.dx[t] : xx; }
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int r = g.dy[t] - yy, c = g.dx[t] - xx;
        if (r > 2 || c > 2) return false;
        seen |= 1u << (r * 3 + c);
        if (sl) sl->s[r * 3 + c] = g.ws[t];
    }
    if (y0) *y0 = yy;
    if (x0) *x0 = xx;
    return seen == 0x1ffu;
}

// which instance takes a layer: 0 none, 1 = <1,128,64,4,1>, 2 = <2,128,32,2,1>, 3 = <1,32,32,4,4>, 4 = <1,64,64,2,1> (round 3: the 64-channel stride-1
// layers -- ResNet-18 layer1 on 56 x 56 maps, G.convblock3 -- ran on the narrow instance, whose four waves stage a 32-channel dy row and a
// 32-channel x patch for ONE 32 x 32 tile; four tiles per block halve the staging per MFMA)
static int pick(const ConvGeom& g, int Cout, int C1, int C2) {
    if (g.run || g.ly != 1 || g.lx != 1 || g.my != g.mx || (g.my != 1 && g.my != 2)) return 0;
    if (wgrad_patch_tiles(g) < 64) return 0;
    // partial tiles multiply zeros: refuse maps that would waste more than ~2/3 of the MFMAs (7 x 7: two images per tile, 98 of 128 pixels; 4 x 4: not taken)
    if ((long)g.N * g.OH * g.OW * 3 < wgrad_patch_tiles(g) * WP_TH * WP_TW) return 0;
    if (!window9(g, nullptr, nullptr, nullptr)) return 0;
    if (g.my == 2) return (Cout % 128 == 0 && C1 % 32 == 0 && C2 % 32 == 0 && C1 >= 32) ? 2 : 0;
    if (Cout % 128 == 0 && C1 % 64 == 0 && C2 % 64 == 0 && C1 >= 64) return 1;
    if (Cout % 64 == 0 && C1 % 64 == 0 && C2 % 64 == 0 && C1 >= 64) return 4;
    if (Cout % 32 == 0 && C1 % 32 == 0 && C2 % 32 == 0 && C1 >= 32) return 3;
    return 0;
}
static int bm_of(int cfg) { return cfg == 3 ? 32 : cfg == 4 ? 64 : 128; }
static int bn_of(int cfg) { return (cfg == 1 || cfg == 4) ? 64 : 32; }
static int wk_of(int cfg) { (void)cfg; return 1; }      // slabs per block (the k-splitting waves of the narrow instance reduce in the block)

template <int S, int BM, int BN, int HR, int WK, bool PD, bool PX>
static int launch_patch_p(WgradArgs& a, int y0, int x0, const WgPatchSlots& sl, hipStream_t st) {
    using C = WgCfg<S, BM, BN, HR, WK>;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_patch_f16_kernel<S, BM, BN, HR, WK, PD, PX>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS); attr_done = true; }
    viai_tag_kernel(S == 2 ? "wgrad_patch_s2_f16x2" : BM == 32 ? "wgrad_patch_narrow_f16x2" : BM == 64 ? "wgrad_patch64_f16x2" : "wgrad_patch_f16x2");
    VIAI_LAUNCH((wgrad_patch_f16_kernel<S, BM, BN, HR, WK, PD, PX>), dim3(a.nblk_co * a.nblk_ci * a.ksplit), dim3(C::NTHR), C::LDS, st, a, y0, x0, sl);
    return viai_launch_status();
}
template <int S, int BM, int BN, int HR, int WK>
static int launch_patch(WgradArgs& a, int y0, int x0, const WgPatchSlots& sl, hipStream_t st) {
    if (a.x_p16 && (a.C2 != 0 || a.xmax == nullptr)) return (int)hipErrorInvalidValue;       // one pre-split source, with its scale
    if (a.dy_p16 && a.x_p16) return launch_patch_p<S, BM, BN, HR, WK, true, true>(a, y0, x0, sl, st);
    if (a.dy_p16) return launch_patch_p<S, BM, BN, HR, WK, true, false>(a, y0, x0, sl, st);
    if (a.x_p16) return launch_patch_p<S, BM, BN, HR, WK, false, true>(a, y0, x0, sl, st);
    return launch_patch_p<S, BM, BN, HR, WK, false, false>(a, y0, x0, sl, st);
}

// number of block-level K slabs
static int block_ksplit(const ConvGeom& g, int Cout, int Cin, int cfg) {
    const long tiles = wgrad_patch_tiles(g);
    const long per = (long)(Cout / bm_of(cfg)) * (Cin / bn_of(cfg));
    constexpr long blk1 = 192, blk2 = 192, blk3 = 128, blk4 = 192;      // stride-1 / stride-2 / narrow / 64 x 64 instance: measured optimum of the three-stream step (swept in rounds 3 and 4, see below)
    long ks = (cfg == 2 ? blk2 : cfg == 3 ? blk3 : cfg == 4 ? blk4 : blk1) / per;
    // the sub-CU grids above suit the audio step, whose weight gradients are short and share the chip with the main chain; a layer with
    // hundreds of tiles per block (the ResNet branch on 1024 frames) is worth every CU
    // (ResNet layer2 - 4 sit at 85 tiles per block and keep the 192-block grid: alone they would be a quarter faster on every CU -- tools/probes/wgrad_patch_bench.hip,
    // 947 us, in-loop MFMA use 0.79 -- but inside the vision-infused step, beside the main chain, 256 blocks measured 107.1 ms against 106.5)
    if (ks >= 1 && tiles / ks > 128) ks = 256 / per;
    if (ks > tiles / 4) ks = tiles / 4;
    if (ks < 1) ks = 1;
    const long tps = (tiles + ks - 1) / ks;                  // tiles per slab
    return (int)((tiles + tps - 1) / tps);                   // no empty slab
}

}  // namespace

// 3 x 3 layers with the full window, stride 1 or 2 (plain conv), output extent a multiple of the 8 x 16 tile, enough tiles to give every
// block a K loop worth its prologue / epilogue, and channel counts one of the instances tiles (a Cin tile must not straddle the two
// concatenated sources).  `shape_ok` is the pure shape predicate (workspace sizing); `ok` adds the switches.
//
// Same-box A/B of the full three-stream step (the weight gradients run beside the main backward chain, so a kernel that is faster alone
// but takes a CU's whole LDS / register file can still lose): neither patch kernel 8.69 ms, stride-1 instance only 8.54, both with two
// stride-2 blocks per CU 8.45, both with ONE stride-2 block per CU (256 blocks) 8.37; + the narrow instance 8.17; and with the grids
// cut below one block per CU (192 / 192 / 128 blocks: CUs left to the main chain, fewer slabs to reduce) 8.00 -- the defaults
// VIAI_WGRAD_PATCH_S2=0 switches the stride-2 instance off.
bool viai_wgrad_patch_shape_ok(const ConvGeom& g, int Cout, int C1, int C2) { return pick(g, Cout, C1, C2) != 0; }

bool viai_wgrad_patch_ok(const ConvGeom& g, int Cout, int C1, int C2) {
    const int cfg = pick(g, Cout, C1, C2);
    if (cfg == 2) { const char* e = getenv("VIAI_WGRAD_PATCH_S2"); if (e && !atoi(e)) return false; }         // read per call: tests/test_fullsize_gpu.py switches it (the round-1 kernel stays covered)
    return cfg != 0;
}

// total number of K slabs the launch writes (block-level slabs x waves that split the k-steps)
int viai_wgrad_patch_ksplit(const ConvGeom& g, int Cout, int C1, int C2) {
    const int cfg = pick(g, Cout, C1, C2);
    if (cfg == 0) return 1;
    return block_ksplit(g, Cout, C1 + C2, cfg) * wk_of(cfg);
}

int viai_wgrad_patch_launch(WgradArgs& a, hipStream_t st) {
    const ConvGeom& g = a.g;
    const int Cin = a.C1 + a.C2;
    if (a.amax == nullptr || !viai_wgrad_patch_ok(g, a.Cout, a.C1, a.C2)) return (int)hipErrorInvalidValue;
    const int cfg = pick(g, a.Cout, a.C1, a.C2);
    int y0, x0; WgPatchSlots sl;
    window9(g, &y0, &x0, &sl);
    const long tiles = wgrad_patch_tiles(g);
    a.ksplit = block_ksplit(g, a.Cout, Cin, cfg);
    a.chunks_per_split = (int)((tiles + a.ksplit - 1) / a.ksplit);
    a.nblk_co = a.Cout / bm_of(cfg);
    a.nblk_ci = Cin / bn_of(cfg);
    if (cfg == 2) return launch_patch<2, 128, 32, 2, 1>(a, y0, x0, sl, st);
    if (cfg == 3) return launch_patch<1, 32, 32, 4, 4>(a, y0, x0, sl, st);
    if (cfg == 4) return launch_patch<1, 64, 64, 2, 1>(a, y0, x0, sl, st);
    return launch_patch<1, 128, 64, 4, 1>(a, y0, x0, sl, st);
}
