// Patch-staged weight gradient for the stride-1 3 x 3 layers (f16x2 split MFMA, gfx950).
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
// Split-K over tile ranges, deterministic slab reduce (viai_wgrad_reduce) as before.
// Reference call sites: the autograd backward of nn.Conv2d(…, 3, 1, 1) / nn.ConvTranspose2d(…, 3, 1, 1)
// (networks/Discriminator_Networks.py:30 conv3, networks/New_Inpainting_Networks.py:24 TransConvBlock).
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"

namespace {

constexpr int WP_BM = 128, WP_BN = 64;              // Cout x Cin tile of a block
constexpr int WP_TW = 16, WP_TH = 8;                // output pixel tile
constexpr int WP_PC = WP_TW + 2;                    // x patch columns
constexpr int WP_DROW = WP_BM * 2;                  // dy LDS row: 128 fp16 = 256 B, 64-byte groups XOR-swizzled by (pixel & 3)
constexpr int WP_XPITCH = WP_BN * 2 + 64;           // x LDS row: 64 fp16 + 64 B pad = 192 B (four consecutive rows tile the 256-B bank row)

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgPatchSlots { int s[9]; };

template <int HR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_patch_f16_kernel(const WgradArgs a, int y0, int x0, WgPatchSlots slots) {
    constexpr int DPIX = HR * WP_TW;                         // dy pixels per stage
    constexpr int XPIX = (HR + 2) * WP_PC;                   // x patch pixels per stage
    constexpr int DPLANE = DPIX * WP_DROW, XPLANE = XPIX * WP_XPITCH;
    constexpr int STAGE = 2 * DPLANE + 2 * XPLANE;
    constexpr int NTHR = 512;
    constexpr int ND = DPIX * (WP_BM / 4) / NTHR;            // dy float4 items per thread and stage (= HR: one tile row each)
    constexpr int NX = (XPIX * (WP_BN / 4) + NTHR - 1) / NTHR;   // x float4 items per thread and stage (last one partial)
    constexpr int SPT = WP_TH / HR;                          // stages per tile
    static_assert(ND == HR && NX <= HR, "staging parts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];      // [2 stages]{dy plane 0, dy plane 1, x plane 0, x plane 1}

    const ConvGeom& g = a.g;
    const float dscale = f16_scale_from_amax(a.amax), dlim = f16_clamp_for_scale(dscale);
    const float xscale = F16_ASCALE, xlim = 65504.f / F16_ASCALE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // 32 output channels x 32 input channels per wave

    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_slab = a.nblk_ci * a.nblk_co;
    const int z = b / per_slab; b -= z * per_slab;
    const int bci = b % a.nblk_ci, bco = b / a.nblk_ci;
    const int co0 = bco * WP_BM, ci0 = bci * WP_BN;
    const int Cin = a.C1 + a.C2;
    const bool first = ci0 < a.C1;
    const int xcs = first ? a.C1 : a.C2;
    const int xoff = first ? ci0 : ci0 - a.C1;
    constexpr int OOB = 0x7fffffff;
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? a.x : a.x2), 0, (int)(in_pixels * xcs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)a.M * a.Cout * 4), 0x00020000);

    const int tiles_x = g.OW / WP_TW, tiles_y = g.OH / WP_TH;
    const int ntiles = g.N * tiles_y * tiles_x;
    const int tile0 = z * a.chunks_per_split;                 // chunks_per_split = tiles per K slab here
    int tile1 = tile0 + a.chunks_per_split;
    if (tile1 > ntiles) tile1 = ntiles;
    const int nst = (tile1 - tile0) * SPT;

    // ---- staging maps (thread -> items; everything but the stage origin is fixed per thread)
    // dy: item j = pixel (column tid >> 5 of tile row j), channel quad q = tid & 31
    const int dq = tid & 31, dp0 = tid >> 5;
    const int d_goff = (dp0 * a.Cout + co0 + dq * 4) * 4;                                   // + stage origin + j rows
    const int d_lds = dp0 * WP_DROW + (((dq >> 3) ^ (dp0 & 3)) * 64) + (dq & 7) * 8;         // + j * 16 * WP_DROW
    // x: item j = patch pixel (tid >> 4) + 32 j, channel quad q = tid & 15
    const int xq = tid & 15, xp0 = tid >> 4;
    const int x_lds = xp0 * WP_XPITCH + xq * 8;                                              // + j * 32 * WP_XPITCH

    u32x4 draw[ND], xraw[NX];
    auto gload = [&](int s_) {
        const int s = __builtin_amdgcn_readfirstlane(s_);
        const int tile = tile0 + s / SPT, h = s % SPT;
        const int tx = tile % tiles_x; const int r_ = tile / tiles_x; const int ty = r_ % tiles_y, n = r_ / tiles_y;
        const int oy0 = ty * WP_TH + h * HR, ox0 = tx * WP_TW;
        const int dbase = ((n * g.OH + oy0) * g.OW + ox0) * a.Cout * 4;
#pragma unroll
        for (int j = 0; j < ND; ++j)
            draw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, d_goff + j * g.OW * a.Cout * 4, dbase, 0);
        const int iy0 = oy0 + y0, ix0 = ox0 + x0;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int pp = xp0 + 32 * j, ppr = (pp * 3641) >> 16;          // pp / 18 for pp < 128
            const int iy = iy0 + ppr, ix = ix0 + pp - ppr * WP_PC;
            const int dead = (((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | (XPIX - 1 - pp)) >> 31) & OOB;
            const int off = ((((n * g.IH + iy) * g.IW + ix) * xcs + xoff + xq * 4) * 4) | dead;
            xraw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        }
    };
    // split + store of one staged float4 in two halves (unit 0: channels 0, 1; unit 1: channels 2, 3 + the two 8-byte LDS stores), so the
    // work can be dealt out between the taps of a k-step, a few instructions at a time
    unsigned hp1, hp2;
    auto put_unit = [&](unsigned char* d, int plane, const u32x4& raw, float S, float L, int unit) {
        const f32x4 v = __builtin_bit_cast(f32x4, raw);
        if (unit == 0) {
            split2_pair(v[0], v[1], S, L, hp1, hp2);
        } else {
            unsigned b1, b2;
            split2_pair(v[2], v[3], S, L, b1, b2);
            const u32x2 p1 = {hp1, b1}, p2 = {hp2, b2};
            *reinterpret_cast<u32x2*>(d) = p1;
            *reinterpret_cast<u32x2*>(d + plane) = p2;
        }
    };
    // part `part` of a stage = item `part` of each operand; unit 0..1 = the dy item, 2..3 = the x item
    auto lstore_unit = [&](int buf, int part, int unit) {
        unsigned char* base = smem_p + buf * STAGE;
        if (unit < 2) {
#pragma unroll
            for (int j = 0; j < ND; ++j)
                if (j == part) put_unit(base + d_lds + j * 16 * WP_DROW, DPLANE, draw[j], dscale, dlim, unit);
        } else {
#pragma unroll
            for (int j = 0; j < NX; ++j)
                if (j == part && xp0 + 32 * j < XPIX) put_unit(base + 2 * DPLANE + x_lds + j * 32 * WP_XPITCH, XPLANE, xraw[j], xscale, xlim, unit - 2);
        }
    };
    auto lstore = [&](int buf, int part) {
#pragma unroll
        for (int u = 0; u < 4; ++u) lstore_unit(buf, part, u);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- fragment addresses (lane parts).  16-lane group grp, lane i in it: channel block m0 = 16 (grp & 1), pixels kb + 4 rd + (i >> 2)
    const int grp = lane >> 4, li = lane & 15;
    const int m0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const int a_lane = (kb + (li >> 2)) * WP_DROW + ((wm ^ (li >> 2)) * 64) + m0 * 2 + (li & 3) * 8;
    const int b_lane = 2 * DPLANE + (kb + (li >> 2)) * WP_XPITCH + wn * 64 + m0 * 2 + (li & 3) * 8;

    auto frag = [&](const unsigned char* p, int rowpitch) -> f16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * rowpitch));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // one k-step (16 pixels = one tile row) of all nine taps.  The B fragments of tap t + 1 are fetched before the MFMAs of tap t
    // (explicit two-deep pipeline); a scheduling barrier per tap keeps the compiler from hoisting all 36 reads of a k-step to its
    // top, which costs 144 registers next to the 288 accumulators and spills.
    auto bfrag = [&](const unsigned char* S, int r, int t, f16x8 (&b)[2]) {
        const unsigned char* bp = S + b_lane + ((r + t / 3) * WP_PC + (t % 3)) * WP_XPITCH;
        b[0] = frag(bp, WP_XPITCH);
        b[1] = frag(bp + XPLANE, WP_XPITCH);
    };
    // A wave's three MFMAs of a tap accumulate into the same registers: they must issue back to back (the accumulator is forwarded;
    // ONE instruction between them costs ~43 cycles, MI355X_MICROARCH.md), while instructions between MFMAs on different accumulators
    // hide behind the running MFMA.  So the fragment reads of the next tap and a slice of the staging work (`fill`) sit BETWEEN the
    // taps, fenced by scheduling barriers.
    auto kstep = [&](const unsigned char* S, int r, auto&& fill) {
        f16x8 af[2], bq[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) af[p] = frag(S + a_lane + p * DPLANE + r * 16 * WP_DROW, WP_DROW);
        bfrag(S, r, 0, bq[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) bfrag(S, r, t + 1, bq[(t + 1) & 1]);
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

    // Loads run a whole stage ahead of their LDS stores: the registers of stage s + 1 are split / stored during the k-steps of stage s,
    // and re-filled with stage s + 2 right after the last store, so a load has a barrier and most of a stage to land.
    if (nst > 0) {
        gload(0);
#pragma unroll
        for (int part = 0; part < HR; ++part) lstore(0, part);
        if (nst > 1) gload(1);
    }
    __syncthreads();
    for (int s = 0; s + 1 < nst; ++s) {
        const unsigned char* S = smem_p + (s & 1) * STAGE;
#pragma unroll
        for (int r = 0; r < HR; ++r)
            kstep(S, r, [&](int t) {
                if ((t & 1) && t < 8) lstore_unit((s & 1) ^ 1, r, t >> 1);       // units 0..3 behind taps 1, 3, 5, 7
            });
        if (s + 2 < nst) gload(s + 2);
        __syncthreads();
    }
    if (nst > 0) {                                  // last stage: nothing left to stage
        const unsigned char* S = smem_p + ((nst - 1) & 1) * STAGE;
#pragma unroll
        for (int r = 0; r < HR; ++r) kstep(S, r, [&](int) {});
    }

    // ---- epilogue: G slab [z][slot][co][ci]
    const float inv = 1.0f / (dscale * F16_ASCALE);
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

constexpr int WP_HR = 4;

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

}  // namespace

// stride-1 3 x 3 layers with Cout % 128 == 0, Cin % 64 == 0 (a tile must not straddle the two concatenated sources), output
// extent a multiple of the 8 x 16 tile and enough tiles to give every block a K loop worth its prologue / epilogue
bool viai_wgrad_patch_ok(const ConvGeom& g, int Cout, int C1, int C2) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VIAI_WGRAD_PATCH"); on = e ? atoi(e) : 1; }
    if (!on || g.run || g.my != 1 || g.mx != 1 || g.ly != 1 || g.lx != 1) return false;
    if (Cout % WP_BM != 0 || C1 % WP_BN != 0 || C2 % WP_BN != 0 || C1 < WP_BN) return false;
    if (g.OH % WP_TH != 0 || g.OW % WP_TW != 0) return false;
    if ((long)g.N * (g.OH / WP_TH) * (g.OW / WP_TW) < 64) return false;
    return window9(g, nullptr, nullptr, nullptr);
}

// K slabs: one round of one block per CU, at least four tiles per block
int viai_wgrad_patch_ksplit(const ConvGeom& g, int Cout, int Cin) {
    const long tiles = (long)g.N * (g.OH / WP_TH) * (g.OW / WP_TW);
    const long per = (long)(Cout / WP_BM) * (Cin / WP_BN);
    long ks = 256 / per;
    if (ks > tiles / 4) ks = tiles / 4;
    if (ks < 1) ks = 1;
    const long tps = (tiles + ks - 1) / ks;                  // tiles per slab
    return (int)((tiles + tps - 1) / tps);                   // no empty slab
}

int viai_wgrad_patch_launch(WgradArgs& a, hipStream_t st) {
    const ConvGeom& g = a.g;
    const int Cin = a.C1 + a.C2;
    if (a.amax == nullptr || !viai_wgrad_patch_ok(g, a.Cout, a.C1, a.C2)) return (int)hipErrorInvalidValue;
    int y0, x0; WgPatchSlots sl;
    window9(g, &y0, &x0, &sl);
    const long tiles = (long)g.N * (g.OH / WP_TH) * (g.OW / WP_TW);
    a.ksplit = viai_wgrad_patch_ksplit(g, a.Cout, Cin);
    a.chunks_per_split = (int)((tiles + a.ksplit - 1) / a.ksplit);
    a.nblk_co = a.Cout / WP_BM;
    a.nblk_ci = Cin / WP_BN;
    constexpr int lds = 2 * (2 * WP_HR * WP_TW * WP_DROW + 2 * (WP_HR + 2) * WP_PC * WP_XPITCH);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_patch_f16_kernel<WP_HR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
    VIAI_LAUNCH((wgrad_patch_f16_kernel<WP_HR>), dim3(a.nblk_co * a.nblk_ci * a.ksplit), dim3(512), lds, st, a, y0, x0, sl);
    return viai_launch_status();
}
