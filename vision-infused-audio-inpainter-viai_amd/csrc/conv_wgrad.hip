// Weight gradient of the conv family on fp32 MFMA (gfx950).
//
//   G[t][co][ci] = sum over output pixels p of  dy[p][co] * x[p_t][ci]
// (p_t = the input pixel tap t reads for output pixel p; forward geometry).
// GEMM view per tap: M = Cout, N = Cin, K = output pixels (huge) -> split-K over
// the grid; each block writes a partial slab, wgrad_reduce sums the slabs in a
// fixed order (deterministic) straight into the torch weight layout.
//
// Both operands are "pixel-major" in memory (NHWC rows), which is exactly the
// k-major image v_mfma_f32_32x32x2_f32 wants: lane (i = l&31, k = l>>5) reads
// LDS[k][i] -- 32 consecutive floats per half-wave, conflict-free ds_read_b32.
// Reference call sites: the autograd backward of every nn.Conv2d /
// nn.ConvTranspose2d listed in include/viai_hip.h.
#include "viai_common.h"
#include "viai_internal.h"
#include <cstdlib>

namespace {

constexpr int BKP = 32;   // pixels per staged chunk

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const WgradArgs a) {
    constexpr int BM = 32 * TM * WM;       // Cout tile
    constexpr int BN = 32 * TN * WN;       // Cin tile
    constexpr int WK = 4 / (WM * WN);      // waves splitting the pixel chunk
    constexpr int QA = BM / 4, QB = BN / 4;               // float4 per staged row
    constexpr int NA = (BKP * QA) / 256, NB = (BKP * QB) / 256;   // float4 per thread
    static_assert((BKP * QA) % 256 == 0 && (BKP * QB) % 256 == 0, "staging");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ds = smem;                       // [2][BKP][BM]
    float* Xs = smem + 2 * BKP * BM;        // [2][BKP][BN]

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN);
    const int wmn = wave % (WM * WN);
    const int wm = wmn / WN, wn = wmn % WN;

    // 1-D grid, XCD-aware: consecutive logical ids (all taps and channel tiles of one pixel slab) share an XCD's L2,
    // so the slab of dy / x is fetched into ONE L2 instead of eight
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_slab = a.nblk_ci * a.nblk_co * a.g.ntaps;
    const int z = b / per_slab; b -= z * per_slab;
    const int bci = b % a.nblk_ci; b /= a.nblk_ci;
    const int bco = b % a.nblk_co; b /= a.nblk_co;
    const int t = b;                        // tap index of this block
    const int co0 = bco * BM, ci0 = bci * BN;
    const int Cin = a.C1 + a.C2;

    const bool first = ci0 < a.C1;
    const int xcs = g.run ? 4 : (first ? a.C1 : a.C2);
    const int xoff = g.run ? 0 : (first ? ci0 : ci0 - a.C1);
    const int dyt = g.dy[t], dxt = g.dx[t];
    constexpr int OOB = 0x7fffffff;
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(first ? a.x : a.x2), 0, (int)(in_pixels * xcs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)a.M * a.Cout * 4), 0x00020000);

    const int chunk0 = z * a.chunks_per_split;
    const int nchunks_total = (a.M + BKP - 1) / BKP;
    int chunk1 = chunk0 + a.chunks_per_split;
    if (chunk1 > nchunks_total) chunk1 = nchunks_total;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // per-thread staged rows: pixel coordinates are decoded once and advanced by BKP pixels per chunk
    int dcol[NA];                          // byte offset of the dy column quad (OOB if past Cout)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int idx = tid + 256 * j;
        int co = co0 + (idx % QA) * 4;
        dcol[j] = co < a.Cout ? co * 4 : OOB;
    }
    int xo[NB], xy[NB], xn[NB], xcol[NB];  // ox, oy, n of the staged x rows; channel byte offset
    const int step_x = BKP % g.OW, step_y = BKP / g.OW;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int idx = tid + 256 * j;
        int row = idx / QB, c4 = idx % QB;
        int p = chunk0 * BKP + row;
        xo[j] = p % g.OW; int r = p / g.OW; xy[j] = r % g.OH; xn[j] = r / g.OH;
        xcol[j] = g.run ? c4 : ((ci0 + c4 * 4 < Cin) ? (xoff + c4 * 4) * 4 : OOB);   // run mode: pixel offset of the quad
    }

    u32x4 dreg[NA], xreg[NB];
    auto gload = [&](int chunk) {
        const int p0 = chunk * BKP;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int idx = tid + 256 * j;
            int p = p0 + idx / QA;
            int off = (p < a.M && dcol[j] != OOB) ? p * a.Cout * 4 + dcol[j] : OOB;
            dreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int iy = xy[j] * g.my + dyt, ix = xo[j] * g.mx + dxt + (g.run ? xcol[j] : 0);
            bool ok = xn[j] < g.N && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW && xcol[j] != OOB;
            int off = ok ? ((xn[j] * g.IH + iy) * g.IW + ix) * xcs * 4 + (g.run ? 0 : xcol[j]) : OOB;
            xreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            // advance this row by BKP output pixels
            int nx = xo[j] + step_x;
            int carry = nx >= g.OW ? 1 : 0;
            xo[j] = nx - carry * g.OW;
            int ny = xy[j] + step_y + carry;
            while (ny >= g.OH) { ny -= g.OH; ++xn[j]; }
            xy[j] = ny;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int idx = tid + 256 * j;
            *reinterpret_cast<u32x4*>(Ds + buf * BKP * BM + idx * 4) = dreg[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int idx = tid + 256 * j;
            *reinterpret_cast<u32x4*>(Xs + buf * BKP * BN + idx * 4) = xreg[j];
        }
    };

    if (chunk0 < chunk1) {
        gload(chunk0);
        lstore(0);
    }
    __syncthreads();
    const int half = lane >> 5, col = lane & 31;
    for (int c = chunk0; c < chunk1; ++c) {
        const int cur = (c - chunk0) & 1;
        const bool more = (c + 1 < chunk1);
        if (more) gload(c + 1);
        const float* Db = Ds + cur * BKP * BM + wm * TM * 32 + col;
        const float* Xb = Xs + cur * BKP * BN + wn * TN * 32 + col;
#pragma unroll
        for (int s = 0; s < BKP / 2 / WK; ++s) {
            const int p = 2 * (s * WK + wk) + half;
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = Db[p * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Xb[p * BN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    // cross-wave reduction of the K-split (WK > 1), through LDS
    if (WK > 1) {
        float* red = smem;      // [WK-1][WM*WN][TM*TN*16][64]
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        red[(((wk - 1) * (WM * WN) + wmn) * (TM * TN * 16) + (i * TN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int w = 1; w < WK; ++w)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            acc[i][j][e] += red[(((w - 1) * (WM * WN) + wmn) * (TM * TN * 16) + (i * TN + j) * 16 + e) * 64 + lane];
        }
    }
    if (wk == 0) {
        float* dst = a.ws + ((size_t)z * g.wtaps + g.ws[t]) * a.Cout * Cin;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int ci = ci0 + (wn * TN + j) * 32 + col;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int row = (e & 3) + 8 * (e >> 2) + 4 * half;
                    int co = co0 + (wm * TM + i) * 32 + row;
                    if (co < a.Cout && ci < Cin) dst[(size_t)co * Cin + ci] = acc[i][j][e];
                }
            }
    }
}

// All-taps kernel for the <= 32 x <= 32 channel stride-1 layers (3 x 3 window, output width a multiple of 32).  The per-tap
// kernel above re-reads the dy and x tiles of a pixel chunk once per tap (nine blocks, 72 KB per 32-pixel chunk through L2
// and the vector-memory path: 57 TFLOP/s).  Here a block owns a slab of 32-pixel row chunks and ALL taps: per chunk it stages
// the dy tile (32 px x 32 co) and the 3 x 34 pixel x patch once (17 KB), and tap t's B operand is the same patch read at a
// row / column offset -- v_mfma_f32_32x32x2_f32 takes ONE value per lane, so a shifted read is just another LDS address.
// The four waves split the chunk's 16 k-steps; nine accumulator tiles per wave (144 registers).
struct Wg32Taps { int off[9]; int slot[9]; int nt; int y0, x0; };

// FULL: all nine taps present (no per-tap branch in the K loop)
template <bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad32_halo_kernel(const WgradArgs a, const Wg32Taps tp) {
    constexpr int PW = 34, PR = 3;                   // patch: 3 rows x 34 pixels
    constexpr int DS_F = 32 * 32, XS_F = PR * PW * 32, STAGE_F = DS_F + XS_F;
    constexpr int NX = (PR * PW * 8 + 255) / 256;    // x float4 per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];           // [2][Ds | Xs], reused by the final reduction
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = blockIdx.x;
    const int Cin = a.C1;
    constexpr int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)g.N * g.IH * g.IW * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)a.M * a.Cout * 4), 0x00020000);
    const int chunk0 = z * a.chunks_per_split;
    const int nchunks_total = a.M / 32;
    int chunk1 = chunk0 + a.chunks_per_split;
    if (chunk1 > nchunks_total) chunk1 = nchunks_total;
    const int cpr = g.OW / 32;                       // chunks per output row

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // staging roles: dy float4 (pixel tid >> 3, channel quad tid & 7); x float4 idx = tid + 256 j -> (row, patch column, quad)
    const int dpx = tid >> 3, dq = tid & 7;
    const int ddead = (dq * 4 < a.Cout) ? 0 : OOB;
    int xr[NX], xc[NX], xq[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int idx = tid + 256 * j;
        xr[j] = idx / (PW * 8); const int rem = idx % (PW * 8); xc[j] = rem >> 3; xq[j] = rem & 7;
        if (idx >= PR * PW * 8 || xq[j] * 4 >= Cin) xr[j] = -1;
    }
    u32x4 dreg, xreg[NX];
    auto gload = [&](int chunk_) {
        const int chunk = __builtin_amdgcn_readfirstlane(chunk_);
        const int cx = chunk % cpr; int r = chunk / cpr; const int oy = r % g.OH, n = r / g.OH;
        const int ox0 = cx * 32;
        dreg = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (((chunk * 32 + dpx) * a.Cout + dq * 4) * 4) | ddead, 0, 0);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int iy = oy + tp.y0 + xr[j], ix = ox0 + tp.x0 + xc[j];
            const int dead = (((g.IH - 1 - iy) | iy | (g.IW - 1 - ix) | ix | xr[j]) >> 31) & OOB;
            xreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ((((n * g.IH + iy) * g.IW + ix) * Cin + xq[j] * 4) * 4) | dead, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
        float* Ds = smem + buf * STAGE_F;
        *reinterpret_cast<u32x4*>(Ds + dpx * 32 + dq * 4) = dreg;
#pragma unroll
        for (int j = 0; j < NX; ++j)
            if (xr[j] >= 0) *reinterpret_cast<u32x4*>(Ds + DS_F + (xr[j] * PW + xc[j]) * 32 + xq[j] * 4) = xreg[j];
    };
    if (chunk0 < chunk1) { gload(chunk0); lstore(0); }
    __syncthreads();
    const int half = lane >> 5, col = lane & 31;
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = __builtin_amdgcn_readfirstlane(tp.off[t < tp.nt ? t : 0]);
    for (int c = chunk0; c < chunk1; ++c) {
        const int cur = (c - chunk0) & 1;
        if (c + 1 < chunk1) gload(c + 1);
        const float* Db = smem + cur * STAGE_F + col;
        const float* Xb = Db + DS_F;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int p = 2 * (s * 4 + wave) + half;            // this lane's pixel of the k-step
            const float af = Db[p * 32];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (FULL || t < tp.nt) {
                    const float bf = Xb[p * 32 + toff[t]];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[t], 0, 0, 0);
                }
        }
        if (c + 1 < chunk1) lstore(cur ^ 1);
        __syncthreads();
    }
    // waves 1..3 hand their partial tiles to wave 0 through LDS, one tap at a time (fixed order)
    float* red = smem;                               // [3][16][64]
    float* dst0 = a.ws + (size_t)z * g.wtaps * a.Cout * Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t < tp.nt) {
            if (wave > 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) red[((wave - 1) * 16 + e) * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (wave == 0) {
                float* dst = dst0 + (size_t)tp.slot[t] * a.Cout * Cin;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[t][e];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += red[(w * 16 + e) * 64 + lane];
                    const int co = (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (co < a.Cout && col < Cin) dst[(size_t)co * Cin + col] = v;
                }
            }
            __syncthreads();
        }
    }
}

template <int TM, int TN, int WM, int WN>
int launch_wgrad(WgradArgs& a, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, WK = 4 / (WM * WN);
    const int Cin = a.C1 + a.C2;
    a.nblk_co = (a.Cout + BM - 1) / BM;
    a.nblk_ci = (Cin + BN - 1) / BN;
    size_t lds = (size_t)2 * BKP * (BM + BN) * sizeof(float);
    size_t red = (size_t)(WK - 1) * (WM * WN) * (TM * TN * 16) * 64 * sizeof(float);
    if (red > lds) lds = red;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<TM, TN, WM, WN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid(a.nblk_co * a.nblk_ci * a.g.ntaps * a.ksplit);
    viai_tag_kernel("wgrad_mfma_f32");
    VIAI_LAUNCH((wgrad_mfma_kernel<TM, TN, WM, WN>), grid, dim3(256), lds, st, a);
    return viai_launch_status();
}

}  // namespace

// layers the all-taps 32-channel kernel takes
bool viai_wgrad32_ok(const ConvGeom& g, int Cout, int C1, int C2) {
    constexpr int on = 1;
    if (!on || C2 != 0 || Cout > 32 || C1 > 32 || Cout % 4 != 0 || C1 % 4 != 0 || g.run) return false;
    if (g.mx != 1 || g.my != 1 || g.ly != 1 || g.lx != 1 || g.SH != g.OH || g.SW != g.OW || g.OW % 32 != 0 || g.ntaps < 1 || g.ntaps > 9) return false;
    int y0 = g.dy[0], y1 = g.dy[0], x0 = g.dx[0], x1 = g.dx[0];
    for (int t = 1; t < g.ntaps; ++t) {
        y0 = g.dy[t] < y0 ? g.dy[t] : y0; y1 = g.dy[t] > y1 ? g.dy[t] : y1;
        x0 = g.dx[t] < x0 ? g.dx[t] : x0; x1 = g.dx[t] > x1 ? g.dx[t] : x1;
    }
    return (y1 - y0) <= 2 && (x1 - x0) <= 2;
}

// split-K for that kernel: one slab per block, ONE block per CU -- the weight gradients trail on a side stream next to the main
// chain, and at 200 registers per lane a second resident block per CU only displaces the main chain's kernels (same-box A/B:
// 512 blocks 9.29 ms/step, 256 blocks 9.35 vs 9.40 without the kernel; standalone it is 85 -> 61 us on 16 x 128 x 128 x 32)
int viai_wgrad32_ksplit(long M) {
    long chunks = M / 32;
    long ks = chunks / 4;                  // >= 4 chunks per block
    constexpr long cap = 256;
    if (ks > cap) ks = cap;
    if (ks < 1) ks = 1;
    return (int)ks;
}

int viai_wgrad32_launch(WgradArgs& a, int ksplit, hipStream_t st) {
    const ConvGeom& g = a.g;
    if (!viai_wgrad32_ok(g, a.Cout, a.C1, a.C2)) return (int)hipErrorInvalidValue;
    Wg32Taps tp{};
    int y0 = g.dy[0], x0 = g.dx[0];
    for (int t = 1; t < g.ntaps; ++t) { y0 = g.dy[t] < y0 ? g.dy[t] : y0; x0 = g.dx[t] < x0 ? g.dx[t] : x0; }
    tp.nt = g.ntaps; tp.y0 = y0; tp.x0 = x0;
    for (int t = 0; t < g.ntaps; ++t) { tp.off[t] = ((g.dy[t] - y0) * 34 + (g.dx[t] - x0)) * 32; tp.slot[t] = g.ws[t]; }
    const long chunks = a.M / 32;
    a.ksplit = ksplit;
    a.chunks_per_split = (int)((chunks + ksplit - 1) / ksplit);
    constexpr int lds = 2 * (32 * 32 + 3 * 34 * 32) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad32_halo_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad32_halo_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    viai_tag_kernel("wgrad32_all_taps_f32");
    if (g.ntaps == 9) VIAI_LAUNCH(wgrad32_halo_kernel<true>, dim3(ksplit), dim3(256), lds, st, a, tp);
    else VIAI_LAUNCH(wgrad32_halo_kernel<false>, dim3(ksplit), dim3(256), lds, st, a, tp);
    return viai_launch_status();
}

static inline int tile_of(int c) { return c > 64 ? 128 : (c > 32 ? 64 : 32); }

int viai_wgrad_pick_ksplit(int Cout, int Cin, int ntaps, long M) {
    int bm = tile_of(Cout), bn = tile_of(Cin);
    long tiles = (long)((Cout + bm - 1) / bm) * ((Cin + bn - 1) / bn) * ntaps;
    long chunks = (M + BKP - 1) / BKP;
    // blocks per launch: one round of 2 blocks/CU x 256 CUs.  The weight gradients run on a side stream next to the
    // main backward chain, so a second round buys nothing and every extra K slab costs reduce traffic (measured:
    // 512 beats 1024 by 1 %, 256 loses 4 %).
    constexpr long target = 512;
    long ks = target / tiles;
    long maxks = chunks / 8; if (maxks < 1) maxks = 1;  // at least 8 chunks (256 pixels) per block
    if (ks > maxks) ks = maxks;
    if (ks > 512) ks = 512;
    if (ks < 1) ks = 1;
    return (int)ks;
}

int viai_wgrad_mfma_launch(WgradArgs& a, int ksplit, hipStream_t st) {
    const int Cin = a.C1 + a.C2;
    if (Cin % 4 != 0 || a.Cout % 4 != 0) return (int)hipErrorInvalidValue;
    long chunks = ((long)a.M + BKP - 1) / BKP;
    a.ksplit = ksplit;
    a.chunks_per_split = (int)((chunks + ksplit - 1) / ksplit);
    int bm = tile_of(a.Cout), bn = tile_of(Cin);
    if (a.C2 > 0) {                       // a Cin tile must not straddle the two concatenated sources
        while (bn > 32 && (a.C1 % bn) != 0) bn >>= 1;
        if ((a.C1 % bn) != 0) return (int)hipErrorInvalidValue;
    }
    if (bm == 128 && bn == 128) return launch_wgrad<2, 2, 2, 2>(a, st);
    if (bm == 128 && bn == 64) return launch_wgrad<2, 1, 2, 2>(a, st);
    if (bm == 128 && bn == 32) return launch_wgrad<2, 1, 2, 1>(a, st);   // 128 x 32, WK = 2
    if (bm == 64 && bn == 128) return launch_wgrad<1, 2, 2, 2>(a, st);
    if (bm == 64 && bn == 64) return launch_wgrad<1, 1, 2, 2>(a, st);
    if (bm == 64 && bn == 32) return launch_wgrad<1, 1, 2, 1>(a, st);    // WK = 2
    if (bm == 32 && bn == 128) return launch_wgrad<1, 2, 1, 2>(a, st);   // WK = 2
    if (bm == 32 && bn == 64) return launch_wgrad<1, 1, 1, 2>(a, st);    // WK = 2
    return launch_wgrad<1, 1, 1, 1>(a, st);                               // 32 x 32, WK = 4
}
