// Weight gradient on the bf16 matrix cores with fp32-grade accuracy (bf16x3 split), gfx950.
//
//   G[t][co][ci] = sum over output pixels p of  dy[p][co] * x[p_t][ci]        (same contract as conv_wgrad.hip)
//
// GEMM per tap: M = Cout, N = Cin, K = pixels (split-K over the grid, deterministic slab reduce afterwards).
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive K values per lane, but both operands are channel-contiguous
// (NHWC) in memory, so the tiles are transposed on their way into LDS: a thread loads a 4-pixel x 4-channel
// block (four coalesced float4 rows), splits every value into three bf16 terms and stores, per channel and
// plane, the four pixels as one 8-byte LDS write into [plane][channel][32 pixels + pad] rows -- the layout
// whose ds_read_b128 is an MFMA operand fragment.  Six partial products per block as in conv_igemm_bf3.hip.
// Reference call sites: the autograd backward of every nn.Conv2d / nn.ConvTranspose2d in include/viai_hip.h.
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"

namespace {

constexpr int WB_BKP = 32;     // pixels per staged chunk

// NP = 3: bf16x3 (six partial products); NP = 2: f16x2 (three; dy scaled by a power of two from a.amax, x by F16_ASCALE)
template <int NP, int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_bf3_kernel(const WgradArgs a) {
    constexpr int WM = 2, WN = 2;
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr int BM = 32 * TM * WM;       // Cout tile
    constexpr int BN = 32 * TN * WN;       // Cin tile
    constexpr int QA = BM / 4, QB = BN / 4;             // channel quads per pixel row
    constexpr int NA = QA / 8, NB = QB / 8;             // float4 (= pixels) per thread per chunk: 4 or 2
    constexpr int APLANE = BM * BF3_PITCH, BPLANE = BN * BF3_PITCH;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    unsigned char* As = smem_w;                         // [3][BM][80]
    unsigned char* Bs = smem_w + NP * APLANE;           // [NP][BN][80]

    const ConvGeom& g = a.g;
    const float dscale = (NP == 2) ? f16_scale_from_amax(a.amax) : 1.f;
    const float xscale = (NP == 2 && a.xmax != nullptr) ? f16_scale_from_amax(a.xmax) : F16_ASCALE;
    const float dlim = f16_clamp_for_scale(dscale), xlim = f16_clamp_for_scale(xscale);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // 1-D grid, XCD-aware: consecutive logical ids (all taps and channel tiles of one pixel slab) share an XCD's L2,
    // so the slab of dy / x is fetched into ONE L2 instead of eight
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_slab = a.nblk_ci * a.nblk_co * a.g.ntaps;
    const int z = b / per_slab; b -= z * per_slab;
    const int bci = b % a.nblk_ci; b /= a.nblk_ci;
    const int bco = b % a.nblk_co; b /= a.nblk_co;
    const int t = b;
    const int co0 = bco * BM, ci0 = bci * BN;
    const int Cin = a.C1 + a.C2;

    const bool first = ci0 < a.C1;
    const int xcs = first ? a.C1 : a.C2;
    const int xoff = first ? ci0 : ci0 - a.C1;
    const int dyt = g.dy[t], dxt = g.dx[t];
    constexpr int OOB = 0x7fffffff;
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? a.x : a.x2), 0, (int)(in_pixels * xcs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((long)a.M * a.Cout * 4), 0x00020000);

    const int chunk0 = z * a.chunks_per_split;
    const int nchunks_total = (a.M + WB_BKP - 1) / WB_BKP;
    int chunk1 = chunk0 + a.chunks_per_split;
    if (chunk1 > nchunks_total) chunk1 = nchunks_total;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // thread -> (pixel group pg of 4 pixels, channel quad): lanes of a wave cover 8 pixel groups x 8 quads, so every
    // global row segment is a full 128-byte line and the LDS writes of a half-wave hit 32 distinct bank pairs
    const int pg = tid & 7, cq = tid >> 3;                       // cq in [0, 32)
    const int qa = cq % QA, suba = cq / QA;                      // NA = 4: suba = 0; NA = 2: suba in {0, 1}
    const int qb = cq % QB, subb = cq / QB;
    const int pa0 = pg * 4 + suba * NA, pb0 = pg * 4 + subb * NB;   // first pixel (within the chunk) of this thread
    const int dcol = (co0 + qa * 4 < a.Cout) ? (co0 + qa * 4) * 4 : OOB;
    const int xcol = (ci0 + qb * 4 < Cin) ? (xoff + qb * 4) * 4 : OOB;

    int xo[NB], xy[NB], xn[NB];
    const int step_x = WB_BKP % g.OW, step_y = WB_BKP / g.OW;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int p = chunk0 * WB_BKP + pb0 + j;
        xo[j] = p % g.OW; int r = p / g.OW; xy[j] = r % g.OH; xn[j] = r / g.OH;
    }

    u32x4 dreg[NA], xreg[NB];
    auto gload = [&](int chunk) {
        const int p0 = chunk * WB_BKP + pa0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int p = p0 + j;
            int off = (p < a.M && dcol != OOB) ? p * a.Cout * 4 + dcol : OOB;
            dreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int iy = xy[j] * g.my + dyt, ix = xo[j] * g.mx + dxt;
            bool ok = xn[j] < g.N && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW && xcol != OOB;
            int off = ok ? ((xn[j] * g.IH + iy) * g.IW + ix) * xcs * 4 + xcol : OOB;
            xreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            int nx = xo[j] + step_x;
            int carry = nx >= g.OW ? 1 : 0;
            xo[j] = nx - carry * g.OW;
            int ny = xy[j] + step_y + carry;
            while (ny >= g.OH) { ny -= g.OH; ++xn[j]; }
            xy[j] = ny;
        }
    };
    // transpose + split in registers: pk[c][plane] holds the thread's pixels of channel c as packed bf16
    unsigned pka[4][NP][NA / 2], pkb[4][NP][NB / 2];
    auto convert = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int h = 0; h < NA / 2; ++h) {
                const f32x4 v0 = __builtin_bit_cast(f32x4, dreg[2 * h]), v1 = __builtin_bit_cast(f32x4, dreg[2 * h + 1]);
                if constexpr (NP == 3) split3_pair(v0[c], v1[c], pka[c][0][h], pka[c][1][h], pka[c][2][h]);
                else split2_pair(v0[c], v1[c], dscale, dlim, pka[c][0][h], pka[c][1][h]);
            }
#pragma unroll
            for (int h = 0; h < NB / 2; ++h) {
                const f32x4 v0 = __builtin_bit_cast(f32x4, xreg[2 * h]), v1 = __builtin_bit_cast(f32x4, xreg[2 * h + 1]);
                if constexpr (NP == 3) split3_pair(v0[c], v1[c], pkb[c][0][h], pkb[c][1][h], pkb[c][2][h]);
                else split2_pair(v0[c], v1[c], xscale, xlim, pkb[c][0][h], pkb[c][1][h]);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                unsigned char* da = As + p * APLANE + (qa * 4 + c) * BF3_PITCH + pa0 * 2;
                if (NA == 4) { const u32x2 v = {pka[c][p][0], pka[c][p][NA / 2 - 1]}; *reinterpret_cast<u32x2*>(da) = v; }
                else *reinterpret_cast<unsigned*>(da) = pka[c][p][0];
                unsigned char* db = Bs + p * BPLANE + (qb * 4 + c) * BF3_PITCH + pb0 * 2;
                if (NB == 4) { const u32x2 v = {pkb[c][p][0], pkb[c][p][NB / 2 - 1]}; *reinterpret_cast<u32x2*>(db) = v; }
                else *reinterpret_cast<unsigned*>(db) = pkb[c][p][0];
            }
    };

    if (chunk0 < chunk1) {
        gload(chunk0);
        convert();
        lstore();
    }
    __syncthreads();
    const int aoff = (wm * TM * 32 + (lane & 31)) * BF3_PITCH + 16 * (lane >> 5);
    const int boff = (wn * TN * 32 + (lane & 31)) * BF3_PITCH + 16 * (lane >> 5);
    constexpr int PA[6] = {1, NP == 3 ? 2 : 0, 0, 1, 0, 0}, PB[6] = {NP == 3 ? 1 : 0, NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 0, 1, 0};

    for (int c = chunk0; c < chunk1; ++c) {
        const bool more = (c + 1 < chunk1);
        if (more) gload(c + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
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
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (NP == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][PA[pr]]), __builtin_bit_cast(bf16x8, bf[j][PB[pr]]), acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][PA[pr]]), __builtin_bit_cast(f16x8, bf[j][PB[pr]]), acc[i][j], 0, 0, 0);
                    }
        }
        if (more) convert();
        __syncthreads();
        if (more) lstore();
        __syncthreads();
    }

    const float inv = (NP == 2) ? 1.0f / (dscale * xscale) : 1.0f;
    const int half = lane >> 5, col = lane & 31;
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
                if (co < a.Cout && ci < Cin) dst[(size_t)co * Cin + ci] = acc[i][j][e] * inv;
            }
        }
}

template <int TM, int TN>
int launch_wgrad_bf3(WgradArgs& a, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const int Cin = a.C1 + a.C2;
    a.nblk_co = (a.Cout + BM - 1) / BM;
    a.nblk_ci = (Cin + BN - 1) / BN;
    size_t lds = (size_t)3 * (BM + BN) * BF3_PITCH;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf3_kernel<3, TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf3_kernel<2, TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    dim3 grid(a.nblk_co * a.nblk_ci * a.g.ntaps * a.ksplit);
    viai_tag_kernel(a.amax != nullptr ? "wgrad_bf3_f16x2" : "wgrad_bf3_bf16x3");
    if (a.amax != nullptr) VIAI_LAUNCH((wgrad_bf3_kernel<2, TM, TN>), grid, dim3(256), (size_t)2 * (BM + BN) * BF3_PITCH, st, a);   // f16x2
    else VIAI_LAUNCH((wgrad_bf3_kernel<3, TM, TN>), grid, dim3(256), lds, st, a);
    return viai_launch_status();
}

}  // namespace

// bf16x3 covers the tiles with >= 64 channels on both sides; narrower layers stay on the fp32-MFMA kernel.
bool viai_wgrad_bf3_ok(int Cout, int C1, int C2) {
    const int Cin = C1 + C2;
    if (Cout <= 32 || Cin <= 32 || Cout % 4 != 0 || Cin % 4 != 0) return false;
    if (C2 > 0 && (C1 % 64) != 0) return false;
    return true;
}

int viai_wgrad_bf3_launch(WgradArgs& a, int ksplit, hipStream_t st) {
    const int Cin = a.C1 + a.C2;
    if (!viai_wgrad_bf3_ok(a.Cout, a.C1, a.C2) || a.g.run) return (int)hipErrorInvalidValue;
    long chunks = ((long)a.M + WB_BKP - 1) / WB_BKP;
    a.ksplit = ksplit;
    a.chunks_per_split = (int)((chunks + ksplit - 1) / ksplit);
    int bm = a.Cout > 64 ? 128 : 64, bn = Cin > 64 ? 128 : 64;
    if (a.C2 > 0 && (a.C1 % bn) != 0) bn = 64;               // a Cin tile must not straddle the two concatenated sources
    if (bm == 128 && bn == 128) return launch_wgrad_bf3<2, 2>(a, st);
    if (bm == 128) return launch_wgrad_bf3<2, 1>(a, st);
    if (bn == 128) return launch_wgrad_bf3<1, 2>(a, st);
    return launch_wgrad_bf3<1, 1>(a, st);
}
