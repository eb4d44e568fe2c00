// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (MI355X).
//
// One kernel family covers every contraction-shaped layer of the VIAI
// generator / discriminator (reference: networks/Inpainting_Networks.py:55-63,
// networks/New_Inpainting_Networks.py:17-24,53-63, networks/Discriminator_Networks.py:17-33):
//   * forward Conv2d 3x3 (any stride) and forward stride-1 ConvTranspose2d,
//   * the data gradient of both (strided Conv2d dgrad is launched once per
//     output parity class so no MFMA work is spent on structural zeros).
// A "tap table" (ConvGeom) describes which input pixel each tap reads, so the
// kernel itself is a plain gather-GEMM:
//     out[pix][co] = bias[co] + sum_t sum_ci in[pix_t][ci] * Wp[co][ws[t]][ci]
// GEMM view: M = output pixels, N = Cout, K = taps x Cin.  A rows are 128-byte
// channel runs of the NHWC input (coalesced), B rows are packed weights.
//
// Math: v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak).  K order inside a
// chunk is permuted (lane half h owns k = 8j+4h .. 8j+4h+3) so that each lane
// fetches its A and B operands for four consecutive MFMAs with one
// ds_read_b128; A and B use the same permutation, so the sum is complete.
//
// LDS: double-buffered [rows][BK+4] fp32 tiles (row stride 144 B -> the 16-lane
// groups of ds_read_b128 and the 8-lane groups of ds_write_b128 are
// conflict-free).  One barrier per K chunk.
//
// Epilogue: + bias, NHWC store (each store instruction writes two full 128-B
// channel runs), and optional per-channel BatchNorm partials: block-local
// (mean, M2) pairs that bn_finalize merges with Chan's formula in fp64.
#include "viai_common.h"
#include "viai_internal.h"
#include <string>
#include <cstdlib>

namespace {

template <int BK, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int LS = BK + 4;            // padded LDS row (floats)
    constexpr int Q = BK / 4;             // float4 per row
    constexpr int RP = 256 / Q;           // rows covered per staging pass
    constexpr int NA = BM / RP;           // A rows staged per thread
    constexpr int NB = (BN + RP - 1) / RP;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BM % RP == 0, "tile/staging mismatch");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM*LS]
    float* Bs = smem + 2 * BM * LS;       // [2][BN*LS]

    const ConvGeom& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % a.nblk_n, bm = bid / a.nblk_n;
    const int m0 = bm * BM, n0 = bn * BN;
    const int Cin = a.C1 + a.C2;

    // ---- per-thread staging rows.  Loads go through buffer descriptors: a lane whose row is out of the
    // image (halo / M tail / Cout tail) gets an out-of-range offset and the hardware returns zeros, so
    // the K loop has no divergent branches and 32-bit address math.
    const int q = tid % Q;
    const int r0 = tid / Q;
    int pixbase[NA];
    int iy0[NA], ix0[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        int m = m0 + r0 + RP * j;
        if (m < a.M) {
            int ox = m % g.SW;
            int t = m / g.SW;
            int oy = t % g.SH;
            int n = t / g.SH;
            iy0[j] = oy * g.my;
            ix0[j] = ox * g.mx;
            pixbase[j] = (n * g.IH + iy0[j]) * g.IW + ix0[j];
        } else {
            iy0[j] = -100000; ix0[j] = -100000; pixbase[j] = 0;
        }
    }
    constexpr int OOB = 0x7fffffff;
    int brow[NB];                       // byte offset of this thread's weight rows (OOB when masked)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int rr = r0 + RP * j;
        int co = n0 + rr;
        bool ok = (rr < BN) && (co < a.Cout);
        brow[j] = ok ? (co * g.wtaps * Cin + q * 4) * 4 : OOB;
    }
    const long in_pixels = (long)g.N * g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rs_in1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)(in_pixels * (g.run ? 4 : a.C1) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in2 ? a.in2 : a.in), 0,
                                                                             (int)(in_pixels * (a.in2 ? a.C2 : a.C1) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, a.Cout * g.wtaps * Cin * 4, 0x00020000);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int cchunks = (Cin + BK - 1) / BK;     // the last chunk of a tap may be partial (Cin % 4 == 0): masked lanes read zeros
    const int nchunks = g.ntaps * cchunks;

    u32x4 areg[NA], breg[NB];
    auto gload = [&](int t_, int c0_) {
        const int t = __builtin_amdgcn_readfirstlane(t_);      // wave-uniform -> scalar table loads
        const int c0 = __builtin_amdgcn_readfirstlane(c0_);
        const int dyt = g.dy[t], dxt = g.dx[t];
        const int toff = dyt * g.IW + dxt;
        const bool first = c0 < a.C1;
        const int cs = g.run ? 4 : (first ? a.C1 : a.C2);
        const int coff = g.run ? 0 : ((first ? c0 : c0 - a.C1) + q * 4);
        const int qpix = g.run ? q : 0;
        const bool kin = g.run || (c0 + q * 4 < Cin);            // K tail of the last chunk
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            int iy = iy0[j] + dyt, ix = ix0[j] + dxt + qpix;
            bool ok = (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW && kin;
            int off = ok ? ((pixbase[j] + toff + qpix) * cs + coff) * 4 : OOB;
            areg[j] = first ? __builtin_amdgcn_raw_buffer_load_b128(rs_in1, off, 0, 0)
                            : __builtin_amdgcn_raw_buffer_load_b128(rs_in2, off, 0, 0);
        }
        const int woff = (g.ws[t] * Cin + c0) * 4;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            breg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (brow[j] == OOB || !kin) ? OOB : brow[j] + woff, 0, 0);
    };
    auto lstore = [&](int buf) {
        float* Ad = As + buf * BM * LS;
        float* Bd = Bs + buf * BN * LS;
#pragma unroll
        for (int j = 0; j < NA; ++j)
            *reinterpret_cast<u32x4*>(Ad + (r0 + RP * j) * LS + q * 4) = areg[j];
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (RP * (j + 1) <= BN || r0 + RP * j < BN)
                *reinterpret_cast<u32x4*>(Bd + (r0 + RP * j) * LS + q * 4) = breg[j];
    };

    int t_next = 0, c_next = 0;
    gload(0, 0);
    lstore(0);
    c_next = BK;
    if (c_next >= Cin) { c_next = 0; t_next = 1; }   // (c_next advances in BK steps; a partial last chunk still counts as one)
    __syncthreads();

    const int arow = (wm * TM * 32 + (lane & 31)) * LS + 4 * (lane >> 5);
    const int brow_l = (wn * TN * 32 + (lane & 31)) * LS + 4 * (lane >> 5);

    // Blocked summation: a chunk (BK = 32 products per output) accumulates from zero and joins the running total once.  One fmaf chain over the whole
    // K (2304 .. 4608 terms in the wide layers) rounds K times in sequence and was LESS accurate than the f16x2 kernels it is the reference point for
    // (whose MFMAs add 16 exact products per rounding): sums with heavy cancellation behind it -- the BatchNorm bias gradients of the
    // discriminator -- came out 3e-3 off fp64 where CPU fp32 (blocked GEMM) is at 6e-6 (tools/grad_table.py under VIAI_MATH=fp32).
    f32x16 tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) tot[i][j] = acc[i][j];
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = kc & 1;
        const bool more = (kc + 1 < nchunks);
        if (more) {
            gload(t_next, c_next);
            c_next += BK;
            if (c_next >= Cin) { c_next = 0; ++t_next; }
        }
        const float* Ab = As + cur * BM * LS + arow;
        const float* Bb = Bs + cur * BN * LS + brow_l;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LS + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LS + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                tot[i][j] += acc[i][j];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            }
        if (more) lstore(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];

    // ---------------------------------------------------------------- epilogue
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
        // block-local per-channel (mean, M2) over the valid rows of this tile
        float* red = smem;                  // [WM][BN], main-loop LDS is dead after the last barrier
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
            for (int j = 0; j < TN; ++j) {
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
}

template <int BK, int TM, int TN, int WM, int WN>
int launch_igemm(ConvArgs& a, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, LS = BK + 4;
    a.nblk_m = (a.M + BM - 1) / BM;
    a.nblk_n = (a.Cout + BN - 1) / BN;
    size_t lds = (size_t)2 * (BM + BN) * LS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BK, TM, TN, WM, WN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    static const std::string fam = "igemm" + std::to_string(BM) + "x" + std::to_string(BN) + "_f32";
    viai_tag_kernel(fam.c_str());
    VIAI_LAUNCH((conv_igemm_kernel<BK, TM, TN, WM, WN>), dim3(a.nblk_m * a.nblk_n), dim3(256), lds, st, a);
    return viai_launch_status();
}

// Wp[no][t][ki] = W[no*s_no + ki*s_ki + t]   (taps are innermost in both torch layouts)
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                   int n_out, int k_in, int taps, long s_no, long s_ki) {
    long total = (long)n_out * taps * k_in;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int ki = (int)(i % k_in);
        long r = i / k_in;
        int t = (int)(r % taps);
        int no = (int)(r / taps);
        wp[i] = w[no * s_no + ki * s_ki + t];
    }
}

}  // namespace

// Tile choice.  128-row tiles give the best MFMA:load ratio but a launch needs >= ~2 blocks per CU
// (256 CUs) to fill the chip; the small-M layers (E.conv4/5, G.deconv1_*, convblock2/3 at batch 16)
// have only 16..256 such tiles, so they take 64x64 tiles (4x the blocks, 1/4 the per-block latency).
int viai_igemm_tile_m(long M, int n_out) {
    if (n_out <= 32) return 128;
    constexpr int force = 0;
    if (force == 64 || force == 128) return force;
    long b128 = ((M + 127) / 128) * ((n_out + 127) / 128);
    if (n_out > 64) return b128 >= 512 ? 128 : 64;
    long b64 = ((M + 127) / 128) * ((n_out + 63) / 64);
    return b64 >= 512 ? 128 : 64;
}

int viai_conv_igemm_launch(ConvArgs& a, hipStream_t st) {
    const int Cin = a.C1 + a.C2;
    if (Cin % 4 != 0 || (a.C2 > 0 && a.C1 % 32 != 0)) return (int)hipErrorInvalidValue;
    if (a.OC1 % 32 != 0 && a.OC1 != a.Cout) return (int)hipErrorInvalidValue;
    if (a.Cout % 4 != 0 && a.Cout < 4) return (int)hipErrorInvalidValue;
    const int bm = viai_igemm_tile_m(a.M, a.Cout);
    if (bm == 64) return launch_igemm<32, 1, 1, 2, 2>(a, st);
    if (a.Cout > 64) return launch_igemm<32, 2, 2, 2, 2>(a, st);
    if (a.Cout > 32) return launch_igemm<32, 2, 1, 2, 2>(a, st);
    return launch_igemm<32, 1, 1, 4, 1>(a, st);
}

extern "C" int viai_pack_weight(const float* w, float* wp, int n_out, int k_in, int taps,
                                long s_no, long s_ki, void* stream) {
    long total = (long)n_out * taps * k_in;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    VIAI_LAUNCH(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, n_out, k_in, taps, s_no, s_ki);
    return viai_launch_status();
}
