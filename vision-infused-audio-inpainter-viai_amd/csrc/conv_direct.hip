// Bandwidth-bound members of the conv family (gfx950): layers with Cin == 1
// (E.conv1 3x3 s2, D.conv1 1x4 s(1,2)) or Cout == 1 (G.conv6_2, D.conv4).
// K is 4..9 (or N is 1) there, so these are streaming kernels, not GEMMs:
// coalesced 16-byte channel accesses, weights in registers / L1, wave shuffles
// for the channel reductions.  Reference call sites:
//   networks/Inpainting_Networks.py:55, networks/Discriminator_Networks.py:17,33,
//   networks/New_Inpainting_Networks.py:63.
#include "viai_common.h"
#include "viai_internal.h"
#include "viai_bf3.h"

namespace {

struct DirectArgs {
    const float* x; const float* w; const float* bias; const float* dy;
    float* y; float* dx; float* ws; float* stat;
    int N, IH, IW, OH, OW, Cin, Cout;
    int kh, kw, sh, sw, ph, pw;      // forward-form geometry: in = o*s + d, d = r - p (conv) or p - r (convT)
    int transposed;
    int M;                            // N*OH*OW
    int nblk;
    int act; float slope;
    // fused Cin = 1 conv + BatchNorm(train) + activation layer (viai_conv2d_cin1_bn_*): y is never stored, it is recomputed from x
    const float* scale; const float* shift; const float* mean; const float* invstd; const float* sums; const float* dz;
    float* part;
    float* zmax;                      // fused layer, apply pass: max |z| (operand scale of the f16x2 kernels that consume z), or null
    int dy_p16;                       // pair kernels: dy is written pre-split (P16): the reduce pass adds max |dpre| partials (part stride 3), the apply pass scales by the
                                      // bound k_sums[2 C + c] (bn_bwd_final_kernel, ps = 3) and stores it in *amax
    int z_p16;                        // apply pass: z is written pre-split (P16, viai_bf3.h) with the scale of the bound |gamma| p16_rad + |beta|, stored in *zmax
    const float* gamma; const float* beta; float p16_rad;
    // fused (conv + BatchNorm + activation) -> (Cout = 1 conv) pair (viai_pair_cout1_*): the Cout = 1 layer reads the PRE-BatchNorm tensor
    // y of the layer in front of it and applies z = act_in(scale * y + shift) on load; its data gradient dz is never stored either, the
    // BatchNorm backward forms it from du (the gradient of the Cout = 1 layer's pre-activation output) with the nine taps in registers
    int act_in;
    const float* xmask;               // Cin = 1 kernels: optional (N, IW) factor per input column -- x is read as x[n][iy][ix] * xmask[n][ix] (the time
                                      // mask of the inpainting step, s_in = s * mask, applied where E.conv1 loads s instead of in a pass of its own)
    const float* k_sums;              // {k0, k1} of bn_bwd_final (apply pass)
    float* amax;                      // apply pass: max |dy|
    // block = whole output rows of one image (host-checked): the block's input rows are staged in LDS once (xrows x xpitch floats)
    int xfast, xrows, xpitch, rows_blk;
};

__device__ __forceinline__ int tap_dy(const DirectArgs& a, int r) { return a.transposed ? a.ph - r : r - a.ph; }
__device__ __forceinline__ int tap_dx(const DirectArgs& a, int s) { return a.transposed ? a.pw - s : s - a.pw; }

// q / d and q % d for 0 <= q < 2^24 by a float reciprocal and one correction step (an integer division is ~30 instructions)
__device__ __forceinline__ void fast_divmod(int q, int d, float inv_d, int& quo, int& rem) {
    quo = (int)((float)q * inv_d);
    rem = q - quo * d;
    if (rem < 0) { --quo; rem += d; } else if (rem >= d) { ++quo; rem -= d; }
}


// The taps of a Cin = 1 layer are 4-byte loads that 16 .. 64 lanes share (every channel lane of a pixel reads the same x): as global
// loads each costs a full 64-lane pass through the address unit (16 cycles), 4 .. 9 taps x 16 pixels per thread, and the kernels were
// bound by THAT, not by HBM (the statistics-only forward read 4 MB in 31 us).  When a block covers whole output rows of one image
// its input rows are staged in LDS once (coalesced), zero padding included, and the taps become LDS broadcasts.
__device__ __forceinline__ float cin1_x(const DirectArgs& a, const float* xb, int n, int iy, int ix) {
    const float v = xb[iy * a.IW + ix];
    return a.xmask != nullptr ? v * a.xmask[(size_t)n * a.IW + ix] : v;
}
constexpr int CIN1_XP = 2560;             // floats of LDS for the staged rows
// (tid = index among the 256 threads that work the chunk; live = false: a chunk past the last pixel stages nothing but joins the barrier)
__device__ __forceinline__ void cin1_stage_rows(const DirectArgs& a, float* xp, int n, int oy0, int tid, bool live = true) {
    const int iy0 = oy0 * a.sh - a.ph;
    const float* xb = a.x + (size_t)n * a.IH * a.IW;
    for (int idx = tid; live && idx < a.xrows * a.xpitch; idx += 256) {
        const int r = idx / a.xpitch, c = idx - r * a.xpitch;
        const int iy = iy0 + r, ix = c - a.pw;
        xp[idx] = ((unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW) ? cin1_x(a, xb, n, iy, ix) : 0.f;
    }
    __syncthreads();
}
// The taps are read from LDS one value per VGPR, pinned by an empty asm: left to itself the compiler fetched them with ds_read2_b32 into
// register PAIRS and fed v_pk_fma_f32 op_sel broadcasts from the pair, and that sequence produced wrong sums (one tap's term missing
// for some lanes) whenever waves of the patch weight-gradient kernel shared the CU -- found as a run-to-run difference of D.bn1's
// gradients in the three-stream step, reproduced with an in-kernel check (LDS values correct, packed-FMA result wrong), gone with
// single registers.  Bit-identical to the global-load path.
template <int KH, int KW>
__device__ __forceinline__ void cin1_lds_taps(const float* xr, int pitch, float (&xt)[KH * KW]) {
#pragma unroll
    for (int r = 0; r < KH; ++r)
#pragma unroll
        for (int q = 0; q < KW; ++q) {
            xt[r * KW + q] = xr[r * pitch + q];
            asm volatile("" : "+v"(xt[r * KW + q]));
        }
}
static bool cin1_rows_ok(DirectArgs& a, int pix_per_blk, int KH, int KW, int which = 1) {
    a.xfast = 0;
    constexpr int on = 7;
    if (!(on & which) || a.transposed || pix_per_blk % a.OW != 0 || (a.OH * a.OW) % pix_per_blk != 0) return false;
    a.rows_blk = pix_per_blk / a.OW;
    a.xrows = (a.rows_blk - 1) * a.sh + KH;
    a.xpitch = (a.OW - 1) * a.sw + KW;
    if (a.xrows * a.xpitch > CIN1_XP) return false;
    a.xfast = 1;
    return true;
}

// All kernels are templated on the (compile-time) kernel window KH x KW so that tap offsets, validity
// tests and the weight registers are resolved at compile time (a runtime tap loop costs ~30 VALU
// instructions of integer division per tap per pixel and made these kernels VALU-bound).

// ------------------------------------------------------------------ Cin == 1
// y[p][co] = sum_t x[p_t] * w[co][t];  thread = (pixel lane, 4 output channels)
constexpr int CIN1_PB = 256;   // pixels per block == rows_per_blk of the BN partials

// MODE 0: y (+ BatchNorm partials when a.stat);  MODE 1: the partials only, y is NOT stored;  MODE 2: z = act(scale * y + shift)
// (modes 1 + 2 = the fused conv + BatchNorm(train) + activation layer: with K = 4 .. 9 the conv is cheaper to recompute than its
// 64-channel output is to write and read back -- D.conv1: 134 MB per pass over y)
// NT = 1024 (MODE 2 only): four 256-pixel chunks per block, each worked by its own four waves exactly as a 256-thread block would -- the
// pass ends in the abs-max atomics, and a quarter of the blocks is a quarter of the same-address atomics queued behind the last store
// (viai_common.h block_absmax_to)
template <int CG, int KH, int KW, int MODE = 0, int NT = 256>   // channel groups of 4 (Cout = 4*CG)
__global__ __launch_bounds__(NT) void cin1_fwd_kernel(const DirectArgs a) {
    constexpr int PG = 256 / CG;
    constexpr int IT = CIN1_PB / PG;
    constexpr int T = KH * KW;
    static_assert(NT == 256 || (NT == 1024 && MODE == 2), "fat blocks: apply pass only");
    __shared__ float red[MODE == 2 ? 1 : PG][CG * 4];
    const int sub = NT == 256 ? 0 : (int)(threadIdx.x >> 8);
    const int chunk = (int)blockIdx.x * (NT / 256) + sub;
    const int tid = threadIdx.x & 255, cg = tid % CG, pg = tid / CG;
    f32x4 wv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        wv[t][0] = a.w[(cg * 4 + 0) * T + t]; wv[t][1] = a.w[(cg * 4 + 1) * T + t];
        wv[t][2] = a.w[(cg * 4 + 2) * T + t]; wv[t][3] = a.w[(cg * 4 + 3) * T + t];
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + cg * 4);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    float pS = 1.f, pL = 0.f;
    if constexpr (MODE == 2) {
        sc = *reinterpret_cast<const f32x4*>(a.scale + cg * 4); sh = *reinterpret_cast<const f32x4*>(a.shift + cg * 4);
        if (a.z_p16) { pS = p16_fwd_scale(a.gamma, a.beta, a.Cout, a.p16_rad, a.zmax); pL = f16_clamp_for_scale(pS); }
    }
    const int p0 = chunk * CIN1_PB;
    __shared__ float xp_all[NT / 256][CIN1_XP];
    float* xp = xp_all[sub];
    const int oy_blk = (p0 / a.OW) % a.OH;
    if (a.xfast) cin1_stage_rows(a, xp, p0 / (a.OH * a.OW), oy_blk, tid, p0 < a.M);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    f32x4 vals[IT];
    float zmx = 0.f;
    // decode the first pixel once, then walk: the per-pixel 32-bit divisions (p % OW, p / OW, ... = ~100 instructions) cost more than
    // the 4 .. 9 multiply-adds of the pixel itself and kept this streaming kernel at 2.7 TB/s of writes
    int ox, oy, n;
    {
        const int p = p0 + pg;
        ox = p % a.OW; const int r_ = p / a.OW; oy = r_ % a.OH; n = r_ / a.OH;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        int p = p0 + it * PG + pg;
        f32x4 v = bv;
        if (it > 0) {
            ox += PG;
            while (ox >= a.OW) { ox -= a.OW; if (++oy >= a.OH) { oy = 0; ++n; } }
        }
        if (p < a.M) {
            const float* xb = a.x + (size_t)n * a.IH * a.IW;
            const int iy0 = oy * a.sh, ix0 = ox * a.sw;
            if (a.xfast) {
                float xt[KH * KW];
                cin1_lds_taps<KH, KW>(xp + (oy - oy_blk) * a.sh * a.xpitch + ix0, a.xpitch, xt);
#pragma unroll
                for (int t = 0; t < KH * KW; ++t) v += xt[t] * wv[t];
            } else {
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const int iy = iy0 + tap_dy(a, r);
                const bool yok = (unsigned)iy < (unsigned)a.IH;
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    const int ix = ix0 + tap_dx(a, q);
                    float xv = (yok && (unsigned)ix < (unsigned)a.IW) ? cin1_x(a, xb, n, iy, ix) : 0.f;
                    v += xv * wv[r * KW + q];
                }
            }
            }
            if constexpr (MODE == 2) {
                f32x4 z;
#pragma unroll
                for (int e = 0; e < 4; ++e) { z[e] = viai_act(v[e] * sc[e] + sh[e], a.act, a.slope); zmx = fmaxf(zmx, fabsf(z[e])); }
                if (a.z_p16) p16_store_quad(a.y + (size_t)p * a.Cout, cg, z, pS, pL);
                else *reinterpret_cast<f32x4*>(a.y + (size_t)p * a.Cout + cg * 4) = z;
            } else {
                if (a.stat == nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = viai_act(v[e], a.act, a.slope);
                }
                if constexpr (MODE == 0) *reinterpret_cast<f32x4*>(a.y + (size_t)p * a.Cout + cg * 4) = v;
                sum += v;
            }
        } else {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        vals[it] = v;
    }
    if constexpr (MODE == 2) {
        if (a.zmax != nullptr && !a.z_p16) block_absmax_to(a.zmax, zmx);
    } else {
    if (a.stat == nullptr) return;
    // block-local (mean, M2) per channel over the valid pixels of this block
    const int cnt = min(CIN1_PB, a.M - p0);
    *reinterpret_cast<f32x4*>(&red[pg][cg * 4]) = sum;
    __syncthreads();
    f32x4 mean = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < PG; ++k) mean += *reinterpret_cast<f32x4*>(&red[k][cg * 4]);
    mean *= 1.f / (float)cnt;
    __syncthreads();
    f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        int p = p0 + it * PG + pg;
        if (p < a.M) { f32x4 d = vals[it] - mean; m2 += d * d; }
    }
    *reinterpret_cast<f32x4*>(&red[pg][cg * 4]) = m2;
    __syncthreads();
    if (pg == 0) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < PG; ++k) t += *reinterpret_cast<f32x4*>(&red[k][cg * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a.stat[(size_t)(cg * 4 + e) * a.nblk + blockIdx.x] = mean[e];
            a.stat[(size_t)(a.Cout + cg * 4 + e) * a.nblk + blockIdx.x] = t[e];
        }
    }
    }
}

// Statistics pass of the fused Cin = 1 conv + BatchNorm(train) layer WITHOUT computing the conv (round 5).  y[p][c] = b[c] + sum_t w[c][t] x_t(p) is
// linear in the T = KH KW taps, so over any set of pixels   mean_c = b[c] + w_c . m   and   M2_c = w_c^T C w_c   with m = the tap means and
// C = the centred second-moment matrix of the tap vectors: T + T (T + 1) / 2 sums per pixel block instead of Cout x T multiply-adds per pixel and
// a per-channel two-pass reduction -- D.conv1 (1 -> 64 channels, 1 x 4): 14 sums against 256 FMAs per pixel.  Same output as cin1_fwd_kernel<.., 1>
// (one (mean, M2) per 256-pixel block and channel, the layout bn_finalize reduces); the moments and the quadratic form are accumulated in fp64
// (neighbouring mel bins are strongly correlated, so w^T C w cancels), which makes the partials MORE accurate than the fp32 two-pass they replace;
// they agree with it to fp32 rounding, not bit for bit.  25.7 -> ~5 us per D pass on the metric config.
template <int KH, int KW>
__global__ __launch_bounds__(256) void cin1_stats_cov_kernel(const DirectArgs a) {
    constexpr int T = KH * KW, NC = T * (T + 1) / 2;
    constexpr int SL = 256 / NC;                       // pixel slices per (t, u) pair: thread = (pair, slice)
    constexpr int MS = 256 / T;                        // ... per tap for the means
    __shared__ float xs[T][257];                       // the block's tap vectors, tap-major; rows of 257 words: lanes that read the same pixel of different taps hit different banks
    __shared__ double part[256], mt[T], cm[NC];
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * CIN1_PB, p = p0 + tid;
    const int cnt = min(CIN1_PB, a.M - p0);
    {
        float xt[T];
#pragma unroll
        for (int t = 0; t < T; ++t) xt[t] = 0.f;
        if (p < a.M) {
            const int ox = p % a.OW; const int r_ = p / a.OW; const int oy = r_ % a.OH, n = r_ / a.OH;
            const float* xb = a.x + (size_t)n * a.IH * a.IW;
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const int iy = oy * a.sh + tap_dy(a, r);
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    const int ix = ox * a.sw + tap_dx(a, q);
                    if ((unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW) xt[r * KW + q] = cin1_x(a, xb, n, iy, ix);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) xs[t][tid] = xt[t];
    }
    __syncthreads();
    // tap means: thread (t, slice) sums its pixels, thread t adds the slices in a fixed order
    if (tid < T * MS) {
        const int t = tid / MS, sl = tid % MS;
        double sv = 0.0;
        for (int i = sl; i < cnt; i += MS) sv += (double)xs[t][i];
        part[tid] = sv;
    }
    __syncthreads();
    if (tid < T) {
        double sv = 0.0;
        for (int sl = 0; sl < MS; ++sl) sv += part[tid * MS + sl];
        mt[tid] = sv / (double)cnt;
    }
    __syncthreads();
    // centred second moments: thread (pair k = (t, u), slice)
    if (tid < NC * SL) {
        const int k = tid / SL, sl = tid % SL;
        int t = 0, rem = k;
        while (rem >= T - t) { rem -= T - t; ++t; }
        const int u = t + rem;
        const double m0 = mt[t], m1 = mt[u];
        double sv = 0.0;
        for (int i = sl; i < cnt; i += SL) sv += ((double)xs[t][i] - m0) * ((double)xs[u][i] - m1);
        part[tid] = sv;
    }
    __syncthreads();
    if (tid < NC) {
        double sv = 0.0;
        for (int sl = 0; sl < SL; ++sl) sv += part[tid * SL + sl];
        cm[tid] = sv;
    }
    __syncthreads();
    for (int c = tid; c < a.Cout; c += 256) {
        double w[T], mean = a.bias ? (double)a.bias[c] : 0.0, m2 = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) { w[t] = (double)a.w[c * T + t]; mean += w[t] * mt[t]; }
        int k = 0;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int u = t; u < T; ++u, ++k) m2 += (t == u ? 1.0 : 2.0) * w[t] * w[u] * cm[k];
        a.stat[(size_t)c * a.nblk + blockIdx.x] = (float)mean;
        a.stat[(size_t)(a.Cout + c) * a.nblk + blockIdx.x] = (float)(m2 > 0.0 ? m2 : 0.0);
    }
}

__device__ __forceinline__ float dact(float pre, int act, float slope) {
    if (act == VIAI_ACT_SIGMOID) { float s_ = 1.f / (1.f + __expf(-pre)); return s_ * (1.f - s_); }
    return viai_act_grad_pl(pre, act, slope);
}

// Backward of the fused Cin = 1 conv + BatchNorm(train) + activation layer with y RECOMPUTED from x (same expression, same order as
// the forward kernel):  dp = dz * act'(scale * y + shift)
//   APPLY = false: part[blk][0][c] = sum dp, part[blk][1][c] = sum dp * (y - mean) * invstd over the block's 256 pixels (the layout
//                  bn_bwd_final_kernel reduces);
//   APPLY = true:  dy = scale * dp + k1 * (y - mean) + k0  (sums = {k0, k1}) -- only where a data gradient needs dy in memory (the
//                  frozen-D pass of the G step); the weight gradient applies it on the fly (cin1_wgrad_kernel<.., true>).
template <int CG, int KH, int KW, bool APPLY>
__global__ __launch_bounds__(256) void cin1_bn_bwd_kernel(const DirectArgs a) {
    constexpr int PG = 256 / CG;
    constexpr int IT = CIN1_PB / PG;
    constexpr int T = KH * KW;
    __shared__ float red[2][PG][CG * 4];
    const int tid = threadIdx.x, cg = tid % CG, pg = tid / CG;
    f32x4 wv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        wv[t][0] = a.w[(cg * 4 + 0) * T + t]; wv[t][1] = a.w[(cg * 4 + 1) * T + t];
        wv[t][2] = a.w[(cg * 4 + 2) * T + t]; wv[t][3] = a.w[(cg * 4 + 3) * T + t];
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + cg * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + cg * 4), sh = *reinterpret_cast<const f32x4*>(a.shift + cg * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cg * 4);
    f32x4 is = {0.f, 0.f, 0.f, 0.f}, k0 = is, k1 = is;
    if constexpr (APPLY) { k0 = *reinterpret_cast<const f32x4*>(a.sums + cg * 4); k1 = *reinterpret_cast<const f32x4*>(a.sums + a.Cout + cg * 4); }
    else is = *reinterpret_cast<const f32x4*>(a.invstd + cg * 4);
    const int p0 = blockIdx.x * CIN1_PB;
    __shared__ float xp[CIN1_XP];
    const int oy_blk = (p0 / a.OW) % a.OH;
    if (a.xfast) cin1_stage_rows(a, xp, p0 / (a.OH * a.OW), oy_blk, threadIdx.x);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    int ox, oy, n;
    {
        const int p = p0 + pg;
        ox = p % a.OW; const int r_ = p / a.OW; oy = r_ % a.OH; n = r_ / a.OH;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int p = p0 + it * PG + pg;
        if (it > 0) {
            ox += PG;
            while (ox >= a.OW) { ox -= a.OW; if (++oy >= a.OH) { oy = 0; ++n; } }
        }
        if (p < a.M) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.dz + (size_t)p * a.Cout + cg * 4);
            f32x4 v = bv;
            const float* xb = a.x + (size_t)n * a.IH * a.IW;
            const int iy0 = oy * a.sh, ix0 = ox * a.sw;
            if (a.xfast) {
                float xt[KH * KW];
                cin1_lds_taps<KH, KW>(xp + (oy - oy_blk) * a.sh * a.xpitch + ix0, a.xpitch, xt);
#pragma unroll
                for (int t = 0; t < KH * KW; ++t) v += xt[t] * wv[t];
            } else {
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const int iy = iy0 + tap_dy(a, r);
                const bool yok = (unsigned)iy < (unsigned)a.IH;
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    const int ix = ix0 + tap_dx(a, q);
                    float xv = (yok && (unsigned)ix < (unsigned)a.IW) ? cin1_x(a, xb, n, iy, ix) : 0.f;
                    v += xv * wv[r * KW + q];
                }
            }
            }
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dp = g[e] * dact(v[e] * sc[e] + sh[e], a.act, a.slope);
                if constexpr (APPLY) o[e] = sc[e] * dp + (k1[e] * (v[e] - mu[e]) + k0[e]);
                else { s1[e] += dp; s2[e] += dp * (v[e] - mu[e]) * is[e]; }
            }
            if constexpr (APPLY) *reinterpret_cast<f32x4*>(a.dx + (size_t)p * a.Cout + cg * 4) = o;
        }
    }
    if constexpr (!APPLY) {
        *reinterpret_cast<f32x4*>(&red[0][pg][cg * 4]) = s1;
        *reinterpret_cast<f32x4*>(&red[1][pg][cg * 4]) = s2;
        __syncthreads();
        if (pg < 2) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < PG; ++k) t += *reinterpret_cast<f32x4*>(&red[pg][k][cg * 4]);
            *reinterpret_cast<f32x4*>(a.part + ((size_t)blockIdx.x * 2 + pg) * a.Cout + cg * 4) = t;
        }
    }
}

// dx[q] = sum_t sum_co dy[o_t(q)][co] * w[co][t]   (Cin == 1), LPP = Cout/4 lanes per input pixel
// SH, SW: the strides as compile-time constants (0 = use a.sh / a.sw): the tap validity test divides by them per tap and
// pixel, and a runtime integer division is ~30 instructions
// FUSED (the fused Cin = 1 conv + BatchNorm layer): a.dy is not read; dy[o] = scale * dz[o] * act'(.) + k1 * (y[o] - mean) + k0 is formed
// per contributing output pixel with y[o] recomputed from a.x (the frozen-D pass of the G step: no dy tensor at all)
template <int LPP, int KH, int KW, int SH = 0, int SW = 0, bool FUSED = false>
__global__ __launch_bounds__(256) void cin1_dgrad_kernel(const DirectArgs a) {
    constexpr int T = KH * KW;
    const int sh = SH ? SH : a.sh, sw = SW ? SW : a.sw;
    const int cl = threadIdx.x % LPP;
    f32x4 wv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        wv[t][0] = a.w[(cl * 4 + 0) * T + t]; wv[t][1] = a.w[(cl * 4 + 1) * T + t];
        wv[t][2] = a.w[(cl * 4 + 2) * T + t]; wv[t][3] = a.w[(cl * 4 + 3) * T + t];
    }
    f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, bk0 = bsc, bk1 = bsc;
    if constexpr (FUSED) {
        bsc = *reinterpret_cast<const f32x4*>(a.scale + cl * 4); bsh = *reinterpret_cast<const f32x4*>(a.shift + cl * 4);
        bmu = *reinterpret_cast<const f32x4*>(a.mean + cl * 4);
        bk0 = *reinterpret_cast<const f32x4*>(a.sums + cl * 4); bk1 = *reinterpret_cast<const f32x4*>(a.sums + a.Cout + cl * 4);
    }
    const int total = a.N * a.IH * a.IW;
    const int qstep = gridDim.x * (256 / LPP);
    const bool small = total < (1 << 24);
    const float inv_iw = 1.0f / (float)a.IW, inv_ih = 1.0f / (float)a.IH;
    for (int q = blockIdx.x * (256 / LPP) + threadIdx.x / LPP; q < total + (256 / LPP); q += qstep) {
        float accv = 0.f;
        const bool live = q < total;
        if (live) {
            int ix, r_, iy, n;
            if (small) { fast_divmod(q, a.IW, inv_iw, r_, ix); fast_divmod(r_, a.IH, inv_ih, n, iy); }
            else { ix = q % a.IW; r_ = q / a.IW; iy = r_ % a.IH; n = r_ / a.IH; }
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const int ny = iy - tap_dy(a, r);
                if (ny < 0 || ny % sh != 0) continue;
                const int oy = ny / sh;
                if (oy >= a.OH) continue;
#pragma unroll
                for (int s_ = 0; s_ < KW; ++s_) {
                    const int nx = ix - tap_dx(a, s_);
                    if (nx < 0 || nx % sw != 0) continue;
                    const int ox = nx / sw;
                    if (ox >= a.OW) continue;
                    f32x4 d = *reinterpret_cast<const f32x4*>((FUSED ? a.dz : a.dy) + ((size_t)(n * a.OH + oy) * a.OW + ox) * a.Cout + cl * 4);
                    if constexpr (FUSED) {
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        const float* xb = a.x + (size_t)n * a.IH * a.IW;
#pragma unroll
                        for (int r2 = 0; r2 < KH; ++r2) {
                            const int iy2 = oy * sh + tap_dy(a, r2);
                            const bool yok = (unsigned)iy2 < (unsigned)a.IH;
#pragma unroll
                            for (int q2 = 0; q2 < KW; ++q2) {
                                const int ix2 = ox * sw + tap_dx(a, q2);
                                const float xv = (yok && (unsigned)ix2 < (unsigned)a.IW) ? cin1_x(a, xb, n, iy2, ix2) : 0.f;
                                v += xv * wv[r2 * KW + q2];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float dp = d[e] * dact(v[e] * bsc[e] + bsh[e], a.act, a.slope);
                            d[e] = bsc[e] * dp + (bk1[e] * (v[e] - bmu[e]) + bk0[e]);
                        }
                    }
                    const f32x4 w = wv[r * KW + s_];
                    accv += d[0] * w[0] + d[1] * w[1] + d[2] * w[2] + d[3] * w[3];
                }
            }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) accv += __shfl_xor(accv, o, 64);
        if (live && cl == 0) a.dx[q] = accv;
        if (q >= total) break;
    }
}

// sum over the LPP consecutive lanes of a lane group (LPP = 8 .. 64, a power of two); every lane of the group gets the sum
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LPP == 64) return wave_sum_dpp(v);      // whole wave: row sums by DPP + four readlanes, no ds_bpermute
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // xor 1
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // xor 2
    v += dpp(v, std::integral_constant<int, 0x141>{});     // mirror within 8 lanes
    if constexpr (LPP >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});     // mirror within 16 lanes
    if constexpr (LPP >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LPP >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// Data gradient of the fused Cin = 1 conv + BatchNorm + activation layer in ONE pass over dz, for windows of one row (KH == 1, sh == 1,
// ph == 0: D.conv1, the frozen-D pass of the G step).  cin1_dgrad_kernel maps threads to INPUT pixels: every dz vector is fetched (and,
// FUSED, its y recomputed) once per tap, and the unfused route writes dy (134 MB) only to read it back: 55 + 80 us.  Here a thread owns
// an OUTPUT pixel's channel quad: dz is loaded once, y recomputed once from the LDS-staged input rows, dy formed in registers, and the
// KW products <dy[o][:], w[:][t]> are reduced over the pixel's channel lanes into c[o][t] in LDS; an output row touches one input row
// only, so the block then gathers dx[iy][ix] = sum over (ox, t) with ox sw + t - pw == ix of c[ox][t] for its own rows.  Fixed order.
template <int CG, int KW>
__global__ __launch_bounds__(256) void cin1_bn_dgrad_rows_kernel(const DirectArgs a) {
    constexpr int PG = 256 / CG;
    constexpr int IT = CIN1_PB / PG;
    __shared__ float cs[CIN1_PB * KW];
    __shared__ float xp[CIN1_XP];
    const int tid = threadIdx.x, cg = tid % CG, pg = tid / CG;
    f32x4 wv[KW];
#pragma unroll
    for (int t = 0; t < KW; ++t) {
        wv[t][0] = a.w[(cg * 4 + 0) * KW + t]; wv[t][1] = a.w[(cg * 4 + 1) * KW + t];
        wv[t][2] = a.w[(cg * 4 + 2) * KW + t]; wv[t][3] = a.w[(cg * 4 + 3) * KW + t];
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + cg * 4), sh = *reinterpret_cast<const f32x4*>(a.shift + cg * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cg * 4);
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(a.sums + cg * 4), k1 = *reinterpret_cast<const f32x4*>(a.sums + a.Cout + cg * 4);
    const int p0 = blockIdx.x * CIN1_PB;
    const int n = p0 / (a.OH * a.OW), oy_blk = (p0 / a.OW) % a.OH;
    cin1_stage_rows(a, xp, n, oy_blk, tid);
    // dz in chunks of four items, one chunk ahead of the arithmetic.  All 2048 blocks of this launch are resident at once and start together:
    // with the sixteen loads of a thread issued up front the whole chip first waited for memory and then computed (27 + 22 us in a row);
    // staged, a wave's next loads fly while it (and the seven other waves of its SIMD) work on the previous chunk.
    constexpr int CH = 4, NCH = IT / CH;
    static_assert(IT % CH == 0, "chunks");
    f32x4 g[2][CH];
    auto gload = [&](int c, f32x4 (&dst)[CH]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) dst[u] = *reinterpret_cast<const f32x4*>(a.dz + (size_t)(p0 + (c * CH + u) * PG + pg) * a.Cout + cg * 4);
    };
    gload(0, g[0]);
    int ox = pg % a.OW, row = pg / a.OW;                  // pixel p0 + pg of the block's rows
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) gload(c + 1, g[(c + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int it = c * CH + u;
            if (it > 0) { ox += PG; while (ox >= a.OW) { ox -= a.OW; ++row; } }
            float xt[KW];
            cin1_lds_taps<1, KW>(xp + row * a.xpitch + ox * a.sw, a.xpitch, xt);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < KW; ++t) v += xt[t] * wv[t];
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dp = g[c & 1][u][e] * dact(v[e] * sc[e] + sh[e], a.act, a.slope);
                d[e] = sc[e] * dp + (k1[e] * (v[e] - mu[e]) + k0[e]);
            }
#pragma unroll
            for (int t = 0; t < KW; ++t) {
                float cc = (d[0] * wv[t][0] + d[1] * wv[t][1]) + (d[2] * wv[t][2] + d[3] * wv[t][3]);
                cc = group_sum<CG>(cc);                   // DPP row operations: __shfl_xor is a ds_bpermute, and sixteen of them per item made the LDS crossbar the bound
                if (cg == 0) cs[(it * PG + pg) * KW + t] = cc;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    float* dxb = a.dx + ((size_t)n * a.IH + oy_blk) * a.IW;        // ph == 0, sh == 1: input row == output row
    for (int idx = tid; idx < a.rows_blk * a.IW; idx += 256) {
        const int r = idx / a.IW, ix = idx - r * a.IW;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < KW; ++t) {
            const int nx = ix + a.pw - t;                            // ix = ox sw + t - pw
            if (nx >= 0 && nx % a.sw == 0 && nx / a.sw < a.OW) acc += cs[(r * a.OW + nx / a.sw) * KW + t];
        }
        dxb[idx] = acc;
    }
}

// ws[z][t][co] = sum over this block's pixels of dy[p][co] * x[p_t]   (Cin == 1)
// FUSED: dy is not read but formed from dz, the recomputed y and the BatchNorm backward coefficients (see cin1_bn_bwd_kernel)
template <int CG, int KH, int KW, bool FUSED = false>
__global__ __launch_bounds__(256) void cin1_wgrad_kernel(const DirectArgs a, int pix_per_blk) {
    constexpr int PG = 256 / CG;
    constexpr int T = KH * KW;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [PG][T][Cout]
    const int tid = threadIdx.x, cg = tid % CG, pg = tid / CG;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wv[FUSED ? T : 1];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = bv, sh = bv, mu = bv, k0 = bv, k1 = bv;
    if constexpr (FUSED) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            wv[t][0] = a.w[(cg * 4 + 0) * T + t]; wv[t][1] = a.w[(cg * 4 + 1) * T + t];
            wv[t][2] = a.w[(cg * 4 + 2) * T + t]; wv[t][3] = a.w[(cg * 4 + 3) * T + t];
        }
        if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + cg * 4);
        sc = *reinterpret_cast<const f32x4*>(a.scale + cg * 4); sh = *reinterpret_cast<const f32x4*>(a.shift + cg * 4);
        mu = *reinterpret_cast<const f32x4*>(a.mean + cg * 4);
        k0 = *reinterpret_cast<const f32x4*>(a.sums + cg * 4); k1 = *reinterpret_cast<const f32x4*>(a.sums + a.Cout + cg * 4);
    }
    const int p0 = blockIdx.x * pix_per_blk;
    const int p1 = min(p0 + pix_per_blk, a.M);
    __shared__ float xp[CIN1_XP];
    const int oy_blk = (p0 / a.OW) % a.OH;
    if (a.xfast && p0 < a.M) cin1_stage_rows(a, xp, p0 / (a.OH * a.OW), oy_blk, threadIdx.x);      // (trailing blocks past the last pixel stage nothing)
    for (int p = p0 + pg; p < p1; p += PG) {
        int ox = p % a.OW; int r_ = p / a.OW; int oy = r_ % a.OH; int n = r_ / a.OH;
        f32x4 d = *reinterpret_cast<const f32x4*>((FUSED ? a.dz : a.dy) + (size_t)p * a.Cout + cg * 4);
        const float* xb = a.x + (size_t)n * a.IH * a.IW;
        const int iy0 = oy * a.sh, ix0 = ox * a.sw;
        float xt[T];
        if (a.xfast) {
            cin1_lds_taps<KH, KW>(xp + (oy - oy_blk) * a.sh * a.xpitch + ix0, a.xpitch, xt);
        } else
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            const int iy = iy0 + tap_dy(a, r);
            const bool yok = (unsigned)iy < (unsigned)a.IH;
#pragma unroll
            for (int q = 0; q < KW; ++q) {
                const int ix = ix0 + tap_dx(a, q);
                xt[r * KW + q] = (yok && (unsigned)ix < (unsigned)a.IW) ? cin1_x(a, xb, n, iy, ix) : 0.f;
            }
        }
        if constexpr (FUSED) {
            f32x4 v = bv;
#pragma unroll
            for (int t = 0; t < T; ++t) v += xt[t] * wv[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dp = d[e] * dact(v[e] * sc[e] + sh[e], a.act, a.slope);
                d[e] = sc[e] * dp + (k1[e] * (v[e] - mu[e]) + k0[e]);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] += d * xt[t];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(smem + ((size_t)pg * T + t) * a.Cout + cg * 4) = acc[t];
    __syncthreads();
    for (int i = tid; i < T * a.Cout; i += 256) {
        float s = 0.f;
        for (int k = 0; k < PG; ++k) s += smem[(size_t)k * T * a.Cout + i];
        a.ws[(size_t)blockIdx.x * T * a.Cout + i] = s;
    }
}

// ------------------------------------------------------------------ Cout == 1 (stride 1)
// y[p] = act(b + sum_t <x[p_t][:], wp[t][:]>),  LPP lanes per pixel, CPL float4 per lane per tap
template <int LPP, int CPL, int KH, int KW>
__global__ __launch_bounds__(256) void cout1_fwd_kernel(const DirectArgs a) {
    constexpr int T = KH * KW;
    const int cl = threadIdx.x % LPP;
    f32x4 wv[T][CPL];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < CPL; ++c) wv[t][c] = *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.Cin + (c * LPP + cl) * 4);
    const float bias = a.bias ? a.bias[0] : 0.f;
    constexpr int PPB = 256 / LPP;
    const int pstep = gridDim.x * PPB;
    for (int p = blockIdx.x * PPB + threadIdx.x / LPP; p < a.M + PPB; p += pstep) {
        float accv = 0.f;
        const bool live = p < a.M;
        if (live) {
            int ox = p % a.OW; int r_ = p / a.OW; int oy = r_ % a.OH; int n = r_ / a.OH;
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const int iy = oy + tap_dy(a, r);
                if ((unsigned)iy >= (unsigned)a.IH) continue;
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    const int ix = ox + tap_dx(a, q);
                    if ((unsigned)ix >= (unsigned)a.IW) continue;
                    const float* xr = a.x + ((size_t)(n * a.IH + iy) * a.IW + ix) * a.Cin;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        f32x4 xv = *reinterpret_cast<const f32x4*>(xr + (c * LPP + cl) * 4);
                        const f32x4 w = wv[r * KW + q][c];
                        accv += xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3];
                    }
                }
            }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) accv += __shfl_xor(accv, o, 64);
        if (live && cl == 0) a.y[p] = viai_act(accv + bias, a.act, a.slope);
        if (p >= a.M) break;
    }
}

// Row-run form for 3 x 3 / pad 1 (output width a multiple of L): a lane group computes L consecutive output pixels of one row from
// the (L + 2) x 3 input vectors it fetches up front (all loads in flight at once), so every input vector is fetched
// 3 (L + 2) / L times instead of nine, and the pixel index is decoded once per run.  FWD: tap offsets ascend with the tap index (Conv2d).
template <int LPP, int CPL, int L, bool FWD, bool BN = false>
__global__ __launch_bounds__(256) void cout1_fwd_run_kernel(const DirectArgs a) {
    constexpr int KH = 3, KW = 3;
    const int cl = threadIdx.x % LPP;
    f32x4 wv[KH * KW][CPL];
#pragma unroll
    for (int t = 0; t < KH * KW; ++t)
#pragma unroll
        for (int c = 0; c < CPL; ++c) wv[t][c] = *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.Cin + (c * LPP + cl) * 4);
    f32x4 bsc[CPL], bsh[CPL];                                       // BN: the input is act_in(scale * x + shift), zero outside the image
    if constexpr (BN) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            bsc[c] = *reinterpret_cast<const f32x4*>(a.scale + (c * LPP + cl) * 4);
            bsh[c] = *reinterpret_cast<const f32x4*>(a.shift + (c * LPP + cl) * 4);
        }
    }
    const float bias = a.bias ? a.bias[0] : 0.f;
    constexpr int GPB = 256 / LPP;                                  // lane groups (= runs in flight) per block
    const int rpr = a.OW / L, nruns = a.N * a.OH * rpr;
    for (int run = blockIdx.x * GPB + threadIdx.x / LPP; run < nruns; run += gridDim.x * GPB) {
        const int rx = run % rpr; int r_ = run / rpr; const int oy = r_ % a.OH, n = r_ / a.OH;
        const int ox0 = rx * L;
        const float* rowp[KH]; bool rowok[KH];
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            const int iy = oy + tap_dy(a, r);
            rowok[r] = (unsigned)iy < (unsigned)a.IH;
            rowp[r] = a.x + ((size_t)(n * a.IH + (rowok[r] ? iy : 0)) * a.IW) * a.Cin + cl * 4;
        }
        auto column = [&](f32x4 (&col)[KH][CPL], int ix) {           // input column ix of the three rows (zeros outside)
            const bool cok = (unsigned)ix < (unsigned)a.IW;
#pragma unroll
            for (int r = 0; r < KH; ++r)
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(rowp[r] + (size_t)(cok ? ix : 0) * a.Cin + c * LPP * 4);
                    if constexpr (BN) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = viai_act(v[e] * bsc[c][e] + bsh[c][e], a.act_in, a.slope);
                    }
                    col[r][c] = (cok & rowok[r]) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                }
        };
        f32x4 win[L + 2][KH][CPL];                                  // input columns ox0 - 1 .. ox0 + L, all requested up front
#pragma unroll
        for (int j = 0; j < L + 2; ++j) column(win[j], ox0 - 1 + j);
#pragma unroll
        for (int p = 0; p < L; ++p) {
            float accv = 0.f;
#pragma unroll
            for (int r = 0; r < KH; ++r)
#pragma unroll
                for (int q = 0; q < KW; ++q) {
                    const int j = FWD ? q : KW - 1 - q;              // window column of tap q: ix = ox + tap_dx(q)
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const f32x4 xv = win[p + j][r][c], w = wv[r * KW + q][c];
                        accv += xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3];
                    }
                }
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) accv += __shfl_xor(accv, o, 64);
            if (cl == 0) a.y[(size_t)(n * a.OH + oy) * a.OW + ox0 + p] = viai_act(accv + bias, a.act, a.slope);
        }
    }
}

// dx[q][ci] = sum_t dy[o_t(q)] * wp[t][ci]; thread = (pixel lane, fixed channel quad)
template <int KH, int KW>
__global__ __launch_bounds__(256) void cout1_dgrad_kernel(const DirectArgs a) {
    constexpr int T = KH * KW;
    const int c4n = a.Cin / 4;              // divides 256
    const int c4 = threadIdx.x % c4n;
    f32x4 wv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) wv[t] = *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.Cin + c4 * 4);
    const int ppb = 256 / c4n;
    const int totq = a.N * a.IH * a.IW;
    for (int q = blockIdx.x * ppb + threadIdx.x / c4n; q < totq; q += gridDim.x * ppb) {
        int ix = q % a.IW; int r_ = q / a.IW; int iy = r_ % a.IH; int n = r_ / a.IH;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            const int oy = iy - tap_dy(a, r);
            if ((unsigned)oy >= (unsigned)a.OH) continue;
#pragma unroll
            for (int s_ = 0; s_ < KW; ++s_) {
                const int ox = ix - tap_dx(a, s_);
                if ((unsigned)ox >= (unsigned)a.OW) continue;
                v += a.dy[(size_t)(n * a.OH + oy) * a.OW + ox] * wv[r * KW + s_];
            }
        }
        *reinterpret_cast<f32x4*>(a.dx + (size_t)q * a.Cin + c4 * 4) = v;
    }
}

// Row-run form of the Cout = 1 data gradient (3 x 3 / pad 1, width a multiple of L): the (L + 2) x 3 dy values of a run of L input
// pixels are fetched once, then every pixel is 9 float4 FMAs and one float4 store.
template <int L, bool FWD>
__global__ __launch_bounds__(256) void cout1_dgrad_run_kernel(const DirectArgs a) {
    constexpr int KH = 3, KW = 3, WN_ = L + KW - 1;
    const int c4n = a.Cin / 4;              // divides 256
    const int c4 = threadIdx.x % c4n;
    f32x4 wv[KH * KW];
#pragma unroll
    for (int t = 0; t < KH * KW; ++t) wv[t] = *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.Cin + c4 * 4);
    const int gpb = 256 / c4n;
    const int rpr = a.IW / L, nruns = a.N * a.IH * rpr;
    for (int run = blockIdx.x * gpb + threadIdx.x / c4n; run < nruns; run += gridDim.x * gpb) {
        const int rx = run % rpr; int r_ = run / rpr; const int iy = r_ % a.IH, n = r_ / a.IH;
        const int ix0 = rx * L;
        float win[KH][WN_];                                         // win[r][j] = dy(oy_r, ix0 - 1 + j)
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            const int oy = iy - tap_dy(a, r);
            const bool yok = (unsigned)oy < (unsigned)a.OH;
            const float* row = a.dy + (size_t)(n * a.OH + (yok ? oy : 0)) * a.OW;
#pragma unroll
            for (int j = 0; j < WN_; ++j) {
                const int ox = ix0 - 1 + j;
                const bool ok = yok & ((unsigned)ox < (unsigned)a.OW);
                const float v = row[ok ? ox : 0];
                win[r][j] = ok ? v : 0.f;
            }
        }
        f32x4* dst = reinterpret_cast<f32x4*>(a.dx + ((size_t)(n * a.IH + iy) * a.IW + ix0) * a.Cin + c4 * 4);
#pragma unroll
        for (int p = 0; p < L; ++p) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < KH; ++r)
#pragma unroll
                for (int s_ = 0; s_ < KW; ++s_) {
                    const int j = p + (FWD ? (KW - 1 - s_) : s_);    // ox = ix - tap_dx(s): column ox - ix0 + 1
                    v += wv[r * KW + s_] * win[r][j];
                }
            dst[(size_t)p * c4n] = v;
        }
    }
}

// ws[z][t][ci] = sum over this block's INPUT pixels q of x[q][ci] * dy[o_t(q)]
template <int KH, int KW>
__global__ __launch_bounds__(256) void cout1_wgrad_kernel(const DirectArgs a, int pix_per_blk) {
    constexpr int T = KH * KW;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [PG][T][Cin]
    const int CG = a.Cin / 4;                 // <= 256, divides 256
    const int PG = 256 / CG;
    const int tid = threadIdx.x, cg = tid % CG, pg = tid / CG;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int totq = a.N * a.IH * a.IW;
    const int q0 = blockIdx.x * pix_per_blk;
    const int q1 = min(q0 + pix_per_blk, totq);
    for (int q = q0 + pg; q < q1; q += PG) {
        int ix = q % a.IW; int r_ = q / a.IW; int iy = r_ % a.IH; int n = r_ / a.IH;
        f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + (size_t)q * a.Cin + cg * 4);
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            const int oy = iy - tap_dy(a, r);
            const bool yok = (unsigned)oy < (unsigned)a.OH;
#pragma unroll
            for (int s_ = 0; s_ < KW; ++s_) {
                const int ox = ix - tap_dx(a, s_);
                float d = (yok && (unsigned)ox < (unsigned)a.OW) ? a.dy[(size_t)(n * a.OH + oy) * a.OW + ox] : 0.f;
                acc[r * KW + s_] += xv * d;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(smem + ((size_t)pg * T + t) * a.Cin + cg * 4) = acc[t];
    __syncthreads();
    for (int i = tid; i < T * a.Cin; i += 256) {
        float s = 0.f;
        for (int k = 0; k < PG; ++k) s += smem[(size_t)k * T * a.Cin + i];
        a.ws[(size_t)blockIdx.x * T * a.Cin + i] = s;
    }
}

// Row-run form of the same sum (image width a multiple of L): a thread walks runs of L consecutive input pixels of one row.
// The (L + KW - 1) x KH dy values a run touches are fetched into registers once, so the pixel loop is one float4 load of x
// and KH*KW float4 FMAs per pixel -- no integer division, no conditional dy load per tap and pixel (the per-pixel form above
// issued 10 vector-memory instructions and ~100 index instructions per KB of x: 322 us on the 16 x 256 x 256 x 32 layer whose
// 134 MB stream in 27 us).  Thread (cg, pg) takes runs base + pg + PG * i; partial sums as above.
// FWD: tap offsets ascend with s (Conv2d); otherwise they descend (ConvTranspose2d) -- keeps every window index a constant
template <int KH, int KW, int L, bool FWD, bool BN = false>
__global__ __launch_bounds__(256) void cout1_wgrad_run_kernel(const DirectArgs a, int runs_per_lane) {
    constexpr int T = KH * KW, WN_ = L + KW - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [PG][T][Cin]
    const int CG = a.Cin / 4;
    const int PG = 256 / CG;
    const int tid = threadIdx.x, cg = tid % CG, pg = tid / CG;
    f32x4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BN) { bsc = *reinterpret_cast<const f32x4*>(a.scale + cg * 4); bsh = *reinterpret_cast<const f32x4*>(a.shift + cg * 4); }
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rpr = a.IW / L;                                       // runs per row
    const int nruns = a.N * a.IH * rpr;
    // window column j of a run holds dy at ox = ix0 + j - (KW - 1) + lo, where lo = min over taps of -tap_dx; tap s reads
    // column (pixel index) + off[s]
    const int dx_first = tap_dx(a, 0), dx_last = tap_dx(a, KW - 1);
    const int dxmax = dx_first > dx_last ? dx_first : dx_last;    // taps are dx_first + s * step, step = +-1
    for (int i = 0; i < runs_per_lane; ++i) {
        const int run = (blockIdx.x * runs_per_lane + i) * PG + pg;
        if (run >= nruns) break;
        const int rx = run % rpr; int r_ = run / rpr; const int iy = r_ % a.IH, n = r_ / a.IH;
        const int ix0 = rx * L;
        float win[KH][WN_];                                         // win[r][j] = dy(oy_r, ix0 - dxmax + j)
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            const int oy = iy - tap_dy(a, r);
            const bool yok = (unsigned)oy < (unsigned)a.OH;
            const float* row = a.dy + (size_t)(n * a.OH + (yok ? oy : 0)) * a.OW;
#pragma unroll
            for (int j = 0; j < WN_; ++j) {
                const int ox = ix0 - dxmax + j;
                const bool ok = yok & ((unsigned)ox < (unsigned)a.OW);
                const float v = row[ok ? ox : 0];
                win[r][j] = ok ? v : 0.f;
            }
        }
        const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + ((size_t)(n * a.IH + iy) * a.IW + ix0) * a.Cin + cg * 4);
#pragma unroll
        for (int p0 = 0; p0 < L; p0 += 4) {
            f32x4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = xp[(size_t)(p0 + u) * CG];
            if constexpr (BN) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[u][e] = viai_act(xv[u][e] * bsc[e] + bsh[e], a.act_in, a.slope);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < KH; ++r)
#pragma unroll
                    for (int s_ = 0; s_ < KW; ++s_) {
                        // ox = ix - tap_dx(s), column = ox - ix0 + dxmax: p + KW-1-s (ascending taps) or p + s (descending)
                        const int jo = FWD ? (KW - 1 - s_) : s_;
                        const int j = (p0 + u) + jo;
                        acc[r * KW + s_] += xv[u] * win[r][j];
                    }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(smem + ((size_t)pg * T + t) * a.Cin + cg * 4) = acc[t];
    __syncthreads();
    for (int i = tid; i < T * a.Cin; i += 256) {
        float s = 0.f;
        for (int k = 0; k < PG; ++k) s += smem[(size_t)k * T * a.Cin + i];
        a.ws[(size_t)blockIdx.x * T * a.Cin + i] = s;
    }
}

// The same forward for WIDE front layers (D.conv3 -> conv4: 512 channels on 64 x 32 maps) as two launches with a small tensor in between:
//   dots:   a lane group owns an input pixel: z = act_in(scale * y + shift) on load, the nine products c[q][t] = <z[q], w[t]> reduced over the
//           group, nine floats per pixel stored.  A plain grid-stride stream over y -- no row blocks, no halo rows read twice, U pixels in
//           flight per group -- where the LDS-plane kernel below had four lane groups per block walking twelve dependent load rounds.
//   gather: out[p] = act(bias + sum_t c[p + d_t][t]) in tap order (zero outside the image).
// Replaces bn_act_fwd (y -> z, 67 + 67 MB) + the row-run forward (z fetched 3 (L + 2) / L times through L1 / L2) of the two-launch route.
template <int LPP, int CPL>
__global__ __launch_bounds__(256) void cout1_pair_dots_kernel(const DirectArgs a, float* __restrict__ cbuf, int npix) {
    constexpr int T = 9, PPB = 256 / LPP, U = CPL >= 2 ? 4 : 8;
    const int tid = threadIdx.x, cl = tid % LPP, grp = tid / LPP;
    f32x4 wv[T][CPL], sc[CPL], sh[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        sc[c] = *reinterpret_cast<const f32x4*>(a.scale + (c * LPP + cl) * 4);
        sh[c] = *reinterpret_cast<const f32x4*>(a.shift + (c * LPP + cl) * 4);
#pragma unroll
        for (int t = 0; t < T; ++t) wv[t][c] = *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.Cin + (c * LPP + cl) * 4);
    }
    for (int q0 = (blockIdx.x * PPB + grp) * U; q0 < npix; q0 += gridDim.x * PPB * U) {
        f32x4 v[U][CPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* src = a.x + (size_t)min(q0 + u, npix - 1) * a.Cin + cl * 4;
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[u][c] = *reinterpret_cast<const f32x4*>(src + c * LPP * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float d[T];
#pragma unroll
            for (int t = 0; t < T; ++t) d[t] = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                f32x4 z;
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = viai_act(v[u][c][e] * sc[c][e] + sh[c][e], a.act_in, a.slope);
#pragma unroll
                for (int t = 0; t < T; ++t) d[t] += z[0] * wv[t][c][0] + z[1] * wv[t][c][1] + z[2] * wv[t][c][2] + z[3] * wv[t][c][3];
            }
#pragma unroll
            for (int t = 0; t < T; ++t) d[t] = group_sum<LPP>(d[t]);
            if (cl == 0 && q0 + u < npix) {
                float* dst = cbuf + (size_t)(q0 + u) * T;
#pragma unroll
                for (int t = 0; t < T; ++t) dst[t] = d[t];
            }
        }
    }
}
__global__ __launch_bounds__(256) void cout1_pair_gather_kernel(const DirectArgs a, const float* __restrict__ cbuf) {
    constexpr int KH = 3, KW = 3;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.M) return;
    const int ox = p % a.OW; const int r_ = p / a.OW; const int oy = r_ % a.OH, n = r_ / a.OH;
    float s = a.bias ? a.bias[0] : 0.f;
#pragma unroll
    for (int r = 0; r < KH; ++r) {
        const int iy = oy + tap_dy(a, r);
        if ((unsigned)iy >= (unsigned)a.IH) continue;
#pragma unroll
        for (int q = 0; q < KW; ++q) {
            const int ix = ox + tap_dx(a, q);
            if ((unsigned)ix < (unsigned)a.IW) s += cbuf[((size_t)(n * a.IH + iy) * a.IW + ix) * (KH * KW) + r * KW + q];
        }
    }
    a.y[p] = viai_act(s, a.act, a.slope);
}

// Forward of the Cout = 1 layer of a fused pair, every input element read ONCE: z = act_in(scale * y + shift) is formed when a lane
// group loads its input pixel, the pixel's nine partial dot products d[t] = <z, w[t]> are reduced across the group and stored into nine
// LDS planes at the output pixel each belongs to; after a barrier an output is bias + the sum of its nine planes in tap order
// (deterministic: every (tap, output) cell has exactly one writer, cells whose input pixel lies outside the image stay zero).
// Block = R output rows x the full width of one image; it walks the R + 2 input rows those outputs touch.  (The row-run kernel above
// fetches every input element 3 (L + 2) / L times and, with BatchNorm on load, normalised it as often: 36 -> 106 us on D.conv4.)
template <int LPP, int CPL, bool FWD>
__global__ __launch_bounds__(256) void cout1_pair_fwd_kernel(const DirectArgs a, int R) {
    constexpr int KH = 3, KW = 3, PPB = 256 / LPP;
    extern __shared__ __attribute__((aligned(16))) float part[];      // [9][R][W]
    const int tid = threadIdx.x, cl = tid % LPP, grp = tid / LPP;
    const int W = a.IW, bpi = (a.IH + R - 1) / R;
    const int n = blockIdx.x / bpi, r0 = (blockIdx.x - n * bpi) * R;
    const int rows = min(R, a.OH - r0);
    const int plane = R * W;
    for (int i = tid; i < 9 * plane; i += 256) part[i] = 0.f;
    f32x4 wv[KH * KW][CPL], sc[CPL], sh[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        sc[c] = *reinterpret_cast<const f32x4*>(a.scale + (c * LPP + cl) * 4);
        sh[c] = *reinterpret_cast<const f32x4*>(a.shift + (c * LPP + cl) * 4);
#pragma unroll
        for (int t = 0; t < KH * KW; ++t) wv[t][c] = *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.Cin + (c * LPP + cl) * 4);
    }
    __syncthreads();
    // input rows r0 - 1 .. r0 + rows (inside the image), all columns: pixel index q over (rows + 2) x W
    const int iy_lo = max(r0 - 1, 0), iy_hi = min(r0 + rows, a.IH - 1);
    const int npix = (iy_hi - iy_lo + 1) * W;
    constexpr int U = CPL >= 2 ? 4 : 8;                     // input pixels in flight per lane group (a block is 4 .. 32 groups: latency, not issue, bounds it)
    for (int q0 = 0; q0 < npix; q0 += U * PPB) {
        f32x4 v[U][CPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = min(q0 + u * PPB + grp, npix - 1);
            const float* src = a.x + ((size_t)(n * a.IH + iy_lo) * W + q) * a.Cin + cl * 4;
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[u][c] = *reinterpret_cast<const f32x4*>(src + c * LPP * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * PPB + grp;
            const int iy = iy_lo + q / W, ix = q % W;
            float d[KH * KW];
#pragma unroll
            for (int t = 0; t < KH * KW; ++t) d[t] = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                f32x4 z;
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = viai_act(v[u][c][e] * sc[c][e] + sh[c][e], a.act_in, a.slope);
#pragma unroll
                for (int t = 0; t < KH * KW; ++t) d[t] += z[0] * wv[t][c][0] + z[1] * wv[t][c][1] + z[2] * wv[t][c][2] + z[3] * wv[t][c][3];
            }
#pragma unroll
            for (int t = 0; t < KH * KW; ++t) d[t] = group_sum<LPP>(d[t]);
            if (cl == 0 && q < npix) {
#pragma unroll
                for (int r = 0; r < KH; ++r) {
                    const int oy = iy - tap_dy(a, r) - r0;              // output row (within the block) this input row feeds through tap row r
                    if ((unsigned)oy >= (unsigned)rows) continue;
#pragma unroll
                    for (int s_ = 0; s_ < KW; ++s_) {
                        const int ox = ix - tap_dx(a, s_);
                        if ((unsigned)ox < (unsigned)W) part[(r * KW + s_) * plane + oy * W + ox] = d[r * KW + s_];
                    }
                }
            }
        }
    }
    __syncthreads();
    const float bias = a.bias ? a.bias[0] : 0.f;
    for (int i = tid; i < rows * W; i += 256) {
        float s = bias;
#pragma unroll
        for (int t = 0; t < KH * KW; ++t) s += part[t * plane + i];
        a.y[(size_t)(n * a.OH + r0) * W + i] = viai_act(s, a.act, a.slope);
    }
}

__device__ __forceinline__ float pair_act_grad(float pre, int act, float slope) {
    if (act == VIAI_ACT_SIGMOID) { float s = 1.f / (1.f + __expf(-pre)); return s * (1.f - s); }
    return viai_act_grad_pl(pre, act, slope);
}

// BatchNorm + activation backward of the layer in FRONT of a Cout = 1 3 x 3 conv, with that conv's data gradient formed on the fly:
// dz[q][c] = sum_t du[o_t(q)] * w[t][c] is nine float4 FMAs from an (L + 2) x 3 register window of the one-channel du -- cheaper than
// the 4 bytes per element it costs to write dz and the 8 to read it back twice (G.conv6_1 -> conv6_2: 3 x 134 MB per step; D.conv3 ->
// conv4: 3 x 67 MB per D pass).  Thread = (run lane, channel quad), runs of L consecutive pixels of one row as in the kernels above.
//   APPLY = false: partial sums part[blk][0][c] = sum dp, part[blk][1][c] = sum dp * xhat  (-> bn_bwd_final_kernel, bn.hip)
//   APPLY = true : dy = scale * dp + k1 * (y - mean) + k0, and max |dy|
// a.x = y (pre-BatchNorm tensor of the front layer), a.dy = du, a.w = the Cout = 1 layer's [tap][channel] weight image.
template <int L, bool FWD, bool APPLY>
__global__ __launch_bounds__(256) void cout1_bn_bwd_rows_kernel(const DirectArgs a, int R) {
    constexpr int KH = 3, KW = 3, WN_ = L + KW - 1;
    extern __shared__ __attribute__((aligned(16))) float tile[];     // du rows r0 - 1 .. r0 + R, columns -1 .. IW (zeros outside the image)
    __shared__ f32x4 r1[256], r2[256];
    __shared__ f32x4 r3[APPLY ? 1 : 256];
    const int CG = a.Cin / 4;               // divides 256
    const int PG = 256 / CG;
    const int tid = threadIdx.x, cg = tid % CG, pg = tid / CG;
    float pS = 1.f, pL = 0.f;
    if constexpr (APPLY) {
        if (a.dy_p16) {
            float b = 0.f;
            for (int c = tid; c < a.Cin; c += 256) b = fmaxf(b, a.k_sums[2 * a.Cin + c]);
            const float bound = block_max_all(b);
            if (blockIdx.x == 0 && tid == 0) *a.amax = bound;
            pS = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(f16_scale_from_amax_value(bound)))); pL = f16_clamp_for_scale(pS);
        }
    }
    const int bpi = a.IH / R;
    const int n = blockIdx.x / bpi, r0 = (blockIdx.x - n * bpi) * R;
    const int pitch = a.IW + 2;
    for (int idx = tid; idx < (R + 2) * pitch; idx += 256) {
        const int rr = idx / pitch, cc = idx - rr * pitch;
        const int oy = r0 - 1 + rr, ox = cc - 1;
        tile[idx] = ((unsigned)oy < (unsigned)a.OH && (unsigned)ox < (unsigned)a.OW) ? a.dy[(size_t)(n * a.OH + oy) * a.OW + ox] : 0.f;
    }
    f32x4 wv[KH * KW];
#pragma unroll
    for (int t = 0; t < KH * KW; ++t) wv[t] = *reinterpret_cast<const f32x4*>(a.w + (size_t)t * a.Cin + cg * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + cg * 4), sh = *reinterpret_cast<const f32x4*>(a.shift + cg * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cg * 4);
    f32x4 is = {0.f, 0.f, 0.f, 0.f}, k0 = is, k1 = is;
    if constexpr (APPLY) { k0 = *reinterpret_cast<const f32x4*>(a.k_sums + cg * 4); k1 = *reinterpret_cast<const f32x4*>(a.k_sums + a.Cin + cg * 4); }
    else is = *reinterpret_cast<const f32x4*>(a.invstd + cg * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
    float mx = 0.f;
    __syncthreads();
    const int rpr = a.IW / L;
    for (int run = pg; run < R * rpr; run += PG) {
        const int rrow = run / rpr, ix0 = (run - rrow * rpr) * L;
        float win[KH][WN_];                                         // win[r][j] = du(iy - tap_dy(r), ix0 - 1 + j); one value per VGPR (see cin1_lds_taps)
#pragma unroll
        for (int r = 0; r < KH; ++r)
#pragma unroll
            for (int j = 0; j < WN_; ++j) {
                win[r][j] = tile[(rrow + 1 - tap_dy(a, r)) * pitch + ix0 + j];
                asm volatile("" : "+v"(win[r][j]));
            }
        const size_t base = ((size_t)(n * a.IH + r0 + rrow) * a.IW + ix0) * CG + cg;
        const f32x4* yp = reinterpret_cast<const f32x4*>(a.x) + base;
        f32x4* dst = reinterpret_cast<f32x4*>(a.dx) + base;
#pragma unroll
        for (int p0 = 0; p0 < L; p0 += 4) {
            f32x4 yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) yv[u] = yp[(size_t)(p0 + u) * CG];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 dz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < KH; ++r)
#pragma unroll
                    for (int s_ = 0; s_ < KW; ++s_) dz += wv[r * KW + s_] * win[r][(p0 + u) + (FWD ? (KW - 1 - s_) : s_)];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dp = dz[e] * pair_act_grad(yv[u][e] * sc[e] + sh[e], a.act_in, a.slope);
                    if constexpr (APPLY) {
                        o[e] = sc[e] * dp + (k1[e] * (yv[u][e] - mu[e]) + k0[e]);
                        mx = fmaxf(mx, fabsf(o[e]));
                    } else {
                        s1[e] += dp;
                        s2[e] += dp * (yv[u][e] - mu[e]) * is[e];
                        s3[e] = fmaxf(s3[e], fabsf(dp));
                    }
                }
                if constexpr (APPLY) {
                    if (a.dy_p16) p16_store_quad(reinterpret_cast<float*>(dst - cg) + (size_t)(p0 + u) * a.Cin, cg, o, pS, pL);
                    else dst[(size_t)(p0 + u) * CG] = o;
                }
            }
        }
    }
    if constexpr (APPLY) {
        if (a.amax != nullptr && !a.dy_p16) block_absmax_to(a.amax, mx);
    } else {
        r1[tid] = s1; r2[tid] = s2; r3[tid] = s3;
        __syncthreads();
        if (tid < CG) {
            const int ps = a.dy_p16 ? 3 : 2;
            f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f}, t3 = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < PG; ++k) {
                t1 += r1[k * CG + tid]; t2 += r2[k * CG + tid];
                const f32x4 m = r3[k * CG + tid];
                for (int e = 0; e < 4; ++e) t3[e] = fmaxf(t3[e], m[e]);
            }
            *reinterpret_cast<f32x4*>(a.part + ((size_t)blockIdx.x * ps + 0) * a.Cin + tid * 4) = t1;
            *reinterpret_cast<f32x4*>(a.part + ((size_t)blockIdx.x * ps + 1) * a.Cin + tid * 4) = t2;
            if (a.dy_p16) *reinterpret_cast<f32x4*>(a.part + ((size_t)blockIdx.x * 3 + 2) * a.Cin + tid * 4) = t3;
        }
    }
}

// dw[co*s_co + ci*s_ci + t] (+)= sum_z ws[z][t][co][ci];  (256 / ZL) outputs x ZL z-lanes per block, fixed order.
// ZL = 4 for the big weight tensors (bandwidth-bound: 64 consecutive outputs per load instruction); ZL = 32 when there are
// few outputs but many slabs (the Cin = 1 / Cout = 1 / 32-channel layers: 288 ... 9216 outputs x up to 2048 slabs, where four
// lanes walking 512 slabs each took longer than the gradient kernel itself).
template <int ZL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nz, int T,
                                                           int Cout, int Cin, long s_co, long s_ci, int accumulate) {
    constexpr int NO = 256 / ZL;
    __shared__ float red[ZL][NO];
    const long total = (long)T * Cout * Cin;
    const int il = threadIdx.x % NO, zl = threadIdx.x / NO;
    const long i = blockIdx.x * (long)NO + il;
    float s = 0.f;
    if (i < total) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = zl;
        for (; z + 3 * ZL < nz; z += 4 * ZL) {
            s0 += ws[(size_t)z * total + i];
            s1 += ws[(size_t)(z + ZL) * total + i];
            s2 += ws[(size_t)(z + 2 * ZL) * total + i];
            s3 += ws[(size_t)(z + 3 * ZL) * total + i];
        }
        for (; z < nz; z += ZL) s0 += ws[(size_t)z * total + i];
        s = (s0 + s1) + (s2 + s3);
    }
    red[zl][il] = s;
    __syncthreads();
    if (zl == 0 && i < total) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < ZL; k += 4) s += (red[k][il] + red[k + 1][il]) + (red[k + 2][il] + red[k + 3][il]);
        int ci = (int)(i % Cin); long r = i / Cin; int co = (int)(r % Cout); int t = (int)(r / Cout);
        long o = co * s_co + ci * s_ci + t;
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

// out[c] (+)= sum_p x[p][c]: per-block partial sums over a row range.  C % 4 == 0: float4 loads, 256 threads =
// channel quads x pixel lanes, four rows in flight per lane (HBM-bound streaming reduce); otherwise a scalar sweep.
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                          long M, int C, long rows_per_blk) {
    __shared__ f32x4 red4[256];
    const int tid = threadIdx.x;
    const long r0 = blockIdx.x * rows_per_blk;
    long r1 = r0 + rows_per_blk; if (r1 > M) r1 = M;
    if ((C & 3) == 0) {
        const int CG = C / 4;
        for (int g0 = 0; g0 < CG; g0 += 256) {
            const int cgw = min(256, CG - g0);
            const int pg = 256 / cgw;
            const int cg = tid % cgw, pl = tid / cgw;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (pl < pg) {
                const float* col = x + (size_t)(g0 + cg) * 4;
                for (long r = r0 + pl; r < r1; r += 4L * pg) {
                    f32x4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        long rr = r + (long)u * pg;
                        v[u] = rr < r1 ? *reinterpret_cast<const f32x4*>(col + rr * C) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    s += (v[0] + v[1]) + (v[2] + v[3]);
                }
            }
            red4[tid] = s;
            __syncthreads();
            if (tid < cgw) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < pg; ++k) t += red4[k * cgw + tid];
                *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + (size_t)(g0 + tid) * 4) = t;
            }
            __syncthreads();
        }
        return;
    }
    float* red = reinterpret_cast<float*>(red4);
    for (int c0 = 0; c0 < C; c0 += 256) {
        int cw = min(256, C - c0);
        int rl = 256 / cw;                // row lanes
        int c = tid % cw, rr = tid / cw;
        float s = 0.f;
        if (rr < rl)
            for (long r = r0 + rr; r < r1; r += rl) s += x[r * C + c0 + c];
        red[tid] = (rr < rl) ? s : 0.f;
        __syncthreads();
        if (tid < cw) {
            float t = 0.f;
            for (int k = 0; k < rl; ++k) t += red[k * cw + tid];
            part[(size_t)blockIdx.x * C + c0 + tid] = t;
        }
        __syncthreads();
    }
}

// fixed-order fp64 sum of the block partials: 16 channels x 16 partial lanes per block
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int C, int accumulate) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < C)
        for (int b = pl; b < nblk; b += 16) s += (double)part[(size_t)b * C + c];
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < C) {
        for (int k = 1; k < 16; ++k) s += red[k][cl];
        out[c] = accumulate ? out[c] + (float)s : (float)s;
    }
}

DirectArgs make_args(const viai_conv2d* c) {
    DirectArgs a{};
    a.N = c->N; a.IH = c->IH; a.IW = c->IW; a.Cin = c->C1 + c->C2; a.Cout = c->Cout;
    a.kh = c->kh; a.kw = c->kw; a.sh = c->sh; a.sw = c->sw; a.ph = c->ph; a.pw = c->pw;
    a.transposed = c->transposed;
    viai_conv2d_out_hw(c, &a.OH, &a.OW);
    a.M = a.N * a.OH * a.OW;
    return a;
}

}  // namespace

// ---- entry points used by conv_api.hip ------------------------------------
// window dispatch: the VIAI layers use 3x3 and 1x4; others are rejected (hipErrorInvalidValue)
#define VIAI_WINDOW_DISPATCH(c, CALL)                                  \
    do {                                                               \
        if ((c)->kh == 3 && (c)->kw == 3) { CALL(3, 3); }              \
        else if ((c)->kh == 1 && (c)->kw == 4) { CALL(1, 4); }         \
        else if ((c)->kh == 1 && (c)->kw == 1) { CALL(1, 1); }         \
        else if ((c)->kh == 1 && (c)->kw == 3) { CALL(1, 3); }         \
        else if ((c)->kh == 1 && (c)->kw == 6) { CALL(1, 6); }         \
        else return (int)hipErrorInvalidValue;                         \
    } while (0)

int viai_cin1_fwd(const viai_conv2d* c, const float* x, const float* w, const float* bias, float* y,
                  float* stat, int act, hipStream_t st) {
    DirectArgs a = make_args(c);
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stat = stat; a.act = act; a.slope = 0.2f;
    a.nblk = (a.M + CIN1_PB - 1) / CIN1_PB;
    cin1_rows_ok(a, CIN1_PB, c->kh, c->kw);
#define CALL(KH, KW)                                                                                             \
    if (c->Cout == 32) VIAI_LAUNCH((cin1_fwd_kernel<8, KH, KW>), dim3(a.nblk), dim3(256), 0, st, a);            \
    else if (c->Cout == 64) VIAI_LAUNCH((cin1_fwd_kernel<16, KH, KW>), dim3(a.nblk), dim3(256), 0, st, a);      \
    else if (c->Cout == 128) VIAI_LAUNCH((cin1_fwd_kernel<32, KH, KW>), dim3(a.nblk), dim3(256), 0, st, a);     \
    else return (int)hipErrorInvalidValue
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    return viai_launch_status();
}

int viai_cin1_stat_geom(const viai_conv2d* c, int* nblk, int* rows) {
    DirectArgs a = make_args(c);
    *nblk = (a.M + CIN1_PB - 1) / CIN1_PB; *rows = CIN1_PB;
    return 0;
}

int viai_cin1_dgrad(const viai_conv2d* c, const float* dy, const float* w, float* dx, hipStream_t st) {
    DirectArgs a = make_args(c);
    a.dy = dy; a.w = w; a.dx = dx;
    long tot = (long)a.N * a.IH * a.IW;
    int lpp = c->Cout / 4;
    long nb = (tot * lpp + 255) / 256;
    // 4096 blocks measured best on D.conv1 (2048 ... 8192 within 4 %; 32768+ slower: every block reloads the filter)
    constexpr long cap = 4096;
    if (nb > cap) nb = cap;
    int blocks = (int)nb;
    if (c->kh == 1 && c->kw == 4 && c->sh == 1 && c->sw == 2 && lpp == 16) {          // D.conv1
        VIAI_LAUNCH((cin1_dgrad_kernel<16, 1, 4, 1, 2>), dim3(blocks), dim3(256), 0, st, a);
        return viai_launch_status();
    }
    if (c->kh == 3 && c->kw == 3 && c->sh == 2 && c->sw == 2 && lpp == 8) {           // E.conv1
        VIAI_LAUNCH((cin1_dgrad_kernel<8, 3, 3, 2, 2>), dim3(blocks), dim3(256), 0, st, a);
        return viai_launch_status();
    }
#define CALL(KH, KW)                                                                                            \
    if (lpp == 8) VIAI_LAUNCH((cin1_dgrad_kernel<8, KH, KW>), dim3(blocks), dim3(256), 0, st, a);               \
    else if (lpp == 16) VIAI_LAUNCH((cin1_dgrad_kernel<16, KH, KW>), dim3(blocks), dim3(256), 0, st, a);        \
    else if (lpp == 32) VIAI_LAUNCH((cin1_dgrad_kernel<32, KH, KW>), dim3(blocks), dim3(256), 0, st, a);        \
    else return (int)hipErrorInvalidValue
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    return viai_launch_status();
}

static int direct_wgrad_blocks(long pixels) {
    long b = (pixels + 127) / 128;       // >= 128 pixels per block, up to 2048 partial slabs
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

int viai_wgrad_reduce(const float* ws, float* dw, int nz, int T, int Cout, int Cin, long s_co, long s_ci,
                      int accumulate, hipStream_t st) {
    long total = (long)T * Cout * Cin;
    if (total <= 16384 && nz >= 128)
        VIAI_LAUNCH(wgrad_reduce_kernel<32>, dim3((int)((total + 7) / 8)), dim3(256), 0, st, ws, dw, nz, T, Cout, Cin, s_co, s_ci, accumulate);
    else
        VIAI_LAUNCH(wgrad_reduce_kernel<4>, dim3((int)((total + 63) / 64)), dim3(256), 0, st, ws, dw, nz, T, Cout, Cin, s_co, s_ci, accumulate);
    return viai_launch_status();
}

size_t viai_cin1_wgrad_ws_floats(const viai_conv2d* c) {
    DirectArgs a = make_args(c);
    return (size_t)direct_wgrad_blocks(a.M) * c->kh * c->kw * c->Cout;
}

int viai_cin1_wgrad(const viai_conv2d* c, const float* x, const float* dy, float* ws, float* dw, int accumulate, hipStream_t st) {
    DirectArgs a = make_args(c);
    a.x = x; a.dy = dy; a.ws = ws;
    const int T = c->kh * c->kw;
    int nb = direct_wgrad_blocks(a.M);
    int ppb = (a.M + nb - 1) / nb;
    cin1_rows_ok(a, ppb, c->kh, c->kw, 4);
    int cg = c->Cout / 4;
    size_t lds = (size_t)(256 / cg) * T * c->Cout * sizeof(float);
#define CALL(KH, KW)                                                                                                  \
    if (cg == 8) VIAI_LAUNCH((cin1_wgrad_kernel<8, KH, KW>), dim3(nb), dim3(256), lds, st, a, ppb);                   \
    else if (cg == 16) VIAI_LAUNCH((cin1_wgrad_kernel<16, KH, KW>), dim3(nb), dim3(256), lds, st, a, ppb);            \
    else if (cg == 32) VIAI_LAUNCH((cin1_wgrad_kernel<32, KH, KW>), dim3(nb), dim3(256), lds, st, a, ppb);            \
    else return (int)hipErrorInvalidValue
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    int e = viai_launch_status();
    if (e) return e;
    // torch layout [Cout][1][kh][kw] -> co*T + t
    return viai_wgrad_reduce(ws, dw, nb, T, c->Cout, 1, T, T, accumulate, st);
}

// ---- fused Cin = 1 conv + BatchNorm(train) + activation layer (E.conv1, D.conv1): entry points of the public ABI ----------------
int viai_bn_bwd_final_launch(const float* part, int nblk, int C, long M, const float* mean, const float* invstd, const float* scale,
                             int training, float* sums, float* dgamma, float* dbeta, int accumulate, hipStream_t st, int ps = 2);

extern "C" int viai_conv2d_cin1_bn_ok(const viai_conv2d* c) {
    if (c == nullptr || c->C1 + c->C2 != 1 || c->transposed) return 0;
    if (!(c->Cout == 32 || c->Cout == 64 || c->Cout == 128)) return 0;
    const bool win = (c->kh == 3 && c->kw == 3) || (c->kh == 1 && c->kw == 4) || (c->kh == 1 && c->kw == 1) || (c->kh == 1 && c->kw == 3) ||
                     (c->kh == 1 && c->kw == 6);
    return win;
}

// z == NULL: BatchNorm partials only (the conv output is not stored);  z != NULL: z = act(scale * conv(x) + shift)
static int cin1_bn_fwd_impl(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, float* stat_part,
                            const float* scale, const float* shift, float* z, int act, float* z_amax, void* stream,
                            int p16, const float* gamma, const float* beta, long m_stat);
extern "C" int viai_conv2d_cin1_bn_fwd(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, float* stat_part,
                                       const float* scale, const float* shift, float* z, int act, float* z_amax, void* stream) {
    return cin1_bn_fwd_impl(c, x, x_mask, w, bias, stat_part, scale, shift, z, act, z_amax, stream, 0, nullptr, nullptr, 0);
}
// (ABI 13) the apply pass writing z pre-split (P16); gamma / beta / m_stat give the bound stored in *z_amax.  Cout = 32 / 64 / 128
extern "C" int viai_conv2d_cin1_bn_fwd_p16(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias,
                                           const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                           float* z, int act, float* z_amax, void* stream) {
    if (z == nullptr || z_amax == nullptr || m_stat < 1 || act == VIAI_ACT_SIGMOID) return (int)hipErrorInvalidValue;
    return cin1_bn_fwd_impl(c, x, x_mask, w, bias, nullptr, scale, shift, z, act, z_amax, stream, 1, gamma, beta, m_stat);
}
static int cin1_bn_fwd_impl(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, float* stat_part,
                            const float* scale, const float* shift, float* z, int act, float* z_amax, void* stream,
                            int p16, const float* gamma, const float* beta, long m_stat) {
    if (!viai_conv2d_cin1_bn_ok(c) || (z == nullptr) == (stat_part == nullptr)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    viai_tag_reset();
    viai_tag_kernel("direct");
    DirectArgs a = make_args(c);
    a.x = x; a.w = w; a.bias = bias; a.y = z; a.stat = stat_part; a.scale = scale; a.shift = shift; a.act = act; a.slope = 0.2f;
    a.zmax = z_amax; a.xmask = x_mask;
    a.z_p16 = p16; a.gamma = gamma; a.beta = beta; a.p16_rad = p16 ? sqrtf((float)(m_stat > 1 ? m_stat - 1 : 1)) : 0.f;
    a.nblk = (a.M + CIN1_PB - 1) / CIN1_PB;
    cin1_rows_ok(a, CIN1_PB, c->kh, c->kw);
#define CALL(KH, KW)                                                                                                               \
    if (z == nullptr && KH * KW <= 4) {           /* tap-covariance statistics: 25.7 -> 9.5 us on D.conv1 (1 x 4); the 3 x 3 window (45 moments, 18 scattered loads per pixel: 18.6 against 14.4 us on E.conv1) keeps the conv-then-reduce pass */ \
        VIAI_LAUNCH((cin1_stats_cov_kernel<KH, KW>), dim3(a.nblk), dim3(256), 0, st, a);                                          \
    } else if (z == nullptr) {                                                                                                     \
        if (c->Cout == 32) VIAI_LAUNCH((cin1_fwd_kernel<8, KH, KW, 1>), dim3(a.nblk), dim3(256), 0, st, a);                       \
        else if (c->Cout == 64) VIAI_LAUNCH((cin1_fwd_kernel<16, KH, KW, 1>), dim3(a.nblk), dim3(256), 0, st, a);                 \
        else VIAI_LAUNCH((cin1_fwd_kernel<32, KH, KW, 1>), dim3(a.nblk), dim3(256), 0, st, a);                                    \
    } else if (z_amax != nullptr && a.nblk >= 512 && !p16) {                                                                                                       \
        if (c->Cout == 32) VIAI_LAUNCH((cin1_fwd_kernel<8, KH, KW, 2, 1024>), dim3((a.nblk + 3) / 4), dim3(1024), 0, st, a);      \
        else if (c->Cout == 64) VIAI_LAUNCH((cin1_fwd_kernel<16, KH, KW, 2, 1024>), dim3((a.nblk + 3) / 4), dim3(1024), 0, st, a);\
        else VIAI_LAUNCH((cin1_fwd_kernel<32, KH, KW, 2, 1024>), dim3((a.nblk + 3) / 4), dim3(1024), 0, st, a);                   \
    } else {                                                                                                                       \
        if (c->Cout == 32) VIAI_LAUNCH((cin1_fwd_kernel<8, KH, KW, 2>), dim3(a.nblk), dim3(256), 0, st, a);                       \
        else if (c->Cout == 64) VIAI_LAUNCH((cin1_fwd_kernel<16, KH, KW, 2>), dim3(a.nblk), dim3(256), 0, st, a);                 \
        else VIAI_LAUNCH((cin1_fwd_kernel<32, KH, KW, 2>), dim3(a.nblk), dim3(256), 0, st, a);                                    \
    }
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    return viai_launch_status();
}

// BatchNorm(train) + activation backward of the fused layer: partial sums from (dz, recomputed y) -> k0 / k1, dgamma, dbeta (the
// same final kernel as viai_bn_act_bwd); dy (optional) is written only when a data gradient needs it in memory.
// part: 2 * Cout * ceil(M / 256) floats; sums: 2 * Cout floats (read by viai_conv2d_cin1_bn_wgrad).
extern "C" int viai_conv2d_cin1_bn_bwd(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, const float* dz,
                                       const float* mean, const float* invstd, const float* scale, const float* shift, float* part,
                                       float* sums, float* dgamma, float* dbeta, float* dy, int act, int training, void* stream) {
    if (!viai_conv2d_cin1_bn_ok(c)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    DirectArgs a = make_args(c);
    a.xmask = x_mask;
    a.x = x; a.w = w; a.bias = bias; a.dz = dz; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.part = part;
    a.sums = sums; a.dx = dy; a.act = act; a.slope = 0.2f;
    a.nblk = (a.M + CIN1_PB - 1) / CIN1_PB;
    cin1_rows_ok(a, CIN1_PB, c->kh, c->kw, 2);
#define CALL(KH, KW)                                                                                                               \
    if (c->Cout == 32) VIAI_LAUNCH((cin1_bn_bwd_kernel<8, KH, KW, false>), dim3(a.nblk), dim3(256), 0, st, a);                    \
    else if (c->Cout == 64) VIAI_LAUNCH((cin1_bn_bwd_kernel<16, KH, KW, false>), dim3(a.nblk), dim3(256), 0, st, a);              \
    else VIAI_LAUNCH((cin1_bn_bwd_kernel<32, KH, KW, false>), dim3(a.nblk), dim3(256), 0, st, a)
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    int e = viai_launch_status();
    if (e) return e;
    e = viai_bn_bwd_final_launch(part, a.nblk, c->Cout, a.M, mean, invstd, scale, training & 1, sums, dgamma, dbeta, (training >> 1) & 1, st);
    if (e || dy == nullptr) return e;
#define CALL(KH, KW)                                                                                                               \
    if (c->Cout == 32) VIAI_LAUNCH((cin1_bn_bwd_kernel<8, KH, KW, true>), dim3(a.nblk), dim3(256), 0, st, a);                     \
    else if (c->Cout == 64) VIAI_LAUNCH((cin1_bn_bwd_kernel<16, KH, KW, true>), dim3(a.nblk), dim3(256), 0, st, a);               \
    else VIAI_LAUNCH((cin1_bn_bwd_kernel<32, KH, KW, true>), dim3(a.nblk), dim3(256), 0, st, a)
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    return viai_launch_status();
}

// dw (+)= weight gradient of the fused layer from dz: dy = scale * dp + k1 * (y - mean) + k0 is formed per element inside the kernel
// ws: viai_conv2d_wgrad_ws_bytes(c) bytes
extern "C" int viai_conv2d_cin1_bn_wgrad(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, const float* dz,
                                         const float* mean, const float* scale, const float* shift, const float* sums, float* ws,
                                         float* dw, int accumulate, int act, void* stream) {
    if (!viai_conv2d_cin1_bn_ok(c)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    viai_tag_reset();
    viai_tag_kernel("direct");
    DirectArgs a = make_args(c);
    a.xmask = x_mask;
    a.x = x; a.w = w; a.bias = bias; a.dz = dz; a.mean = mean; a.scale = scale; a.shift = shift; a.sums = sums; a.ws = ws;
    a.act = act; a.slope = 0.2f;
    const int T = c->kh * c->kw;
    int nb = direct_wgrad_blocks(a.M);
    int ppb = (a.M + nb - 1) / nb;
    cin1_rows_ok(a, ppb, c->kh, c->kw, 4);
    int cg = c->Cout / 4;
    size_t lds = (size_t)(256 / cg) * T * c->Cout * sizeof(float);
#define CALL(KH, KW)                                                                                                               \
    if (cg == 8) VIAI_LAUNCH((cin1_wgrad_kernel<8, KH, KW, true>), dim3(nb), dim3(256), lds, st, a, ppb);                         \
    else if (cg == 16) VIAI_LAUNCH((cin1_wgrad_kernel<16, KH, KW, true>), dim3(nb), dim3(256), lds, st, a, ppb);                  \
    else VIAI_LAUNCH((cin1_wgrad_kernel<32, KH, KW, true>), dim3(nb), dim3(256), lds, st, a, ppb)
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    int e = viai_launch_status();
    if (e) return e;
    return viai_wgrad_reduce(ws, dw, nb, T, c->Cout, 1, T, T, accumulate, st);
}

// dx = data gradient of the fused layer straight from dz (no dy tensor): the frozen-D pass of the G step
extern "C" int viai_conv2d_cin1_bn_dgrad(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* dz, const float* mean,
                                         const float* scale, const float* shift, const float* sums, float* dx, int act, void* stream) {
    if (!viai_conv2d_cin1_bn_ok(c)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    DirectArgs a = make_args(c);
    a.xmask = x_mask;
    a.x = x; a.w = w; a.dz = dz; a.mean = mean; a.scale = scale; a.shift = shift; a.sums = sums; a.dx = dx; a.act = act; a.slope = 0.2f;
    long tot = (long)a.N * a.IH * a.IW;
    int lpp = c->Cout / 4;
    // one-row windows (D.conv1): the single-pass kernel, thread = output pixel x channel quad
    if (c->kh == 1 && c->kw == 4 && c->sh == 1 && c->ph == 0 && !c->transposed && (lpp == 8 || lpp == 16 || lpp == 32) && a.IH == a.OH) {
        if (cin1_rows_ok(a, CIN1_PB, 1, 4)) {
            const int nblk = a.M / CIN1_PB;
            if (lpp == 8) VIAI_LAUNCH((cin1_bn_dgrad_rows_kernel<8, 4>), dim3(nblk), dim3(256), 0, st, a);
            else if (lpp == 16) VIAI_LAUNCH((cin1_bn_dgrad_rows_kernel<16, 4>), dim3(nblk), dim3(256), 0, st, a);
            else VIAI_LAUNCH((cin1_bn_dgrad_rows_kernel<32, 4>), dim3(nblk), dim3(256), 0, st, a);
            return viai_launch_status();
        }
    }
    long nb = (tot * lpp + 255) / 256;
    if (nb > 4096) nb = 4096;
    int blocks = (int)nb;
    if (c->kh == 1 && c->kw == 4 && c->sh == 1 && c->sw == 2 && lpp == 16) {          // D.conv1
        VIAI_LAUNCH((cin1_dgrad_kernel<16, 1, 4, 1, 2, true>), dim3(blocks), dim3(256), 0, st, a);
        return viai_launch_status();
    }
    if (c->kh == 3 && c->kw == 3 && c->sh == 2 && c->sw == 2 && lpp == 8) {           // E.conv1
        VIAI_LAUNCH((cin1_dgrad_kernel<8, 3, 3, 2, 2, true>), dim3(blocks), dim3(256), 0, st, a);
        return viai_launch_status();
    }
#define CALL(KH, KW)                                                                                                     \
    if (lpp == 8) VIAI_LAUNCH((cin1_dgrad_kernel<8, KH, KW, 0, 0, true>), dim3(blocks), dim3(256), 0, st, a);            \
    else if (lpp == 16) VIAI_LAUNCH((cin1_dgrad_kernel<16, KH, KW, 0, 0, true>), dim3(blocks), dim3(256), 0, st, a);     \
    else VIAI_LAUNCH((cin1_dgrad_kernel<32, KH, KW, 0, 0, true>), dim3(blocks), dim3(256), 0, st, a)
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    return viai_launch_status();
}

int viai_cout1_fwd(const viai_conv2d* c, const float* x, const float* wp, const float* bias, float* y, int act, hipStream_t st) {
    if (c->sh != 1 || c->sw != 1 || c->C2 != 0) return (int)hipErrorInvalidValue;
    DirectArgs a = make_args(c);
    a.x = x; a.w = wp; a.bias = bias; a.y = y; a.act = act; a.slope = 0.2f;
    const int cin = a.Cin;
    const int lpp = cin >= 256 ? 64 : cin / 4;
    long nb = ((long)a.M * lpp + 255) / 256;
    if (nb > 16384) nb = 16384;
    const int blocks = (int)nb;
    constexpr int runk = 1;
    if (runk && c->kh == 3 && c->kw == 3 && c->ph == 1 && c->pw == 1 && a.OW % 4 == 0 && a.OW == a.IW && a.OH == a.IH) {
        const int L = cin == 512 ? 2 : 4;                            // (L + 2) x 3 x CPL float4 of window registers
        long nr = (long)a.N * a.OH * (a.OW / L);
        long nb2 = (nr * lpp + 255) / 256;
        if (nb2 > 8192) nb2 = 8192;
        const dim3 g2((unsigned)nb2), b2(256);
#define RUN(LPP_, CPL_, L_)                                                                                                    \
        do { if (a.transposed) VIAI_LAUNCH((cout1_fwd_run_kernel<LPP_, CPL_, L_, false>), g2, b2, 0, st, a);                   \
             else VIAI_LAUNCH((cout1_fwd_run_kernel<LPP_, CPL_, L_, true>), g2, b2, 0, st, a); } while (0)
        if (cin == 32) RUN(8, 1, 4); else if (cin == 64) RUN(16, 1, 4); else if (cin == 128) RUN(32, 1, 4);
        else if (cin == 256) RUN(64, 1, 4); else if (cin == 512) RUN(64, 2, 2); else return (int)hipErrorInvalidValue;
#undef RUN
        return viai_launch_status();
    }
#define CALL(KH, KW)                                                                                              \
    if (cin == 32) VIAI_LAUNCH((cout1_fwd_kernel<8, 1, KH, KW>), dim3(blocks), dim3(256), 0, st, a);              \
    else if (cin == 64) VIAI_LAUNCH((cout1_fwd_kernel<16, 1, KH, KW>), dim3(blocks), dim3(256), 0, st, a);        \
    else if (cin == 128) VIAI_LAUNCH((cout1_fwd_kernel<32, 1, KH, KW>), dim3(blocks), dim3(256), 0, st, a);       \
    else if (cin == 256) VIAI_LAUNCH((cout1_fwd_kernel<64, 1, KH, KW>), dim3(blocks), dim3(256), 0, st, a);       \
    else if (cin == 512) VIAI_LAUNCH((cout1_fwd_kernel<64, 2, KH, KW>), dim3(blocks), dim3(256), 0, st, a);       \
    else return (int)hipErrorInvalidValue
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    return viai_launch_status();
}

int viai_cout1_dgrad(const viai_conv2d* c, const float* dy, const float* wp, float* dx, hipStream_t st) {
    if (c->sh != 1 || c->sw != 1 || c->C2 != 0 || (c->C1 % 4) != 0) return (int)hipErrorInvalidValue;
    DirectArgs a = make_args(c);
    if (a.Cin / 4 > 256 || 256 % (a.Cin / 4) != 0) return (int)hipErrorInvalidValue;
    a.dy = dy; a.w = wp; a.dx = dx;
    long total = (long)a.N * a.IH * a.IW * (a.Cin / 4);
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    const int blocks = (int)nb;
    constexpr int runk = 1;
    constexpr int L = 8;
    if (runk && c->kh == 3 && c->kw == 3 && c->ph == 1 && c->pw == 1 && a.IW % L == 0 && a.OW == a.IW && a.OH == a.IH) {
        long nb2 = ((long)a.N * a.IH * (a.IW / L) * (a.Cin / 4) + 255) / 256;
        if (nb2 > 4096) nb2 = 4096;
        if (a.transposed) VIAI_LAUNCH((cout1_dgrad_run_kernel<L, false>), dim3((unsigned)nb2), dim3(256), 0, st, a);
        else VIAI_LAUNCH((cout1_dgrad_run_kernel<L, true>), dim3((unsigned)nb2), dim3(256), 0, st, a);
        return viai_launch_status();
    }
#define CALL(KH, KW) VIAI_LAUNCH((cout1_dgrad_kernel<KH, KW>), dim3(blocks), dim3(256), 0, st, a)
    VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    return viai_launch_status();
}

size_t viai_cout1_wgrad_ws_floats(const viai_conv2d* c) {
    long q = (long)c->N * c->IH * c->IW;
    return (size_t)direct_wgrad_blocks(q) * c->kh * c->kw * (c->C1 + c->C2);
}

int viai_cout1_wgrad(const viai_conv2d* c, const float* x, const float* dy, float* ws, float* dw, int accumulate, hipStream_t st) {
    if (c->sh != 1 || c->sw != 1 || c->C2 != 0) return (int)hipErrorInvalidValue;
    DirectArgs a = make_args(c);
    if (a.Cin % 4 != 0 || a.Cin / 4 > 256 || 256 % (a.Cin / 4) != 0) return (int)hipErrorInvalidValue;
    a.x = x; a.dy = dy; a.ws = ws;
    const int T = c->kh * c->kw;
    long q = (long)a.N * a.IH * a.IW;
    int nb = direct_wgrad_blocks(q);
    int ppb = (int)((q + nb - 1) / nb);
    int pg = 256 / (a.Cin / 4);
    size_t lds = (size_t)pg * T * a.Cin * sizeof(float);
    constexpr int runk = 1;
    constexpr int L = 16;
    if (runk && c->kh == 3 && c->kw == 3 && a.IW % L == 0 && c->ph == 1 && c->pw == 1) {
        const long nruns = (long)a.N * a.IH * (a.IW / L);
        const int rpl = (int)((nruns + (long)nb * pg - 1) / ((long)nb * pg));      // runs per pixel lane, so that nb slabs cover all runs
        if (a.transposed) VIAI_LAUNCH((cout1_wgrad_run_kernel<3, 3, L, false>), dim3(nb), dim3(256), lds, st, a, rpl);
        else VIAI_LAUNCH((cout1_wgrad_run_kernel<3, 3, L, true>), dim3(nb), dim3(256), lds, st, a, rpl);
    } else {
#define CALL(KH, KW) VIAI_LAUNCH((cout1_wgrad_kernel<KH, KW>), dim3(nb), dim3(256), lds, st, a, ppb)
        VIAI_WINDOW_DISPATCH(c, CALL);
#undef CALL
    }
    int e = viai_launch_status();
    if (e) return e;
    // conv [1][Cin][kh][kw] and convT [Cin][1][kh][kw] both flatten to ci*T + t
    return viai_wgrad_reduce(ws, dw, nb, T, 1, a.Cin, (long)a.Cin * T, T, accumulate, st);
}

// ---- fused (conv + BatchNorm + activation) -> (3 x 3 stride-1 pad-1 conv with ONE output channel) pair ----------------------------
// c2 describes the Cout = 1 layer (C1 = the front layer's channel count).  The front layer's post-activation tensor z is never stored:
// these entry points read its pre-BatchNorm output y with (scale, shift, act_in) applied on load, and the BatchNorm backward forms the
// Cout = 1 layer's data gradient in registers (kernels above).  Reference layer pairs: G.conv6_1 + conv6_1_bn + ReLU -> conv6_2
// (New_Inpainting_Networks.py:85-88), D.conv3 + norm3 + LeakyReLU -> conv4 (Discriminator_Networks.py:44-49).
static int pair_rows(const DirectArgs& a) {       // rows of du staged per block of the BatchNorm-backward kernels
    int R = 4;
    while (R > 1 && (a.IH % R != 0 || (long)a.N * (a.IH / R) < 512)) R >>= 1;
    return a.IH % R == 0 ? R : 1;
}
extern "C" int viai_pair_cout1_ok(const viai_conv2d* c) {
    if (!c || c->Cout != 1 || c->C2 != 0 || c->kh != 3 || c->kw != 3 || c->sh != 1 || c->sw != 1 || c->ph != 1 || c->pw != 1) return 0;
    if (c->dh > 1 || c->dw > 1 || c->ph2 >= 0 || c->pw2 >= 0) return 0;
    if (!(c->C1 == 32 || c->C1 == 64 || c->C1 == 128 || c->C1 == 256 || c->C1 == 512)) return 0;
    if (c->IW % 16 != 0 || c->IW + 2 > 4096) return 0;
    return 1;
}
extern "C" int viai_pair_cout1_bn_bwd_blocks(const viai_conv2d* c) {
    if (!viai_pair_cout1_ok(c)) return 0;
    DirectArgs a = make_args(c);
    return a.N * (a.IH / pair_rows(a));
}
extern "C" int viai_pair_cout1_fwd(const viai_conv2d* c, const float* y, const float* scale, const float* shift, int act_in,
                                   const float* wp, const float* bias, float* out, int act, void* stream) {
    if (!viai_pair_cout1_ok(c)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    DirectArgs a = make_args(c);
    a.x = y; a.w = wp; a.bias = bias; a.y = out; a.act = act; a.slope = 0.2f; a.scale = scale; a.shift = shift; a.act_in = act_in;
    const int cin = a.Cin;
    // rows per block: as many as keep >= 512 blocks in flight and the nine planes within 64 KB of LDS
    int R = 8;
    while (R > 1 && ((long)a.N * ((a.IH + R - 1) / R) < 256 || (size_t)9 * R * a.IW * sizeof(float) > 65536)) R >>= 1;
    const dim3 g2((unsigned)(a.N * ((a.IH + R - 1) / R))), b2(256);
    const size_t lds = (size_t)9 * R * a.IW * sizeof(float);
    viai_tag_reset();
    viai_tag_kernel("direct");
#define RUN(LPP_, CPL_)                                                                                                          \
    do { if (a.transposed) VIAI_LAUNCH((cout1_pair_fwd_kernel<LPP_, CPL_, false>), g2, b2, lds, st, a, R);                       \
         else VIAI_LAUNCH((cout1_pair_fwd_kernel<LPP_, CPL_, true>), g2, b2, lds, st, a, R); } while (0)
    if (cin == 32) RUN(8, 1); else if (cin == 64) RUN(16, 1); else if (cin == 128) RUN(32, 1);
    else if (cin == 256) RUN(64, 1); else RUN(64, 2);
#undef RUN
    return viai_launch_status();
}
// the same result through the dots / gather pair (wide front layers); ws: 9 * N * IH * IW floats
extern "C" int viai_pair_cout1_fwd_dots(const viai_conv2d* c, const float* y, const float* scale, const float* shift, int act_in,
                                        const float* wp, const float* bias, float* ws, float* out, int act, void* stream) {
    if (!viai_pair_cout1_ok(c) || ws == nullptr) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    DirectArgs a = make_args(c);
    a.x = y; a.w = wp; a.bias = bias; a.y = out; a.act = act; a.slope = 0.2f; a.scale = scale; a.shift = shift; a.act_in = act_in;
    const int cin = a.Cin;
    const int npix = a.N * a.IH * a.IW;
    const int lpp = cin >= 256 ? 64 : cin / 4, cpl = cin == 512 ? 2 : 1, u = cpl >= 2 ? 4 : 8;
    long nb = ((long)npix + (256 / lpp) * u - 1) / ((256 / lpp) * u);
    if (nb > 2048) nb = 2048;
    const dim3 g2((unsigned)nb), b2(256);
    viai_tag_reset();
    viai_tag_kernel("direct");
    if (cin == 32) VIAI_LAUNCH((cout1_pair_dots_kernel<8, 1>), g2, b2, 0, st, a, ws, npix);
    else if (cin == 64) VIAI_LAUNCH((cout1_pair_dots_kernel<16, 1>), g2, b2, 0, st, a, ws, npix);
    else if (cin == 128) VIAI_LAUNCH((cout1_pair_dots_kernel<32, 1>), g2, b2, 0, st, a, ws, npix);
    else if (cin == 256) VIAI_LAUNCH((cout1_pair_dots_kernel<64, 1>), g2, b2, 0, st, a, ws, npix);
    else VIAI_LAUNCH((cout1_pair_dots_kernel<64, 2>), g2, b2, 0, st, a, ws, npix);
    VIAI_LAUNCH(cout1_pair_gather_kernel, dim3((a.M + 255) / 256), dim3(256), 0, st, a, (const float*)ws);
    return viai_launch_status();
}
// dw (+)= weight gradient of the Cout = 1 layer from du and z = act_in(scale * y + shift) formed on load; ws: viai_conv2d_wgrad_ws_bytes(c)
extern "C" int viai_pair_cout1_wgrad(const viai_conv2d* c, const float* y, const float* scale, const float* shift, int act_in,
                                     const float* du, float* ws, float* dw, int accumulate, void* stream) {
    if (!viai_pair_cout1_ok(c)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    DirectArgs a = make_args(c);
    a.x = y; a.dy = du; a.ws = ws; a.scale = scale; a.shift = shift; a.act_in = act_in; a.slope = 0.2f;
    constexpr int T = 9, L = 16;
    const long q = (long)a.N * a.IH * a.IW;
    const int nb = direct_wgrad_blocks(q);
    const int pg = 256 / (a.Cin / 4);
    const size_t lds = (size_t)pg * T * a.Cin * sizeof(float);
    const long nruns = (long)a.N * a.IH * (a.IW / L);
    const int rpl = (int)((nruns + (long)nb * pg - 1) / ((long)nb * pg));
    viai_tag_reset();
    viai_tag_kernel("direct");
    if (a.transposed) VIAI_LAUNCH((cout1_wgrad_run_kernel<3, 3, L, false, true>), dim3(nb), dim3(256), lds, st, a, rpl);
    else VIAI_LAUNCH((cout1_wgrad_run_kernel<3, 3, L, true, true>), dim3(nb), dim3(256), lds, st, a, rpl);
    int e = viai_launch_status();
    if (e) return e;
    return viai_wgrad_reduce(ws, dw, nb, T, 1, a.Cin, (long)a.Cin * T, T, accumulate, st);
}
// BatchNorm + activation backward of the front layer from du (N, H, W, 1) -- the gradient of the Cout = 1 layer's PRE-activation output:
// part: 2 * C * viai_pair_cout1_bn_bwd_blocks(c) floats; sums / dgamma / dbeta / training / dy_amax as in viai_bn_act_bwd_amax;
// dy (optional) = the gradient of the front layer's conv output.
static int pair_bn_bwd_impl(const viai_conv2d* c, const float* du, const float* wp, const float* y, const float* mean,
                            const float* invstd, const float* scale, const float* shift, int act_in, float* part, float* sums,
                            float* dgamma, float* dbeta, float* dy, int training, float* dy_amax, void* stream, int p16);
extern "C" int viai_pair_cout1_bn_bwd(const viai_conv2d* c, const float* du, const float* wp, const float* y, const float* mean,
                                      const float* invstd, const float* scale, const float* shift, int act_in, float* part, float* sums,
                                      float* dgamma, float* dbeta, float* dy, int training, float* dy_amax, void* stream) {
    return pair_bn_bwd_impl(c, du, wp, y, mean, invstd, scale, shift, act_in, part, sums, dgamma, dbeta, dy, training, dy_amax, stream, 0);
}
// (ABI 13) the same with dy written pre-split (P16): part 3 * C * blocks floats, sums 3 * C floats, *dy_amax receives the bound; C % 32 == 0
extern "C" int viai_pair_cout1_bn_bwd_p16(const viai_conv2d* c, const float* du, const float* wp, const float* y, const float* mean,
                                          const float* invstd, const float* scale, const float* shift, int act_in, float* part, float* sums,
                                          float* dgamma, float* dbeta, float* dy, int training, float* dy_amax, void* stream) {
    if (c == nullptr || (c->C1 + c->C2) % 32 != 0 || dy == nullptr || dy_amax == nullptr || act_in == VIAI_ACT_SIGMOID) return (int)hipErrorInvalidValue;
    return pair_bn_bwd_impl(c, du, wp, y, mean, invstd, scale, shift, act_in, part, sums, dgamma, dbeta, dy, training, dy_amax, stream, 1);
}
static int pair_bn_bwd_impl(const viai_conv2d* c, const float* du, const float* wp, const float* y, const float* mean,
                            const float* invstd, const float* scale, const float* shift, int act_in, float* part, float* sums,
                            float* dgamma, float* dbeta, float* dy, int training, float* dy_amax, void* stream, int p16) {
    if (!viai_pair_cout1_ok(c)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    DirectArgs a = make_args(c);
    a.dy_p16 = p16;
    a.x = y; a.dy = du; a.w = wp; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.act_in = act_in; a.slope = 0.2f;
    a.part = part; a.k_sums = sums; a.dx = dy; a.amax = dy_amax;
    const int R = pair_rows(a);
    const int nb = a.N * (a.IH / R);
    const size_t lds = (size_t)(R + 2) * (a.IW + 2) * sizeof(float);
    constexpr int L = 8;
    if (a.transposed) VIAI_LAUNCH((cout1_bn_bwd_rows_kernel<L, false, false>), dim3(nb), dim3(256), lds, st, a, R);
    else VIAI_LAUNCH((cout1_bn_bwd_rows_kernel<L, true, false>), dim3(nb), dim3(256), lds, st, a, R);
    int e = viai_launch_status();
    if (e) return e;
    e = viai_bn_bwd_final_launch(part, nb, a.Cin, (long)a.N * a.IH * a.IW, mean, invstd, scale, training & 1, sums, dgamma, dbeta, (training >> 1) & 1, st, p16 ? 3 : 2);
    if (e || dy == nullptr) return e;
    if (a.transposed) VIAI_LAUNCH((cout1_bn_bwd_rows_kernel<L, false, true>), dim3(nb), dim3(256), lds, st, a, R);
    else VIAI_LAUNCH((cout1_bn_bwd_rows_kernel<L, true, true>), dim3(nb), dim3(256), lds, st, a, R);
    return viai_launch_status();
}

extern "C" int viai_colsum_blocks(long M, int C) {
    // ~2048 row blocks for large tensors, never fewer rows than one unrolled pass of the reduce covers
    long rows_min = 4096 / (C > 0 ? C : 1);
    if (rows_min < 4) rows_min = 4;
    long rows = (M + 2047) / 2048;
    if (rows < rows_min) rows = rows_min;
    long b = (M + rows - 1) / rows;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int viai_colsum(const float* x, long M, int C, float* part, float* out, int accumulate, void* stream) {
    int nb = viai_colsum_blocks(M, C);
    long rpb = (M + nb - 1) / nb;
    VIAI_LAUNCH(colsum_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, part, M, C, rpb);
    VIAI_LAUNCH(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, out, nb, C, accumulate);
    return viai_launch_status();
}
