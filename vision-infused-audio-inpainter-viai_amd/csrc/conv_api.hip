// C-ABI entry points of the convolution family: geometry (tap tables), weight
// packing, and dispatch between the MFMA implicit-GEMM kernels and the
// streaming kernels for the Cin == 1 / Cout == 1 layers.
#include "viai_common.h"
#include "viai_internal.h"
#include <cstdlib>
#include <cstring>

// direct kernels (conv_direct.hip)
int viai_cin1_fwd(const viai_conv2d* c, const float* x, const float* w, const float* bias, float* y, float* stat, int act, hipStream_t st);
int viai_cin1_stat_geom(const viai_conv2d* c, int* nblk, int* rows);
int viai_cin1_dgrad(const viai_conv2d* c, const float* dy, const float* w, float* dx, hipStream_t st);
size_t viai_cin1_wgrad_ws_floats(const viai_conv2d* c);
int viai_cin1_wgrad(const viai_conv2d* c, const float* x, const float* dy, float* ws, float* dw, int accumulate, hipStream_t st);
int viai_cout1_fwd(const viai_conv2d* c, const float* x, const float* wp, const float* bias, float* y, int act, hipStream_t st);
int viai_cout1_dgrad(const viai_conv2d* c, const float* dy, const float* wp, float* dx, hipStream_t st);
size_t viai_cout1_wgrad_ws_floats(const viai_conv2d* c);
int viai_cout1_wgrad(const viai_conv2d* c, const float* x, const float* dy, float* ws, float* dw, int accumulate, hipStream_t st);
int viai_wgrad_reduce(const float* ws, float* dw, int nz, int T, int Cout, int Cin, long s_co, long s_ci, int accumulate, hipStream_t st);
extern "C" int viai_colsum_blocks(long M, int C);
extern "C" int viai_colsum(const float* x, long M, int C, float* part, float* out, int accumulate, void* stream);

static inline int cin_of(const viai_conv2d* c) { return c->C1 + c->C2; }
static inline int dil_h(const viai_conv2d* c) { return c->dh > 1 ? c->dh : 1; }
static inline int dil_w(const viai_conv2d* c) { return c->dw > 1 ? c->dw : 1; }
static inline int pad_b(const viai_conv2d* c) { return c->ph2 >= 0 ? c->ph2 : c->ph; }
static inline int pad_r(const viai_conv2d* c) { return c->pw2 >= 0 ? c->pw2 : c->pw; }
enum { K_IGEMM = 0, K_CIN1 = 1, K_COUT1 = 2, K_RUN = 3 };
static inline int kind_of(const viai_conv2d* c) {
    if (cin_of(c) == 1) return K_CIN1;
    if (c->Cout == 1) return K_COUT1;
    if (cin_of(c) <= 4) return K_RUN;       // ResNet conv1 (Cin 3 / 2): input stored with channel stride 4
    return K_IGEMM;
}

static inline bool valid(const viai_conv2d* c) {
    if (!c || c->N <= 0 || c->IH <= 0 || c->IW <= 0 || c->C1 <= 0 || c->C2 < 0 || c->Cout <= 0) return false;
    const bool runk = (cin_of(c) > 1 && cin_of(c) <= 4 && c->Cout > 1);     // row-run kind: taps are kernel rows
    if (c->kh <= 0 || c->kw <= 0) return false;
    if (runk ? (c->kh > VIAI_MAX_TAPS || c->kw > 8) : (c->kh * c->kw > VIAI_MAX_TAPS)) return false;
    if (c->sh <= 0 || c->sw <= 0 || c->ph < 0 || c->pw < 0) return false;
    if (c->transposed && (c->sh != 1 || c->sw != 1)) return false;
    if ((dil_h(c) > 1 || dil_w(c) > 1 || c->ph2 >= 0 || c->pw2 >= 0) && (c->transposed || kind_of(c) != K_IGEMM)) return false;
    if (cin_of(c) > 1 && cin_of(c) <= 4 && c->Cout > 1 && (c->kw > 8 || c->transposed || c->C2 != 0)) return false;
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    return oh > 0 && ow > 0;
}
// ---- math mode of the contraction-shaped layers ------------------------------------------------------
// default: "bf16x3" split-bf16 MFMA (fp32-grade accuracy, 2.67x the fp32-MFMA ceiling) wherever the
// 128x128 tile applies; VIAI_MATH=fp32 forces the exact-fp32 MFMA kernels everywhere.
static bool bf3_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VIAI_MATH"); v = (e && (!strcmp(e, "fp32") || !strcmp(e, "f32"))) ? 0 : 1; }
    return v == 1;
}
static bool use_bf3_shape(long M, int n_out, int k_in) {
    (void)M; (void)n_out;
    return bf3_enabled() && (k_in % 16) == 0;
}
// GEMM rows that decide the bf16x3 weight layout (pack and launch must agree; dgrad uses its largest parity class)
static long bf3_rows_fwd(const viai_conv2d* c) { int oh, ow; viai_conv2d_out_hw(c, &oh, &ow); return (long)c->N * oh * ow; }
static long bf3_rows_dgrad(const viai_conv2d* c) { return (long)c->N * ((c->IH + c->sh - 1) / c->sh) * ((c->IW + c->sw - 1) / c->sw); }

static bool use_bf3_fwd(const viai_conv2d* c);
static bool use_bf3_dgrad(const viai_conv2d* c);
static bool s2_dgrad(const viai_conv2d* c);
static bool f16x2_enabled();
static bool frag_dgrad(const viai_conv2d* c);
// LDS-resident-tile kernel (conv_halo_bf3.hip) for the small-channel stride-1 layers
static bool halo_fwd(const viai_conv2d* c) {
    if (!use_bf3_fwd(c)) return false;
    ConvGeom g{}; viai_geom_fwd(c, &g);
    return viai_conv_halo_ok(g, c->C1, c->C2, c->Cout);
}
static bool halo_dgrad(const viai_conv2d* c) {
    if (!use_bf3_dgrad(c) || c->sw != 1 || (c->sh != 1 && c->sh != 2)) return false;
    if (c->sh == 2) {                                      // round 6: the stride-(2, 1) layer (MelEncoder.conv2): both row-parity classes on the halo kernel
        if (!f16x2_enabled() || c->transposed) return false;
        for (int a_ = 0; a_ < 2; ++a_) {
            ConvGeom g{}; if (viai_geom_dgrad_class(c, a_, 0, &g) == 0) return false;
            if (!viai_conv_halo_ok(g, c->Cout, 0, cin_of(c))) return false;
        }
        return true;
    }
    ConvGeom g{}; if (viai_geom_dgrad_class(c, 0, 0, &g) == 0) return false;
    return viai_conv_halo_ok(g, c->Cout, 0, cin_of(c));
}
// f16x2 halo kernel with the filter in registers (32 -> <= 32 channels, full 3 x 3 window)
static bool halo16_fwd(const viai_conv2d* c) {
    if (!f16x2_enabled() || !halo_fwd(c)) return false;
    ConvGeom g{}; viai_geom_fwd(c, &g);
    return viai_conv_halo16_ok(g, c->C1, c->C2, c->Cout);
}
static bool halo16_dgrad(const viai_conv2d* c) {
    if (!f16x2_enabled() || !halo_dgrad(c)) return false;
    ConvGeom g{}; viai_geom_dgrad_class(c, 0, 0, &g);
    return viai_conv_halo16_ok(g, c->Cout, 0, cin_of(c));
}
// weight layout of the bf16x3 kernels: fragment-major for the wide-tile and halo kernels, planar otherwise
static bool f16x2_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VIAI_F16X2"); on = e ? atoi(e) : 1; }
    return on != 0;
}
// wide halo kernel (conv_halo_bf3.hip, f16x2 fragment-major weights): stride-1 3 x 3 layers with Cin >= 32 and 32 / 64 / 128k outputs
static bool halo_wide_fwd(const viai_conv2d* c) {
    if (!f16x2_enabled() || !use_bf3_fwd(c)) return false;
    ConvArgs a{};
    viai_geom_fwd(c, &a.g);
    a.C1 = c->C1; a.C2 = c->C2; a.Cout = c->Cout; a.OC1 = c->Cout; a.M = a.g.N * a.g.OH * a.g.OW;
    return viai_conv_halo_wide_ok(a);
}
static bool halo_wide_dgrad(const viai_conv2d* c) {
    if (!f16x2_enabled() || !use_bf3_dgrad(c) || c->sh != 1 || c->sw != 1) return false;
    ConvArgs a{};
    if (viai_geom_dgrad_class(c, 0, 0, &a.g) == 0) return false;
    a.C1 = c->Cout; a.C2 = 0; a.Cout = cin_of(c); a.OC1 = c->C1; a.M = a.g.N * a.g.SH * a.g.SW;
    return viai_conv_halo_wide_ok(a);
}
// loader / consumer kernel over linear pixel tiles (conv_halo_dma.hip, round 5): stride-1 3 x 3 layers on maps that are not whole 8 x 16 tiles, when
// the operand arrives pre-split (geometry only here: the launch decides by a.in_p16)
static bool lin_fwd(const viai_conv2d* c) {
    if (!f16x2_enabled() || !use_bf3_fwd(c)) return false;
    ConvArgs a{};
    viai_geom_fwd(c, &a.g);
    a.C1 = c->C1; a.C2 = c->C2; a.Cout = c->Cout; a.OC1 = c->Cout; a.M = a.g.N * a.g.OH * a.g.OW;
    return viai_conv_lin_dma_geom_ok(a);
}
static bool lin_dgrad(const viai_conv2d* c) {
    if (!f16x2_enabled() || !use_bf3_dgrad(c) || c->sh != 1 || c->sw != 1) return false;
    ConvArgs a{};
    if (viai_geom_dgrad_class(c, 0, 0, &a.g) == 0) return false;
    a.C1 = c->Cout; a.C2 = 0; a.Cout = cin_of(c); a.OC1 = c->C1; a.M = a.g.N * a.g.SH * a.g.SW;
    return viai_conv_lin_dma_geom_ok(a);
}
// f16x2 for the LDS-weight / split-K kernels too (planar fp16 planes); VIAI_F16_PLANAR=0 keeps them on bf16x3
static bool planar16_enabled() {
    constexpr int on = 1;
    return on != 0 && f16x2_enabled();
}
// forward weight layout: 0 planar bf16x3, 1 fragment-major bf16x3, 3 fragment-major f16x2 (wide-tile / halo kernels), 4 planar f16x2
static int frag_fwd(const viai_conv2d* c) {
    if (halo_fwd(c)) return f16x2_enabled() ? 3 : 1;            // f16x2: filter in registers (32 -> <= 32 channels) or streamed
    if (viai_bf3_frag_layout(bf3_rows_fwd(c), c->Cout)) return f16x2_enabled() ? 3 : 1;
    if (halo_wide_fwd(c) || lin_fwd(c)) return 3;
    return planar16_enabled() ? 4 : 0;
}
// data gradient on the f16x2 wide-tile kernel (needs the abs-max of dy): the layers whose classes run on the fragment-major kernel
static bool dgrad_f16(const viai_conv2d* c) {
    if (!f16x2_enabled() || !use_bf3_dgrad(c)) return false;
    if (halo_dgrad(c)) return true;
    if (halo_wide_dgrad(c) || lin_dgrad(c)) return true;
    if (!frag_dgrad(c)) return planar16_enabled();              // LDS-weight / split-K kernels: planar fp16 planes
    return s2_dgrad(c) || viai_bf3_frag_layout(bf3_rows_dgrad(c), cin_of(c));
}
static bool sk_fwd(const viai_conv2d* c) {
    const int lay = frag_fwd(c);                            // planar layouts only (bf16x3 or f16x2)
    return use_bf3_fwd(c) && (lay == 0 || lay == 4) && viai_bf3_sk_ok(bf3_rows_fwd(c), c->Cout, c->C1, c->C2);
}
static bool s2_dgrad(const viai_conv2d* c) { return use_bf3_dgrad(c) && viai_dgrad_s2_ok(c); }     // fused parity classes (conv_dgrad_s2_bf3.hip)
static bool frag_dgrad(const viai_conv2d* c) { return halo_dgrad(c) || s2_dgrad(c) || viai_bf3_frag_layout(bf3_rows_dgrad(c), cin_of(c)); }
// layout of the f16x2 data-gradient image: fragment-major also for the wide halo kernel's layers
static bool frag_dgrad16(const viai_conv2d* c) { return frag_dgrad(c) || halo_wide_dgrad(c) || lin_dgrad(c); }

static bool use_bf3_fwd(const viai_conv2d* c) {
    if (kind_of(c) != K_IGEMM) return false;
    int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
    return use_bf3_shape((long)c->N * oh * ow, c->Cout, cin_of(c));
}
static bool use_bf3_dgrad(const viai_conv2d* c) {
    if (kind_of(c) != K_IGEMM) return false;
    for (int a_ = 0; a_ < c->sh; ++a_)
        for (int b_ = 0; b_ < c->sw; ++b_) {
            ConvGeom g; int nt = viai_geom_dgrad_class(c, a_, b_, &g);
            if (g.SH <= 0 || g.SW <= 0 || nt == 0) continue;
            if (!use_bf3_shape((long)g.N * g.SH * g.SW, cin_of(c), c->Cout)) return false;
        }
    return true;
}

extern "C" int viai_abi_version(void) { return VIAI_ABI_VERSION; }

thread_local ViaiKernelTag viai_kernel_tag = {nullptr, 0};
// name of the kernel family the LAST convolution entry point of this thread (viai_conv2d_fwd / _dgrad[_f16] / _wgrad[_f16] /
// viai_conv2d_cin1_bn_*) launched, copied into buf (NUL-terminated, truncated to cap); returns the number of conv-kernel launches of
// that call (a strided data gradient on the gather kernel is one launch per parity class), 0 if none.
extern "C" int viai_conv2d_last_kernel(char* buf, int cap) {
    if (buf != nullptr && cap > 0) {
        const char* s = viai_kernel_tag.family ? viai_kernel_tag.family : "";
        int i = 0;
        for (; i < cap - 1 && s[i]; ++i) buf[i] = s[i];
        buf[i] = 0;
    }
    return viai_kernel_tag.launches;
}

extern "C" int viai_conv2d_out_hw(const viai_conv2d* c, int* OH, int* OW) {
    if (c->transposed) {   // ConvTranspose2d, stride 1: (I-1) - 2p + k
        *OH = c->IH - 1 - 2 * c->ph + c->kh;
        *OW = c->IW - 1 - 2 * c->pw + c->kw;
    } else {
        *OH = (c->IH + c->ph + pad_b(c) - dil_h(c) * (c->kh - 1) - 1) / c->sh + 1;
        *OW = (c->IW + c->pw + pad_r(c) - dil_w(c) * (c->kw - 1) - 1) / c->sw + 1;
    }
    return 0;
}

extern "C" size_t viai_conv2d_packed_floats(const viai_conv2d* c) {
    if (kind_of(c) == K_RUN) return (size_t)c->Cout * c->kh * 32;
    size_t n = (size_t)c->Cout * cin_of(c) * c->kh * c->kw;
    if (kind_of(c) == K_IGEMM && bf3_enabled()) {                                  // room for fragment-major bf16 planes (fwd or dgrad form)
        size_t f = viai_bf3_packed_floats(c->Cout, cin_of(c), c->kh * c->kw), d = viai_bf3_packed_floats(cin_of(c), c->Cout, c->kh * c->kw);
        size_t m = f > d ? f : d;
        return m > n ? m : n;
    }
    return n;
}

static void geom_base(const viai_conv2d* c, ConvGeom* g) {
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    g->run = 0;
    g->N = c->N; g->IH = c->IH; g->IW = c->IW; g->OH = oh; g->OW = ow; g->SH = oh; g->SW = ow;
    g->ly = g->lx = 1; g->ay = g->ax = 0;
    g->my = c->sh; g->mx = c->sw;
}

void viai_geom_fwd(const viai_conv2d* c, ConvGeom* g) {
    geom_base(c, g);
    g->ntaps = g->wtaps = c->kh * c->kw;
    for (int r = 0; r < c->kh; ++r)
        for (int s = 0; s < c->kw; ++s) {
            int t = r * c->kw + s;
            g->dy[t] = (c->transposed ? c->ph - r : r * dil_h(c) - c->ph);
            g->dx[t] = (c->transposed ? c->pw - s : s * dil_w(c) - c->pw);
            g->ws[t] = t;
        }
}

// row-run geometry (1 < Cin <= 4): one "tap" per kernel row, K = 8 pixels x 4 channels per tap
static void geom_run(const viai_conv2d* c, ConvGeom* g) {
    geom_base(c, g);
    g->run = 1;
    g->ntaps = g->wtaps = c->kh;
    for (int r = 0; r < c->kh; ++r) { g->dy[r] = r - c->ph; g->dx[r] = -c->pw; g->ws[r] = r; }
}
// the ResNet stem (7 x 7, stride 2, 64 channels) on the f16x2 kernels of conv_stem.hip
static bool stem_f16(const viai_conv2d* c) {
    if (kind_of(c) != K_RUN || !f16x2_enabled() || !bf3_enabled() || c->transposed) return false;
    ConvGeom g{}; geom_run(c, &g);
    return viai_conv_stem_ok(g, cin_of(c), c->Cout, c->kh, c->kw, c->sh, c->sw, c->ph, c->pw);
}

// wp[co][r][s*4+ch] = w[co][ch][r][s], zero for s >= kw or ch >= Cin
__global__ void pack_run_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int kh, int kw) {
    const int total = Cout * kh * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int k = i % 32, r = (i / 32) % kh, co = i / (32 * kh);
        int s_ = k / 4, ch = k % 4;
        wp[i] = (s_ < kw && ch < Cin) ? w[((size_t)(co * Cin + ch) * kh + r) * kw + s_] : 0.f;
    }
}

// dw[co][ch][r][s] (+)= sum_z ws[z][r][co][s*4+ch]
__global__ void wgrad_reduce_run_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nz, int Cout, int Cin,
                                        int kh, int kw, int accumulate) {
    const int total = Cout * Cin * kh * kw;
    const size_t slab = (size_t)kh * Cout * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int s_ = i % kw, r = (i / kw) % kh, ch = (i / (kw * kh)) % Cin, co = i / (kw * kh * Cin);
        size_t src = ((size_t)r * Cout + co) * 32 + s_ * 4 + ch;
        float acc = 0.f;
        for (int z = 0; z < nz; ++z) acc += ws[z * slab + src];
        dw[i] = accumulate ? dw[i] + acc : acc;
    }
}

// data gradient: produced tensor = dx (N, IH, IW, Cin), gathered tensor = dy (N, OH, OW, Cout).
// class (a, b) = (iy mod sh, ix mod sw).
int viai_geom_dgrad_class(const viai_conv2d* c, int a, int b, ConvGeom* g) {
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    g->N = c->N; g->IH = oh; g->IW = ow;           // gathered = dy
    g->OH = c->IH; g->OW = c->IW;                  // produced = dx
    g->ly = c->sh; g->lx = c->sw; g->ay = a; g->ax = b;
    g->SH = (c->IH - a + c->sh - 1) / c->sh;
    g->SW = (c->IW - b + c->sw - 1) / c->sw;
    g->my = g->mx = 1;
    g->run = 0;
    g->wtaps = c->kh * c->kw;
    int nt = 0;
    for (int r = 0; r < c->kh; ++r)
        for (int s = 0; s < c->kw; ++s) {
            int dy, dx;
            if (c->transposed) {                   // fwd: o = i - p + r  ->  gathered y = iy + (r - p)
                dy = r - c->ph; dx = s - c->pw;
            } else {                               // fwd: i = o*s - p + r ->  o = (iy + p - r)/s
                int ny = a + c->ph - r * dil_h(c), nx = b + c->pw - s * dil_w(c);
                if (((ny % c->sh) + c->sh) % c->sh != 0 || ((nx % c->sw) + c->sw) % c->sw != 0) continue;
                dy = ny / c->sh; dx = nx / c->sw;   // exact (divisible), may be negative
            }
            g->dy[nt] = dy; g->dx[nt] = dx; g->ws[nt] = r * c->kw + s;
            ++nt;
        }
    g->ntaps = nt;
    return nt;
}

extern "C" int viai_conv2d_pack_fwd(const viai_conv2d* c, const float* w, float* wp, void* stream) {
    if (!valid(c)) return (int)hipErrorInvalidValue;
    const int T = c->kh * c->kw, Cin = cin_of(c);
    switch (kind_of(c)) {
    case K_CIN1:    // [Cout][T]: torch conv layout as is
        return viai_pack_weight(w, wp, c->Cout, 1, T, c->transposed ? T : (long)T, c->transposed ? (long)c->Cout * T : T, stream);
    case K_COUT1:   // wp[t][ci] == pack with n_out = 1 ... expressed as [1][T][Cin]
        return viai_pack_weight(w, wp, 1, Cin, T, 0, T, stream);
    case K_RUN: {
        if (stem_f16(c)) return viai_conv_stem_pack(w, wp, Cin, (hipStream_t)stream);
        int total = c->Cout * c->kh * 32;
        VIAI_LAUNCH(pack_run_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wp, c->Cout, Cin, c->kh, c->kw);
        return viai_launch_status();
    }
    default:
        if (use_bf3_fwd(c)) {
            const int M = frag_fwd(c);
            if (c->transposed) return viai_pack_weight_bf3(w, wp, c->Cout, Cin, T, T, (long)c->Cout * T, M, (hipStream_t)stream);
            return viai_pack_weight_bf3(w, wp, c->Cout, Cin, T, (long)Cin * T, T, M, (hipStream_t)stream);
        }
        if (c->transposed) return viai_pack_weight(w, wp, c->Cout, Cin, T, T, (long)c->Cout * T, stream);
        return viai_pack_weight(w, wp, c->Cout, Cin, T, (long)Cin * T, T, stream);
    }
}

extern "C" int viai_conv2d_pack_dgrad(const viai_conv2d* c, const float* w, float* wp, void* stream) {
    if (!valid(c)) return (int)hipErrorInvalidValue;
    const int T = c->kh * c->kw, Cin = cin_of(c);
    switch (kind_of(c)) {
    case K_RUN: return (int)hipErrorInvalidValue;          // image inputs need no data gradient
    case K_CIN1:
    case K_COUT1:
        return viai_conv2d_pack_fwd(c, w, wp, stream);     // the streaming kernels share one image
    default:            // wp[ci][t][co]
        if (use_bf3_dgrad(c)) {
            const int M = frag_dgrad(c);
            if (c->transposed) return viai_pack_weight_bf3(w, wp, Cin, c->Cout, T, (long)c->Cout * T, T, M, (hipStream_t)stream);
            return viai_pack_weight_bf3(w, wp, Cin, c->Cout, T, T, (long)Cin * T, M, (hipStream_t)stream);
        }
        if (c->transposed) return viai_pack_weight(w, wp, Cin, c->Cout, T, (long)c->Cout * T, T, stream);
        return viai_pack_weight(w, wp, Cin, c->Cout, T, T, (long)Cin * T, stream);
    }
}

// Job descriptor of this layer's weight image for viai_pack_jobs_run; returns 1 for the one image kind that is not
// batched (row-run mode of the image-input 7x7 conv: pack it with viai_conv2d_pack_fwd).
extern "C" int viai_conv2d_pack_job(const viai_conv2d* c, int dgrad, const float* w, float* wp, viai_pack_job* job) {
    if (!valid(c) || job == nullptr) return (int)hipErrorInvalidValue;
    const int T = c->kh * c->kw, Cin = cin_of(c);
    switch (kind_of(c)) {                    // same cases as viai_conv2d_pack_fwd / _dgrad; frag = 2: fp32 [no][t][ki]
    case K_RUN: return 1;
    case K_CIN1: return viai_pack_job_bf3(w, wp, c->Cout, 1, T, c->transposed ? T : (long)T, c->transposed ? (long)c->Cout * T : T, 2, job);
    case K_COUT1: return viai_pack_job_bf3(w, wp, 1, Cin, T, 0, T, 2, job);
    default: break;
    }
    if (!dgrad) {
        const int frag = use_bf3_fwd(c) ? frag_fwd(c) : 2;
        if (c->transposed) return viai_pack_job_bf3(w, wp, c->Cout, Cin, T, T, (long)c->Cout * T, frag, job);
        return viai_pack_job_bf3(w, wp, c->Cout, Cin, T, (long)Cin * T, T, frag, job);
    }
    if (dgrad == 2 && !dgrad_f16(c)) return (int)hipErrorInvalidValue;
    const int frag = dgrad == 2 ? (frag_dgrad16(c) ? 3 : 4) : use_bf3_dgrad(c) ? (frag_dgrad(c) ? 1 : 0) : 2;
    if (c->transposed) return viai_pack_job_bf3(w, wp, Cin, c->Cout, T, (long)c->Cout * T, T, frag, job);
    return viai_pack_job_bf3(w, wp, Cin, c->Cout, T, T, (long)Cin * T, frag, job);
}

extern "C" int viai_conv2d_stat_geom(const viai_conv2d* c, int* nblk, int* rows_per_blk) {
    if (!valid(c)) return (int)hipErrorInvalidValue;
    if (kind_of(c) == K_CIN1) return viai_cin1_stat_geom(c, nblk, rows_per_blk);
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    long M = (long)c->N * oh * ow;
    int bm = (kind_of(c) == K_COUT1 || stem_f16(c) || halo_fwd(c) || halo_wide_fwd(c)) ? 128 : sk_fwd(c) ? 32 : viai_igemm_tile_m(M, c->Cout);
    if (kind_of(c) == K_IGEMM && !halo_fwd(c) && halo_wide_fwd(c) && c->sh == 2) {        // stride-2 wide halo forward: 64-pixel tiles where it runs them
        ConvGeom g{}; viai_geom_fwd(c, &g);
        bm = 16 * viai_halo_s2_rows(g);
    }
    *rows_per_blk = bm;
    *nblk = (int)((M + bm - 1) / bm);
    int th, tw;
    if (viai_conv2d_stat_tiles(c, &th, &tw) == 0 && th > 0) *nblk = c->N * ((oh + th - 1) / th) * ((ow + tw - 1) / tw);
    return 0;
}

extern "C" int viai_conv2d_stat_tiles(const viai_conv2d* c, int* tile_h, int* tile_w) {
    if (!valid(c)) return (int)hipErrorInvalidValue;
    *tile_h = *tile_w = 0;
    if (kind_of(c) == K_CIN1 || kind_of(c) == K_COUT1 || !halo_wide_fwd(c)) return 0;
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    if (oh % 8 != 0 || ow % 16 != 0) { *tile_h = 8; *tile_w = 16; }       // the wide halo kernel's 8 x 16 tiles, clipped at the map's edge
    return 0;
}

extern "C" int viai_conv2d_fwd(const viai_conv2d* c, const float* x, const float* x2, const float* wp,
                               const float* bias, float* y, float* stat_part, int act, void* stream) {
    return viai_conv2d_fwd_amax(c, x, x2, wp, bias, y, stat_part, act, nullptr, stream);
}

// 1 if the forward launch of this layer splits its activations into fp16 terms (then x_amax matters)
extern "C" int viai_conv2d_fwd_f16_ok(const viai_conv2d* c) {
    if (valid(c) && stem_f16(c)) return 1;
    if (!valid(c) || kind_of(c) != K_IGEMM || !use_bf3_fwd(c)) return 0;
    const int lay = frag_fwd(c);
    return (lay == 3 || lay == 4) ? 1 : 0;
}

// x_amax: device float >= max |x| (and |x2|), or NULL.  The f16x2 kernels scale their activation operand by a power of two before the
// split: from x_amax when it is given (any magnitude is then representable), by the static 16 otherwise (|x| beyond 4094 saturates)
static int fwd_impl(const viai_conv2d* c, const float* x, const float* x2, const float* wp,
                    const float* bias, float* y, float* stat_part, int act, const float* x_amax, void* stream, int p16);
extern "C" int viai_conv2d_fwd_amax(const viai_conv2d* c, const float* x, const float* x2, const float* wp,
                                    const float* bias, float* y, float* stat_part, int act, const float* x_amax, void* stream) {
    return fwd_impl(c, x, x2, wp, bias, y, stat_part, act, x_amax, stream, 0);
}
static bool p16_fwd_ok(const viai_conv2d* c) {
    return valid(c) && kind_of(c) == K_IGEMM && f16x2_enabled() && c->C2 == 0 && (halo_fwd(c) || halo_wide_fwd(c) || lin_fwd(c));
}
// (ABI 13) the forward with x pre-split (P16 planes, scale from *x_amax): layers with VIAI_P16_OK_FWD_X in viai_conv2d_p16_ok
extern "C" int viai_conv2d_fwd_p16(const viai_conv2d* c, const float* x, const float* wp, const float* bias, float* y, float* stat_part,
                                   int act, const float* x_amax, void* stream) {
    if (!p16_fwd_ok(c) || x_amax == nullptr) return (int)hipErrorInvalidValue;
    return fwd_impl(c, x, nullptr, wp, bias, y, stat_part, act, x_amax, stream, 1);
}
static int fwd_impl(const viai_conv2d* c, const float* x, const float* x2, const float* wp,
                    const float* bias, float* y, float* stat_part, int act, const float* x_amax, void* stream, int p16) {
    if (!valid(c) || (c->C2 > 0) != (x2 != nullptr)) return (int)hipErrorInvalidValue;
    if (stat_part != nullptr && act != VIAI_ACT_NONE) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    viai_tag_reset();
    switch (kind_of(c)) {
    case K_CIN1: viai_tag_kernel("direct"); return viai_cin1_fwd(c, x, wp, bias, y, stat_part, act, st);
    case K_COUT1:
        if (stat_part) return (int)hipErrorInvalidValue;
        viai_tag_kernel("direct");
        return viai_cout1_fwd(c, x, wp, bias, y, act, st);
    default: break;
    }
    ConvArgs a{};
    a.in = x; a.in2 = x2; a.wp = wp; a.bias = bias; a.out = y; a.out2 = nullptr; a.stat = stat_part;
    a.C1 = c->C1; a.C2 = c->C2; a.Cout = c->Cout; a.OC1 = c->Cout;
    a.act = act; a.slope = 0.2f;
    a.in_p16 = p16;
    if (kind_of(c) == K_RUN) { geom_run(c, &a.g); a.C1 = 32; a.C2 = 0; }
    else viai_geom_fwd(c, &a.g);
    a.M = a.g.N * a.g.OH * a.g.OW;
    if (stem_f16(c)) { a.amax = x_amax; return viai_conv_stem_fwd_launch(a, st); }
    if (use_bf3_fwd(c)) {
        a.wfrag = frag_fwd(c);
        if (a.wfrag == 3 || a.wfrag == 4) a.amax = x_amax;            // f16x2 weight image = f16x2 kernel
        if (halo_fwd(c)) return viai_conv_halo_bf3_launch(a, st);
        a.sk = sk_fwd(c);
        return viai_conv_igemm_bf3_launch(a, st);
    }
    return viai_conv_igemm_launch(a, st);
}

static int dgrad_impl(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2, const float* amax, void* stream, int p16 = 0);

extern "C" int viai_conv2d_dgrad(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2, void* stream) {
    return dgrad_impl(c, dy, wp, dx, dx2, nullptr, stream);
}

// f16x2 data gradient: 1 if this layer has one (then pack with viai_conv2d_pack_dgrad_f16 and pass the abs-max of dy)
extern "C" int viai_conv2d_dgrad_f16_ok(const viai_conv2d* c) { return (valid(c) && kind_of(c) == K_IGEMM && dgrad_f16(c)) ? 1 : 0; }

extern "C" int viai_conv2d_pack_dgrad_f16(const viai_conv2d* c, const float* w, float* wp, void* stream) {
    if (!viai_conv2d_dgrad_f16_ok(c)) return (int)hipErrorInvalidValue;
    const int T = c->kh * c->kw, Cin = cin_of(c);
    const int lay = frag_dgrad16(c) ? 3 : 4;
    if (c->transposed) return viai_pack_weight_bf3(w, wp, Cin, c->Cout, T, (long)c->Cout * T, T, lay, (hipStream_t)stream);
    return viai_pack_weight_bf3(w, wp, Cin, c->Cout, T, T, (long)Cin * T, lay, (hipStream_t)stream);
}

// dy_amax: device float holding max |dy| (viai_bn_act_bwd_amax): the f16x2 operand scale is derived from it on the device
extern "C" int viai_conv2d_dgrad_f16(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2,
                                     const float* dy_amax, void* stream) {
    if (!viai_conv2d_dgrad_f16_ok(c) || dy_amax == nullptr) return (int)hipErrorInvalidValue;
    return dgrad_impl(c, dy, wp, dx, dx2, dy_amax, stream);
}

// the patch-staged stride-2 data gradient (conv_dgrad_s2_bf3.hip) takes this layer: base lattice a multiple of 8 x 16
static bool s2_patch_dgrad(const viai_conv2d* c) {
    if (!s2_dgrad(c)) return false;
    int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
    return oh % 8 == 0 && ow % 16 == 0 && c->IH == 2 * oh && c->IW == 2 * ow && c->Cout % 32 == 0;
}
static bool p16_dgrad_ok(const viai_conv2d* c) {
    if (!valid(c) || kind_of(c) != K_IGEMM || !dgrad_f16(c)) return false;
    if (s2_dgrad(c)) return s2_patch_dgrad(c) || c->Cout % 32 == 0;          // (round 5: the gather kernel of the fused classes stages pieces too)
    return c->sh == 1 && c->sw == 1 && (halo_dgrad(c) || halo_wide_dgrad(c) || lin_dgrad(c));
}
// (ABI 13) viai_conv2d_dgrad_f16 with dy pre-split (P16 planes, scale from *dy_amax): layers with VIAI_P16_OK_DGRAD_DY
extern "C" int viai_conv2d_dgrad_f16_p16(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2,
                                         const float* dy_amax, void* stream) {
    if (!p16_dgrad_ok(c) || dy_amax == nullptr) return (int)hipErrorInvalidValue;
    return dgrad_impl(c, dy, wp, dx, dx2, dy_amax, stream, 1);
}

static int dgrad_impl(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2, const float* amax, void* stream, int p16) {
    if (!valid(c) || (c->C2 > 0) != (dx2 != nullptr)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    viai_tag_reset();
    switch (kind_of(c)) {
    case K_CIN1: viai_tag_kernel("direct"); return viai_cin1_dgrad(c, dy, wp, dx, st);
    case K_COUT1: viai_tag_kernel("direct"); return viai_cout1_dgrad(c, dy, wp, dx, st);
    case K_RUN: return (int)hipErrorInvalidValue;
    default: break;
    }
    {   // a parity class with no valid tap (e.g. 1x1 stride 2) receives no gradient: zero-fill first
        bool empty = false;
        for (int a_ = 0; a_ < c->sh && !empty; ++a_)
            for (int b_ = 0; b_ < c->sw; ++b_) { ConvGeom g; if (viai_geom_dgrad_class(c, a_, b_, &g) == 0 && g.SH > 0 && g.SW > 0) { empty = true; break; } }
        if (empty) {
            size_t px = (size_t)c->N * c->IH * c->IW;
            if (hipMemsetAsync(dx, 0, px * c->C1 * sizeof(float), st) != hipSuccess) return (int)hipErrorInvalidValue;
            if (dx2 && hipMemsetAsync(dx2, 0, px * c->C2 * sizeof(float), st) != hipSuccess) return (int)hipErrorInvalidValue;
        }
    }
    const bool bf3 = use_bf3_dgrad(c);
    if (s2_dgrad(c)) {                                     // all four parity classes in one launch
        ConvArgs a{};
        a.in = dy; a.wp = wp; a.out = dx; a.out2 = dx2;
        a.C1 = c->Cout; a.C2 = 0; a.Cout = cin_of(c); a.OC1 = c->C1;
        int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
        a.g.N = c->N; a.g.IH = oh; a.g.IW = ow; a.g.OH = c->IH; a.g.OW = c->IW;
        a.M = c->N * (c->IH / 2) * (c->IW / 2);
        a.wfrag = amax != nullptr ? 3 : 1;
        a.amax = amax;
        a.in_p16 = p16;
        return viai_conv_dgrad_s2_bf3_launch(a, st);
    }
    for (int a_ = 0; a_ < c->sh; ++a_)
        for (int b_ = 0; b_ < c->sw; ++b_) {
            ConvArgs a{};
            a.in = dy; a.in2 = nullptr; a.wp = wp; a.bias = nullptr; a.out = dx; a.out2 = dx2; a.stat = nullptr;
            a.C1 = c->Cout; a.C2 = 0; a.Cout = cin_of(c); a.OC1 = c->C1;
            a.act = VIAI_ACT_NONE; a.slope = 0.f;
            a.in_p16 = p16;
            int nt = viai_geom_dgrad_class(c, a_, b_, &a.g);
            if (a.g.SH <= 0 || a.g.SW <= 0) continue;
            if (nt == 0) continue;                            // zero-filled above
            a.M = a.g.N * a.g.SH * a.g.SW;
            a.wfrag = bf3 && frag_dgrad(c);
            if (amax != nullptr) { a.wfrag = frag_dgrad16(c) ? 3 : 4; a.amax = amax; }       // f16x2 weights + dynamic operand scale
            a.sk = bf3 && (a.wfrag == 0 || a.wfrag == 4) && viai_bf3_sk_ok(a.M, a.Cout, a.C1, 0);
            int e = (bf3 && halo_dgrad(c)) ? viai_conv_halo_bf3_launch(a, st) : bf3 ? viai_conv_igemm_bf3_launch(a, st) : viai_conv_igemm_launch(a, st);
            if (e) return e;
        }
    return 0;
}

// weight-gradient split-K of the igemm-class layers: the all-taps 32-channel kernel has its own rule (one slab per block)
static bool wgrad32(const viai_conv2d* c) {
    if (kind_of(c) != K_IGEMM) return false;
    ConvGeom g{}; viai_geom_fwd(c, &g);
    return viai_wgrad32_ok(g, c->Cout, c->C1, c->C2);
}
static int wgrad_ksplit(const viai_conv2d* c, long M) {
    if (wgrad32(c)) return viai_wgrad32_ksplit(M);
    return viai_wgrad_pick_ksplit(c->Cout, cin_of(c), c->kh * c->kw, M);
}
// the all-taps patch kernel (f16x2 launches of the stride-1 3 x 3 layers with >= 128 x 64 channels)
static bool wgrad_patch(const viai_conv2d* c, bool shape_only = false) {
    if (kind_of(c) != K_IGEMM || !f16x2_enabled() || !bf3_enabled()) return false;
    ConvGeom g{}; viai_geom_fwd(c, &g);
    return shape_only ? viai_wgrad_patch_shape_ok(g, c->Cout, c->C1, c->C2) : viai_wgrad_patch_ok(g, c->Cout, c->C1, c->C2);
}

extern "C" size_t viai_conv2d_wgrad_ws_bytes(const viai_conv2d* c) {
    if (!valid(c)) return 0;
    size_t fl;
    switch (kind_of(c)) {
    case K_CIN1: fl = viai_cin1_wgrad_ws_floats(c); break;
    case K_COUT1: fl = viai_cout1_wgrad_ws_floats(c); break;
    case K_RUN: {
        int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
        int ks = viai_wgrad_pick_ksplit(c->Cout, 32, c->kh, (long)c->N * oh * ow);
        if (stem_f16(c)) { ConvGeom g{}; geom_run(c, &g); const int kz = viai_conv_stem_wgrad_slabs(g); if (kz > ks) ks = kz; }
        fl = (size_t)ks * viai_conv2d_packed_floats(c); break;
    }
    default: {
        int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
        long M = (long)c->N * oh * ow;
        int ks = wgrad_ksplit(c, M);
        if (wgrad_patch(c, true)) {                            // workspace covers every form the layer can take, whatever the switches say
            ConvGeom g{}; viai_geom_fwd(c, &g);
            int kp = viai_wgrad_patch_ksplit(g, c->Cout, c->C1, c->C2);
            if (kp > ks) ks = kp;
        }
        fl = (size_t)ks * viai_conv2d_packed_floats(c);
    } }
    // + column-sum partials for the bias gradient
    int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
    fl += (size_t)viai_colsum_blocks((long)c->N * oh * ow, c->Cout) * c->Cout;
    return fl * sizeof(float);
}

static int wgrad_impl(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                      float* ws, float* dw, float* db, int accumulate, const float* amax, const float* xmax, void* stream, int p16 = 0);

extern "C" int viai_conv2d_wgrad(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                                 float* ws, float* dw, float* db, int accumulate, void* stream) {
    return wgrad_impl(c, x, x2, dy, ws, dw, db, accumulate, nullptr, nullptr, stream);
}

// f16x2 weight gradient (layers on the bf16x3 wgrad kernel): dy scaled on the device from dy_amax = max |dy|, x by the
// static activation scale; 1 from viai_conv2d_wgrad_f16_ok if the layer has this form
extern "C" int viai_conv2d_wgrad_f16_ok(const viai_conv2d* c) {
    // the layers of the f16x2 wgrad_bf3 kernel (> 32 channels on both sides) and every layer an instance of the patch kernel takes
    if (valid(c) && stem_f16(c)) return 1;
    return (valid(c) && kind_of(c) == K_IGEMM && f16x2_enabled() && bf3_enabled() && (viai_wgrad_bf3_ok(c->Cout, c->C1, c->C2) || wgrad_patch(c))) ? 1 : 0;
}
extern "C" int viai_conv2d_wgrad_f16(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                                     float* ws, float* dw, float* db, int accumulate, const float* dy_amax, const float* x_amax, void* stream) {
    if (!viai_conv2d_wgrad_f16_ok(c) || dy_amax == nullptr) return (int)hipErrorInvalidValue;
    return wgrad_impl(c, x, x2, dy, ws, dw, db, accumulate, dy_amax, x_amax, stream);
}

// (ABI 13) which operands of this layer's f16x2 kernels may arrive pre-split (P16 planes, csrc/viai_bf3.h): a mask of VIAI_P16_OK_*
extern "C" int viai_conv2d_p16_ok(const viai_conv2d* c) {
    if (!valid(c) || kind_of(c) != K_IGEMM || !f16x2_enabled() || !bf3_enabled()) return 0;
    int m = 0;
    if (wgrad_patch(c)) { m |= VIAI_P16_OK_WGRAD_DY; if (c->C2 == 0) m |= VIAI_P16_OK_WGRAD_X; }
    if (p16_fwd_ok(c)) m |= VIAI_P16_OK_FWD_X;
    if (p16_fwd_ok(c) && lin_fwd(c)) m |= VIAI_P16_OK_FWD_LIN;
    if (p16_dgrad_ok(c)) m |= VIAI_P16_OK_DGRAD_DY;
    return m;
}
extern "C" int viai_conv2d_wgrad_f16_p16(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                                         float* ws, float* dw, float* db, int accumulate, const float* dy_amax, const float* x_amax, int flags, void* stream) {
    if (!viai_conv2d_wgrad_f16_ok(c) || dy_amax == nullptr) return (int)hipErrorInvalidValue;
    return wgrad_impl(c, x, x2, dy, ws, dw, db, accumulate, dy_amax, x_amax, stream, flags);
}

static int wgrad_impl(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                      float* ws, float* dw, float* db, int accumulate, const float* amax, const float* xmax, void* stream, int p16) {
    if (!valid(c) || (c->C2 > 0) != (x2 != nullptr)) return (int)hipErrorInvalidValue;
    if (p16 != 0 && (kind_of(c) != K_IGEMM || amax == nullptr || !wgrad_patch(c) || ((p16 & VIAI_P16_DY) && db != nullptr)
                     || ((p16 & VIAI_P16_X) && (x2 != nullptr || xmax == nullptr)))) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
    const long M = (long)c->N * oh * ow;
    const int T = c->kh * c->kw, Cin = cin_of(c);
    int e = 0;
    size_t used = 0;
    viai_tag_reset();
    if (kind_of(c) == K_CIN1 || kind_of(c) == K_COUT1) viai_tag_kernel("direct");
    switch (kind_of(c)) {
    case K_CIN1: e = viai_cin1_wgrad(c, x, dy, ws, dw, accumulate, st); used = viai_cin1_wgrad_ws_floats(c); break;
    case K_COUT1: e = viai_cout1_wgrad(c, x, dy, ws, dw, accumulate, st); used = viai_cout1_wgrad_ws_floats(c); break;
    case K_RUN: {
        WgradArgs a{};
        a.x = x; a.x2 = nullptr; a.dy = dy; a.ws = ws; a.C1 = 32; a.C2 = 0; a.Cout = c->Cout; a.M = (int)M;
        geom_run(c, &a.g);
        if (amax != nullptr && stem_f16(c)) {                 // f16x2 launch (viai_conv2d_wgrad_f16): conv_stem.hip, slabs + reduce in one call
            a.amax = amax; a.xmax = xmax;
            used = (size_t)viai_conv_stem_wgrad_slabs(a.g) * viai_conv2d_packed_floats(c);
            e = viai_conv_stem_wgrad_launch(a, Cin, dw, accumulate, st);
            break;
        }
        int ks = viai_wgrad_pick_ksplit(c->Cout, 32, c->kh, M);
        used = (size_t)ks * viai_conv2d_packed_floats(c);
        e = viai_wgrad_mfma_launch(a, ks, st);
        if (e) return e;
        int total = c->Cout * Cin * T;
        VIAI_LAUNCH(wgrad_reduce_run_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws, dw, ks, c->Cout, Cin, c->kh, c->kw, accumulate);
        e = viai_launch_status();
        break;
    }
    default: {
        WgradArgs a{};
        a.x = x; a.x2 = x2; a.dy = dy; a.ws = ws; a.C1 = c->C1; a.C2 = c->C2; a.Cout = c->Cout; a.M = (int)M;
        a.amax = amax;
        a.xmax = xmax;
        a.dy_p16 = (p16 & VIAI_P16_DY) ? 1 : 0; a.x_p16 = (p16 & VIAI_P16_X) ? 1 : 0;
        viai_geom_fwd(c, &a.g);
        int ks = wgrad_ksplit(c, M);
        const bool patch = amax != nullptr && wgrad_patch(c);
        if (patch) ks = viai_wgrad_patch_ksplit(a.g, c->Cout, c->C1, c->C2);
        used = (size_t)ks * viai_conv2d_packed_floats(c);
        e = patch ? viai_wgrad_patch_launch(a, st)
          : wgrad32(c) ? viai_wgrad32_launch(a, ks, st)
          : (bf3_enabled() && viai_wgrad_bf3_ok(c->Cout, c->C1, c->C2)) ? viai_wgrad_bf3_launch(a, ks, st) : viai_wgrad_mfma_launch(a, ks, st);
        if (e) return e;
        if (c->transposed) e = viai_wgrad_reduce(ws, dw, ks, T, c->Cout, Cin, T, (long)c->Cout * T, accumulate, st);
        else e = viai_wgrad_reduce(ws, dw, ks, T, c->Cout, Cin, (long)Cin * T, T, accumulate, st);
    } }
    if (e) return e;
    if (db != nullptr) e = viai_colsum(dy, M, c->Cout, ws + used, db, accumulate, stream);
    return e;
}
