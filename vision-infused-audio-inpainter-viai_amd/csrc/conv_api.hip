// C-ABI entry points of the convolution family: geometry (tap tables), weight
// packing, and dispatch between the MFMA implicit-GEMM kernels and the
// streaming kernels for the Cin == 1 / Cout == 1 layers.
#include "viai_common.h"
#include "viai_internal.h"

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
static inline bool valid(const viai_conv2d* c) {
    if (!c || c->N <= 0 || c->IH <= 0 || c->IW <= 0 || c->C1 <= 0 || c->C2 < 0 || c->Cout <= 0) return false;
    if (c->kh <= 0 || c->kw <= 0 || c->kh * c->kw > VIAI_MAX_TAPS) return false;
    if (c->sh <= 0 || c->sw <= 0 || c->ph < 0 || c->pw < 0) return false;
    if (c->transposed && (c->sh != 1 || c->sw != 1)) return false;
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    return oh > 0 && ow > 0;
}
enum { K_IGEMM = 0, K_CIN1 = 1, K_COUT1 = 2 };
static inline int kind_of(const viai_conv2d* c) {
    if (cin_of(c) == 1) return K_CIN1;
    if (c->Cout == 1) return K_COUT1;
    return K_IGEMM;
}

extern "C" int viai_abi_version(void) { return VIAI_ABI_VERSION; }

extern "C" int viai_conv2d_out_hw(const viai_conv2d* c, int* OH, int* OW) {
    if (c->transposed) {   // ConvTranspose2d, stride 1: (I-1) - 2p + k
        *OH = c->IH - 1 - 2 * c->ph + c->kh;
        *OW = c->IW - 1 - 2 * c->pw + c->kw;
    } else {
        *OH = (c->IH + 2 * c->ph - c->kh) / c->sh + 1;
        *OW = (c->IW + 2 * c->pw - c->kw) / c->sw + 1;
    }
    return 0;
}

extern "C" size_t viai_conv2d_packed_floats(const viai_conv2d* c) {
    return (size_t)c->Cout * cin_of(c) * c->kh * c->kw;
}

void viai_geom_fwd(const viai_conv2d* c, ConvGeom* g) {
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    g->N = c->N; g->IH = c->IH; g->IW = c->IW; g->OH = oh; g->OW = ow; g->SH = oh; g->SW = ow;
    g->ly = g->lx = 1; g->ay = g->ax = 0;
    g->my = c->sh; g->mx = c->sw;
    g->ntaps = g->wtaps = c->kh * c->kw;
    for (int r = 0; r < c->kh; ++r)
        for (int s = 0; s < c->kw; ++s) {
            int t = r * c->kw + s;
            g->dy[t] = (c->transposed ? c->ph - r : r - c->ph);
            g->dx[t] = (c->transposed ? c->pw - s : s - c->pw);
            g->ws[t] = t;
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
    g->wtaps = c->kh * c->kw;
    int nt = 0;
    for (int r = 0; r < c->kh; ++r)
        for (int s = 0; s < c->kw; ++s) {
            int dy, dx;
            if (c->transposed) {                   // fwd: o = i - p + r  ->  gathered y = iy + (r - p)
                dy = r - c->ph; dx = s - c->pw;
            } else {                               // fwd: i = o*s - p + r ->  o = (iy + p - r)/s
                int ny = a + c->ph - r, nx = b + c->pw - s;
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
    default:
        if (c->transposed) return viai_pack_weight(w, wp, c->Cout, Cin, T, T, (long)c->Cout * T, stream);
        return viai_pack_weight(w, wp, c->Cout, Cin, T, (long)Cin * T, T, stream);
    }
}

extern "C" int viai_conv2d_pack_dgrad(const viai_conv2d* c, const float* w, float* wp, void* stream) {
    if (!valid(c)) return (int)hipErrorInvalidValue;
    const int T = c->kh * c->kw, Cin = cin_of(c);
    switch (kind_of(c)) {
    case K_CIN1:
    case K_COUT1:
        return viai_conv2d_pack_fwd(c, w, wp, stream);     // the streaming kernels share one image
    default:            // wp[ci][t][co]
        if (c->transposed) return viai_pack_weight(w, wp, Cin, c->Cout, T, (long)c->Cout * T, T, stream);
        return viai_pack_weight(w, wp, Cin, c->Cout, T, T, (long)Cin * T, stream);
    }
}

extern "C" int viai_conv2d_stat_geom(const viai_conv2d* c, int* nblk, int* rows_per_blk) {
    if (!valid(c)) return (int)hipErrorInvalidValue;
    if (kind_of(c) == K_CIN1) return viai_cin1_stat_geom(c, nblk, rows_per_blk);
    int oh, ow;
    viai_conv2d_out_hw(c, &oh, &ow);
    long M = (long)c->N * oh * ow;
    const int bm = (kind_of(c) == K_COUT1) ? 128 : viai_igemm_tile_m(M, c->Cout);
    *rows_per_blk = bm;
    *nblk = (int)((M + bm - 1) / bm);
    return 0;
}

extern "C" int viai_conv2d_fwd(const viai_conv2d* c, const float* x, const float* x2, const float* wp,
                               const float* bias, float* y, float* stat_part, int act, void* stream) {
    if (!valid(c) || (c->C2 > 0) != (x2 != nullptr)) return (int)hipErrorInvalidValue;
    if (stat_part != nullptr && act != VIAI_ACT_NONE) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch (kind_of(c)) {
    case K_CIN1: return viai_cin1_fwd(c, x, wp, bias, y, stat_part, act, st);
    case K_COUT1:
        if (stat_part) return (int)hipErrorInvalidValue;
        return viai_cout1_fwd(c, x, wp, bias, y, act, st);
    default: break;
    }
    ConvArgs a{};
    a.in = x; a.in2 = x2; a.wp = wp; a.bias = bias; a.out = y; a.out2 = nullptr; a.stat = stat_part;
    a.C1 = c->C1; a.C2 = c->C2; a.Cout = c->Cout; a.OC1 = c->Cout;
    a.act = act; a.slope = 0.2f;
    viai_geom_fwd(c, &a.g);
    a.M = a.g.N * a.g.OH * a.g.OW;
    return viai_conv_igemm_launch(a, st);
}

extern "C" int viai_conv2d_dgrad(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2, void* stream) {
    if (!valid(c) || (c->C2 > 0) != (dx2 != nullptr)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    switch (kind_of(c)) {
    case K_CIN1: return viai_cin1_dgrad(c, dy, wp, dx, st);
    case K_COUT1: return viai_cout1_dgrad(c, dy, wp, dx, st);
    default: break;
    }
    for (int a_ = 0; a_ < c->sh; ++a_)
        for (int b_ = 0; b_ < c->sw; ++b_) {
            ConvArgs a{};
            a.in = dy; a.in2 = nullptr; a.wp = wp; a.bias = nullptr; a.out = dx; a.out2 = dx2; a.stat = nullptr;
            a.C1 = c->Cout; a.C2 = 0; a.Cout = cin_of(c); a.OC1 = c->C1;
            a.act = VIAI_ACT_NONE; a.slope = 0.f;
            int nt = viai_geom_dgrad_class(c, a_, b_, &a.g);
            if (a.g.SH <= 0 || a.g.SW <= 0) continue;
            if (nt == 0) return (int)hipErrorInvalidValue;   // a class with no taps would need a zero fill
            a.M = a.g.N * a.g.SH * a.g.SW;
            int e = viai_conv_igemm_launch(a, st);
            if (e) return e;
        }
    return 0;
}

extern "C" size_t viai_conv2d_wgrad_ws_bytes(const viai_conv2d* c) {
    if (!valid(c)) return 0;
    size_t fl;
    switch (kind_of(c)) {
    case K_CIN1: fl = viai_cin1_wgrad_ws_floats(c); break;
    case K_COUT1: fl = viai_cout1_wgrad_ws_floats(c); break;
    default: {
        int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
        long M = (long)c->N * oh * ow;
        int ks = viai_wgrad_pick_ksplit(c->Cout, cin_of(c), c->kh * c->kw, M);
        fl = (size_t)ks * viai_conv2d_packed_floats(c);
    } }
    // + column-sum partials for the bias gradient
    int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
    fl += (size_t)viai_colsum_blocks((long)c->N * oh * ow, c->Cout) * c->Cout;
    return fl * sizeof(float);
}

extern "C" int viai_conv2d_wgrad(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                                 float* ws, float* dw, float* db, int accumulate, void* stream) {
    if (!valid(c) || (c->C2 > 0) != (x2 != nullptr)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    int oh, ow; viai_conv2d_out_hw(c, &oh, &ow);
    const long M = (long)c->N * oh * ow;
    const int T = c->kh * c->kw, Cin = cin_of(c);
    int e = 0;
    size_t used = 0;
    switch (kind_of(c)) {
    case K_CIN1: e = viai_cin1_wgrad(c, x, dy, ws, dw, accumulate, st); used = viai_cin1_wgrad_ws_floats(c); break;
    case K_COUT1: e = viai_cout1_wgrad(c, x, dy, ws, dw, accumulate, st); used = viai_cout1_wgrad_ws_floats(c); break;
    default: {
        WgradArgs a{};
        a.x = x; a.x2 = x2; a.dy = dy; a.ws = ws; a.C1 = c->C1; a.C2 = c->C2; a.Cout = c->Cout; a.M = (int)M;
        viai_geom_fwd(c, &a.g);
        int ks = viai_wgrad_pick_ksplit(c->Cout, Cin, T, M);
        used = (size_t)ks * viai_conv2d_packed_floats(c);
        e = viai_wgrad_mfma_launch(a, ks, st);
        if (e) return e;
        if (c->transposed) e = viai_wgrad_reduce(ws, dw, ks, T, c->Cout, Cin, T, (long)c->Cout * T, accumulate, st);
        else e = viai_wgrad_reduce(ws, dw, ks, T, c->Cout, Cin, (long)Cin * T, T, accumulate, st);
    } }
    if (e) return e;
    if (db != nullptr) e = viai_colsum(dy, M, c->Cout, ws + used, db, accumulate, stream);
    return e;
}
