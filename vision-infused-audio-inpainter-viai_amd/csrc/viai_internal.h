// Internal (non-ABI) declarations shared between the kernel translation units.
#pragma once
#include "viai_common.h"
#include "../../include/viai_hip.h"

struct ConvArgs {
    const float* in;      // gathered NHWC tensor, channels [0, C1)
    const float* in2;     // optional second source, channels [C1, C1+C2) (virtual concat)
    const float* wp;      // packed weights [Cout][wtaps][C1+C2]
    const float* bias;    // [Cout] or null
    float* out;           // NHWC, channels [0, OC1)
    float* out2;          // optional second destination, channels [OC1, Cout)
    float* stat;          // optional BN partials [2][Cout][nblk_m]
    int C1, C2, Cout, OC1;
    int M, nblk_m, nblk_n;
    int act;              // fused activation (only when stat == null)
    float slope;
    int wfrag;            // bf16x3 only: 1 = weights packed fragment-major (viai_bf3_frag_layout)
    int sk;               // bf16x3 only: 1 = 32x32-tile kernel whose four waves split K (viai_bf3_sk_ok)
    const float* amax;    // f16x2 launches: device scalar max |gathered tensor| (dynamic operand scale, data gradients); null = static F16_ASCALE
    int in_p16;           // 1: the gathered tensor (in; no in2) is stored pre-split (P16 planes, viai_bf3.h) with the scale of *amax
    ConvGeom g;
};

struct WgradArgs {
    const float* x; const float* x2;   // forward input (virtual concat C1 + C2)
    const float* dy;                   // [M][Cout]
    float* ws;                         // [ksplit][wtaps][Cout][Cin]
    const float* amax;                 // f16x2 launch: device scalar max |dy| (dynamic operand scale); null = bf16x3
    const float* xmax;                 // f16x2 launch: device scalar max |x| (and |x2|) if known (dynamic operand scale for the forward input); null = static F16_ASCALE
    int C1, C2, Cout;
    int M;
    int nblk_co, nblk_ci, ksplit, chunks_per_split;
    int dy_p16, x_p16;                 // 1: dy / x (no x2) stored pre-split (P16 planes, viai_bf3.h) with the scales of *amax / *xmax
    ConvGeom g;                        // forward geometry (ly = lx = 1)
};

int viai_conv_igemm_launch(ConvArgs& a, hipStream_t st);
int viai_igemm_tile_m(long M, int n_out);
int viai_conv_igemm_bf3_launch(ConvArgs& a, hipStream_t st);
size_t viai_bf3_packed_floats(int n_out, int k_in, int taps);
int viai_pack_weight_bf3(const float* w, void* wp, int n_out, int k_in, int taps, long s_no, long s_ki, int frag, hipStream_t st);
int viai_pack_job_bf3(const float* w, void* wp, int n_out, int k_in, int taps, long s_no, long s_ki, int frag, viai_pack_job* job);
bool viai_bf3_frag_layout(long M, int n_out);
bool viai_bf3_sk_ok(long M, int n_out, int C1, int C2);
bool viai_conv_halo_ok(const ConvGeom& g, int C1, int C2, int Cout);
bool viai_conv_halo16_ok(const ConvGeom& g, int C1, int C2, int Cout);
int viai_conv_halo_bf3_launch(ConvArgs& a, hipStream_t st);
// conv_halo_dma.hip: P16 input patches by LDS-DMA (round 5)
int viai_conv_halo_c32_dma_launch(ConvArgs& a, int y0, int x0, const int* slots9, hipStream_t st);
bool viai_halo_dma_on();
bool viai_conv_halo_c32_dma_ok(const ConvArgs& a);
bool viai_conv_s2_dma_ok(const ConvArgs& a);          // stride-2 forward, loader / consumer waves
int viai_conv_s2_dma_launch(ConvArgs& a, hipStream_t st);
bool viai_conv_s1_dma_ok(const ConvArgs& a);          // stride-1 256 k-channel layers on the same kernel (D.conv3)
int viai_conv_s1_dma_launch(ConvArgs& a, hipStream_t st);
bool viai_conv_lin_dma_geom_ok(const ConvArgs& a);   // stride-1 3 x 3 layers on linear pixel tiles (maps that are not whole 8 x 16 tiles: the ResNet branch)
bool viai_conv_lin_dma_ok(const ConvArgs& a);
int viai_conv_lin_dma_launch(ConvArgs& a, hipStream_t st);
int viai_lin_dma_stat_merge(long M, int Cout, int* grid, int* pw, int* nitems);   // > 0: the kernel's BatchNorm partials are merged per block (that many per channel)
// conv_stem.hip: the 7 x 7 stride-2 image conv of the ResNet branch on the f16x2 matrix-core path (forward + weight gradient)
bool viai_conv_stem_ok(const ConvGeom& g, int Cin, int Cout, int kh, int kw, int sh, int sw, int ph, int pw);
int viai_conv_stem_fwd_launch(ConvArgs& a, hipStream_t st);
int viai_conv_stem_wgrad_slabs(const ConvGeom& g);
int viai_conv_stem_wgrad_launch(WgradArgs& a, int Cin, float* dw, int accumulate, hipStream_t st);
int viai_conv_stem_pack(const float* w, float* wp, int Cin, hipStream_t st);
bool viai_conv_halo_wide_ok(const ConvArgs& a);
int viai_halo_tiles_y(const ConvGeom& g);      // 8 x 16 output tiles of the wide halo kernel (the last row / column of tiles may be partial)
int viai_halo_tiles_x(const ConvGeom& g);
int viai_halo_s2_rows(const ConvGeom& g);      // tile rows (8 or 4) of the wide halo kernel's stride-2 forward instance: BatchNorm partial blocks = 16 x rows pixels
int viai_conv_halo_wide_launch(ConvArgs& a, hipStream_t st);
bool viai_dgrad_s2_ok(const viai_conv2d* c);
int viai_conv_dgrad_s2_bf3_launch(ConvArgs& a, hipStream_t st);
int viai_wgrad_mfma_launch(WgradArgs& a, int ksplit, hipStream_t st);
bool viai_wgrad32_ok(const ConvGeom& g, int Cout, int C1, int C2);
int viai_wgrad32_ksplit(long M);
int viai_wgrad32_launch(WgradArgs& a, int ksplit, hipStream_t st);
int viai_wgrad_bf3_launch(WgradArgs& a, int ksplit, hipStream_t st);
bool viai_wgrad_bf3_ok(int Cout, int C1, int C2);
int viai_wgrad_pick_ksplit(int Cout, int Cin, int ntaps, long M);
bool viai_wgrad_patch_ok(const ConvGeom& g, int Cout, int C1, int C2);      // conv_wgrad_patch.hip: all-taps f16x2 kernel (3 x 3, stride 1 / 2)
bool viai_wgrad_patch_shape_ok(const ConvGeom& g, int Cout, int C1, int C2);
int viai_wgrad_patch_ksplit(const ConvGeom& g, int Cout, int C1, int C2);
int viai_wgrad_patch_launch(WgradArgs& a, hipStream_t st);

// Which kernel family ran: every conv launcher tags its launch; the C-ABI entry points reset the tag on entry and
// viai_conv2d_last_kernel() reports it (bench.py prices each family against the ceiling of its arithmetic -- the name ends in
// _f16x2 / _bf16x3 / _f32, or is "direct" for the Cin = 1 / Cout = 1 streaming kernels).  Per thread, like the error slot.
struct ViaiKernelTag { const char* family; int launches; };
extern thread_local ViaiKernelTag viai_kernel_tag;
static inline void viai_tag_kernel(const char* family) { viai_kernel_tag.family = family; viai_kernel_tag.launches += 1; }
static inline void viai_tag_reset() { viai_kernel_tag.family = nullptr; viai_kernel_tag.launches = 0; }

// geometry builders (conv_api.hip)
void viai_geom_fwd(const viai_conv2d* c, ConvGeom* g);
int viai_geom_dgrad_class(const viai_conv2d* c, int a, int b, ConvGeom* g);   // returns ntaps
