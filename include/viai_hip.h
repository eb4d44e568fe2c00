/* viai_hip.h — C ABI of libviai_hip.so, the MI355X (gfx950) kernel library for the
 * VIAI spectrogram-inpainting GAN hot path.
 *
 * The reference (Hangz-nju-cuhk/Vision-Infused-Audio-Inpainter-VIAI) has no FFI /
 * operator registry: its hot path is torch.nn calls inside nn.Modules (SURVEY.md
 * §8b).  Each entry point below therefore names the reference call site(s) whose
 * ATen dispatch it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller.  The library allocates no device memory and keeps no state BETWEEN
 *     calls that affects results.  What it does keep, per process: the VIAI_* environment switches (read once), the
 *     max-dynamic-LDS attribute of each kernel (set once), a per-thread first-error slot (cleared by every entry point), and the
 *     launch-plan log / plans (viai_plan_*: host memory and hipEvents owned by the plan, freed by viai_plan_destroy);
 *   - activations are fp32 NHWC: [N][H][W][C], C contiguous.  A (N,1,H,W) NCHW
 *     tensor is bit-identical to its NHWC form;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it,
 *     the library never synchronises;
 *   - return value: 0 (hipSuccess) or a hipError_t / hipErrorInvalidValue code.
 *     Nothing throws across this boundary.
 *   - entry points may be called from several host threads on different streams (the launch-plan log is the exception: one
 *     recording at a time); one process per GPU under data parallelism.
 */
#ifndef VIAI_HIP_H
#define VIAI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIAI_ABI_VERSION 17

enum { VIAI_ACT_NONE = 0, VIAI_ACT_RELU = 1, VIAI_ACT_LRELU = 2, VIAI_ACT_SIGMOID = 3 };

int viai_abi_version(void);

/* ---------------------------------------------------------------- convolution
 * nn.Conv2d / stride-1 nn.ConvTranspose2d as used by
 *   MelEncoder            networks/Inpainting_Networks.py:55-63
 *   TransConvBlock        networks/New_Inpainting_Networks.py:24
 *   MelDecoder            networks/New_Inpainting_Networks.py:53-63
 *   MelDiscriminator      networks/Discriminator_Networks.py:17-33
 * `x2` (may be NULL) is a second input concatenated after x on the channel axis
 * (replaces torch.cat at New_Inpainting_Networks.py:81 without the copy).       */
typedef struct viai_conv2d {
    int N, IH, IW;       /* input batch / height / width                           */
    int C1, C2;          /* channels of x and of x2 (C2 = 0 when x2 == NULL)       */
    int Cout;
    int kh, kw, sh, sw, ph, pw;
    int transposed;      /* 0 = nn.Conv2d (weight [Cout][Cin][kh][kw]);
                            1 = nn.ConvTranspose2d, stride 1 (weight [Cin][Cout][kh][kw]) */
    int dh, dw;          /* dilation (0 or 1 = none); nn.Conv2d only.  WaveNet's dilated causal Conv1d
                            (wavenet_vocoder/modules.py:124-126) is kh = 1, kw = 3, dw = 2^i            */
    int ph2, pw2;        /* bottom / right padding; -1 = same as ph / pw (torch's symmetric padding).
                            pw = (k-1)*d, pw2 = 0 computes exactly the T causal outputs the reference keeps
                            after `x[:, :, :residual.size(-1)]` (modules.py:181)                          */
} viai_conv2d;

/* output extent (torch formulas) */
int viai_conv2d_out_hw(const viai_conv2d* c, int* OH, int* OW);
/* floats needed for one packed copy of the weights (forward or dgrad form).  The packed image is opaque:
 * fp32 [Cout][taps][Cin], or three bf16 planes (row-major or MFMA-fragment-major) for the bf16x3 kernels. */
size_t viai_conv2d_packed_floats(const viai_conv2d* c);
/* repack torch-layout weights for the forward / data-gradient kernels */
int viai_conv2d_pack_fwd(const viai_conv2d* c, const float* w, float* wp, void* stream);
int viai_conv2d_pack_dgrad(const viai_conv2d* c, const float* w, float* wp, void* stream);
/* Batched weight packing: every bf16x3 weight image of a model in one launch (the weights change once per optimizer
 * step; ~60 five-microsecond pack launches per step otherwise sit on the critical path).  viai_conv2d_pack_job fills
 * one job (frag: 0 = row-major bf16x3 planes, 1 = fragment-major bf16x3 planes, 2 = fp32 [n_out][tap][k_in]; returns 1
 * for the row-run image of the 7x7 image-input conv: pack that one with viai_conv2d_pack_fwd); the
 * caller sets blk0 to the running sum of nblk, uploads the array and launches it with viai_pack_jobs_run.      */
typedef struct viai_pack_job {
    const void* w; void* wp;
    int n_out, k_in, taps, frag;
    long s_no, s_ki;
    int blk0, nblk;
} viai_pack_job;
int viai_conv2d_pack_job(const viai_conv2d* c, int dgrad, const float* w, float* wp, viai_pack_job* job);
int viai_pack_jobs_run(const viai_pack_job* jobs_dev, int njobs, int total_blocks, void* stream);
/* BatchNorm partial-statistics geometry of the forward kernel: number of row
 * blocks and rows per block; stat_part holds 2*Cout*nblk floats.                */
int viai_conv2d_stat_geom(const viai_conv2d* c, int* nblk, int* rows_per_blk);
/* (ABI 9) Where the forward kernel works on 2-D output tiles and the map is not a whole number of them, the partial blocks are
 * tiles clipped at the map's edge (block = (n, tile row, tile column), row-major), not runs of rows_per_blk rows: tile_h x tile_w
 * is returned here (0, 0 otherwise) and the partials go to viai_bn_finalize_tiles instead of viai_bn_finalize.                    */
int viai_conv2d_stat_tiles(const viai_conv2d* c, int* tile_h, int* tile_w);
/* y = conv(x ++ x2, w) + bias [; act].  stat_part (optional) receives per-block
 * per-channel (mean, M2) of y for training-mode BatchNorm; act must be NONE then. */
int viai_conv2d_fwd(const viai_conv2d* c, const float* x, const float* x2, const float* wp_fwd,
                    const float* bias, float* y, float* stat_part, int act, void* stream);
/* The same with the magnitude of the input known.  The f16x2 kernels (two-term fp16 operand split) pre-scale the activation
 * operand by a power of two: with x_amax -- a DEVICE float >= max |x| (and |x2|), produced by viai_bn_act_fwd_amax /
 * viai_conv2d_cin1_bn_fwd / viai_absmax -- the scale is derived on the device and every magnitude is representable; with
 * x_amax = NULL (and in viai_conv2d_fwd) it is the static 16, right for normalised activations, and |x| beyond 65504 / 16 = 4094
 * saturates.  Kernels that do not split (fp32, bf16x3, the streaming kernels) ignore it.                                        */
int viai_conv2d_fwd_f16_ok(const viai_conv2d* c);      /* 1: the forward launch of this layer is an f16x2 kernel */
int viai_conv2d_fwd_amax(const viai_conv2d* c, const float* x, const float* x2, const float* wp_fwd,
                         const float* bias, float* y, float* stat_part, int act, const float* x_amax, void* stream);
/* amax = max(amax, max |x|) over n floats (x 16-byte aligned): the operand magnitude of a tensor this library did not produce;
 * *amax must hold 0 or an earlier maximum.  One streaming pass.                                                              */
int viai_absmax(const float* x, long n, float* amax, void* stream);
/* dx (++ dx2) = conv_backward_data(dy, w) */
int viai_conv2d_dgrad(const viai_conv2d* c, const float* dy, const float* wp_dgrad,
                      float* dx, float* dx2, void* stream);
/* f16x2 data gradient for the layers whose data gradient runs on the wide-tile kernel (viai_conv2d_dgrad_f16_ok): half the
 * MFMA work of the bf16x3 form.  fp16 has a narrow exponent range and gradients span many decades, so the kernel scales dy
 * by a power of two derived ON THE DEVICE from dy_amax = max |dy| (one float, written by viai_bn_act_bwd_amax).
 * Weights: viai_conv2d_pack_dgrad_f16 (or viai_conv2d_pack_job with dgrad = 2).                                        */
int viai_conv2d_dgrad_f16_ok(const viai_conv2d* c);
int viai_conv2d_pack_dgrad_f16(const viai_conv2d* c, const float* w, float* wp, void* stream);
int viai_conv2d_dgrad_f16(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2,
                          const float* dy_amax, void* stream);
/* dw (torch layout) = conv_backward_weight(x ++ x2, dy); dw += if accumulate.
 * db (optional, [Cout]) = sum of dy.  ws: scratch of viai_conv2d_wgrad_ws_bytes. */
size_t viai_conv2d_wgrad_ws_bytes(const viai_conv2d* c);
int viai_conv2d_wgrad(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                      float* ws, float* dw, float* db, int accumulate, void* stream);
/* f16x2 form of the weight gradient (viai_conv2d_wgrad_f16_ok: layers with more than 32 channels on both sides): dy is
 * scaled on the device from dy_amax = max |dy| (viai_bn_act_bwd_amax), x from x_amax >= max |x|, |x2| (see
 * viai_conv2d_fwd_amax) or, with x_amax = NULL, by the static 16                                                        */
int viai_conv2d_wgrad_f16_ok(const viai_conv2d* c);
int viai_conv2d_wgrad_f16(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                          float* ws, float* dw, float* db, int accumulate, const float* dy_amax, const float* x_amax, void* stream);

/* Which kernel ran.  The convolution entry points above (and viai_conv2d_cin1_bn_fwd / _wgrad) choose between kernel families by
 * shape; this reports the family the LAST such call of the calling thread launched and how many conv-kernel launches it made
 * (a strided data gradient on the gather kernel is one launch per parity class).  The name ends in the arithmetic of the
 * kernel -- "_f16x2" (two-term fp16 split, 3 MFMA products per MAC), "_bf16x3" (three-term bf16 split, 6 products), "_f32"
 * (exact fp32 MFMA) -- or is "direct" for the Cin = 1 / Cout = 1 streaming kernels; bench.py prices each family against the
 * ceiling of that arithmetic.  buf receives the NUL-terminated name (truncated to cap); returns the launch count, 0 if none.
 * (measurement aid: no reference counterpart)                                                                             */
int viai_conv2d_last_kernel(char* buf, int cap);

/* generic weight repack used by the two pack entry points (exposed for tests):
 * wp[no][t][ki] = w[no*s_no + ki*s_ki + t]                                       */
int viai_pack_weight(const float* w, float* wp, int n_out, int k_in, int taps,
                     long s_no, long s_ki, void* stream);

/* ------------------------------------------------------------ batch-norm + act
 * nn.BatchNorm2d in training mode followed by LeakyReLU(0.2) / ReLU
 * (Inpainting_Networks.py:72-76, New_Inpainting_Networks.py:33-36,72-75,86,
 *  Discriminator_Networks.py:39-46).                                            */
/* merge block partials -> mean / invstd / (scale, shift); update running stats
 * (momentum, unbiased variance) and num_batches_tracked (int64, may be NULL).    */
int viai_bn_finalize(const float* stat_part, int nblk, int rows_per_blk, long M, int C,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float momentum, float eps,
                     float* mean, float* invstd, float* scale, float* shift, void* stream);
/* (ABI 9) the same merge where block b = tile (n, ty, tx) of tile_h x tile_w output pixels clipped at the edge of the OH x OW map
 * (viai_conv2d_stat_tiles): N * ceil(OH / tile_h) * ceil(OW / tile_w) blocks of min(tile_h, OH - ty tile_h) * min(tile_w, OW - tx tile_w) rows */
/* (ABI 17) the finalize behind a pre-split forward on the linear-tile kernel (VIAI_P16_OK_FWD_LIN in viai_conv2d_p16_ok): its partials are per 128 consecutive pixels
 * or, for layers with one channel block (Cout 64 / 128 / 256), merged per persistent block inside the conv kernel -- the library applies the same rule here as at
 * the launch.  Replaces the nn.BatchNorm2d statistics of networks/ResNet.py:26-55 behind those convs; stat_part as sized by viai_conv2d_stat_geom. */
int viai_bn_finalize_lin(const float* stat_part, long M, int C,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         int64_t* nbt, float momentum, float eps,
                         float* mean, float* invstd, float* scale, float* shift, void* stream);
int viai_bn_finalize_tiles(const float* stat_part, int N, int OH, int OW, int tile_h, int tile_w, int C,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float momentum, float eps,
                           float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: (scale, shift) from the running statistics */
int viai_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* mean, float* invstd,
                        float* scale, float* shift, void* stream);
/* z = act(y*scale[c] + shift[c]) */
int viai_bn_act_fwd(const float* y, const float* scale, const float* shift, float* z,
                    long M, int C, int act, float slope, void* stream);
/* the same, also reducing max |z| into the device float z_amax (zero-initialised by the caller; NULL = off): the operand
 * magnitude the f16x2 kernels that consume z are handed (viai_conv2d_fwd_amax, viai_conv2d_wgrad_f16)                   */
int viai_bn_act_fwd_amax(const float* y, const float* scale, const float* shift, float* z,
                         long M, int C, int act, float slope, float* z_amax, void* stream);
/* backward of the pair above (training-mode statistics):
 *   dgamma, dbeta (optional outputs) and dy.  part: scratch of 2*C*nblk floats,
 *   nblk from viai_bn_bwd_blocks(M, C).  `training` bit 0: 1 = batch statistics, 0 = eval-mode rule;
 *   bit 1: accumulate into dgamma/dbeta instead of overwriting them.               */
int viai_bn_bwd_blocks(long M, int C);
int viai_bn_act_bwd(const float* dz, const float* y, const float* mean, const float* invstd,
                    const float* scale, const float* shift, float* part, float* sums,
                    float* dgamma, float* dbeta, float* dy, long M, int C, int act, float slope,
                    int training, void* stream);
/* the same, and *amax (one float, zero-initialised by the caller) receives max |dy| */
int viai_bn_act_bwd_amax(const float* dz, const float* y, const float* mean, const float* invstd,
                         const float* scale, const float* shift, float* part, float* sums,
                         float* dgamma, float* dbeta, float* dy, long M, int C, int act, float slope,
                         int training, float* amax, void* stream);
/* elementwise activation backward for layers without BN (sigmoid heads):
 * dx = dz * act'(.) expressed through the activation OUTPUT z                    */
int viai_act_bwd_from_output(const float* dz, const float* z, float* dx, long n, int act,
                             float slope, void* stream);
/* (ABI 15) the same on a gradient that arrives as two addends: dx = (dz + dz2) * act'(.); n a multiple of 4.  The residual join of
 * networks/ResNet.py:46-53: d(out) = conv1's data gradient + the next join's residual gradient, summed where the ReLU mask is applied */
int viai_add_act_bwd_from_output(const float* dz, const float* dz2, const float* z, float* dx, long n, int act,
                                 float slope, void* stream);

/* ------------------------------------------------------------------ resampling
 * F.interpolate(mode='bilinear', align_corners=True)  New_Inpainting_Networks.py:78,83 */
int viai_bilinear_ac_fwd(const float* x, float* y, int N, int IH, int IW, int OH, int OW, int C, void* stream);
int viai_bilinear_ac_bwd(const float* dy, float* dx, int N, int IH, int IW, int OH, int OW, int C, void* stream);
/* BatchNorm apply + activation + the resize in one pass: out = interpolate(act(scale * y + shift), size=(OH, OW), mode="bilinear",
 * align_corners=True) -- the last layer of each decoder block and the F.interpolate behind it (New_Inpainting_Networks.py:76-83); the
 * post-activation map is not stored.  act: none / ReLU / LeakyReLU; C / 4 a power of two <= 256; N * OH * OW < 2^24.  *z_amax (optional,
 * zero-initialised): max |act(..)| over the taps read, the whole map's when the resize does not shrink it (ABI 12)                      */
int viai_bn_act_bilinear_fwd_amax(const float* y, const float* scale, const float* shift, float* out, int N, int IH, int IW,
                                  int OH, int OW, int C, int act, float slope, float* z_amax, void* stream);
/* nn.AvgPool2d((3,1)) on the bottleneck  Inpainting_Networks.py:65,77 (floor mode) */
int viai_avgpool_h_fwd(const float* x, float* y, int N, int IH, int W, int C, int k, void* stream);
int viai_avgpool_h_bwd(const float* dy, float* dx, int N, int IH, int W, int C, int k, void* stream);

/* ResNet-18 visual branch (networks/Image_Embedding.py:13-71, networks/ResNet.py:26-55).
 * For 1 < C1 <= 4 (conv1 on RGB / flow frames) the conv entry points expect x stored with channel stride 4
 * (zero padded): viai_nchw_to_nhwc4 produces that layout from the loader's NCHW frames.                */
int viai_nchw_to_nhwc4(const float* x, float* y, long N, int C, long HW, void* stream);
/* (ABI 15) the same, and max |x| into *amax (one zero-initialised float): the operand scale of the stem conv's f16x2 kernels */
int viai_nchw_to_nhwc4_amax(const float* x, float* y, long N, int C, long HW, float* amax, void* stream);
/* nn.MaxPool2d(k, s, p); idx: one byte per output element (window argmax) kept for the backward */
int viai_maxpool_fwd(const float* x, float* y, unsigned char* idx, int N, int IH, int IW, int C, int k, int s, int p, void* stream);
int viai_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int N, int IH, int IW, int C, int k, int s, int p, void* stream);
/* nn.AvgPool2d(7) on a 7x7 map: mean over the P = H*W positions */
int viai_avgpool_hw_fwd(const float* x, float* y, int N, int P, int C, void* stream);
int viai_avgpool_hw_bwd(const float* dy, float* dx, int N, int P, int C, void* stream);
/* F.avg_pool2d(x, k, s, p, count_include_pad=False): input pyramid of the multi-scale discriminator (cfg 4) */
int viai_avgpool2d_fwd(const float* x, float* y, int N, int IH, int IW, int C, int k, int s, int p, void* stream);
int viai_avgpool2d_bwd(const float* dy, float* dx, int N, int IH, int IW, int C, int k, int s, int p, void* stream);
/* BasicBlock join: out = relu(a + b); backward d = g * (out > 0) (same for both addends) */
int viai_add_relu_fwd(const float* a, const float* b, float* out, long n, void* stream);
/* (ABI 10) the BatchNorm-apply pass fused with what follows it in the ResNet branch (networks/ResNet.py:43-55, Image_Embedding.py:20-23):
 *   viai_bn_add_act_fwd_amax : z = act(scale*y + shift + res)            bn2 -> += identity -> ReLU of a BasicBlock in one pass
 *   viai_bn_act_maxpool_fwd  : out = maxpool_{k,s,p}(act(scale*y + shift)) + argmax bytes; the stem's post-activation map is never stored
 *   viai_bn_act_pool_bwd_amax: viai_bn_act_bwd_amax with the gradient of that map gathered from (dpool, idx) where it is loaded
 * act = VIAI_ACT_RELU or VIAI_ACT_NONE.                                                                                            */
int viai_bn_add_act_fwd_amax(const float* y, const float* scale, const float* shift, const float* res, float* z,
                             long M, int C, int act, float slope, float* z_amax, void* stream);
int viai_bn_act_maxpool_fwd(const float* y, const float* scale, const float* shift, float* out, unsigned char* idx,
                            int N, int IH, int IW, int C, int k, int s, int p, int act, float slope, float* out_amax, void* stream);
int viai_bn_act_pool_bwd_amax(const float* dpool, const unsigned char* idx, int N, int IH, int IW, int k, int s, int p,
                              const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                              float* part, float* sums, float* dgamma, float* dbeta, float* dy, int C, int act, float slope,
                              int training, float* amax, void* stream);
/* (ABI 15) the pooled gradient as two addends, summed where it is loaded (dpool2 may be NULL) */
int viai_bn_act_pool_bwd_amax2(const float* dpool, const float* dpool2, const unsigned char* idx, int N, int IH, int IW, int k, int s, int p,
                               const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                               float* part, float* sums, float* dgamma, float* dbeta, float* dy, int C, int act, float slope,
                               int training, float* amax, void* stream);
int viai_relu_bwd(const float* g, const float* out, float* d, long n, void* stream);

/* ----------------------------------------------------------------------- losses
 * GANLoss = BCELoss / MSELoss against an expanded scalar label (loss_functions.py:79-104);
 * L1 is the `loss_mel_L1_item` metric (train_whole_sync.py:111).  Each forward
 * writes ONE float (mean reduction) to `loss`; `part` is scratch of
 * viai_reduce_blocks(n) floats.  Backward multiplies by the scalar *gscale
 * (device pointer; e.g. d(total)/d(loss)).                                       */
int viai_reduce_blocks(long n);
int viai_bce_fwd(const float* p, float target, long n, float* part, float* loss, void* stream);
int viai_bce_bwd(const float* p, float target, long n, const float* gscale, float* dp, void* stream);
int viai_mse_fwd(const float* p, float target, long n, float* part, float* loss, void* stream);
int viai_mse_bwd(const float* p, float target, long n, const float* gscale, float* dp, void* stream);
int viai_l1_fwd(const float* a, const float* b, long n, float* part, float* loss, void* stream);
int viai_l1_bwd(const float* a, const float* b, long n, const float* gscale, float* da, void* stream);

/* L2ContrastiveLoss (loss_functions.py:107-148): scores[a][b] = ||f1[a]-f2[b]||_2 (n x n, kept for the backward),
 * loss = (sum_{a!=b} clamp(margin - s_ab, 0)^2 [or max over b per row if max_violation] + sum_a s_aa^2) / (2n).
 * argmax_ws: n ints of scratch (max_violation only).                                                          */
int viai_l2c_fwd(const float* f1, const float* f2, int n, int d, float margin, int max_violation,
                 float* scores, float* loss, void* stream);
int viai_l2c_bwd(const float* f1, const float* f2, const float* scores, int n, int d, float margin,
                 int max_violation, const float* gscale, int* argmax_ws, float* df1, float* df2, void* stream);

/* ------------------------------------------------------------------- WaveNet
 * (wavenet_vocoder/{wavenet,modules,mixture}.py).  The dilated causal and 1x1 Conv1d layers are
 * viai_conv2d_* with kh = 1 on (B, 1, T, C) tensors; the entry points below are the remaining pieces.  */
/* torch weight_norm(dim=0): w[r][:] = g[r] * v[r][:] / ||v[r]||  (modules.py:39,59); norm: [rows] kept for bwd */
int viai_weight_norm_fwd(const float* v, const float* g, float* w, float* norm, int rows, int L, void* stream);
int viai_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm, float* dv, float* dg,
                         int rows, int L, int accumulate, void* stream);
/* gated activation tanh(a + ca) * sigmoid(b + cb), y/yc rows = [a | b] of 2*H channels (modules.py:183-201);
 * the backward writes ONE tensor that is the gradient of both y and yc                                 */
int viai_glu_fwd(const float* y, const float* yc, float* z, long rows, int H, void* stream);
int viai_glu_bwd(const float* dz, const float* y, const float* yc, float* dy, long rows, int H, void* stream);
/* out = (a + b) * s (b may be NULL): residual / skip scaling by sqrt(0.5) (modules.py:209, wavenet.py:222-226) */
int viai_add_scale(const float* a, const float* b, float* out, float s, long n, void* stream);
int viai_relu_fwd(const float* a, float* out, long n, void* stream);
/* first_conv for scalar input, Conv1d1x1(1, C): y[p][c] = x[p]*w[c] + b[c] (wavenet.py:118) and its weight grads */
int viai_outer_fwd(const float* x, const float* w, const float* b, float* y, long rows, int C, void* stream);
int viai_outer_bwd_blocks(long rows);
int viai_outer_bwd(const float* dy, const float* x, float* part, float* dw, float* db, long rows, int C, int accumulate, void* stream);
/* conditioning up-sampler: ConvTranspose2d(1,1,(KH,S),stride (1,S),padding ((KH-1)/2,0)) + ReLU on (B,F,T)
 * (wavenet.py:153-164); part: (KH*16+1)*viai_upsample_bwd_blocks() floats                               */
int viai_upsample_fwd(const float* x, const float* w, const float* bias, float* y, int B, int F, int T, int KH, int S, void* stream);
int viai_upsample_bwd_blocks(void);
int viai_upsample_bwd(const float* dy, const float* y, const float* x, const float* w, float* part, float* dx, float* dw, float* db,
                      int B, int F, int T, int KH, int S, int accumulate, void* stream);
/* DiscretizedMixturelogisticLoss (mixture.py:25-105 + loss_functions.py:43-62): yhat rows of `pitch` floats
 * ([logit|mean|log_scale] x nr_mix first), target y[row], optional mask[row];  loss = sum(l*mask)/sum(mask);
 * dyhat (optional) = d loss / d yhat.  loss_rows, wrow: [rows] scratch.                                   */
int viai_mol_loss(const float* yhat, const float* y, const float* mask, float* loss_rows, float* wrow, float* loss,
                  float* dyhat, long rows, int pitch, int nr_mix, float num_classes, float log_scale_min, void* stream);
int viai_scale_by_scalar(float* d, const float* gscale, long n, void* stream);
/* sample_from_discretized_mix_logistic (mixture.py:117-153) with injected uniforms u1[rows][nr_mix], u2[rows] */
int viai_mol_sample(const float* yhat, const float* u1, const float* u2, float* out, long rows, int pitch, int nr_mix,
                    float log_scale_min, void* stream);

/* Incremental synthesis, ONE time step (wavenet.py:322-357 loop body; conv.py:17-46 linearised convolutions).
 * Weights are plain (weight norm already applied) and linearised: w_conv[g][j*C + ci] = w[g][ci][j].
 * `step` is a device int (time index, advanced by the call); ring buffers are (B, ring_len, C) zero-initialised,
 * ring_len > 2*dilation.  All arguments are static across steps, so the call can be captured in a hipGraph.
 * B in {1,2,4,8}.  cond: (B, T, cin) up-sampled conditioning; u1 (B,T,out_ch/3), u2 (B,T): uniforms for the
 * sampler; test_inputs (B, n_test) teacher-forced prefix; out (B, T).                                       */
typedef struct viai_wn_layer {
    const float *w_conv, *b_conv, *w_c, *b_c, *w_out, *b_out, *w_skip, *b_skip;
    float* ring;
    int dilation, ring_len;
    const float* g_add;     /* optional [B][G]: the time-invariant gate contribution of global conditioning,
                               conv1x1g(embed_speakers(g)) + bias (modules.py:195-199); NULL without it (ABI v3) */
    /* fused stages (ABI v7, viai_wn_synth.fused): extended gate rows [G][K] = [Wc^0 | Wc^1 | Wc^2] for layer 0 (K = 3 C) and
     * [Wc^0 | Wc^1 | r Wc^2 | r Wc^2 Wo_prev] for layer l > 0 (K = 3 C + G/2, r = sqrt(.5), Wo_prev = the previous layer's w_out), and
     * the folded bias [G] = b_conv + b_c (+ r Wc^2 bo_prev): gate_l then needs only z_{l-1} and x_{l-1}(t), i.e. ONE dependent launch
     * per layer instead of two                                                                                                      */
    const float *w_stage, *b_stage;
} viai_wn_layer;
typedef struct viai_wn_synth {
    int B, C, G, S, cin, n_layers, out_ch, T, n_test;
    float log_scale_min;
    const viai_wn_layer* layers;            /* HOST array of n_layers descriptors (device pointers inside) */
    const float *w_first, *b_first, *w_l1, *b_l1, *w_l2, *b_l2;
    const float *cond, *test_inputs, *u1, *u2;
    float *out, *z, *skips, *yhat_dbg;      /* z: (B, G/2), skips: (B, S) scratch; yhat_dbg optional (B,T,out_ch) */
    int* step;                              /* device int, 0 before the first call: counts the calls (the step advances it itself) */
    float* z2;                              /* fused stages: second (B, G/2) buffer (z is double-buffered) */
    int fused;                              /* 1: run the fused-stage form (needs layers[].w_stage / b_stage, z2; S, out_ch, G/2 <= 256) */
} viai_wn_synth;
int viai_wavenet_synth_step(const viai_wn_synth* s, void* stream);
/* The same time steps t0 .. t0 + n_steps - 1 launched from a host loop with the time index passed BY VALUE (ABI v5): no kernel starts
 * with a load of `*step` in front of its address arithmetic (a full memory round trip at these grid sizes).  `step` is not touched.
 * wavenet.py:237-364 incremental_forward's loop body, n_steps at a time.                                                         */
int viai_wavenet_synth_run(const viai_wn_synth* s, int t0, int n_steps, void* stream);

/* Incremental synthesis as ONE persistent launch (ABI v16, csrc/wavenet_pipe.hip): a weight-stationary pipeline for the reference-size
 * network (24 layers, 512 residual / 512 gate / 256 skip channels, 80 conditioning channels, 30 outputs, no global conditioning).  Every
 * stage owns compute units (10 per layer, 4 + 4 + 1 for the head) and keeps its weights in registers / LDS; the streams travel round the
 * stages as tokens of 8-byte {tag, value} granules, so up to B stages work at once where the chain form (viai_wavenet_synth_run) has one.
 * Same time steps, same folded weights (layers[].w_stage / b_stage), fp32, fixed summation order; replaces the per-step loop of
 * wavenet.py:322-357 for this configuration.
 *   viai_wn_pipe_ok            1 if `s` is that configuration with 1 .. 32 streams and the device has >= 256 compute units (all blocks must be resident)
 *   viai_wn_pipe_image_floats  sizes of the five weight images the HOST packs (viai_amd.wavenet._pipe_images): 0 wreg [24][10][8][156][64],
 *                              1 wlds [24][10][130][260], 2 bias [24][10][136], 3 head_w [544][256], 4 head_b [544], 5 wcond [24][10][64][80]
 *   viai_wn_pipe_token_granules  8-byte granules of the token rings for B streams (dil: the 24 dilations)
 *   viai_wn_pipe_run           time steps [t0, t0 + n_steps) of every stream.  tok: the rings, ZERO before t0 == 0 and carried over between
 *                              calls; err: 4 zeroed uint32, err[0] != 0 afterwards = failure (1: a wait timed out at stage / stream / t =
 *                              err[1..3]) -- the caller must check it after synchronising.                                                 */
int viai_wn_pipe_ok(const viai_wn_synth* s);
/* debug aid (tools/wn_pipe_stamps.py): later viai_wn_pipe_run calls record wall-clock stamps (100 MHz) of time step t on compute unit 0 of every stage
 * into buf, 2 x [27 stages][8 streams][8] uint64 (wall clock, then shader cycles; 0 wait begins / 1 x part complete / 2 z part complete / 3 published / 4 .. 7 inside a layer
 * stage: residual rows done, past barrier 1, gate rows done, past barrier 2); buf = NULL switches it off */
int viai_wn_pipe_profile(void* buf, int t);
long viai_wn_pipe_image_floats(int which);
long viai_wn_pipe_token_granules(int B, const int* dil);
int viai_wn_pipe_run(const viai_wn_synth* s, const float* wreg, const float* wcond, const float* wlds, const float* bias, const float* head_w, const float* head_b,
                     void* tok, unsigned* err, int t0, int n_steps, void* stream);

/* ------------------------------------------------------------ mask / optimizer
 * s_in = s * mask, mask (N, T) broadcast over frequency (the missing
 * AudioModel.set_inputs; figure misc/pipeline2.png)                             */
int viai_mask_mul(const float* s, const float* mask, float* out, int N, int F, int T, void* stream);
/* the loss scalars an iteration reports (train_whole_sync.py:85-112 reads them through get_loss_items / get_current_errors), from the
 * device scalars the loss kernels left: out[0] = 0.5 (d_fake + d_real), out[1] = g_gan + lambda_l1 l1 (+ lambda_c contrast), out[2] = g_gan,
 * out[3] = l1, out[4] = d_real, out[5] = contrast (only when contrast != NULL).  One launch instead of ten one-element kernels.     */
int viai_step_scalars(const float* d_real, const float* d_fake, const float* g_gan, const float* l1, const float* contrast,
                      float lambda_l1, float lambda_c, float* out, void* stream);
/* torch.optim.Adam step on a flat fp32 arena (optimizer_G / optimizer_D,
 * utils/util.py:149-150).  `state` is 4 device doubles {step, lr, beta1^t, beta2^t}
 * (initialise to {0, lr, 1, 1}) advanced ON DEVICE so that the launch is
 * replayable inside a hipGraph.                                                  */
int viai_adam_step(float* p, const float* g, float* m, float* v, long n, double* state,
                   double beta1, double beta2, double eps, float grad_scale, void* stream);

/* out[c] (+)= sum over rows of x[M][C] (bias gradients); part: scratch of
 * viai_colsum_blocks(M, C) * C floats                                            */
int viai_colsum_blocks(long M, int C);
int viai_colsum(const float* x, long M, int C, float* part, float* out, int accumulate, void* stream);

/* Debug aid for the f16x2 conv arithmetic (operands are pre-scaled by powers of two and SATURATE beyond the fp16 range:
 * activations above 65504 / 16 ~ 4094, weights above 65504 / 256 ~ 255): counts[0] += #{|x_i| > limit},
 * counts[1] = max(counts[1], bit pattern of max |x_i|), counts[2] += #{non-finite x_i}.  counts: 3 zero-initialised uint32. */
int viai_range_count(const float* x, long n, float limit, unsigned* counts, void* stream);

/* y = a*x + y (flat); used for gradient accumulation on arenas */
int viai_axpy(float a, const float* x, float* y, long n, void* stream);

/* ----------------------------------------------------------------- audio front
 * melspectrogram(y) of utils/audio.py:70-75 (lws STFT -> |.| -> mel -> dB -> [0,1])
 * fused with the inpainting mask.  basis_t: [fft/2+1][n_mels] (TRANSPOSED mel basis); window: [fft]; mask: [B][frames] or NULL.
 * wav: [B][n_samples]; mel out: [B][n_mels][frames] (== NHWC with C = 1).        */
int viai_stft_mel(const float* wav, const float* window, const float* basis_t, const float* mask,
                  float* mel, int B, int n_samples, int fft, int hop, int n_mels, int frames,
                  float min_level_db, float ref_level_db, void* stream);
/* The same stage (utils/audio.py:70-75) on the frame-batched kernel, fft = 1024 (ABI v4): eight frames per block, two real frames per
 * complex radix-4 FFT, and a BANDED mel basis -- the caller states the support of each band of `basis_t`:
 * basis_t[k][m] == 0 outside k in [band_lo[m], band_lo[m] + band_cnt[m])  (librosa.filters.mel rows are triangles a few bins wide;
 * a dense basis is expressed by band_lo = 0, band_cnt = fft/2 + 1).  Same outputs as viai_stft_mel.                                  */
int viai_stft_mel_banded(const float* wav, const float* window, const float* basis_t, const int* band_lo, const int* band_cnt,
                         const float* mask, float* mel, int B, int n_samples, int fft, int hop, int n_mels, int frames,
                         float min_level_db, float ref_level_db, void* stream);

/* -------------------------------------------------- callers either side of the path
 * Batch assembly on the device, Data_loaders/audio_loader.py:185-245: frames uint8 [n][S][S][C] (RGB or
 * (flow_x, flow_y), already resized to S x S) -> float NHWC4 [n][size][size][4] = (px - 127) / 128 after the
 * left-right flip and the crop of rows [crop_x, +size) x columns [crop_y, +size).                           */
int viai_frames_prep(const unsigned char* frames, float* out, long n, int S, int C, int size,
                     int crop_x, int crop_y, int flip, void* stream);
/* audio_loader.py:471-475,508,523: clip b = mel frames [3 + 4*start[b], +L) of c [T_total][D] -> c_out [B][D][L]
 * (channel first), and samples [(3 + 4*start[b]) * hop, +L*hop) of x -> x_out [B][L*hop]; zero padded past the end. */
int viai_slice_clips(const float* c, const float* x, const int* start, float* c_out, float* x_out,
                     int B, int D, int L, int hop, long T_total, long samples, void* stream);
/* loss_functions.py:65-76 ExponentialMovingAverage.update on a flat buffer: shadow -= (1 - decay) * (shadow - x) */
int viai_ema_update(float* shadow, const float* x, long n, double decay, void* stream);
/* utils/audio.py:135-144: out = 10 ^ ((clip(S, 0, 1) * -min_level_db + min_level_db) / 20)  (_denormalize, _db_to_amp) */
int viai_mel_denorm_amp(const float* S, float* out, long n, float min_level_db, void* stream);
/* utils/util.py:99-121 L2retrieval: for caption i, ranks[i] = position of clip i in the ascending order of
 * |captions[i] - clips[j]|_2 over j, top1[i] = argmin_j; dist (optional) [n_captions][n_clips].             */
int viai_l2_ranks(const float* clips, const float* captions, int n_clips, int n_captions, int dim,
                  int* ranks, int* top1, float* dist, void* stream);

/* Fused pair: (conv + BatchNorm + activation) -> (3 x 3, stride 1, pad 1 conv or stride-1 transposed conv with ONE output channel).
 * Reference pairs: G.conv6_1 + conv6_1_bn + ReLU -> conv6_2 (New_Inpainting_Networks.py:85-88, 134 MB per tensor at the benchmark size)
 * and D.conv3 + norm3 + LeakyReLU -> conv4 (Discriminator_Networks.py:44-49).  Neither the front layer's post-activation tensor z nor the
 * Cout = 1 layer's data gradient dz is ever stored: the Cout = 1 kernels read the front layer's PRE-BatchNorm output y and apply
 * z = act_in(scale * y + shift) on load; the BatchNorm backward forms dz = conv_backward_data(du, w) -- nine float4 FMAs per element from
 * a register window of the one-channel du -- instead of reading it back twice.  `c` describes the Cout = 1 layer (C1 = the front layer's
 * channels: 32 .. 512, width a multiple of 16); wp = its packed image (viai_conv2d_pack_fwd); du = gradient of its PRE-activation output.
 *   fwd:    out = act(conv(z, w) + bias)
 *   wgrad:  dw (+)= conv_backward_weight(z, du);  ws: viai_conv2d_wgrad_ws_bytes(c)
 *   bn_bwd: the front layer's BatchNorm + activation backward (viai_bn_act_bwd_amax semantics: sums = {k0, k1}, dgamma, dbeta, dy, max |dy|);
 *           part: 2 * C1 * viai_pair_cout1_bn_bwd_blocks(c) floats                                                                          */
int viai_pair_cout1_ok(const viai_conv2d* c);
int viai_pair_cout1_bn_bwd_blocks(const viai_conv2d* c);
int viai_pair_cout1_fwd(const viai_conv2d* c, const float* y, const float* scale, const float* shift, int act_in,
                        const float* wp, const float* bias, float* out, int act, void* stream);
/* fwd through a per-pixel tensor of the nine tap products (ws: 9 * N * IH * IW floats): one grid-stride pass over y + a gather -- the form
 * wide front layers take (D.conv3 -> conv4, 512 channels), where a block of whole rows leaves too few pixels in flight (ABI 11)            */
int viai_pair_cout1_fwd_dots(const viai_conv2d* c, const float* y, const float* scale, const float* shift, int act_in,
                             const float* wp, const float* bias, float* ws, float* out, int act, void* stream);
int viai_pair_cout1_wgrad(const viai_conv2d* c, const float* y, const float* scale, const float* shift, int act_in,
                          const float* du, float* ws, float* dw, int accumulate, void* stream);
int viai_pair_cout1_bn_bwd(const viai_conv2d* c, const float* du, const float* wp, const float* y, const float* mean,
                           const float* invstd, const float* scale, const float* shift, int act_in, float* part, float* sums,
                           float* dgamma, float* dbeta, float* dy, int training, float* dy_amax, void* stream);

/* -------------------------------------------------- fused Cin = 1 conv + BatchNorm2d(train) + activation layer (ABI v5)
 * MelEncoder.conv1 + bn1 + LeakyReLU (Inpainting_Networks.py:55,71) and MelDiscriminator's first block (Discriminator_Networks.py:17-19,
 * 38-39): with 4 .. 9 taps and one input channel the convolution is cheaper to RECOMPUTE than its 32 / 64-channel output is to write
 * and read back, so the pre-BatchNorm tensor y never exists in memory (same arithmetic, same order as viai_conv2d_fwd + viai_bn_*):
 *   fwd(z = NULL): BatchNorm partials (layout of viai_conv2d_fwd's stat_part, geometry of viai_conv2d_stat_geom) -> viai_bn_finalize
 *   fwd(stat_part = NULL): z = act(scale * conv(x) + shift); z_amax (optional, zero-initialised device float) receives max |z|
 *   bwd: partial sums from (dz, recomputed y) -> sums = {k0, k1}, dgamma, dbeta;  dy only if a data gradient needs it in memory
 *        (`training` as in viai_bn_act_bwd; part: 2 * Cout * nblk floats, nblk from viai_conv2d_stat_geom)
 *   wgrad: dw (+)= sum dy * x with dy formed on the fly from dz, y and sums
 * w: the packed image of viai_conv2d_pack (= the torch layout for this kind).
 * x_mask (optional, (N, IW) floats): x is read as x[n][iy][ix] * x_mask[n][ix] -- the time mask of the inpainting step (s_in = s * mask,
 * the missing AudioModel.set_inputs; misc/pipeline2.png) applied where E.conv1 loads s, not in a pass of its own (ABI 8).           */
int viai_conv2d_cin1_bn_ok(const viai_conv2d* c);
int viai_conv2d_cin1_bn_fwd(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, float* stat_part,
                            const float* scale, const float* shift, float* z, int act, float* z_amax, void* stream);
int viai_conv2d_cin1_bn_bwd(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, const float* dz,
                            const float* mean, const float* invstd, const float* scale, const float* shift, float* part,
                            float* sums, float* dgamma, float* dbeta, float* dy, int act, int training, void* stream);
int viai_conv2d_cin1_bn_wgrad(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias, const float* dz,
                              const float* mean, const float* scale, const float* shift, const float* sums, float* ws,
                              float* dw, int accumulate, int act, void* stream);
/* dx straight from dz (dy formed per contributing output pixel): no dy tensor at all; needs bias == NULL */
int viai_conv2d_cin1_bn_dgrad(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* dz, const float* mean,
                              const float* scale, const float* shift, const float* sums, float* dx, int act, void* stream);

/* -------------------------------------------------- (ABI 13) pre-split activations and gradients: the P16 layout
 * Between a BatchNorm pass and the f16x2 conv kernels that consume its output nothing needs fp32: every consumer (the next layer's forward,
 * this layer's data gradient, the weight gradients on either side) splits each value into two fp16 terms, value * S = lead + rem, S a power
 * of two, while it stages the tile -- up to four times per tensor and step, 6 VALU per element pair each time, in kernels that rocprofv3
 * shows issue-bound (6 - 7 non-MFMA VALU per MFMA in the stride-2 and 32-channel families, profiles/r04_a_pmc_*.json).  A P16 tensor holds
 * the two terms instead of the fp32 value, in the SAME bytes: shape (N, H, W, C) "fp32", C % 32 == 0; per pixel and 32-channel group g the
 * 128 bytes at g * 128 are [32 x fp16 lead (64 B)][32 x fp16 rem (64 B)].  A consumer stages 16-byte PIECES (8 channels of one plane) where
 * it staged 16-byte channel quads: same addresses, no conversion, one 16-byte LDS store.
 * The scale S = 2^(14 - e), magnitude < 2^e, must be known BEFORE the pass that writes the planes, so it is derived from an a-priori BOUND
 * of the tensor (written to *amax for the consumers), not from its measured maximum:
 *   forward (training-mode statistics over m_stat samples):  |act(gamma xhat + beta)| <= |gamma| sqrt(m_stat - 1) + |beta|   (Samuelson)
 *   backward: |dy| <= |scale| max|dpre| + |k1| sqrt(M - 1) / invstd + |k0| per channel, max|dpre| reduced with the sums
 * fp16 keeps 11 bits at every magnitude down to 2^-14 S^-1, so values 2^10 below the bound still carry 22 bits (tests/test_p16_gpu.py).
 * The arithmetic is that of the fp32-input kernels handed the same *amax, bit for bit.
 * Reference semantics: the tensors between nn.BatchNorm2d / LeakyReLU and the next nn.Conv2d (Discriminator_Networks.py:38-46,
 * New_Inpainting_Networks.py:31-37) -- an internal format; the module API keeps returning fp32.                                       */
#define VIAI_P16_DY 1              /* viai_conv2d_wgrad_f16_p16 flags: dy is P16 (scale from *dy_amax) */
#define VIAI_P16_X 2               /*                                  x is P16 (scale from *x_amax; no x2) */
#define VIAI_P16_OK_FWD_X 1        /* viai_conv2d_p16_ok mask: the forward kernel takes a P16 x */
#define VIAI_P16_OK_DGRAD_DY 2     /*                          the data-gradient kernel takes a P16 dy */
#define VIAI_P16_OK_WGRAD_DY 4     /*                          the weight-gradient kernel takes a P16 dy */
#define VIAI_P16_OK_WGRAD_X 8      /*                          ... and a P16 x */
#define VIAI_P16_OK_FWD_LIN 16     /*                          the P16 forward writes its BatchNorm partials per 128 CONSECUTIVE pixels
                                                                (viai_bn_finalize with rows = 128, nblk = M / 128), whatever viai_conv2d_stat_geom says of the fp32-input launch */
int viai_conv2d_p16_ok(const viai_conv2d* c);
/* z (P16) = act(scale * y + shift); gamma / beta (NULL = 1 / 0) and m_stat give the bound; *z_amax receives it.  act: none / ReLU / LeakyReLU */
int viai_bn_act_fwd_p16(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                        float* z, long M, int C, int act, float slope, float* z_amax, void* stream);
/* viai_bn_add_act_fwd_amax (the residual join of networks/ResNet.py:49-53) writing z twice: fp32 (z, exact maximum to *z_amax: the next join's residual, the
 * backward's mask) and P16 (z_p16, bound |gamma| sqrt(m_stat - 1) + |beta| + *res_amax to *p_amax: what the next block's conv1 stages).  C / 4 a power of two <= 256 */
int viai_bn_add_act_fwd_twin(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                             const float* res, const float* res_amax, float* z, float* z_p16, long M, int C, int act, float slope,
                             float* z_amax, float* p_amax, void* stream);
/* viai_bn_act_maxpool_fwd (3 x 3 / stride 2 / pad 1 only) writing the pooled tensor twice: fp32 (out, exact maximum to *out_amax) and P16 (out_p16, bound to
 * *p_amax) -- the stem of networks/Image_Embedding.py:20-23 in front of the first BasicBlock.  C % 32 == 0 */
int viai_bn_act_maxpool_fwd_twin(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                 float* out, float* out_p16, unsigned char* idx, int N, int IH, int IW, int C, int k, int s, int p, int act, float slope,
                                 float* out_amax, float* p_amax, void* stream);
/* viai_bn_act_bilinear_fwd_amax with the resized tensor written as P16 */
int viai_bn_act_bilinear_fwd_p16(const float* y, const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                 float* out, int N, int IH, int IW, int OH, int OW, int C, int act, float slope, float* z_amax, void* stream);
/* the apply pass of the fused Cin = 1 layer (viai_conv2d_cin1_bn_fwd with z != NULL) writing z as P16 */
int viai_conv2d_cin1_bn_fwd_p16(const viai_conv2d* c, const float* x, const float* x_mask, const float* w, const float* bias,
                                const float* scale, const float* shift, const float* gamma, const float* beta, long m_stat,
                                float* z, int act, float* z_amax, void* stream);
/* viai_pair_cout1_bn_bwd with dy written as P16 (part: 3 * C * blocks floats, sums: 3 * C floats) */
int viai_pair_cout1_bn_bwd_p16(const viai_conv2d* c, const float* du, const float* wp, const float* y, const float* mean,
                               const float* invstd, const float* scale, const float* shift, int act_in, float* part, float* sums,
                               float* dgamma, float* dbeta, float* dy, int training, float* dy_amax, void* stream);
/* viai_bn_act_bwd_amax with dy written as P16; part: 3 * C * viai_bn_bwd_blocks(M, C) floats, sums: 3 * C floats; *amax receives the bound */
int viai_bn_act_bwd_p16(const float* dz, const float* y, const float* mean, const float* invstd,
                        const float* scale, const float* shift, float* part, float* sums,
                        float* dgamma, float* dbeta, float* dy, long M, int C, int act, float slope,
                        int training, float* amax, void* stream);
/* (ABI 15) the same with the fp32 tensor written beside the planes (dy32: M x C floats, the values of viai_bn_act_bwd_amax): for a layer
 * whose data-gradient kernel takes fp32 while its weight-gradient kernel takes planes */
int viai_bn_act_bwd_p16_twin(const float* dz, const float* y, const float* mean, const float* invstd,
                             const float* scale, const float* shift, float* part, float* sums,
                             float* dgamma, float* dbeta, float* dy, float* dy32, long M, int C, int act, float slope,
                             int training, float* amax, void* stream);
/* (ABI 15) the BatchNorm backward behind a residual join with a ReLU (networks/ResNet.py:46-53): the gradient of the join's output arrives as
 * dz (+ dz2, may be NULL); dres = (dz + dz2) * [zj > 0] is written (M x C floats: the residual branch's gradient) and stands in for dz in
 * viai_bn_act_bwd_p16 with act = none -- the same values as viai_add_act_bwd_from_output + viai_bn_act_bwd_p16, one pass over memory fewer */
int viai_bn_join_bwd_p16(const float* dz, const float* dz2, const float* zj, float* dres, const float* y, const float* mean, const float* invstd,
                         const float* scale, const float* shift, float* part, float* sums, float* dgamma, float* dbeta, float* dy,
                         long M, int C, int training, float* amax, void* stream);
/* x (fp32) = the values a P16 tensor holds, (lead + rem) / S(*amax) */
int viai_p16_decode(const float* p16, float* x, long M, int C, const float* amax, void* stream);
/* viai_conv2d_fwd_amax / viai_conv2d_dgrad_f16 with the gathered tensor pre-split (one source; scale from *x_amax / *dy_amax) */
int viai_conv2d_fwd_p16(const viai_conv2d* c, const float* x, const float* wp_fwd, const float* bias, float* y, float* stat_part,
                        int act, const float* x_amax, void* stream);
int viai_conv2d_dgrad_f16_p16(const viai_conv2d* c, const float* dy, const float* wp, float* dx, float* dx2,
                              const float* dy_amax, void* stream);
/* viai_conv2d_wgrad_f16 with pre-split operands (flags: VIAI_P16_DY | VIAI_P16_X); db must be NULL when dy is P16 */
int viai_conv2d_wgrad_f16_p16(const viai_conv2d* c, const float* x, const float* x2, const float* dy,
                              float* ws, float* dw, float* db, int accumulate, const float* dy_amax, const float* x_amax, int flags, void* stream);

/* -------------------------------------------------- launch plans: the train step recorded once, replayed from C
 * Replaces the per-launch host work of `model.optimize_parameters()` (train_whole_sync.py:76).  Protocol:
 *   viai_plan_log_begin();  <stream-capture the step: every library launch notes (kernel, stream)>;  n = viai_plan_log_end();
 *   viai_plan_build(captured hipGraph_t, capture origin stream, &plan);   // reads nodes + edges, never instantiates the graph
 *   viai_plan_replay(plan, stream);                                       // each step: same kernels, arguments, streams, edges
 * The hipGraph_t (it owns the kernel-argument arrays) and every buffer the capture touched must outlive the plan.
 * Kernel, 1-D copy, fill and empty nodes are accepted; anything else returns hipErrorNotSupported.
 * viai_plan_info fills out[0..n) with: nodes, kernels, kernels with a noted stream, copies, fills, streams, events, waits.   */
typedef struct viai_plan viai_plan;
int viai_plan_log_begin(void);
int viai_plan_log_end(void);
int viai_plan_build(void* hip_graph, void* capture_stream, viai_plan** plan);
int viai_plan_replay(viai_plan* plan, void* stream);
int viai_plan_info(const viai_plan* plan, int* out, int n);
void viai_plan_destroy(viai_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* VIAI_HIP_H */
