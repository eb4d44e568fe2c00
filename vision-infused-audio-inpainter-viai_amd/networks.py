"""nn.Module shells over the HIP ops: the reference's generator / discriminator API.

Same constructor signatures, forward I/O and `state_dict()` keys/shapes as
  MelEncoder        networks/Inpainting_Networks.py:49-88
  TransConvBlock    networks/New_Inpainting_Networks.py:12-45
  MelDecoder        networks/New_Inpainting_Networks.py:48-89
  MelDiscriminator  networks/Discriminator_Networks.py:9-50
so checkpoints written by `utils/util.save_inpainting_checkpoint` load here and
vice versa.  torch's nn.Conv2d / nn.ConvTranspose2d / nn.BatchNorm2d objects are
used ONLY as parameter/buffer holders (their own forward is never called): the
arithmetic is `ops.conv_bn_act` etc., i.e. libviai_hip.so.

Layout: inputs/outputs are NCHW-shaped like the reference's; internally the
data is NHWC.  Tensors returned to the caller are NCHW *views* of NHWC storage
(torch channels_last strides) — same shape, same values, no copy.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID


class DefaultHParams:
    """The few attributes the reference reads from its (missing) Options_inpainting.Inpainting_Config."""
    cin_channels = 256
    max_mel_lengths = 256
    normlayer = nn.BatchNorm2d
    length_feature = 256
    image_size = 224


hparams = DefaultHParams()


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) any strides -> (N,H,W,C) contiguous (free when already channels-last or C == 1)."""
    y = x.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


def to_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    return x_nhwc.permute(0, 3, 1, 2)


def _move_to_end(d, keys):
    for k in keys:
        d[k] = d.pop(k)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _check_norm(norm_layer):
    """`norm_layer` of the reference constructors (Inpainting_Networks.py:52, New_Inpainting_Networks.py:16,
    Discriminator_Networks.py:14): nn.BatchNorm2d (hparams.normlayer, the trained configuration) or nn.InstanceNorm2d."""
    if norm_layer not in (nn.BatchNorm2d, nn.InstanceNorm2d):
        raise NotImplementedError("the HIP path implements nn.BatchNorm2d and nn.InstanceNorm2d; got %r" % (norm_layer,))


class _InstanceNormHolder:
    """What ops.conv_bn_act reads from a norm module, for an nn.InstanceNorm2d: instance statistics are batch statistics of a
    one-sample batch, so the layer runs the conv + BatchNorm(train) kernels once per sample (no running statistics; gamma / beta
    are the module's when affine, constant 1 / 0 otherwise)."""
    track_running_stats = False
    running_mean = running_var = num_batches_tracked = None
    training = True

    def __init__(self, inorm, device):
        if inorm.track_running_stats:
            raise NotImplementedError("InstanceNorm2d(track_running_stats=True)")
        self.eps, self.momentum = inorm.eps, 0.1
        c = inorm.num_features
        self.weight = inorm.weight if inorm.affine else torch.ones(c, device=device)
        self.bias = inorm.bias if inorm.affine else torch.zeros(c, device=device)


def _instance_norm_layer(x, conv, inorm, act, x2, transposed):
    h = getattr(inorm, "_viai_holder", None)
    if h is None or h.weight.device != x.device:
        h = _InstanceNormHolder(inorm, x.device)
        object.__setattr__(inorm, "_viai_holder", h)
    if inorm.affine:
        h.weight, h.bias = inorm.weight, inorm.bias
    outs = [ops.conv_bn_act(x[i:i + 1], conv.weight, conv.bias, h, kernel=_pair(conv.kernel_size), stride=_pair(conv.stride),
                            padding=_pair(conv.padding), transposed=transposed, act=act,
                            x2=(x2[i:i + 1] if x2 is not None else None), training=True) for i in range(x.shape[0])]
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


_FLOW_STREAMS = {}
FLOW_STREAM = True           # ImageEmbedding: the flow ResNet on its own stream (module switch)
FUSE_BN_TAIL = True      # BatchNorm apply + (residual add + ReLU | ReLU + max-pool) in one pass (module switch: tests/test_resnet_gpu.py flips it)


# Residual joins / the stem's pool can write a pre-split (P16) twin of their output for the next BasicBlock's conv1 (ops.conv_bn_act `out_p16` on a layer with
# `residual` / `pool`).  Round 4, on the register-staged kernels: kernel-time sum 224 -> 212 ms, the step did not move (118.7 / 119.0 ms) -- OFF then.  Round 5: a
# pre-split conv1 runs on the linear-tile loader / consumer kernel (csrc/conv_halo_dma.hip), 110.1 -> 108.1 ms on the vision-infused step -- ON.
P16_TWIN = True


def takes_p16(x_shape, conv):
    """will `conv` (the layer behind a BatchNorm pass) stage a pre-split (P16) input of this NHWC shape?  (ops.conv_takes_p16)"""
    tr = isinstance(conv, nn.ConvTranspose2d)
    if not isinstance(conv, (nn.Conv2d, nn.ConvTranspose2d)) or getattr(conv, "groups", 1) != 1 or _pair(conv.dilation) != (1, 1):
        return False
    if tr and _pair(conv.stride) != (1, 1):
        return False
    return ops.conv_takes_p16(tuple(x_shape), conv.weight, _pair(conv.kernel_size), _pair(conv.stride), _pair(conv.padding), tr)


def wgrad_takes_p16(x_shape, conv):
    """... or at least in its weight-gradient kernel (ops.conv_wgrad_takes_p16: the stride-2 3 x 3 convs of the ResNet branch)?"""
    tr = isinstance(conv, nn.ConvTranspose2d)
    if not isinstance(conv, (nn.Conv2d, nn.ConvTranspose2d)) or getattr(conv, "groups", 1) != 1 or _pair(conv.dilation) != (1, 1) or tr:
        return False
    return ops.conv_wgrad_takes_p16(tuple(x_shape), conv.weight, _pair(conv.kernel_size), _pair(conv.stride), _pair(conv.padding), tr)


def out_shape(x_shape, conv):
    """NHWC shape of conv(x)"""
    N, H, W, _ = x_shape
    k, s, p = _pair(conv.kernel_size), _pair(conv.stride), _pair(conv.padding)
    if isinstance(conv, nn.ConvTranspose2d):
        return (N, (H - 1) * s[0] - 2 * p[0] + k[0], (W - 1) * s[1] - 2 * p[1] + k[1], conv.out_channels)
    return (N, (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1, conv.out_channels)


def fused_layer(x, conv, bn, act, x2=None, training=True, xmask=None, residual=None, pool=None, upsample=None, next_conv=None):
    """`next_conv`: the conv layer that consumes this layer's output (and nothing else does): where its kernels stage pre-split pieces, the
    BatchNorm apply pass writes them (ops.conv_bn_act(out_p16=True)).  THE RESULT MAY THEN BE A P16 TENSOR (`ops.is_p16`): the bytes of an fp32
    tensor holding fp16 planes.  Hand it only to ops.conv_bn_act / conv_bn_act_cout1 (they check the tag); any torch op on it -- detach, clone, view, +, cat,
    a hook that reads values -- drops the tag and reinterprets the planes as fp32.  ops.p16_decode gives the values.
    conv (nn.Conv2d | nn.ConvTranspose2d holder) -> bn (nn.BatchNorm2d | nn.InstanceNorm2d holder | None) -> act, on NHWC.
    `xmask`: the layer convolves x * xmask (ops.conv_bn_act).  `upsample` = (H, W): F.interpolate(.., align_corners=True) behind the
    layer -- inside the BatchNorm apply pass where ops.upsample_fusable allows, as a pass of its own otherwise."""
    transposed = isinstance(conv, nn.ConvTranspose2d)
    if transposed and _pair(conv.stride) != (1, 1):
        raise NotImplementedError("ConvTranspose2d stride != 1")
    if upsample is not None:
        up = (int(upsample[0]), int(upsample[1]))
        if residual is None and pool is None and xmask is None and ops.upsample_fusable(x, conv.weight, conv.bias, bn, transposed, act, up):
            p16 = (next_conv is not None and isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.training
                   and takes_p16((x.shape[0], up[0], up[1], conv.out_channels), next_conv))
            return ops.conv_bn_act(x, conv.weight, conv.bias, bn, kernel=_pair(conv.kernel_size), stride=_pair(conv.stride),
                                   padding=_pair(conv.padding), transposed=transposed, act=act, x2=x2, training=bn.training, upsample=up, out_p16=p16)
        return ops.bilinear_ac(fused_layer(x, conv, bn, act, x2=x2, training=training, xmask=xmask, residual=residual, pool=pool), up)
    if isinstance(bn, nn.InstanceNorm2d):
        if xmask is not None:
            x = ops.mask_mul(x, xmask)
        out = _instance_norm_layer(x, conv, bn, ACT_NONE if residual is not None else act, x2, transposed)
        if residual is not None:
            out = ops.add_relu(out, residual) if act == ACT_RELU else out + residual
        return ops.maxpool(out, *pool) if pool is not None else out
    if FUSE_BN_TAIL and isinstance(bn, nn.modules.batchnorm._BatchNorm) and act in (ACT_RELU, ACT_NONE) and (residual is not None or pool is not None):
        # a residual join whose output feeds `next_conv` AND other readers (the next join, a downsample conv) writes a pre-split twin beside the fp32 tensor
        oshape = out_shape(x.shape, conv)
        if pool is not None:
            oshape = (oshape[0], (oshape[1] + 2 * pool[2] - pool[0]) // pool[1] + 1, (oshape[2] + 2 * pool[2] - pool[0]) // pool[1] + 1, oshape[3])
        twin = P16_TWIN and next_conv is not None and bn.training and (takes_p16(oshape, next_conv) or wgrad_takes_p16(oshape, next_conv))
        return ops.conv_bn_act(x, conv.weight, conv.bias, bn, kernel=_pair(conv.kernel_size), stride=_pair(conv.stride),
                               padding=_pair(conv.padding), transposed=transposed, act=act, x2=x2, training=bn.training, xmask=xmask,
                               residual=residual, pool=pool, out_p16=twin)
    p16 = (next_conv is not None and residual is None and pool is None and isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.training
           and takes_p16(out_shape(x.shape, conv), next_conv))
    out = ops.conv_bn_act(x, conv.weight, conv.bias, bn, kernel=_pair(conv.kernel_size), stride=_pair(conv.stride),
                          padding=_pair(conv.padding), transposed=transposed, act=(ACT_NONE if residual is not None else act), x2=x2,
                          training=(bn.training if bn is not None else training), xmask=xmask, out_p16=p16)
    if residual is not None:
        out = ops.add_relu(out, residual) if act == ACT_RELU else out + residual
    return ops.maxpool(out, *pool) if pool is not None else out


def fused_pair(x, conv, bn, act, conv2, act2, x2=None):
    """conv -> bn -> act -> conv2 (one output channel) -> act2.  Where the pair kernels apply (BatchNorm, conv2 3 x 3 / stride 1 / pad 1,
    width a multiple of 16: G.conv6_1 -> conv6_2 and D.conv3 -> conv4 at every shape the step runs) neither the tensor between the two
    layers nor conv2's data gradient is ever stored (ops.conv_bn_act_cout1); otherwise two fused layers."""
    tr, tr2 = isinstance(conv, nn.ConvTranspose2d), isinstance(conv2, nn.ConvTranspose2d)
    if (isinstance(bn, nn.modules.batchnorm._BatchNorm) and conv2.out_channels == 1 and (not tr or _pair(conv.stride) == (1, 1))
            and _pair(conv2.dilation) == (1, 1) and _pair(conv.dilation) == (1, 1) and getattr(conv2, "groups", 1) == 1
            and ops.conv_bn_act_cout1_ok(x, conv.weight, bn, conv2.weight, kernel=_pair(conv.kernel_size), stride=_pair(conv.stride),
                                         padding=_pair(conv.padding), transposed=tr, kernel2=_pair(conv2.kernel_size),
                                         stride2=_pair(conv2.stride), padding2=_pair(conv2.padding), x2=x2)):
        return ops.conv_bn_act_cout1(x, conv.weight, conv.bias, bn, conv2.weight, conv2.bias, kernel=_pair(conv.kernel_size),
                                     stride=_pair(conv.stride), padding=_pair(conv.padding), transposed=tr, act=act, transposed2=tr2,
                                     act2=act2, x2=x2, training=bn.training)
    return fused_layer(fused_layer(x, conv, bn, act, x2=x2), conv2, None, act2)


class TransConvBlock(nn.Module):
    """`nums` x [ConvTranspose2d 3x3 s1 p1 (no bias) -> BN -> ReLU]; sub-module names
    conv{name}_{i}, conv{name}_{i}_bn (New_Inpainting_Networks.py:12-45)."""

    def __init__(self, inplanes, outplanes, name, nums=3, kernel_size=3, padding=(1, 1), stride=(1, 1),
                 norm_layer=nn.BatchNorm2d):
        super().__init__()
        _check_norm(norm_layer)
        if not isinstance(name, str):
            raise Exception("name should be str")
        self.nums, self.name = nums, name
        use_bias = norm_layer is nn.InstanceNorm2d          # New_Inpainting_Networks.py:17
        c = inplanes
        for i in range(nums):
            self.add_module("conv%s_%d" % (name, i),
                            nn.ConvTranspose2d(c, outplanes, kernel_size=kernel_size, stride=stride, padding=padding, bias=use_bias))
            self.add_module("conv%s_%d_bn" % (name, i), norm_layer(outplanes))
            c = outplanes
        self.initial()

    def initial(self):
        for m in self.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward_nhwc(self, x, x2=None, upsample=None, next_conv=None):
        """`upsample` = (H, W): the F.interpolate the decoder applies to the block's output (New_Inpainting_Networks.py:78,83);
        `next_conv`: the layer that alone consumes the (resized) output, if the caller knows it (fused_layer) -- the result may then be a P16 tensor
        (see fused_layer: consume it with ops.conv_bn_act only)"""
        for i in range(self.nums):
            conv = self._modules["conv%s_%d" % (self.name, i)]
            bn = self._modules["conv%s_%d_bn" % (self.name, i)]
            nxt = self._modules["conv%s_%d" % (self.name, i + 1)] if i + 1 < self.nums else next_conv
            x = fused_layer(x, conv, bn, ACT_RELU, x2=x2 if i == 0 else None, upsample=upsample if i == self.nums - 1 else None, next_conv=nxt)
        return x

    def forward(self, x):
        return to_nchw_view(self.forward_nhwc(to_nhwc(x)))


class MelEncoder(nn.Module):
    """E_a: 5 x [Conv2d 3x3 (no bias) -> BN -> LeakyReLU(0.2)], strides (2,2),(2,1),(2,2),(2,2),(2,2);
    AvgPool2d((3,1)) on the last map; returns the list of 5 maps (Inpainting_Networks.py:49-78)."""

    STRIDES = ((2, 2), (2, 1), (2, 2), (2, 2), (2, 2))
    WIDTHS = (1, 32, 64, 128, 256, 256)

    def __init__(self, hparams=hparams, norm_layer=nn.BatchNorm2d):
        super().__init__()
        _check_norm(norm_layer)
        self.hparams = hparams
        use_bias = norm_layer is nn.InstanceNorm2d          # Inpainting_Networks.py:52
        for i in range(5):
            self.add_module("conv%d" % (i + 1), nn.Conv2d(self.WIDTHS[i], self.WIDTHS[i + 1], (3, 3), self.STRIDES[i], (1, 1), bias=use_bias))
            self.add_module("bn%d" % (i + 1), norm_layer(self.WIDTHS[i + 1], affine=True))
        self.initial()

    def initial(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward_nhwc(self, c, mask=None):
        """c: (B, F, T) or (B,1,F,T) -> list of 5 NHWC maps.  `mask` (B, T) / (B,1,1,T): the encoder sees c * mask (the inpainting
        step's time gap), multiplied where conv1 loads c."""
        b = c.size(0)
        f = c.size(-2) if c.dim() >= 3 else self.hparams.cin_channels
        x = c.reshape(b, f, -1, 1)                      # (B,1,F,T) NCHW == (B,F,T,1) NHWC
        net = []
        for i in range(5):
            x = fused_layer(x, self._modules["conv%d" % (i + 1)], self._modules["bn%d" % (i + 1)], ACT_LRELU, xmask=mask if i == 0 else None)
            net.append(x)
        net[-1] = ops.avgpool_h(net[-1], 3)
        return net

    def forward(self, c):
        return [to_nchw_view(t) for t in self.forward_nhwc(c)]


class MelDecoder(nn.Module):
    """G: stride-1 transposed convs + align-corners bilinear up-sampling to each encoder scale,
    skip concat with e2 at the third scale, Sigmoid (New_Inpainting_Networks.py:48-89).
    `convblock1` exists for state_dict parity but is never used by forward (as in the reference)."""

    UNUSED_PREFIXES = ("convblock1.",)        # registered, never reached by forward: .grad stays None in the reference

    def __init__(self, hparams=hparams, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer if norm_layer is not None else getattr(hparams, "normlayer", nn.BatchNorm2d)
        _check_norm(norm_layer)
        self.hparams = hparams
        self.deconv1_1 = nn.ConvTranspose2d(256, 256, 3, 1, (0, 1))
        self.deconv1_1_bn = norm_layer(256)
        self.deconv1_2 = nn.ConvTranspose2d(256, 256, 3, 1, 1)
        self.deconv1_2_bn = norm_layer(256)
        self.convblock1 = TransConvBlock(256, 256, "1", nums=2, norm_layer=norm_layer)
        self.convblock2 = TransConvBlock(256, 128, "2", nums=3, norm_layer=norm_layer)
        self.convblock3 = TransConvBlock(128, 64, "3", nums=3, norm_layer=norm_layer)
        self.convblock4 = TransConvBlock(64 * 2, 32, "4", nums=3, norm_layer=norm_layer)
        self.convblock5 = TransConvBlock(32, 32, "5", nums=4, norm_layer=norm_layer)
        self.conv6_1 = nn.ConvTranspose2d(32, 32, 3, 1, 1)
        self.conv6_2 = nn.ConvTranspose2d(32, 1, 3, 1, 1)
        self.conv6_1_bn = norm_layer(32)
        self.orig_size = [getattr(hparams, "cin_channels", None), getattr(hparams, "max_mel_lengths", None)]
        self.upsample_mode = "bilinear"
        self.skip_at = 3            # i == 3: concat with net[-4]

    def _head(self, net, upsample=None):
        out = fused_layer(net[-1], self.deconv1_1, self.deconv1_1_bn, ACT_RELU, next_conv=self.deconv1_2)
        nb = self._modules["convblock2"]
        return fused_layer(out, self.deconv1_2, self.deconv1_2_bn, ACT_RELU, upsample=upsample,
                           next_conv=None if self.skip_at == 1 else nb._modules["conv%s_0" % nb.name])

    def forward_nhwc(self, net, out_hw, head=None):
        # every F.interpolate of the reference's loop (New_Inpainting_Networks.py:76-83) follows the last layer of the block in front of it:
        # it is handed to that layer (resize inside its BatchNorm apply pass) instead of being a pass of its own
        sizes = [(net[-1 - i].shape[1], net[-1 - i].shape[2]) for i in range(1, len(net))] + [(int(out_hw[0]), int(out_hw[1]))]
        out = self._head(net, upsample=sizes[0]) if head is None else ops.bilinear_ac(head, sizes[0])
        for i in range(1, len(net)):
            skip = net[-(i + 1)] if i == self.skip_at else None     # virtual concat: two source pointers
            # the block's resized output feeds the first layer of the next block (pre-split where that layer stages pieces and has no second source)
            if i + 1 < len(net):
                nb = self._modules["convblock%d" % (i + 2)]
                nxt = None if i + 1 == self.skip_at else nb._modules["conv%s_0" % nb.name]
            else:
                nxt = self.conv6_1
            out = self._modules["convblock%d" % (i + 1)].forward_nhwc(out, skip, upsample=sizes[i], next_conv=nxt)
        return fused_pair(out, self.conv6_1, self.conv6_1_bn, ACT_RELU, self.conv6_2, ACT_SIGMOID)

    def forward(self, net, x_size):
        net = [to_nhwc(t) for t in net]
        return to_nchw_view(self.forward_nhwc(net, (int(x_size[2]), int(x_size[3]))))


class MelDecoderImage(MelDecoder):
    """AV decoder: the video feature (B,256,1,T/16) is concatenated to the bottleneck on C and
    `deconv1_1_1` (512->256) replaces `deconv1_1` (New_Inpainting_Networks.py:92-143).  `deconv1_1(+_bn)`
    stay in the state_dict (unused by forward, exactly as in the reference); `convblock1` does not exist."""

    UNUSED_PREFIXES = ("deconv1_1.", "deconv1_1_bn.")

    def __init__(self, hparams=hparams, norm_layer=None):
        super().__init__(hparams, norm_layer)
        nl = type(self.deconv1_1_bn)
        del self.convblock1
        self.deconv1_1_1 = nn.ConvTranspose2d(256 * 2, 256, 3, 1, (0, 1))
        self.deconv1_1_1_bn = nl(256)
        self._reorder()

    _ORDER = ("deconv1_1", "deconv1_1_bn", "deconv1_1_1", "deconv1_1_1_bn", "deconv1_2", "deconv1_2_bn", "convblock2",
              "convblock3", "convblock4", "convblock5", "conv6_1", "conv6_2", "conv6_1_bn")

    def _reorder(self):
        # keep the reference's registration order so state_dict() key order matches too
        _move_to_end(self._modules, self._ORDER)

    def _head_av(self, net, video_net):
        b, h, w = net[-1].shape[0], net[-1].shape[1], net[-1].shape[2]
        v = video_net.reshape(b, -1, h, w)                      # reference: video_net.view(B, -1, h, w)   (:119)
        v = to_nhwc(v)
        out = fused_layer(net[-1], self.deconv1_1_1, self.deconv1_1_1_bn, ACT_RELU, x2=v)   # virtual concat (:120-122)
        return fused_layer(out, self.deconv1_2, self.deconv1_2_bn, ACT_RELU)

    def forward(self, net, x_size, video_net=None):
        net = [to_nhwc(t) for t in net]
        head = self._head_av(net, video_net)
        return to_nchw_view(self.forward_nhwc(net, (int(x_size[2]), int(x_size[3])), head=head))

    def init_deconv_1_1_1(self):
        """duplicate the audio weight into both halves of the 512-ch layer (New_Inpainting_Networks.py:140-143)."""
        w = self.deconv1_1.weight.detach().unsqueeze(0).expand(2, 256, 256, 3, 3).contiguous()
        self.deconv1_1_1.weight.data.copy_(w.view(512, 256, 3, 3))


class MelDecoderImage2(MelDecoderImage):
    """as MelDecoderImage but the skip concat is with e1 at the LAST scale: convblock4 64->32,
    convblock5 (32*2)->32 x2 (New_Inpainting_Networks.py:146-197)."""

    def __init__(self, hparams=hparams, norm_layer=None):
        super().__init__(hparams, norm_layer)
        nl = type(self.deconv1_1_bn)
        self.convblock4 = TransConvBlock(64, 32, "4", nums=3, norm_layer=nl)
        self.convblock5 = TransConvBlock(32 * 2, 32, "5", nums=2, norm_layer=nl)
        self.skip_at = 4
        self._reorder()


class MelDecoder_old(MelDecoder):
    """audio-only decoder with the skip at the last scale (New_Inpainting_Networks.py:201-242)."""

    def __init__(self, hparams=hparams, norm_layer=None):
        super().__init__(hparams, norm_layer)
        nl = type(self.deconv1_1_bn)
        self.convblock4 = TransConvBlock(64, 32, "4", nums=3, norm_layer=nl)
        self.convblock5 = TransConvBlock(32 * 2, 32, "5", nums=2, norm_layer=nl)
        self.skip_at = 4
        _move_to_end(self._modules, ("convblock4", "convblock5", "conv6_1", "conv6_2", "conv6_1_bn"))


class MelDiscriminator(nn.Module):
    """PatchGAN D: Conv(1->ndf,(1,4),s(1,2),p(0,1)) BN LReLU; n_layers-1 x [Conv3x3 s2 BN LReLU];
    Conv3x3 s1 BN LReLU; Conv3x3 -> 1; Sigmoid (always on, as in Discriminator_Networks.py:13)."""

    def __init__(self, input_nc=1, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=True):
        super().__init__()
        _check_norm(norm_layer)
        self.n_layers = n_layers
        self.use_sigmoid = True
        use_bias = norm_layer is nn.InstanceNorm2d          # Discriminator_Networks.py:14
        self.conv1 = nn.Conv2d(input_nc, ndf, kernel_size=(1, 4), stride=(1, 2), padding=(0, 1), bias=use_bias)
        self.bn1 = norm_layer(ndf)
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            self.add_module("conv2_%d" % n, nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=(3, 3), stride=2, padding=1, bias=use_bias))
            self.add_module("norm_%d" % n, norm_layer(ndf * nf_mult))
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        self.conv3 = nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=3, stride=1, padding=1, bias=use_bias)
        self.norm3 = norm_layer(ndf * nf_mult)
        self.conv4 = nn.Conv2d(ndf * nf_mult, 1, kernel_size=3, stride=1, padding=1, bias=use_bias)

    def forward_nhwc(self, x):
        net = fused_layer(x, self.conv1, self.bn1, ACT_LRELU, next_conv=self._modules["conv2_1"] if self.n_layers > 1 else self.conv3)
        for n in range(1, self.n_layers):
            nxt = self._modules["conv2_%d" % (n + 1)] if n + 1 < self.n_layers else self.conv3
            net = fused_layer(net, self._modules["conv2_%d" % n], self._modules["norm_%d" % n], ACT_LRELU, next_conv=nxt)
        return fused_pair(net, self.conv3, self.norm3, ACT_LRELU, self.conv4, ACT_SIGMOID if self.use_sigmoid else ACT_NONE)

    def forward(self, input):
        return to_nchw_view(self.forward_nhwc(to_nhwc(input)))


class MultiScaleDiscriminator(nn.Module):
    """BASELINE cfg 4 "multi-scale D".  The reference has NO such class; following SURVEY.md §8d it is defined as
    k reference `MelDiscriminator`s applied to the 1x, 1/2x, 1/4x ... inputs obtained with
    avg_pool2d(3, stride 2, padding 1, count_include_pad=False) (pix2pixHD convention).  forward returns the list
    of the k PatchGAN maps; state_dict keys are `scale{i}.<MelDiscriminator keys>`."""

    def __init__(self, num_D=3, input_nc=1, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.num_D = num_D
        for i in range(num_D):
            self.add_module("scale%d" % i, MelDiscriminator(input_nc, ndf, n_layers, norm_layer))

    def forward_nhwc(self, x):
        outs = []
        for i in range(self.num_D):
            outs.append(self._modules["scale%d" % i].forward_nhwc(x))
            if i + 1 < self.num_D:
                x = ops.avgpool2d(x, 3, 2, 1)
        return outs

    def forward(self, input):
        return [to_nchw_view(o) for o in self.forward_nhwc(to_nhwc(input))]


class Inpainting_Dis(nn.Module):
    """joint mel x video sync discriminator (Discriminator_Networks.py:53-87): mel path 3x[Conv3x3 s2 BN LReLU]
    -> Conv(256,256,(10,1)); video path Conv1d(512,256,3,s2,p1) BN1d LReLU on fea_cat; cat on C ->
    Conv1d(512,1,k6) -> Sigmoid.  Valid only when the mel height is 80 (F/8 == 10), as in the reference."""

    def __init__(self):
        super().__init__()
        self.mel_conv1 = nn.Conv2d(1, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.mel_bn1 = nn.BatchNorm2d(64)
        self.mel_conv2 = nn.Conv2d(64, 128, 3, 2, 1, bias=False)
        self.mel_bn2 = nn.BatchNorm2d(128)
        self.mel_conv3 = nn.Conv2d(128, 256, 3, 2, 1, bias=False)
        self.mel_bn3 = nn.BatchNorm2d(256)
        self.mel_conv4 = nn.Conv2d(256, 256, (10, 1), 1, bias=False)
        self.vid_conv1 = nn.Conv1d(512, 256, 3, 2, 1, bias=False)
        self.vid_bn1 = nn.BatchNorm1d(256)
        self.conv = nn.Conv1d(512, 1, 6, bias=False)

    def forward(self, mel_inpainting, fea_inpainting):
        m = fused_layer(to_nhwc(mel_inpainting), self.mel_conv1, self.mel_bn1, ACT_LRELU)
        m = fused_layer(m, self.mel_conv2, self.mel_bn2, ACT_LRELU)
        m = fused_layer(m, self.mel_conv3, self.mel_bn3, ACT_LRELU)
        m = fused_layer(m, self.mel_conv4, None, ACT_NONE)              # (B, 1, W', 256)
        v = fea_inpainting.transpose(1, 2).unsqueeze(1)                 # (B,512,N) -> NHWC (B,1,N,512)
        v = conv1d_layer(v, self.vid_conv1, self.vid_bn1, ACT_LRELU)    # (B,1,N/2,256)
        net = torch.cat((m, v), dim=3)                                  # cat on C (tiny tensors)
        net = conv1d_layer(net, self.conv, None, ACT_SIGMOID)           # (B,1,W'-5,1)
        return net.reshape(net.shape[0], net.shape[2])


class DomainDis(nn.Module):
    """Conv1d(256,256,k13) -> ReLU -> Linear(256,256) -> Linear(256,1) -> Sigmoid on 13-step embeddings
    (Discriminator_Networks.py:90-107)."""

    def __init__(self, hparams=hparams):
        super().__init__()
        self.length_feature = getattr(hparams, "length_feature", 256)
        self.conv1 = nn.Conv1d(self.length_feature, 256, 13, 1, 0, bias=False)
        self.fc1 = nn.Linear(256, 256)
        self.fc2 = nn.Linear(256, 1)

    def forward(self, input):
        x = input.reshape(-1, self.length_feature, 13)                  # (B, C, 13)
        x = x.transpose(1, 2).unsqueeze(1)                              # NHWC (B,1,13,C)
        out = conv1d_layer(x, self.conv1, None, ACT_RELU)               # (B,1,1,256)
        out = linear_layer(out, self.fc1, ACT_NONE)
        out = linear_layer(out, self.fc2, ACT_SIGMOID)                  # (B,1,1,1)
        return out.reshape(-1, 1)


def conv1d_layer(x, conv, bn, act):
    """nn.Conv1d (+BatchNorm1d) (+act) on NHWC (B,1,L,C): a (1,k) convolution."""
    w = conv.weight.unsqueeze(2)                                         # (Cout,Cin,k) -> (Cout,Cin,1,k), a view
    return ops.conv_bn_act(x, w, conv.bias, bn, kernel=(1, conv.kernel_size[0]), stride=(1, conv.stride[0]),
                           padding=(0, conv.padding[0]), transposed=False, act=act,
                           training=(bn.training if bn is not None else True))


def linear_layer(x, fc, act):
    """nn.Linear on NHWC (B,1,1,C): a 1x1 convolution with the (out,in) weight viewed as (out,in,1,1)."""
    w = fc.weight.unsqueeze(2).unsqueeze(3)
    return ops.conv_bn_act(x, w, fc.bias, None, kernel=(1, 1), stride=(1, 1), padding=(0, 0), transposed=False, act=act)


# --------------------------------------------------------------------------- visual branch (ResNet-18)
class BasicBlock(nn.Module):
    """networks/ResNet.py:26-55: conv3x3(s) BN ReLU conv3x3 BN (+ downsample(x)) -> add -> ReLU."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward_nhwc(self, x, next_conv=None):
        """`next_conv`: the next block's conv1 (fused_layer: the join then writes the planes its kernels stage beside the fp32 tensor)"""
        x, xr = ops.fork2(x)                          # two readers: their gradients reach the producer as two addends (no pass for the sum)
        out = fused_layer(x, self.conv1, self.bn1, ACT_RELU, next_conv=self.conv2)
        res = xr if self.downsample is None else fused_layer(xr, self.downsample[0], self.downsample[1], ACT_NONE)
        return fused_layer(out, self.conv2, self.bn2, ACT_RELU, residual=res, next_conv=next_conv)   # relu(bn2(conv2(out)) + res), networks/ResNet.py:46-53

    def forward(self, x):
        return to_nchw_view(self.forward_nhwc(to_nhwc(x)))


class ResNet(nn.Module):
    """the reference's local ResNet (networks/Image_Embedding.py:13-71): conv7x7 s2 -> BN -> ReLU -> maxpool 3x3 s2
    -> layer1..4 -> AvgPool2d(7) -> fc(512 -> length_feature).  224x224 inputs only (as the reference)."""

    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), channel_size=3, length_feature=256):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(channel_size, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.fc = nn.Linear(512 * block.expansion, length_feature)
        import math
        for m in self.modules():                      # Image_Embedding.py:30-36
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        """x: (N, C, 224, 224) NCHW frames -> (N, length_feature)."""
        h = ops.frames_to_nhwc4(x)
        blocks = [blk for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for blk in layer]
        h = fused_layer(h, self.conv1, self.bn1, ACT_RELU, pool=(3, 2, 1), next_conv=blocks[0].conv1)   # conv1 -> bn1 -> relu -> maxpool, Image_Embedding.py:20-23
        for i, blk in enumerate(blocks):
            h = blk.forward_nhwc(h, next_conv=blocks[i + 1].conv1 if i + 1 < len(blocks) else None)
        if h.shape[1] != 7 or h.shape[2] != 7:
            raise RuntimeError("the reference ResNet (AvgPool2d(7) + fc) needs 224x224 frames")
        h = ops.avgpool_hw(h)
        return linear_layer(h, self.fc, ACT_NONE).reshape(h.shape[0], -1)


def ImageResnet18(hparams=hparams):
    return ResNet(BasicBlock, [2, 2, 2, 2], channel_size=3, length_feature=getattr(hparams, "length_feature", 256))


def FlowResnet18(hparams=hparams):
    return ResNet(BasicBlock, [2, 2, 2, 2], channel_size=2, length_feature=getattr(hparams, "length_feature", 256))


class _ImageEmbeddingBase(nn.Module):
    def _temporal(self, fea_bcn, use_dead_bn):
        """fea (B, C, N) -> conv_1 (s2) -> [dead bn_1/relu: running stats only] -> conv_2 (s2) -> (B, C', N/4)."""
        x = fea_bcn.transpose(1, 2).unsqueeze(1)                       # NHWC (B,1,N,C)
        out = conv1d_layer(x, self.conv_1, None, ACT_NONE)
        if use_dead_bn and self.bn_1.training:
            with torch.no_grad():                                       # `self.relu(self.bn_1(out))` result is discarded in the
                conv1d_layer(x, self.conv_1, self.bn_1, ACT_RELU)       # reference (Image_Embedding.py:123): only the buffers move
        out = conv1d_layer(out, self.conv_2, None, ACT_NONE)
        return out.squeeze(1).transpose(1, 2)                           # (B, C', N/4)


class ImageEmbedding2(_ImageEmbeddingBase):
    """E_v (networks/Image_Embedding.py:174-200): RGB ResNet-18 + flow ResNet-18 per frame -> cat -> Conv1d(512,512,3,2,1)
    -> Conv1d(512,256,3,2,1); returns (out (B,256,1,N/4), fea_cat (B,512,N)).  bn_1 / bn_2 exist but are unused."""

    UNUSED_PREFIXES = ("bn_1.", "bn_2.")

    def __init__(self, hparams=hparams):
        super().__init__()
        self.hparams = hparams
        lf = getattr(hparams, "length_feature", 256)
        self.image_single_model = ImageResnet18(hparams)
        self.flow_single_model = FlowResnet18(hparams)
        self.conv_1 = nn.Conv1d(2 * lf, 2 * lf, 3, 2, 1, bias=False)
        self.bn_1 = nn.BatchNorm1d(2 * lf)
        self.conv_2 = nn.Conv1d(2 * lf, lf, 3, 2, 1, bias=False)
        self.bn_2 = nn.BatchNorm1d(lf)
        self.dead_bn = False

    def _features(self, video_block, flow_block):
        sz = getattr(self.hparams, "image_size", 224)
        lf = getattr(self.hparams, "length_feature", 256)
        b = video_block.size(0)
        side = self._flow_stream(flow_block)
        if side is None:
            img = self.image_single_model(video_block.reshape(-1, 3, sz, sz)).reshape(b, -1, lf)
            flw = self.flow_single_model(flow_block.reshape(-1, 2, sz, sz)).reshape(b, -1, lf)
        else:
            # the two ResNets share nothing: the flow network runs on its own stream next to the RGB network (autograd replays each
            # backward node on its forward stream, so the two backward chains overlap the same way).  What either chain leaves idle --
            # kernel tails, the BatchNorm passes' memory latency -- the other fills.
            cur = torch.cuda.current_stream(flow_block.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                flw = self.flow_single_model(flow_block.reshape(-1, 2, sz, sz)).reshape(b, -1, lf)
            flow_block.record_stream(side)
            img = self.image_single_model(video_block.reshape(-1, 3, sz, sz)).reshape(b, -1, lf)
            cur.wait_stream(side)
            flw.record_stream(cur)
        return torch.cat((img, flw), 2).transpose(2, 1)                # (B, 512, N)

    def _flow_stream(self, t):
        if not (FLOW_STREAM and t.is_cuda) or torch.cuda.is_current_stream_capturing():
            return None
        st = _FLOW_STREAMS.get(t.device.index)          # one per device for the process (ops.SIDE_STREAMS must not grow with every model)
        if st is None:
            st = _FLOW_STREAMS[t.device.index] = torch.cuda.Stream(device=t.device)
            ops.SIDE_STREAMS.append(st)
        return st

    def forward(self, video_block, flow_block):
        fea_cat = self._features(video_block, flow_block)
        out = self._temporal(fea_cat, self.dead_bn).unsqueeze(2)
        return out, fea_cat


class ImageEmbedding(ImageEmbedding2):
    """networks/Image_Embedding.py:100-126: as ImageEmbedding2 but bn_1 runs (result discarded) and only `out` is returned."""

    def __init__(self, hparams=hparams):
        super().__init__(hparams)
        self.dead_bn = True

    def forward(self, video_block, flow_block):
        return self._temporal(self._features(video_block, flow_block), True).unsqueeze(2)


class ImageEmbedding_single(_ImageEmbeddingBase):
    """networks/Image_Embedding.py:129-150."""

    def __init__(self, hparams=hparams, image=1):
        super().__init__()
        self.image, self.hparams = image, hparams
        lf = getattr(hparams, "length_feature", 256)
        self.image_single_model = ImageResnet18(hparams) if image else FlowResnet18(hparams)
        self.conv_1 = nn.Conv1d(lf, lf, 3, 2, 1, bias=False)
        self.bn_1 = nn.BatchNorm1d(lf)
        self.conv_2 = nn.Conv1d(lf, lf, 3, 2, 1, bias=False)
        self.bn_2 = nn.BatchNorm1d(lf)

    def forward(self, video_block):
        sz = getattr(self.hparams, "image_size", 224)
        lf = getattr(self.hparams, "length_feature", 256)
        f = self.image_single_model(video_block.reshape(-1, 3 if self.image else 2, sz, sz)).reshape(video_block.size(0), -1, lf)
        return self._temporal(f.transpose(1, 2), True)


class ImageEmbedding_finetune(_ImageEmbeddingBase):
    """networks/Image_Embedding.py:152-171: temporal convs on precomputed per-frame features (B, N, C)."""

    def __init__(self, hparams=hparams, image=1):
        super().__init__()
        self.image, self.hparams = image, hparams
        lf = getattr(hparams, "length_feature", 256)
        self.conv_1 = nn.Conv1d(lf, lf, 3, 2, 1, bias=False)
        self.bn_1 = nn.BatchNorm1d(lf)
        self.conv_2 = nn.Conv1d(lf, lf, 3, 2, 1, bias=False)
        self.bn_2 = nn.BatchNorm1d(lf)

    def forward(self, image_out):
        return self._temporal(image_out.transpose(1, 2), True).unsqueeze(2)
