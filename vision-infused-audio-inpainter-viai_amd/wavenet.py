"""WaveNet vocoder on the HIP kernels: the reference's `wavenet_vocoder.WaveNet` API and state_dict.

Reference: wavenet_vocoder/wavenet.py:62-393 (WaveNet), modules.py:30-216 (weight-normed Conv1d / Conv1d1x1 /
ConvTranspose2d factories, ResidualConv1dGLU), conv.py:7-65 (incremental Conv1d), mixture.py:25-153 (MoL).
Tensors are (B, 1, T, C) NHWC internally; the module API takes / returns the reference's (B, C, T).
Known latent bugs of the reference that are NOT reproduced as features: `self.softmax(x, dim=1)` TypeError
(wavenet.py:233) and the 5x `self.conv.clear_buffer()` (modules.py:213-216).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import ACT_NONE, ACT_RELU, _c, _ptr, _require, _stream


# ----------------------------------------------------------------------------- autograd ops
class _WeightNorm(torch.autograd.Function):
    """w = g * v / ||v||  per output row (torch.nn.utils.weight_norm, dim=0)."""

    @staticmethod
    def forward(ctx, v, g):
        lib = _lib.load()
        _require(v, g)
        v, g = _c(v), _c(g)
        rows = v.shape[0]
        L = v.numel() // rows
        w = torch.empty_like(v)
        norm = torch.empty(rows, device=v.device, dtype=torch.float32)
        _lib.check(lib.viai_weight_norm_fwd(v.data_ptr(), g.data_ptr(), w.data_ptr(), norm.data_ptr(), rows, L, _stream()), "viai_weight_norm_fwd")
        ctx.save_for_backward(v, g, norm)
        return w

    @staticmethod
    def backward(ctx, dw):
        lib = _lib.load()
        v, g, norm = ctx.saved_tensors
        dw = _c(dw)
        rows = v.shape[0]
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        _lib.check(lib.viai_weight_norm_bwd(dw.data_ptr(), v.data_ptr(), g.data_ptr(), norm.data_ptr(), dv.data_ptr(), dg.data_ptr(),
                                            rows, v.numel() // rows, 0, _stream()), "viai_weight_norm_bwd")
        return dv, dg


def normed_weight(m):
    """effective weight of a (possibly weight-normed) holder module."""
    if hasattr(m, "weight_g"):
        return _WeightNorm.apply(m.weight_v, m.weight_g)
    return m.weight


class _GLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, yc):
        lib = _lib.load()
        _require(y, yc)
        y = _c(y)
        yc = _c(yc) if yc is not None else None
        H = y.shape[-1] // 2
        rows = y.numel() // (2 * H)
        z = torch.empty(y.shape[:-1] + (H,), device=y.device, dtype=torch.float32)
        _lib.check(lib.viai_glu_fwd(y.data_ptr(), _ptr(yc), z.data_ptr(), rows, H, _stream()), "viai_glu_fwd")
        ctx.save_for_backward(y, yc)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        y, yc = ctx.saved_tensors
        dz = _c(dz)
        H = y.shape[-1] // 2
        dy = torch.empty_like(y)
        _lib.check(lib.viai_glu_bwd(dz.data_ptr(), y.data_ptr(), _ptr(yc), dy.data_ptr(), y.numel() // (2 * H), H, _stream()), "viai_glu_bwd")
        return dy, (dy if yc is not None else None)


class _AddScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, s):
        lib = _lib.load()
        _require(a, b)
        a = _c(a)
        b = _c(b) if b is not None else None
        out = torch.empty_like(a)
        _lib.check(lib.viai_add_scale(a.data_ptr(), _ptr(b), out.data_ptr(), s, a.numel(), _stream()), "viai_add_scale")
        ctx.s, ctx.hb = s, b is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        g = _c(g)
        d = torch.empty_like(g)
        _lib.check(lib.viai_add_scale(g.data_ptr(), 0, d.data_ptr(), ctx.s, g.numel(), _stream()), "viai_add_scale")
        return d, (d if ctx.hb else None), None


def add_scale(a, b, s):
    return _AddScale.apply(a, b, float(s))


class _Relu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        lib = _lib.load()
        _require(a)
        a = _c(a)
        out = torch.empty_like(a)
        _lib.check(lib.viai_relu_fwd(a.data_ptr(), out.data_ptr(), a.numel(), _stream()), "viai_relu_fwd")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (out,) = ctx.saved_tensors
        g = _c(g)
        d = torch.empty_like(g)
        _lib.check(lib.viai_relu_bwd(g.data_ptr(), out.data_ptr(), d.data_ptr(), g.numel(), _stream()), "viai_relu_bwd")
        return d


class _Outer(torch.autograd.Function):
    """Conv1d1x1(1, C) on a scalar signal: y[p][c] = x[p]*w[c] + b[c]."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.load()
        _require(x, w, b)
        x, w, b = _c(x), _c(w), _c(b)
        Cc = w.numel()
        rows = x.numel()
        y = torch.empty(x.shape[:3] + (Cc,), device=x.device, dtype=torch.float32)
        _lib.check(lib.viai_outer_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, Cc, _stream()), "viai_outer_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = _c(dy)
        Cc, rows = w.numel(), x.numel()
        part = torch.empty(2 * Cc * lib.viai_outer_bwd_blocks(rows), device=dy.device, dtype=torch.float32)
        dw, db = torch.empty_like(w), torch.empty(Cc, device=dy.device, dtype=torch.float32)
        _lib.check(lib.viai_outer_bwd(dy.data_ptr(), x.data_ptr(), part.data_ptr(), dw.data_ptr(), db.data_ptr(), rows, Cc, 0, _stream()), "viai_outer_bwd")
        return None, dw, db


class _Upsample(torch.autograd.Function):
    """ConvTranspose2d(1,1,(KH,S),stride (1,S),padding ((KH-1)/2,0)) + ReLU on (B,F,T) (wavenet.py:153-164)."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.load()
        _require(x, w, b)
        x, w, b = _c(x), _c(w), _c(b)
        B, Fq, T = x.shape
        KH, S = w.shape[2], w.shape[3]
        y = torch.empty((B, Fq, T * S), device=x.device, dtype=torch.float32)
        _lib.check(lib.viai_upsample_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, Fq, T, KH, S, _stream()), "viai_upsample_fwd")
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        B, Fq, T = x.shape
        KH, S = w.shape[2], w.shape[3]
        part = torch.empty((KH * 16 + 1) * lib.viai_upsample_bwd_blocks(), device=dy.device, dtype=torch.float32)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, db = torch.empty_like(w), torch.empty(1, device=dy.device, dtype=torch.float32)
        _lib.check(lib.viai_upsample_bwd(dy.data_ptr(), y.data_ptr(), x.data_ptr(), w.data_ptr(), part.data_ptr(), _ptr(dx), dw.data_ptr(),
                                         db.data_ptr(), B, Fq, T, KH, S, 0, _stream()), "viai_upsample_bwd")
        return dx, dw, db


class _MoLLoss(torch.autograd.Function):
    """DiscretizedMixturelogisticLoss: masked mean of the MoL negative log-likelihood (loss_functions.py:43-62)."""

    @staticmethod
    def forward(ctx, yhat, y, mask, num_classes, log_scale_min):
        lib = _lib.load()
        _require(yhat, y, mask)
        yhat, y = _c(yhat), _c(y)
        mask = _c(mask) if mask is not None else None
        pitch = yhat.shape[-1]
        rows = yhat.numel() // pitch
        dev = yhat.device
        loss_rows = torch.empty(rows, device=dev, dtype=torch.float32)
        wrow = torch.empty(rows, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        dyh = torch.empty_like(yhat) if yhat.requires_grad else None
        _lib.check(lib.viai_mol_loss(yhat.data_ptr(), y.data_ptr(), _ptr(mask), loss_rows.data_ptr(), wrow.data_ptr(), loss.data_ptr(),
                                     _ptr(dyh), rows, pitch, 10, float(num_classes), float(log_scale_min), _stream()), "viai_mol_loss")
        ctx.dyh = dyh
        ctx.mark_non_differentiable(loss_rows)
        return loss, loss_rows

    @staticmethod
    def backward(ctx, g, _g2):
        lib = _lib.load()
        d = ctx.dyh
        g = _c(g)
        _lib.check(lib.viai_scale_by_scalar(d.data_ptr(), g.data_ptr(), d.numel(), _stream()), "viai_scale_by_scalar")
        return d, None, None, None, None


def mol_loss(yhat_nhwc, y, mask=None, num_classes=65536, log_scale_min=math.log(1e-14)):
    """yhat_nhwc: (B,1,T,P>=30) rows [logit|mean|log_scale]x10; y: (B,T[,1]) targets in [-1,1]; mask (B,T[,1])."""
    loss, _ = _MoLLoss.apply(yhat_nhwc, y.reshape(-1), None if mask is None else mask.reshape(-1), num_classes, log_scale_min)
    return loss


def mol_sample(yhat_nhwc, u1, u2, log_scale_min=-7.0):
    """sample_from_discretized_mix_logistic with injected uniforms u1 (rows,10), u2 (rows,)."""
    lib = _lib.load()
    yh = _c(yhat_nhwc)
    pitch = yh.shape[-1]
    rows = yh.numel() // pitch
    out = torch.empty(rows, device=yh.device, dtype=torch.float32)
    _lib.check(lib.viai_mol_sample(yh.data_ptr(), _c(u1).data_ptr(), _c(u2).data_ptr(), out.data_ptr(), rows, pitch, 10,
                                   float(log_scale_min), _stream()), "viai_mol_sample")
    return out


def receptive_field_size(total_layers, num_cycles, kernel_size, dilation=lambda x: 2 ** x):
    """samples of context one output sees: (k - 1) * sum of the layer dilations + 1 (wavenet_vocoder/wavenet.py:41-59; 505 for the
    reference's 24 layers / 4 cycles / k = 3).  `dilation(i)` maps the position inside a cycle to the dilation."""
    if total_layers % num_cycles != 0:
        raise AssertionError("total_layers must be a multiple of num_cycles")
    per = total_layers // num_cycles
    return (kernel_size - 1) * sum(dilation(i % per) for i in range(total_layers)) + 1


def sequence_mask(sequence_length, max_len=None):
    """(B,) lengths -> (B, max_len) float mask, 1 where t < length (loss_functions.py:11-21); stays on the lengths' device."""
    if max_len is None:
        max_len = int(sequence_length.max())
    t = torch.arange(0, max_len, device=sequence_length.device, dtype=torch.long)
    return (t.unsqueeze(0) < sequence_length.long().unsqueeze(1)).float()


def to_one_hot(tensor, n, fill_with=1.):
    """integer tensor (...) -> float (..., n), one-hot along a new last axis (wavenet_vocoder/mixture.py:108-114)."""
    out = torch.zeros(tuple(tensor.shape) + (n,), dtype=torch.float32, device=tensor.device)
    return out.scatter_(tensor.dim(), tensor.long().unsqueeze(-1), fill_with)


# ----------------------------------------------------------------------------- modules
def _wn(m, on):
    return nn.utils.weight_norm(m) if on else m


def Conv1d(in_channels, out_channels, kernel_size=1, padding=0, dilation=1, bias=True, weight_normalization=True,
           dropout=0, std_mul=1.0):
    """modules.py:30-41 (parameter holder; the arithmetic is viai_conv2d_* with kh = 1)."""
    m = nn.Conv1d(in_channels, out_channels, kernel_size, padding=padding, dilation=dilation, bias=bias)
    if weight_normalization:
        std = math.sqrt((std_mul * (1.0 - dropout)) / (m.kernel_size[0] * in_channels))
        m.weight.data.normal_(mean=0, std=std)
        m.bias.data.zero_()
    return _wn(m, weight_normalization)


def Conv1d1x1(in_channels, out_channels, bias=True, weight_normalization=True):
    return Conv1d(in_channels, out_channels, 1, 0, 1, bias, weight_normalization)


def conv1d_apply(x, m, act=ACT_NONE, causal_crop=False):
    """x (B,1,T,Cin) -> (B,1,T',Cout) with holder m (nn.Conv1d, maybe weight-normed)."""
    w = normed_weight(m).unsqueeze(2)                         # (Cout,Cin,1,k)
    k, d, p = m.kernel_size[0], m.dilation[0], m.padding[0]
    return ops.conv_bn_act(x, w, m.bias, None, kernel=(1, k), stride=(1, 1), padding=(0, p), dilation=(1, d),
                           padding2=(-1, 0 if causal_crop else -1), act=act)


def _pipe_images(w_stage, b_stage, w_c, w_out, b_out, w_skip, b_skip, w_l1, b_l1, w_l2, b_l2, dev):
    """Weight images of the pipelined synthesis kernel (csrc/wavenet_pipe.hip; layouts in include/viai_hip.h, viai_wn_pipe_image_floats).
    Inputs: per layer the fused gate rows `w_stage[l]` [G][3C (+H for l > 0)] / `b_stage[l]` [G] of the chain form, the conditioning rows `w_c[l]`
    [G][cin], and the out / skip 1x1s; the head's two 1x1s.  Compute unit j of layer l owns gate pairs h in [26 j, 26 j + 26), residual rows
    [52 j, 52 j + 52) and skip rows [26 j, 26 j + 26) -- row SLOTS beyond a range are zero rows.  Pure re-arrangement: no arithmetic on the weights."""
    NL, NCU, NW, GW, BW, C, H, S = 24, 10, 8, 7, 10, 512, 256, 256
    G = 2 * H
    j = torch.arange(NCU).view(NCU, 1)
    r = torch.arange(NW * GW).view(1, -1)
    hh = 26 * j + torch.where(r < 26, r, r - 26)
    grow = torch.where((r < 52) & (hh < H), hh + torch.where(r < 26, 0, H), torch.full_like(hh, G)).to(dev)          # [10][56] -> gate row, G = the zero row
    q = torch.arange(NW * BW).view(1, -1)
    xrow = 52 * j + q
    srow = 26 * j + (q - 52)
    # B slots: < 52 residual row, 52 .. 77 skip row (offset C in the stacked [Wo; Ws] matrix), C + S = the zero row
    brow = torch.where((q < 52) & (xrow < C), xrow, torch.where((q >= 52) & (q < 78) & (srow < S), C + srow, torch.full_like(xrow, C + S))).to(dev)
    wreg, wcond, wlds, bias = [], [], [], []
    z1 = lambda n: torch.zeros(1, n, device=dev)
    for l in range(NL):
        ws = w_stage[l]
        if ws.size(1) == 3 * C:
            ws = torch.cat((ws, torch.zeros(G, H, device=dev)), 1)                      # layer 0: no z columns
        W = torch.cat((ws, z1(3 * C + H)), 0)[grow]                                     # [10][56][1792]
        # register image: wave = 128 past-tap / 64 current-tap columns of all 52 rows; lane (g, cg) = (lane / 16, lane % 16) holds rows 13 g + i (i < 13):
        # register 8 i + 4 m + e = past-tap column 128 wave + 64 m + 4 cg + e, register 104 + 4 i + e = current-tap column 64 wave + 4 cg + e
        pre = W[:, :52, :2 * C].reshape(NCU, 4, 13, NW, 2, 16, 4).permute(0, 3, 2, 4, 6, 1, 5).reshape(NCU, NW, 104, 64)
        cur = W[:, :52, 2 * C:3 * C].reshape(NCU, 4, 13, NW, 16, 4).permute(0, 3, 2, 5, 1, 4).reshape(NCU, NW, 52, 64)
        wreg.append(torch.cat((pre, cur), 2))                                           # 104 + 52 registers
        wc = torch.cat((w_c[l], z1(w_c[l].size(1))), 0)[grow]                           # [10][56][80]
        wcond.append(torch.cat((wc, torch.zeros(NCU, 8, wc.size(2), device=dev)), 1))   # 64 row slots
        bg = torch.cat((b_stage[l], torch.zeros(1, device=dev)))[grow]                  # [10][56]
        if l == 0:
            wB, bB = torch.zeros(NCU, NW * BW, H, device=dev), torch.zeros(NCU, NW * BW, device=dev)
        else:
            wB = torch.cat((w_out[l - 1], w_skip[l - 1], z1(H)), 0)[brow]               # [10][80][256]
            bB = torch.cat((b_out[l - 1], b_skip[l - 1], torch.zeros(1, device=dev)))[brow]
        rows = torch.cat((W[:, :52, 3 * C:], wB[:, :78]), 1)                            # LDS rows: 52 gate rows (z columns), 78 out / skip rows
        wlds.append(torch.nn.functional.pad(rows, (0, 4)))                              # rows padded to 260 floats (lane = row reads without bank conflicts)
        bias.append(torch.cat((bg, bB), 1))
    head_w = torch.cat((w_skip[NL - 1], w_l1, w_l2, torch.zeros(32 - w_l2.size(0), S, device=dev)), 0)
    head_b = torch.cat((b_skip[NL - 1], b_l1, b_l2, torch.zeros(32 - b_l2.numel(), device=dev)))
    c = lambda ts: torch.stack(ts).float().contiguous()
    return c(wreg), c(wcond), c(wlds), c(bias), head_w.float().contiguous(), head_b.float().contiguous()



class ResidualConv1dGLU(nn.Module):
    """modules.py:84-216."""

    def __init__(self, residual_channels, gate_channels, kernel_size, skip_out_channels=None, cin_channels=-1, gin_channels=-1,
                 dropout=1 - 0.95, padding=None, dilation=1, causal=True, bias=True, weight_normalization=True):
        super().__init__()
        self.dropout = dropout
        skip_out_channels = residual_channels if skip_out_channels is None else skip_out_channels
        if padding is None:
            padding = (kernel_size - 1) * dilation if causal else (kernel_size - 1) // 2 * dilation
        self.causal = causal
        self.conv = Conv1d(residual_channels, gate_channels, kernel_size, padding=padding, dilation=dilation, bias=bias,
                           weight_normalization=weight_normalization)
        self.conv1x1c = Conv1d1x1(cin_channels, gate_channels, bias, weight_normalization) if cin_channels > 0 else None
        self.conv1x1g = Conv1d1x1(gin_channels, gate_channels, bias, weight_normalization) if gin_channels > 0 else None
        self.conv1x1_out = Conv1d1x1(gate_channels // 2, residual_channels, bias, weight_normalization)
        self.conv1x1_skip = Conv1d1x1(gate_channels // 2, skip_out_channels, bias, weight_normalization)

    def forward_nhwc(self, x, c=None, g=None):
        residual = x
        if self.training and self.dropout > 0:
            x = torch.nn.functional.dropout(x, p=self.dropout, training=True)   # RNG-dependent: parity tests use dropout = 0
        y = conv1d_apply(x, self.conv, causal_crop=self.causal)                 # (B,1,T,gate)  modules.py:176-181
        yc = None
        if c is not None:
            yc = conv1d_apply(c, self.conv1x1c)                                 # :187-191
        if g is not None:
            yg = conv1d_apply(g, self.conv1x1g)                                 # :194-198
            yc = yg if yc is None else yc + yg
        z = _GLU.apply(y, yc)                                                   # :201
        s = conv1d_apply(z, self.conv1x1_skip)                                  # :204
        out = conv1d_apply(z, self.conv1x1_out)                                 # :207
        return add_scale(out, residual, math.sqrt(0.5)), s                      # :209


class WaveNet(nn.Module):
    """wavenet.py:62-393 (scalar_input=True / mixture-of-logistics output is the reference's configuration)."""

    def __init__(self, out_channels=30, layers=24, stacks=4, residual_channels=512, gate_channels=512, skip_out_channels=256,
                 kernel_size=3, dropout=1 - 0.95, cin_channels=80, gin_channels=-1, n_speakers=None, weight_normalization=True,
                 upsample_conditional_features=True, upsample_scales=(4, 4, 4, 4), freq_axis_kernel_size=3, scalar_input=True,
                 use_speaker_embedding=True):
        super().__init__()
        assert layers % stacks == 0
        self.scalar_input, self.out_channels, self.cin_channels = scalar_input, out_channels, cin_channels
        per = layers // stacks
        self.first_conv = Conv1d1x1(1 if scalar_input else out_channels, residual_channels, True, weight_normalization)
        self.conv_layers = nn.ModuleList([
            ResidualConv1dGLU(residual_channels, gate_channels, kernel_size, skip_out_channels, cin_channels, gin_channels, dropout,
                              dilation=2 ** (i % per), bias=True, weight_normalization=weight_normalization) for i in range(layers)])
        self.last_conv_layers = nn.ModuleList([nn.ReLU(inplace=True), Conv1d1x1(skip_out_channels, skip_out_channels, True, weight_normalization),
                                               nn.ReLU(inplace=True), Conv1d1x1(skip_out_channels, out_channels, True, weight_normalization)])
        self.embed_speakers = None
        if gin_channels > 0 and use_speaker_embedding:
            self.embed_speakers = nn.Embedding(n_speakers, gin_channels)
            self.embed_speakers.weight.data.normal_(0, 0.1)
        self.upsample_conv = None
        if upsample_conditional_features:
            self.upsample_conv = nn.ModuleList()
            for s in upsample_scales:
                m = nn.ConvTranspose2d(1, 1, (freq_axis_kernel_size, s), padding=((freq_axis_kernel_size - 1) // 2, 0), dilation=1, stride=(1, s))
                m.weight.data.fill_(1.0 / freq_axis_kernel_size)
                m.bias.data.zero_()
                self.upsample_conv.append(_wn(m, weight_normalization))
                self.upsample_conv.append(nn.ReLU(inplace=True))
        self.receptive_field = (kernel_size - 1) * sum(2 ** (i % per) for i in range(layers)) + 1

    def has_speaker_embedding(self):
        return self.embed_speakers is not None

    def local_conditioning_enabled(self):
        return self.cin_channels > 0

    def _upsample(self, c):
        """c (B, cin, T') -> (B, cin, T) through the weight-normed transposed-conv stack (wavenet.py:208-215)."""
        if c is None or self.upsample_conv is None:
            return c
        for m in self.upsample_conv:
            if isinstance(m, nn.ReLU):
                continue                                             # fused into the kernel
            c = _Upsample.apply(c, normed_weight(m), m.bias)
        return c

    def _global(self, g, B, T):
        if g is None:
            return None
        if self.embed_speakers is not None:
            g = self.embed_speakers(g.view(B, -1)).transpose(1, 2)
        g = g.unsqueeze(-1) if g.dim() == 2 else g
        return g.expand(B, -1, T).transpose(1, 2).unsqueeze(1).contiguous()          # (B,1,T,gin)

    def forward_nhwc(self, x, c=None, g=None):
        """x (B,1,T) or (B,out,T); returns (B,1,T,P) with P = out_channels padded to a multiple of 4."""
        B, _, T = x.size()
        g_n = self._global(g, B, T)
        c = self._upsample(c)
        c_n = None
        if c is not None:
            assert c.size(-1) == T
            c_n = c.transpose(1, 2).unsqueeze(1).contiguous()                         # (B,1,T,cin)
        if self.scalar_input:
            h = _Outer.apply(x.reshape(B, 1, T).contiguous(), normed_weight(self.first_conv).reshape(-1), self.first_conv.bias)
        else:
            h = conv1d_apply(x.transpose(1, 2).unsqueeze(1).contiguous(), self.first_conv)
        skips = None
        for f in self.conv_layers:
            h, s = f.forward_nhwc(h, c_n, g_n)
            skips = s if skips is None else add_scale(skips, s, math.sqrt(0.5))       # wavenet.py:222-226
        h = _Relu.apply(skips)
        h = conv1d_apply(h, self.last_conv_layers[1], act=ACT_RELU)
        last = self.last_conv_layers[3]
        w = normed_weight(last)
        pad = (-self.out_channels) % 4
        bias = last.bias
        if pad:                                                                       # 30 -> 32 output rows (16-byte rows)
            w = torch.cat((w, w.new_zeros((pad,) + tuple(w.shape[1:]))), 0)
            bias = torch.cat((bias, bias.new_zeros(pad)), 0)
        return ops.conv_bn_act(h, w.unsqueeze(2), bias, None, kernel=(1, 1), stride=(1, 1), padding=(0, 0))

    def forward(self, x, c=None, g=None, softmax=False):
        y = self.forward_nhwc(x, c, g)                                                # (B,1,T,P)
        y = y[..., :self.out_channels].squeeze(1).transpose(1, 2)                     # (B, out, T) view
        return torch.softmax(y, dim=1) if softmax else y

    def clear_buffer(self):
        pass                                      # incremental state (ring buffers) is local to incremental_forward

    @torch.no_grad()
    def incremental_forward(self, initial_input=None, c=None, g=None, T=100, test_inputs=None, tqdm=lambda x: x, softmax=True,
                            quantize=True, log_scale_min=-7.0, uniforms=None, use_graph=False, return_logits=False, timing=None):
        """Sample-by-sample synthesis (wavenet.py:237-364) for the scalar-input / MoL configuration.

        A time step = first conv, 2 GEMV-batch kernels per layer, head + MoL sample.  Default: `viai_wavenet_synth_run` loops over the
        steps in C with the time index passed by value; `use_graph=True`: `viai_wavenet_synth_step` (time index on the device, the first
        conv advances it) captured once into a HIP graph and replayed.
        `uniforms=(u1 (B,T,10), u2 (B,T))` injects the sampler's two uniform draws (parity tests); default torch.rand.
        `timing={"warmup": W}` (bench.py): the first W time steps run untimed, the remaining T - W are bracketed by device
        synchronisations and reported as timing["ms"] / timing["steps"] (set-up -- weight norm, linearised weights -- excluded)."""
        import ctypes as Ct
        lib = _lib.load()
        if not self.scalar_input:
            raise NotImplementedError("incremental_forward: scalar-input (mixture of logistics) WaveNet only")
        dev = self.first_conv.bias.device
        if test_inputs is not None:
            if test_inputs.size(1) == 1:
                test_inputs = test_inputs.transpose(1, 2)                             # -> (B, n, 1)
            B = test_inputs.size(0)
            T = test_inputs.size(1) if T is None else max(int(T), test_inputs.size(1))
            tin = test_inputs.reshape(B, -1).to(dev).float().contiguous()
        else:
            B = c.size(0) if c is not None else 1
            tin = None
        T = int(T)
        if not 1 <= B <= 32:
            raise NotImplementedError("incremental_forward: 1 to 32 streams")
        cond = None
        if c is not None:
            cu = self._upsample(c.to(dev).float())
            assert cu.size(-1) == T
            cond = cu.transpose(1, 2).contiguous()                                    # (B, T, cin)
        if uniforms is None:
            u1 = torch.empty(B, T, self.out_channels // 3, device=dev).uniform_(1e-5, 1.0 - 1e-5)
            u2 = torch.empty(B, T, device=dev).uniform_(1e-5, 1.0 - 1e-5)
        else:
            u1, u2 = uniforms[0].to(dev).float().contiguous(), uniforms[1].to(dev).float().contiguous()
        g_vec = None
        if g is not None:                                                            # wavenet.py:284-290: time-invariant
            g = g.to(dev)
            g_vec = (self.embed_speakers(g.view(B, -1)).transpose(1, 2) if self.embed_speakers is not None else g.float().view(B, -1, 1)).contiguous()
        Cc = self.first_conv.bias.numel()
        f0 = self.conv_layers[0]
        G, S = f0.conv.bias.numel(), f0.conv1x1_skip.bias.numel()
        keep = []                                                                     # keep tensors alive

        def t(x):
            x = x.detach().float().contiguous()
            keep.append(x)
            return x.data_ptr()
        layers = (_lib.WnLayer * len(self.conv_layers))()
        held = {k: [] for k in ("w_stage", "b_stage", "w_c", "w_out", "b_out", "w_skip", "b_skip")}         # the tensors behind the pointers (pipelined form)

        def th(key, x):
            x = x.detach().float().contiguous()
            keep.append(x)
            held[key].append(x)
            return x.data_ptr()
        # fused stages (csrc/wavenet.hip, ABI 7): gate_l from z_{l-1} and x_{l-1}(t) through the extended rows built below -- one
        # dependent launch per layer instead of two
        import os
        fuse = os.environ.get("VIAI_WN_FUSED", "1") != "0" and S <= 256 and self.out_channels <= 256 and G // 2 <= 256
        r5 = math.sqrt(0.5)
        prev = None
        for i, f in enumerate(self.conv_layers):
            d = f.conv.dilation[0]
            ring = torch.zeros(B, 2 * d + 1, Cc, device=dev)
            keep.append(ring)
            L = layers[i]
            L.w_conv = t(normed_weight(f.conv).permute(0, 2, 1).reshape(G, -1))       # linearised (conv.py:53-57)
            L.b_conv = t(f.conv.bias)
            L.w_c = th("w_c", normed_weight(f.conv1x1c).reshape(G, -1)) if (f.conv1x1c is not None and cond is not None) else None
            L.b_c = t(f.conv1x1c.bias) if (f.conv1x1c is not None and cond is not None) else None
            L.w_out, L.b_out = th("w_out", normed_weight(f.conv1x1_out).reshape(Cc, -1)), th("b_out", f.conv1x1_out.bias)
            L.w_skip, L.b_skip = th("w_skip", normed_weight(f.conv1x1_skip).reshape(S, -1)), th("b_skip", f.conv1x1_skip.bias)
            L.ring, L.dilation, L.ring_len = ring.data_ptr(), d, 2 * d + 1
            if fuse:
                # (set-up arithmetic, once per synthesis call, in fp64 on the host: no library GEMM on the device path)
                wlin = normed_weight(f.conv).permute(0, 2, 1).reshape(G, -1).double().cpu()     # [Wc^0 | Wc^1 | Wc^2]
                bias = f.conv.bias.double().cpu()
                if f.conv1x1c is not None and cond is not None:
                    bias = bias + f.conv1x1c.bias.double().cpu()
                if prev is not None:
                    wc2 = wlin[:, 2 * Cc:]
                    wo, bo = normed_weight(prev.conv1x1_out).reshape(Cc, -1).double().cpu(), prev.conv1x1_out.bias.double().cpu()
                    wlin = torch.cat((wlin[:, :2 * Cc], r5 * wc2, r5 * (wc2 @ wo)), 1)
                    bias = bias + r5 * (wc2 @ bo)
                L.w_stage, L.b_stage = th("w_stage", wlin.float().to(dev)), th("b_stage", bias.float().to(dev))
                prev = f
            # global conditioning adds conv1x1g(g) + bias to the gate pre-activation at every step (modules.py:195-199):
            # computed once per layer by the HIP 1x1 conv and handed to the step kernel as a per-stream constant
            L.g_add = (t(conv1d_apply(g_vec.transpose(1, 2).reshape(B, 1, 1, -1).contiguous(), f.conv1x1g).reshape(B, G))
                       if (g_vec is not None and f.conv1x1g is not None) else None)
        out = torch.zeros(B, T, device=dev)
        logits = torch.zeros(B, T, self.out_channels, device=dev) if return_logits else None
        z = torch.zeros(B, G // 2, device=dev)
        z2 = torch.zeros(B, G // 2, device=dev)
        skips = torch.zeros(B, S, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev)        # time index, advanced on the device by each step
        st = _lib.WnSynth()
        st.B, st.C, st.G, st.S, st.cin, st.n_layers, st.out_ch, st.T = B, Cc, G, S, (cond.size(2) if cond is not None else 4), len(self.conv_layers), self.out_channels, T
        st.n_test = tin.size(1) if tin is not None else 0
        st.log_scale_min = float(log_scale_min)
        st.layers = layers
        st.w_first, st.b_first = t(normed_weight(self.first_conv).reshape(-1)), t(self.first_conv.bias)
        st.w_l1, st.b_l1 = t(normed_weight(self.last_conv_layers[1]).reshape(S, -1)), t(self.last_conv_layers[1].bias)
        st.w_l2, st.b_l2 = t(normed_weight(self.last_conv_layers[3]).reshape(self.out_channels, -1)), t(self.last_conv_layers[3].bias)
        st.cond = cond.data_ptr() if cond is not None else None
        st.test_inputs = tin.data_ptr() if tin is not None else None
        st.u1, st.u2, st.out, st.z, st.skips, st.step = u1.data_ptr(), u2.data_ptr(), out.data_ptr(), z.data_ptr(), skips.data_ptr(), step.data_ptr()
        st.yhat_dbg = logits.data_ptr() if logits is not None else None
        st.z2, st.fused = z2.data_ptr(), 1 if fuse else 0
        ref = Ct.byref(st)
        # the pipelined form (csrc/wavenet_pipe.hip): one persistent launch, the stages work on different streams at the same time.  Reference-size
        # network with local conditioning only; everything else (and use_graph) takes the chain of launches below.
        pipe = (not use_graph) and fuse and os.environ.get("VIAI_WN_PIPE", "1") != "0" and bool(lib.viai_wn_pipe_ok(ref))
        if not pipe and B not in (1, 2, 4, 8):
            raise NotImplementedError("incremental_forward: the chain of launches takes 1, 2, 4 or 8 streams; any other count up to 32 needs the pipelined form "
                                      "(reference-size network, local conditioning only, no use_graph, VIAI_WN_PIPE != 0, a device with 256 compute units)")
        if pipe:
            imgs = _pipe_images(held["w_stage"], held["b_stage"], held["w_c"], held["w_out"], held["b_out"], held["w_skip"], held["b_skip"],
                                normed_weight(self.last_conv_layers[1]).reshape(S, -1).detach().float(), self.last_conv_layers[1].bias.detach().float(),
                                normed_weight(self.last_conv_layers[3]).reshape(self.out_channels, -1).detach().float(), self.last_conv_layers[3].bias.detach().float(), dev)
            for k, im in enumerate((imgs[0], imgs[2], imgs[3], imgs[4], imgs[5], imgs[1])):
                assert im.numel() == lib.viai_wn_pipe_image_floats(k), (k, im.numel(), lib.viai_wn_pipe_image_floats(k))
            dil = (Ct.c_int * len(self.conv_layers))(*[f.conv.dilation[0] for f in self.conv_layers])
            tok = torch.zeros(lib.viai_wn_pipe_token_granules(B, dil), dtype=torch.int64, device=dev)
            err = torch.zeros(4, dtype=torch.int32, device=dev)
            w0 = min(int(timing.get("warmup", 0)), T) if timing is not None else 0

            def run(t0, n):
                _lib.check(lib.viai_wn_pipe_run(ref, imgs[0].data_ptr(), imgs[1].data_ptr(), imgs[2].data_ptr(), imgs[3].data_ptr(), imgs[4].data_ptr(), imgs[5].data_ptr(),
                                                tok.data_ptr(), err.data_ptr(), t0, n, torch.cuda.current_stream().cuda_stream), "viai_wn_pipe_run")
            if w0 > 0:
                run(0, w0)
            if timing is not None:
                import time
                torch.cuda.synchronize()
                t_start = time.perf_counter()
            for t0 in tqdm(range(w0, T, 1024)):
                run(t0, min(1024, T - t0))
            if timing is not None:
                torch.cuda.synchronize()
                timing["ms"], timing["steps"], timing["form"] = (time.perf_counter() - t_start) * 1e3, T - w0, "pipe"
            e = err.tolist()
            if e[0] != 0:
                raise _lib.ViaiLibraryError("viai_wn_pipe_run failed on the device: %s at stage %d, stream %d, t = %d (the pipelined form needs all of its 249 blocks "
                                            "resident at once, i.e. the whole chip to itself; VIAI_WN_PIPE=0 selects the chain of launches)"
                                            % ("a wait timed out" if e[0] == 1 else "a past tap was missing", e[1], e[2], e[3]))
        elif use_graph and T > 2:
            # device-side time index: one step captured into a HIP graph and replayed (every kernel starts with a load of the index)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.check(lib.viai_wavenet_synth_step(ref, torch.cuda.current_stream().cuda_stream), "viai_wavenet_synth_step")   # step 0 (warm-up)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                _lib.check(lib.viai_wavenet_synth_step(ref, torch.cuda.current_stream().cuda_stream), "viai_wavenet_synth_step")   # captured: step 1
            for _ in tqdm(range(T - 1)):
                graph.replay()
        else:
            # default: the C side loops over the time steps and hands every kernel its time index by value
            chunk = 64
            w0 = min(int(timing.get("warmup", 0)), T) if timing is not None else 0
            if w0 > 0:
                _lib.check(lib.viai_wavenet_synth_run(ref, 0, w0, torch.cuda.current_stream().cuda_stream), "viai_wavenet_synth_run")
            if timing is not None:
                import time
                torch.cuda.synchronize()
                t_start = time.perf_counter()
            for t0 in tqdm(range(w0, T, chunk)):
                _lib.check(lib.viai_wavenet_synth_run(ref, t0, min(chunk, T - t0), torch.cuda.current_stream().cuda_stream), "viai_wavenet_synth_run")
            if timing is not None:
                torch.cuda.synchronize()
                timing["ms"], timing["steps"] = (time.perf_counter() - t_start) * 1e3, T - w0
        torch.cuda.current_stream().synchronize()
        del keep
        res = out.unsqueeze(1)                                                        # (B, 1, T) like the reference
        return (res, logits) if return_logits else res

    def make_generation_fast_(self):
        def rm(m):
            try:
                nn.utils.remove_weight_norm(m)
            except ValueError:
                return
        self.apply(rm)


class DiscretizedMixturelogisticLoss(nn.Module):
    """loss_functions.py:43-62; input (B, C, T) or the NHWC tensor of WaveNet.forward_nhwc, target (B, T, 1)."""

    def __init__(self, quantize_channels=65536, log_scale_min=math.log(1e-14)):
        super().__init__()
        self.quantize_channels, self.log_scale_min = quantize_channels, log_scale_min

    def forward(self, input, target, lengths=None, mask=None, max_len=None):
        if lengths is None and mask is None:
            raise RuntimeError("Should provide either lengths or mask")
        if mask is None:
            mask = sequence_mask(lengths, max_len).unsqueeze(-1)
        if input.dim() == 3:                                                          # (B, C, T) -> NHWC rows, padded to 32
            B, Cc, T = input.shape
            yh = torch.nn.functional.pad(input.transpose(1, 2), (0, (-Cc) % 4)).reshape(B, 1, T, -1).contiguous()
        else:
            yh = input
        return mol_loss(yh, target, mask, self.quantize_channels, self.log_scale_min)
