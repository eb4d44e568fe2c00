"""CPU oracle for the WaveNet vocoder path.  TEST INFRASTRUCTURE ONLY (see oracle/viai_oracle.py header).

Plain-torch functional restatement of wavenet_vocoder/{wavenet,modules,conv,mixture}.py.  PINNED:
tools/make_goldens.py imports the reference's own `WaveNet`, `discretized_mix_logistic_loss` and
`sample_from_discretized_mix_logistic` (stub Config module) and checks this file against them;
the reference's outputs are committed as tests/golden/wavenet.npz.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .viai_oracle import cf_std, cf_uniform


class WNConfig:
    """small test configuration (the reference's defaults: 24 layers / 4 stacks / 512 / 512 / 256, scales [4,4,4,4])."""
    out_channels = 30
    layers = 4
    stacks = 2
    residual_channels = 64
    gate_channels = 64
    skip_out_channels = 32
    kernel_size = 3
    cin_channels = 80
    upsample_scales = (4, 4)
    freq_axis_kernel_size = 3
    gin_channels = -1           # > 0: global (speaker) conditioning through an embedding of n_speakers rows
    n_speakers = None


class WNConfigG(WNConfig):
    """the small configuration with global conditioning (wavenet.py:146-151, modules.py:140-145)."""
    gin_channels = 8
    n_speakers = 3


class WNConfigFull(WNConfig):
    """the reference's own size (24 layers / 4 stacks / 512 residual + gate / 256 skip channels,
    up-sampling [4,4,4,4] = hop 256; 24.7 M parameters): BASELINE.json configs[4]."""
    layers = 24
    stacks = 4
    residual_channels = 512
    gate_channels = 512
    skip_out_channels = 256
    upsample_scales = (4, 4, 4, 4)


class WNConfigDeep(WNConfig):
    """the reference's DEPTH (24 layers / 4 stacks: dilations 1 .. 32 four times, receptive field 505, wavenet.py:116-131) at reduced
    width, so the reference's Python `incremental_forward` (wavenet.py:237-364, ring buffers of conv.py:17-46) finishes in seconds on the
    CPU: the fixture that pins incremental synthesis at every dilation the benchmark's configs[4] runs."""
    layers = 24
    stacks = 4
    residual_channels = 32
    gate_channels = 32
    skip_out_channels = 32


class WNConfigOneHot(WNConfig):
    """the small configuration with one-hot (mu-law, 256 classes) input and softmax-logit output (`scalar_input=False`)."""
    out_channels = 256
    scalar_input = False


def wavenet_state(cfg=WNConfig, tag="WN."):
    """state_dict of WaveNet with weight normalisation (keys: *.weight_g, *.weight_v, *.bias)."""
    sd = OrderedDict()

    def conv(name, cout, cin, k):
        v = cf_std(tag + name + ".v", (cout, cin, k), math.sqrt(1.0 / (cin * k)))
        sd[name + ".bias"] = cf_uniform(tag + name + ".b", (cout,), -0.05, 0.05)
        sd[name + ".weight_g"] = (v.reshape(cout, -1).norm(dim=1) * cf_uniform(tag + name + ".g", (cout,), 0.8, 1.2)).reshape(cout, 1, 1)
        sd[name + ".weight_v"] = v
    conv("first_conv", cfg.residual_channels, 1 if getattr(cfg, "scalar_input", True) else cfg.out_channels, 1)   # wavenet.py:116-119
    for i in range(cfg.layers):
        p = "conv_layers.%d." % i
        conv(p + "conv", cfg.gate_channels, cfg.residual_channels, cfg.kernel_size)
        conv(p + "conv1x1c", cfg.gate_channels, cfg.cin_channels, 1)
        if cfg.gin_channels > 0:
            conv(p + "conv1x1g", cfg.gate_channels, cfg.gin_channels, 1)
        conv(p + "conv1x1_out", cfg.residual_channels, cfg.gate_channels // 2, 1)
        conv(p + "conv1x1_skip", cfg.skip_out_channels, cfg.gate_channels // 2, 1)
    conv("last_conv_layers.1", cfg.skip_out_channels, cfg.skip_out_channels, 1)
    conv("last_conv_layers.3", cfg.out_channels, cfg.skip_out_channels, 1)
    if cfg.gin_channels > 0:
        sd["embed_speakers.weight"] = cf_std(tag + "embed", (cfg.n_speakers, cfg.gin_channels), 0.1)
    for j, s in enumerate(cfg.upsample_scales):
        n = "upsample_conv.%d" % (2 * j)
        v = cf_uniform(tag + n + ".v", (1, 1, cfg.freq_axis_kernel_size, s), 0.1, 0.5)
        sd[n + ".bias"] = cf_uniform(tag + n + ".b", (1,), -0.01, 0.01)
        sd[n + ".weight_g"] = (v.norm() * 1.1).reshape(1, 1, 1, 1)
        sd[n + ".weight_v"] = v
    return sd


def wn_weight(sd, name):
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v|| per output row (modules.py:39)."""
    v, g = sd[name + ".weight_v"], sd[name + ".weight_g"]
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape((-1,) + (1,) * (v.dim() - 1))
    return v * (g / n)


def wavenet_forward(sd, x, c, cfg=WNConfig, g=None):
    """WaveNet.forward, scalar input, local conditioning with up-sampling, optional global conditioning by speaker id
    g (B,) or (B, 1), eval-mode dropout (wavenet.py:177-235; ResidualConv1dGLU._forward modules.py:162-210)."""
    B, _, T = x.shape
    g_bct = None
    if g is not None:                                                        # wavenet.py:198-206
        e = F.embedding(g.view(B, -1), sd["embed_speakers.weight"]).transpose(1, 2)      # (B, gin, 1)
        g_bct = e.expand(B, -1, T)
    c = c.unsqueeze(1)                                                       # :210
    for j, s in enumerate(cfg.upsample_scales):                              # :211-212
        n = "upsample_conv.%d" % (2 * j)
        c = F.relu(F.conv_transpose2d(c, wn_weight(sd, n), sd[n + ".bias"], stride=(1, s), padding=((cfg.freq_axis_kernel_size - 1) // 2, 0)))
    c = c.squeeze(1)                                                         # :214
    assert c.shape[-1] == T
    h = F.conv1d(x, wn_weight(sd, "first_conv"), sd["first_conv.bias"])      # :218
    skips = None
    per = cfg.layers // cfg.stacks
    for i in range(cfg.layers):
        p = "conv_layers.%d." % i
        d = 2 ** (i % per)
        residual = h
        y = F.conv1d(h, wn_weight(sd, p + "conv"), sd[p + "conv.bias"], padding=(cfg.kernel_size - 1) * d, dilation=d)[:, :, :T]   # modules.py:179-181
        a, b = y.split(y.shape[1] // 2, dim=1)
        yc = F.conv1d(c, wn_weight(sd, p + "conv1x1c"), sd[p + "conv1x1c.bias"])
        ca, cb = yc.split(yc.shape[1] // 2, dim=1)
        if g_bct is not None:                                                # modules.py:195-199
            yg = F.conv1d(g_bct, wn_weight(sd, p + "conv1x1g"), sd[p + "conv1x1g.bias"])
            ga, gb = yg.split(yg.shape[1] // 2, dim=1)
            ca, cb = ca + ga, cb + gb
        z = torch.tanh(a + ca) * torch.sigmoid(b + cb)                       # :201
        s = F.conv1d(z, wn_weight(sd, p + "conv1x1_skip"), sd[p + "conv1x1_skip.bias"])
        o = F.conv1d(z, wn_weight(sd, p + "conv1x1_out"), sd[p + "conv1x1_out.bias"])
        h = (o + residual) * math.sqrt(0.5)                                  # :209
        skips = s if skips is None else (skips + s) * math.sqrt(0.5)         # wavenet.py:222-226
    h = F.relu(skips)
    h = F.relu(F.conv1d(h, wn_weight(sd, "last_conv_layers.1"), sd["last_conv_layers.1.bias"]))
    return F.conv1d(h, wn_weight(sd, "last_conv_layers.3"), sd["last_conv_layers.3.bias"])


def mol_loss_rows(y_hat, y, num_classes=65536, log_scale_min=math.log(1e-14)):
    """discretized_mix_logistic_loss(reduce=False) (mixture.py:25-105): y_hat (B,C,T), y (B,T,1) -> (B,T,1)."""
    nr_mix = y_hat.shape[1] // 3
    y_hat = y_hat.transpose(1, 2)
    logit_probs = y_hat[:, :, :nr_mix]
    means = y_hat[:, :, nr_mix:2 * nr_mix]
    log_scales = torch.clamp(y_hat[:, :, 2 * nr_mix:3 * nr_mix], min=log_scale_min)
    y = y.expand_as(means)
    centered = y - means
    inv = torch.exp(-log_scales)
    plus_in = inv * (centered + 1. / (num_classes - 1))
    min_in = inv * (centered - 1. / (num_classes - 1))
    cdf_delta = torch.sigmoid(plus_in) - torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    mid_in = inv * centered
    log_pdf_mid = mid_in - log_scales - 2. * F.softplus(mid_in)
    c3 = (cdf_delta > 1e-5).float()
    inner = c3 * torch.log(torch.clamp(cdf_delta, min=1e-12)) + (1. - c3) * (log_pdf_mid - np.log((num_classes - 1) / 2))
    c2 = (y > 0.999).float()
    inner = c2 * log_one_minus_cdf_min + (1. - c2) * inner
    c1 = (y < -0.999).float()
    log_probs = c1 * log_cdf_plus + (1. - c1) * inner + F.log_softmax(logit_probs, -1)
    return -torch.logsumexp(log_probs, dim=-1, keepdim=True)


def mol_loss(y_hat, y, mask, num_classes=65536, log_scale_min=math.log(1e-14)):
    """DiscretizedMixturelogisticLoss.forward (loss_functions.py:43-62)."""
    losses = mol_loss_rows(y_hat, y, num_classes, log_scale_min)
    m = mask.expand_as(y)
    return (losses * m).sum() / m.sum()


def mol_sample(y, u1, u2, log_scale_min=-7.0):
    """sample_from_discretized_mix_logistic (mixture.py:117-153) with the two uniform draws injected:
    u1 (B,T,nr_mix), u2 (B,T)."""
    nr_mix = y.shape[1] // 3
    y = y.transpose(1, 2)
    logit = y[:, :, :nr_mix]
    arg = (logit - torch.log(-torch.log(u1))).max(dim=-1)[1]
    onehot = F.one_hot(arg, nr_mix).float()
    means = (y[:, :, nr_mix:2 * nr_mix] * onehot).sum(-1)
    log_scales = torch.clamp((y[:, :, 2 * nr_mix:3 * nr_mix] * onehot).sum(-1), min=log_scale_min)
    x = means + torch.exp(log_scales) * (torch.log(u2) - torch.log(1. - u2))
    return torch.clamp(x, -1., 1.)


def incremental_forward(sd, c, T, u1, u2, cfg=WNConfig, test_inputs=None, log_scale_min=-7.0, g=None):
    """WaveNet.incremental_forward (wavenet.py:237-364) by definition of causality: sample t depends only on
    samples < t, so it equals running the batch forward on the growing prefix (O(T^2); tiny T only)."""
    B = c.shape[0]
    x = torch.zeros(B, 1, T)
    out = torch.zeros(B, 1, T)
    cur = torch.zeros(B)                                          # initial input: zeros (wavenet.py:305-306)
    for t in range(T):
        if test_inputs is not None and t < test_inputs.shape[-1]:
            cur = test_inputs[:, 0, t]
        elif t > 0:
            cur = out[:, 0, t - 1]
        x[:, 0, t] = cur
        y = wavenet_forward(sd, x, c, cfg, g)[:, :, t:t + 1]
        out[:, 0, t] = mol_sample(y, u1[:, t:t + 1], u2[:, t:t + 1], log_scale_min)[:, 0]
    return out


def incremental_forward_ring(sd, c, T, u1, u2, cfg=WNConfig, test_inputs=None, log_scale_min=-7.0, g=None, return_logits=False):
    """WaveNet.incremental_forward (wavenet.py:237-364) the way the reference computes it: sample by sample, every dilated conv as a
    LINEAR layer over a ring buffer of its last (k - 1) d + 1 inputs (conv.py:17-46: shift the buffer, append the new input, gather
    every d-th entry, one matrix product with the (Cout, k Cin) linearised weight conv.py:53-57) -- O(T) work, unlike
    `incremental_forward` above (the definition-of-causality form used to pin it).  This is the CPU restatement bench.py times as the
    `cpu_baseline` of configs[4]."""
    B = c.shape[0] if c is not None else (test_inputs.shape[0] if test_inputs is not None else 1)
    k = cfg.kernel_size
    per = cfg.layers // cfg.stacks
    sq = math.sqrt(0.5)
    with torch.no_grad():
        cu = None
        if c is not None:                                                    # wavenet.py:291-299: up-sample once
            cu = c.unsqueeze(1)
            for j, s in enumerate(cfg.upsample_scales):
                n = "upsample_conv.%d" % (2 * j)
                cu = F.relu(F.conv_transpose2d(cu, wn_weight(sd, n), sd[n + ".bias"], stride=(1, s), padding=((cfg.freq_axis_kernel_size - 1) // 2, 0)))
            cu = cu.squeeze(1).transpose(1, 2).contiguous()                      # (B, T, cin)
            assert cu.shape[1] == T
        g_bc = None
        if g is not None:
            g_bc = F.embedding(g.view(B, -1), sd["embed_speakers.weight"]).reshape(B, -1)
        w_first = wn_weight(sd, "first_conv").reshape(cfg.residual_channels, -1)
        layers = []
        for i in range(cfg.layers):
            p = "conv_layers.%d." % i
            d = 2 ** (i % per)
            w = wn_weight(sd, p + "conv")                                        # (G, C, k) -> (G, k C): tap-major, as conv.py:53-57
            layers.append(dict(
                d=d, ring=torch.zeros(B, (k - 1) * d + 1, cfg.residual_channels),
                w=w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous(), b=sd[p + "conv.bias"],
                wc=wn_weight(sd, p + "conv1x1c").reshape(cfg.gate_channels, -1) if cu is not None else None, bc=sd.get(p + "conv1x1c.bias"),
                wg=wn_weight(sd, p + "conv1x1g").reshape(cfg.gate_channels, -1) if g_bc is not None else None, bg=sd.get(p + "conv1x1g.bias"),
                wo=wn_weight(sd, p + "conv1x1_out").reshape(cfg.residual_channels, -1), bo=sd[p + "conv1x1_out.bias"],
                ws=wn_weight(sd, p + "conv1x1_skip").reshape(cfg.skip_out_channels, -1), bs=sd[p + "conv1x1_skip.bias"]))
        w1 = wn_weight(sd, "last_conv_layers.1").reshape(cfg.skip_out_channels, -1)
        w2 = wn_weight(sd, "last_conv_layers.3").reshape(cfg.out_channels, -1)
        out = torch.zeros(B, 1, T)
        logits = torch.zeros(B, cfg.out_channels, T) if return_logits else None
        cur = torch.zeros(B, 1)                                                  # initial input: zeros (wavenet.py:305-306)
        half = cfg.gate_channels // 2
        for t in range(T):
            if test_inputs is not None and t < test_inputs.shape[-1]:
                cur = test_inputs[:, :, t].reshape(B, -1)
            elif t > 0:
                cur = out[:, :, t - 1].reshape(B, 1)
            h = F.linear(cur, w_first, sd["first_conv.bias"])                   # (B, C)
            skips = None
            for L in layers:
                ring = L["ring"]
                ring[:, :-1] = ring[:, 1:].clone()                               # conv.py:39
                ring[:, -1] = h
                x = ring[:, ::L["d"]].reshape(B, -1)                             # conv.py:44: every d-th entry = the k taps
                y = F.linear(x, L["w"], L["b"])
                a, b = y[:, :half], y[:, half:]
                if L["wc"] is not None:
                    yc = F.linear(cu[:, t], L["wc"], L["bc"])
                    a, b = a + yc[:, :half], b + yc[:, half:]
                if L["wg"] is not None:
                    yg = F.linear(g_bc, L["wg"], L["bg"])
                    a, b = a + yg[:, :half], b + yg[:, half:]
                z = torch.tanh(a) * torch.sigmoid(b)
                s = F.linear(z, L["ws"], L["bs"])
                h = (F.linear(z, L["wo"], L["bo"]) + h) * sq
                skips = s if skips is None else (skips + s) * sq
            y = F.linear(F.relu(F.linear(F.relu(skips), w1, sd["last_conv_layers.1.bias"])), w2, sd["last_conv_layers.3.bias"])
            if logits is not None:
                logits[:, :, t] = y
            out[:, 0, t] = mol_sample(y.unsqueeze(-1), u1[:, t:t + 1], u2[:, t:t + 1], log_scale_min)[:, 0]
    return (out, logits) if return_logits else out
