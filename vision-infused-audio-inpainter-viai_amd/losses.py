"""HIP-backed drop-ins for the reference's loss modules (loss_functions.py).

`GANLoss` and `L2ContrastiveLoss` keep the reference's constructor / call signatures; the arithmetic is the
fused kernels of libviai_hip.so.  (The reference's own `loss_functions.GANLoss` also works unchanged on the
tensors this package produces — these classes only remove the torch elementwise launches.)"""
from __future__ import annotations

import random

import torch
import torch.nn as nn

from . import ops
from .wavenet import DiscretizedMixturelogisticLoss, sequence_mask   # noqa: F401  (loss_functions.py:11-21,43-62 live in this module in the reference)


class GANLoss(nn.Module):
    """loss_functions.py:79-104: BCELoss (use_lsgan=False) or MSELoss against an expanded scalar label,
    optional random soft label (real - U(0,0.1), fake + U(0,0.1))."""

    def __init__(self, use_lsgan=True, device=None, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.device = device
        self.register_buffer("real_label", torch.tensor(target_real_label))
        self.register_buffer("fake_label", torch.tensor(target_fake_label))
        self.use_lsgan = use_lsgan
        self._real, self._fake = float(target_real_label), float(target_fake_label)

    def get_target_value(self, target_is_real, softlabel):
        soft = random.random() * 0.1 if softlabel else 0.0
        return self._real - soft if target_is_real else self._fake + soft

    def __call__(self, input, target_is_real, softlabel=False):
        t = self.get_target_value(target_is_real, softlabel)
        return ops.mse_mean(input, t) if self.use_lsgan else ops.bce_mean(input, t)


class L2ContrastiveLoss(nn.Module):
    """loss_functions.py:112-148."""

    def __init__(self, margin=0, measure=False, max_violation=False):
        super().__init__()
        self.margin = margin
        self.max_violation = max_violation

    def forward(self, feature1, feature2):
        return ops.l2_contrastive(feature1, feature2, self.margin, self.max_violation)


class ExponentialMovingAverage(object):
    """loss_functions.py:65-76, same methods; the update is one HIP kernel per registered tensor (register the flat
    parameter arena of a model once and the whole model is averaged by a single launch)."""

    def __init__(self, decay):
        self.decay = decay
        self.shadow = {}

    def register(self, name, val):
        self.shadow[name] = val.detach().clone()

    def update(self, name, x):
        assert name in self.shadow
        from . import _lib
        from .ops import _require, _stream
        sh = self.shadow[name]
        x = x.detach()
        _require(sh, x)
        x = x if x.is_contiguous() else x.contiguous()
        _lib.check(_lib.load().viai_ema_update(sh.data_ptr(), x.data_ptr(), sh.numel(), float(self.decay), _stream()), "viai_ema_update")
