"""The fused layer as a PyTorch custom op: `torch.ops.viai.conv_bn_act` (BASELINE.json north_star: "Python host code calling [the kernels] through
PyTorch-ROCm custom ops"; SURVEY.md section 7 stage 2, section 8b).

`viai_amd.ops.conv_bn_act` -- what the networks of this package call -- is a `torch.autograd.Function` over the C ABI: it works with autograd and
`torch.optim`, but the dispatcher does not know it (no schema, nothing for `torch.compile`, `torch.library.opcheck` or a profiler's op view to see).
This module registers the same forward and backward with `torch.library` ON TOP OF the same code: the op's implementation runs
`ops._ConvBnAct.forward / backward` against a stand-in context object, so there is one body of launch logic and the two entry points cannot drift.

    z = torch.ops.viai.conv_bn_act(x, weight, bias, gamma, beta, running_mean, running_var,
                                   kernel, stride, padding, transposed, act, training, momentum, eps)[0]

x: NHWC fp32 (N, H, W, Cin); weight in the torch layout of nn.Conv2d / nn.ConvTranspose2d; gamma .. num_batches_tracked = the BatchNorm2d's tensors or
None (no BatchNorm); act = ops.ACT_*.  The op is FUNCTIONAL (torch.library registers autograd formulas for functional operators only): the updated running
statistics are OUTPUTS, which the `conv_bn_act()` wrapper below copies into the module's buffers.  Reference semantics: nn.Conv2d / nn.ConvTranspose2d -> nn.BatchNorm2d -> LeakyReLU(0.2) / ReLU / Sigmoid
(Inpainting_Networks.py:71-76, New_Inpainting_Networks.py:31-37, Discriminator_Networks.py:38-49).  Outputs: (z, y, coef, xa, new_running_mean,
new_running_var) -- the layer's output, the three tensors its backward reads (pre-BatchNorm map, BatchNorm coefficients, the operand magnitude of the
f16x2 split) and the statistics; use `conv_bn_act()` below for just z.  Plain form only: one source tensor, fp32 in and out, no residual / pool / resize tail (those stay with `ops.conv_bn_act`).

The registered backward is an op of its own, `torch.ops.viai.conv_bn_act_backward`.  As with every op of this package there is no CPU implementation:
a CPU tensor raises.  `ops.begin_step(device)` must have been called once per step (it re-arms the magnitude slots), as the networks do."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _lib, ops

__all__ = ["conv_bn_act"]


class _Ctx:
    """what ops._ConvBnAct.forward / backward need of an autograd context"""

    def __init__(self, needs):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _cfg(kernel, stride, padding, transposed, act, training, momentum, eps):
    return {"k": tuple(kernel), "s": tuple(stride), "p": tuple(padding), "transposed": bool(transposed), "act": int(act), "training": bool(training),
            "momentum": float(momentum), "eps": float(eps), "d": (1, 1), "p2": (-1, -1), "xa_in": (None, None), "xmask": None, "pool": None, "up": None,
            "p16_out": False, "x_twin": None}


def _empty(dev):
    return torch.empty(0, device=dev, dtype=torch.float32)


@torch.library.custom_op("viai::conv_bn_act", mutates_args=(), device_types="cuda")
def _conv_bn_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                 running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
                 kernel: List[int], stride: List[int], padding: List[int], transposed: bool, act: int, training: bool, momentum: float,
                 eps: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    if ops.is_p16(x):
        raise ValueError("torch.ops.viai.conv_bn_act takes fp32 tensors (pre-split P16 tensors stay inside viai_amd.ops.conv_bn_act)")
    ctx = _Ctx((True, False, True, bias is not None, gamma is not None, beta is not None, False, False, False, False, False))
    cfg = _cfg(kernel, stride, padding, transposed, act, training, momentum, eps)
    track = running_mean is not None and running_var is not None
    if gamma is not None and not training and not track:
        cfg["training"] = True                                                   # nn.BatchNorm2d without running statistics normalises with batch statistics
    # a FUNCTIONAL op (torch.library registers autograd formulas for those only): the kernels update copies of the running statistics, which are returned
    rm = running_mean.clone() if track else None
    rv = running_var.clone() if track else None
    z = ops._ConvBnAct.forward(ctx, x, None, weight, bias, gamma, beta, rm, rv, None, None, cfg)
    saved = ctx.saved_tensors
    y, coef = saved[3], saved[4]
    dev = x.device
    # outputs must not alias each other: without BatchNorm the saved map IS z (the backward op takes z for it)
    return (z, y if (y is not None and y is not z) else _empty(dev), coef if coef is not None else _empty(dev),
            ctx.xa.reshape(-1).clone() if ctx.xa is not None else _empty(dev), rm if track else _empty(dev), rv if track else _empty(dev))


def _out_hw(x, weight, kernel, stride, padding, transposed):
    kh, kw = kernel
    if transposed:
        return (x.shape[1] - 1) * stride[0] - 2 * padding[0] + kh, (x.shape[2] - 1) * stride[1] - 2 * padding[1] + kw, weight.shape[1]
    return (x.shape[1] + 2 * padding[0] - kh) // stride[0] + 1, (x.shape[2] + 2 * padding[1] - kw) // stride[1] + 1, weight.shape[0]


@_conv_bn_act.register_fake
def _(x, weight, bias, gamma, beta, running_mean, running_var, kernel, stride, padding, transposed, act, training, momentum, eps):
    oh, ow, cout = _out_hw(x, weight, kernel, stride, padding, transposed)
    z = x.new_empty((x.shape[0], oh, ow, cout))
    has_bn = gamma is not None
    fused1 = has_bn and bias is None and x.shape[3] == 1                       # the fused Cin = 1 layer stores no pre-BatchNorm map (it may still: a data-dependent library decision)
    track = running_mean is not None and running_var is not None
    return (z, x.new_empty((0,)) if (not has_bn or fused1) else torch.empty_like(z), x.new_empty((4, cout)) if has_bn else x.new_empty((0,)),
            x.new_empty((1,)), torch.empty_like(running_mean) if track else x.new_empty((0,)), torch.empty_like(running_var) if track else x.new_empty((0,)))


@torch.library.custom_op("viai::conv_bn_act_backward", mutates_args=(), device_types="cuda")
def _conv_bn_act_backward(dz: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, z: torch.Tensor, y: torch.Tensor, coef: torch.Tensor, xa: torch.Tensor,
                          has_bias: bool, kernel: List[int], stride: List[int], padding: List[int], transposed: bool, act: int, training: bool,
                          momentum: float, eps: float, needs: List[bool]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    has_bn = coef.numel() > 0
    ctx = _Ctx((needs[0], False, needs[1], needs[2] and has_bias, needs[3] and has_bn, needs[4] and has_bn, False, False, False, False, False))
    cfg = _cfg(kernel, stride, padding, transposed, act, training, momentum, eps)
    N, IH, IW, C1 = x.shape
    cout = weight.shape[1] if transposed else weight.shape[0]
    cin_w = weight.shape[0] if transposed else weight.shape[1]
    if C1 == 4 and 1 < cin_w < 4:
        C1 = cin_w
    d = ops.conv_desc(N, IH, IW, C1, 0, cout, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], 1 if transposed else 0)
    fused1 = bool(has_bn and training and not has_bias and C1 == 1 and lib.viai_conv2d_cin1_bn_ok(d["ref"]))
    ctx.d, ctx.cfg, ctx.has_bn, ctx.has_bias = d, cfg, has_bn, has_bias
    ctx.dims = (N, IH, IW, C1, 0, cout, d["OH"], d["OW"])
    ctx.fused1, ctx.tail, ctx.x_p16, ctx.x_twin_w, ctx.xmask = fused1, None, False, None, None
    ctx.xa = xa.reshape(1) if xa.numel() else None
    ctx.saved_tensors = (x, None, weight, None if fused1 else (y if has_bn else z), coef if has_bn else None)
    g = ops._ConvBnAct.backward(ctx, dz)
    dev = dz.device
    pick = lambda t: t if t is not None else _empty(dev)
    return pick(g[0]), pick(g[2]), pick(g[3]), pick(g[4]), pick(g[5])


@_conv_bn_act_backward.register_fake
def _(dz, x, weight, z, y, coef, xa, has_bias, kernel, stride, padding, transposed, act, training, momentum, eps, needs):
    cout = weight.shape[1] if transposed else weight.shape[0]
    e = x.new_empty((0,))
    has_bn = coef.numel() > 0
    return (torch.empty_like(x) if needs[0] else e, torch.empty_like(weight) if needs[1] else e, x.new_empty((cout,)) if (needs[2] and has_bias) else e,
            x.new_empty((cout,)) if (needs[3] and has_bn) else e, x.new_empty((cout,)) if (needs[4] and has_bn) else e)


def _setup_context(ctx, inputs, output):
    x, weight, bias, gamma, beta, rm, rv, kernel, stride, padding, transposed, act, training, momentum, eps = inputs
    z, y, coef, xa, new_rm, new_rv = output
    ctx.save_for_backward(x, weight, z, y, coef, xa)
    track = rm is not None and rv is not None
    ctx.args = (bias is not None, list(kernel), list(stride), list(padding), bool(transposed), int(act),
                bool(training) or (gamma is not None and not track), float(momentum), float(eps))
    ctx.mark_non_differentiable(y, coef, xa, new_rm, new_rv)


def _backward(ctx, dz, _dy, _dcoef, _dxa, _drm, _drv):
    x, weight, z, y, coef, xa = ctx.saved_tensors
    needs = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3], ctx.needs_input_grad[4]]
    dx, dw, db, dg, dbeta = torch.ops.viai.conv_bn_act_backward(dz.contiguous(), x, weight, z, y, coef, xa, *ctx.args, needs)
    opt = lambda t, need: t if (need and t.numel()) else None
    return (opt(dx, needs[0]), opt(dw, needs[1]), opt(db, needs[2]), opt(dg, needs[3]), opt(dbeta, needs[4])) + (None,) * 10


_conv_bn_act.register_autograd(_backward, setup_context=_setup_context)


def conv_bn_act(x, weight, bias=None, bn=None, *, kernel, stride=(1, 1), padding=(0, 0), transposed=False, act=ops.ACT_NONE, training=True):
    """`torch.ops.viai.conv_bn_act` with an nn.BatchNorm2d (or None) as the parameter holder, like `ops.conv_bn_act`; returns z (NHWC).  The op is
    functional; this wrapper writes the new running statistics into the module's buffers (nn.BatchNorm2d semantics: training mode, momentum)."""
    if bn is None:
        return torch.ops.viai.conv_bn_act(x, weight, bias, None, None, None, None, list(kernel), list(stride), list(padding), transposed, int(act),
                                          training, 0.1, 1e-5)[0]
    track = bn.track_running_stats and bn.running_mean is not None
    out = torch.ops.viai.conv_bn_act(x, weight, bias, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                                     list(kernel), list(stride), list(padding), transposed, int(act), training,
                                     0.1 if bn.momentum is None else float(bn.momentum), float(bn.eps))
    if track and training:
        with torch.no_grad():
            bn.running_mean.copy_(out[4])
            bn.running_var.copy_(out[5])
            bn.num_batches_tracked += 1
    return out[0]
