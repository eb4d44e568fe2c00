#!/usr/bin/env python3
"""Run ONE conv layer (forward + backward) a few times so rocprofv3 can attribute counters to its kernels.

    rocprofv3 --kernel-trace --stats -d out -- python tools/profile_layer.py            # D.conv3 of cfg 2
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- python tools/profile_layer.py   # one --pmc pass per counter set
    python tools/profile_layer.py --shape 16 256 256 32 32 --transposed                 # a halo-kernel layer

Default = D.conv3 (Discriminator_Networks.py:44, 256 -> 512 channels, 3x3, on the 64 x 32 map of a 256 x 256 mel at batch 16),
the layer that holds 51 % of the step's FLOPs and runs on the dominant kernel of bench.py's roofline."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viai_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=5, default=[16, 64, 32, 256, 512], metavar=("N", "H", "W", "CIN", "COUT"))
ap.add_argument("--transposed", action="store_true")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--bn", action="store_true", help="BatchNorm + LeakyReLU behind the conv: the backward then takes the f16x2 gradient kernels")
ap.add_argument("--stride", type=int, nargs="+", default=[1])
ap.add_argument("--kernel", type=int, nargs=2, default=[3, 3])
ap.add_argument("--pad", type=int, nargs=2, default=None)
ap.add_argument("--zeros", action="store_true", help="all-zero input, weights and gradient: same instruction stream, no toggling -- the DVFS headroom of the layer's kernels")
ap.add_argument("--p16", action="store_true", help="hand the layer a pre-split (P16) input, as the BatchNorm pass in front of it does in the networks")
a = ap.parse_args()
N, H, W, Ci, Co = a.shape
x = ((torch.rand(N, H, W, Ci, device="cuda") * 2 - 1) * (0.0 if a.zeros else 1.0)).requires_grad_(True)
kh, kw = a.kernel
sh, sw = (a.stride * 2)[:2]
pad = tuple(a.pad) if a.pad else (kh // 2, kw // 2)
wshape = (Ci, Co, kh, kw) if a.transposed else (Co, Ci, kh, kw)
w = ((torch.rand(*wshape, device="cuda") - 0.5) * (0.0 if a.zeros else 0.1)).requires_grad_(True)
bn = torch.nn.BatchNorm2d(Co).cuda() if a.bn else None
if a.p16:
    from viai_amd import _lib
    one, zero = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    xp = torch.empty_like(x)
    am = torch.zeros(1, device="cuda")
    _lib.check(_lib.load().viai_bn_act_fwd_p16(x.data_ptr(), one.data_ptr(), zero.data_ptr(), one.data_ptr(), zero.data_ptr(), 16, xp.data_ptr(), N * H * W, Ci, 0, 0.2,
                                               am.data_ptr(), torch.cuda.current_stream().cuda_stream), "p16")
    xp._viai_p16, xp._viai_amax = True, am
    x = xp.requires_grad_(True)
    x._viai_p16, x._viai_amax = True, am
for _ in range(a.iters):
    ops.begin_step(x.device)
    y = ops.conv_bn_act(x, w, None, bn, kernel=(kh, kw), stride=(sh, sw), padding=pad, transposed=a.transposed,
                        act=ops.ACT_LRELU if a.bn else ops.ACT_NONE)
    y.backward((torch.rand_like(y) - 0.5) * (0.0 if a.zeros else 1.0))
torch.cuda.synchronize()
print("ok", tuple(y.shape))
