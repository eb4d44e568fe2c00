#!/usr/bin/env python3
"""the fused (conv + BN + act) -> (Cout = 1 conv) pair alone, forward + backward, for rocprofv3 (tools/layer_stats.sh-style):
    rocprofv3 --kernel-trace --stats -d out -- python tools/profile_pair.py [D|G]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viai_amd import networks as N_, ops
which = sys.argv[1] if len(sys.argv) > 1 else "D"
if which == "D":
    n, ci, cm, h, w, tr, act = 16, 256, 512, 64, 32, False, ops.ACT_LRELU
else:
    n, ci, cm, h, w, tr, act = 16, 32, 32, 256, 256, True, ops.ACT_RELU
mk = (lambda a, b: torch.nn.ConvTranspose2d(a, b, 3, 1, 1)) if tr else (lambda a, b: torch.nn.Conv2d(a, b, 3, 1, 1, bias=False))
c1, c2, bn = mk(ci, cm).cuda(), mk(cm, 1).cuda(), torch.nn.BatchNorm2d(cm).cuda()
x = torch.rand(n, h, w, ci, device="cuda").requires_grad_(True)
for _ in range(10):
    ops.begin_step(x.device)
    p = N_.fused_pair(x, c1, bn, act, c2, ops.ACT_SIGMOID)
    p.backward(torch.rand_like(p) - 0.5)
torch.cuda.synchronize()
print("ok", tuple(p.shape))
