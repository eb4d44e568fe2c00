#!/usr/bin/env python3
"""What a plain streaming pass reaches on this box: fill (write only), copy (read + write), sum-free read (abs-max) of an n-MB fp32 tensor."""
import sys
import torch
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 134.2
n = int(mb * 1e6 / 4)
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
def t(f, it=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for name, f, b in (("fill", lambda: x.fill_(1.0), 1), ("copy", lambda: y.copy_(x), 2), ("mul", lambda: torch.mul(x, 2.0, out=y), 2), ("absmax", lambda: x.abs().max(), 1)):
    us = t(f)
    print("%-7s %7.1f MB  %7.1f us  %6.2f TB/s" % (name, mb, us, b * mb / us))
