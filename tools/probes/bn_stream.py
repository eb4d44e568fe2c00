#!/usr/bin/env python3
"""Standalone BatchNorm passes at the step's shapes through the C ABI: us per launch and bytes / time against a plain copy.
    python tools/probes/bn_stream.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viai_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
SHAPES = [("D.bn2_1", 131072, 128), ("D.bn2_2", 32768, 256), ("D.bn3", 32768, 512), ("G.cb5", 262144, 32), ("G.cb4", 131072, 32), ("G.cb3", 32768, 64),
          ("G.cb2", 8192, 128), ("E.bn2", 131072, 64), ("E.bn3", 32768, 128), ("E.bn4", 8192, 256)]
def timeit(f, it=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
p = lambda t: t.data_ptr()
COLD = "--cold" in sys.argv          # rotate over buffer sets of > 1.5 GB in total: every pass misses the 256 MB Infinity Cache
for name, M, C in SHAPES:
    R = max(1, int(1.5e9 / (4 * M * C * 4))) if COLD else 1
    ys = [torch.randn(M, C, device=dev) for _ in range(R)]; zs = [torch.empty(M, C, device=dev) for _ in range(R)]
    dzs = [torch.randn(M, C, device=dev) for _ in range(R)]; dys = [torch.empty(M, C, device=dev) for _ in range(R)]
    k = [0]
    def rot():
        k[0] = (k[0] + 1) % R
        return ys[k[0]], zs[k[0]], dzs[k[0]], dys[k[0]]
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev); mean = torch.randn(C, device=dev) * 0.1; inv = torch.rand(C, device=dev) + 0.5
    amax = torch.zeros(1, device=dev)
    nblk = lib.viai_bn_bwd_blocks(M, C)
    part = torch.empty(nblk * 2 * C, device=dev); sums = torch.empty(2 * C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    mb = M * C * 4 / 1e6
    def f_copy():
        y, z, dz, dy = rot(); z.copy_(y)
    t_copy = timeit(f_copy)
    def f_fwd():
        y, z, dz, dy = rot(); lib.viai_bn_act_fwd_amax(p(y), p(sc), p(sh), p(z), M, C, 2, 0.2, p(amax), st)
    t_fwd = timeit(f_fwd)
    def f_bwd():
        y, z, dz, dy = rot(); lib.viai_bn_act_bwd_amax(p(dz), p(y), p(mean), p(inv), p(sc), p(sh), p(part), p(sums), p(dg), p(db), p(dy), M, C, 2, 0.2, 1, p(amax), st)
    t_bwd = timeit(f_bwd)
    def f_red():
        y, z, dz, dy = rot(); lib.viai_bn_act_bwd_amax(p(dz), p(y), p(mean), p(inv), p(sc), p(sh), p(part), p(sums), p(dg), p(db), None, M, C, 2, 0.2, 1, p(amax), st)
    t_red = timeit(f_red)
    print("%-8s M=%7d C=%4d %6.1f MB | copy %6.1f us %5.2f TB/s | fwd %6.1f us %5.2f TB/s | bwd reduce+final %6.1f us (2x: %5.2f TB/s) | bwd all three %6.1f us (apply ~%6.1f us, 3x: %5.2f TB/s)"
          % (name, M, C, mb, t_copy, 2 * mb / t_copy, t_fwd, 2 * mb / t_fwd, t_red, 2 * mb / t_red, t_bwd, t_bwd - t_red, 3 * mb / max(t_bwd - t_red, 1e-3)))
