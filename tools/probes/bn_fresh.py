#!/usr/bin/env python3
"""Why is bn_act_fwd slower inside the step than alone?  Variants on G.cb5 / G.cb4 / D.bn2_1 shapes, cold buffers (rotation over > 1.5 GB):
  a) as in bn_stream.py (amax already at its maximum: no atomics)   b) amax = null   c) a fresh zeroed amax slot per launch (the step's case)
  d) y written by a kernel just before (producer -> consumer, as behind a conv)      e) c + d
Per-kernel times from torch profiler-free event pairs around EACH launch (so a producer's time is not included)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viai_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr() if t is not None else None
def run(name, M, C, it=24):
    R = max(2, int(1.5e9 / (3 * M * C * 4)))
    ys = [torch.randn(M, C, device=dev) for _ in range(R)]; zs = [torch.empty(M, C, device=dev) for _ in range(R)]; srcs = [torch.randn(M, C, device=dev) for _ in range(R)]
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
    big = torch.zeros(1, device=dev) + 1e30
    slots = torch.zeros(4096, device=dev)
    res = {}
    for variant in "abcde":
        ev = []
        slots.zero_()
        for i in range(it + 4):
            k = i % R
            if variant in "de":
                torch.mul(srcs[k], 1.0001, out=ys[k])          # producer kernel: y freshly written
            am = big if variant in "ad" else None if variant == "b" else slots[i:i + 1]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.viai_bn_act_fwd_amax(p(ys[k]), p(sc), p(sh), p(zs[k]), M, C, 1, 0.0, p(am), st)
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[4:])
        res[variant] = ts[len(ts) // 2]
    print("%-8s M=%7d C=%4d %5.1f MB  " % (name, M, C, M * C * 4e-6) + "  ".join("%s %6.1f us" % (v, res[v]) for v in "abcde"))
for a in (("G.cb5", 262144, 32), ("G.cb4", 131072, 32), ("G.cb3", 32768, 64), ("D.bn2_1", 131072, 128), ("D.bn2_2", 32768, 256)):
    run(*a)
