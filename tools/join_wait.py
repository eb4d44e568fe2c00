#!/usr/bin/env python3
"""How long does the main stream wait for the trailing weight-gradient stream at each join?  (eager three-stream step, no profiler)
    python tools/join_wait.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viai_amd import ops
from viai_amd.model import AudioModel, StepConfig

from viai_amd import synth

dev = torch.device("cuda:0")
hp = StepConfig()
B, F, T = 16, 256, 256
hp.cin_channels, hp.max_mel_lengths = F, T
m = AudioModel(hp, device=dev)
s = synth.mel_batch(B, F, T, "bench.s", 0).to(dev)
mask = synth.time_mask(B, T, "bench.mask", 0).to(dev)
m.set_inputs(s, mask)
for i in range(5):
    m.optimize_parameters(i)
torch.cuda.synchronize()
rec = []
orig = ops.join_wgrad
def timed():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(); e1.record(); rec.append((e0, e1))
ops.join_wgrad = timed
import viai_amd.model as M
steps = 20
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(steps):
    m.optimize_parameters(5 + i)
t1.record(); torch.cuda.synchronize()
w = [a.elapsed_time(b) for a, b in rec]
per = len(w) // steps
print("step %.3f ms; joins per step %d; wait at each join (ms, mean over steps):" % (t0.elapsed_time(t1) / steps, per),
      [round(sum(w[k::per]) / steps, 3) for k in range(per)])
