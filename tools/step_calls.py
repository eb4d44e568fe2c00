#!/usr/bin/env python3
"""One eager train step of the metric config with every libviai_hip.so entry point wrapped: prints entry point, call count and one argument sample
(integers / floats only; pointers as 'p') -- the list bench.py's whole-step floor model is written against."""
import collections, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viai_amd import _lib, synth
from viai_amd.model import AudioModel, StepConfig
lib = _lib.load()
hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths, hp.batch_size = 256, 256, 16
dev = torch.device("cuda", 0)
m = AudioModel(hp, device=dev)
s = synth.mel_batch(16, 256, 256, "bench.s", 0).to(dev); mask = synth.time_mask(16, 256, "bench.mask", 0).to(dev)
m.set_inputs(s, mask)
for i in range(2): m.optimize_parameters(i)
torch.cuda.synchronize()
calls = collections.OrderedDict()
def wrap(name):
    fn = getattr(lib, name)
    def w(*a):
        rec = calls.setdefault(name, [0, None])
        rec[0] += 1
        if rec[1] is None:
            rec[1] = ["p" if (isinstance(x, int) and x > 1 << 32) else (x if isinstance(x, (int, float)) else type(x).__name__) for x in a]
        return fn(*a)
    setattr(lib, name, w)
for name in _lib.SIGNATURES:
    if hasattr(lib, name): wrap(name)
m.optimize_parameters(2)
torch.cuda.synchronize()
for k, (n, a) in calls.items():
    print(n, k, a)
