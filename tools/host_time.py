"""host time to enqueue ONE train step into an idle device queue (no back-pressure): eager vs launch plan vs hipGraph"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viai_amd.model import AudioModel, StepConfig
from viai_amd import synth

hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 256, 256
s = synth.mel_batch(16, 256, 256, "b.s", 0).cuda(); mask = synth.time_mask(16, 256, "b.m", 0).cuda()
for name, kw in (("eager", {}), ("plan", {"use_plan": True}), ("graph", {"use_graph": True})):
    m = AudioModel(hp, device="cuda", **kw)
    m.set_inputs(s, mask)
    for i in range(10):
        m.optimize_parameters(i)
    torch.cuda.synchronize()
    host, total = [], []
    for i in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.optimize_parameters(10 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
    print("%-6s host enqueue %.3f ms (min %.3f)   one isolated step %.3f ms" % (name, statistics.median(host), min(host), statistics.median(total)))
    del m
