"""which statements of the train step issue device-to-device copies / fills (they become memcpy / memset nodes of a capture)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from viai_amd.model import AudioModel, StepConfig
from viai_amd import synth

hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 256, 256
m = AudioModel(hp, device="cuda")
s = synth.mel_batch(16, 256, 256, "b.s", 0).cuda(); mask = synth.time_mask(16, 256, "b.m", 0).cuda()
m.set_inputs(s, mask)
for i in range(3):
    m.optimize_parameters(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.optimize_parameters(3)
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::add", "aten::mul", "aten::add_", "aten::ones_like", "aten::clone", "aten::sum"):
        st = [f for f in (ev.stack or []) if "viai" in f or "model.py" in f or "ops.py" in f or "networks.py" in f]
        key = (ev.name, tuple(st[:3]))
        seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k[0], " <- ".join(k[1]) if k[1] else "(no python frame: autograd engine)")
