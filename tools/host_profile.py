import os, sys, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch
from viai_amd.model import AudioModel, StepConfig
from viai_amd import synth
dev = torch.device("cuda:0")
hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 256, 256
m = AudioModel(hp, device=dev)
s = synth.mel_batch(16, 256, 256, "bench.s", 0).to(dev); mask = synth.time_mask(16, 256, "bench.mask", 0).to(dev)
m.set_inputs(s, mask)
for i in range(5): m.optimize_parameters(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(20): m.optimize_parameters(5 + i)
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); ps = pstats.Stats(pr, stream=st).sort_stats("tottime"); ps.print_stats(22)
print(st.getvalue()[:5000])
