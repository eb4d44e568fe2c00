import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viai_amd import synth
from viai_amd.model import AudioModel, StepConfig
hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths, hp.batch_size = 256, 256, 16
m = AudioModel(hp, device=torch.device("cuda"), use_graph=False)
s = synth.mel_batch(16, 256, 256, "b.s", 0).cuda()
mask = synth.time_mask(16, 256, "b.m", 0).cuda()
m.set_inputs(s, mask)
for i in range(5): m.optimize_parameters(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): m.optimize_parameters(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("cpu enqueue ms/step %.2f, total ms/step %.2f" % ((t1 - t0) * 50, (t2 - t0) * 50))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(10): m.optimize_parameters(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
