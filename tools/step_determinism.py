"""run-to-run determinism of the three-stream step at the benchmark size: gradient arenas and `fake` of fresh models must be BIT-identical
(found the packed-FMA / ds_read2 miscompute described in csrc/conv_direct.hip: cin1_lds_taps)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viai_amd.model import AudioModel, StepConfig
from viai_amd import synth
hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 256, 256
s = synth.mel_batch(16, 256, 256, "b.s", 0).cuda(); mask = synth.time_mask(16, 256, "b.m", 0).cuda()
res = []
print('mel', hex(s.data_ptr()))
for rep in range(4):
    torch.manual_seed(0)
    m = AudioModel(hp, device="cuda")
    m.set_inputs(s, mask)
    for i in range(1):
        m.forward_backward_no_update()
    torch.cuda.synchronize()
    print('rep', rep, 'mel', hex(m.mel.data_ptr()), 'fake', hex(m._fake.data_ptr()), 'mel intact', bool(torch.equal(m.mel, s)))
    res.append((m.arena_D.grad.clone(), m.arena_G.grad.clone(), m.fake.clone()))
for r in res[1:]:
    print([bool(torch.equal(a, b)) for a, b in zip(res[0], r)], [float((a - b).abs().max()) for a, b in zip(res[0], r)])
# which parameters differ
m0 = m
dD = (res[0][0] - res[1][0]).abs()
for nme, o, p in zip(m.arena_D.names, m.arena_D.offsets, m.arena_D.params):
    d = float(dD[o:o + p.numel()].max())
    if d > 0: print("D", nme, d)
dG = (res[0][1] - res[1][1]).abs()
for nme, o, p in zip(m.arena_G.names, m.arena_G.offsets, m.arena_G.params):
    d = float(dG[o:o + p.numel()].max())
    if d > 0: print("G", nme, d)
