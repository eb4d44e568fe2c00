import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import viai_oracle as O
from viai_amd import ops
from viai_amd.model import AudioModel, StepConfig
B2 = int(sys.argv[1]); DROP = len(sys.argv) > 2
hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 80, 32
s = O.cf_uniform("gc.s", (2, 1, 80, 32), 0, 1).cuda(); mask = O.make_mask(2, 32, "gc.mask").cuda()
s2 = O.cf_uniform("gc.s2", (B2, 1, 80, 32), 0, 1).cuda(); mask2 = O.make_mask(B2, 32, "gc.mask2").cuda()
m = AudioModel(hp, device="cuda", use_graph=True)
m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
m.set_inputs(s, mask)
m.optimize_parameters(0); torch.cuda.synchronize(); print("step0 ok", flush=True)
m.set_inputs(s2, mask2); print("set_inputs new shape ok", flush=True)
if not DROP:
    pass
ops.DIRECT_GRAD = True
segs = (m._seg_forward_dstep, m._seg_dupdate_gstep, m._seg_gupdate)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for it in range(2):
        for i, f in enumerate(segs):
            f(); torch.cuda.synchronize(); print("warm", it, i, "ok", flush=True)
torch.cuda.current_stream().wait_stream(st)
graphs = []; pool = None
for i, f in enumerate(segs):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool):
        f()
    pool = g.pool(); graphs.append(g); print("captured", i, flush=True)
for i, g in enumerate(graphs):
    g.replay(); torch.cuda.synchronize(); print("replayed", i, flush=True)
