import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import viai_oracle as O
from viai_amd.model import AudioModel, StepConfig
graph = sys.argv[1] == "graph"; B2 = int(sys.argv[2])
hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 80, 32
s = O.cf_uniform("gc.s", (2, 1, 80, 32), 0, 1).cuda(); mask = O.make_mask(2, 32, "gc.mask").cuda()
s2 = O.cf_uniform("gc.s2", (B2, 1, 80, 32), 0, 1).cuda(); mask2 = O.make_mask(B2, 32, "gc.mask2").cuda()
m = AudioModel(hp, device="cuda", use_graph=graph)
m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
m.set_inputs(s, mask)
m.optimize_parameters(0); torch.cuda.synchronize(); print("step0 ok", flush=True)
m.optimize_parameters(1); torch.cuda.synchronize(); print("step1 ok", flush=True)
m.set_inputs(s2, mask2); print("set_inputs new shape ok", flush=True)
m.optimize_parameters(2); torch.cuda.synchronize(); print("step2 ok", graph, B2, flush=True)
m.optimize_parameters(3); torch.cuda.synchronize(); print("step3 ok", graph, B2, flush=True)
