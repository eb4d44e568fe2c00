"""WaveNet incremental synthesis at the reference size (24 layers, 512/512/256), 8 streams: steps/s with setup excluded"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viai_amd.wavenet import WaveNet
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
GRAPH = len(sys.argv) > 2 and sys.argv[2] == 'graph'
net = WaveNet(dropout=0.0).cuda().eval()
cs = torch.rand(8, 80, T // 256, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.incremental_forward(None, c=cs, T=T, log_scale_min=-7.0, use_graph=GRAPH)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter()
    net.incremental_forward(None, c=cs[:, :, :T // 512], T=T // 2, log_scale_min=-7.0, use_graph=GRAPH)
    dt2 = time.perf_counter() - t1
    per = (dt - dt2) / (T - T // 2)
    print("T=%d: %.3f s, T=%d: %.3f s -> %.1f us per time step = %.0f steps/s = %.0f samples/s (8 streams); setup %.3f s" % (T, dt, T // 2, dt2, per * 1e6, 1 / per, 8 / per, dt - per * T))
