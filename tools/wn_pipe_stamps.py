"""Where a time step of the pipelined WaveNet synthesis goes (csrc/wavenet_pipe.hip, viai_wn_pipe_profile): wall-clock stamps of one time step on
compute unit 0 of every stage -> per stage: wait (token complete - wait begins), compute (results ready - token complete), publish, and the HOP from the
previous stage's publish to this stage's token-complete.   python tools/wn_pipe_stamps.py [t]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viai_amd import _lib  # noqa: E402
from viai_amd.wavenet import WaveNet  # noqa: E402

t_prof = int(sys.argv[1]) if len(sys.argv) > 1 else 700
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
B, hop, T = 8, 256, 1024
net = WaveNet(dropout=0.0).to(dev).eval()
c = torch.rand(B, 80, T // hop, device=dev)
lib = _lib.load()
buf = torch.zeros(2 * 27 * 8 * 8, dtype=torch.int64, device=dev)
lib.viai_wn_pipe_profile(buf.data_ptr(), t_prof)
timing = {"warmup": 256}
net.incremental_forward(None, c=c, T=T, log_scale_min=-7.0, timing=timing)
lib.viai_wn_pipe_profile(None, -1)
print("form %s, %.2f us per time step" % (timing.get("form"), timing["ms"] / timing["steps"] * 1e3))
st = buf[:27 * 8 * 8].view(27, 8, 8).cpu().double() * 0.01          # us (100 MHz)
cy = buf[27 * 8 * 8:2 * 27 * 8 * 8].view(27, 8, 8).cpu().double()
t0 = st[0, 0, 0].item()
print("stream 0 of time step %d (us since stage 0 began to wait):" % t_prof)
# layer stages: stamp 1 = x part of the token complete, 2 = z part complete, 3 = z_l published; head stages: 1 = token complete, 3 = published
print("%5s %9s %9s %9s %9s %9s" % ("stage", "wait_x", "x->z", "z->pub", "hop_z", "t_done"))
for k in range(27):
    lay = st[k, 0, 2] > 0
    wx, xz, zp = st[k, 0, 1] - st[k, 0, 0], st[k, 0, 2] - st[k, 0, 1], st[k, 0, 3] - (st[k, 0, 2] if lay else st[k, 0, 1])
    hop_ = (st[k, 0, 2] if lay else st[k, 0, 1]) - st[k - 1, 0, 3] if k > 0 else float("nan")
    print("%5d %9.2f %9.2f %9.2f %9.2f %9.2f" % (k, wx, xz if lay else float("nan"), zp, hop_, st[k, 0, 3] - t0))
rev = [(st[26, s, 3] - st[0, s, 1]).item() for s in range(B)]
print("revolution (stage 0 token complete -> sample published) per stream:", ["%.1f" % r for r in rev])
print("stage 0: start of stream s relative to stream 0:", ["%.1f" % (st[0, s, 1] - st[0, 0, 1]).item() for s in range(B)])
ghz = [((cy[k, 0, 3] - cy[k, 0, 1]) / ((st[k, 0, 3] - st[k, 0, 1]) * 1e3)).item() for k in range(1, 24)]
print("shader clock over (x complete -> published), stages 1 .. 23: %.2f - %.2f GHz" % (min(ghz), max(ghz)))
occ = [(st[k, 1, 0] - st[k, 0, 3]).item() for k in range(1, 24)]
print("published(stream 0) -> next token's wait begins (past taps + conditioning of stream 1), stages 1 .. 23: %.2f - %.2f us" % (min(occ), max(occ)))
# inside z -> pub of the layer stages (a -DVIAI_WN_FINE_STAMPS build), in shader cycles of thread 0: residual rows | barrier 1 | gate rows | barrier 2 | tanh / sigmoid + stores
if float(cy[1, 0, 4]) > 0:
    for a_, b_, name in [(2, 4, "residual rows"), (4, 5, "barrier 1"), (5, 6, "gate rows"), (6, 7, "barrier 2"), (7, 3, "finalize + stores")]:
        v = [(cy[k, 0, b_] - cy[k, 0, a_]).item() for k in range(1, 24)]
        print("  %-20s %6.0f - %6.0f cycles (median %6.0f)" % (name, min(v), max(v), sorted(v)[len(v) // 2]))
