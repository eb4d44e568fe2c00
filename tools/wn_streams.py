"""Aggregate synthesis rate of the pipelined WaveNet kernel against the number of streams (the benchmark's configs[4] is 8 streams: 8 tokens in a ring of 27 stages,
i.e. latency-bound -- more streams fill the idle stages).   python tools/wn_streams.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viai_amd.wavenet import WaveNet  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
net = WaveNet(dropout=0.0).to(dev).eval()
hop, T = 256, 1280
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 12, 16, 24, 32]:
    c = torch.rand(B, 80, T // hop, device=dev)
    timing = {"warmup": 256}
    net.incremental_forward(None, c=c, T=T, log_scale_min=-7.0, timing=timing)
    us = timing["ms"] / timing["steps"] * 1e3
    print("streams %2d  form %-5s  %7.2f us per time step  %9.1f samples/s  (%.2f x real time per stream at 16 kHz)" % (B, timing.get("form"), us, B * 1e6 / us, 1e6 / us / 16000.0))
