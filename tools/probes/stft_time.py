#!/usr/bin/env python3
"""STFT -> mel kernel time on N clips (direct library call, output preallocated): python tools/probes/stft_time.py [clips] [mels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viai_amd import _lib, synth
from viai_amd.audio import AudioConfig, MelFrontEnd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mels = int(sys.argv[2]) if len(sys.argv) > 2 else 256
class Cfg(AudioConfig):
    num_mels = mels
fe = MelFrontEnd(Cfg)
n = 65536
wav = synth.waveform(16, n).cuda().repeat(B // 16, 1).contiguous()
fr = fe.num_frames(n)
out = torch.empty(B, 1, mels, fr, device="cuda")
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.viai_stft_mel_banded(wav.data_ptr(), fe.window.data_ptr(), fe.basis_t.data_ptr(), fe.band_lo.data_ptr(), fe.band_cnt.data_ptr(), 0, out.data_ptr(),
                                        B, n, 1024, 256, mels, fr, -100.0, 20.0, st), "stft")
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
by = B * (n * 4 + mels * fr * 4)
print("clips %d mels %d frames %d: %.1f us, %.1f GB/s (%.3f of 8 TB/s), band_cnt max %d sum %d" % (B, mels, fr, t * 1e6, by / t * 1e-9, by / t / 8e12, int(fe.band_cnt.max()), int(fe.band_cnt.sum())))
