import sys; sys.path.insert(0, "/root/repo")
import torch
from viai_amd import synth
from viai_amd.audio import AudioConfig, MelFrontEnd
class Cfg(AudioConfig):
    num_mels = 256
fe = MelFrontEnd(Cfg, device="cuda")
wav = synth.waveform(16, 65536).cuda().repeat(64, 1)
for _ in range(3): fe(wav)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): fe(wav)
e1.record(); torch.cuda.synchronize()
print("fe(wav) %.1f us" % (e0.elapsed_time(e1) / 10 * 1e3), wav.is_contiguous(), wav.data_ptr() % 16)
