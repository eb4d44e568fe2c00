import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from viai_amd import _lib
from viai_amd.audio import AudioConfig, MelFrontEnd
class Cfg(AudioConfig):
    num_mels = 256
fe = MelFrontEnd(Cfg, device="cuda")
n = 16384
g = torch.Generator().manual_seed(1)
wav = (torch.rand(1, n, generator=g) - 0.5).cuda()
fr = fe.num_frames(n)
out = torch.zeros(1, 1, 256, fr, device="cuda")
lib = _lib.load()
_lib.check(lib.viai_stft_mel_banded(wav.data_ptr(), fe.window.data_ptr(), fe.basis_t.data_ptr(), fe.band_lo.data_ptr(), fe.band_cnt.data_ptr(), 0, out.data_ptr(),
                                    1, n, 1024, 256, 256, fr, 100.0, 20.0, torch.cuda.current_stream().cuda_stream), "stft")
torch.cuda.synchronize()
Z = out.flatten()[:1024].cpu().numpy().astype(np.float64)
Z = Z[0::2] + 1j * Z[1::2]
w = fe.window.cpu().numpy().astype(np.float64)
x = np.concatenate([np.zeros(768), wav[0].cpu().numpy().astype(np.float64), np.zeros(2048)])
xf = x[0:1024] * w
z = xf[0::2] + 1j * xf[1::2]
ref = np.fft.fft(z)
print("Z err", np.abs(Z - ref).max(), "ref max", np.abs(ref).max())
cands = {"swap re/im": np.fft.fft(xf[1::2] + 1j * xf[0::2]), "conj": np.conj(ref), "no window": np.fft.fft(x[0:1024:2] + 1j * x[1:1024:2]),
         "reversed k": ref[(-np.arange(512)) % 512]}
for k, v in cands.items():
    print(k, np.abs(Z - v).max())
print("first 6 got", np.round(Z[:6], 3), "want", np.round(ref[:6], 3))
# stage structure: which k are right?
good = np.abs(Z - ref) < 1e-3 * np.abs(ref).max()
print("good count", good.sum(), "good k (first 40):", np.where(good)[0][:40].tolist())
