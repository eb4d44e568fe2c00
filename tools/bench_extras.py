#!/usr/bin/env python
"""Side measurements for the §8 rows that are not the headline metric (run on the GPU box; results quoted in DESIGN.md):
STFT/mel front end (GB/s), WaveNet teacher-forced step (TFLOP/s) and incremental synthesis (samples/s, vs the CPU
oracle), ResNet-18 visual branch forward+backward (frames/s)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    from viai_amd import synth
    from viai_amd.audio import AudioConfig, MelFrontEnd
    from viai_amd.networks import ImageEmbedding2
    from viai_amd.wavenet import WaveNet, mol_loss
    res = {}
    # ---- STFT -> mel (+mask): 16 clips x 65536 samples, 256 mels
    class Cfg(AudioConfig):
        num_mels = 256
    fe = MelFrontEnd(Cfg)
    wav = synth.waveform(16, 65536).cuda()
    dt = timed(lambda: fe(wav), 50)
    frames = fe.num_frames(65536)
    alg = 16 * (65536 * 4 + 256 * frames * 4)
    res["stft_mel"] = {"ms": round(dt * 1e3, 3), "clips_per_s": round(16 / dt, 1), "algorithmic_GBps": round(alg / dt * 1e-9, 1)}
    # ---- WaveNet teacher-forced fwd+bwd, reference size (24 layers, 512/512/256), B=8, T=8192
    net = WaveNet(dropout=0.0).cuda().train()
    B, T = 8, 8192
    x = torch.rand(B, 1, T, device="cuda") * 2 - 1
    c = torch.rand(B, 80, T // 256, device="cuda")
    y = torch.rand(B, T, 1, device="cuda") * 2 - 1

    def step():
        net.zero_grad(set_to_none=True)
        mol_loss(net.forward_nhwc(x, c), y, None).backward()
    dt = timed(step, 3, 1)
    flops = 3 * 49.3e6 * B * T                                    # fwd + 2x bwd, 49.3 MFLOP/sample
    res["wavenet_train_step"] = {"B": B, "T": T, "ms": round(dt * 1e3, 1), "samples_per_s": round(B * T / dt), "TFLOPs": round(flops / dt * 1e-12, 1)}
    # ---- WaveNet incremental synthesis, 8 streams
    net.eval()
    Ts = 2048
    cs = torch.rand(8, 80, Ts // 256, device="cuda")
    t0 = time.perf_counter()
    net.incremental_forward(None, c=cs, T=Ts, log_scale_min=-7.0)
    dt = time.perf_counter() - t0
    res["wavenet_incremental"] = {"streams": 8, "T": Ts, "s": round(dt, 3), "steps_per_s": round(Ts / dt), "samples_per_s": round(8 * Ts / dt)}
    del net
    # ---- ResNet-18 visual branch fwd+bwd, 64 frames (B=1, N=64)
    V = ImageEmbedding2().cuda().train()
    video = torch.rand(1, 64, 3, 224, 224, device="cuda") * 2 - 1
    flow = torch.rand(1, 64, 2, 224, 224, device="cuda") * 2 - 1

    def vstep():
        V.zero_grad(set_to_none=True)
        o, f = V(video, flow)
        (o.pow(2).mean() + f.pow(2).mean()).backward()
    dt = timed(vstep, 3, 1)
    res["image_embedding2_fwd_bwd"] = {"frames": 64, "ms": round(dt * 1e3, 1), "frames_per_s": round(64 / dt, 1),
                                       "TFLOPs": round(3 * 2 * 3.6e9 * 64 / dt * 1e-12, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
