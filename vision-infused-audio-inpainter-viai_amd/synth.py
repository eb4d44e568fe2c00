"""Closed-form synthetic inputs ("MUSICES-shaped"): integer-hash
uniforms that are bit-identical on every machine, so tests, smoke and bench can
regenerate exactly the tensors the golden fixtures were produced from without
shipping data."""
from __future__ import annotations

import math

import numpy as np
import torch


def _tag_id(tag: str) -> int:
    h = 2166136261
    for ch in tag.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def uniform(tag: str, shape, lo=0.0, hi=1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(_tag_id(tag))) & np.uint64(0xFFFFFFFF)
    for _ in range(2):
        h ^= h >> np.uint64(16)
        h = (h * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    u = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32).reshape(shape))


def mel_batch(batch, bins, frames, tag="s", rank=0):
    """s ~ U[0,1) fp32 (B,1,F,T); rank-keyed so data-parallel ranks draw different clips."""
    return uniform("%s.r%d" % (tag, rank), (batch, 1, bins, frames))


def time_mask(batch, frames, tag="mask", rank=0):
    """one full-height gap per clip: L = T/4, t0 ~ U{T/8 .. 5T/8} -> (B,1,1,T) in {0,1}."""
    u = uniform("%s.r%d" % (tag, rank), (batch,)).numpy()
    m = np.ones((batch, 1, 1, frames), dtype=np.float32)
    L = max(frames // 4, 1)
    lo, hi = frames // 8, (5 * frames) // 8
    for i in range(batch):
        t0 = min(lo + int(u[i] * (hi - lo + 1)), frames - L)
        m[i, :, :, t0:t0 + L] = 0.0
    return torch.from_numpy(m)


def waveform(batch, n_samples, sr=16000, tag="wav", rank=0):
    """0.5*U(-1,1) noise + 3 sinusoids per clip."""
    y = 0.25 * uniform("%s.n.r%d" % (tag, rank), (batch, n_samples), -1, 1)
    f = uniform("%s.f.r%d" % (tag, rank), (batch, 3), 100.0, 4000.0)
    t = torch.arange(n_samples, dtype=torch.float64)[None, :] / sr
    for k in range(3):
        y = y + (0.2 * torch.sin(2 * math.pi * f[:, k:k + 1].double() * t)).float()
    return y.contiguous()
