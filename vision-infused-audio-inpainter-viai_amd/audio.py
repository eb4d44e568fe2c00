"""On-device mel front end: host side of `viai_stft_mel` (reference: utils/audio.py:70-144).

`MelFrontEnd(cfg)(wav, mask=None)` == normalised mel spectrograms of a batch of clips, computed by the
fused STFT -> |.| -> mel -> dB -> [0,1] (-> mask) HIP kernel.  The analysis window and mel basis are built
once on the host (numpy) following the lws / librosa conventions the reference relies on; since neither
library is available to check against, this stage is "parity unpinned" (DESIGN.md §4).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class AudioConfig:
    sample_rate = 16000
    fft_size = 1024
    hop_size = 256
    num_mels = 80
    fmin = 125
    fmax = 7600
    min_level_db = -100
    ref_level_db = 20


def lws_num_frames(length, fsize, fshift):
    """utils/audio.py:90-98"""
    pad = fsize - fshift
    return (length + pad * 2 - fsize) // fshift + (1 if length % fshift == 0 else 2)


def lws_pad_lr(length, fsize, fshift):
    """utils/audio.py:101-108"""
    M = lws_num_frames(length, fsize, fshift)
    pad = fsize - fshift
    return pad, pad + (M - 1) * fshift + fsize - (length + 2 * pad)


def sqrt_hann_window(fsize, fshift):
    n = np.arange(fsize, dtype=np.float64)
    return np.sqrt((0.5 - 0.5 * np.cos(2.0 * np.pi * n / (fsize - 1))) * 2.0 * fshift / fsize)


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)
    freqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    edges = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    ramps = edges[:, None] - freqs[None, :]
    fd = np.diff(edges)
    w = np.maximum(0, np.minimum(-ramps[:-2] / fd[:-1, None], ramps[2:] / fd[1:, None]))
    return w * (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]


class MelFrontEnd:
    def __init__(self, cfg=AudioConfig, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.window = torch.tensor(sqrt_hann_window(cfg.fft_size, cfg.hop_size), dtype=torch.float32, device=self.device)
        basis = slaney_mel_basis(cfg.sample_rate, cfg.fft_size, cfg.num_mels, cfg.fmin, cfg.fmax)
        self.basis_t = torch.tensor(np.ascontiguousarray(basis.T), dtype=torch.float32, device=self.device)   # [bins][mels]
        # support of every band (the triangles are a few bins wide): what viai_stft_mel_banded reads instead of all fft/2 + 1 bins
        nz = basis.astype(np.float32) != 0
        lo = np.where(nz.any(1), nz.argmax(1), 0)
        hi = np.where(nz.any(1), nz.shape[1] - nz[:, ::-1].argmax(1), 0)
        self.band_lo = torch.tensor(lo, dtype=torch.int32, device=self.device)
        self.band_cnt = torch.tensor(hi - lo, dtype=torch.int32, device=self.device)

    def num_frames(self, n_samples):
        return lws_num_frames(n_samples, self.cfg.fft_size, self.cfg.hop_size)

    def __call__(self, wav: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
        """wav (B, n_samples) fp32 on the GPU -> (B, 1, num_mels, frames); mask (B, frames) optional."""
        lib = _lib.load()
        if not wav.is_cuda:
            raise _lib.ViaiLibraryError("MelFrontEnd runs on the GPU only; no CPU fallback")
        wav = wav.contiguous().float()
        B, n = wav.shape
        frames = self.num_frames(n)
        c = self.cfg
        out = torch.empty((B, 1, c.num_mels, frames), device=wav.device, dtype=torch.float32)
        m = None
        if mask is not None:
            m = mask.reshape(B, frames).contiguous().float()
        st = torch.cuda.current_stream().cuda_stream
        if c.fft_size == 1024 and not getattr(self, "force_dense", False):
            _lib.check(lib.viai_stft_mel_banded(wav.data_ptr(), self.window.data_ptr(), self.basis_t.data_ptr(), self.band_lo.data_ptr(),
                                                self.band_cnt.data_ptr(), 0 if m is None else m.data_ptr(), out.data_ptr(), B, n, c.fft_size,
                                                c.hop_size, c.num_mels, frames, float(c.min_level_db), float(c.ref_level_db), st),
                       "viai_stft_mel_banded")
        else:
            _lib.check(lib.viai_stft_mel(wav.data_ptr(), self.window.data_ptr(), self.basis_t.data_ptr(),
                                         0 if m is None else m.data_ptr(), out.data_ptr(), B, n, c.fft_size, c.hop_size,
                                         c.num_mels, frames, float(c.min_level_db), float(c.ref_level_db), st), "viai_stft_mel")
        return out


def inv_mel_amplitude(S, min_level_db=None):
    """`_db_to_amp(_denormalize(S))` of utils/audio.py:135-144 in one pass: the normalised mel the generator emits
    -> linear amplitudes (the step after the path; the vocoder consumes the normalised mel directly)."""
    from .ops import _require, _stream
    _require(S)
    S = S if S.is_contiguous() else S.contiguous()
    out = torch.empty_like(S)
    db = float(AudioConfig.min_level_db if min_level_db is None else min_level_db)
    _lib.check(_lib.load().viai_mel_denorm_amp(S.data_ptr(), out.data_ptr(), S.numel(), db, _stream()), "viai_mel_denorm_amp")
    return out
