"""CPU oracle for the STFT -> mel -> dB -> [0,1] front end.  TEST INFRASTRUCTURE ONLY.

Restates `melspectrogram` of the reference's utils/audio.py:70-75 in numpy (fp64).

PARITY PARTLY PINNED (round 2): the plain-numpy parts of utils/audio.py -- lws_num_frames / lws_pad_lr (:90-108), _amp_to_db
(:130-132), _normalize (:139-140) and the inverse pair _denormalize / _db_to_amp (:135-144) -- are now exercised through the
REFERENCE's own functions by tools/make_goldens.py::loader_goldens (utils/audio.py imports with empty stand-ins for `lws` and
`librosa`, which those functions never touch) and committed as tests/golden/loader.npz.  What remains
UNPINNED is what lives inside the two absent third-party packages: utils/audio.py cannot be run end to end here — its arithmetic lives in two third-party
packages that are neither installed, vendored nor version-pinned by the reference (no requirements
file): `lws` (`lws.lws(fft_size, hop, mode="speech").stft`, call site utils/audio.py:71,86-87) and
`librosa.filters.mel(sr, n_fft, fmin=, fmax=, n_mels=)` (utils/audio.py:125-127; positional sr/n_fft =>
librosa < 0.10).  The reference holds no test vectors for this stage.  What IS restated from the
reference's own text: the pipeline (:70-75), the lws frame/pad arithmetic (:90-108), `_amp_to_db`
(:130-132) and `_normalize` (:139-140).  What is taken from the libraries' documented conventions
(not verifiable in this container): lws's default analysis window sqrt(hann(fsize, symmetric) * 2*hop/fsize)
with zero padding of (fsize - hop) samples on both sides and an unnormalised one-sided FFT; librosa's
Slaney mel scale with Slaney (area) normalisation.  tests/test_audio_*.py cross-check the mel basis against
`transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` and the framing/FFT
against `torch.stft`, plus known-answer cases (pure tone -> one mel peak, silence -> zeros).
"""
from __future__ import annotations

import numpy as np


class AudioConfig:
    """upstream r9y9/wavenet_vocoder defaults (the reference's Config module is missing)."""
    sample_rate = 16000
    fft_size = 1024
    hop_size = 256
    num_mels = 80
    fmin = 125
    fmax = 7600
    min_level_db = -100
    ref_level_db = 20


def lws_num_frames(length, fsize, fshift):
    """utils/audio.py:90-98."""
    pad = fsize - fshift
    if length % fshift == 0:
        return (length + pad * 2 - fsize) // fshift + 1
    return (length + pad * 2 - fsize) // fshift + 2


def lws_pad_lr(length, fsize, fshift):
    """utils/audio.py:101-108 (takes the length instead of the array)."""
    M = lws_num_frames(length, fsize, fshift)
    pad = fsize - fshift
    T = length + 2 * pad
    r = (M - 1) * fshift + fsize - T
    return pad, pad + r


def lws_window(fsize, fshift):
    """lws default analysis window for a scalar window size: sqrt(hann(fsize, symmetric) * 2*hop/fsize)."""
    n = np.arange(fsize, dtype=np.float64)
    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (fsize - 1))
    return np.sqrt(hann * 2.0 * fshift / fsize)


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults (htk=False, norm='slaney')
    -> [n_mels][n_fft//2 + 1]."""
    fftfreqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def stft_lws(y, fsize, fshift, window=None):
    """frames of `fsize` samples every `fshift`, (fsize - fshift) zeros prepended, zeros appended to cover
    lws_num_frames frames; one-sided unnormalised FFT -> complex [M][fsize//2 + 1]."""
    y = np.asarray(y, dtype=np.float64)
    M = lws_num_frames(len(y), fsize, fshift)
    l, r = lws_pad_lr(len(y), fsize, fshift)
    yp = np.concatenate([np.zeros(l), y, np.zeros(max(r, 0))])
    win = lws_window(fsize, fshift) if window is None else np.asarray(window, dtype=np.float64)
    frames = np.stack([yp[m * fshift: m * fshift + fsize] for m in range(M)])
    return np.fft.rfft(frames * win[None, :], axis=1)


def amp_to_db(x, min_level_db):
    """utils/audio.py:130-132 (pinned: tests/golden/loader.npz `amp_to_db_norm`, produced by the reference's own function)"""
    min_level = np.exp(min_level_db / 20.0 * np.log(10.0))
    return 20.0 * np.log10(np.maximum(min_level, x))


def normalize(S, min_level_db):
    """utils/audio.py:139-140 (pinned, same fixture)"""
    return np.clip((S - min_level_db) / -min_level_db, 0.0, 1.0)


def melspectrogram(y, cfg=AudioConfig, window=None, basis=None):
    """utils/audio.py:70-75 -> [num_mels][M] in [0, 1]."""
    D = stft_lws(y, cfg.fft_size, cfg.hop_size, window).T                         # :71  (F, M)
    B = mel_basis(cfg.sample_rate, cfg.fft_size, cfg.num_mels, cfg.fmin, cfg.fmax) if basis is None else basis
    mel = B @ np.abs(D)                                                            # :116-120
    S = amp_to_db(mel, cfg.min_level_db) - cfg.ref_level_db                        # :132, :72
    return normalize(S, cfg.min_level_db)                                          # :139-140
