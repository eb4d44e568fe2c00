"""CPU: the STFT/mel oracle against the cross-checks that exist in this environment (parity with lws/librosa
themselves is UNPINNED — neither is installed; see oracle/audio_oracle.py header)."""
import numpy as np
import torch

from oracle import audio_oracle as A


def test_lws_frame_arithmetic_table():
    # expected values computed from the formulas of utils/audio.py:90-108 for (length, 1024, 256)
    for length, M in ((65536, 259), (65537, 260), (66560, 263), (1000, 7), (256, 4), (1, 4)):
        assert A.lws_num_frames(length, 1024, 256) == M, length
        l, r = A.lws_pad_lr(length, 1024, 256)
        assert l == 768
        assert (length + l + r - 1024) % 256 == 0
        assert (length + l + r - 1024) // 256 + 1 == M


def test_mel_basis_matches_transformers_slaney():
    from transformers.audio_utils import mel_filter_bank
    for n_mels in (80, 256):
        ours = A.mel_basis(16000, 1024, n_mels, 125, 7600)
        ref = mel_filter_bank(513, n_mels, 125.0, 7600.0, 16000, norm="slaney", mel_scale="slaney").T
        assert ours.shape == ref.shape == (n_mels, 513)
        assert np.abs(ours - ref).max() < 1e-6 * np.abs(ref).max()


def test_stft_matches_torch_stft():
    rng = np.random.RandomState(0)
    y = rng.uniform(-0.5, 0.5, 4096)
    win = A.lws_window(1024, 256)
    D = A.stft_lws(y, 1024, 256, win)                            # [M][513]
    l, r = A.lws_pad_lr(len(y), 1024, 256)
    yp = torch.from_numpy(np.concatenate([np.zeros(l), y, np.zeros(r)]))
    T = torch.stft(yp, 1024, hop_length=256, win_length=1024, window=torch.from_numpy(win), center=False, return_complex=True)
    assert T.shape[1] == D.shape[0]
    assert np.abs(T.numpy().T - D).max() < 1e-9


def test_known_answers():
    cfg = A.AudioConfig
    t = np.arange(16384) / cfg.sample_rate
    mel = A.melspectrogram(0.5 * np.sin(2 * np.pi * 1000.0 * t), cfg)
    assert mel.shape == (80, A.lws_num_frames(16384, 1024, 256))
    basis = A.mel_basis(cfg.sample_rate, cfg.fft_size, cfg.num_mels, cfg.fmin, cfg.fmax)
    expect = int(np.argmax(basis[:, int(round(1000.0 / (cfg.sample_rate / cfg.fft_size)))]))
    mid = mel[:, mel.shape[1] // 2]
    assert abs(int(np.argmax(mid)) - expect) <= 1                 # pure tone -> one mel peak
    assert mel.min() >= 0.0 and mel.max() <= 1.0
    silent = A.melspectrogram(np.zeros(4096), cfg)
    assert np.all(silent == 0.0)                                  # silence -> floor -> 0 after normalise
