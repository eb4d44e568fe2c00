"""GPU: fused STFT->mel->dB->[0,1] (+mask) kernel vs the numpy oracle (fp64) on synthetic waveforms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import audio_oracle as A


@pytest.mark.parametrize("n_mels,n_samples", [(80, 16384), (256, 65536), (80, 5000), (80, 5001)])      # (odd sample counts take the 16-points-per-lane kernel: no 8-byte pairs)
def test_stft_mel_matches_oracle(n_mels, n_samples):
    from viai_amd import synth
    from viai_amd.audio import AudioConfig, MelFrontEnd

    class Cfg(AudioConfig):
        num_mels = n_mels

    class OCfg(A.AudioConfig):
        num_mels = n_mels
    fe = MelFrontEnd(Cfg, device="cuda")
    wav = synth.waveform(3, n_samples, tag="a.wav")
    out = fe(wav.cuda()).cpu().numpy()
    frames = A.lws_num_frames(n_samples, 1024, 256)
    assert out.shape == (3, 1, n_mels, frames)
    for b in range(3):
        ref = A.melspectrogram(wav[b].numpy().astype(np.float64), OCfg)
        # normalised dB scale: 1e-4 == 0.01 dB
        assert np.abs(out[b, 0] - ref).max() < 2e-4, (b, np.abs(out[b, 0] - ref).max())
    # fused mask
    mask = synth.time_mask(3, frames, "a.mask").reshape(3, frames)
    outm = fe(wav.cuda(), mask.cuda()).cpu().numpy()
    assert np.array_equal(outm[:, 0], out[:, 0] * mask.numpy()[:, None, :])
    # the frame-batched banded kernel (default for fft 1024) and the one-frame-per-block dense kernel are the same function
    fe.force_dense = True
    dense = fe(wav.cuda()).cpu().numpy()
    assert np.abs(dense - out).max() < 2e-5, np.abs(dense - out).max()
    assert int(fe.band_cnt.max()) < 64 and int(fe.band_cnt.min()) >= 1            # the Slaney triangles are narrow: banded = sparse


def test_front_end_feeds_the_model():
    """on-device front end -> AudioModel.set_inputs (the north star's first stage)."""
    from viai_amd import synth
    from viai_amd.audio import AudioConfig, MelFrontEnd
    from viai_amd.model import AudioModel, StepConfig

    class Cfg(AudioConfig):
        num_mels = 80
    fe = MelFrontEnd(Cfg, device="cuda")
    wav = synth.waveform(2, 7424, tag="b.wav").cuda()          # -> 32 frames
    mel = fe(wav)
    assert mel.shape == (2, 1, 80, 32)
    hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 80, 32
    m = AudioModel(hp, device="cuda")
    m.set_inputs(mel)
    m.optimize_parameters(0)
    v = m.get_loss_items()
    assert all(np.isfinite(v))
