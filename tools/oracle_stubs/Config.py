"""Stub for the reference's missing `Config` module.

Values follow the upstream r9y9/wavenet_vocoder defaults the reference fork
(wavenet_vocoder/version.py: 0.0.5+2092a64) was configured from; the reference
snapshot itself does not pin them.  Used only by tools/make_goldens.py.
"""
import math


class Config(object):
    def __init__(self):
        self.name = "viai_golden"
        self.input_type = "raw"
        self.quantize_channels = 65536
        self.sample_rate = 16000
        self.out_channels = 10 * 3
        self.decode_layers = 24
        self.decode_stacks = 4
        self.residual_channels = 512
        self.gate_channels = 512
        self.skip_out_channels = 256
        self.kernel_size = 3
        self.dropout = 1 - 0.95
        self.cin_channels = 80
        self.gin_channels = -1
        self.n_speakers = 1
        self.weight_normalization = True
        self.upsample_conditional_features = True
        self.upsample_scales = [4, 4, 4, 4]
        self.freq_axis_kernel_size = 3
        self.log_scale_min = float(math.log(1e-14))
        # audio front end (utils/audio.py)
        self.fft_size = 1024
        self.hop_size = 256
        self.frame_shift_ms = None
        self.num_mels = 80
        self.fmin = 125
        self.fmax = 7600
        self.min_level_db = -100
        self.ref_level_db = 20
        self.allow_clipping_in_normalization = True
        self.silence_threshold = 2
