"""Stub for the reference's missing `Options_inpainting` module.

Used ONLY by tools/make_goldens.py in the build container, to make the
reference's network modules importable from /root/reference.  Not shipped in
the product path.  Attribute names are the ones the reference reads
(`grep -rnoE "hparams\\.[A-Za-z_]+" networks/`); values are the build's choice
because the reference does not pin them (the real file is absent upstream).
"""
import torch.nn as nn


class Inpainting_Config(object):
    def __init__(self):
        self.name = "viai_golden"
        self.cin_channels = 80          # mel bins F; overwritten per shape
        self.max_mel_lengths = 208
        self.normlayer = nn.BatchNorm2d
        self.length_feature = 256
        self.image_size = 224
        self.resnet_pretrain = False
        self.resnet_pretrain_path = None
        self.sample_rate = 16000
        self.batch_size = 16
        self.save_optimizer_state = True
        # Data_loaders/audio_loader.py reads these (values = the native MUSICES geometry inferred in SURVEY.md section 8d:
        # 1280 samples per used video frame, 4 mel frames per video frame, hop 320, 52 frames = 66 560 samples per clip)
        self.max_time_sec = None
        self.max_time_steps = 66560
        self.image_hope_size = 2           # use_image_num = floor(4.16 s / (0.04 * 2)) = 52
        self.hop_size = 256            # == Config.hop_size (assert_ready_for_upsampling, audio_loader.py:44-45)
        self.load_num = 2
        self.image_rescal_size = 256
        self.image = True
        self.flow = True
        self.file_channel = -1
        self.upsample_conditional_features = True
        self.input_type = "raw"
        self.quantize_channels = 65536

