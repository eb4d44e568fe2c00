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
