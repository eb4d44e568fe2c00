"""the reference's train loop (train_whole_sync.py:32-145) walked over `AudioModel` -- every method
the loop calls, in the loop's order, with a synthetic loader that yields the 8-tuples of Data_loaders/audio_loader.py:532, a stub
SummaryWriter and a stub visualizer; checkpoints written on the loop's cadence load into a fresh model and continue bit for bit."""
import glob
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import viai_oracle as O

B, F_BINS, T, HOP = 2, 80, 32, 4


class Writer:
    """tensorboardX.SummaryWriter stand-in"""

    def __init__(self):
        self.scalars = []

    def add_scalar(self, tag, value, step=None):
        assert isinstance(tag, str) and float(value) == float(value), (tag, value)
        self.scalars.append((tag, float(value), step))


class Visualizer:
    """visdom_utils/visualizer.py stand-in: the three methods the loop calls (:34,98,114)"""

    def __init__(self):
        self.shown, self.printed, self.plotted = [], [], []

    def display_current_results(self, visuals, epoch, step=0):
        assert set(visuals) >= {"real_mel", "masked_mel", "fake_mel"}
        for v in visuals.values():
            assert tuple(v.shape) == (B, 1, F_BINS, T)
        self.shown.append(step)

    def print_current_errors(self, epoch, i, errors, t):
        assert all(isinstance(v, float) for v in errors.values()) and t >= 0
        self.printed.append((i, dict(errors)))

    def plot_current_errors(self, epoch, counter_ratio, opt, errors):
        self.plotted.append(counter_ratio)


class Recorder:
    """forwards everything to the model and logs the method calls in order"""

    def __init__(self, model):
        object.__setattr__(self, "_m", model)
        object.__setattr__(self, "log", [])

    def __getattr__(self, name):
        v = getattr(self._m, name)
        if callable(v) and not isinstance(v, torch.nn.Module) and not name.startswith("_"):
            def call(*a, **k):
                self.log.append(name)
                return v(*a, **k)
            return call
        return v

    def __setattr__(self, name, value):
        setattr(self._m, name, value)


def batch(tag):
    """(video, flow, c (B,C,T), x (B,1,T*hop), y, g, lengths, paths): audio_loader.py:532 (the audio-only model reads c)"""
    c = O.cf_uniform("loop.c." + tag, (B, F_BINS, T))
    x = O.cf_uniform("loop.x." + tag, (B, 1, T * HOP), -1, 1)
    y = (x.transpose(1, 2) * 0).long()
    return (None, None, c, x, y, None, torch.full((B,), T * HOP, dtype=torch.long), ["clip%s_%d" % (tag, i) for i in range(B)])


def hparams(tmp):
    from viai_amd.model import StepConfig
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths, hp.batch_size = F_BINS, T, B
    hp.name, hp.nepochs = "viai_loop", 1
    hp.print_freq, hp.display_freq, hp.display_id = 1, 2, 1
    hp.checkpoint_interval, hp.checkpoint_dir = 2, str(tmp)
    hp.train_eval_interval, hp.test_eval_epoch_interval = 1, 1
    return hp


def fresh(hp):
    from viai_amd.model import AudioModel
    m = AudioModel(hp, device="cuda")
    m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
    return m


def test_reference_train_loop_sequence_runs_on_audiomodel(tmp_path):
    from viai_amd import train
    hp = hparams(tmp_path)
    torch.manual_seed(11)                       # set_inputs draws the time gaps from torch's generator
    model = Recorder(fresh(hp))
    loaders = {"train": [batch("tr%d" % i) for i in range(4)], "test": [batch("te%d" % i) for i in range(2)]}
    writer, vis = Writer(), Visualizer()
    state, history = train.train_loop(model, loaders, writer, hp, visualizer=vis, checkpoint_dir=str(tmp_path))
    assert (state.global_step, state.global_epoch, state.global_test_step) == (4, 1, 2)

    # ---- the call order of train_whole_sync.py:46-112, iteration by iteration
    head = ["get_blank_space_length", "set_inputs"]
    tail = ["TF_writer", "del_no_need"]
    want = []
    for gs_after in (1, 2, 3, 4):               # global_step after the iteration's optimize_parameters
        it = list(head)
        if gs_after == 4:                       # evals are scheduled at steps 1, 2, 3; every third one runs (:65-69)
            it.append("eval_model_test")
        it += ["optimize_parameters", "get_loss_items"]
        if gs_after % hp.display_freq == 0:
            it.append("get_current_visuals")
        it.append("get_current_errors")         # print_freq = 1
        if gs_after % hp.checkpoint_interval == 0:
            it.append("save_inpainting_checkpoint")
        want += it + tail
    for _ in range(2):                          # test phase (:78-84): global_step stays 4 -> display fires (4 % 2 == 0), no print
        want += head + ["test", "get_loss_items", "get_current_visuals"] + tail
    # (what the model calls on itself -- TF_writer -> get_current_errors, eval_model_test -> test -- does not pass the recorder)
    assert model.log == want, (model.log, want)

    # ---- per-step scalars, per-epoch scalars of both phases, retrieval scalars
    tags = [t for t, _, _ in writer.scalars]
    assert tags.count("viai_loop_mel_L1") == 6 and tags.count("viai_loop_D") == 6
    for phase in ("train", "test"):
        assert "viai_loop_mel_L1_%s loss (per epoch)" % phase in tags
        assert "viai_loop_reconstruction_%s loss (per epoch)" % phase in tags
        assert "viai_loop_%s EmbeddingL2loss (per epoch)" % phase in tags
    assert "val_video_retrieval top1" in tags and "val_audio_retrieval top1" in tags
    assert history[0]["train"]["mel_l1"] > 0 and history[0]["test"]["mel_l1"] > 0
    assert len(history[0]["test"]["video_retrieval"]) == 6 and len(history[0]["test"]["audio_retrieval"]) == 6
    assert [i for i, _ in vis.printed] == [1, 2, 3, 4] and set(vis.printed[0][1]) == {"D", "G", "G_GAN", "mel_L1"}
    assert vis.shown == [2, 4, 4, 4] and len(vis.plotted) == 4
    assert model.train == 0                     # the loop left the model in the test phase's mode
    assert tuple(model.mel_net_norm.shape) == (B, 256) and tuple(model.video_net_norm.shape) == (B, 256)
    assert os.path.isfile(os.path.join(str(tmp_path), "train_eval", "step%09d_mel.pt" % 3))

    # ---- checkpoints on the loop's cadence; a fresh model resumes from the last one and continues bit for bit
    files = sorted(glob.glob(os.path.join(str(tmp_path), "viai_loop_checkpoint_step*.pth.tar")))
    assert [os.path.basename(f) for f in files] == ["viai_loop_checkpoint_step%09d.pth.tar" % s for s in (2, 4)]
    ck = torch.load(files[-1], map_location="cpu", weights_only=False)
    assert list(ck.keys()) == ["Mel_Encoder", "Mel_Decoder", "netD", "optimizer_G", "optimizer_D", "global_step", "global_epoch", "global_test_step"]
    resumed = fresh(hp)
    assert resumed.load_inpainting_checkpoint(files[-1]) == (4, 0, 0)
    s, mask = O.cf_uniform("s.tiny", (B, 1, F_BINS, T)), O.make_mask(B, T, "mask.tiny")
    orig = model._m
    orig.train = 1
    for m in (orig, resumed):
        m.set_inputs(s, mask)
        m.optimize_parameters(4)
        m.get_loss_items()
    torch.cuda.synchronize()
    assert torch.equal(orig.losses, resumed.losses)
    assert torch.equal(orig.arena_G.flat, resumed.arena_G.flat) and torch.equal(orig.arena_D.flat, resumed.arena_D.flat)
    assert torch.equal(orig.netD.norm3.running_var, resumed.netD.norm3.running_var)


def test_run_resumes_and_saves_on_the_way_out(tmp_path):
    """train_whole_sync.py:148-187: resume from a checkpoint, run the loop, save whatever ends it (here: an interrupt)."""
    from viai_amd import train
    hp = hparams(tmp_path)
    hp.checkpoint_interval = 0
    model = fresh(hp)
    first = model.save_inpainting_checkpoint(7, 3, str(tmp_path), 0, hparams=hp)

    class Interrupting(list):
        def __iter__(self):
            yield self[0]
            raise KeyboardInterrupt

    torch.manual_seed(5)
    state = train.run(fresh(hp), {"train": Interrupting([batch("r0"), batch("r1")])}, Writer(), hp, resume_path=first)
    assert (state.global_step, state.global_test_step) == (8, 3)
    assert os.path.isfile(os.path.join(str(tmp_path), "viai_loop_checkpoint_step%09d.pth.tar" % 8))
