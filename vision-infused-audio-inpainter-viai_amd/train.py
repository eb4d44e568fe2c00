"""The caller of the hot path: the epoch / phase / iteration driver around `AudioModel`.

The reference's driver is `train_whole_sync.py:32-145` (`train_loop`), a script bound to module-level globals
(`hparams`, `visualizer`, `global_step`, ...) and to loaders this build does not reproduce (jpg / npy I/O is out of
scope).  What IS part of the boundary is the order in which it calls the model --
SURVEY.md section 3.2 -- so this module restates that contract as a function with explicit arguments:

    per iteration (train_whole_sync.py:46-112)
        get_blank_space_length(global_step) -> set_inputs(data)
        [eval_model_test every third scheduled eval]                               (:53-72)
        train phase:  model.train = 1; optimize_parameters(global_step); global_step += 1          (:75-77)
        test phase:   model.train = 0; no_grad: test(); collect mel_net_norm / video_net_norm       (:78-84)
        get_loss_items()                                                                            (:85)
        display / print hooks of the visualizer                                                    (:87-99)
        save_inpainting_checkpoint every checkpoint_interval train steps                            (:101-103)
        TF_writer(writer, step) -> running sums -> del_no_need()                                    (:105-112)
    per phase (:113-143): three per-epoch scalars to the writer; test phase: the two retrieval metric sets

`data_loaders` is any mapping phase -> iterable of the loader's 8-tuples (Data_loaders/audio_loader.py:532) with
`len()`; `writer` anything with `add_scalar(tag, value, step)`; `visualizer` (optional) anything with the three
methods of visdom_utils/visualizer.py the loop uses.  tests/test_train_loop_gpu.py drives it with synthetic clips.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import util


class LoopState:
    """the three counters the reference keeps as module globals (train_whole_sync.py:14-16) and writes into checkpoints"""

    def __init__(self, global_step=0, global_epoch=0, global_test_step=0):
        self.global_step, self.global_epoch, self.global_test_step = int(global_step), int(global_epoch), int(global_test_step)
        self.eval_count = 0                   # `count` of train_whole_sync.py:35,65-69: every third scheduled eval really runs


def to_np(x):
    """utils/util.py `to_np`: tensor -> numpy on the host"""
    return x.detach().cpu().numpy()


def _opt(hp, name, default):
    return getattr(hp, name, default)


def _wants_eval(hp, state, train, already):
    """the eval schedule of train_whole_sync.py:57-63"""
    if train:
        every = _opt(hp, "train_eval_interval", 0)
        return bool(every) and state.global_step > 0 and state.global_step % every == 0
    every = _opt(hp, "test_eval_epoch_interval", 0)
    return bool(every) and not already and state.global_epoch > 0 and state.global_epoch % every == 0


def run_phase(model, phase, data_loader, writer, hparams, state, visualizer=None, checkpoint_dir=None):
    """one pass over one loader; returns the per-phase averages (and the retrieval metrics of a test phase)"""
    train = phase == "train"
    sums = {"reconstruct": 0.0, "embedding_l2": 0.0, "mel_l1": 0.0}
    audio_ebds, image_ebds = [], []
    evaluated = False
    ckpt_root = checkpoint_dir if checkpoint_dir is not None else _opt(hparams, "checkpoint_dir", ".")
    batch = max(int(_opt(hparams, "batch_size", 1)), 1)
    n_iter = 0
    for it, data in enumerate(data_loader):
        t0 = time.time()
        n_iter += 1
        model.get_blank_space_length(state.global_step)
        model.set_inputs(data)
        if _wants_eval(hparams, state, train, evaluated):
            evaluated = evaluated or not train
            eval_dir = os.path.join(ckpt_root, "%s_eval" % phase)
            os.makedirs(eval_dir, exist_ok=True)
            if state.eval_count == 2:
                model.eval_model_test(state.global_step, eval_dir)
                state.eval_count = 0
            else:
                state.eval_count += 1
        if train:
            model.train = 1
            model.optimize_parameters(state.global_step)
            state.global_step += 1
        else:
            model.train = 0
            with torch.no_grad():
                model.test()
            audio_ebds.append(to_np(model.mel_net_norm))
            image_ebds.append(to_np(model.video_net_norm))
            state.global_test_step += 1
        model.get_loss_items()
        if state.global_step > 0:
            if visualizer is not None and _opt(hparams, "display_freq", 0) and state.global_step % hparams.display_freq == 0:
                visualizer.display_current_results(model.get_current_visuals(), state.global_epoch, step=state.global_step)
            if train and _opt(hparams, "print_freq", 0) and state.global_step % hparams.print_freq == 0:
                errors = model.get_current_errors()
                if visualizer is not None:
                    visualizer.print_current_errors(state.global_epoch, state.global_step, errors, (time.time() - t0) / batch)
                    if _opt(hparams, "display_id", 0) > 0:
                        visualizer.plot_current_errors(state.global_epoch, float(state.global_step) / len(data_loader), hparams, errors)
        every = _opt(hparams, "checkpoint_interval", 0)
        if train and every and state.global_step > 0 and state.global_step % every == 0:
            model.save_inpainting_checkpoint(state.global_step, state.global_test_step, ckpt_root, state.global_epoch, hparams=hparams)
        model.TF_writer(writer, step=state.global_step)
        if model.update_wavenet:
            sums["reconstruct"] += model.reconstruct_loss_item
        sums["embedding_l2"] += model.EmbeddingL2_item
        sums["mel_l1"] += model.loss_mel_L1_item
        model.del_no_need()
    n = max(len(data_loader) if hasattr(data_loader, "__len__") else n_iter, 1)
    avg = {k: v / n for k, v in sums.items()}
    name = _opt(hparams, "name", "viai")
    if writer is not None:
        writer.add_scalar(name + "_reconstruction_%s loss (per epoch)" % phase, avg["reconstruct"], state.global_epoch)
        writer.add_scalar(name + "_mel_L1_%s loss (per epoch)" % phase, avg["mel_l1"], state.global_epoch)
        writer.add_scalar(name + "_%s EmbeddingL2loss (per epoch)" % phase, avg["embedding_l2"], state.global_epoch)
    out = dict(avg)
    if not train and audio_ebds:
        a, v = np.concatenate(audio_ebds, axis=0), np.concatenate(image_ebds, axis=0)
        dev = model.device
        out["video_retrieval"] = util.L2retrieval(torch.from_numpy(a).to(dev), torch.from_numpy(v).to(dev))
        out["audio_retrieval"] = util.L2retrieval(torch.from_numpy(v).to(dev), torch.from_numpy(a).to(dev))
        if writer is not None:
            writer.add_scalar("val_video_retrieval top1", out["video_retrieval"][0], state.global_epoch)
            writer.add_scalar("val_audio_retrieval top1", out["audio_retrieval"][0], state.global_epoch)
    return out


def train_loop(model, data_loaders, writer, hparams=None, state=None, visualizer=None, checkpoint_dir=None):
    """epochs x phases (train_whole_sync.py:32-145); returns (state, [per-epoch {phase: averages}])."""
    hparams = hparams if hparams is not None else model.hparams
    state = state if state is not None else LoopState()
    history = []
    while state.global_epoch < int(_opt(hparams, "nepochs", 1)):
        epoch = {}
        for phase, loader in data_loaders.items():
            epoch[phase] = run_phase(model, phase, loader, writer, hparams, state, visualizer, checkpoint_dir)
        history.append(epoch)
        state.global_epoch += 1
    return state, history


def run(model, data_loaders, writer, hparams=None, visualizer=None, resume_path=None, load_pretrain=False):
    """the `__main__` of train_whole_sync.py:148-187 as a function: optional resume / partial load, the loop, and a checkpoint on
    the way out whatever ends it (KeyboardInterrupt included)."""
    hparams = hparams if hparams is not None else model.hparams
    state = LoopState()
    ckpt_dir = _opt(hparams, "checkpoint_dir", ".")
    os.makedirs(ckpt_dir, exist_ok=True)
    if resume_path is not None and _opt(hparams, "resume", True):
        state = LoopState(*model.load_inpainting_checkpoint(resume_path, _opt(hparams, "reset_optimizer", False)))
    if load_pretrain:
        model.load_part_checkpoint()
    try:
        train_loop(model, data_loaders, writer, hparams, state, visualizer, ckpt_dir)
    except KeyboardInterrupt:
        pass
    finally:
        model.save_inpainting_checkpoint(state.global_step, state.global_test_step, ckpt_dir, state.global_epoch, hparams=hparams)
    return state
