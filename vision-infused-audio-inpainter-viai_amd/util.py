"""Counterparts of the reference's `utils/util.py` / `utils/model_util.py` helpers on either side of the path:
retrieval metrics on the GPU, tolerant state-dict copy, and the WaveNet checkpoint dict."""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .ops import _stream


def l2_ranks(clips_embed, captions_embed, return_dist=False):
    """Device part of L2retrieval (utils/util.py:99-109): ranks[i] = position of clip i when the clips are sorted by
    L2 distance to caption i, top1[i] = nearest clip.  int32 CUDA tensors."""
    lib = _lib.load()
    for t in (clips_embed, captions_embed):
        if not t.is_cuda:
            raise _lib.ViaiLibraryError("viai retrieval runs on the GPU only (got a %s tensor); no CPU fallback" % t.device)
    a, b = clips_embed.float().contiguous(), captions_embed.float().contiguous()
    n, d = a.shape
    m = b.shape[0]
    ranks = torch.empty(m, dtype=torch.int32, device=a.device)
    top1 = torch.empty(m, dtype=torch.int32, device=a.device)
    dist = torch.empty((m, n), dtype=torch.float32, device=a.device) if return_dist else None
    _lib.check(lib.viai_l2_ranks(a.data_ptr(), b.data_ptr(), n, m, d, ranks.data_ptr(), top1.data_ptr(),
                                 0 if dist is None else dist.data_ptr(), _stream()), "viai_l2_ranks")
    return (ranks, top1, dist) if return_dist else (ranks, top1)


def L2retrieval(clips_embed, captions_embed, return_ranks=False):
    """Same name, arguments and return tuple as utils/util.py:99-121: (r1, r5, r10, r50, medr, meanr)."""
    ranks_t, top1_t = l2_ranks(torch.as_tensor(clips_embed), torch.as_tensor(captions_embed))
    ranks, top1 = ranks_t.cpu().numpy().astype(np.int64), top1_t.cpu().numpy().astype(np.int64)
    r1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    r5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    r10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    r50 = 100.0 * len(np.where(ranks < 50)[0]) / len(ranks)
    medr = np.floor(np.median(ranks)) + 1
    meanr = ranks.mean() + 1
    if return_ranks:
        return (r1, r5, r10, r50, medr, meanr), (ranks, top1)
    return (r1, r5, r10, r50, medr, meanr)


def copy_state_dict(state_dict, model, strip=None):
    """utils/util.py:124-144 / utils/model_util.py:151-172: copy what matches by name and shape, report the rest."""
    tgt_state = model.state_dict()
    copied = set()
    for name, param in state_dict.items():
        if strip is not None and name.startswith(strip):
            name = name[len(strip):]
        if name not in tgt_state:
            continue
        if isinstance(param, torch.nn.Parameter):
            param = param.data
        if param.size() != tgt_state[name].size():
            print("mismatch:", name, param.size(), tgt_state[name].size())
            continue
        tgt_state[name].copy_(param)
        copied.add(name)
    missing = set(tgt_state.keys()) - copied
    if len(missing) > 0:
        print("missing keys in state_dict:", missing)
    return model


def save_checkpoint(model, optimizer, global_step, global_test_step, checkpoint_dir, epoch, name="wavenet", ema=None,
                    save_optimizer_state=True):
    """WaveNet checkpoint dict of utils/model_util.py:122-148: {"model", "optimizer", "global_step", "global_epoch",
    "global_test_step"}, plus the `_ema` twin whose parameters are the EMA shadows (clone_as_averaged_model :113-119)."""
    os.makedirs(checkpoint_dir, exist_ok=True)

    def cpu_sd(sd):
        return OrderedDict((k, v.detach().cpu().clone()) for k, v in sd.items())
    opt = optimizer.state_dict() if (optimizer is not None and save_optimizer_state) else None
    path = os.path.join(checkpoint_dir, name + "_checkpoint_step{:09d}.pth.tar".format(global_step))
    meta = {"optimizer": opt, "global_step": global_step, "global_epoch": epoch, "global_test_step": global_test_step}
    torch.save(dict(model=cpu_sd(model.state_dict()), **meta), path)
    paths = [path]
    if ema is not None:
        sd = cpu_sd(model.state_dict())
        for k, v in ema.shadow.items():
            if k in sd:
                sd[k] = v.detach().cpu().clone()
        path = os.path.join(checkpoint_dir, "checkpoint_step{:09d}_ema.pth".format(global_step))
        torch.save(dict(model=sd, **meta), path)
        paths.append(path)
    return paths


def load_checkpoint(path, model, optimizer=None, reset_optimizer=False, strip="module."):
    """utils/model_util.py:175-199; `strip` drops the DataParallel prefix the reference's checkpoints carry."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    copy_state_dict(ck["model"], model, strip=strip)
    if optimizer is not None and not reset_optimizer and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    return ck["global_step"], ck["global_epoch"], ck.get("global_test_step", 0)
