"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code at all; correctness is
"N replicas + mean-reduced gradients == one replica on the N-times batch, except
BatchNorm statistics, which stay per replica" (no SyncBN in the reference).
The exchange step is one all-reduce per optimizer over its flat gradient arena
(model.FlatArena), so there is nothing to pack or bucket by hand; the 1/N
scaling is folded into the Adam kernel's grad_scale.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # VIAI_DIST_BACKEND=gloo: CPU-staged collectives (tests, or several ranks sharing one GPU, which RCCL refuses)
            backend = os.environ.get("VIAI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            # eager communicator on the device this rank owns (device_id): RCCL's set-up happens here, not inside the first collective of the step
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_arena(flat: torch.Tensor, src=0, group=None):
    """identical initial parameters on every rank (DDP semantics)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            h = flat.detach().cpu()
            dist.broadcast(h, src=src, group=group)
            flat.copy_(h)
        else:
            dist.broadcast(flat, src=src, group=group)


def allreduce_mean_(flat_grad: torch.Tensor, group=None, scale_in_optimizer=False):
    """sum-all-reduce of a flat gradient arena; divides by world size unless the optimizer does."""
    if not dist.is_initialized():
        return flat_grad
    w = dist.get_world_size(group)
    if w == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    if not scale_in_optimizer:
        flat_grad.div_(w)
    return flat_grad


def all_reduce_sum_(t: torch.Tensor, group=None):
    """in-place sum-all-reduce of a (view of a) flat gradient arena on the CURRENT stream.
    RCCL ("nccl" backend): the collective is enqueued behind the current stream and the current stream waits for it -- the
    host does not block.  gloo (CPU tests, and the two-processes-on-one-GPU test): staged through host memory, blocking."""
    if not dist.is_initialized():
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_max_(t: torch.Tensor, group=None):
    """in-place max-all-reduce of a small tensor (the divergence flag of model.get_loss_items); gloo: staged through the host."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def barrier_max_ms(elapsed_ms: float, device=None) -> float:
    """max over ranks of a per-rank time (bench contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_ms
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([elapsed_ms], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
