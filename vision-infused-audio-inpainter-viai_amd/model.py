"""AudioModel — the G+D train step of the inpainting GAN on the HIP kernels.

The reference's `Models/Whole_Sync_inpainting_modify.AudioModel` is ABSENT from
the snapshot; its interface is pinned only by the call sites in
train_whole_sync.py:49-112,159-183 and utils/util.py:146-173.  This class
implements that interface; the step itself follows the pix2pix ordering the
reference credits (README.md:39) and is the build's declared spec:

    s_in  = s * mask
    fake  = Mel_Decoder(Mel_Encoder(s_in), s.size())
    D:  loss_D = 0.5*[GAN(netD(fake.detach()), False) + GAN(netD(s), True)] -> Adam(D)
    G:  loss_G = GAN(netD(fake), True) + lambda_L1 * L1(fake, s)            -> Adam(E, G)

Memory layout: all parameters of an optimizer live in ONE flat fp32 arena (and
their gradients / Adam moments in three more), so the optimizer is a single
streaming kernel and data-parallel gradient exchange is one RCCL all-reduce per
optimizer with no packing copies.  The whole step can be captured into HIP
graphs (three segments, split at the two all-reduce points).
"""
from __future__ import annotations

import ctypes
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from ._lib import check, load as lib
from .networks import (ImageEmbedding2, MelDecoder, MelDecoderImage, MelDiscriminator, MelEncoder, MultiScaleDiscriminator,
                       to_nchw_view)


class StepConfig:
    lr = 2e-4
    beta1 = 0.5
    beta2 = 0.999
    eps = 1e-8
    lambda_l1 = 100.0
    use_lsgan = False
    batch_size = 16
    cin_channels = 256          # mel bins F
    max_mel_lengths = 256       # frames T
    name = "viai"
    save_optimizer_state = True
    # BASELINE configs[2] / configs[3] (declared adaptations):
    use_video = False           # ResNet-18 ImageEmbedding2 fused into the G bottleneck through MelDecoderImage; when the
                                # bottleneck height h > 1 (F = 256) the video feature is TILED over h
    num_D = 1                   # > 1: multi-scale D = num_D MelDiscriminators on an avg-pool pyramid
    lambda_contrast = 0.0       # weight of L2ContrastiveLoss(audio bottleneck, video feature)  (`EmbeddingL2_item`)
    contrast_margin = 1.0


class FlatArena:
    """Moves the parameters of `modules` into one contiguous fp32 buffer (views
    keep every nn.Parameter usable as before) and gives them gradient views of a
    second buffer, so `.grad` accumulation lands in the arena."""

    ALIGN = 64

    def __init__(self, named_params):
        self.names, self.params, self.offsets = [], [], []
        off = 0
        for name, p in named_params:
            self.names.append(name)
            self.params.append(p)
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.size = off
        dev = self.params[0].device
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach in case something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FusedAdam:
    """torch.optim.Adam semantics (no amsgrad / weight decay) as one HIP kernel over a FlatArena.
    The step counter and bias corrections live on the device (graph-replayable)."""

    def __init__(self, arena: FlatArena, lr, betas, eps, stateless=()):
        """`stateless`: indices of parameters the forward never reaches (e.g. MelDecoder.convblock1.*): torch.optim.Adam keeps
        no state for a parameter whose .grad stays None, and neither does state_dict() here (their gradient is exactly zero, so
        the kernel leaves them and their moments untouched anyway)."""
        self.arena = arena
        self.stateless = frozenset(int(i) for i in stateless)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.state = torch.tensor([0.0, self.lr, 1.0, 1.0], dtype=torch.float64, device=arena.flat.device)

    def zero_grad(self):
        self.arena.zero_grad()

    def set_lr(self, lr):
        """writes the device-side rate on the CURRENT stream: callers that run the optimizer on another stream (AudioModel's deferred
        G update) order the write themselves -- use AudioModel.set_lr"""
        self.lr = float(lr)
        self.state[1] = self.lr

    def step(self, grad_scale=1.0):
        ops.adam_step(self.arena.flat, self.arena.grad, self.exp_avg, self.exp_avg_sq, self.state,
                      self.betas[0], self.betas[1], self.eps, grad_scale)
        ops.weights_changed(self.arena.params, owner=self)     # the kernel wrote the arena: re-pack every conv weight image in one launch

    # torch.optim.Adam-compatible (de)serialisation so utils/util.py:149-150 style checkpoints interchange
    def state_dict(self):
        step = float(self.state[0].item())
        st = {}
        if step > 0:                                      # torch creates the per-parameter state lazily, at the first step
            for i, (p, o) in enumerate(zip(self.arena.params, self.arena.offsets)):
                if i in self.stateless:
                    continue
                n = p.numel()
                st[i] = {"step": torch.tensor(step), "exp_avg": self.exp_avg[o:o + n].view(p.shape).detach().cpu().clone(),
                         "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).detach().cpu().clone()}
        # the param_group of the installed torch.optim.Adam (its key set differs between torch versions), with our values
        group = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=self.lr, betas=self.betas, eps=self.eps).state_dict()["param_groups"][0]
        group["params"] = list(range(len(self.arena.params)))
        return {"state": st, "param_groups": [group]}

    def load_state_dict(self, sd):
        step = 0.0
        for i, (p, o) in enumerate(zip(self.arena.params, self.arena.offsets)):
            s = sd["state"].get(i)
            if s is None:
                continue
            n = p.numel()
            self.exp_avg[o:o + n].copy_(s["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
            step = max(step, float(s["step"]))
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        self.state.copy_(torch.tensor([step, self.lr, self.betas[0] ** step, self.betas[1] ** step], dtype=torch.float64))


_PARKED_PLANS = []      # launch plans of models that were collected while a stream capture was in progress (see AudioModel.close)


def make_time_mask(batch, frames, blank_length, generator=None, device="cpu"):
    """One full-height time gap [t0, t0+L) per clip (misc/pipeline2.png); t0 ~ U{T/8 .. 5T/8}."""
    lo, hi = frames // 8, (5 * frames) // 8
    t0 = torch.randint(lo, hi + 1, (batch,), generator=generator)
    t0 = torch.clamp(t0, max=frames - blank_length)
    ar = torch.arange(frames)[None, :]
    m = ((ar < t0[:, None]) | (ar >= t0[:, None] + blank_length)).float()
    return m.view(batch, 1, 1, frames).to(device)


class AudioModel:
    """Interface inferred from train_whole_sync.py:49,50,68,75-85,91-112,145,159-183."""

    def __init__(self, hparams=None, device=None, process_group=None, use_graph=False, use_plan=False):
        self.hparams = hparams if hparams is not None else StepConfig()
        hp = self.hparams
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if self.device.type != "cuda":
            raise RuntimeError("AudioModel runs on an MI355X (HIP kernels); there is no CPU path")
        self.use_video = bool(getattr(hp, "use_video", False))
        self.num_D = int(getattr(hp, "num_D", 1))
        self.Mel_Encoder = MelEncoder(hp).to(self.device)
        self.Mel_Decoder = (MelDecoderImage(hp) if self.use_video else MelDecoder(hp)).to(self.device)
        self.netD = (MultiScaleDiscriminator(self.num_D) if self.num_D > 1 else MelDiscriminator()).to(self.device)
        self.VideoEncoder = ImageEmbedding2(hp).to(self.device) if self.use_video else None
        self.video = self.flow = None
        self.cfg = StepConfig()
        for k in ("lr", "beta1", "beta2", "eps", "lambda_l1", "use_lsgan", "lambda_contrast", "contrast_margin"):
            if hasattr(hp, k):
                setattr(self.cfg, k, getattr(hp, k))
        self._build_optimizers()
        self.pg = process_group
        self.world = 1
        self._force_allreduce = False      # tests: run the RCCL exchange even at world size 1 (identity)
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.train = 1
        self.update_wavenet = False
        self.blank_length = max(getattr(hp, "max_mel_lengths", 256) // 4, 1)
        self.current_lr = self.cfg.lr
        self.reconstruct_loss_item = 0.0
        self.EmbeddingL2_item = 0.0
        self.loss_mel_L1_item = 0.0
        self.mel_net_norm = None
        self.video_net_norm = None
        self.losses = torch.zeros(6, device=self.device)      # loss_D, loss_G, loss_G_GAN, loss_L1, loss_D_real, EmbeddingL2
        self.mel = self.mask = self.fake = None
        self._wmax = None
        # use_plan: the step is stream-captured once and replayed from C as a launch plan (csrc/plan.hip): the eager step's
        # kernels, arguments, streams and cross-stream edges without the per-launch host work.  Shares the segment structure
        # (and the static-buffer discipline) of graph mode, so it sets use_graph too.
        self.use_plan = bool(use_plan)
        self.use_graph = bool(use_graph) or self.use_plan
        self._plans = None
        self._graph_scratch = None
        self._consts = {}
        # weight gradients trail on a side stream (ops.WGRAD_STREAM) in eager mode: -3.5 % step time on one MI355X.  Inside
        # a captured hipGraph the fork/join edges cost more than the overlap returns (+2 %), so graph mode stays on one
        # stream.  VIAI_WGRAD_STREAM=0/1 overrides.
        # With the visual branch the two ResNets already run as two chains on two streams (networks.ImageEmbedding2), each carrying its
        # own weight gradients; funnelling all of them through one trailing stream costs more than it overlaps (120.8 vs 124.1 ms).
        one_stream = self.use_graph and not self.use_plan
        from . import networks as _nets
        # (eager mode only: inside a capture ImageEmbedding2 does not fork its flow stream, so plan / graph mode keeps the trailing stream)
        two_chains = self.use_video and _nets.FLOW_STREAM and not self.use_graph
        side = os.environ.get("VIAI_WGRAD_STREAM", "0" if (one_stream or two_chains) else "1") != "0"
        self._wgrad_stream = torch.cuda.Stream(device=self.device) if side else None
        dreal = os.environ.get("VIAI_DREAL_STREAM", "0" if one_stream else "1") != "0"
        self._dreal_stream = torch.cuda.Stream(device=self.device) if dreal else None
        self._graphs = None
        self._comm = None                  # data-parallel exchange stream (created on first use)
        self._g_update_pending = False     # the G exchange + Adam(G) + re-pack of the previous step are still on _comm
        self._skip_exchange = False        # bench.py: measure the step without the collectives (comm_ms_exposed)
        self._plan_exchange()

    # ------------------------------------------------------------------ setup
    def _build_optimizers(self):
        c = self.cfg
        g_named = [("E." + n, p) for n, p in self.Mel_Encoder.named_parameters()] + \
                  [("G." + n, p) for n, p in self.Mel_Decoder.named_parameters()]
        if self.VideoEncoder is not None:
            g_named += [("V." + n, p) for n, p in self.VideoEncoder.named_parameters()]
        d_named = [("D." + n, p) for n, p in self.netD.named_parameters()]
        self.arena_G = FlatArena(g_named)
        self.arena_D = FlatArena(d_named)
        dead = tuple("G." + p for p in getattr(self.Mel_Decoder, "UNUSED_PREFIXES", ())) + \
            tuple("V." + p for p in getattr(self.VideoEncoder, "UNUSED_PREFIXES", ()))
        stateless = [i for i, n in enumerate(self.arena_G.names) if n.startswith(dead)] if dead else []
        self.optimizer_G = FusedAdam(self.arena_G, c.lr, (c.beta1, c.beta2), c.eps, stateless)
        self.optimizer_D = FusedAdam(self.arena_D, c.lr, (c.beta1, c.beta2), c.eps)

    def _plan_exchange(self):
        """Gradient buckets of the data-parallel exchange: contiguous ranges of the flat gradient arenas in the order the
        backward pass completes them.  An EARLY bucket (trigger parameter, lo, hi) goes to RCCL from a gradient-ready hook
        (ops.GRAD_HOOKS) as soon as the layer owning the trigger has queued its gradients -- the rest of the backward chain
        runs beside the collective; the LATE ranges follow after the last weight gradient.
          D  (6.2 MB):  early [conv3.weight .. end) = conv3 + norm3 + conv4, 76 % of the bytes, ready after the second layer of
                        the backward chain;  late [0 .. conv3.weight)
          G (16.7 MB):  early [convblock3 .. end of G), then [G start .. convblock3) at deconv1_1;  late = E (+ the visual branch).
        The G exchange, Adam(E,G) and the weight re-pack stay on the exchange stream and overlap the NEXT step's D(real) forward +
        backward, which reads none of them (see _seg_forward_dstep)."""
        aD, aG = self.arena_D, self.arena_G

        def off(arena, name):
            return arena.offsets[arena.names.index(name)]
        self._early_D, self._late_D = [], [(0, aD.size)]
        if self.num_D == 1:
            o = off(aD, "D.conv3.weight")
            self._early_D, self._late_D = [("D.conv3.weight", o, aD.size)], [(0, o)]
        g_lo = off(aG, next(n for n in aG.names if n.startswith("G.")))
        v_names = [n for n in aG.names if n.startswith("V.")]
        g_hi = off(aG, v_names[0]) if v_names else aG.size
        c3 = off(aG, "G.convblock3.conv3_0.weight")
        head = "G.deconv1_1_1.weight" if self.use_video else "G.deconv1_1.weight"
        self._early_G = [("G.convblock3.conv3_0.weight", c3, g_hi), (head, g_lo, c3)]
        self._late_G = [(0, g_lo)] + ([(g_hi, aG.size)] if g_hi < aG.size else [])
        # An InstanceNorm2d layer runs conv_bn_act once PER SAMPLE with the same weight (networks._instance_norm_layer): the trigger
        # layer's backward is then invoked B times per pass and a hook armed for one invocation would hand the bucket to RCCL while the
        # other samples are still accumulating into it.  Such models exchange each arena whole, after its last weight gradient.
        def has_instance_norm(*mods):
            return any(isinstance(m, nn.InstanceNorm2d) for mod in mods if mod is not None for m in mod.modules())
        if has_instance_norm(self.netD):
            self._early_D, self._late_D = [], [(0, aD.size)]
        if has_instance_norm(self.Mel_Encoder, self.Mel_Decoder, self.VideoEncoder):
            self._early_G, self._late_G = [], [(0, aG.size)]

    def _exchanging(self):
        return (self.world > 1 or self._force_allreduce) and not self._skip_exchange

    def _overlapped(self):
        """bucketed exchange from inside the step (eager mode); graph mode exchanges between its captured segments instead"""
        return self._exchanging() and not self.use_graph

    def _comm_stream(self):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=self.device)
        return self._comm

    def _reduce_range(self, arena, lo, hi):
        """sum-all-reduce of arena.grad[lo:hi] on the exchange stream, behind everything queued so far on the current stream
        and on the weight-gradient stream (in-place on the arena: no packing copies)."""
        from . import ddp
        comm = self._comm_stream()
        comm.wait_stream(torch.cuda.current_stream())
        if ops.WGRAD_STREAM is not None:
            comm.wait_stream(ops.WGRAD_STREAM)
        with torch.cuda.stream(comm):
            ddp.all_reduce_sum_(arena.grad[lo:hi], self.pg)

    def _arm_hooks(self, arena, early, fire_at=1):
        """register the early buckets of `arena` as gradient-ready hooks; `fire_at` = which invocation of the trigger layer's
        backward completes its gradient (2 when both halves of loss_D are back-propagated in one pass)."""
        ops.GRAD_HOOKS.clear()
        if not self._overlapped():
            return
        for name, lo, hi in early:
            p = arena.params[arena.names.index(name)]
            state = {"n": fire_at}

            def hook(state=state, lo=lo, hi=hi):
                state["n"] -= 1
                if state["n"] == 0:
                    self._reduce_range(arena, lo, hi)
            ops.GRAD_HOOKS[p.data_ptr()] = hook

    def _finish_exchange(self, arena, late):
        """after the backward pass: the late ranges, then hand the reduced gradients back to the main stream."""
        ops.GRAD_HOOKS.clear()
        if not self._overlapped():
            return False
        for lo, hi in late:
            self._reduce_range(arena, lo, hi)
        return True

    def sync_pending_update(self):
        """the main stream waits for the deferred G exchange + Adam(E,G) + re-pack of the previous step (no host sync).  Called at
        the point of the next step that first touches E / G, and by everything that reads parameters between steps."""
        if self._g_update_pending:
            torch.cuda.current_stream().wait_stream(self._comm)
            self._g_update_pending = False

    def load_states(self, E=None, G=None, D=None, V=None):
        """load state_dicts (e.g. the oracle's closed-form tables) without breaking the arenas."""
        if V is not None and self.VideoEncoder is None:
            raise ValueError("a VideoEncoder state was given but the model was built without use_video")
        self.sync_pending_update()
        for mod, sd in ((self.Mel_Encoder, E), (self.Mel_Decoder, G), (self.netD, D), (self.VideoEncoder, V)):
            if sd is None:
                continue
            own = mod.state_dict()
            for k, v in sd.items():
                own[k].copy_(v.to(own[k].device))
        self.weights_changed()

    def weights_changed(self):
        self.sync_pending_update()
        self._weights_changed()

    def set_lr(self, lr, which=("G", "D")):
        """learning rate of one or both optimizers (viai_amd.lrschedule.apply goes through here when handed the model).  The rate lives
        in the device-side Adam state; the deferred Adam(E,G) of the previous step may still be reading it on the exchange stream,
        so the write is ordered behind it."""
        self.sync_pending_update()
        if "G" in which:
            self.optimizer_G.set_lr(lr)
        if "D" in which:
            self.optimizer_D.set_lr(lr)
        self.current_lr = float(lr)
        return self.current_lr

    def _weights_changed(self):
        """Parameters were written outside the optimizers (checkpoint load, manual surgery): re-pack the conv weight
        images now.  Eager steps would notice through the tensor versions; a captured graph contains no per-layer pack
        launches (the optimizers' batched re-pack is what it replays), so it relies on this call."""
        ops.weights_changed(self.arena_G.params, owner=self.optimizer_G)
        ops.weights_changed(self.arena_D.params, owner=self.optimizer_D)

    # ------------------------------------------------------------- step pieces
    def get_blank_space_length(self, global_step):
        """mask-length curriculum hook (train_whole_sync.py:49); the policy is unpinned -> fixed T/4."""
        T = self.mel.shape[-1] if self.mel is not None else getattr(self.hparams, "max_mel_lengths", 256)
        self.blank_length = max(T // 4, 1)
        return self.blank_length

    def set_inputs(self, data, mask=None, video=None, flow=None):
        """data: the loader's 8-tuple (audio_loader.py:532: video, flow, c (B,C,T), x, y, g, lengths, paths) or a mel
        tensor (B,F,T)/(B,1,F,T) in [0,1].  Copies into static device buffers (graph-replay safe)."""
        if isinstance(data, (tuple, list)):
            video = data[0] if video is None else video
            flow = data[1] if flow is None else flow
        if self.use_video:
            if video is None or flow is None:
                raise ValueError("use_video: set_inputs needs the video (B,N,3,224,224) and flow (B,N,2,224,224) blocks")
            video = video.to(self.device, dtype=torch.float32)
            flow = flow.to(self.device, dtype=torch.float32)
            if self.video is None or self.video.shape != video.shape:
                self.video, self.flow = torch.empty_like(video), torch.empty_like(flow)
                self._drop_graphs()
            self.video.copy_(video)
            self.flow.copy_(flow)
        mel = data[2] if isinstance(data, (tuple, list)) else data
        if mel.dim() == 3:
            mel = mel.unsqueeze(1)
        mel = mel.to(self.device, dtype=torch.float32, non_blocking=True)
        B, _, F, T = mel.shape
        if mask is None:
            mask = make_time_mask(B, T, min(self.blank_length, T), device=self.device)
        mask = mask.to(self.device, dtype=torch.float32).reshape(B, 1, 1, T)
        if self.mel is None or self.mel.shape != mel.shape:
            self.mel = torch.empty_like(mel)
            self.mask = torch.empty_like(mask)
            self._drop_graphs()
        self.mel.copy_(mel)
        self.mask.copy_(mask)

    def _drop_graphs(self):
        """new input shape: the captured graphs are stale.  They may still be replaying (steps are asynchronous) and a hipGraphExec
        must not be destroyed while in flight, so this is one of the few places that waits for the device."""
        if self._graphs is not None:
            torch.cuda.synchronize(self.device)
            if self._plans is not None:
                for p in self._plans:
                    lib().viai_plan_destroy(p)
                self._plans = None
            self._graphs = None
            self._graph_scratch = None
            ops.drop_scratch()            # scratch buffers allocated while capturing live in the dead graphs' private memory pool

    def close(self):
        """release what the launch plans hold (events, plan nodes).  Also runs from __del__, i.e. whenever the garbage collector gets
        to a dropped model -- possibly in the middle of ANOTHER model's stream capture, where a device synchronisation would invalidate
        that capture: plans are then parked and destroyed at the next close() outside a capture.  The process-wide scratch pool is
        not touched from here (buffers a capture of THIS model allocated live in self._graph_scratch and go with the object)."""
        plans, self._plans = (getattr(self, "_plans", None) or []), None      # __del__ of a half-built object: __init__ may have raised early
        self._graphs = None
        self._graph_scratch = None
        if not plans and not _PARKED_PLANS:
            return
        try:
            if torch.cuda.is_current_stream_capturing():
                _PARKED_PLANS.extend(plans)
                return
            torch.cuda.synchronize(self.device)         # a plan's events / kernel nodes must not be destroyed while a replay is in flight
            for p in plans + _PARKED_PLANS:
                lib().viai_plan_destroy(p)
            del _PARKED_PLANS[:]
        except Exception:
            pass                                        # interpreter shutdown: the driver is gone, nothing left to release

    def __del__(self):
        self.close()

    def _gan(self, pred, real):
        t = 1.0 if real else 0.0
        if isinstance(pred, (list, tuple)):                  # multi-scale D: mean of the per-scale losses
            tot = None
            for p in pred:
                l = ops.mse_mean(p, t) if self.cfg.use_lsgan else ops.bce_mean(p, t)
                tot = l if tot is None else tot + l
            return tot / float(len(pred))
        return ops.mse_mean(pred, t) if self.cfg.use_lsgan else ops.bce_mean(pred, t)

    def _generate(self, s_nhwc, want_feats=False):
        """E (+ E_v) + G forward on NHWC; returns fake (B,F,T,1) and the contrastive term (or None)
        [+ the encoder maps and the video feature (B,256,1,T/16) when `want_feats`]."""
        B, F, T, _ = s_nhwc.shape
        feats = self.Mel_Encoder.forward_nhwc(s_nhwc.view(B, F, T), mask=self.mask)       # s_in = s * mask, where E.conv1 loads s
        if not self.use_video:
            fake = self.Mel_Decoder.forward_nhwc(feats, (F, T))
            return (fake, None, feats, None) if want_feats else (fake, None)
        f_v, _fea = self.VideoEncoder(self.video, self.flow)                 # (B,256,1,N/4), (B,512,N)
        h, w = feats[-1].shape[1], feats[-1].shape[2]
        if f_v.shape[3] != w:
            raise ValueError("use_video: need N = T/4 frames per clip (video steps %d != bottleneck steps %d)" % (f_v.shape[3], w))
        fv_nhwc = f_v.permute(0, 2, 3, 1)                                     # (B,1,w,256)
        head = self.Mel_Decoder._head_av(feats, fv_nhwc.expand(B, h, w, 256).permute(0, 3, 1, 2))   # tiled over h (declared adaptation)
        fake = self.Mel_Decoder.forward_nhwc(feats, (F, T), head=head)
        lc = None
        if self.cfg.lambda_contrast > 0:
            f_a = feats[-1].mean(dim=1).reshape(B * w, 256)
            lc = ops.l2_contrastive(f_a.contiguous(), fv_nhwc.reshape(B * w, 256).contiguous(), self.cfg.contrast_margin, False)
        return (fake, lc, feats, f_v) if want_feats else (fake, lc)

    def _const(self, v):
        """device scalar with a fixed value (gradient seeds), made once"""
        v = float(v)
        t = self._consts.get(v)
        if t is None:
            t = self._consts[v] = torch.full((), v, device=self.device, dtype=torch.float32)
        return t

    def _put(self, i, value):
        """losses[i] = value through an elementwise kernel: a `copy_` is a device-to-device memcpy, whose node parameters a
        stream capture does not give back (plan mode reads the captured launches)."""
        torch.mul(value.detach().reshape(1), 1.0, out=self.losses[i:i + 1])

    def _seg_forward_dstep(self):
        s = self.mel
        B, _, F, T = s.shape
        s_nhwc = s.view(B, F, T, 1)
        ops.begin_step(self.device)                    # re-arm the abs-max slots of the f16x2 backward kernels (one fill)
        self.optimizer_D.zero_grad()
        self.netD.requires_grad_(True)
        main = torch.cuda.current_stream()
        side = self._dreal_stream
        # D on the real clip needs nothing from the generator: in eager mode its forward AND backward run on a second
        # stream next to the E+G forward (whose BatchNorm / streaming kernels leave the matrix pipes idle, and vice
        # versa).  loss_D = 0.5 * (loss_fake + loss_real) is back-propagated as two halves; each parameter gradient is
        # the sum of the two contributions in either order (two addends: bitwise commutative).  The declared order of
        # the BatchNorm running-statistics updates is real first, then fake.
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pred_real = self.netD.forward_nhwc(s_nhwc)
                fwd_done = torch.cuda.Event()
                fwd_done.record(side)
                loss_real = self._gan(pred_real, True)
                loss_real.backward(self._const(0.5))                   # d(0.5 L): the factor goes in as the seed, no elementwise launches
        else:
            pred_real = self.netD.forward_nhwc(s_nhwc)
            loss_real = self._gan(pred_real, True)
        # everything above reads D only: it ran beside the previous step's G exchange + Adam(E,G) + re-pack (data parallel);
        # E / G and their gradient arena are touched from here on
        self.sync_pending_update()
        self.optimizer_G.zero_grad()
        fake, self._lc = self._generate(s_nhwc)                        # (B,F,T,1)
        self._fake = fake
        self.fake = to_nchw_view(fake)
        if side is not None:
            main.wait_event(fwd_done)                                  # running statistics: real before fake
        pred_fake = self.netD.forward_nhwc(fake.detach())
        loss_fake = self._gan(pred_fake, False)
        if side is not None:
            main.wait_stream(side)                                     # real-branch gradients are in the arena
            loss_real.record_stream(main)                              # allocated on `side`, read below on main
            self._arm_hooks(self.arena_D, self._early_D, 1)
            loss_fake.backward(self._const(0.5))
        else:
            self._arm_hooks(self.arena_D, self._early_D, 2)
            torch.autograd.backward([loss_fake, loss_real], [self._const(0.5), self._const(0.5)])      # d(0.5 (fake + real)): the factor as seeds
        ops.join_wgrad()
        if self._finish_exchange(self.arena_D, self._late_D):
            main.wait_stream(self._comm)                               # Adam(D) and the G step need the reduced D gradients
        self._loss_real, self._loss_fake = loss_real.detach(), loss_fake.detach()      # -> losses[0], losses[4] at the end of the step (one launch)

    def _seg_dupdate_gstep(self, update=True):
        if update:
            self.optimizer_D.step(1.0 / self.world)
        s = self.mel
        B, _, F, T = s.shape
        self.netD.requires_grad_(False)
        pred = self.netD.forward_nhwc(self._fake)
        loss_gan = self._gan(pred, True)
        loss_l1 = ops.l1_mean(self._fake, s.view(B, F, T, 1))
        # loss_G = gan + lambda_l1 * l1 (+ lambda_c * contrast) is back-propagated from its terms with the weights as seeds: the same
        # gradients bit for bit, without the scalar mul / add / ones_like launches between D's forward and the backward chain
        roots, seeds = [loss_gan, loss_l1], [self._const(1.0), self._const(self.cfg.lambda_l1)]
        if self._lc is not None:
            roots.append(self._lc)
            seeds.append(self._const(self.cfg.lambda_contrast))
        self._arm_hooks(self.arena_G, self._early_G, 1)
        torch.autograd.backward(roots, seeds)
        ops.join_side_streams()                   # (the visual branch's flow network back-propagates on its own stream)
        if self._lc is not None:
            self.EmbeddingL2 = self._lc.detach()
        ops.join_wgrad()
        self._g_exchanged = self._finish_exchange(self.arena_G, self._late_G)
        self.netD.requires_grad_(True)
        # the six scalars get_loss_items reports, in one launch
        check(lib().viai_step_scalars(self._loss_real.data_ptr(), self._loss_fake.data_ptr(), loss_gan.data_ptr(), loss_l1.data_ptr(),
                                      self._lc.data_ptr() if self._lc is not None else 0, float(self.cfg.lambda_l1), float(self.cfg.lambda_contrast),
                                      self.losses.data_ptr(), torch.cuda.current_stream().cuda_stream), "viai_step_scalars")
        self._scalars_src = (self._loss_real, self._loss_fake, loss_gan, loss_l1, self._lc)       # alive until the launch has run
        self._pred_fake_g = pred

    def _seg_gupdate(self):
        if getattr(self, "_g_exchanged", False) and not self.use_graph:
            # Adam(E,G) + the batched weight re-pack stay on the exchange stream behind the collective; the main stream picks them
            # up at the first use of E / G in the next step (sync_pending_update)
            with torch.cuda.stream(self._comm):
                self.optimizer_G.step(1.0 / self.world)
            self._g_update_pending = True
            self._g_exchanged = False
        else:
            self.optimizer_G.step(1.0 / self.world)

    def _allreduce(self, arena):
        """graph mode only: blocking whole-arena exchange between the captured segments."""
        if self._exchanging():
            from . import ddp
            ddp.all_reduce_sum_(arena.grad, self.pg)

    def _capture(self):
        """three HIP graphs split at the two gradient all-reduce points."""
        segs = (self._seg_forward_dstep, self._seg_dupdate_gstep, self._seg_gupdate)
        # Warm-up on a side stream (allocator + lazy init).  The warm-up runs two REAL steps (both Adam updates, the
        # BatchNorm running statistics) without any gradient exchange, so everything a step mutates is snapshotted first
        # and put back afterwards: a (re-)capture -- first step, or a new input shape -- then leaves parameters, Adam
        # moments / step counters and BatchNorm buffers exactly where they were, on every rank.
        snap = self._mutable_state()
        saved = [t.clone() for t in snap]
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):
                for f in segs:
                    f()
        torch.cuda.current_stream().wait_stream(st)
        for t, v in zip(snap, saved):
            t.copy_(v)
        self.weights_changed()
        graphs, plans = [], []
        pool = None
        before = ops.scratch_snapshot()
        # torch hands out streams from a pool of 32 per device, round robin, and torch.cuda.graph's default capture stream is one pool
        # stream made once per process: in a long-lived process it can be the very stream this model got as a side stream, and the
        # captured branch would collapse into the origin.  Capture on a stream known to differ from both side streams.
        taken = {st.cuda_stream for st in (self._wgrad_stream, self._dreal_stream) if st is not None}
        cap = torch.cuda.Stream(device=self.device)
        while cap.cuda_stream in taken:
            cap = torch.cuda.Stream(device=self.device)
        # With a process group up, ProcessGroupNCCL's watchdog THREAD polls its work events (hipEventQuery) whenever it likes; under the default
        # capture mode ("global") that call is illegal from any thread while this one captures, the watchdog dies with
        # hipErrorStreamCaptureUnsupported and takes the process with it -- the "RCCL abort now and then" of rounds 2 and 3 (1 of 20 runs of the
        # one-rank test; it was never RCCL's set-up or teardown).  "thread_local" confines the check to the capturing thread.
        dist_up = torch.distributed.is_available() and torch.distributed.is_initialized()
        gkw = {"capture_error_mode": "thread_local"} if dist_up else {}
        for f in segs:
            if self.use_plan:
                # the capture is a recorder: the hipGraph is kept (it owns the kernel-argument arrays) but never instantiated
                g = torch.cuda.CUDAGraph(keep_graph=True)
                check(lib().viai_plan_log_begin(), "viai_plan_log_begin")
                try:
                    with torch.cuda.graph(g, pool=pool, stream=cap, **gkw):
                        origin = torch.cuda.current_stream().cuda_stream
                        f()
                finally:
                    lib().viai_plan_log_end()
                plan = ctypes.c_void_p()
                check(lib().viai_plan_build(int(g.raw_cuda_graph()), origin, ctypes.byref(plan)), "viai_plan_build")
                plans.append(plan)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=cap, **gkw):
                    f()
            pool = g.pool()
            graphs.append(g)
        self._graphs = graphs
        self._plans = plans if self.use_plan else None
        # scratch buffers first requested inside the captures came from the graphs' private pool: they belong to this model now
        self._graph_scratch = ops.scratch_take_new(before)

    def plan_info(self):
        """per captured segment: nodes, kernels, kernels with a noted stream, copies, fills, streams, events, waits"""
        out = []
        for p in self._plans or ():
            v = (ctypes.c_int * 8)()
            check(lib().viai_plan_info(p, v, 8), "viai_plan_info")
            out.append(list(v))
        return out

    def _mutable_state(self):
        """every device tensor a train step writes and the next step reads: parameter arenas, Adam moments and step
        state, BatchNorm buffers."""
        ts = []
        for opt in (self.optimizer_G, self.optimizer_D):
            ts += [opt.arena.flat, opt.exp_avg, opt.exp_avg_sq, opt.state]
        for mod in (self.Mel_Encoder, self.Mel_Decoder, self.netD, self.VideoEncoder):
            if mod is not None:
                ts += [b for b in mod.buffers()]
        return ts

    def optimize_parameters(self, global_step=0):
        """one G+D train step (train_whole_sync.py:76)."""
        prev, ops.DIRECT_GRAD = ops.DIRECT_GRAD, True      # gradients land in the arenas in place
        prev_s, ops.WGRAD_STREAM = ops.WGRAD_STREAM, self._wgrad_stream
        try:
            self._optimize_parameters()
            self._stepped_since_range_check = True
        finally:
            ops.DIRECT_GRAD, ops.WGRAD_STREAM = prev, prev_s

    def _optimize_parameters(self):
        if self.use_graph:
            if self._graphs is None:
                self._capture()
            if self.use_plan:
                run = [lambda p=p: check(lib().viai_plan_replay(p, torch.cuda.current_stream().cuda_stream), "viai_plan_replay")
                       for p in self._plans]
            else:
                run = [g.replay for g in self._graphs]
            run[0]()
            self._allreduce(self.arena_D)
            run[1]()
            self._allreduce(self.arena_G)
            run[2]()
        else:
            self._seg_forward_dstep()          # D exchange: early bucket from inside the backward, the rest at its end
            self._seg_dupdate_gstep()          # G exchange: two early buckets + the encoder range
            self._seg_gupdate()

    def forward_backward_no_update(self):
        """the step WITHOUT the two Adam updates (parity target, see oracle.step_no_update).  In a process group the gradient
        arenas hold the all-reduced SUMS afterwards (the 1/N lives in the Adam kernel)."""
        prev, ops.DIRECT_GRAD = ops.DIRECT_GRAD, True
        prev_s, ops.WGRAD_STREAM = ops.WGRAD_STREAM, self._wgrad_stream
        try:
            self._seg_forward_dstep()
            self._seg_dupdate_gstep(update=False)
            if getattr(self, "_g_exchanged", False):
                torch.cuda.current_stream().wait_stream(self._comm)
                self._g_exchanged = False
        finally:
            ops.DIRECT_GRAD, ops.WGRAD_STREAM = prev, prev_s
            ops.GRAD_HOOKS.clear()

    def test(self):
        """forward only (train_whole_sync.py:79-80,159-183: the caller sets `model.train = 0` and wraps in no_grad).
        Runs the SAME generator path as the train step (`_generate`: with `use_video` the video feature goes through
        `deconv1_1_1`).  With `self.train == 0` the modules are put in eval mode for the call (running BatchNorm
        statistics, buffers untouched) and restored afterwards."""
        self.sync_pending_update()
        s = self.mel
        B, _, F, T = s.shape
        mods = [m for m in (self.Mel_Encoder, self.Mel_Decoder, self.VideoEncoder) if m is not None]
        was = [m.training for m in mods]
        if not self.train:
            for m in mods:
                m.eval()
        try:
            with torch.no_grad():
                fake, _lc, feats, f_v = self._generate(s.view(B, F, T, 1), want_feats=True)
                self.fake = to_nchw_view(fake)
                self._put(3, ops.l1_mean(fake, s.view(B, F, T, 1)))
                emb = feats[-1].mean(dim=(1, 2))                   # bottleneck (B, h, T/16, 256) -> (B, 256)
                self.mel_net_norm = torch.nn.functional.normalize(emb, p=2, dim=1)
                # the retrieval metrics of the reference loop (utils/util.py:99-121) pair the audio embedding with the VIDEO
                # embedding, and the loop reads `video_net_norm` unconditionally after test() (train_whole_sync.py:82-83:
                # util.to_np(model.video_net_norm)), so it is always a tensor.  Without a visual branch there is nothing to pair the
                # audio embedding with: the attribute is a zero tensor of the same shape and the retrieval numbers computed from it
                # carry no information (every "video" is equidistant from every clip).
                self.video_net_norm = (torch.nn.functional.normalize(f_v.mean(dim=(2, 3)), p=2, dim=1)
                                       if f_v is not None else torch.zeros_like(self.mel_net_norm))
        finally:
            for m, t in zip(mods, was):
                m.train(t)
        return self.fake

    # ------------------------------------------------------------ bookkeeping
    def _weight_range_check(self):
        """max |w| over both parameter arenas next to the loss scalars (same host read): the f16x2 weight images are pre-scaled by a
        static 256 and clamp beyond |w| = 255.9 -- a model that gets there has diverged, and is told so instead of training on clipped
        weights."""
        if self._wmax is None:
            self._wmax = torch.zeros(2, device=self.device)
        self._wmax.zero_()
        for i, arena in enumerate((self.arena_G, self.arena_D)):
            check(lib().viai_absmax(arena.flat.data_ptr(), arena.flat.numel(), self._wmax[i:i + 1].data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), "viai_absmax")
        return self._wmax

    def get_loss_items(self):
        """host sync point (train_whole_sync.py:85).  COLLECTIVE in a data-parallel run, but only when a train step ran since the last call: train steps
        are collective themselves, so every rank gets here the same number of times; calls in the no-grad test phase (train_whole_sync.py:79-80 reads the
        losses there too, and ranks may hold unequal numbers of validation batches) and repeated reads on one rank do not communicate."""
        self.sync_pending_update()
        wr = self._weight_range_check()
        stepped, self._stepped_since_range_check = getattr(self, "_stepped_since_range_check", False), False
        if self._exchanging() and stepped:
            # data-parallel runs: the weights are replicas, so every rank should see the same maxima -- but a rank whose replica has gone bad on its own (a
            # corrupted exchange, a NaN that only its clips produce) must not raise alone and leave the others waiting in the next collective: the flag is
            # the MAX over the ranks (NaN propagates through max), one tiny all-reduce at a point where the host synchronises anyway
            from . import ddp
            wr = ddp.all_reduce_max_(torch.nan_to_num(wr, nan=float("inf"), posinf=float("inf")), self.pg)
        both = torch.cat((self.losses, wr)).tolist()
        v, wmax = both[:self.losses.numel()], both[self.losses.numel():]
        if max(wmax) > ops.F16_WEIGHT_LIMIT and os.environ.get("VIAI_F16X2", "1") != "0" and os.environ.get("VIAI_MATH", "") != "fp32":
            raise FloatingPointError("max |weight| = %.4g (E+G) / %.4g (D) is beyond %.1f, where the f16x2 weight images of the conv kernels "
                                     "clamp: the model has diverged (VIAI_F16X2=0 selects the bf16x3 kernels, which have the fp32 exponent range)"
                                     % (wmax[0], wmax[1], ops.F16_WEIGHT_LIMIT))
        self.loss_D_item, self.loss_G_item, self.loss_G_GAN_item, self.loss_mel_L1_item = v[0], v[1], v[2], v[3]
        self.reconstruct_loss_item = v[3]                      # the L1 reconstruction term (train_whole_sync.py:98 accumulates it)
        self.EmbeddingL2_item = v[5] if (self.use_video and self.cfg.lambda_contrast > 0) else 0.0
        return v

    def get_current_errors(self):
        return OrderedDict([("D", self.loss_D_item), ("G", self.loss_G_item), ("G_GAN", self.loss_G_GAN_item),
                            ("mel_L1", self.loss_mel_L1_item)])

    def get_current_visuals(self):
        out = OrderedDict()
        if self.mel is not None:
            out["real_mel"] = self.mel.detach()
            out["masked_mel"] = (self.mel * self.mask).detach()
        if self.fake is not None:
            out["fake_mel"] = self.fake.detach()
        return out

    def TF_writer(self, writer, step=0):
        if writer is None:
            return
        for k, v in self.get_current_errors().items():
            writer.add_scalar(getattr(self.hparams, "name", "viai") + "_" + k, v, step)

    def del_no_need(self):
        self._pred_fake_g = None

    def eval_model_test(self, global_step, eval_dir):
        self.test()
        os.makedirs(eval_dir, exist_ok=True)
        torch.save({"fake": self.fake.cpu(), "real": self.mel.cpu(), "mask": self.mask.cpu()},
                   os.path.join(eval_dir, "step%09d_mel.pt" % global_step))

    # ---------------------------------------------------------- checkpointing
    def save_inpainting_checkpoint(self, global_step, global_test_step, checkpoint_dir, epoch, hparams=None):
        """same dict layout as utils/util.py:146-162."""
        hp = hparams if hparams is not None else self.hparams
        self.sync_pending_update()
        path = os.path.join(checkpoint_dir, getattr(hp, "name", "viai") + "_checkpoint_step{:09d}.pth.tar".format(global_step))
        if self.world > 1 and torch.distributed.get_rank(self.pg) != 0:
            return path                                     # replicas are identical: rank 0 writes, the others do not race it
        os.makedirs(checkpoint_dir, exist_ok=True)
        keep = getattr(hp, "save_optimizer_state", True)

        def cpu_sd(m):
            return OrderedDict((k, v.detach().cpu().clone()) for k, v in m.state_dict().items())
        ck = OrderedDict([
            ("Mel_Encoder", cpu_sd(self.Mel_Encoder)), ("Mel_Decoder", cpu_sd(self.Mel_Decoder)), ("netD", cpu_sd(self.netD)),
            ("optimizer_G", self.optimizer_G.state_dict() if keep else None),
            ("optimizer_D", self.optimizer_D.state_dict() if keep else None),
            ("global_step", global_step), ("global_epoch", epoch), ("global_test_step", global_test_step),
        ])
        if self.VideoEncoder is not None:
            # utils/util.py:148 has this entry commented out -- there the visual branch was not in this optimizer; here it is
            # trained by optimizer_G (whose state dict indexes its parameters too), so a resume needs its weights and buffers
            ck["VideoEncoder"] = cpu_sd(self.VideoEncoder)
        torch.save(dict(ck), path)          # a plain dict, as utils/util.py:150 writes
        return path

    def load_inpainting_checkpoint(self, path, reset_optimizer=False):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if self.VideoEncoder is not None and "VideoEncoder" not in ck and not reset_optimizer:
            raise KeyError("checkpoint has no 'VideoEncoder' entry: optimizer_G's moments would not belong to the freshly "
                           "initialised visual branch (pass reset_optimizer=True to load E/G/D only)")
        self.load_states(ck["Mel_Encoder"], ck["Mel_Decoder"], ck["netD"], ck.get("VideoEncoder") if self.VideoEncoder is not None else None)
        if not reset_optimizer:
            if ck.get("optimizer_G") is not None:
                self.optimizer_G.load_state_dict(ck["optimizer_G"])
            if ck.get("optimizer_D") is not None:
                self.optimizer_D.load_state_dict(ck["optimizer_D"])
        return ck["global_step"], ck["global_epoch"], ck["global_test_step"]

    def load_part_checkpoint(self, path=None):
        """tolerant partial load of E and G only (utils/util.py:124-144,165-173)."""
        path = path if path is not None else getattr(self.hparams, "resume_path", None)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.sync_pending_update()          # the deferred Adam(E,G) of the last step may still be writing these parameters
        for mod, key in ((self.Mel_Encoder, "Mel_Encoder"), (self.Mel_Decoder, "Mel_Decoder")):
            own = mod.state_dict()
            for k, v in ck[key].items():
                if k in own and tuple(own[k].shape) == tuple(v.shape):
                    own[k].copy_(v)
        self.weights_changed()
        return self
