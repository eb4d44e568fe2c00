"""PyTorch autograd front of the HIP kernels (host side of the drop-in boundary).

Every op here is a `torch.autograd.Function` whose forward AND backward are
launches into libviai_hip.so on torch's current HIP stream; outputs are ordinary
autograd-connected CUDA tensors, so the reference's `loss_functions.GANLoss`,
`torch.optim.Adam` and `train_whole_sync.py`-style loops work on them unchanged
(SURVEY.md §8b "Autograd").  Tensors cross this layer as fp32 NHWC
((N,H,W,C) contiguous); `networks.py` does the NCHW<->NHWC views.

There is no CPU path: a non-CUDA tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os

import weakref

import torch

from . import _lib
from ._lib import Conv2dDesc

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3
BN_EPS = 1e-5

# When True (set by model.AudioModel around its step), a parameter whose .grad already exists (a view of
# the flat gradient arena) receives its gradient IN PLACE from the wgrad / BN-backward kernels
# (accumulate mode) and autograd gets None for it: no per-parameter torch add launches.
DIRECT_GRAD = False

# Weight gradients on a side stream (set by model.AudioModel together with DIRECT_GRAD): the wgrad of a layer only
# feeds the optimizer, so it is launched on WGRAD_STREAM behind an event and overlaps the rest of the backward chain
# (the HBM-bound BatchNorm-backward kernels leave the matrix pipes idle; the MFMA-bound wgrad fills them).  The
# operands are kept alive in _deferred until join_wgrad() makes the main stream wait for the side stream.
WGRAD_STREAM = None
_deferred = []
F16_BACKWARD = os.environ.get("VIAI_F16_BACKWARD", "1") != "0"     # f16x2 data gradients (dynamic per-tensor scale)
F16_DYNAMIC = True                                                 # False: static x16 activation scale in the forward / weight-gradient kernels (module switch, A/B only)

# Pre-split tensors (P16, include/viai_hip.h ABI 13).  A tensor tagged `_viai_p16` holds, in the bytes of an fp32 NHWC tensor of the same
# shape, the two fp16 planes the f16x2 kernels would make of its values (scale from its `_viai_amax` slot, filled by the producer with an
# a-priori bound).  Producers: the BatchNorm apply passes (forward: where the caller asks with out_p16 -- networks.py knows the consumer;
# backward: where this layer's own data- / weight-gradient kernels take it).  Consumers: conv_bn_act / conv_bn_act_cout1 (anything else
# refuses a P16 tensor; p16_decode() gives the fp32 values).  VIAI_P16=0 switches the format off (A/B: a kernel returns the same bits on P16 operands as on the fp32 tensors AT EQUAL SCALE -- tests/test_p16_gpu.py;
# the P16 producers derive the scale from an a-priori bound where the fp32 path measures the maximum, so whole-network results agree to rounding, not bit for bit).
P16 = os.environ.get("VIAI_P16", "1") != "0"
P16_OK_FWD_X, P16_OK_DGRAD_DY, P16_OK_WGRAD_DY, P16_OK_WGRAD_X, P16_OK_FWD_LIN = 1, 2, 4, 8, 16


def is_p16(t):
    return t is not None and getattr(t, "_viai_p16", False)


def p16_mask(d):
    # (the library reads two of its kernel-family switches per call -- tests flip them at run time -- so the cached mask is keyed on them)
    key = ("p16", os.environ.get("VIAI_WGRAD_PATCH_S2"))
    m = d.get(key)
    if m is None:
        m = d[key] = int(_lib.load().viai_conv2d_p16_ok(d["ref"]))
    return m


def _p16_decode(t, amax):
    out = torch.empty_like(t)
    Cc = t.shape[-1]
    _lib.check(_lib.load().viai_p16_decode(t.data_ptr(), out.data_ptr(), t.numel() // Cc, Cc, amax.data_ptr(), _stream()), "viai_p16_decode")
    out._viai_amax = amax
    return out


def p16_decode(t):
    """fp32 values of a pre-split tensor (one streaming pass): for consumers without a P16 loader, and for tests"""
    return _p16_decode(t, t._viai_amax) if is_p16(t) else t


def conv_takes_p16(x_shape, weight, kernel, stride, padding, transposed):
    """will conv_bn_act on an input of this NHWC shape stage a pre-split x in both its forward and its weight-gradient kernel?
    (what a producer asks before it writes its output as P16: networks.py)"""
    if not P16:
        return False
    N, IH, IW, C1 = x_shape
    Cout = weight.shape[1] if transposed else weight.shape[0]
    if C1 % 32 != 0:
        return False
    d = conv_desc(N, IH, IW, C1, 0, Cout, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], 1 if transposed else 0)
    return (p16_mask(d) & (P16_OK_FWD_X | P16_OK_WGRAD_X)) == (P16_OK_FWD_X | P16_OK_WGRAD_X)


def conv_wgrad_takes_p16(x_shape, weight, kernel, stride, padding, transposed):
    """... or at least in its weight-gradient kernel?  (a residual join then still writes the twin: the forward of such a layer reads the fp32 tensor)"""
    if not (P16 and F16_BACKWARD):
        return False
    N, IH, IW, C1 = x_shape
    Cout = weight.shape[1] if transposed else weight.shape[0]
    if C1 % 32 != 0:
        return False
    d = conv_desc(N, IH, IW, C1, 0, Cout, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], 1 if transposed else 0)
    if d.get("wgrad_f16") is None:
        d["wgrad_f16"] = bool(_lib.load().viai_conv2d_wgrad_f16_ok(d["ref"]))
    return bool(p16_mask(d) & P16_OK_WGRAD_X) and d["wgrad_f16"]


# Gradient-ready hooks (set by model.AudioModel for the data-parallel exchange): {weight.data_ptr(): callable}.  The callable runs
# right after the backward of the layer owning that weight has queued its LAST gradient launch (weight gradient on WGRAD_STREAM,
# BatchNorm gamma / beta gradients on the current stream), i.e. every parameter gradient of that layer and of all layers behind it
# in the network is queued: the bucket they form can go to RCCL while the rest of the backward chain still runs.
GRAD_HOOKS = {}


# f16x2 range guard.  Activations: the operand scale of the fp16 split follows the tensor's magnitude on the device (`_viai_amax`
# provenance below), so nothing saturates.  Weights are pre-scaled by the static 256 in the pack kernels and CLAMP beyond
# |w| > 65504 / 256 = 255.9: model.AudioModel checks max |w| of its arenas at its host sync point (get_loss_items) and raises; in
# debug mode (VIAI_DEBUG_RANGE=1 or ops.DEBUG_RANGE = True) every conv call scans its weight (one extra streaming pass) and
# range_report() tells how many elements were outside; `strict` raises at the offending layer (a host sync per layer).
DEBUG_RANGE = os.environ.get("VIAI_DEBUG_RANGE", "0") not in ("0", "")
DEBUG_RANGE_STRICT = os.environ.get("VIAI_DEBUG_RANGE", "0") == "strict"
F16_ACT_LIMIT, F16_WEIGHT_LIMIT = 65504.0 / 16.0, 65504.0 / 256.0
_range_counts = {}


def _range_scan(kind, t, limit):
    c = _range_counts.get((kind, t.device))
    if c is None:
        c = _range_counts[(kind, t.device)] = torch.zeros(3, dtype=torch.int32, device=t.device)
    _lib.check(_lib.load().viai_range_count(t.data_ptr(), t.numel(), limit, c.data_ptr(), _stream()), "viai_range_count")
    if DEBUG_RANGE_STRICT and int(c[0]) + int(c[2]) > 0:
        raise FloatingPointError("f16x2 range guard: %s tensor of shape %s has %d element(s) beyond +-%.0f and %d non-finite; "
                                 "run with VIAI_F16X2=0 (bf16x3 has the fp32 exponent range)" % (kind, tuple(t.shape), int(c[0]), limit, int(c[2])))


def range_report(reset=True):
    """{'activation': (n_saturating, max_abs, n_nonfinite), 'weight': (...)} accumulated since the last reset (host sync)."""
    out = {}
    for (kind, _dev), c in _range_counts.items():
        v = c.cpu()
        mx = float(v[1:2].view(torch.float32)[0])
        a = out.get(kind, (0, 0.0, 0))
        out[kind] = (a[0] + int(v[0]), max(a[1], mx), a[2] + int(v[2]))
        if reset:
            c.zero_()
    return out


def join_wgrad():
    """main stream waits for every deferred weight-gradient launch; call before the gradients are consumed."""
    if WGRAD_STREAM is not None and _deferred:
        torch.cuda.current_stream().wait_stream(WGRAD_STREAM)
    _deferred.clear()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """handle of the current stream.  torch.cuda.current_stream() builds a Stream object (~10 us; ~90 calls per step = 0.9 ms of
    host time); the raw-handle query is a plain C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _require(*tensors):
    for t in tensors:
        if t is None:
            continue
        if getattr(t, "_viai_p16", False):
            raise TypeError("this op does not take a pre-split (P16) tensor; ops.p16_decode() gives its fp32 values")
        if not t.is_cuda:
            raise _lib.ViaiLibraryError("viai ops run on the GPU only (got a %s tensor); no CPU fallback" % t.device)
        if t.dtype != torch.float32:
            raise TypeError("viai ops are fp32, got %s" % t.dtype)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_desc_cache = {}

# Scratch buffers (BatchNorm partials, split-K slabs) are dead as soon as the call that fills them has queued its
# last kernel, and kernels on one stream run in order: one growing buffer per (stream, purpose) replaces an
# allocator round trip per layer and direction.
_scratch_pool = {}

# abs-max slots for the f16x2 backward kernels: the BatchNorm-backward apply kernel atomically maxes |dy| into a
# zero-initialised float that the data-gradient kernel turns into its power-of-two operand scale.  A ring of slots
# is zeroed by ONE fill per step (begin_step, called by model.AudioModel); callers that never call begin_step get a
# fresh zeroed scalar per layer instead.
_amax_ring = None
_amax_next = 0
_amax_managed = False


def begin_step(device=None):
    """start of a training step: re-arm the abs-max slots (one fill launch)."""
    global _amax_ring, _amax_next, _amax_managed
    if _amax_ring is None:
        _amax_ring = torch.zeros(1024, device=device if device is not None else "cuda", dtype=torch.float32)
    else:
        _amax_ring.zero_()
    _amax_next = 0
    _amax_managed = True


_free_ring = {}     # device -> [zeroed ring, next index]: slots for callers that never call begin_step (one fill per 1024 slots)


def _amax_slot(dev):
    global _amax_next
    if _amax_managed and _amax_ring is not None and _amax_ring.device == dev and _amax_next < _amax_ring.numel():
        t = _amax_ring[_amax_next:_amax_next + 1]
        _amax_next += 1
        return t
    r = _free_ring.get(dev)
    if r is None or r[1] >= r[0].numel():
        r = _free_ring[dev] = [torch.zeros(1024, device=dev, dtype=torch.float32), 0]     # the old ring lives on in the slices handed out
    t = r[0][r[1]:r[1] + 1]
    r[1] += 1
    return t


# Operand magnitudes for the f16x2 kernels.  Every tensor an op of this module produces behind a BatchNorm (or a bounded activation)
# carries `_viai_amax`: a device float >= max |tensor| that its producer reduced on the way out (viai_bn_act_fwd_amax, the fused
# Cin = 1 layer) or inherited (resampling never increases the maximum).  The consumer hands it to the kernel, which derives the
# power-of-two scale of the fp16 split from it ON THE DEVICE: no host round trip, and no magnitude saturates.  A tensor without
# provenance (user data, torch ops in between) gets one streaming viai_absmax pass when -- and only when -- the layer's forward
# kernel is an f16x2 kernel.
_unit_amax = {}


def _const_amax(dev, value=1.0):
    t = _unit_amax.get((dev, value))
    if t is None:
        t = _unit_amax[(dev, value)] = torch.full((1,), float(value), device=dev, dtype=torch.float32)
    return t


def amax_of(t):
    return getattr(t, "_viai_amax", None) if t is not None else None


def inherit_amax(out, *srcs):
    """out = a resampling / masking of srcs[0] (|out| <= max |src|): the magnitude carries over"""
    a = amax_of(srcs[0])
    if a is not None:
        out._viai_amax = a
    return out


def _input_amax(x, x2, known, st):
    """device float >= max(|x|, |x2|) for an f16x2 forward launch; `known` = the magnitudes the caller's tensors carried"""
    a1, a2 = known
    if x2 is None and a1 is not None:
        return a1
    lib = _lib.load()
    slot = _amax_slot(x.device)
    if a1 is not None and a2 is not None:
        torch.maximum(a1, a2, out=slot)                 # both magnitudes travelled with their tensors (the virtual concat of G.convblock4): one launch
        return slot
    for t, a in ((x, a1), (x2, a2)):
        if t is None:
            continue
        if a is not None:
            torch.maximum(slot, a, out=slot)
        else:
            _lib.check(lib.viai_absmax(t.data_ptr(), t.numel(), slot.data_ptr(), st), "viai_absmax")
    return slot


def scratch_snapshot():
    """the pooled scratch buffers as they are now (see scratch_take_new)"""
    return dict(_scratch_pool)


def scratch_take_new(before):
    """Remove from the pool -- and hand to the caller -- every scratch buffer that was created (or grown) since `before`.  Buffers first
    requested DURING a stream capture are allocated from that capture's private memory pool: they must live and die with the captured
    graphs, not stay in this process-wide pool where a later model would pick up memory of a pool that no longer exists (found as an
    abort two test files after a captured vision-infused model had been deleted)."""
    new = {k: t for k, t in _scratch_pool.items() if before.get(k) is not t}
    for k in new:
        del _scratch_pool[k]
    return new


def drop_scratch():
    """forget every pooled scratch buffer (the next use re-allocates).  Needed when captured hipGraphs are dropped: buffers first
    requested during a capture were allocated from that graph's private pool."""
    _scratch_pool.clear()


# Streams a module forks work onto inside the forward pass (networks.ImageEmbedding2: the flow ResNet).  Autograd replays their backward
# nodes on the same streams, and a branch that ends in a parameter (no data gradient flows back to the caller's stream) is joined by
# nothing: whoever reads the gradients after backward() calls join_side_streams() first.
SIDE_STREAMS = []


def join_side_streams():
    if SIDE_STREAMS:
        cur = torch.cuda.current_stream()
        for s in SIDE_STREAMS:
            if s.device == cur.device:
                cur.wait_stream(s)


def _scratch(tag, n, dev, stream=None):
    key = (tag, dev.index, stream.cuda_stream if stream is not None else _stream())
    t = _scratch_pool.get(key)
    if t is None or t.numel() < n:
        if stream is not None:                      # allocate in the pool of the stream that will use it
            with torch.cuda.stream(stream):
                t = torch.empty(max(n, 1), device=dev, dtype=torch.float32)
        else:
            t = torch.empty(max(n, 1), device=dev, dtype=torch.float32)
        _scratch_pool[key] = t
    return t

# ---------------------------------------------------------------------------------------------- packed-weight cache
# The conv kernels read weights from a packed image (bf16x3 planes, fragment- or row-major) that has to be rebuilt
# whenever the weights change: once per optimizer step.  Parameters get a persistent image per (layer descriptor,
# forward / data-gradient form), stamped with (WEIGHT_EPOCH, tensor version); a conv call re-packs only if the stamp
# is stale.  model.FusedAdam bumps WEIGHT_EPOCH (its kernel writes the arena behind autograd's back) and calls
# repack(), which rebuilds every registered image of its parameters in ONE launch (viai_pack_jobs_run) instead of
# ~60 small launches spread over the step's critical path.  In-place torch updates bump the tensor version and are
# caught by the stamp.  Non-parameter weights (e.g. weight-normed tensors recomputed every call) are packed per call.
WEIGHT_EPOCH = 0
_packs = {}        # (data_ptr, form, id(desc dict)) -> entry
_packs_by_ptr = {}  # data_ptr -> [entries]


_ptr_epoch = {}     # data_ptr -> epoch of the last out-of-band write to that parameter (per parameter: one optimizer's
                    # step must not invalidate the images of the other optimizer's weights)


def _stamp(weight):
    return (WEIGHT_EPOCH, _ptr_epoch.get(weight.data_ptr(), 0), weight._version)


def _packed(weight, d, form, st):
    """packed image of `weight` for descriptor d; form 0 = forward, 1 = data gradient, 2 = f16x2 data gradient."""
    lib = _lib.load()
    fn = (lib.viai_conv2d_pack_fwd, lib.viai_conv2d_pack_dgrad, lib.viai_conv2d_pack_dgrad_f16)[form]
    if not isinstance(weight, torch.nn.Parameter):
        wp = torch.empty(d["packed"], device=weight.device, dtype=torch.float32)
        _lib.check(fn(d["ref"], weight.data_ptr(), wp.data_ptr(), st), "viai_conv2d_pack")
        return wp
    key = (weight.data_ptr(), form, id(d))
    e = _packs.get(key)
    if e is None or e["w"]() is not weight:
        e = {"w": weakref.ref(weight), "wp": torch.empty(d["packed"], device=weight.device, dtype=torch.float32), "d": d, "form": form,
             "stamp": None, "ptr": weight.data_ptr()}
        _packs[key] = e
        lst = [x for x in _packs_by_ptr.get(e["ptr"], []) if x["w"]() is weight and not (x["form"] == form and x["d"] is d)]
        lst.append(e)
        _packs_by_ptr[e["ptr"]] = lst
    stamp = _stamp(weight)
    if e["stamp"] != stamp:
        _lib.check(fn(d["ref"], weight.data_ptr(), e["wp"].data_ptr(), st), "viai_conv2d_pack")
        e["stamp"] = stamp
    return e["wp"]


_job_tables = {}   # id(params list owner) -> (signature, device table, njobs, total_blocks, singles)


def repack(params, owner=None):
    """Rebuild, in one launch, every registered packed image of `params` (call right after the weights changed and
    WEIGHT_EPOCH was bumped).  Images that are not bf16x3 images (streaming / fp32 kernels) are re-packed one by one."""
    lib = _lib.load()
    st = _stream()
    entries = []
    for p in params:
        for e in _packs_by_ptr.get(p.data_ptr(), ()):
            if e["w"]() is p:
                entries.append((p, e))
    if not entries:
        return 0
    sig = tuple((e["ptr"], e["form"], id(e["d"]), e["wp"].data_ptr()) for _, e in entries)
    key = id(owner) if owner is not None else id(params)
    tab = _job_tables.get(key)
    if tab is None or tab[0] != sig:
        jobs, singles, blk = [], [], 0
        for p, e in entries:
            j = _lib.PackJob()
            r = lib.viai_conv2d_pack_job(e["d"]["ref"], e["form"], p.data_ptr(), e["wp"].data_ptr(), C.byref(j))
            if r == 0:
                j.blk0 = blk
                blk += j.nblk
                jobs.append(j)
            elif r == 1:
                singles.append((p, e))
            else:
                _lib.check(r, "viai_conv2d_pack_job")
        dev_tab = None
        if jobs:
            arr = (_lib.PackJob * len(jobs))(*jobs)
            dev_tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(entries[0][0].device)
        tab = (sig, dev_tab, len(jobs), blk, singles)
        _job_tables[key] = tab
    _, dev_tab, njobs, total, singles = tab
    if njobs:
        _lib.check(lib.viai_pack_jobs_run(dev_tab.data_ptr(), njobs, total, st), "viai_pack_jobs_run")
    for p, e in singles:
        fn = (lib.viai_conv2d_pack_fwd, lib.viai_conv2d_pack_dgrad, lib.viai_conv2d_pack_dgrad_f16)[e["form"]]
        _lib.check(fn(e["d"]["ref"], p.data_ptr(), e["wp"].data_ptr(), st), "viai_conv2d_pack")
    for p, e in entries:
        e["stamp"] = _stamp(p)
    return len(entries)


def weights_changed(params=None, owner=None):
    """Call after writing parameters behind autograd's back (raw kernels, arena copies); with `params`, re-packs now."""
    global WEIGHT_EPOCH
    if params is None:
        WEIGHT_EPOCH += 1                      # unknown extent: every cached image is stale
        return
    params = list(params)
    for p in params:
        k = p.data_ptr()
        _ptr_epoch[k] = _ptr_epoch.get(k, 0) + 1
    repack(params, owner)


def _bn_finalize(lib, d, stat, M, Cc, gamma, beta, rmean, rvar, nbt, cfg, coef, st, lin=False):
    """partials of the conv forward -> (mean, invstd, scale, shift) + running statistics; tile-shaped partial blocks where the forward
    kernel's tiles are clipped at the map's edge (viai_conv2d_stat_tiles); `lin`: the pre-split forward ran on the linear-tile kernel, whose
    partial blocks are 128 consecutive pixels (VIAI_P16_OK_FWD_LIN)"""
    th, tw = (0, 0) if lin else d["tiles"]
    tail = (Cc, gamma.data_ptr(), beta.data_ptr(), _ptr(rmean), _ptr(rvar), _ptr(nbt), cfg["momentum"], cfg["eps"],
            coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), st)
    if lin:
        # (the kernel's partials are per 128 pixels or merged per persistent block: the library applies the launch's rule, round 6)
        _lib.check(lib.viai_bn_finalize_lin(stat.data_ptr(), M, *tail), "viai_bn_finalize_lin")
        return
    if th > 0:
        _lib.check(lib.viai_bn_finalize_tiles(stat.data_ptr(), d["N"], d["OH"], d["OW"], th, tw, *tail), "viai_bn_finalize_tiles")
    else:
        _lib.check(lib.viai_bn_finalize(stat.data_ptr(), d["nblk"], d["rows"], M, *tail), "viai_bn_finalize")


def conv_desc(N, IH, IW, C1, C2, Cout, kh, kw, sh, sw, ph, pw, transposed, dh=1, dw=1, ph2=-1, pw2=-1):
    key = (N, IH, IW, C1, C2, Cout, kh, kw, sh, sw, ph, pw, transposed, dh, dw, ph2, pw2)
    d = _desc_cache.get(key)
    if d is None:
        lib = _lib.load()
        desc = Conv2dDesc(*key)
        oh, ow = C.c_int(), C.c_int()
        lib.viai_conv2d_out_hw(C.byref(desc), C.byref(oh), C.byref(ow))
        nblk, rows = C.c_int(), C.c_int()
        _lib.check(lib.viai_conv2d_stat_geom(C.byref(desc), C.byref(nblk), C.byref(rows)), "viai_conv2d_stat_geom")
        th, tw = C.c_int(), C.c_int()
        _lib.check(lib.viai_conv2d_stat_tiles(C.byref(desc), C.byref(th), C.byref(tw)), "viai_conv2d_stat_tiles")
        d = {
            "desc": desc, "ref": C.byref(desc), "OH": oh.value, "OW": ow.value, "N": N,
            "nblk": nblk.value, "rows": rows.value, "tiles": (th.value, tw.value),
            "packed": int(lib.viai_conv2d_packed_floats(C.byref(desc))),
            "ws_floats": int(lib.viai_conv2d_wgrad_ws_bytes(C.byref(desc))) // 4,
        }
        _desc_cache[key] = d
    return d


def _wgrad_call(lib, d, x, x2, dy, ws, dw, db, acc, amax, xa, flags, handle):
    if flags:
        _lib.check(lib.viai_conv2d_wgrad_f16_p16(d["ref"], x.data_ptr(), _ptr(x2), dy.data_ptr(), ws.data_ptr(), dw.data_ptr(), db, acc,
                                                 amax.data_ptr(), _ptr(xa), flags, handle), "viai_conv2d_wgrad_f16_p16")
    elif amax is not None and d["wgrad_f16"]:
        _lib.check(lib.viai_conv2d_wgrad_f16(d["ref"], x.data_ptr(), _ptr(x2), dy.data_ptr(), ws.data_ptr(), dw.data_ptr(), db, acc,
                                             amax.data_ptr(), _ptr(xa), handle), "viai_conv2d_wgrad_f16")
    else:
        _lib.check(lib.viai_conv2d_wgrad(d["ref"], x.data_ptr(), _ptr(x2), dy.data_ptr(), ws.data_ptr(), dw.data_ptr(), db, acc, handle),
                   "viai_conv2d_wgrad")


def _conv_grads(lib, d, cfg, dims, has_bn, has_bias, needs, xa, x, x2, weight, dy, amax, st, dy_p16=False, x_p16=False, dy_w=None):
    """weight / bias / data gradients of one conv layer from dy (the gradient of its output): the tail every fused layer's backward
    shares.  needs = (need_x, need_x2, need_w, need_b); returns (dx, dx2, dw, db) with None where the gradient went into the arena."""
    N, IH, IW, C1, C2, Cout, OH, OW = dims
    M = N * OH * OW
    dev = dy.device
    need_x, need_x2, need_w, need_b = needs
    gt = cfg.get("gt") or (None, None, None, None)
    dw = db = dx = dx2 = None
    if need_w and x_p16 and not (p16_mask(d) & P16_OK_WGRAD_X and (amax is not None and d.get("wgrad_f16"))):
        x = _p16_decode(x, xa)                 # (a layer whose weight gradient cannot stage pieces: not reached by the networks of this package)
        x_p16 = False
    # dy_w: dy once more as planes, for the weight gradient only (the data gradient below reads the fp32 `dy`)
    dyw = dy_w if dy_w is not None else dy
    flags = (1 if (dy_p16 or dy_w is not None) else 0) | (2 if x_p16 else 0)
    if need_w or (need_b and has_bias):
        ws = None
        shadowed = has_bn and cfg["training"]     # bias in front of train-mode BN: gradient is exactly 0
        acc_w = gt[0] is not None and need_w
        dw = gt[0] if acc_w else torch.empty_like(weight)
        acc_b = False
        if has_bias and need_b:
            if gt[1] is not None:
                acc_b, db = True, gt[1]              # (+= 0 when shadowed: nothing to do)
            elif shadowed:
                db = torch.zeros(Cout, device=dev, dtype=torch.float32)
            else:
                db = torch.empty(Cout, device=dev, dtype=torch.float32)
        want_db = db is not None and not shadowed
        if WGRAD_STREAM is not None and acc_w and (acc_b or not want_db):
            # in-place accumulation into the arenas: nothing flows back through autograd, so the launch can trail
            ev = torch.cuda.Event()
            ev.record()
            WGRAD_STREAM.wait_event(ev)
            # (no `with torch.cuda.stream(...)` here: the launch takes the stream handle explicitly, and entering / leaving the
            # context costs ~25 us of host time per layer)
            ws = _scratch("wgrad", d["ws_floats"], dev, WGRAD_STREAM)
            _wgrad_call(lib, d, x, x2, dyw, ws, dw, db.data_ptr() if want_db else 0, 1, amax, xa, flags, WGRAD_STREAM.cuda_stream)
            _deferred.append((x, x2, dyw, weight, amax, xa))
        elif acc_w == acc_b or not want_db:
            ws = _scratch("wgrad", d["ws_floats"], dev)
            _wgrad_call(lib, d, x, x2, dyw, ws, dw, db.data_ptr() if want_db else 0, 1 if acc_w else 0, amax, xa, flags, st)
        else:   # mixed modes (only outside AudioModel): two calls keep the accumulate flag consistent
            if flags:
                raise RuntimeError("pre-split operands with a mixed accumulate / overwrite bias gradient")
            ws = _scratch("wgrad", d["ws_floats"], dev)
            _lib.check(lib.viai_conv2d_wgrad(d["ref"], x.data_ptr(), _ptr(x2), dy.data_ptr(), ws.data_ptr(),
                                             dw.data_ptr(), 0, 1 if acc_w else 0, st), "viai_conv2d_wgrad")
            part_b = torch.empty(lib.viai_colsum_blocks(M, Cout) * Cout, device=dev, dtype=torch.float32)
            _lib.check(lib.viai_colsum(dy.data_ptr(), M, Cout, part_b.data_ptr(), db.data_ptr(), 1 if acc_b else 0, st),
                       "viai_colsum")
        if acc_w or not need_w:
            dw = None
        if acc_b:
            db = None
        if GRAD_HOOKS:
            hook = GRAD_HOOKS.get(weight.data_ptr())
            if hook is not None:
                hook()
    if need_x or need_x2:
        dx = torch.empty((N, IH, IW, C1), device=dev, dtype=torch.float32)
        dx2 = torch.empty((N, IH, IW, C2), device=dev, dtype=torch.float32) if C2 > 0 else None
        if dy_p16:
            wp = _packed(weight, d, 2, st)
            _lib.check(lib.viai_conv2d_dgrad_f16_p16(d["ref"], dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), _ptr(dx2), amax.data_ptr(), st),
                       "viai_conv2d_dgrad_f16_p16")
        elif amax is not None and d["dgrad_f16"]:
            wp = _packed(weight, d, 2, st)
            _lib.check(lib.viai_conv2d_dgrad_f16(d["ref"], dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), _ptr(dx2), amax.data_ptr(), st),
                       "viai_conv2d_dgrad_f16")
        else:
            wp = _packed(weight, d, 1, st)
            _lib.check(lib.viai_conv2d_dgrad(d["ref"], dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), _ptr(dx2), st),
                       "viai_conv2d_dgrad")
    return dx, dx2, dw, db


class _ConvBnAct(torch.autograd.Function):
    """y = conv(x ++ x2, w) + b ; [BatchNorm2d] ; activation  — one fused layer.

    Reference semantics: nn.Conv2d / nn.ConvTranspose2d -> nn.BatchNorm2d ->
    LeakyReLU(0.2)/ReLU/Sigmoid (Inpainting_Networks.py:71-76,
    New_Inpainting_Networks.py:31-37,71-75,85-88, Discriminator_Networks.py:38-49)."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, gamma, beta, rmean, rvar, nbt, res, cfg):
        lib = _lib.load()
        xp = is_p16(x)
        _require(None if xp else x, x2, weight, bias, gamma, beta, res)
        x = _c(x)
        x2 = _c(x2) if x2 is not None else None
        weight = _c(weight)
        N, IH, IW, C1 = x.shape
        C2 = x2.shape[3] if x2 is not None else 0
        kh, kw = cfg["k"]
        transposed = cfg["transposed"]
        Cout = weight.shape[1] if transposed else weight.shape[0]
        cin_w = weight.shape[0] if transposed else weight.shape[1]
        if C2 == 0 and C1 == 4 and 1 < cin_w < 4:
            C1 = cin_w                      # frames stored with channel stride 4 (zero padded): ResNet conv1
        dil, p2 = cfg.get("d", (1, 1)), cfg.get("p2", (-1, -1))
        d = conv_desc(N, IH, IW, C1, C2, Cout, kh, kw, cfg["s"][0], cfg["s"][1], cfg["p"][0], cfg["p"][1],
                      1 if transposed else 0, dil[0], dil[1], p2[0], p2[1])
        st = _stream()
        dev = x.device
        OH, OW = d["OH"], d["OW"]
        M = N * OH * OW
        if DEBUG_RANGE:
            _range_scan("weight", weight, F16_WEIGHT_LIMIT)
        wp = _packed(weight, d, 0, st)
        has_bn = gamma is not None
        act = cfg["act"]
        training = cfg["training"]
        f16f = d.get("fwd_f16")
        if f16f is None:
            f16f = d["fwd_f16"] = bool(lib.viai_conv2d_fwd_f16_ok(d["ref"]))
        twin = cfg.pop("x_twin", None)          # a pre-split copy of x beside the fp32 tensor (the residual join of a ResNet block writes both)
        # (both consumers of x must stage pieces: otherwise the weight gradient would decode the twin -- coarser scale -- although the exact fp32 x is at hand)
        ctx.x_twin_w = None
        if twin is not None and not xp and x2 is None and P16 and (p16_mask(d) & P16_OK_FWD_X) and (p16_mask(d) & P16_OK_WGRAD_X) and F16_BACKWARD:
            x, xp = twin, True                  # the kernels read the planes; the gradient still goes to the fp32 tensor this op was applied to
        elif twin is not None and not xp and x2 is None and P16 and (p16_mask(d) & P16_OK_WGRAD_X) and F16_BACKWARD and ctx.needs_input_grad[2]:
            # only the weight-gradient kernel stages pieces (the stride-2 3 x 3 convs of ResNet-18 on 28 / 14 / 7-pixel maps: their forward runs on the
            # gather kernel): the forward reads the fp32 tensor, the weight gradient the planes -- it splits nothing (1143 -> ~650 us per launch)
            ctx.x_twin_w = twin
        if xp and (x2 is not None or (p16_mask(d) & P16_OK_FWD_X) == 0):
            x = p16_decode(x)                   # (a layer without a P16 loader: the networks of this package ask conv_takes_p16 first)
            xp = False
        if xp:
            xa = amax_of(x)                     # the scale the planes were written with
        else:
            xa = _input_amax(x, x2, cfg.get("xa_in", (None, None)), st) if (f16f and F16_DYNAMIC) else None   # operand magnitude of the f16x2 split (forward and weight gradient)
        ctx.xa = xa
        ctx.x_p16 = xp

        def conv_fwd(out, stat, act_):
            if xp:
                _lib.check(lib.viai_conv2d_fwd_p16(d["ref"], x.data_ptr(), wp.data_ptr(), _ptr(bias), out.data_ptr(), stat, act_, xa.data_ptr(), st), "viai_conv2d_fwd_p16")
            else:
                _lib.check(lib.viai_conv2d_fwd_amax(d["ref"], x.data_ptr(), _ptr(x2), wp.data_ptr(), _ptr(bias), out.data_ptr(), stat, act_, _ptr(xa), st),
                           "viai_conv2d_fwd")
        za = None
        fused1 = False
        if has_bn and training and bias is None and C1 + C2 == 1:
            fused1 = d.get("cin1_bn")
            if fused1 is None:
                fused1 = d["cin1_bn"] = bool(lib.viai_conv2d_cin1_bn_ok(d["ref"]))
        ctx.fused1 = fused1
        xmask = cfg.get("xmask")
        if xmask is not None and not fused1:
            raise RuntimeError("conv_bn_act: xmask reached a layer that is not the fused Cin = 1 layer (conv_bn_act applies it up front otherwise)")
        ctx.xmask = xmask
        if fused1:
            # Cin = 1 conv + BatchNorm(train) + activation: the pre-BatchNorm tensor is never stored (recomputed from x where needed)
            coef = torch.empty((4, Cout), device=dev, dtype=torch.float32)
            stat = _scratch("stat", 2 * Cout * d["nblk"], dev)
            _lib.check(lib.viai_conv2d_cin1_bn_fwd(d["ref"], x.data_ptr(), _ptr(xmask), wp.data_ptr(), 0, stat.data_ptr(), 0, 0, 0, act, 0, st),
                       "viai_conv2d_cin1_bn_fwd")
            _bn_finalize(lib, d, stat, M, Cout, gamma, beta, rmean, rvar, nbt, cfg, coef, st)
            z = torch.empty((N, OH, OW, Cout), device=dev, dtype=torch.float32)
            za = _amax_slot(dev)
            if cfg.get("p16_out") and P16 and act in (ACT_NONE, ACT_RELU, ACT_LRELU):
                _lib.check(lib.viai_conv2d_cin1_bn_fwd_p16(d["ref"], x.data_ptr(), _ptr(xmask), wp.data_ptr(), 0, coef[2].data_ptr(), coef[3].data_ptr(),
                                                           gamma.data_ptr(), beta.data_ptr(), M, z.data_ptr(), act, za.data_ptr(), st), "viai_conv2d_cin1_bn_fwd_p16")
                cfg["z_p16"] = True
            else:
                _lib.check(lib.viai_conv2d_cin1_bn_fwd(d["ref"], x.data_ptr(), _ptr(xmask), wp.data_ptr(), 0, 0, coef[2].data_ptr(), coef[3].data_ptr(),
                                                       z.data_ptr(), act, za.data_ptr(), st), "viai_conv2d_cin1_bn_fwd")
            ctx.save_for_backward(x, None, weight, None, coef)
        elif has_bn:
            y = torch.empty((N, OH, OW, Cout), device=dev, dtype=torch.float32)
            coef = torch.empty((4, Cout), device=dev, dtype=torch.float32)   # mean, invstd, scale, shift
            if training:
                stat = _scratch("stat", 2 * Cout * d["nblk"], dev)
                conv_fwd(y, stat.data_ptr(), ACT_NONE)
                _bn_finalize(lib, d, stat, M, Cout, gamma, beta, rmean, rvar, nbt, cfg, coef, st, lin=xp and bool(p16_mask(d) & P16_OK_FWD_LIN))
            else:
                conv_fwd(y, 0, ACT_NONE)
                _lib.check(lib.viai_bn_eval_coeffs(Cout, gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(),
                                                   rvar.data_ptr(), cfg["eps"], coef[0].data_ptr(), coef[1].data_ptr(),
                                                   coef[2].data_ptr(), coef[3].data_ptr(), st), "viai_bn_eval_coeffs")
            za = _amax_slot(dev)
            pool = cfg.get("pool")
            if pool is not None:
                # BatchNorm + activation + max-pool in one pass over y: the post-activation map is never stored (the backward gathers its
                # gradient from the pooled gradient and the argmax bytes inside the BatchNorm-backward passes)
                k_, s_, p_ = pool
                PH, PW = (OH + 2 * p_ - k_) // s_ + 1, (OW + 2 * p_ - k_) // s_ + 1
                z = torch.empty((N, PH, PW, Cout), device=dev, dtype=torch.float32)
                pidx = torch.empty((N, PH, PW, Cout), device=dev, dtype=torch.uint8)
                if cfg.get("p16_out") and P16 and training and Cout % 32 == 0 and (k_, s_, p_) == (3, 2, 1) and z.numel() // 4 < (1 << 31) - (1 << 24):
                    zp = torch.empty_like(z)
                    pa = _amax_slot(dev)
                    _lib.check(lib.viai_bn_act_maxpool_fwd_twin(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), gamma.data_ptr(), beta.data_ptr(), M,
                                                                z.data_ptr(), zp.data_ptr(), pidx.data_ptr(), N, OH, OW, Cout, k_, s_, p_, act, 0.2,
                                                                za.data_ptr(), pa.data_ptr(), st), "viai_bn_act_maxpool_fwd_twin")
                    zp._viai_p16, zp._viai_amax = True, pa
                    cfg["z_twin"] = zp
                else:
                    _lib.check(lib.viai_bn_act_maxpool_fwd(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), z.data_ptr(), pidx.data_ptr(),
                                                           N, OH, OW, Cout, k_, s_, p_, act, 0.2, za.data_ptr(), st), "viai_bn_act_maxpool_fwd")
                ctx.save_for_backward(x, x2, weight, y, coef, pidx)
            elif res is not None:
                # BatchNorm + residual add + activation in one pass (ResNet BasicBlock); z is kept: the activation's mask needs the sum
                res = _c(res)
                z = torch.empty_like(y)
                ra = amax_of(res)
                c4 = Cout // 4
                if (cfg.get("p16_out") and P16 and training and ra is not None and Cout % 32 == 0 and c4 <= 256 and (c4 & (c4 - 1)) == 0):
                    # the next block's conv1 stages pre-split pieces: the join writes z twice (fp32 for the next join and the mask, P16 for the convs)
                    zp = torch.empty_like(y)
                    pa = _amax_slot(dev)
                    _lib.check(lib.viai_bn_add_act_fwd_twin(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), gamma.data_ptr(), beta.data_ptr(), M,
                                                            res.data_ptr(), ra.data_ptr(), z.data_ptr(), zp.data_ptr(), M, Cout, act, 0.2,
                                                            za.data_ptr(), pa.data_ptr(), st), "viai_bn_add_act_fwd_twin")
                    zp._viai_p16, zp._viai_amax = True, pa
                    cfg["z_twin"] = zp
                else:
                    _lib.check(lib.viai_bn_add_act_fwd_amax(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), res.data_ptr(), z.data_ptr(),
                                                            M, Cout, act, 0.2, za.data_ptr(), st), "viai_bn_add_act_fwd")
                ctx.save_for_backward(x, x2, weight, y, coef, z)
            elif cfg.get("up") is not None:
                # BatchNorm + activation + the F.interpolate behind the layer in one pass over y: the post-activation map is not stored
                # (the backward gathers its gradient with the resize's backward and goes on from y)
                UH, UW = cfg["up"]
                z = torch.empty((N, UH, UW, Cout), device=dev, dtype=torch.float32)
                if cfg.get("p16_out") and P16 and training and Cout % 32 == 0:
                    _lib.check(lib.viai_bn_act_bilinear_fwd_p16(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), gamma.data_ptr(), beta.data_ptr(), M,
                                                                z.data_ptr(), N, OH, OW, UH, UW, Cout, act, 0.2, za.data_ptr(), st), "viai_bn_act_bilinear_fwd_p16")
                    cfg["z_p16"] = True
                else:
                    _lib.check(lib.viai_bn_act_bilinear_fwd_amax(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), z.data_ptr(), N, OH, OW, UH, UW,
                                                                 Cout, act, 0.2, za.data_ptr(), st), "viai_bn_act_bilinear_fwd")
                ctx.save_for_backward(x, x2, weight, y, coef)
            elif cfg.get("p16_out") and P16 and training and Cout % 32 == 0 and act in (ACT_NONE, ACT_RELU, ACT_LRELU):
                # the consumer stages pre-split pieces (networks.py asked conv_takes_p16): z is written as the two fp16 planes, scale from the bound
                z = torch.empty_like(y)
                _lib.check(lib.viai_bn_act_fwd_p16(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, z.data_ptr(),
                                                   M, Cout, act, 0.2, za.data_ptr(), st), "viai_bn_act_fwd_p16")
                cfg["z_p16"] = True
                ctx.save_for_backward(x, x2, weight, y, coef)
            else:
                z = torch.empty_like(y)
                _lib.check(lib.viai_bn_act_fwd_amax(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), z.data_ptr(),
                                                    M, Cout, act, 0.2, za.data_ptr(), st), "viai_bn_act_fwd")
                ctx.save_for_backward(x, x2, weight, y, coef)
            ctx.tail = "pool" if pool is not None else ("res" if res is not None else ("up" if cfg.get("up") is not None else None))
        else:
            z = torch.empty((N, OH, OW, Cout), device=dev, dtype=torch.float32)
            conv_fwd(z, 0, act)
            if act == ACT_SIGMOID:
                za = _const_amax(dev, 1.0)
            ctx.save_for_backward(x, x2, weight, z, None)
        if not has_bn or fused1:
            if res is not None or cfg.get("pool") is not None or cfg.get("up") is not None:
                raise RuntimeError("conv_bn_act: residual / pool / upsample need a BatchNorm layer on the MFMA path")
            ctx.tail = None
        cfg["za"] = za                       # conv_bn_act attaches it to the returned tensor
        ctx.d = d
        ctx.cfg = cfg
        ctx.has_bn = has_bn
        ctx.has_bias = bias is not None
        ctx.dims = (N, IH, IW, C1, C2, Cout, OH, OW)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        x, x2, weight, y_or_z, coef = ctx.saved_tensors[:5]
        d, cfg = ctx.d, ctx.cfg
        N, IH, IW, C1, C2, Cout, OH, OW = ctx.dims
        M = N * OH * OW
        st = _stream()
        dev = dz.device
        addend = _take_addend(ctx)
        dz = _c(dz)
        act = cfg["act"]
        # (the pool tail takes two addends as well -- viai_bn_act_pool_bwd_amax2.  With the per-pixel apply pass, which gathered up to four windows per pixel,
        # two tensors to gather cost more than the add they save (103.5 against 102.2 ms on the vision-infused step); with the 2 x 2-block pass it is a
        # small gain: 102.17 -> 101.99, two same-box pairs)
        if addend is not None and not ((ctx.tail == "res" and act != ACT_NONE and not ctx.fused1 and dz.numel() % 4 == 0) or (POOL_ADDENDS and ctx.tail == "pool" and ctx.has_bn)):
            dz = dz + addend                              # no pass of this backward to fold the sum into
            addend = None
        need_x, need_x2, need_w, need_b, need_g, need_be = ctx.needs_input_grad[:6]
        gt = cfg.get("gt") or (None, None, None, None)       # in-place gradient targets (arena views)
        dgamma = dbeta = None
        amax = None
        dy_p16 = False
        dy_w = None                                    # dy as planes for the weight gradient only (see dy_tw below)
        if ctx.fused1:
            return _ConvBnAct._backward_cin1(ctx, lib, dz, x, weight, coef, st)
        dres = None
        if ctx.tail == "up":
            UH, UW = cfg["up"]
            dlo = torch.empty((N, OH, OW, Cout), device=dev, dtype=torch.float32)
            _lib.check(lib.viai_bilinear_ac_bwd(dz.data_ptr(), dlo.data_ptr(), N, OH, OW, UH, UW, Cout, st), "viai_bilinear_ac_bwd")
            dz = dlo
        join = None                                    # (dz, addend, saved join output): the masked sum is made inside the BatchNorm backward's reduce pass
        if ctx.tail == "res" and act == ACT_RELU and JOIN_FUSED and ctx.has_bn and P16 and F16_BACKWARD and Cout % 32 == 0 and (need_x or need_w):
            pm_ = p16_mask(d)
            f16d_ = d.get("dgrad_f16")
            if f16d_ is None:
                f16d_ = d["dgrad_f16"] = bool(lib.viai_conv2d_dgrad_f16_ok(d["ref"]))
            f16w_ = d.get("wgrad_f16")
            if f16w_ is None:
                f16w_ = d["wgrad_f16"] = bool(lib.viai_conv2d_wgrad_f16_ok(d["ref"]))
            # (exactly the layers whose dy the plain path below would write as planes: same condition)
            if ((f16d_ and need_x) or (f16w_ and need_w)) and (not need_x or (f16d_ and pm_ & P16_OK_DGRAD_DY)) and (not need_w or (f16w_ and pm_ & P16_OK_WGRAD_DY)) \
                    and not (need_b and ctx.has_bias and not cfg["training"]):
                join = (dz, addend, ctx.saved_tensors[5])
                dres = torch.empty_like(dz)
                dz = dres
                act = ACT_NONE
        if ctx.tail == "res" and join is None:
            # d/d(sum) through the activation (mask from the saved output); the same tensor is the residual branch's gradient
            if act != ACT_NONE:
                dres = torch.empty_like(dz)
                if addend is not None:
                    _lib.check(lib.viai_add_act_bwd_from_output(dz.data_ptr(), addend.data_ptr(), ctx.saved_tensors[5].data_ptr(), dres.data_ptr(), dz.numel(), act, 0.2,
                                                                st), "viai_add_act_bwd_from_output")
                else:
                    _lib.check(lib.viai_act_bwd_from_output(dz.data_ptr(), ctx.saved_tensors[5].data_ptr(), dres.data_ptr(), dz.numel(), act, 0.2, st),
                               "viai_act_bwd_from_output")
                dz = dres
            else:
                dres = dz
            act = ACT_NONE
        if ctx.has_bn:
            nblk = lib.viai_bn_bwd_blocks(M, Cout)
            part = _scratch("bnpart", 2 * Cout * nblk, dev)
            sums = _scratch("bnsums", 2 * Cout, dev)
            acc_bn = gt[2] is not None and gt[3] is not None and need_g and need_be
            if acc_bn:
                pg, pb = gt[2], gt[3]
            else:
                dgamma = torch.empty(Cout, device=dev, dtype=torch.float32) if need_g else None
                dbeta = torch.empty(Cout, device=dev, dtype=torch.float32) if need_be else None
                pg, pb = dgamma, dbeta
            dy = torch.empty_like(dz)
            f16d = d.get("dgrad_f16")
            if f16d is None:
                f16d = d["dgrad_f16"] = bool(lib.viai_conv2d_dgrad_f16_ok(d["ref"]))
            f16w = d.get("wgrad_f16")
            if f16w is None:
                f16w = d["wgrad_f16"] = bool(lib.viai_conv2d_wgrad_f16_ok(d["ref"]))
            # max |dy|: operand scale of the f16x2 data- and weight-gradient kernels
            amax = _amax_slot(dev) if (F16_BACKWARD and ((f16d and need_x) or (f16w and need_w))) else None
            # dy pre-split (P16) when every kernel that reads it stages pieces: this layer's data gradient (if needed) and weight gradient
            # (if needed); a bias gradient (column sums of dy) needs the fp32 tensor
            pm = p16_mask(d)
            dy_p16 = (P16 and amax is not None and Cout % 32 == 0 and act != ACT_SIGMOID and ctx.tail in (None, "up", "res")
                      and (need_x or need_w) and (not need_x or (f16d and pm & P16_OK_DGRAD_DY)) and (not need_w or (f16w and pm & P16_OK_WGRAD_DY))
                      and not (need_b and ctx.has_bias and not cfg["training"]))
            # ... and where only the weight gradient does (its data-gradient kernel reads fp32): both forms in one apply pass
            dy_tw = (not dy_p16 and P16 and amax is not None and Cout % 32 == 0 and act != ACT_SIGMOID and ctx.tail in (None, "up", "res")
                     and need_x and need_w and f16w and bool(pm & P16_OK_WGRAD_DY) and not (need_b and ctx.has_bias and not cfg["training"]))
            if dy_tw:
                part = _scratch("bnpart", 3 * Cout * nblk, dev)
                sums = _scratch("bnsums", 3 * Cout, dev)
                dy_w = torch.empty_like(dz)
                _lib.check(lib.viai_bn_act_bwd_p16_twin(dz.data_ptr(), y_or_z.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                                        coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(), sums.data_ptr(),
                                                        _ptr(pg), _ptr(pb), dy_w.data_ptr(), dy.data_ptr(), M, Cout, act, 0.2,
                                                        (1 if cfg["training"] else 0) | (2 if acc_bn else 0), amax.data_ptr(), st), "viai_bn_act_bwd_p16_twin")
            elif join is not None:
                if not dy_p16:
                    raise RuntimeError("conv_bn_act backward: the fused join pass was chosen for a layer whose dy is not written as planes")
                part = _scratch("bnpart", 3 * Cout * nblk, dev)
                sums = _scratch("bnsums", 3 * Cout, dev)
                _lib.check(lib.viai_bn_join_bwd_p16(join[0].data_ptr(), _ptr(join[1]), join[2].data_ptr(), dres.data_ptr(), y_or_z.data_ptr(),
                                                    coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(), sums.data_ptr(),
                                                    _ptr(pg), _ptr(pb), dy.data_ptr(), M, Cout, (1 if cfg["training"] else 0) | (2 if acc_bn else 0), amax.data_ptr(), st),
                           "viai_bn_join_bwd_p16")
            elif dy_p16:
                part = _scratch("bnpart", 3 * Cout * nblk, dev)
                sums = _scratch("bnsums", 3 * Cout, dev)
                _lib.check(lib.viai_bn_act_bwd_p16(dz.data_ptr(), y_or_z.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                                   coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(), sums.data_ptr(),
                                                   _ptr(pg), _ptr(pb), dy.data_ptr(), M, Cout, act, 0.2,
                                                   (1 if cfg["training"] else 0) | (2 if acc_bn else 0), amax.data_ptr(), st), "viai_bn_act_bwd_p16")
            elif ctx.tail == "pool":
                k_, s_, p_ = cfg["pool"]
                dy = torch.empty_like(y_or_z)
                # (the pooled gradient may have arrived as two addends -- ops.fork2: the stem's output feeds layer1's conv1 and its first join -- summed on load)
                _lib.check(lib.viai_bn_act_pool_bwd_amax2(dz.data_ptr(), _ptr(addend), ctx.saved_tensors[5].data_ptr(), N, OH, OW, k_, s_, p_, y_or_z.data_ptr(),
                                                         coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(),
                                                         part.data_ptr(), sums.data_ptr(), _ptr(pg), _ptr(pb), dy.data_ptr(), Cout, act, 0.2,
                                                          (1 if cfg["training"] else 0) | (2 if acc_bn else 0), _ptr(amax), st), "viai_bn_act_pool_bwd")
            else:
                _lib.check(lib.viai_bn_act_bwd_amax(dz.data_ptr(), y_or_z.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                                    coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(), sums.data_ptr(),
                                                    _ptr(pg), _ptr(pb), dy.data_ptr(), M, Cout, act, 0.2,
                                                    (1 if cfg["training"] else 0) | (2 if acc_bn else 0), _ptr(amax), st), "viai_bn_act_bwd")
        elif act == ACT_NONE:
            dy = dz
        else:
            dy = torch.empty_like(dz)
            _lib.check(lib.viai_act_bwd_from_output(dz.data_ptr(), y_or_z.data_ptr(), dy.data_ptr(), dz.numel(), act,
                                                    0.2, st), "viai_act_bwd_from_output")
        xw, xwa, xwp = x, ctx.xa, ctx.x_p16
        if ctx.x_twin_w is not None and need_w:
            xw, xwa, xwp = ctx.x_twin_w, amax_of(ctx.x_twin_w), True          # planes for the weight gradient (the forward read the fp32 tensor)
        dx, dx2, dw, db = _conv_grads(lib, d, cfg, ctx.dims, ctx.has_bn, ctx.has_bias, (need_x, need_x2, need_w, need_b), xwa,
                                      xw, x2, weight, dy, amax, st, dy_p16=dy_p16, x_p16=xwp, dy_w=dy_w)
        return dx, dx2, dw, db, dgamma, dbeta, None, None, None, dres, None


# The first layer's weight gradient is the LAST launch of a backward: on the side stream it lengthens the tail the main chain waits for at the join in front
# of Adam (84 / 60 us of waiting in the D / G step, profiles/r04_d_queues_plan.txt); on the main stream the two queues end together.  6.672 -> 6.622 ms
# (six same-box pairs, ranges disjoint); module switch.
CIN1_WGRAD_MAIN = True


def _backward_cin1(ctx, lib, dz, x, weight, coef, st):
    """backward of the fused Cin = 1 conv + BatchNorm(train) + activation layer: y is recomputed from x; dy is written to memory only
    when a data gradient needs it (the frozen-D pass of the G step), the weight gradient forms it on the fly."""
    d, cfg = ctx.d, ctx.cfg
    N, IH, IW, C1, C2, Cout, OH, OW = ctx.dims
    M = N * OH * OW
    dev = dz.device
    act = cfg["act"]
    need_x, _, need_w, _, need_g, need_be = ctx.needs_input_grad[:6]
    gt = cfg.get("gt") or (None, None, None, None)
    wp = _packed(weight, d, 0, st)
    part = _scratch("bnpart1", 2 * Cout * d["nblk"], dev)
    sums = torch.empty(2 * Cout, device=dev, dtype=torch.float32)     # private: the trailing weight gradient reads it
    acc_bn = gt[2] is not None and gt[3] is not None and need_g and need_be
    dgamma = dbeta = None
    if acc_bn:
        pg, pb = gt[2], gt[3]
    else:
        dgamma = torch.empty(Cout, device=dev, dtype=torch.float32) if need_g else None
        dbeta = torch.empty(Cout, device=dev, dtype=torch.float32) if need_be else None
        pg, pb = dgamma, dbeta
    # the data gradient straight from dz (viai_conv2d_cin1_bn_dgrad).  Windows of one row (D.conv1: 1 x 4, the frozen-D pass of the G step)
    # take the single-pass kernel -- thread = output pixel, dz read once, no dy tensor: 55 + 80 us -> one pass over dz.  Other windows map
    # threads to INPUT pixels and recompute y per contributing output pixel: measured SLOWER than writing dy once with the recomputing
    # apply pass (7.48 - 7.50 vs 7.38 - 7.40 ms per step), so there it stays opt-in (VIAI_CIN1_BN_DGRAD=1; =0 switches both off)
    one_row = (cfg["k"][0] == 1 and cfg["s"][0] == 1 and cfg["p"][0] == 0 and not cfg["transposed"])
    fused_dx = need_x and os.environ.get("VIAI_CIN1_BN_DGRAD", "1" if one_row else "0") != "0"
    dy = torch.empty_like(dz) if (need_x and not fused_dx) else None
    xmask = ctx.xmask
    _lib.check(lib.viai_conv2d_cin1_bn_bwd(d["ref"], x.data_ptr(), _ptr(xmask), wp.data_ptr(), 0, dz.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                           coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(), sums.data_ptr(), _ptr(pg), _ptr(pb),
                                           _ptr(dy), act, 1 | (2 if acc_bn else 0), st), "viai_conv2d_cin1_bn_bwd")
    dw = dx = None
    if need_w:
        acc_w = gt[0] is not None
        dw = gt[0] if acc_w else torch.empty_like(weight)

        def wgrad(stream_obj, handle):
            ws = _scratch("wgrad", d["ws_floats"], dev, stream_obj) if stream_obj is not None else _scratch("wgrad", d["ws_floats"], dev)
            _lib.check(lib.viai_conv2d_cin1_bn_wgrad(d["ref"], x.data_ptr(), _ptr(xmask), wp.data_ptr(), 0, dz.data_ptr(), coef[0].data_ptr(),
                                                     coef[2].data_ptr(), coef[3].data_ptr(), sums.data_ptr(), ws.data_ptr(), dw.data_ptr(),
                                                     1 if acc_w else 0, act, handle), "viai_conv2d_cin1_bn_wgrad")
        if WGRAD_STREAM is not None and acc_w and not CIN1_WGRAD_MAIN:
            ev = torch.cuda.Event()
            ev.record()
            WGRAD_STREAM.wait_event(ev)
            wgrad(WGRAD_STREAM, WGRAD_STREAM.cuda_stream)
            _deferred.append((x, dz, weight, wp, coef, sums, xmask))
        else:
            wgrad(None, st)
        if acc_w:
            dw = None
        if GRAD_HOOKS:
            hook = GRAD_HOOKS.get(weight.data_ptr())
            if hook is not None:
                hook()
    if need_x:
        dx = torch.empty((N, IH, IW, C1), device=dev, dtype=torch.float32)
        wpd = _packed(weight, d, 1, st)
        if fused_dx:
            _lib.check(lib.viai_conv2d_cin1_bn_dgrad(d["ref"], x.data_ptr(), _ptr(xmask), wpd.data_ptr(), dz.data_ptr(), coef[0].data_ptr(),
                                                     coef[2].data_ptr(), coef[3].data_ptr(), sums.data_ptr(), dx.data_ptr(), act, st),
                       "viai_conv2d_cin1_bn_dgrad")
        else:
            _lib.check(lib.viai_conv2d_dgrad(d["ref"], dy.data_ptr(), wpd.data_ptr(), dx.data_ptr(), 0, st), "viai_conv2d_dgrad")
        if xmask is not None:                                   # d/ds of conv(s * mask)
            N_, T_ = xmask.shape[0], xmask.shape[-1]
            _lib.check(lib.viai_mask_mul(dx.data_ptr(), xmask.data_ptr(), dx.data_ptr(), N_, dx.numel() // (N_ * T_), T_, st), "viai_mask_mul")
    return dx, None, dw, None, dgamma, dbeta, None, None, None, None, None


_ConvBnAct._backward_cin1 = staticmethod(_backward_cin1)


def _cin1_fused_applies(x, weight, bias, bn, kernel, stride, padding, transposed, training):
    """will this call take the fused Cin = 1 conv + BatchNorm(train) layer?  (the predicate of _ConvBnAct.forward)"""
    if bn is None or not training or bias is not None or x.shape[3] != 1 or not isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
        return False
    Cout = weight.shape[1] if transposed else weight.shape[0]
    d = conv_desc(x.shape[0], x.shape[1], x.shape[2], 1, 0, Cout, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], 1 if transposed else 0)
    f = d.get("cin1_bn")
    if f is None:
        f = d["cin1_bn"] = bool(_lib.load().viai_conv2d_cin1_bn_ok(d["ref"]))
    return f


def conv_bn_act(x, weight, bias=None, bn=None, *, kernel, stride=(1, 1), padding=(0, 0), transposed=False,
                act=ACT_NONE, x2=None, training=True, dilation=(1, 1), padding2=(-1, -1), xmask=None, residual=None, pool=None, upsample=None,
                out_p16=False):
    """Fused layer on NHWC tensors.  `bn` is an nn.BatchNorm2d (parameter/buffer holder) or None.
    `padding2` = (bottom, right) padding when it differs from `padding` (-1 = symmetric).
    `xmask` (N, W) or (N,1,1,W): the layer convolves x * xmask (the inpainting step's time mask).  The fused Cin = 1 layer multiplies
    while it loads x; every other kernel gets a masked copy first.
    `residual` (BatchNorm layers): act(BN(conv(x)) + residual) in the BatchNorm-apply pass -- the join of networks/ResNet.py:49-53.
    `pool` = (k, s, p) (BatchNorm layers, act ReLU / none): max_pool2d(act(BN(conv(x))), k, s, p) without storing the un-pooled map --
    the stem of networks/Image_Embedding.py:20-23.
    `upsample` = (H, W) (BatchNorm layers where upsample_fusable() says so): F.interpolate(act(BN(conv(x))), (H, W), mode="bilinear",
    align_corners=True) without storing the map in between -- the decoder blocks of New_Inpainting_Networks.py:76-83."""
    if (residual is not None or pool is not None) and (bn is None or (residual is not None and pool is not None) or act not in (ACT_RELU, ACT_NONE)):
        raise ValueError("conv_bn_act: residual / pool take a BatchNorm layer, ReLU or no activation, and exclude each other")
    if upsample is not None and (residual is not None or pool is not None or not upsample_fusable(x, weight, bias, bn, transposed, act, upsample)):
        raise ValueError("conv_bn_act: upsample takes a BatchNorm layer on the MFMA path (see upsample_fusable) and excludes residual / pool")
    if xmask is not None:
        xmask = _c(xmask.reshape(xmask.shape[0], xmask.shape[-1]))
        trainmode = training if (bn is None or (bn.track_running_stats and bn.running_mean is not None)) else True
        if not (x2 is None and tuple(dilation) == (1, 1) and tuple(padding2) == (-1, -1)
                and _cin1_fused_applies(x, weight, bias, bn, tuple(kernel), tuple(stride), tuple(padding), transposed, trainmode)):
            x = mask_mul(x, xmask)
            xmask = None
    cfg = {"k": tuple(kernel), "s": tuple(stride), "p": tuple(padding), "transposed": bool(transposed),
           "act": int(act), "training": bool(training), "momentum": 0.1, "eps": BN_EPS,
           "d": tuple(dilation), "p2": tuple(padding2), "xa_in": (amax_of(x), amax_of(x2)), "xmask": xmask,
           "pool": tuple(int(v) for v in pool) if pool is not None else None,
           "up": (int(upsample[0]), int(upsample[1])) if upsample is not None else None,
           "p16_out": bool(out_p16) and bn is not None and isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and bn.weight is not None,
           "x_twin": getattr(x, "_viai_twin", None) if x2 is None else None}
    if bn is not None:
        cfg["momentum"] = 0.1 if bn.momentum is None else float(bn.momentum)
        cfg["eps"] = float(bn.eps)
        track = bn.track_running_stats and bn.running_mean is not None
        if not training and not track:
            cfg["training"] = True
        if DIRECT_GRAD:
            cfg["gt"] = tuple(p.grad if (p is not None and p.is_leaf and p.requires_grad and p.grad is not None) else None
                              for p in (weight, bias, bn.weight, bn.bias))
        z = _tag_amax(_ConvBnAct.apply(x, x2, weight, bias, bn.weight, bn.bias,
                                       bn.running_mean if track else None, bn.running_var if track else None,
                                       bn.num_batches_tracked if (track and cfg["training"]) else None, residual, cfg), cfg)
        z._viai_lazy_sum_ok = True                        # its backward takes a gradient with an addend attached (fork2)
        return z
    if DIRECT_GRAD:
        cfg["gt"] = tuple(p.grad if (p is not None and p.is_leaf and p.requires_grad and p.grad is not None) else None
                          for p in (weight, bias, None, None))
    return _tag_amax(_ConvBnAct.apply(x, x2, weight, bias, None, None, None, None, None, None, cfg), cfg)


FUSE_BN_UP = True      # BatchNorm apply + the resize behind a decoder block in one pass (module switch: tests/test_kernels_gpu.py flips it)


def upsample_fusable(x, weight, bias, bn, transposed, act, size=None):
    """can conv_bn_act(..., upsample=size) take this layer?  A BatchNorm2d layer that is not the fused Cin = 1 layer, a piecewise-linear
    activation, a channel count the per-pixel resize kernels take (C / 4 a power of two <= 256), and a target of fewer than 2^24
    pixels (the fused kernel's pixel index; the two-pass path has a generic-index kernel for more)."""
    if size is not None and x.shape[0] * int(size[0]) * int(size[1]) >= (1 << 24):
        return False
    if not FUSE_BN_UP or bn is None or not isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) or act not in (ACT_NONE, ACT_RELU, ACT_LRELU):
        return False
    if x.shape[3] == 1:
        return False
    cout = weight.shape[1] if transposed else weight.shape[0]
    c4 = cout // 4
    return cout % 4 == 0 and 1 <= c4 <= 256 and (c4 & (c4 - 1)) == 0


def _tag_amax(z, cfg):
    za = cfg.pop("za", None)
    if za is not None:
        z._viai_amax = za
    if cfg.pop("z_p16", False):
        z._viai_p16 = True
    zt = cfg.pop("z_twin", None)
    if zt is not None:
        z._viai_twin = zt
    return z


PAIR_FUSED = True     # (conv + BN + act) -> (Cout = 1 conv) pairs as one op (module switch: tests/test_kernels_gpu.py flips it)
PAIR_FWD_DOTS = True        # wide pairs: tap products per pixel + gather instead of bn_act + the row-run forward (7.00 -> 6.94 ms; module switch for A/B)
PAIR_FWD_FUSED = True     # ... and their forward without the tensor in between (module switch)


class _ConvBnActCout1(torch.autograd.Function):
    """p = act2(conv2(act1(BN(conv1(x ++ x2, w1) + b1)), w2) + b2) with conv2 a 3 x 3 / stride 1 / pad 1 (transposed) conv to ONE channel:
    G.conv6_1 + conv6_1_bn + ReLU -> conv6_2 + Sigmoid (New_Inpainting_Networks.py:85-88) and D.conv3 + norm3 + LeakyReLU -> conv4 +
    Sigmoid (Discriminator_Networks.py:44-49).  Same arithmetic as two _ConvBnAct layers, minus three tensors: the front layer's
    post-activation z (the Cout = 1 kernels apply BatchNorm + activation to y on load) and the Cout = 1 layer's data gradient dz
    (the BatchNorm backward forms it from du in registers) -- csrc/conv_direct.hip `viai_pair_cout1_*`."""

    @staticmethod
    def forward(ctx, x, x2, w1, b1, gamma, beta, rmean, rvar, nbt, w2, b2, cfg):
        lib = _lib.load()
        xp = is_p16(x)
        _require(None if xp else x, x2, w1, b1, gamma, beta, w2, b2)
        x = _c(x)
        x2 = _c(x2) if x2 is not None else None
        N, IH, IW, C1 = x.shape
        C2 = x2.shape[3] if x2 is not None else 0
        kh, kw = cfg["k"]
        tr1 = cfg["transposed"]
        Cmid = w1.shape[1] if tr1 else w1.shape[0]
        d = conv_desc(N, IH, IW, C1, C2, Cmid, kh, kw, cfg["s"][0], cfg["s"][1], cfg["p"][0], cfg["p"][1], 1 if tr1 else 0)
        OH, OW = d["OH"], d["OW"]
        d2 = conv_desc(N, OH, OW, Cmid, 0, 1, 3, 3, 1, 1, 1, 1, 1 if cfg["transposed2"] else 0)
        st = _stream()
        dev = x.device
        M = N * OH * OW
        if DEBUG_RANGE:
            _range_scan("weight", w1, F16_WEIGHT_LIMIT)
        wp1 = _packed(w1, d, 0, st)
        wp2 = _packed(w2, d2, 0, st)
        f16f = d.get("fwd_f16")
        if f16f is None:
            f16f = d["fwd_f16"] = bool(lib.viai_conv2d_fwd_f16_ok(d["ref"]))
        if xp and (x2 is not None or (p16_mask(d) & P16_OK_FWD_X) == 0):
            x = p16_decode(x)
            xp = False
        if xp:
            xa = amax_of(x)
        else:
            xa = _input_amax(x, x2, cfg.get("xa_in", (None, None)), st) if (f16f and F16_DYNAMIC) else None
        y = torch.empty((N, OH, OW, Cmid), device=dev, dtype=torch.float32)
        coef = torch.empty((4, Cmid), device=dev, dtype=torch.float32)       # mean, invstd, scale, shift

        def conv_fwd(stat):
            if xp:
                _lib.check(lib.viai_conv2d_fwd_p16(d["ref"], x.data_ptr(), wp1.data_ptr(), _ptr(b1), y.data_ptr(), stat, ACT_NONE, xa.data_ptr(), st), "viai_conv2d_fwd_p16")
            else:
                _lib.check(lib.viai_conv2d_fwd_amax(d["ref"], x.data_ptr(), _ptr(x2), wp1.data_ptr(), _ptr(b1), y.data_ptr(), stat, ACT_NONE, _ptr(xa), st), "viai_conv2d_fwd")
        if cfg["training"]:
            stat = _scratch("stat", 2 * Cmid * d["nblk"], dev)
            conv_fwd(stat.data_ptr())
            _bn_finalize(lib, d, stat, M, Cmid, gamma, beta, rmean, rvar, nbt, cfg, coef, st, lin=xp and bool(p16_mask(d) & P16_OK_FWD_LIN))
        else:
            conv_fwd(0)
            _lib.check(lib.viai_bn_eval_coeffs(Cmid, gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), cfg["eps"],
                                               coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), st),
                       "viai_bn_eval_coeffs")
        p = torch.empty((N, OH, OW, 1), device=dev, dtype=torch.float32)
        if PAIR_FWD_FUSED and Cmid <= 64:
            # BatchNorm + activation applied where the one-channel conv loads y: no z at all.  `cout1_pair_fwd_kernel` reads every input
            # element once (a lane group owns an INPUT pixel and deposits its nine partial dot products in LDS planes); the first
            # attempt -- BatchNorm on load inside the row-run forward kernel, which fetches every element 4.5 - 6 times -- was 3x slower.
            # Standalone (tools/profile_pair.py): G.conv6_1 -> conv6_2 (32 channels, 256 x 256) 62 us against 46 + 57 for the two
            # launches; D.conv3 -> conv4 (512 channels on 64 x 32 maps) 67 us against 24 + 38 -- a whole wave per pixel leaves four
            # lane groups per block and 12 dependent load rounds each, so wide layers keep the two launches
            _lib.check(lib.viai_pair_cout1_fwd(d2["ref"], y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), cfg["act"], wp2.data_ptr(),
                                               _ptr(b2), p.data_ptr(), cfg["act2"], st), "viai_pair_cout1_fwd")
        elif PAIR_FWD_DOTS:
            # wide front layers (D.conv3 -> conv4): one grid-stride pass over y leaves the nine tap products of every pixel, a gather sums
            # them: no z, y read once (was: bn_act_fwd 22 us + the row-run forward 36 us on the 67 MB tensor)
            ws = _scratch("pairdots", 9 * M, dev)
            _lib.check(lib.viai_pair_cout1_fwd_dots(d2["ref"], y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), cfg["act"], wp2.data_ptr(),
                                                    _ptr(b2), ws.data_ptr(), p.data_ptr(), cfg["act2"], st), "viai_pair_cout1_fwd_dots")
        else:
            # z exists only between these two launches: the backward works from y
            z = torch.empty_like(y)
            _lib.check(lib.viai_bn_act_fwd(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), z.data_ptr(), M, Cmid, cfg["act"], 0.2, st),
                       "viai_bn_act_fwd")
            _lib.check(lib.viai_conv2d_fwd(d2["ref"], z.data_ptr(), 0, wp2.data_ptr(), _ptr(b2), p.data_ptr(), 0, cfg["act2"], st),
                       "viai_conv2d_fwd")
            del z
        ctx.save_for_backward(x, x2, w1, y, coef, w2, p)
        ctx.d, ctx.d2, ctx.cfg, ctx.xa = d, d2, cfg, xa
        ctx.x_p16 = xp
        ctx.has_bias, ctx.has_bias2 = b1 is not None, b2 is not None
        ctx.dims = (N, IH, IW, C1, C2, Cmid, OH, OW)
        return p

    @staticmethod
    def backward(ctx, dp):
        lib = _lib.load()
        x, x2, w1, y, coef, w2, p = ctx.saved_tensors
        d, d2, cfg = ctx.d, ctx.d2, ctx.cfg
        N, IH, IW, C1, C2, Cmid, OH, OW = ctx.dims
        M = N * OH * OW
        st = _stream()
        dev = dp.device
        dp = _c(dp)
        need_x, need_x2, need_w1, need_b1, need_g, need_be = ctx.needs_input_grad[:6]
        need_w2, need_b2 = ctx.needs_input_grad[9], ctx.needs_input_grad[10]
        gt = cfg.get("gt") or (None, None, None, None)
        gt2 = cfg.get("gt2") or (None, None)
        # du = gradient of the Cout = 1 layer's pre-activation output (N, OH, OW, 1): 4 bytes per pixel
        if cfg["act2"] == ACT_NONE:
            du = dp
        else:
            du = torch.empty_like(dp)
            _lib.check(lib.viai_act_bwd_from_output(dp.data_ptr(), p.data_ptr(), du.data_ptr(), dp.numel(), cfg["act2"], 0.2, st),
                       "viai_act_bwd_from_output")
        wp2 = _packed(w2, d2, 0, st)
        # ---- the Cout = 1 layer's own parameter gradients
        dw2 = db2 = None
        if need_w2 or (need_b2 and ctx.has_bias2):
            acc_w2 = gt2[0] is not None and need_w2
            dw2 = gt2[0] if acc_w2 else torch.empty_like(w2)
            acc_b2 = False
            if ctx.has_bias2 and need_b2:
                acc_b2 = gt2[1] is not None
                db2 = gt2[1] if acc_b2 else torch.empty(1, device=dev, dtype=torch.float32)
            side = WGRAD_STREAM if (WGRAD_STREAM is not None and acc_w2 and (acc_b2 or db2 is None)) else None
            if side is not None:
                ev = torch.cuda.Event()
                ev.record()
                side.wait_event(ev)
            handle = side.cuda_stream if side is not None else st
            ws = _scratch("wgrad", d2["ws_floats"] + lib.viai_colsum_blocks(M, 1) + 8, dev, side) if side is not None else \
                _scratch("wgrad", d2["ws_floats"] + lib.viai_colsum_blocks(M, 1) + 8, dev)
            if need_w2:
                _lib.check(lib.viai_pair_cout1_wgrad(d2["ref"], y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), cfg["act"], du.data_ptr(),
                                                     ws.data_ptr(), dw2.data_ptr(), 1 if acc_w2 else 0, handle), "viai_pair_cout1_wgrad")
            if db2 is not None:
                _lib.check(lib.viai_colsum(du.data_ptr(), M, 1, ws[d2["ws_floats"]:].data_ptr(), db2.data_ptr(), 1 if acc_b2 else 0, handle),
                           "viai_colsum")
            if side is not None:
                _deferred.append((y, coef, du, w2, ws))
            if acc_w2 or not need_w2:
                dw2 = None
            if acc_b2:
                db2 = None
        # ---- BatchNorm + activation backward of the front layer, dz formed on the fly from du
        nblk = d2.get("pair_blk")
        if nblk is None:
            nblk = d2["pair_blk"] = int(lib.viai_pair_cout1_bn_bwd_blocks(d2["ref"]))
        part = _scratch("bnpart", 2 * Cmid * nblk, dev)
        sums = _scratch("bnsums", 2 * Cmid, dev)
        acc_bn = gt[2] is not None and gt[3] is not None and need_g and need_be
        dgamma = dbeta = None
        if acc_bn:
            pg, pb = gt[2], gt[3]
        else:
            dgamma = torch.empty(Cmid, device=dev, dtype=torch.float32) if need_g else None
            dbeta = torch.empty(Cmid, device=dev, dtype=torch.float32) if need_be else None
            pg, pb = dgamma, dbeta
        f16d = d.get("dgrad_f16")
        if f16d is None:
            f16d = d["dgrad_f16"] = bool(lib.viai_conv2d_dgrad_f16_ok(d["ref"]))
        f16w = d.get("wgrad_f16")
        if f16w is None:
            f16w = d["wgrad_f16"] = bool(lib.viai_conv2d_wgrad_f16_ok(d["ref"]))
        amax = _amax_slot(dev) if (F16_BACKWARD and ((f16d and (need_x or need_x2)) or (f16w and need_w1))) else None
        want_dy = need_x or need_x2 or need_w1 or (need_b1 and ctx.has_bias)
        dy = torch.empty_like(y) if want_dy else None
        pm = p16_mask(d)
        nx = need_x or need_x2
        dy_p16 = (P16 and want_dy and amax is not None and Cmid % 32 == 0 and cfg["act"] != ACT_SIGMOID and (nx or need_w1)
                  and (not nx or (f16d and pm & P16_OK_DGRAD_DY)) and (not need_w1 or (f16w and pm & P16_OK_WGRAD_DY))
                  and not (need_b1 and ctx.has_bias and not cfg["training"]))
        if dy_p16:
            part = _scratch("bnpart", 3 * Cmid * nblk, dev)
            sums = _scratch("bnsums", 3 * Cmid, dev)
        fn = lib.viai_pair_cout1_bn_bwd_p16 if dy_p16 else lib.viai_pair_cout1_bn_bwd
        _lib.check(fn(d2["ref"], du.data_ptr(), wp2.data_ptr(), y.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                      coef[2].data_ptr(), coef[3].data_ptr(), cfg["act"], part.data_ptr(), sums.data_ptr(), _ptr(pg), _ptr(pb),
                      _ptr(dy), (1 if cfg["training"] else 0) | (2 if acc_bn else 0), _ptr(amax), st), "viai_pair_cout1_bn_bwd")
        dx = dx2 = dw1 = db1 = None
        if want_dy:
            dx, dx2, dw1, db1 = _conv_grads(lib, d, cfg, ctx.dims, True, ctx.has_bias, (need_x, need_x2, need_w1, need_b1), ctx.xa,
                                            x, x2, w1, dy, amax, st, dy_p16=dy_p16, x_p16=ctx.x_p16)
        return dx, dx2, dw1, db1, dgamma, dbeta, None, None, None, dw2, db2, None


def conv_bn_act_cout1_ok(x, weight, bn, conv2_weight, *, kernel, stride, padding, transposed, kernel2, stride2, padding2, x2=None):
    """can (conv + bn + act) -> conv2 run as the fused pair?  conv2: 3 x 3, stride 1, pad 1, one output channel; bn a BatchNorm."""
    if not PAIR_FUSED or bn is None or not isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) or not x.is_cuda:
        return False
    if tuple(kernel2) != (3, 3) or tuple(stride2) != (1, 1) or tuple(padding2) != (1, 1):
        return False
    N, IH, IW, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    Cmid = weight.shape[1] if transposed else weight.shape[0]
    d = conv_desc(N, IH, IW, C1, C2, Cmid, kernel[0], kernel[1], stride[0], stride[1], padding[0], padding[1], 1 if transposed else 0)
    d2 = conv_desc(N, d["OH"], d["OW"], Cmid, 0, 1, 3, 3, 1, 1, 1, 1, 0)
    ok = d2.get("pair_ok")
    if ok is None:
        ok = d2["pair_ok"] = bool(_lib.load().viai_pair_cout1_ok(d2["ref"]))
    return ok


def conv_bn_act_cout1(x, weight, bias, bn, weight2, bias2, *, kernel, stride=(1, 1), padding=(0, 0), transposed=False, act=ACT_NONE,
                      transposed2=False, act2=ACT_NONE, x2=None, training=True):
    """the fused pair on NHWC tensors (check conv_bn_act_cout1_ok first)"""
    cfg = {"k": tuple(kernel), "s": tuple(stride), "p": tuple(padding), "transposed": bool(transposed), "act": int(act), "training": bool(training),
           "momentum": 0.1 if bn.momentum is None else float(bn.momentum), "eps": float(bn.eps), "transposed2": bool(transposed2),
           "act2": int(act2), "xa_in": (amax_of(x), amax_of(x2))}
    track = bn.track_running_stats and bn.running_mean is not None
    if not training and not track:
        cfg["training"] = True
    if DIRECT_GRAD:
        def tgt(p):
            return p.grad if (p is not None and p.is_leaf and p.requires_grad and p.grad is not None) else None
        cfg["gt"] = tuple(tgt(p) for p in (weight, bias, bn.weight, bn.bias))
        cfg["gt2"] = (tgt(weight2), tgt(bias2))
    out = _ConvBnActCout1.apply(x, x2, weight, bias, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                                bn.num_batches_tracked if (track and cfg["training"]) else None, weight2, bias2, cfg)
    if act2 == ACT_SIGMOID:
        out._viai_amax = _const_amax(x.device, 1.0)
    return out


class _BilinearAC(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=True) on NHWC (New_Inpainting_Networks.py:78,83)."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        lib = _lib.load()
        _require(x)
        x = _c(x)
        N, IH, IW, Cc = x.shape
        y = torch.empty((N, oh, ow, Cc), device=x.device, dtype=torch.float32)
        _lib.check(lib.viai_bilinear_ac_fwd(x.data_ptr(), y.data_ptr(), N, IH, IW, oh, ow, Cc, _stream()), "viai_bilinear_ac_fwd")
        ctx.dims = (N, IH, IW, oh, ow, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        N, IH, IW, oh, ow, Cc = ctx.dims
        dy = _c(dy)
        dx = torch.empty((N, IH, IW, Cc), device=dy.device, dtype=torch.float32)
        _lib.check(lib.viai_bilinear_ac_bwd(dy.data_ptr(), dx.data_ptr(), N, IH, IW, oh, ow, Cc, _stream()), "viai_bilinear_ac_bwd")
        return dx, None, None


def bilinear_ac(x, size):
    if x.shape[1] == size[0] and x.shape[2] == size[1]:
        return x            # identity resize (align_corners): exact copy semantics
    return inherit_amax(_BilinearAC.apply(x, int(size[0]), int(size[1])), x)      # a convex combination of its inputs


class _AvgPoolH(torch.autograd.Function):
    """nn.AvgPool2d((k,1)) on NHWC (Inpainting_Networks.py:65,77)."""

    @staticmethod
    def forward(ctx, x, k):
        lib = _lib.load()
        _require(x)
        x = _c(x)
        N, IH, W, Cc = x.shape
        y = torch.empty((N, IH // k, W, Cc), device=x.device, dtype=torch.float32)
        _lib.check(lib.viai_avgpool_h_fwd(x.data_ptr(), y.data_ptr(), N, IH, W, Cc, k, _stream()), "viai_avgpool_h_fwd")
        ctx.dims = (N, IH, W, Cc, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        N, IH, W, Cc, k = ctx.dims
        dy = _c(dy)
        dx = torch.empty((N, IH, W, Cc), device=dy.device, dtype=torch.float32)
        _lib.check(lib.viai_avgpool_h_bwd(dy.data_ptr(), dx.data_ptr(), N, IH, W, Cc, k, _stream()), "viai_avgpool_h_bwd")
        return dx, None


def avgpool_h(x, k=3):
    return inherit_amax(_AvgPoolH.apply(x, int(k)), x)


class _ScalarLoss(torch.autograd.Function):
    """mean-reduced BCE / MSE against a scalar label, or L1 between two tensors."""

    @staticmethod
    def forward(ctx, kind, a, b, target):
        lib = _lib.load()
        _require(a, b)
        a = _c(a)
        b = _c(b) if b is not None else None
        n = a.numel()
        part = torch.empty(lib.viai_reduce_blocks(n), device=a.device, dtype=torch.float32)
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        st = _stream()
        if kind == "bce":
            _lib.check(lib.viai_bce_fwd(a.data_ptr(), target, n, part.data_ptr(), loss.data_ptr(), st), "viai_bce_fwd")
        elif kind == "mse":
            _lib.check(lib.viai_mse_fwd(a.data_ptr(), target, n, part.data_ptr(), loss.data_ptr(), st), "viai_mse_fwd")
        else:
            _lib.check(lib.viai_l1_fwd(a.data_ptr(), b.data_ptr(), n, part.data_ptr(), loss.data_ptr(), st), "viai_l1_fwd")
        ctx.kind, ctx.target = kind, target
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = _c(g)
        n = a.numel()
        da = torch.empty_like(a)
        st = _stream()
        if ctx.kind == "bce":
            _lib.check(lib.viai_bce_bwd(a.data_ptr(), ctx.target, n, g.data_ptr(), da.data_ptr(), st), "viai_bce_bwd")
        elif ctx.kind == "mse":
            _lib.check(lib.viai_mse_bwd(a.data_ptr(), ctx.target, n, g.data_ptr(), da.data_ptr(), st), "viai_mse_bwd")
        else:
            _lib.check(lib.viai_l1_bwd(a.data_ptr(), b.data_ptr(), n, g.data_ptr(), da.data_ptr(), st), "viai_l1_bwd")
        db = None
        if ctx.kind == "l1" and ctx.needs_input_grad[2]:
            db = -da
        return None, da, db, None


def bce_mean(p, target: float):
    return _ScalarLoss.apply("bce", p, None, float(target))


def mse_mean(p, target: float):
    return _ScalarLoss.apply("mse", p, None, float(target))


def l1_mean(a, b):
    return _ScalarLoss.apply("l1", a, b, 0.0)


class _MaxPool(torch.autograd.Function):
    """nn.MaxPool2d(k, s, p) on NHWC (networks/Image_Embedding.py:21)."""

    @staticmethod
    def forward(ctx, x, k, s, p):
        lib = _lib.load()
        _require(x)
        x = _c(x)
        N, IH, IW, Cc = x.shape
        OH, OW = (IH + 2 * p - k) // s + 1, (IW + 2 * p - k) // s + 1
        y = torch.empty((N, OH, OW, Cc), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, OH, OW, Cc), device=x.device, dtype=torch.uint8)
        _lib.check(lib.viai_maxpool_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, IH, IW, Cc, k, s, p, _stream()), "viai_maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.dims = (N, IH, IW, Cc, k, s, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        N, IH, IW, Cc, k, s, p = ctx.dims
        dy = _c(dy)
        dx = torch.empty((N, IH, IW, Cc), device=dy.device, dtype=torch.float32)
        _lib.check(lib.viai_maxpool_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), N, IH, IW, Cc, k, s, p, _stream()), "viai_maxpool_bwd")
        return dx, None, None, None


def maxpool(x, k=3, s=2, p=1):
    return inherit_amax(_MaxPool.apply(x, int(k), int(s), int(p)), x)


class _AvgPool2d(torch.autograd.Function):
    """F.avg_pool2d(x, k, s, p, count_include_pad=False) on NHWC."""

    @staticmethod
    def forward(ctx, x, k, s, p):
        lib = _lib.load()
        _require(x)
        x = _c(x)
        N, IH, IW, Cc = x.shape
        OH, OW = (IH + 2 * p - k) // s + 1, (IW + 2 * p - k) // s + 1
        y = torch.empty((N, OH, OW, Cc), device=x.device, dtype=torch.float32)
        _lib.check(lib.viai_avgpool2d_fwd(x.data_ptr(), y.data_ptr(), N, IH, IW, Cc, k, s, p, _stream()), "viai_avgpool2d_fwd")
        ctx.dims = (N, IH, IW, Cc, k, s, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        N, IH, IW, Cc, k, s, p = ctx.dims
        dy = _c(dy)
        dx = torch.empty((N, IH, IW, Cc), device=dy.device, dtype=torch.float32)
        _lib.check(lib.viai_avgpool2d_bwd(dy.data_ptr(), dx.data_ptr(), N, IH, IW, Cc, k, s, p, _stream()), "viai_avgpool2d_bwd")
        return dx, None, None, None


def avgpool2d(x, k=3, s=2, p=1):
    return inherit_amax(_AvgPool2d.apply(x, int(k), int(s), int(p)), x)


class _AvgPoolHW(torch.autograd.Function):
    """mean over the spatial positions: nn.AvgPool2d(7) on a 7x7 map (networks/Image_Embedding.py:27)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        _require(x)
        x = _c(x)
        N, H, W, Cc = x.shape
        y = torch.empty((N, 1, 1, Cc), device=x.device, dtype=torch.float32)
        _lib.check(lib.viai_avgpool_hw_fwd(x.data_ptr(), y.data_ptr(), N, H * W, Cc, _stream()), "viai_avgpool_hw_fwd")
        ctx.dims = (N, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        N, H, W, Cc = ctx.dims
        dy = _c(dy)
        dx = torch.empty((N, H, W, Cc), device=dy.device, dtype=torch.float32)
        _lib.check(lib.viai_avgpool_hw_bwd(dy.data_ptr(), dx.data_ptr(), N, H * W, Cc, _stream()), "viai_avgpool_hw_bwd")
        return dx


def avgpool_hw(x):
    return inherit_amax(_AvgPoolHW.apply(x), x)


POOL_ADDENDS = True   # the stem's pool backward sums two gradient addends on load (module switch; see _ConvBnAct.backward)
JOIN_FUSED = True    # a residual join's masked gradient sum is made inside the BatchNorm backward's reduce pass (viai_bn_join_bwd_p16; module switch: tests flip it)
LAZY_SUM = True      # gradients of a tensor with two readers reach its producer as two addends (fork2; module switch: tests/test_resnet_gpu.py flips it)


class _AddendSlot:
    """where a lazily summed gradient's second addend waits for the producer's backward.  The slot is shared by the _Fork2 node (which fills it)
    and the producer's autograd node (which empties it): it does not ride on the gradient TENSOR, so it survives whatever autograd does to that
    tensor on the way (accumulation with a third reader's gradient, hooks, a contiguous copy)."""
    __slots__ = ("addend",)

    def __init__(self):
        self.addend = None


class _Fork2(torch.autograd.Function):
    """x -> (x, x) for a tensor with two readers (a BasicBlock's input: conv1 and the residual join / the downsample conv, networks/ResNet.py:40-53).
    The backward does NOT add the two gradients: it hands the first on and leaves the second in the slot it shares with the producer's node, and the
    producer's backward -- _ConvBnAct, the only producer fork2() is applied to -- adds them in the pass that reads the gradient anyway (the join: where
    the ReLU mask is applied, viai_add_act_bwd_from_output).  The autograd engine would have spent a pass of its own on the sum."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            return (g1 if g2 is None else g2), None
        if g1.shape != g2.shape or g1.dtype != torch.float32 or g2.dtype != torch.float32:
            return g1 + g2, None
        if ctx.slot.addend is not None:                      # a second backward through the same graph before the producer consumed the first addend
            g2 = g2 + ctx.slot.addend
        ctx.slot.addend = _c(g2)
        return _c(g1), None


def fork2(x):
    """two handles of x for its two readers, when x comes straight out of conv_bn_act (whose backward takes the gradient as two addends) and the
    sum is worth fusing; (x, x) otherwise.  The producer's node and the fork share an _AddendSlot; a tensor whose node is not a _ConvBnAct node, or that
    was forked before, is left to autograd's own sum."""
    if not (LAZY_SUM and getattr(x, "_viai_lazy_sum_ok", False) and x.requires_grad and x.is_cuda):
        return x, x
    node = x.grad_fn
    if not isinstance(node, _ConvBnAct._backward_cls) or getattr(node, "_viai_addend_slot", None) is not None:
        return x, x
    slot = _AddendSlot()
    node._viai_addend_slot = slot
    a, b = _Fork2.apply(x, slot)
    for t in (a, b):                                      # the tags conv_bn_act left for the next layer
        for k in ("_viai_amax", "_viai_twin", "_viai_p16"):
            v = getattr(x, k, None)
            if v is not None:
                setattr(t, k, v)
    return a, b


def _take_addend(ctx):
    """the second addend of a gradient that arrives lazily summed (fork2) at the node `ctx`, or None"""
    slot = getattr(ctx, "_viai_addend_slot", None)
    if slot is None:
        return None
    a, slot.addend = slot.addend, None
    return a


class _AddRelu(torch.autograd.Function):
    """out = relu(a + b): the BasicBlock join (networks/ResNet.py:51-52)."""

    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        _require(a, b)
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        _lib.check(lib.viai_add_relu_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "viai_add_relu_fwd")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (out,) = ctx.saved_tensors
        g = _c(g)
        d = torch.empty_like(g)
        _lib.check(lib.viai_relu_bwd(g.data_ptr(), out.data_ptr(), d.data_ptr(), g.numel(), _stream()), "viai_relu_bwd")
        return d, d


def add_relu(a, b):
    out = _AddRelu.apply(a, b)
    ma, mb = amax_of(a), amax_of(b)
    if ma is not None and mb is not None:
        out._viai_amax = ma + mb                        # |relu(a + b)| <= max |a| + max |b|
    return out


def frames_to_nhwc4(x):
    """(N, C<=4, H, W) NCHW frames -> (N, H, W, 4) NHWC, zero-padded channels (no gradient: loader output)."""
    lib = _lib.load()
    _require(x)
    x = _c(x.detach())
    N, Cc, H, W = x.shape
    y = torch.empty((N, H, W, 4), device=x.device, dtype=torch.float32)
    za = _amax_slot(x.device)                   # max |x| on the way through: the stem conv's operand scale (otherwise a pass of its own over the frames)
    _lib.check(lib.viai_nchw_to_nhwc4_amax(x.data_ptr(), y.data_ptr(), N, Cc, H * W, za.data_ptr(), _stream()), "viai_nchw_to_nhwc4_amax")
    y._viai_amax = za
    return y


class _L2Contrastive(torch.autograd.Function):
    """L2ContrastiveLoss.forward (loss_functions.py:125-148)."""

    @staticmethod
    def forward(ctx, f1, f2, margin, max_violation):
        lib = _lib.load()
        _require(f1, f2)
        f1, f2 = _c(f1), _c(f2)
        n, d = f1.shape
        scores = torch.empty((n, n), device=f1.device, dtype=torch.float32)
        loss = torch.empty((), device=f1.device, dtype=torch.float32)
        _lib.check(lib.viai_l2c_fwd(f1.data_ptr(), f2.data_ptr(), n, d, margin, 1 if max_violation else 0,
                                    scores.data_ptr(), loss.data_ptr(), _stream()), "viai_l2c_fwd")
        ctx.save_for_backward(f1, f2, scores)
        ctx.margin, ctx.mv = margin, bool(max_violation)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        f1, f2, scores = ctx.saved_tensors
        n, d = f1.shape
        g = _c(g)
        arg = torch.empty(n, device=f1.device, dtype=torch.int32)
        df1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        df2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        _lib.check(lib.viai_l2c_bwd(f1.data_ptr(), f2.data_ptr(), scores.data_ptr(), n, d, ctx.margin, 1 if ctx.mv else 0,
                                    g.data_ptr(), arg.data_ptr(), _ptr(df1), _ptr(df2), _stream()), "viai_l2c_bwd")
        return df1, df2, None, None


def l2_contrastive(f1, f2, margin=0.0, max_violation=False):
    return _L2Contrastive.apply(f1, f2, float(margin), bool(max_violation))


class _MaskMul(torch.autograd.Function):
    """s_in = s * mask, mask (N,T) broadcast over F (the missing AudioModel.set_inputs)."""

    @staticmethod
    def forward(ctx, s, mask):
        lib = _lib.load()
        _require(s, mask)
        s, mask = _c(s), _c(mask)
        N, T = mask.shape[0], mask.shape[-1]
        F = s.numel() // (N * T)
        out = torch.empty_like(s)
        _lib.check(lib.viai_mask_mul(s.data_ptr(), mask.data_ptr(), out.data_ptr(), N, F, T, _stream()), "viai_mask_mul")
        ctx.save_for_backward(mask)
        ctx.dims = (N, F, T)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        N, F, T = ctx.dims
        g = _c(g)
        ds = torch.empty_like(g)
        _lib.check(lib.viai_mask_mul(g.data_ptr(), mask.data_ptr(), ds.data_ptr(), N, F, T, _stream()), "viai_mask_mul")
        return ds, None


def mask_mul(s, mask):
    """s: (N,1,F,T) or (N,F,T); mask: (N,1,1,T) or (N,T)."""
    return inherit_amax(_MaskMul.apply(s, mask.reshape(mask.shape[0], mask.shape[-1])), s)      # mask in {0, 1}


def adam_step(p, g, m, v, state, beta1, beta2, eps, grad_scale=1.0):
    """In-place torch.optim.Adam step on flat fp32 arenas; `state` = 4 fp64 {step, lr, b1^t, b2^t} on device."""
    lib = _lib.load()
    _require(p, g, m, v)
    assert state.dtype == torch.float64 and state.numel() == 4
    _lib.check(lib.viai_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), state.data_ptr(),
                                  beta1, beta2, eps, grad_scale, _stream()), "viai_adam_step")
