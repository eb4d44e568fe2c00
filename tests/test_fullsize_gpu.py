"""Full-size checks (BASELINE.json configs[1]: batch 16, 256 x 256 mel): the kernels the benchmark times, at the shapes it times.

The layers of the benchmark size select other kernel instances than the cfg 0 / cfg 1 shapes (128 x 256 wide halo tile, patch-staged
stride-2 forward / data gradient, register-filter halo kernel, split-K, row-run streaming kernels, all-taps weight gradient ...).
They are pinned three ways:

* VALUES, layer by layer: forward, data gradient and weight gradient of every distinct layer shape of the cfg-2 step against
  torch.nn.functional on the CPU in fp64 (the reference's own ops: Discriminator_Networks.py:37-50, New_Inpainting_Networks.py:70-89,
  Inpainting_Networks.py:69-78), <= 3e-6 relative -- `test_full_size_layer_values_against_fp64`;
* the whole no-update step against digests produced by the REFERENCE's modules at 16 x 256 x 256 (tests/golden/step_cfg2.npz,
  tools/make_goldens.py --cfg2-only) -- tests/test_networks_gpu.py::test_step_no_update_matches_reference_golden_at_benchmark_size;
* identities that hold for any size:

* adjointness: a convolution is bilinear in (x, w), so  <y, gy> = <x, dgrad(gy)> = <w, wgrad(x, gy)>  exactly -- this ties
  the forward, data-gradient and weight-gradient kernels of a layer to each other (eval-mode BatchNorm behind the conv keeps
  the map affine and routes the backward through the abs-max-scaled f16x2 gradient kernels);
* the three-stream step is bit-identical to the single-stream step at the benchmark size too;
* eval-mode G is a per-clip function: 16 clips at once = two batches of 8.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import viai_oracle as O


def dot(a, b):
    return (a.double().reshape(-1) * b.double().reshape(-1)).sum().item()


# (name, N, H, W, Cin, Cout, kernel, stride, pad, transposed): the distinct kernel classes of the cfg-2 step at their real sizes
FULL_LAYERS = [
    ("D.conv3 (128x256 wide tile, f16x2 wgrad)", 16, 64, 32, 256, 512, (3, 3), (1, 1), (1, 1), False),
    ("D.conv2_2 (fused stride-2 dgrad)", 16, 128, 64, 128, 256, (3, 3), (2, 2), (1, 1), False),
    ("D.conv2_1 (stride-2 forward on the parity-sub-patch kernel, 128 outputs)", 16, 256, 128, 64, 128, (3, 3), (2, 2), (1, 1), False),
    ("G 32->32 @256x256 (register-filter halo, all-taps wgrad)", 16, 256, 256, 32, 32, (3, 3), (1, 1), (1, 1), True),
    ("G 64->64 @64x128 (streamed-filter halo)", 16, 64, 128, 64, 64, (3, 3), (1, 1), (1, 1), True),
    ("E deep 512->512 @8x8 (split-K)", 16, 8, 8, 512, 512, (3, 3), (1, 1), (1, 1), False),
    ("G 128->64 @32x64 (64x64 LDS-weight tile)", 16, 32, 64, 128, 64, (3, 3), (1, 1), (1, 1), True),
    ("G.conv6_2 32->1 @256x256 (row-run streaming)", 16, 256, 256, 32, 1, (3, 3), (1, 1), (1, 1), True),
    ("D.conv1 1->64 1x4 s(1,2) (Cin = 1 streaming)", 16, 256, 256, 1, 64, (1, 4), (1, 2), (0, 1), False),
]

# the remaining layer shapes of the cfg-2 step (SURVEY.md section 8a per-layer table), so that EVERY conv launch the benchmark
# times has a value test at its own shape; (.., C2) = channels of the virtually concatenated second source
MORE_LAYERS = [
    ("E.conv1 1->32 s2 @256x256 (Cin = 1 streaming)", 16, 256, 256, 1, 32, (3, 3), (2, 2), (1, 1), False, 0),
    ("E.conv2 32->64 s(2,1) @128x128", 16, 128, 128, 32, 64, (3, 3), (2, 1), (1, 1), False, 0),
    ("E.conv3 64->128 s2 @64x128", 16, 64, 128, 64, 128, (3, 3), (2, 2), (1, 1), False, 0),
    ("E.conv4 128->256 s2 @32x64", 16, 32, 64, 128, 256, (3, 3), (2, 2), (1, 1), False, 0),
    ("E.conv5 256->256 s2 @16x32", 16, 16, 32, 256, 256, (3, 3), (2, 2), (1, 1), False, 0),
    ("G.deconv1_1 256->256 pad(0,1) @2x16", 16, 2, 16, 256, 256, (3, 3), (1, 1), (0, 1), True, 0),
    ("G.deconv1_2 256->256 @4x16", 16, 4, 16, 256, 256, (3, 3), (1, 1), (1, 1), True, 0),
    ("G.cb2_0 256->128 @16x32", 16, 16, 32, 256, 128, (3, 3), (1, 1), (1, 1), True, 0),
    ("G.cb2_1 128->128 @16x32", 16, 16, 32, 128, 128, (3, 3), (1, 1), (1, 1), True, 0),
    ("G.cb3_0 128->64 @32x64", 16, 32, 64, 128, 64, (3, 3), (1, 1), (1, 1), True, 0),
    ("G.cb4_0 cat(64,64)->32 @64x128 (virtual concat, two dgrad destinations)", 16, 64, 128, 64, 32, (3, 3), (1, 1), (1, 1), True, 64),
    ("G.cb4_1 32->32 @64x128", 16, 64, 128, 32, 32, (3, 3), (1, 1), (1, 1), True, 0),
    ("G.cb5 32->32 @128x128", 16, 128, 128, 32, 32, (3, 3), (1, 1), (1, 1), True, 0),
    ("D.conv4 512->1 @64x32 (Cout = 1 streaming)", 16, 64, 32, 512, 1, (3, 3), (1, 1), (1, 1), False, 0),
]


@pytest.mark.parametrize("case", FULL_LAYERS, ids=[c[0].split(" (")[0] for c in FULL_LAYERS])
@pytest.mark.parametrize("bn", [False, True], ids=["conv", "conv+bn_eval"])
def test_forward_dgrad_wgrad_are_adjoint_at_full_size(case, bn):
    from viai_amd import ops
    name, N, H, W, Ci, Co, k, s, p, tr = case
    if bn and (Co == 1 or Ci == 1):
        pytest.skip("no BatchNorm behind the streaming layers in the reference")
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = (torch.rand(N, H, W, Ci, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    wshape = (Ci, Co) + k if tr else (Co, Ci) + k
    w = ((torch.rand(*wshape, device="cuda", generator=g) - 0.5) * (2.0 / (Ci * k[0] * k[1]) ** 0.5)).requires_grad_(True)
    norm = None
    if bn:
        norm = torch.nn.BatchNorm2d(Co).cuda()
        with torch.no_grad():
            norm.weight.copy_(torch.rand(Co, device="cuda", generator=g) + 0.5)
            norm.bias.copy_(torch.rand(Co, device="cuda", generator=g) - 0.5)
            norm.running_mean.copy_(torch.rand(Co, device="cuda", generator=g) * 0.2 - 0.1)
            norm.running_var.copy_(torch.rand(Co, device="cuda", generator=g) + 0.5)
        norm.eval()
    ops.begin_step(x.device)
    y = ops.conv_bn_act(x, w, None, norm, kernel=k, stride=s, padding=p, transposed=tr, act=ops.ACT_NONE, training=False)
    gy = torch.rand(y.shape, device="cuda", generator=g) * 2 - 1
    y.backward(gy)
    torch.cuda.synchronize()
    if bn:      # y = a_c * conv(x, w) + b_c: remove the offset, the rest is bilinear in (x, w)
        a_c = norm.weight / torch.sqrt(norm.running_var + norm.eps)
        b_c = norm.bias - norm.running_mean * a_c
        lhs = dot(y.detach() - b_c, gy)
    else:
        lhs = dot(y.detach(), gy)
    dx, dw = dot(x.detach(), x.grad), dot(w.detach(), w.grad)
    scale = (y.detach().double().norm() * gy.double().norm()).item()
    assert abs(lhs - dx) < 2e-5 * scale, (name, lhs, dx, scale)
    assert abs(lhs - dw) < 2e-5 * scale, (name, lhs, dw, scale)


def _cpu_layer64(x, w, k, s, p, tr, gy, bn=None):
    """the reference's op (F.conv2d / F.conv_transpose2d [+ eval BatchNorm]) in fp64 on the CPU; returns y, dx, dw (NCHW)."""
    import torch.nn.functional as F
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv_transpose2d(xd, wd, None, stride=s, padding=p) if tr else F.conv2d(xd, wd, None, stride=s, padding=p)
    if bn is not None:
        g_, b_, rm, rv = (t.double() for t in bn)
        y = F.batch_norm(y, rm, rv, g_, b_, False, 0.1, 1e-5)
    y.backward(gy.double())
    return y.detach(), xd.grad, wd.grad


def _relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _worst_tile(a, b, th=8, tw=16):
    """largest relative error of any th x tw output tile of an NCHW tensor (a corrupted tile cannot hide in the global norm)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    N, C, H, W = a.shape
    if H % th or W % tw:
        return _relerr(a, b)
    d = (a - b).pow(2).reshape(N, C, H // th, th, W // tw, tw).sum(dim=(1, 3, 5))
    r = b.pow(2).reshape(N, C, H // th, th, W // tw, tw).sum(dim=(1, 3, 5))
    return float((d / (r + 1e-30)).sqrt().max())


@pytest.mark.parametrize("case", [c for c in FULL_LAYERS + MORE_LAYERS if c[7] == (2, 2) and c[5] % 128 == 0 and c[4] >= 64],
                         ids=lambda c: c[0].split(" (")[0] + " [wgrad_bf3]")
def test_full_size_stride2_layers_without_the_patch_weight_gradient(case, monkeypatch):
    """VIAI_WGRAD_PATCH_S2=0: the stride-2 layers fall back to wgrad_bf3_kernel (the round-1 kernel stays covered at the benchmark shapes)"""
    monkeypatch.setenv("VIAI_WGRAD_PATCH_S2", "0")
    test_full_size_layer_values_against_fp64(case, True)


@pytest.mark.parametrize("case", FULL_LAYERS + MORE_LAYERS, ids=[c[0].split(" (")[0] for c in FULL_LAYERS + MORE_LAYERS])
@pytest.mark.parametrize("bn", [False, True], ids=["conv", "conv+bn_eval"])
def test_full_size_layer_values_against_fp64(case, bn):
    """VALUES of y, dx, dw at the benchmark shapes against the reference's torch ops evaluated in fp64 on the CPU.  With the
    eval-mode BatchNorm behind the conv the backward runs through the abs-max-scaled f16x2 data- and weight-gradient kernels,
    exactly as in the timed step; without it through the bf16x3 / streaming ones."""
    from viai_amd import ops
    name, N, H, W, C1, Co, k, s, p, tr = case[:10]
    C2 = case[10] if len(case) > 10 else 0
    Ci = C1 + C2
    if bn and Co == 1:
        pytest.skip("no BatchNorm behind the Cout = 1 layers in the reference")
    tag = "fsv.%d.%d.%d.%d" % (H, W, Ci, Co)
    x = O.cf_uniform(tag + ".x", (N, Ci, H, W), -1, 1)
    wshape = (Ci, Co) + k if tr else (Co, Ci) + k
    w = O.cf_std(tag + ".w", wshape, 1.0 / (Ci * k[0] * k[1]) ** 0.5)
    bnp = None
    if bn:
        bnp = (O.cf_uniform(tag + ".g", (Co,), 0.5, 1.5), O.cf_uniform(tag + ".b", (Co,), -0.5, 0.5),
               O.cf_uniform(tag + ".rm", (Co,), -0.1, 0.1), O.cf_uniform(tag + ".rv", (Co,), 0.5, 1.5))
    norm = None
    if bn:
        norm = torch.nn.BatchNorm2d(Co).cuda()
        with torch.no_grad():
            norm.weight.copy_(bnp[0]); norm.bias.copy_(bnp[1]); norm.running_mean.copy_(bnp[2]); norm.running_var.copy_(bnp[3])
        norm.eval()
    xg = x[:, :C1].permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    xg2 = x[:, C1:].permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True) if C2 else None
    wg = w.cuda().requires_grad_(True)
    ops.begin_step(xg.device)
    yg = ops.conv_bn_act(xg, wg, None, norm, kernel=k, stride=s, padding=p, transposed=tr, act=ops.ACT_NONE, training=False, x2=xg2)
    gy = O.cf_uniform(tag + ".gy", (N, Co, yg.shape[1], yg.shape[2]), -1, 1)
    yg.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    torch.cuda.synchronize()
    y, dx, dw = _cpu_layer64(x, w, k, s, p, tr, gy, bnp)
    y_hip = yg.detach().permute(0, 3, 1, 2)
    dx_hip = xg.grad.permute(0, 3, 1, 2)
    if C2:
        dx_hip = torch.cat((dx_hip, xg2.grad.permute(0, 3, 1, 2)), 1)
    assert tuple(y_hip.shape) == tuple(y.shape), name
    TOL = 3e-6
    assert _relerr(y_hip, y) < TOL, (name, "y", _relerr(y_hip, y))
    assert _relerr(dx_hip, dx) < TOL, (name, "dx", _relerr(dx_hip, dx))
    assert _relerr(wg.grad, dw) < TOL, (name, "dw", _relerr(wg.grad, dw))
    # tile-local: no 8 x 16 output tile (the unit every patch-staged kernel works in) may be off either
    assert _worst_tile(y_hip, y) < 10 * TOL, (name, "y tile", _worst_tile(y_hip, y))
    assert _worst_tile(dx_hip, dx) < 10 * TOL, (name, "dx tile", _worst_tile(dx_hip, dx))


def _full_model(monkeypatch, wgrad, dreal):
    from viai_amd.model import AudioModel, StepConfig
    monkeypatch.setenv("VIAI_WGRAD_STREAM", wgrad)
    monkeypatch.setenv("VIAI_DREAL_STREAM", dreal)
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = 256, 256
    m = AudioModel(hp, device="cuda", use_graph=False)
    m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
    return m


def test_three_stream_step_is_bitwise_serial_at_benchmark_size(monkeypatch):
    from viai_amd import synth
    s = synth.mel_batch(16, 256, 256, "full.s", 0).cuda()
    mask = synth.time_mask(16, 256, "full.mask", 0).cuda()

    def run(wgrad, dreal):
        m = _full_model(monkeypatch, wgrad, dreal)
        m.set_inputs(s, mask)
        for i in range(2):
            m.optimize_parameters(i)
        torch.cuda.synchronize()
        out = [m.fake.detach().clone(), m.losses.clone(), m.arena_D.grad.clone(), m.arena_G.grad.clone(), m.arena_G.flat.clone()]
        del m
        return out
    serial, streams = run("0", "0"), run("1", "1")
    for a, b in zip(serial, streams):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)


def test_three_stream_step_is_run_to_run_deterministic_at_benchmark_size(monkeypatch):
    """Three fresh models, one no-update step each on the same clips with all three streams on: gradient arenas and `fake` must be
    BIT-identical from run to run.  (tools/step_determinism.py; this is the check that exposed the packed-FMA miscompute of the first
    LDS-staged Cin = 1 BatchNorm-backward kernel -- correct alone, wrong for a few lanes per launch next to the stride-2 patch weight
    gradient on the other stream, csrc/conv_direct.hip cin1_lds_taps.)"""
    from viai_amd import synth
    s = synth.mel_batch(16, 256, 256, "det.s", 0).cuda()
    mask = synth.time_mask(16, 256, "det.mask", 0).cuda()
    outs = []
    for _ in range(3):
        m = _full_model(monkeypatch, "1", "1")
        m.set_inputs(s, mask)
        m.forward_backward_no_update()
        torch.cuda.synchronize()
        outs.append([m.fake.detach().clone(), m.arena_D.grad.clone(), m.arena_G.grad.clone()])
        del m
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_eval_discriminator_is_per_clip_at_benchmark_size():
    """running-statistics BatchNorm makes D a per-clip map: the 16-clip forward (128 x 256 tiles, 256+ tiles per layer) must agree
    with two 8-clip forwards (different tile counts, so partly different kernels) to fp32 rounding."""
    from viai_amd import synth
    from viai_amd.networks import MelDiscriminator
    D = MelDiscriminator().cuda()
    sd = O.disc_state()
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = O.cf_uniform("fs." + k, tuple(sd[k].shape), 0.5, 1.5)
        if k.endswith("running_mean"):
            sd[k] = O.cf_uniform("fs." + k, tuple(sd[k].shape), -0.2, 0.2)
    D.load_state_dict(sd)
    D.eval()
    x = synth.mel_batch(16, 256, 256, "full.s", 0).cuda()
    with torch.no_grad():
        whole = D(x)
        halves = torch.cat([D(x[:8]), D(x[8:])], 0)
    assert torch.isfinite(whole).all()
    err = ((whole.double() - halves.double()).norm() / whole.double().norm()).item()
    assert err < 2e-5, err


def _full_av_model(monkeypatch, wgrad_stream, dreal_stream, num_D):
    from viai_amd.model import AudioModel, StepConfig
    monkeypatch.setenv("VIAI_WGRAD_STREAM", wgrad_stream)
    monkeypatch.setenv("VIAI_DREAL_STREAM", dreal_stream)
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = 256, 256
    hp.use_video, hp.num_D, hp.lambda_contrast = True, num_D, 0.1
    m = AudioModel(hp, device="cuda")
    m.load_states(O.encoder_state(), O.decoder_variant_state("image"), O.msd_state(num_D) if num_D > 1 else O.disc_state(),
                  O.image_embedding2_state())
    return m


def test_vision_infused_step_is_bitwise_serial_at_full_size(monkeypatch):
    """BASELINE.json configs[2] / [3] at their real size -- 16 clips of 256 x 256, 64 video + 64 flow frames of 224 x 224 per clip
    (1024 + 1024 frames through two ResNet-18s, Image_Embedding.py:187-200), 3-scale D: two full steps (both Adam updates) on three
    streams are bit-identical to the single-stream steps, and everything is finite.  The ResNet branch at this size takes kernels no
    other test reaches (partial 8 x 16 tiles on the 56 / 28 / 14 / 7-pixel maps at 1024 frames, full-CU weight-gradient grids)."""
    from viai_amd import synth
    B, NF = 16, 64
    s = synth.mel_batch(B, 256, 256, "fullav.s", 0).cuda()
    mask = synth.time_mask(B, 256, "fullav.mask", 0).cuda()
    video = synth.uniform("fullav.video", (B, NF, 3, 224, 224), -1, 1).cuda()
    flow = synth.uniform("fullav.flow", (B, NF, 2, 224, 224), -1, 1).cuda()

    def run(wgrad, dreal):
        m = _full_av_model(monkeypatch, wgrad, dreal, 3)
        m.set_inputs(s, mask, video=video, flow=flow)
        for i in range(2):
            m.optimize_parameters(i)
        v = m.get_loss_items()
        torch.cuda.synchronize()
        out = [m.fake.detach().clone(), m.losses.clone(), m.arena_D.grad.clone(), m.arena_G.grad.clone(), m.arena_G.flat.clone()]
        m.close()
        del m
        torch.cuda.empty_cache()
        return out, v
    (serial, v0), (streams, v1) = run("0", "0"), run("1", "1")
    assert all(x == x for x in v0) and v0[5] > 0                 # the contrastive term is live
    for a, b in zip(serial, streams):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)


def test_eval_image_embedding_is_per_clip_at_full_size():
    """eval-mode ImageEmbedding2 (running-statistics BatchNorm) is a per-clip map: 4 clips x 64 frames in one call (256 + 256 frames)
    agree with two 2-clip calls (different tile counts / grids in every layer) to fp32 rounding."""
    from viai_amd import synth
    from viai_amd.networks import ImageEmbedding2
    V = ImageEmbedding2().cuda()
    V.load_state_dict(O.image_embedding2_state())
    V.eval()
    video = synth.uniform("fullav.ev.video", (4, 64, 3, 224, 224), -1, 1).cuda()
    flow = synth.uniform("fullav.ev.flow", (4, 64, 2, 224, 224), -1, 1).cuda()
    with torch.no_grad():
        whole, fea = V(video, flow)
        h0, f0 = V(video[:2], flow[:2])
        h1, f1 = V(video[2:], flow[2:])
    assert tuple(whole.shape) == (4, 256, 1, 16) and tuple(fea.shape) == (4, 512, 64)
    for a, b in ((whole, torch.cat((h0, h1), 0)), (fea, torch.cat((f0, f1), 0))):
        err = ((a.double() - b.double()).norm() / a.double().norm()).item()
        assert torch.isfinite(a).all() and err < 2e-5, err
