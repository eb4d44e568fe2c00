"""GPU parity of the HIP kernels (through the C ABI / autograd ops) against the
reference's own torch CPU ops on identical inputs.  Tolerance: the north star's
1e-3 relative fp32; the kernels are exact-fp32 MFMA so most checks are ~1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import viai_oracle as O


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def nhwc(t):  # NCHW cpu -> NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):  # NHWC cuda -> NCHW cpu
    return t.detach().permute(0, 3, 1, 2).contiguous().cpu()


CONV_CASES = [
    # name, N, H, W, Cin, Cout, k, s, p, transposed, bias
    ("E.conv2", 2, 20, 24, 32, 64, (3, 3), (2, 1), (1, 1), False, False),
    ("E.conv3", 2, 17, 24, 64, 128, (3, 3), (2, 2), (1, 1), False, False),
    ("E.conv5", 1, 16, 32, 256, 256, (3, 3), (2, 2), (1, 1), False, False),
    ("G.deconv1_1", 2, 2, 8, 256, 256, (3, 3), (1, 1), (0, 1), True, True),
    ("G.cb2", 2, 6, 10, 256, 128, (3, 3), (1, 1), (1, 1), True, False),
    ("G.cb5", 2, 16, 12, 32, 32, (3, 3), (1, 1), (1, 1), True, False),
    ("G.conv6_1", 1, 24, 40, 32, 32, (3, 3), (1, 1), (1, 1), True, True),
    ("D.conv2_1", 2, 16, 20, 64, 128, (3, 3), (2, 2), (1, 1), False, False),
    ("D.conv3", 1, 9, 11, 256, 512, (3, 3), (1, 1), (1, 1), False, False),
    ("odd-sizes", 3, 7, 5, 64, 96, (3, 3), (2, 2), (1, 1), False, True),
    ("E.conv1", 2, 20, 24, 1, 32, (3, 3), (2, 2), (1, 1), False, False),
    ("D.conv1", 2, 12, 32, 1, 64, (1, 4), (1, 2), (0, 1), False, False),
    ("G.conv6_2", 2, 12, 20, 32, 1, (3, 3), (1, 1), (1, 1), True, True),
    ("D.conv4", 2, 8, 6, 512, 1, (3, 3), (1, 1), (1, 1), False, False),
    ("D.conv1-tiny", 2, 80, 32, 1, 64, (1, 4), (1, 2), (0, 1), False, False),
    ("G.conv6_2-tiny", 2, 80, 32, 32, 1, (3, 3), (1, 1), (1, 1), True, True),
    ("E.conv1-tiny", 2, 80, 32, 1, 32, (3, 3), (2, 2), (1, 1), False, False),
    ("G.cb3-tiny", 2, 10, 8, 128, 64, (3, 3), (1, 1), (1, 1), True, False),
    ("D.conv1-wide", 1, 8, 300, 1, 64, (1, 4), (1, 2), (0, 1), False, False),
    # LDS-resident-tile ("halo") kernel shapes: 32/64 channels, stride 1, extent a multiple of 8 x 16
    ("halo32", 2, 16, 32, 32, 32, (3, 3), (1, 1), (1, 1), False, True),
    ("halo32T", 2, 24, 16, 32, 32, (3, 3), (1, 1), (1, 1), True, False),
    ("halo64", 1, 8, 48, 64, 64, (3, 3), (1, 1), (1, 1), False, False),
    ("halo32to64", 2, 16, 16, 32, 64, (3, 3), (1, 1), (1, 1), False, True),
    ("halo64to32T", 2, 8, 32, 64, 32, (3, 3), (1, 1), (1, 1), True, True),
    ("halo48out", 1, 16, 16, 32, 48, (3, 3), (1, 1), (1, 1), False, False),
    ("halo1x3", 2, 8, 64, 32, 32, (1, 3), (1, 1), (0, 1), False, False),
    # round 6: the stride-(2, 1) layer on the halo kernel -- forward with vertical gather stride 2, data gradient as two row-parity classes with scatter stride 2
    ("halo_s21", 2, 32, 32, 32, 64, (3, 3), (2, 1), (1, 1), False, False),
    ("halo_s21_bias", 1, 48, 16, 32, 64, (3, 3), (2, 1), (1, 1), False, True),
    # big enough that the 64x64-tile LDS-weight kernel (not the small-M split-K kernel) takes the layer
    # fused-class stride-2 dgrad (needs >= 256 tiles of 128 base pixels x 64 channels): two channel blocks / ragged M
    ("D.conv2_2", 4, 128, 128, 128, 64, (3, 3), (2, 2), (1, 1), False, False),
    ("s2-ragged-M", 3, 214, 206, 64, 32, (3, 3), (2, 2), (1, 1), False, True),
    ("lds64x64", 4, 64, 64, 64, 128, (3, 3), (1, 1), (1, 1), False, False),
    # all-taps weight-gradient kernel of the <= 32 x <= 32 channel stride-1 layers (output width a multiple of 32)
    ("wg32", 2, 8, 64, 32, 32, (3, 3), (1, 1), (1, 1), False, True),
    ("wg32T", 3, 5, 96, 32, 32, (3, 3), (1, 1), (1, 1), True, False),
    ("wg32-16to24", 1, 4, 32, 16, 24, (3, 3), (1, 1), (1, 1), False, False),
    ("lds64x64-s2T", 2, 128, 64, 128, 96, (3, 3), (1, 1), (1, 1), True, True),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd_matches_torch_cpu(case):
    from viai_amd import ops
    name, N, H, W, Cin, Cout, k, s, p, tr, use_bias = case
    x = O.cf_uniform(name + ".x", (N, Cin, H, W), -1, 1).requires_grad_(True)
    wshape = (Cin, Cout) + k if tr else (Cout, Cin) + k
    w = O.cf_std(name + ".w", wshape, 0.1).requires_grad_(True)
    b = O.cf_uniform(name + ".b", (Cout,), -0.5, 0.5).requires_grad_(True) if use_bias else None
    if tr:
        y = F.conv_transpose2d(x, w, b, stride=1, padding=p)
    else:
        y = F.conv2d(x, w, b, stride=s, padding=p)
    gy = O.cf_uniform(name + ".gy", tuple(y.shape), -1, 1)
    y.backward(gy)

    xg = nhwc(x.detach()).requires_grad_(True)
    wg = w.detach().cuda().requires_grad_(True)
    bg = b.detach().cuda().requires_grad_(True) if use_bias else None
    yg = ops.conv_bn_act(xg, wg, bg, None, kernel=k, stride=s, padding=p, transposed=tr, act=ops.ACT_NONE)
    assert tuple(nchw(yg).shape) == tuple(y.shape)
    assert relerr(nchw(yg), y) < 2e-5, name
    yg.backward(nhwc(gy))
    assert relerr(nchw(xg.grad), x.grad) < 2e-5, name
    assert relerr(wg.grad, w.grad) < 2e-5, name
    if use_bias:
        assert relerr(bg.grad, b.grad) < 2e-5, name


@pytest.mark.parametrize("shape", [(2, 9, 12), (3, 64, 128)], ids=["small", "wide-halo-tiles"])
def test_conv_virtual_concat_matches_cat(shape):
    """second shape: 192 tiles of 8 x 16 pixels, so the wide halo kernel takes the forward (two sources, 32 outputs) and -- behind a
    BatchNorm in the networks -- the two-destination data gradient; without BN here the backward runs the bf16x3 kernels"""
    from viai_amd import ops
    N, H, W = shape
    x1 = O.cf_uniform("cc.x1", (N, 64, H, W), -1, 1).requires_grad_(True)
    x2 = O.cf_uniform("cc.x2", (N, 64, H, W), -1, 1).requires_grad_(True)
    w = O.cf_std("cc.w", (128, 32, 3, 3), 0.1).requires_grad_(True)
    y = F.conv_transpose2d(torch.cat((x1, x2), 1), w, None, stride=1, padding=1)
    gy = O.cf_uniform("cc.gy", tuple(y.shape), -1, 1)
    y.backward(gy)
    a, b = nhwc(x1.detach()).requires_grad_(True), nhwc(x2.detach()).requires_grad_(True)
    wg = w.detach().cuda().requires_grad_(True)
    yg = ops.conv_bn_act(a, wg, None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1), transposed=True, x2=b)
    assert relerr(nchw(yg), y) < 2e-5
    yg.backward(nhwc(gy))
    assert relerr(nchw(a.grad), x1.grad) < 2e-5
    assert relerr(nchw(b.grad), x2.grad) < 2e-5
    assert relerr(wg.grad, w.grad) < 2e-5


@pytest.mark.parametrize("co", [128, 256])
def test_stride2_forward_patch_kernel_against_fp64(co):
    """3 x 3 stride-2 conv at 4 x 128 x 256 x 64 -> co (256 tiles of 8 x 16 outputs): the parity-sub-patch instance of the wide halo
    kernel computes the forward (with BatchNorm partials: train-mode BN behind it), checked against fp64."""
    from viai_amd import ops
    N, Ci, H, W = 4, 64, 128, 256
    x = O.cf_uniform("s2f.x", (N, Ci, H, W), -1, 1)
    w = O.cf_std("s2f.w", (co, Ci, 3, 3), 0.05)
    truth = F.conv2d(x.double(), w.double(), None, stride=2, padding=1)
    y = ops.conv_bn_act(nhwc(x), w.cuda(), None, None, kernel=(3, 3), stride=(2, 2), padding=(1, 1))
    assert relerr(nchw(y), truth) < 3e-6
    bn = torch.nn.BatchNorm2d(co).cuda()
    yb = ops.conv_bn_act(nhwc(x), w.cuda(), None, bn, kernel=(3, 3), stride=(2, 2), padding=(1, 1))
    tb = F.batch_norm(truth, None, None, None, None, True, 0.1, 1e-5)
    assert relerr(nchw(yb), tb) < 1e-5


def test_patch_staged_stride2_dgrad_against_fp64():
    """3 x 3 stride-2 conv + BatchNorm(eval), 4 x 128 x 128 x 128 -> 64: 256 blocks of the fused-class data gradient, base lattice
    64 x 64 = a multiple of the 8 x 16 tile, so the patch-staged f16x2 kernel (conv_dgrad_s2_patch_kernel) computes dx."""
    from viai_amd import ops
    N, Ci, Co, H, W = 4, 128, 64, 128, 128
    x = O.cf_uniform("s2p.x", (N, Ci, H, W), -1, 1)
    w = O.cf_std("s2p.w", (Co, Ci, 3, 3), 0.05)
    g_, b_ = O.cf_uniform("s2p.g", (Co,), 0.5, 1.5), O.cf_uniform("s2p.b", (Co,), -0.5, 0.5)
    rm, rv = O.cf_uniform("s2p.rm", (Co,), -0.1, 0.1), O.cf_uniform("s2p.rv", (Co,), 0.5, 1.5)
    gy = O.cf_uniform("s2p.gy", (N, Co, H // 2, W // 2), -1, 1)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.batch_norm(F.conv2d(xd, wd, None, stride=2, padding=1), rm.double(), rv.double(), g_.double(), b_.double(), False, 0.1, 1e-5)
    y.backward(gy.double())
    bn = torch.nn.BatchNorm2d(Co).cuda()
    with torch.no_grad():
        bn.weight.copy_(g_); bn.bias.copy_(b_); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.eval()
    a = nhwc(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    ops.begin_step(a.device)
    yg = ops.conv_bn_act(a, wg, None, bn, kernel=(3, 3), stride=(2, 2), padding=(1, 1), act=ops.ACT_NONE, training=False)
    assert relerr(nchw(yg), y) < 3e-6
    yg.backward(nhwc(gy))
    assert relerr(nchw(a.grad), xd.grad) < 3e-6
    assert relerr(wg.grad, wd.grad) < 3e-6


@pytest.mark.parametrize("cfg", [(1, 128, 0, 256, False), (1, 64, 64, 128, True), (2, 64, 0, 128, False), (2, 96, 0, 256, False),
                                 (1, 32, 0, 32, True), (1, 64, 64, 32, True), (1, 64, 0, 64, False), (1, 96, 0, 160, True)],
                         ids=["s1_128to256", "s1_cat64+64to128_T", "s2_64to128", "s2_96to256",
                              "narrow_32to32_T", "narrow_cat64+64to32_T", "narrow_64to64", "narrow_96to160_T"])
def test_patch_staged_weight_gradient_against_fp64(cfg):
    """conv_wgrad_patch.hip (all nine taps per block, fragments through ds_read_b64_tr_b16): weight gradient of 3 x 3 layers with
    Cout % 128 == 0 behind an eval-mode BatchNorm (the abs-max-scaled f16x2 path), stride 1 (plain and transposed, one and two
    concatenated sources: 128 x 64 tile, eight waves; narrow layers: 32 x 32 tile whose four waves split the tile rows and write a slab
    each) and stride 2 (128 x 32 tile, parity sub-patches), 2 x (64 x 64 output) = 64 tiles;
    gradients of constant-free random tensors against an fp64 evaluation."""
    from viai_amd import ops
    S, C1, C2, Co, tr = cfg
    N, OHW = 2, 64
    H = W = OHW * S
    Ci = C1 + C2
    x = O.cf_uniform("wgp.x", (N, Ci, H, W), -1, 1)
    w = O.cf_std("wgp.w", (Ci, Co, 3, 3) if tr else (Co, Ci, 3, 3), 0.05)
    g_, b_ = O.cf_uniform("wgp.g", (Co,), 0.5, 1.5), O.cf_uniform("wgp.b", (Co,), -0.5, 0.5)
    rm, rv = O.cf_uniform("wgp.rm", (Co,), -0.1, 0.1), O.cf_uniform("wgp.rv", (Co,), 0.5, 1.5)
    gy = O.cf_uniform("wgp.gy", (N, Co, OHW, OHW), -1, 1) * 1e-3          # small gradients: the dynamic scale has work to do
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    conv = F.conv_transpose2d(xd, wd, None, stride=1, padding=1) if tr else F.conv2d(xd, wd, None, stride=S, padding=1)
    y = F.batch_norm(conv, rm.double(), rv.double(), g_.double(), b_.double(), False, 0.1, 1e-5)
    y.backward(gy.double())
    bn = torch.nn.BatchNorm2d(Co).cuda()
    with torch.no_grad():
        bn.weight.copy_(g_); bn.bias.copy_(b_); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.eval()
    a = nhwc(x[:, :C1]).requires_grad_(True)
    a2 = nhwc(x[:, C1:]).requires_grad_(True) if C2 else None
    wg = w.cuda().requires_grad_(True)
    ops.begin_step(a.device)
    yg = ops.conv_bn_act(a, wg, None, bn, kernel=(3, 3), stride=(S, S), padding=(1, 1), transposed=tr, x2=a2, act=ops.ACT_NONE, training=False)
    d = ops.conv_desc(N, H, W, C1, C2, Co, 3, 3, S, S, 1, 1, 1 if tr else 0)
    from viai_amd import _lib
    assert _lib.load().viai_conv2d_wgrad_f16_ok(d["ref"]) == 1
    yg.backward(nhwc(gy))
    assert relerr(nchw(yg), y) < 3e-6
    assert relerr(wg.grad, wd.grad) < 3e-6
    # every filter position separately (a tap / slot mix-up hides in the global norm of a smooth filter gradient)
    for ky in range(3):
        for kx in range(3):
            assert relerr(wg.grad[:, :, ky, kx], wd.grad[:, :, ky, kx]) < 5e-6, (ky, kx)


@pytest.mark.parametrize("cfg", [(1, 8, 28, 28, 128, 128), (2, 8, 28, 28, 64, 128), (1, 12, 20, 24, 64, 64), (1, 128, 7, 7, 128, 128), (1, 129, 7, 7, 64, 64),
                                 (1, 160, 5, 6, 128, 64)],
                         ids=["s1_28x28_128to128", "s2_to28x28_64to128", "narrow_20x24_64to64", "s1_7x7_128to128_pairs", "s1_7x7_64to64_odd_batch", "s1_5x6_64to128_pairs"])
def test_patch_weight_gradient_on_maps_that_are_not_tile_multiples(cfg):
    """the output maps of the ResNet-18 visual branch (56 / 28 / 14 / 7 pixels) are not multiples of the 8 x 16 tile: the patch weight
    gradient covers them with partial tiles whose outside pixels are staged as zeros (networks/Image_Embedding.py:13-71 shapes)."""
    from viai_amd import _lib, ops
    S, N, OH, OW, Ci, Co = cfg
    H, W = OH * S, OW * S
    x = O.cf_uniform("wgq.x", (N, Ci, H, W), -1, 1)
    w = O.cf_std("wgq.w", (Co, Ci, 3, 3), 0.05)
    g_, b_ = O.cf_uniform("wgq.g", (Co,), 0.5, 1.5), O.cf_uniform("wgq.b", (Co,), -0.5, 0.5)
    rm, rv = O.cf_uniform("wgq.rm", (Co,), -0.1, 0.1), O.cf_uniform("wgq.rv", (Co,), 0.5, 1.5)
    gy = O.cf_uniform("wgq.gy", (N, Co, OH, OW), -1, 1)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.batch_norm(F.conv2d(xd, wd, None, stride=S, padding=1), rm.double(), rv.double(), g_.double(), b_.double(), False, 0.1, 1e-5)
    y.backward(gy.double())
    bn = torch.nn.BatchNorm2d(Co).cuda()
    with torch.no_grad():
        bn.weight.copy_(g_); bn.bias.copy_(b_); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.eval()
    a = nhwc(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    ops.begin_step(a.device)
    yg = ops.conv_bn_act(a, wg, None, bn, kernel=(3, 3), stride=(S, S), padding=(1, 1), act=ops.ACT_NONE, training=False)
    yg.backward(nhwc(gy))
    assert relerr(nchw(yg), y) < 3e-6
    assert relerr(nchw(a.grad), xd.grad) < 3e-6
    assert relerr(wg.grad, wd.grad) < 3e-6
    for ky in range(3):
        for kx in range(3):
            assert relerr(wg.grad[:, :, ky, kx], wd.grad[:, :, ky, kx]) < 5e-6, (ky, kx)


@pytest.mark.parametrize("cfg", [(128, 0, 32), (64, 64, 32), (32, 0, 128), (96, 0, 64), (64, 64, 128)], ids=["128to32", "cat64+64to32", "32to128", "96to64", "cat64+64to128"])
def test_wide_halo_kernel_fwd_and_f16_backward_against_fp64(cfg):
    """stride-1 3 x 3 transposed conv + BatchNorm(eval) at 3 x 64 x 128 (192 tiles): forward, data gradient (one or two
    destinations) and weight gradient of the layers the wide halo kernel takes, against an fp64 evaluation."""
    from viai_amd import ops
    C1, C2, Co = cfg
    N, H, W = 3, 64, 128
    x1 = O.cf_uniform("wh.x1", (N, C1, H, W), -1, 1)
    x2 = O.cf_uniform("wh.x2", (N, C2, H, W), -1, 1) if C2 else None
    w = O.cf_std("wh.w", (C1 + C2, Co, 3, 3), 0.05)
    g_, b_ = O.cf_uniform("wh.g", (Co,), 0.5, 1.5), O.cf_uniform("wh.b", (Co,), -0.5, 0.5)
    rm, rv = O.cf_uniform("wh.rm", (Co,), -0.1, 0.1), O.cf_uniform("wh.rv", (Co,), 0.5, 1.5)
    gy = O.cf_uniform("wh.gy", (N, Co, H, W), -1, 1)
    xs = [t.double().requires_grad_(True) for t in ((x1, x2) if C2 else (x1,))]
    wd = w.double().requires_grad_(True)
    y = F.batch_norm(F.conv_transpose2d(torch.cat(xs, 1), wd, None, stride=1, padding=1), rm.double(), rv.double(), g_.double(), b_.double(), False, 0.1, 1e-5)
    y.backward(gy.double())
    bn = torch.nn.BatchNorm2d(Co).cuda()
    with torch.no_grad():
        bn.weight.copy_(g_); bn.bias.copy_(b_); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.eval()
    a = nhwc(x1).requires_grad_(True)
    b2 = nhwc(x2).requires_grad_(True) if C2 else None
    wg = w.cuda().requires_grad_(True)
    ops.begin_step(a.device)
    yg = ops.conv_bn_act(a, wg, None, bn, kernel=(3, 3), stride=(1, 1), padding=(1, 1), transposed=True, x2=b2, act=ops.ACT_NONE, training=False)
    assert relerr(nchw(yg), y) < 3e-6
    yg.backward(nhwc(gy))
    assert relerr(nchw(a.grad), xs[0].grad) < 3e-6
    if C2:
        assert relerr(nchw(b2.grad), xs[1].grad) < 3e-6
    assert relerr(wg.grad, wd.grad) < 3e-6


@pytest.mark.parametrize("act", ["lrelu", "relu"])
@pytest.mark.parametrize("shape", [(2, 32, 20, 24), (3, 64, 7, 9), (1, 256, 4, 16), (2, 32, 80, 32), (4, 32, 64, 96)])
def test_conv_bn_act_train_matches_torch_cpu(act, shape):
    """conv -> BatchNorm2d(train) -> activation, forward, backward, running stats."""
    from viai_amd import ops
    N, C, H, W = shape
    Cout = 64
    x = O.cf_uniform("bn.x", shape, -1, 1).requires_grad_(True)
    w = O.cf_std("bn.w", (Cout, C, 3, 3), 0.1).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(Cout)
    bn.weight.data.copy_(O.cf_uniform("bn.g", (Cout,), 0.5, 1.5))
    bn.bias.data.copy_(O.cf_uniform("bn.b", (Cout,), -0.5, 0.5))
    bn.train()
    y = bn(F.conv2d(x, w, None, stride=1, padding=1))
    y = F.leaky_relu(y, 0.2) if act == "lrelu" else F.relu(y)
    gy = O.cf_uniform("bn.gy", tuple(y.shape), -1, 1)
    y.backward(gy)

    bng = torch.nn.BatchNorm2d(Cout).cuda()
    bng.weight.data.copy_(O.cf_uniform("bn.g", (Cout,), 0.5, 1.5))
    bng.bias.data.copy_(O.cf_uniform("bn.b", (Cout,), -0.5, 0.5))
    bng.train()
    xg = nhwc(x.detach()).requires_grad_(True)
    wg = w.detach().cuda().requires_grad_(True)
    yg = ops.conv_bn_act(xg, wg, None, bng, kernel=(3, 3), stride=(1, 1), padding=(1, 1),
                         act=ops.ACT_LRELU if act == "lrelu" else ops.ACT_RELU)
    assert relerr(nchw(yg), y) < 1e-4
    yg.backward(nhwc(gy))
    assert relerr(nchw(xg.grad), x.grad) < 1e-3
    assert relerr(wg.grad, w.grad) < 1e-3
    assert relerr(bng.weight.grad, bn.weight.grad) < 1e-3
    assert relerr(bng.bias.grad, bn.bias.grad) < 1e-3
    assert relerr(bng.running_mean, bn.running_mean) < 1e-5
    assert relerr(bng.running_var, bn.running_var) < 1e-5
    assert int(bng.num_batches_tracked) == 1


@pytest.mark.parametrize("need_dx", [False, "fused", "apply"])
@pytest.mark.parametrize("geom", [(3, 80, 96, 32, (3, 3), (2, 2), (1, 1)), (2, 64, 72, 64, (1, 4), (1, 2), (0, 1)), (5, 33, 37, 32, (3, 3), (2, 2), (1, 1)),
                                  (2, 32, 128, 64, (1, 4), (1, 2), (0, 1)), (2, 64, 64, 32, (3, 3), (2, 2), (1, 1))])
def test_cin1_conv_bn_layer_without_the_stored_preactivation(geom, need_dx, monkeypatch):
    """E.conv1 / D.conv1 (Inpainting_Networks.py:55,71; Discriminator_Networks.py:17-19): Cin = 1 conv -> BatchNorm2d(train) ->
    LeakyReLU on the fused path (viai_conv2d_cin1_bn_*: the conv output is never stored, forward and backward recompute it from x;
    the weight gradient forms dy on the fly; the data gradient either reads a dy tensor written by the recomputing apply pass ("apply":
    the default for 3 x 3 windows) or forms dy on the fly too ("fused", VIAI_CIN1_BN_DGRAD=1: the default for one-row windows -- D.conv1's
    1 x 4 -- whose block-of-whole-rows shapes take the single-pass cin1_bn_dgrad_rows_kernel, here the fourth geometry; the other
    geometries take the input-pixel-mapped kernel, slower than "apply" and opt-in).  Forward, dw, dgamma, dbeta,
    dx and the running statistics against fp64 within 5x of torch-CPU-fp32's own rounding error; the third geometry has a ragged last
    statistics block (pixels not a multiple of 256); the last two have blocks of whole output rows, i.e. the kernels that stage the
    block's input rows in LDS (the benchmark shapes take that path)."""
    from viai_amd import ops, _lib
    monkeypatch.setenv("VIAI_CIN1_BN_DGRAD", "0" if need_dx == "apply" else "1")
    N, H, W, Co, k, s_, p_ = geom
    x = O.cf_uniform("c1.x", (N, 1, H, W), 0, 1)
    w = O.cf_std("c1.w", (Co, 1) + k, 0.3)
    g = O.cf_uniform("c1.g", (Co,), 0.8, 1.2)
    b = O.cf_uniform("c1.b", (Co,), -0.1, 0.1)

    def run(dt):
        xs, ws, gs, bs = [t.clone().to(dt).requires_grad_(True) for t in (x, w, g, b)]
        z = F.leaky_relu(F.batch_norm(F.conv2d(xs, ws, None, stride=s_, padding=p_), None, None, gs, bs, True, 0.1, 1e-5), 0.2)
        gy = O.cf_uniform("c1.gy", tuple(z.shape), -1, 1).to(dt)
        return (z,) + torch.autograd.grad(z, [xs, ws, gs, bs], grad_outputs=gy), gy
    (truth, _), (cpu32, gy) = run(torch.float64), run(torch.float32)
    bn = torch.nn.BatchNorm2d(Co).cuda().train()
    bn.weight.data.copy_(g); bn.bias.data.copy_(b)
    xg = nhwc(x).requires_grad_(bool(need_dx))
    wg = w.cuda().requires_grad_(True)
    d = ops.conv_desc(N, H, W, 1, 0, Co, k[0], k[1], s_[0], s_[1], p_[0], p_[1], 0, 1, 1, -1, -1)
    assert _lib.load().viai_conv2d_cin1_bn_ok(d["ref"]) == 1
    zg = ops.conv_bn_act(xg, wg, None, bn, kernel=k, stride=s_, padding=p_, act=ops.ACT_LRELU)
    zg.backward(nhwc(gy))
    hip = (nchw(zg), nchw(xg.grad) if need_dx else None, wg.grad, bn.weight.grad, bn.bias.grad)
    for nm, h, c32, t in zip(("z", "dx", "dw", "dgamma", "dbeta"), hip, cpu32, truth):
        if h is not None:
            assert relerr(h, t) < 5 * relerr(c32, t) + 1e-6, (nm, relerr(h, t), relerr(c32, t))
    ref = torch.nn.BatchNorm2d(Co).double().train()
    ref(F.conv2d(x.double(), w.double(), None, stride=s_, padding=p_))
    assert relerr(bn.running_mean, ref.running_mean) < 1e-5 and relerr(bn.running_var, ref.running_var) < 1e-5


@pytest.mark.parametrize("shape", [(2, 128, 10, 8, 64), (2, 256, 5, 4, 128), (2, 64, 20, 16, 128), (4, 32, 40, 32, 32)])
def test_fused_layer_accuracy_against_fp64(shape):
    """one ConvTranspose2d -> BN(train) -> ReLU layer, forward and all gradients, measured against an fp64
    evaluation: the HIP kernels must be within 5x of torch-CPU-fp32's own rounding error (+1e-6)."""
    from viai_amd import ops
    N, C, H, W, Co = shape
    x = O.cf_uniform("fa.x", (N, C, H, W), 0, 1)
    w = O.cf_std("fa.w", (C, Co, 3, 3), 0.05)
    gy = O.cf_uniform("fa.gy", (N, Co, H, W), -1, 1)
    g = O.cf_uniform("fa.g", (Co,), 0.8, 1.2)
    b = O.cf_uniform("fa.b", (Co,), -0.1, 0.1)

    def run(dt):
        xs, ws, gs, bs = [t.clone().to(dt).requires_grad_(True) for t in (x, w, g, b)]
        z = F.relu(F.batch_norm(F.conv_transpose2d(xs, ws, None, stride=1, padding=1), None, None, gs, bs, True, 0.1, 1e-5))
        return (z,) + torch.autograd.grad(z, [xs, ws, gs, bs], grad_outputs=gy.to(dt))
    truth, cpu32 = run(torch.float64), run(torch.float32)
    bn = torch.nn.BatchNorm2d(Co).cuda().train()
    bn.weight.data.copy_(g); bn.bias.data.copy_(b)
    xg = nhwc(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    zg = ops.conv_bn_act(xg, wg, None, bn, kernel=(3, 3), stride=(1, 1), padding=(1, 1), transposed=True, act=ops.ACT_RELU)
    zg.backward(nhwc(gy))
    hip = (nchw(zg), nchw(xg.grad), wg.grad, bn.weight.grad, bn.bias.grad)
    for nm, h, c32, t in zip(("z", "dx", "dw", "dgamma", "dbeta"), hip, cpu32, truth):
        assert relerr(h, t) < 5 * relerr(c32, t) + 1e-6, (nm, relerr(h, t), relerr(c32, t))


def test_bn_large_mean_is_stable():
    """variance of a channel with mean >> std (the E[x^2]-E[x]^2 trap)."""
    from viai_amd import ops
    N, C, H, W = 4, 32, 64, 64
    x = O.cf_uniform("bnl.x", (N, C, H, W), -1, 1)
    w = torch.zeros(32, C, 3, 3)
    for i in range(32):
        w[i, i, 1, 1] = 1e-2          # y ~ 1e-2 * x
    b = torch.full((32,), 5.0)         # + 5: fp32 E[x^2]-E[x]^2 would be ~6% off here
    bn = torch.nn.BatchNorm2d(32).train()
    y = bn(F.conv2d(x, w, b, padding=1))
    bng = torch.nn.BatchNorm2d(32).cuda().train()
    yg = ops.conv_bn_act(nhwc(x), w.cuda(), b.cuda(), bng, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_NONE)
    assert relerr(nchw(yg), y) < 1e-3
    assert relerr(bng.running_var, bn.running_var) < 1e-4


@pytest.mark.parametrize("C", [32, 24])
@pytest.mark.parametrize("sizes", [((4, 16), (16, 32)), ((3, 8), (7, 20)), ((64, 128), (128, 128)), ((5, 7), (5, 7)), ((9, 9), (4, 5)), ((4, 4), (32, 30)),
                                   ((40, 6), (5, 6))])
def test_bilinear_ac_matches_torch_cpu(sizes, C):
    """F.interpolate(mode="bilinear", align_corners=True) forward and backward (New_Inpainting_Networks.py:78,83).  C = 32 takes the per-pixel
    kernels (a channel quad per thread, exact candidate ranges, <= 6 column weights held in registers; (4, 4) -> (32, 30) has more columns per
    input pixel than that and takes their generic inner loop), C = 24 (six quads: not a power of two) the index-decoding kernels."""
    from viai_amd import ops
    (ih, iw), (oh, ow) = sizes
    x = O.cf_uniform("bl.x", (2, C, ih, iw), -1, 1).requires_grad_(True)
    y = F.interpolate(x, size=[oh, ow], mode="bilinear", align_corners=True)
    gy = O.cf_uniform("bl.gy", tuple(y.shape), -1, 1)
    y.backward(gy)
    xg = nhwc(x.detach()).requires_grad_(True)
    yg = ops.bilinear_ac(xg, (oh, ow))
    assert relerr(nchw(yg), y) < 5e-6
    yg.backward(nhwc(gy))
    assert relerr(nchw(xg.grad), x.grad) < 1e-5


def test_avgpool_h_matches_torch_cpu():
    from viai_amd import ops
    for ih in (8, 5, 3):
        x = O.cf_uniform("ap.x", (2, 256, ih, 16), -1, 1).requires_grad_(True)
        y = F.avg_pool2d(x, (3, 1))
        gy = O.cf_uniform("ap.gy", tuple(y.shape), -1, 1)
        y.backward(gy)
        xg = nhwc(x.detach()).requires_grad_(True)
        yg = ops.avgpool_h(xg, 3)
        assert relerr(nchw(yg), y) < 1e-6
        yg.backward(nhwc(gy))
        assert relerr(nchw(xg.grad), x.grad) < 1e-6
        x.grad = None


def test_losses_match_torch_cpu(golden_dir):
    from viai_amd import ops
    p = O.cf_uniform("ls.p", (4, 1, 32, 16), 0.001, 0.999).requires_grad_(True)
    s = O.cf_uniform("ls.s", (4, 1, 32, 16), 0, 1)
    for target in (0.0, 1.0, 0.9):
        for kind in ("bce", "mse"):
            ref = O.gan_loss(p, None, kind == "mse", real_label=target, fake_label=target) if False else None
            t = torch.full_like(p, target)
            ref = F.binary_cross_entropy(p, t) if kind == "bce" else F.mse_loss(p, t)
            (g_ref,) = torch.autograd.grad(ref * 3.0, p)
            pg = p.detach().cuda().requires_grad_(True)
            out = ops.bce_mean(pg, target) if kind == "bce" else ops.mse_mean(pg, target)
            (g,) = torch.autograd.grad(out * 3.0, pg)
            assert abs(out.item() - ref.item()) <= 1e-5 * abs(ref.item())
            assert relerr(g, g_ref) < 1e-5
    ref = F.l1_loss(p, s)
    (g_ref,) = torch.autograd.grad(ref, p)
    pg = p.detach().cuda().requires_grad_(True)
    out = ops.l1_mean(pg, s.cuda())
    (g,) = torch.autograd.grad(out, pg)
    assert abs(out.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert relerr(g, g_ref) < 1e-6
    # the reference GANLoss golden values incl. the log clamp at p in {0, 1}
    gold = np.load(golden_dir + "/layers.npz")
    pe = torch.tensor([[0.0, 1.0, 0.25, 0.999999, 1e-30, 0.5]]).cuda()
    assert abs(ops.bce_mean(pe, 1.0).item() - float(gold["gan_bce_real"])) < 1e-4 * float(gold["gan_bce_real"])
    assert abs(ops.bce_mean(pe, 0.0).item() - float(gold["gan_bce_fake"])) < 1e-4 * float(gold["gan_bce_fake"])
    assert abs(ops.mse_mean(pe, 1.0).item() - float(gold["gan_mse_real"])) < 1e-5
    assert abs(ops.mse_mean(pe, 0.0).item() - float(gold["gan_mse_fake"])) < 1e-5


def test_adam_matches_torch_optim_golden(golden_dir):
    from viai_amd import ops
    gold = np.load(golden_dir + "/adam.npz")
    c = O.StepConfig
    p = O.cf_uniform("adam.p", (4096,), -1, 1).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.tensor([0.0, c.lr, 1.0, 1.0], dtype=torch.float64, device="cuda")
    for t in range(3):
        g = (O.cf_uniform("adam.g%d" % t, (4096,), -1, 1) * (10.0 ** O.cf_uniform("adam.e%d" % t, (4096,), -9, 0))).cuda()
        ops.adam_step(p, g, m, v, state, c.beta1, c.beta2, c.eps)
        ref = torch.from_numpy(gold["p_after_%d" % (t + 1)])
        assert (p.cpu() - ref).abs().max().item() < 2e-7
    assert state[0].item() == 3.0


def test_mask_mul():
    from viai_amd import ops
    s = O.cf_uniform("mm.s", (3, 1, 20, 32), 0, 1)
    mask = O.make_mask(3, 32, "mm.mask")
    out = ops.mask_mul(s.cuda(), mask.cuda())
    assert torch.equal(out.cpu(), s * mask)


def test_ops_refuse_cpu_tensors():
    from viai_amd import ops, _lib
    with pytest.raises(_lib.ViaiLibraryError):
        ops.bilinear_ac(torch.zeros(1, 2, 2, 4), (4, 4))


@pytest.mark.parametrize("mv", [False, True])
def test_l2_contrastive_matches_reference_golden_and_autograd(mv, golden_dir):
    from viai_amd import losses
    gold = np.load(golden_dir + "/layers.npz")
    f1 = O.cf_uniform("lg.f1", (6, 256), -1, 1)
    f2 = O.cf_uniform("lg.f2", (6, 256), -1, 1)
    crit = losses.L2ContrastiveLoss(margin=12.0, max_violation=mv)
    a, b = f1.cuda().requires_grad_(True), f2.cuda().requires_grad_(True)
    out = crit(a, b)
    ref = float(gold["l2c_mv%d" % mv])                       # the reference's own L2ContrastiveLoss value
    assert abs(out.item() - ref) < 1e-5 * abs(ref)
    (out * 2.0).backward()
    x, y = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    (O.l2_contrastive(x, y, 12.0, mv) * 2.0).backward()
    assert relerr(a.grad, x.grad) < 1e-5
    assert relerr(b.grad, y.grad) < 1e-5
    # bigger, non-trivial margin hits
    f1 = O.cf_uniform("l2c.f1", (52, 256), -1, 1)
    f2 = O.cf_uniform("l2c.f2", (52, 256), -1, 1)
    a, b = f1.cuda().requires_grad_(True), f2.cuda().requires_grad_(True)
    out = losses.L2ContrastiveLoss(margin=13.0, max_violation=mv)(a, b)
    x, y = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    ref = O.l2_contrastive(x, y, 13.0, mv)
    assert abs(out.item() - ref.item()) < 1e-5 * abs(ref.item())
    out.backward(); ref.backward()
    assert relerr(a.grad, x.grad) < 1e-5 and relerr(b.grad, y.grad) < 1e-5


def test_ganloss_module_matches_reference_semantics():
    from viai_amd import losses
    p = O.cf_uniform("gl.p", (4, 1, 8, 4), 0.01, 0.99)
    for lsgan in (False, True):
        crit = losses.GANLoss(use_lsgan=lsgan)
        for real in (False, True):
            out = crit(p.cuda(), real)
            assert abs(out.item() - O.gan_loss(p, real, lsgan).item()) < 1e-5


def test_ganloss_soft_labels_match_reference_golden(golden_dir):
    """GANLoss(..., softlabel=True) (loss_functions.py:90-99): label = real - U(0, 0.1) / fake + U(0, 0.1) drawn from Python's `random`,
    one draw per call: with the generator seeded like the fixture's, the reference's five values per loss kind
    (tests/golden/ganloss_soft.npz, tools/make_goldens.py --ganloss-soft-only), call for call."""
    import random
    from viai_amd import losses
    gold = np.load(golden_dir + "/ganloss_soft.npz")
    p = O.cf_uniform("gl.p", (4, 1, 8, 4), 0.01, 0.99).cuda()
    for lsgan in (False, True):
        crit = losses.GANLoss(use_lsgan=lsgan)
        random.seed(20260929)
        vals = [crit(p, real, softlabel=True).item() for real in (True, False, True, True, False)]
        assert np.allclose(vals, gold["lsgan%d" % int(lsgan)], rtol=2e-6, atol=0), (vals, gold["lsgan%d" % int(lsgan)])
        assert len(set(np.round(vals, 6))) == 5                 # five different labels: the draws really happen


def test_bf16x3_path_is_fp32_grade():
    """Large layers run on the split-bf16 ("bf16x3") MFMA kernel (csrc/conv_igemm_bf3.hip).  Its error against an
    fp64 evaluation must stay at the fp32 level: within 4x of the CPU fp32 convolution's own rounding error
    (the exact-fp32 MFMA kernel, a strictly sequential fmaf chain over K, sits at ~2-4x itself)."""
    from viai_amd import ops
    N, C, H, W, Co = 2, 128, 256, 256, 128           # 1024 tiles of 128x128, forward AND data gradient -> bf16x3 path
    x = O.cf_uniform("b3.x", (N, C, H, W), -1, 1)
    w = O.cf_std("b3.w", (Co, C, 1, 3), 0.05)
    gy = O.cf_uniform("b3.gy", (N, Co, H, W), -1, 1)

    def run(dt):
        xs, ws = x.clone().to(dt).requires_grad_(True), w.clone().to(dt).requires_grad_(True)
        y = F.conv2d(xs, ws, None, stride=1, padding=(0, 1))
        gx, gw = torch.autograd.grad(y, [xs, ws], grad_outputs=gy.to(dt))
        return y.detach(), gx, gw
    truth, cpu32 = run(torch.float64), run(torch.float32)
    xg = nhwc(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    yg = ops.conv_bn_act(xg, wg, None, None, kernel=(1, 3), stride=(1, 1), padding=(0, 1))
    yg.backward(nhwc(gy))
    for nm, h, c32, t in zip(("fwd", "dgrad", "wgrad"), (nchw(yg), nchw(xg.grad), wg.grad), cpu32, truth):
        e, c = relerr(h, t), relerr(c32, t)
        print("%s: hip err %.2e, cpu fp32 err %.2e" % (nm, e, c))
        assert e < 4 * c + 1e-7, (nm, e, c)


@pytest.mark.parametrize("scale", [1.0, 0.01, 100.0])
def test_f16x2_forward_is_fp32_grade(scale):
    """Wide forward layers run the f16x2 split (two fp16 terms per operand, three partial products, power-of-two
    pre-scaling): against an fp64 evaluation the result must be as accurate as fp32 arithmetic, for activations of
    ordinary size and two decades either side.  Shape chosen so the fragment-major 128x128 kernel takes the layer."""
    from viai_amd import ops
    N, H, W, Ci, Co = 16, 64, 64, 64, 128
    x = (O.cf_uniform("f16.x", (N, Ci, H, W), -1, 2).clamp(min=0) * scale)           # ReLU-like: a third zeros
    w = O.cf_std("f16.w", (Co, Ci, 3, 3), 0.05)
    truth = F.conv2d(x.double(), w.double(), None, padding=1)
    cpu32 = F.conv2d(x, w, None, padding=1)
    y = ops.conv_bn_act(nhwc(x), w.cuda(), None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
    e_hip, e_cpu = relerr(nchw(y), truth), relerr(cpu32, truth)
    assert e_hip < 3 * e_cpu + 1e-7, (scale, e_hip, e_cpu)


def test_f16x2_forward_on_the_128x256_tile():
    """256-multiple output channels and >= 256 tiles: the eight-wave 128 x 256 variant of the wide kernel takes the layer
    (conv_igemm_bf3.hip, viai_conv_igemm_bf3_launch).  Same fp32-grade bound against fp64, with a BatchNorm partial-statistics
    epilogue checked through the normalised output."""
    from viai_amd import ops
    N, H, W, Ci, Co = 8, 64, 64, 64, 256
    x = O.cf_uniform("w4.x", (N, Ci, H, W), -1, 1)
    w = O.cf_std("w4.w", (Co, Ci, 3, 3), 0.05)
    truth = F.conv2d(x.double(), w.double(), None, padding=1)
    cpu32 = F.conv2d(x, w, None, padding=1)
    y = ops.conv_bn_act(nhwc(x), w.cuda(), None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
    e_hip, e_cpu = relerr(nchw(y), truth), relerr(cpu32, truth)
    assert e_hip < 3 * e_cpu + 1e-7, (e_hip, e_cpu)
    bn = torch.nn.BatchNorm2d(Co).cuda()
    yb = ops.conv_bn_act(nhwc(x), w.cuda(), None, bn, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
    tb = F.batch_norm(truth, None, None, None, None, True, 0.1, 1e-5)
    assert relerr(nchw(yb), tb) < 1e-5


@pytest.mark.parametrize("mag", [1.0e4, 3.0e-4])
def test_f16x2_operand_scale_follows_the_input_magnitude(mag):
    """The fp16 split pre-scales the activation operand by a power of two.  With a static scale (x16, round 1 / 2) inputs beyond
    |x| = 4094 saturated SILENTLY; now the scale is derived on the device from the magnitude the tensor carries (`_viai_amax`:
    reduced by its producer, or by one viai_absmax pass for a tensor this library did not produce -- the case here), so a layer fed
    |x| ~ 1e4, or ~ 3e-4, is as accurate against fp64 as the fp32 arithmetic allows: forward, data gradient and weight gradient."""
    from viai_amd import ops
    N, C, H, W, Co = 4, 64, 32, 64, 128
    x = O.cf_uniform("mag.x", (N, C, H, W), -1, 1) * mag
    w = O.cf_std("mag.w", (Co, C, 3, 3), 0.05)
    gy = O.cf_uniform("mag.gy", (N, Co, H, W), -1, 1)
    g = O.cf_uniform("mag.g", (Co,), 0.8, 1.2)

    def run(dt):
        xs, ws, gs = [t.clone().to(dt).requires_grad_(True) for t in (x, w, g)]
        y = F.conv2d(xs, ws, None, padding=1)
        z = F.batch_norm(y, None, None, gs, None, True, 0.1, 1e-5)
        return (y.detach(),) + torch.autograd.grad(z, [xs, ws], grad_outputs=gy.to(dt))
    truth, cpu32 = run(torch.float64), run(torch.float32)
    xg = nhwc(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    yg = ops.conv_bn_act(xg.detach(), wg.detach(), None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
    assert relerr(nchw(yg), truth[0]) < 1e-5 and relerr(nchw(yg), truth[0]) < 5 * relerr(cpu32[0], truth[0]) + 1e-7
    bn = torch.nn.BatchNorm2d(Co).cuda().train()
    bn.weight.data.copy_(g)
    zg = ops.conv_bn_act(xg, wg, None, bn, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_NONE)
    assert zg._viai_amax is not None and abs(float(zg._viai_amax) - float(zg.abs().max())) < 1e-6 * float(zg.abs().max())
    zg.backward(nhwc(gy))
    for nm, h, c32, t in zip(("dx", "dw"), (nchw(xg.grad), wg.grad), cpu32[1:], truth[1:]):
        assert relerr(h, t) < 5 * relerr(c32, t) + 1e-6, (mag, nm, relerr(h, t), relerr(c32, t))


def test_operand_magnitude_travels_with_the_tensors_of_a_network():
    """provenance of `_viai_amax` through a network: BatchNorm outputs reduce it on the way out, resampling and masking inherit it,
    the virtual concat takes the larger one, and no extra pass runs for tensors that carry it."""
    from viai_amd import networks as N_, ops
    E = N_.MelEncoder().cuda().train()
    G = N_.MelDecoder().cuda().train()
    E.load_state_dict(O.encoder_state()); G.load_state_dict(O.decoder_state())
    s = O.cf_uniform("s.tiny", (2, 80, 32)).cuda()
    calls = []
    lib = ops._lib.load()
    orig = lib.viai_absmax
    try:
        lib.viai_absmax = lambda *a: (calls.append(a), orig(*a))[1]
        feats = E.forward_nhwc(s)
        fake = G.forward_nhwc(feats, (80, 32))
    finally:
        lib.viai_absmax = orig
    assert calls == []                                     # every MFMA layer's input carried its magnitude
    for f in feats:
        assert abs(float(f._viai_amax) - float(f.abs().max())) <= 1e-6 * float(f.abs().max()) or float(f._viai_amax) >= float(f.abs().max())
    assert float(fake._viai_amax) == 1.0 and float(fake.max()) <= 1.0          # sigmoid output: bounded by construction
    up = ops.bilinear_ac(feats[2], (40, 20))
    assert up._viai_amax is feats[2]._viai_amax


@pytest.mark.parametrize("gscale", [1.0, 1e-6, 1e4])
def test_f16x2_backward_is_fp32_grade_at_any_gradient_scale(gscale):
    """The data- and weight-gradient kernels of the wide layers use the f16x2 split with a power-of-two operand scale
    derived on the device from max |dy| (written by the BatchNorm-backward kernel): gradients six decades smaller or four
    larger must come out as accurate, against fp64, as fp32 arithmetic makes them.  Conv -> BN(train) (no activation: a
    ReLU kink within rounding of zero flips a derivative and would measure luck, not kernels), 128 -> 128 channels at
    16 x 64 x 64 so that the wide-tile dgrad and the bf16x3-class wgrad kernels take the layer."""
    from viai_amd import ops
    N, C, H, W, Co = 16, 128, 64, 64, 128
    x = O.cf_uniform("fb.x", (N, C, H, W), 0, 1)
    w = O.cf_std("fb.w", (Co, C, 3, 3), 0.03)
    gy = O.cf_uniform("fb.gy", (N, Co, H, W), -1, 1) * gscale
    g = O.cf_uniform("fb.g", (Co,), 0.8, 1.2)
    b = O.cf_uniform("fb.b", (Co,), -0.1, 0.1)

    def run(dt):
        xs, ws, gs, bs = [t.clone().to(dt).requires_grad_(True) for t in (x, w, g, b)]
        z = F.batch_norm(F.conv2d(xs, ws, None, padding=1), None, None, gs, bs, True, 0.1, 1e-5)
        return torch.autograd.grad(z, [xs, ws], grad_outputs=gy.to(dt))
    truth, cpu32 = run(torch.float64), run(torch.float32)
    bn = torch.nn.BatchNorm2d(Co).cuda().train()
    bn.weight.data.copy_(g); bn.bias.data.copy_(b)
    xg = nhwc(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    zg = ops.conv_bn_act(xg, wg, None, bn, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_NONE)
    zg.backward(nhwc(gy))
    for nm, h, c32, t in zip(("dx", "dw"), (nchw(xg.grad), wg.grad), cpu32, truth):
        assert relerr(h, t) < 5 * relerr(c32, t) + 1e-6, (gscale, nm, relerr(h, t), relerr(c32, t))


def test_f16x2_range_guard_counts_saturating_operands():
    """debug mode (VIAI_DEBUG_RANGE / ops.DEBUG_RANGE): WEIGHTS beyond the fp16 range of the statically pre-scaled f16x2 weight images
    are counted (and refused in strict mode) instead of being clamped silently.  (The product model checks max |w| at its host sync
    point: tests/test_networks_gpu.py::test_weights_beyond_the_f16x2_range_raise_at_the_sync_point.)"""
    from viai_amd import ops
    x = O.cf_uniform("rg.x", (1, 16, 16, 32), -1, 1).cuda()
    w = O.cf_std("rg.w", (32, 32, 3, 3), 0.05).cuda()
    w[1, 2, 0, 0] = 300.0                        # > 65504 / 256
    old, olds = ops.DEBUG_RANGE, ops.DEBUG_RANGE_STRICT
    try:
        ops.DEBUG_RANGE, ops.DEBUG_RANGE_STRICT = True, False
        ops.range_report()
        y = ops.conv_bn_act(x, w, None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
        assert torch.isfinite(y).all()                                  # saturates, no NaN
        rep = ops.range_report()
        assert "activation" not in rep                                   # activations follow their magnitude: nothing to guard
        assert rep["weight"][0] == 1 and rep["weight"][1] == 300.0
        ops.conv_bn_act(x.clamp(-1, 1), w.clamp(-1, 1), None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
        rep = ops.range_report()
        assert rep["weight"][0] == 0
        ops.DEBUG_RANGE_STRICT = True
        with pytest.raises(FloatingPointError):
            ops.conv_bn_act(x, w, None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
    finally:
        ops.DEBUG_RANGE, ops.DEBUG_RANGE_STRICT = old, olds
        ops.range_report()


@pytest.mark.parametrize("case", ["G.conv6", "D.conv3-4", "D.eval", "wide-frozen"])
def test_fused_bn_cout1_pair_matches_the_two_layers(case):
    """(conv + BatchNorm + act) -> (3 x 3 conv to ONE channel) as one op (ops.conv_bn_act_cout1: the tensor between the layers and the
    second layer's data gradient are never stored; csrc/conv_direct.hip viai_pair_cout1_*) against the same two layers run one after the
    other AND against torch in fp64: G.conv6_1 + BN + ReLU -> conv6_2 + Sigmoid (transposed convs, 32 channels, biases;
    New_Inpainting_Networks.py:85-88), D.conv3 + BN + LeakyReLU -> conv4 + Sigmoid (512 channels; Discriminator_Networks.py:44-49),
    eval-mode BatchNorm, and a frozen pair (data gradient only: the G step's pass through D)."""
    from viai_amd import networks as N_, ops
    tr = case == "G.conv6"
    N, Ci, Cm, H, W = {"G.conv6": (2, 32, 32, 32, 64), "D.conv3-4": (2, 64, 512, 16, 32), "D.eval": (2, 32, 64, 8, 16),
                       "wide-frozen": (3, 128, 256, 8, 48)}[case]
    act = ops.ACT_RELU if tr else ops.ACT_LRELU
    mk = (lambda ci, co: torch.nn.ConvTranspose2d(ci, co, 3, 1, 1, bias=True)) if tr else (lambda ci, co: torch.nn.Conv2d(ci, co, 3, 1, 1, bias=False))
    conv1, conv2, bn = mk(Ci, Cm).cuda(), mk(Cm, 1).cuda(), torch.nn.BatchNorm2d(Cm).cuda()
    with torch.no_grad():
        conv1.weight.copy_(O.cf_std("pair.w1." + case, tuple(conv1.weight.shape), 0.08))
        conv2.weight.copy_(O.cf_std("pair.w2." + case, tuple(conv2.weight.shape), 0.1))
        bn.weight.copy_(O.cf_uniform("pair.g." + case, (Cm,), 0.7, 1.3)); bn.bias.copy_(O.cf_uniform("pair.b." + case, (Cm,), -0.2, 0.2))
        bn.running_mean.copy_(O.cf_uniform("pair.rm." + case, (Cm,), -0.1, 0.1)); bn.running_var.copy_(O.cf_uniform("pair.rv." + case, (Cm,), 0.5, 1.5))
        if tr:
            conv1.bias.copy_(O.cf_uniform("pair.b1", (Cm,), -0.1, 0.1)); conv2.bias.copy_(O.cf_uniform("pair.b2", (1,), -0.1, 0.1))
    if case == "D.eval":
        bn.eval()
    if case == "wide-frozen":
        for m in (conv1, conv2, bn):
            m.requires_grad_(False)
    x = O.cf_uniform("pair.x." + case, (N, Ci, H, W), -1, 1)
    gp = O.cf_uniform("pair.gp." + case, (N, 1, H, W), -1, 1)
    params = [p for p in list(conv1.parameters()) + list(bn.parameters()) + list(conv2.parameters()) if p.requires_grad]

    def run(fused):
        bn_state = {k: v.clone() for k, v in bn.state_dict().items()}
        for p in params:
            p.grad = None
        xg = nhwc(x).requires_grad_(True)
        old, ops.PAIR_FUSED = ops.PAIR_FUSED, fused
        try:
            p_ = N_.fused_pair(xg, conv1, bn, act, conv2, ops.ACT_SIGMOID)
        finally:
            ops.PAIR_FUSED = old
        p_.backward(nhwc(gp))
        out = [p_.detach().clone(), xg.grad.clone()] + [q.grad.clone() for q in params] + [bn.running_mean.clone(), bn.running_var.clone()]
        bn.load_state_dict(bn_state)
        return out
    calls = []
    lib = ops._lib.load()
    orig = lib.viai_pair_cout1_bn_bwd
    try:
        lib.viai_pair_cout1_bn_bwd = lambda *a: (calls.append(1), orig(*a))[1]
        fused = run(True)
    finally:
        lib.viai_pair_cout1_bn_bwd = orig
    assert calls == [1]                                   # the pair kernels really ran
    plain = run(False)
    assert len(fused) == len(plain)
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert relerr(a, b) < 2e-5, (case, i, relerr(a, b))
    # fp64 truth
    c1, c2, b64 = [m_.double().cpu() for m_ in (conv1, conv2, bn)]
    x64 = x.double().requires_grad_(True)
    z = torch.nn.functional.leaky_relu(b64(c1(x64)), 0.2) if not tr else torch.relu(b64(c1(x64)))
    p64 = torch.sigmoid(c2(z))
    p64.backward(gp.double())
    assert relerr(nchw(fused[0]), p64) < 1e-5 and relerr(nchw(fused[1]), x64.grad) < 5e-5
    conv1.float().cuda(); conv2.float().cuda(); bn.float().cuda()


@pytest.mark.parametrize("geom", [(3, 80, 96, 32, (3, 3), (2, 2), (1, 1)), (2, 64, 64, 32, (3, 3), (2, 2), (1, 1)), (5, 33, 37, 32, (3, 3), (2, 2), (1, 1))])
@pytest.mark.parametrize("need_dx", [False, True])
def test_time_mask_applied_where_the_first_conv_loads_its_input(geom, need_dx):
    """s_in = s * mask (the inpainting step's full-height time gap, misc/pipeline2.png) is not a pass of its own any more: the fused
    Cin = 1 layer (E.conv1 + bn1 + LeakyReLU, Inpainting_Networks.py:55,71) multiplies while it loads s, in its forward, its BatchNorm
    backward and its weight gradient (`x_mask` of viai_conv2d_cin1_bn_*, LDS-staged and global-load paths).  Against the same layer
    on an explicitly masked input -- bit for bit, the arithmetic is the same -- and the gradient w.r.t. the UNMASKED s."""
    from viai_amd import ops
    N, H, W, Co, k, s_, p_ = geom
    x = O.cf_uniform("mk.x", (N, 1, H, W), 0, 1)
    mask = (O.cf_uniform("mk.m", (N, 1, 1, W), 0, 1) > 0.3).float()
    w = O.cf_std("mk.w", (Co, 1) + k, 0.3)
    gy = None

    def run(fused_mask):
        nonlocal gy
        bn = torch.nn.BatchNorm2d(Co).cuda().train()
        xg = nhwc(x).requires_grad_(need_dx)
        wg = w.cuda().requires_grad_(True)
        calls = []
        lib = ops._lib.load()
        orig = lib.viai_mask_mul
        lib.viai_mask_mul = lambda *a: (calls.append(1), orig(*a))[1]
        try:
            if fused_mask:
                z = ops.conv_bn_act(xg, wg, None, bn, kernel=k, stride=s_, padding=p_, act=ops.ACT_LRELU, xmask=mask.cuda())
            else:
                z = ops.conv_bn_act(ops.mask_mul(xg, mask.cuda()), wg, None, bn, kernel=k, stride=s_, padding=p_, act=ops.ACT_LRELU)
            if gy is None:
                gy = O.cf_uniform("mk.gy", tuple(z.shape), -1, 1).cuda()
            z.backward(gy)
        finally:
            lib.viai_mask_mul = orig
        return [z.detach(), wg.grad, bn.weight.grad, bn.bias.grad, bn.running_var.clone()] + ([xg.grad] if need_dx else []), len(calls)
    (plain, n_plain), (fused, n_fused) = run(False), run(True)
    assert n_plain >= 1 and n_fused == (1 if need_dx else 0)          # no mask pass in the step's configuration (s needs no gradient)
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert torch.equal(a, b), (i, relerr(a, b))
    # and a layer the fused path does not take (eval-mode BatchNorm) still sees the masked input
    bn = torch.nn.BatchNorm2d(Co).cuda().eval()
    a = ops.conv_bn_act(nhwc(x), w.cuda(), None, bn, kernel=k, stride=s_, padding=p_, act=ops.ACT_LRELU, xmask=mask.cuda(), training=False)
    b = ops.conv_bn_act(ops.mask_mul(nhwc(x), mask.cuda()), w.cuda(), None, bn, kernel=k, stride=s_, padding=p_, act=ops.ACT_LRELU, training=False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("cfg", [(8, 56, 56, 64, 64), (8, 28, 28, 128, 128), (16, 14, 14, 256, 256), (4, 80, 104, 64, 128), (24, 22, 44, 128, 32), (16, 20, 26, 256, 512)],
                         ids=["56x56_64to64", "28x28_128to128", "14x14_256to256", "80x104_64to128", "22x44_128to32", "20x26_256to512"])
def test_wide_halo_kernel_on_maps_with_partial_tiles(cfg):
    """the 8 x 16 tiles of the wide halo kernel clipped at the map's edge (the 56 / 28 / 14-pixel maps of networks/Image_Embedding.py:13-71,
    the 80 x 104 maps of the reference's native 80 x 208 clips): conv -> BatchNorm(train) -> ReLU with the batch statistics merged from
    tile-shaped partial blocks of unequal size, forward + running statistics + every gradient against an fp64 evaluation."""
    from viai_amd import _lib, ops
    import ctypes as C
    N, H, W, Ci, Co = cfg
    d = ops.conv_desc(N, H, W, Ci, 0, Co, 3, 3, 1, 1, 1, 1, 0)
    assert d["tiles"] == (8, 16) and d["nblk"] == N * ((H + 7) // 8) * ((W + 15) // 16)
    x = O.cf_uniform("pt.x", (N, Ci, H, W), -1, 1)
    w = O.cf_std("pt.w", (Co, Ci, 3, 3), 0.05)
    gy = O.cf_uniform("pt.gy", (N, Co, H, W), -1, 1)
    g = O.cf_uniform("pt.g", (Co,), 0.8, 1.2)
    b = O.cf_uniform("pt.b", (Co,), -0.1, 0.1)

    def run(dt):
        xs, ws, gs, bs = [t.clone().to(dt).requires_grad_(True) for t in (x, w, g, b)]
        rm, rv = torch.zeros(Co, dtype=dt), torch.ones(Co, dtype=dt)
        z = F.relu(F.batch_norm(F.conv2d(xs, ws, None, stride=1, padding=1), rm, rv, gs, bs, True, 0.1, 1e-5))
        return (z,) + torch.autograd.grad(z, [xs, ws, gs, bs], grad_outputs=gy.to(dt)) + (rm, rv)
    truth, cpu32 = run(torch.float64), run(torch.float32)
    bn = torch.nn.BatchNorm2d(Co).cuda().train()
    bn.weight.data.copy_(g); bn.bias.data.copy_(b)
    xg = nhwc(x).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    ops.begin_step(xg.device)
    zg = ops.conv_bn_act(xg, wg, None, bn, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_RELU)
    buf = C.create_string_buffer(64)
    zg.backward(nhwc(gy))
    hip = (nchw(zg), nchw(xg.grad), wg.grad, bn.weight.grad, bn.bias.grad, None, bn.running_var)
    for nm, h, c32, t in zip(("z", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var"), hip, cpu32, truth):
        if h is not None:
            assert relerr(h, t) < 5 * relerr(c32, t) + 1e-6, (nm, relerr(h, t), relerr(c32, t))
    # the batch mean is ~1e-3 of the batch deviation here: its error is measured against the deviation, not against itself
    sd = ((truth[6] - 0.9) / 0.1).sqrt()
    assert float(((bn.running_mean.cpu().double() - truth[5]).abs() / (0.1 * sd)).max()) < 1e-6
    # the forward ran on the wide halo kernel (not on the gather kernel these shapes used before)
    ops.begin_step(xg.device)
    with torch.no_grad():
        ops.conv_bn_act(xg, wg, None, bn, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_RELU, training=False)
    _lib.load().viai_conv2d_last_kernel(buf, 64)
    assert buf.value.decode().startswith("halo_wide"), buf.value


def test_tiled_batchnorm_merge_matches_the_uniform_merge():
    """viai_bn_finalize_tiles on a map that IS a whole number of tiles equals viai_bn_finalize on the same partials bit for bit"""
    from viai_amd import _lib
    lib = _lib.load()
    N, OH, OW, Cc = 3, 16, 32, 8
    nblk = N * 2 * 2
    part = torch.randn(2 * Cc * nblk, device="cuda").abs_()
    gam, bet = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda")
    outs = []
    for tiled in (False, True):
        rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
        coef = torch.empty(4, Cc, device="cuda")
        tail = (Cc, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0, 0.1, 1e-5, coef[0].data_ptr(), coef[1].data_ptr(),
                coef[2].data_ptr(), coef[3].data_ptr(), 0)
        if tiled:
            _lib.check(lib.viai_bn_finalize_tiles(part.data_ptr(), N, OH, OW, 8, 16, *tail), "tiles")
        else:
            _lib.check(lib.viai_bn_finalize(part.data_ptr(), nblk, 128, N * OH * OW, *tail), "uniform")
        torch.cuda.synchronize()
        outs.append((coef.clone(), rm.clone(), rv.clone()))
    for u, t in zip(*outs):
        assert torch.equal(u, t)


@pytest.mark.parametrize("cin,transposed", [(32, 1), (128, 0), (256, 0), (512, 0), (512, 1)])
def test_pair_forward_through_the_tap_products_equals_the_lds_plane_kernel(cin, transposed):
    """viai_pair_cout1_fwd_dots (one grid-stride pass over y leaving nine tap products per pixel + a gather: the form D.conv3 -> conv4 takes,
    Discriminator_Networks.py:44-50) against viai_pair_cout1_fwd (nine LDS planes per row block) on the same y, BatchNorm coefficients and
    packed weights, through the C ABI: the same nine products summed in the same tap order, so the outputs agree to the last bits of an fp32
    sigmoid; maps with an odd number of pixel groups per block exercise the clamped tail."""
    from viai_amd import ops, _lib
    lib = _lib.load()
    N, H, W = 3, 12, 48
    y = O.cf_uniform("pd.y.%d" % cin, (N, H, W, cin), -2, 2).cuda()
    w = O.cf_std("pd.w.%d" % cin, (cin, 1, 3, 3) if transposed else (1, cin, 3, 3), 0.05).cuda()
    sc = O.cf_uniform("pd.sc.%d" % cin, (cin,), 0.5, 1.5).cuda(); sh = O.cf_uniform("pd.sh.%d" % cin, (cin,), -0.3, 0.3).cuda()
    bias = torch.tensor([0.1], device="cuda")
    d = ops.conv_desc(N, H, W, cin, 0, 1, 3, 3, 1, 1, 1, 1, transposed)
    assert lib.viai_pair_cout1_ok(d["ref"]) == 1
    wp = torch.empty(d["packed"], device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.viai_conv2d_pack_fwd(d["ref"], w.data_ptr(), wp.data_ptr(), st), "pack")
    a, b = torch.empty(N, H, W, 1, device="cuda"), torch.empty(N, H, W, 1, device="cuda")
    ws = torch.empty(9 * N * H * W, device="cuda")
    _lib.check(lib.viai_pair_cout1_fwd(d["ref"], y.data_ptr(), sc.data_ptr(), sh.data_ptr(), ops.ACT_LRELU, wp.data_ptr(), bias.data_ptr(),
                                       a.data_ptr(), ops.ACT_SIGMOID, st), "fwd")
    _lib.check(lib.viai_pair_cout1_fwd_dots(d["ref"], y.data_ptr(), sc.data_ptr(), sh.data_ptr(), ops.ACT_LRELU, wp.data_ptr(), bias.data_ptr(),
                                            ws.data_ptr(), b.data_ptr(), ops.ACT_SIGMOID, st), "fwd_dots")
    torch.cuda.synchronize()
    assert (a - b).abs().max().item() < 2e-6
    # and against torch in fp64
    z = F.leaky_relu(y.double().cpu() * sc.double().cpu() + sh.double().cpu(), 0.2).permute(0, 3, 1, 2)
    ref = torch.sigmoid((F.conv_transpose2d if transposed else F.conv2d)(z, w.double().cpu(), bias.double().cpu(), 1, 1))
    assert relerr(nchw(b), ref) < 1e-5


@pytest.mark.parametrize("M,C", [(131072, 32), (32768, 64), (5000, 128), (4096, 96)])
def test_operand_magnitude_of_the_fat_block_batchnorm_passes_is_the_exact_maximum(M, C):
    """The BatchNorm apply passes that end in the abs-max atomics run as one 1024-thread block per CU (viai_common.h block_absmax_to): the
    slot -- zeroed, as at the start of a step -- must receive exactly max |z| / max |dy| whatever the grid, including tensors smaller than
    one block per CU, a channel count whose quads do not divide the block (C = 96) and a ragged tail."""
    from viai_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    y = O.cf_uniform("fat.y", (M, C), -3, 3).cuda(); dz = O.cf_uniform("fat.dz", (M, C), -1, 1).cuda()
    sc = O.cf_uniform("fat.sc", (C,), 0.5, 1.5).cuda(); sh = O.cf_uniform("fat.sh", (C,), -0.5, 0.5).cuda()
    mean = O.cf_uniform("fat.mu", (C,), -0.2, 0.2).cuda(); inv = O.cf_uniform("fat.is", (C,), 0.5, 2.0).cuda()
    z, dy = torch.empty_like(y), torch.empty_like(y)
    am = torch.zeros(2, device="cuda")
    _lib.check(lib.viai_bn_act_fwd_amax(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), z.data_ptr(), M, C, 2, 0.2, am[0:1].data_ptr(), st), "fwd")
    nblk = lib.viai_bn_bwd_blocks(M, C)
    part = torch.empty(2 * C * nblk, device="cuda"); sums = torch.empty(2 * C, device="cuda"); dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    _lib.check(lib.viai_bn_act_bwd_amax(dz.data_ptr(), y.data_ptr(), mean.data_ptr(), inv.data_ptr(), sc.data_ptr(), sh.data_ptr(), part.data_ptr(),
                                        sums.data_ptr(), dg.data_ptr(), db.data_ptr(), dy.data_ptr(), M, C, 2, 0.2, 1, am[1:2].data_ptr(), st), "bwd")
    torch.cuda.synchronize()
    assert am[0].item() == z.abs().max().item() and am[1].item() == dy.abs().max().item()
    zr = F.leaky_relu(y * sc + sh, 0.2)
    assert relerr(z.cpu(), zr.cpu()) < 1e-6


def test_resize_inside_the_batchnorm_apply_pass_is_bitwise_the_two_passes():
    """MelDecoder (New_Inpainting_Networks.py:70-89): every F.interpolate follows the last layer of a block; ops.conv_bn_act(..., upsample=)
    applies BatchNorm + ReLU to the four taps while the resize loads them (viai_bn_act_bilinear_fwd_amax) and never stores the map in
    between.  Output, operand magnitudes and every gradient must equal the two-pass route bit for bit (same expressions, same kernels in
    the backward); a 24-channel layer (six quads: the per-pixel kernels do not take it) must fall back to the separate resize."""
    from viai_amd import networks as N_, ops
    torch.manual_seed(3)
    dec = N_.MelDecoder().cuda().train()
    B, F_, T = 2, 64, 32
    shapes = [(B, F_ // 2, T // 2, 32), (B, F_ // 4, T // 2, 64), (B, F_ // 8, T // 4, 128), (B, F_ // 16, T // 8, 256), (B, 2, T // 16, 256)]
    net0 = [O.cf_uniform("up.n%d" % i, sh_, -1, 1).cuda() for i, sh_ in enumerate(shapes)]
    gout = O.cf_uniform("up.g", (B, F_, T, 1), -1, 1).cuda()

    def run(on):
        old, ops.FUSE_BN_UP = ops.FUSE_BN_UP, on
        calls = []
        lib = ops._lib.load()
        orig = lib.viai_bn_act_bilinear_fwd_amax
        lib.viai_bn_act_bilinear_fwd_amax = lambda *a: (calls.append(1), orig(*a))[1]
        try:
            state = {k: v.clone() for k, v in dec.state_dict().items()}
            for p_ in dec.parameters():
                p_.grad = None
            net = [t.clone().requires_grad_(True) for t in net0]
            out = dec.forward_nhwc(net, (F_, T))
            out.backward(gout)
            res = [out.detach().clone()] + [t.grad.clone() for t in net if t.grad is not None] + [p_.grad.clone() for p_ in dec.parameters() if p_.grad is not None]
            dec.load_state_dict(state)
            return res, len(calls)
        finally:
            ops.FUSE_BN_UP = old
            lib.viai_bn_act_bilinear_fwd_amax = orig
    fused, nf = run(True)
    plain, npl = run(False)
    assert nf == 5 and npl == 0                    # head + four blocks resized inside their BatchNorm apply pass
    assert len(fused) == len(plain)
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert torch.equal(a, b), i
    # a channel count the fused pass does not take
    conv = torch.nn.ConvTranspose2d(16, 24, 3, 1, 1, bias=False).cuda(); bn = torch.nn.BatchNorm2d(24).cuda()
    x = O.cf_uniform("up.x24", (2, 8, 8, 16), -1, 1).cuda()
    assert not ops.upsample_fusable(x, conv.weight, None, bn, True, ops.ACT_RELU)
    y = N_.fused_layer(x, conv, bn, ops.ACT_RELU, upsample=(16, 12))
    assert tuple(y.shape) == (2, 16, 12, 24)
