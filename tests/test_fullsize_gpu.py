"""Full-size checks (BASELINE.json configs[1]: batch 16, 256 x 256 mel) through properties that need no oracle run.

The oracle finishes cfg 0 / cfg 1 in seconds and pins the kernels there (test_networks_gpu.py); at the benchmark size it
would take minutes per step on the CPU, so the kernels the full-size layers select (128 x 256 wide tile, fused stride-2
data gradient, register-filter halo kernel, split-K, row-run streaming kernels, all-taps weight gradient ...) are checked here
through identities that hold for any size:

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
