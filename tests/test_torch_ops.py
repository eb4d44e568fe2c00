"""`torch.ops.viai.conv_bn_act`: the fused layer registered with torch.library (north_star: "through PyTorch-ROCm custom ops") on top
of the same launch logic as viai_amd.ops.conv_bn_act.  CPU part: schema, fake-tensor shapes, loud failure without a GPU.  GPU part: the registered op
equals the autograd.Function path bit for bit (outputs, running statistics, all five gradients) and passes torch.library.opcheck."""
import pytest
import torch

import viai_amd.torch_ops as T  # noqa: F401  (registers the ops)
from viai_amd import ops


def test_ops_are_registered_with_schema_and_fake_kernels():
    s = str(torch.ops.viai.conv_bn_act.default._schema)
    assert s.startswith("viai::conv_bn_act(Tensor x, Tensor weight, Tensor? bias, Tensor? gamma, Tensor? beta, Tensor? running_mean, Tensor? running_var")
    assert "viai::conv_bn_act_backward(" in str(torch.ops.viai.conv_bn_act_backward.default._schema)
    x, w, g = torch.empty(2, 16, 16, 8, device="meta"), torch.empty(16, 8, 3, 3, device="meta"), torch.empty(16, device="meta")
    out = torch.ops.viai.conv_bn_act(x, w, None, g, g, g, g, [3, 3], [2, 2], [1, 1], False, ops.ACT_LRELU, True, 0.1, 1e-5)
    assert [tuple(o.shape) for o in out] == [(2, 8, 8, 16), (2, 8, 8, 16), (4, 16), (1,), (16,), (16,)]
    wt = torch.empty(8, 4, 3, 3, device="meta")                      # ConvTranspose2d layout [Cin][Cout][kh][kw], stride 1
    out = torch.ops.viai.conv_bn_act(x, wt, None, None, None, None, None, [3, 3], [1, 1], [1, 1], True, ops.ACT_SIGMOID, True, 0.1, 1e-5)
    assert tuple(out[0].shape) == (2, 16, 16, 4) and out[1].numel() == 0 and out[2].numel() == 0
    with pytest.raises(NotImplementedError):                         # no CPU kernel, as everywhere in this package
        torch.ops.viai.conv_bn_act(torch.zeros(1, 4, 4, 2), torch.zeros(2, 2, 3, 3), None, None, None, None, None, [3, 3], [1, 1], [1, 1], False, 0, True, 0.1, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["conv_bn_lrelu_s2", "convT_bn_relu", "conv_bias_sigmoid", "cin1_bn"])
def test_registered_op_equals_the_autograd_function_path(case):
    torch.manual_seed(3)
    dev = "cuda"
    if case == "conv_bn_lrelu_s2":
        x, conv, bn, kw = torch.randn(4, 32, 32, 32, device=dev), torch.nn.Conv2d(32, 64, 3, 2, 1, bias=False), torch.nn.BatchNorm2d(64), dict(kernel=(3, 3), stride=(2, 2), padding=(1, 1), act=ops.ACT_LRELU)
    elif case == "convT_bn_relu":
        x, conv, bn, kw = torch.randn(4, 16, 16, 64, device=dev), torch.nn.ConvTranspose2d(64, 32, 3, 1, 1, bias=True), torch.nn.BatchNorm2d(32), dict(kernel=(3, 3), stride=(1, 1), padding=(1, 1), transposed=True, act=ops.ACT_RELU)
    elif case == "conv_bias_sigmoid":
        x, conv, bn, kw = torch.randn(4, 16, 16, 32, device=dev), torch.nn.Conv2d(32, 1, 3, 1, 1, bias=True), None, dict(kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_SIGMOID)
    else:
        x, conv, bn, kw = torch.rand(4, 64, 64, 1, device=dev), torch.nn.Conv2d(1, 32, 3, 2, 1, bias=False), torch.nn.BatchNorm2d(32), dict(kernel=(3, 3), stride=(2, 2), padding=(1, 1), act=ops.ACT_LRELU)
    conv = conv.to(dev)
    bn = bn.to(dev) if bn is not None else None
    if bn is not None:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)

    def run(fn):
        for m in (conv, bn):
            if m is not None:
                m.zero_grad(set_to_none=True)
        if bn is not None:
            bn.reset_running_stats()
        xi = x.clone().requires_grad_(True)
        ops.begin_step(xi.device)
        z = fn(xi, conv.weight, conv.bias, bn, **kw)
        g = torch.randn(z.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        z.backward(g)
        torch.cuda.synchronize()
        grads = [xi.grad] + [p.grad for m in (conv, bn) if m is not None for p in m.parameters()]
        stats = [bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()] if bn is not None else []
        return [z.detach().clone()] + [t.clone() for t in grads] + stats
    ref, got = run(ops.conv_bn_act), run(T.conv_bn_act)
    assert len(ref) == len(got)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert float(ref[0].abs().max()) > 0 and float(ref[1].abs().max()) > 0


@pytest.mark.gpu
def test_opcheck_of_the_registered_ops():
    dev = "cuda"
    torch.manual_seed(5)
    x = torch.randn(2, 16, 16, 32, device=dev, requires_grad=True)
    w = (torch.randn(64, 32, 3, 3, device=dev) * 0.1).requires_grad_(True)
    gamma, beta = torch.rand(64, device=dev).add_(0.5).requires_grad_(True), torch.zeros(64, device=dev, requires_grad=True)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    ops.begin_step(x.device)
    args = (x, w, None, gamma, beta, rm, rv, [3, 3], [1, 1], [1, 1], False, ops.ACT_RELU, True, 0.1, 1e-5)
    torch.library.opcheck(torch.ops.viai.conv_bn_act.default, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
