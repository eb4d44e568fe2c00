"""The arithmetic-mode switches of the conv kernels, exercised (round-3 review: "shipped but untested"):

  VIAI_MATH=fp32   exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) in every conv kernel -- the only mode whose products are literally the
                   reference's fp32 products (csrc/conv_api.hip `math_f16`, conv_igemm.hip, conv_wgrad.hip)
  VIAI_F16X2=0     bf16x3 split (three bf16 terms, six partial products) instead of the default f16x2 split

Both are read once per process by the library, so every leg is a child interpreter that runs the step-golden test of
tests/test_networks_gpu.py (tiny and cfg 1: the reference's golden outputs + the fp64 gradient criterion) under the switch and first
asserts that the switch took effect (the kernel family the library reports for a wide layer ends in _f32 / _bf16x3).
Also: NaN must propagate through the piecewise-linear activations (round-3 advice: the one-expression viai_act mapped NaN to 0).
Reference semantics: nn.LeakyReLU / nn.ReLU / identity on torch tensors (networks/Discriminator_Networks.py:38-49)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import ctypes, sys, torch
sys.path.insert(0, %r)
from viai_amd import _lib, ops
x = torch.rand(2, 16, 32, 128, device="cuda")
w = torch.randn(128, 128, 3, 3, device="cuda") * 0.05
with torch.no_grad():
    ops.conv_bn_act(x, w, None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=ops.ACT_NONE)
buf = ctypes.create_string_buffer(64)
_lib.load().viai_conv2d_last_kernel(buf, 64)
print("FAMILY", buf.value.decode())
"""


def _child(env_extra, args):
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_step_goldens_hold_in_the_other_arithmetic_modes(mode):
    env = {"VIAI_MATH": "fp32"} if mode == "fp32" else {"VIAI_F16X2": "0"}
    suffix = "_f32" if mode == "fp32" else "_bf16x3"
    r = _child(env, ["-c", _PROBE % ROOT])
    assert r.returncode == 0, r.stdout[-3000:]
    fam = [l for l in r.stdout.splitlines() if l.startswith("FAMILY")][-1].split()[-1]
    assert fam.endswith(suffix), (mode, fam)
    r = _child(env, ["-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                     "tests/test_networks_gpu.py::test_step_no_update_matches_oracle_and_golden",
                     "tests/test_networks_gpu.py::test_one_adam_step_from_synced_state_matches_oracle"])
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-4000:]


def test_step_goldens_hold_on_the_register_staged_kernels():
    """VIAI_HALO_DMA=0 (round 5): the layers that run on the LDS-DMA kernels of csrc/conv_halo_dma.hip by default (G's 32-channel layers, the stride-2
    forward convs of D and E) fall back to the register-staged kernels of conv_halo_bf3.hip -- the pre-split bitwise / tolerance tests and the step goldens
    must hold on both."""
    env = {"VIAI_HALO_DMA": "0"}
    r = _child(env, ["-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                     "tests/test_networks_gpu.py::test_step_no_update_matches_oracle_and_golden",
                     "tests/test_p16_gpu.py::test_forward_and_data_gradient_on_presplit_operands_are_bitwise_the_fp32_input_kernels"])
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-4000:]


@pytest.mark.parametrize("act", [0, 1, 2], ids=["none", "relu", "lrelu"])
def test_nan_propagates_through_the_piecewise_linear_activations(act):
    """torch: relu(nan) = leaky_relu(nan) = nan.  Through the BatchNorm apply pass (eval coefficients 1 / 0) and through a conv epilogue
    without statistics (the D.conv4-under-LSGAN case of the advice)."""
    from viai_amd import _lib, ops
    lib = _lib.load()
    M, Cc = 1024, 32
    y = torch.randn(M, Cc, device="cuda")
    y[5, 3] = float("nan")
    y[700, 31] = float("nan")
    sc, sh = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    z = torch.empty_like(y)
    am = torch.zeros(1, device="cuda")
    _lib.check(lib.viai_bn_act_fwd_amax(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), z.data_ptr(), M, Cc, act, 0.2, am.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "viai_bn_act_fwd_amax")
    torch.cuda.synchronize()
    assert torch.isnan(z[5, 3]) and torch.isnan(z[700, 31])
    fin = torch.isfinite(y)
    ref = {0: y, 1: torch.relu(y), 2: torch.nn.functional.leaky_relu(y, 0.2)}[act]
    assert torch.equal(z[fin], ref[fin])                      # finite values: unchanged bit for bit
    assert int(torch.isnan(z).sum()) == 2
    # conv epilogue, no BatchNorm: a NaN input pixel poisons the outputs its window reaches, nothing is flushed to zero
    x = torch.rand(1, 8, 16, 1, device="cuda")
    x[0, 4, 7, 0] = float("nan")
    w = torch.randn(32, 1, 3, 3, device="cuda")           # (the Cin = 1 kernels take 32 / 64 / 128 output channels)
    with torch.no_grad():
        o = ops.conv_bn_act(x, w, None, None, kernel=(3, 3), stride=(1, 1), padding=(1, 1), act=act)
    assert bool(torch.isnan(o[0, 3:6, 6:9, :]).all()) and int(torch.isnan(o).sum()) == 9 * 32
