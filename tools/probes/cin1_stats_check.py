"""probe: BatchNorm partials of the Cin = 1 layer -- tap-covariance kernel (viai_conv2d_cin1_bn_fwd, statistics mode) against the conv-then-reduce kernel
(viai_conv2d_fwd_amax with stat_part) and against fp64, per 256-pixel block and channel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viai_amd import _lib, ops
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for (N, H, W, Co, kh, kw, sh, sw, ph, pw) in [(2, 80, 32, 64, 1, 4, 1, 2, 0, 1), (2, 80, 32, 32, 3, 3, 2, 2, 1, 1), (16, 256, 256, 64, 1, 4, 1, 2, 0, 1)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(N, H, W, 1, device="cuda", generator=g)
    x = (x + torch.roll(x, 1, 2) + torch.roll(x, 2, 2)) / 3          # correlated neighbours, like a mel image
    w = torch.randn(Co, 1, kh, kw, device="cuda", generator=g) * 0.5
    d = ops.conv_desc(N, H, W, 1, 0, Co, kh, kw, sh, sw, ph, pw, 0)
    wp = torch.empty(d["packed"], device="cuda")
    _lib.check(lib.viai_conv2d_pack_fwd(d["ref"], w.data_ptr(), wp.data_ptr(), st), "pack")
    OH, OW = d["OH"], d["OW"]
    M = N * OH * OW
    nb = d["nblk"]
    y = torch.empty(N, OH, OW, Co, device="cuda")
    s_old = torch.zeros(2 * Co * nb, device="cuda"); s_new = torch.zeros_like(s_old)
    _lib.check(lib.viai_conv2d_fwd_amax(d["ref"], x.data_ptr(), 0, wp.data_ptr(), 0, y.data_ptr(), s_old.data_ptr(), 0, 0, st), "old")
    _lib.check(lib.viai_conv2d_cin1_bn_fwd(d["ref"], x.data_ptr(), 0, wp.data_ptr(), 0, s_new.data_ptr(), 0, 0, 0, 0, 0, st), "new")
    torch.cuda.synchronize()
    y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(), None, (sh, sw), (ph, pw)).permute(0, 2, 3, 1).reshape(M, Co)
    rows = d["rows"]
    yb = y64.reshape(nb, rows, Co)
    mean64 = yb.mean(1).t(); m2_64 = ((yb - yb.mean(1, keepdim=True)) ** 2).sum(1).t()
    for nm, s in (("conv+reduce", s_old), ("tap-covariance", s_new)):
        v = s.view(2, Co, nb).double().cpu()
        print("%-28s %-15s nblk %5d rows %4d  mean err %.2e (scale %.2e)   M2 rel err max %.2e" % ((N, H, W, Co, kh, kw), nm, nb, rows, (v[0] - mean64).abs().max().item(), mean64.abs().max().item(),
              ((v[1] - m2_64).abs() / m2_64).max().item()))
