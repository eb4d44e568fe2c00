import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_p16_gpu import to_p16, _st
from viai_amd import _lib, ops
lib = _lib.load()
S, Ci, Co, N, OHW = 2, 64, 128, 4, 64
H = W = OHW * S
gen = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(N, H, W, Ci, device="cuda", generator=gen)
w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=gen) * 0.05
d = ops.conv_desc(N, H, W, Ci, 0, Co, 3, 3, S, S, 1, 1, 0)
xp, xa = to_p16(x)
wp = torch.empty(d["packed"], device="cuda")
_lib.check(lib.viai_conv2d_pack_fwd(d["ref"], w.data_ptr(), wp.data_ptr(), _st()), "pack")
y0, y1 = torch.empty(N, OHW, OHW, Co, device="cuda"), torch.empty(N, OHW, OHW, Co, device="cuda")
st0 = torch.zeros(2 * Co * d["nblk"], device="cuda"); st1 = torch.zeros_like(st0)
_lib.check(lib.viai_conv2d_fwd_amax(d["ref"], x.data_ptr(), 0, wp.data_ptr(), 0, y0.data_ptr(), st0.data_ptr(), 0, xa.data_ptr(), _st()), "a")
_lib.check(lib.viai_conv2d_fwd_p16(d["ref"], xp.data_ptr(), wp.data_ptr(), 0, y1.data_ptr(), st1.data_ptr(), 0, xa.data_ptr(), _st()), "b")
torch.cuda.synchronize()
Mb = d["nblk"]
print("nblk", Mb, "rows", d["rows"], "y rel", ((y0 - y1).norm() / y0.norm()).item())
m0, m1 = st0.view(2, Co, Mb), st1.view(2, Co, Mb)
bad = ((m0[0] - m1[0]).abs() > 1e-4).nonzero()
print("bad mean entries", bad.shape[0], "of", Co * Mb)
print(bad[:20].tolist())
# recompute from y1
yb = y1.view(N, OHW // 4, 4, OHW // 16, 16, Co).permute(0, 1, 3, 5, 2, 4).reshape(N * (OHW // 4) * (OHW // 16), Co, 64)
mean_ref = yb.mean(-1).t()
print("st1 mean vs recomputed:", (m1[0] - mean_ref).abs().max().item(), " st0 vs recomputed:", (m0[0] - mean_ref).abs().max().item())
for c, b in bad[:5].tolist():
    print(c, b, m0[0, c, b].item(), m1[0, c, b].item(), mean_ref[c, b].item())
