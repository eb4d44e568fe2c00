"""checksum + timing of the split-K small-map forward kernel on one shape (bitwise A/B between library builds)"""
import ctypes as C, os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from viai_amd import _lib, ops
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for (N, H, W, Ci, Co) in ((16, 8, 16, 256, 256), (16, 16, 32, 128, 256), (16, 4, 8, 256, 256)):
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, H, W, Ci, device="cuda", generator=gen)
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=gen) * 0.05
    d = ops.conv_desc(N, H, W, Ci, 0, Co, 3, 3, 1, 1, 1, 1, 0)
    wp = torch.empty(d["packed"], device="cuda")
    _lib.check(lib.viai_conv2d_pack_fwd(d["ref"], w.data_ptr(), wp.data_ptr(), st), "pack")
    y = torch.empty(N, H, W, Co, device="cuda")
    stat = torch.empty(2 * Co * max(d["nblk"], 1) + 16, device="cuda")
    xa = x.abs().max().reshape(1)
    fam = C.create_string_buffer(64)
    def run():
        _lib.check(lib.viai_conv2d_fwd_amax(d["ref"], x.data_ptr(), 0, wp.data_ptr(), 0, y.data_ptr(), stat.data_ptr(), 0, xa.data_ptr(), st), "fwd")
    run(); lib.viai_conv2d_last_kernel(fam, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    h = hashlib.sha1(y.cpu().numpy().tobytes() + stat[:2 * Co * d["nblk"]].cpu().numpy().tobytes()).hexdigest()[:16]
    print((N, H, W, Ci, Co), fam.value.decode(), "%.1f us" % (e0.elapsed_time(e1) * 20), h)
