"""signed error statistics of the f16x2 stem forward against fp64 (is there a rounding bias?)"""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viai_amd import ops
torch.manual_seed(0)
for off in (0.0, 0.3):
    x = torch.rand(8, 3, 224, 224) * 2 - 1 + off
    w = torch.randn(64, 3, 7, 7) * 0.05
    y64 = F.conv2d(x.double(), w.double(), None, stride=2, padding=3)
    yg = ops.conv_bn_act(ops.frames_to_nhwc4(x.cuda()), w.cuda(), None, None, kernel=(7, 7), stride=(2, 2), padding=(3, 3))
    e = yg.permute(0, 3, 1, 2).double().cpu() - y64
    n = e.numel()
    sig = e.std().item() / n ** 0.5
    print("STEM_F16=%s offset %.1f: rms rel %.2e  mean err %.2e (%.1f sigma)  mean err*sign(y) %.2e (%.1f sigma)  mean y %.2e" % (
        os.environ.get("VIAI_STEM_F16", "1"), off, (e.norm() / y64.norm()).item(), e.mean().item(), e.mean().item() / sig,
        (e * y64.sign()).mean().item(), (e * y64.sign()).mean().item() / sig, y64.mean().item()))
