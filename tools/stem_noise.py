"""How far is the conv1.weight gradient of the 4-frame ImageEmbedding2 golden from an fp64 evaluation, for (a) the reference's own
fp32 values (tests/golden/resnet.npz), (b) the HIP step with the exact-fp32 stem kernels, (c) with the f16x2 stem kernels?
Run once per setting of VIAI_STEM_F16 (the switch is read at load time)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import viai_oracle as O
from viai_amd import networks as N

gold = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "resnet.npz"))
video = O.cf_uniform("ie.video", (1, 4, 3, 224, 224), -1, 1)
flow = O.cf_uniform("ie.flow", (1, 4, 2, 224, 224), -1, 1)
sd = {k: v.double().requires_grad_(v.dtype.is_floating_point) if v.dtype.is_floating_point else v for k, v in O.image_embedding2_state().items()}
out, fea = O.image_embedding2_forward(sd, video.double(), flow.double(), training=True)
(out.pow(2).mean() + fea.pow(2).mean()).backward()
M = N.ImageEmbedding2().cuda(); M.load_state_dict(O.image_embedding2_state()); M.train()
o, f = M(video.cuda(), flow.cuda())
(o.pow(2).mean() + f.pow(2).mean()).backward()
params = dict(M.named_parameters())
for k in ("image_single_model.conv1.weight", "flow_single_model.conv1.weight", "image_single_model.layer4.1.conv2.weight", "flow_single_model.layer1.0.bn1.weight"):
    t = O.digest(sd[k].grad.float())
    h = O.digest(params[k].grad)
    r = gold["g.%s.dg" % k]
    full = ((params[k].grad.double().cpu() - sd[k].grad).norm() / sd[k].grad.norm()).item()
    print("%-45s STEM_F16=%s  samples: ref-vs-fp64 %.2e  hip-vs-fp64 %.2e  hip-vs-ref %.2e   whole tensor hip-vs-fp64 %.2e" % (
        k, os.environ.get("VIAI_STEM_F16", "1"), np.linalg.norm(r[3:] - t[3:]) / np.linalg.norm(t[3:]), np.linalg.norm(h[3:] - t[3:]) / np.linalg.norm(t[3:]),
        np.linalg.norm(h[3:] - r[3:]) / np.linalg.norm(r[3:]), full))
import torch.nn.functional as F
from viai_amd import ops
for name, net, frames in (("image", M.image_single_model, video), ("flow", M.flow_single_model, flow)):
    w = net.conv1.weight.detach()
    x = frames.reshape((-1,) + tuple(frames.shape[2:]))
    y64 = F.conv2d(x.double(), w.double().cpu(), None, stride=2, padding=3)
    y32 = F.conv2d(x, w.cpu(), None, stride=2, padding=3)
    yg = ops.conv_bn_act(ops.frames_to_nhwc4(x.cuda()), w, None, None, kernel=(7, 7), stride=(2, 2), padding=(3, 3))
    yh = yg.permute(0, 3, 1, 2).double().cpu()
    e = (yh - y64)
    print("%s conv1 out: hip-vs-fp64 %.2e (max abs %.2e, mean err %.2e)  cpu32-vs-fp64 %.2e   |w| max %.3f" % (
        name, (e.norm() / y64.norm()).item(), e.abs().max().item(), e.mean().item(), ((y32.double() - y64).norm() / y64.norm()).item(), w.abs().max().item()))
print("fea hip-vs-fp64 %.2e   out hip-vs-fp64 %.2e" % (((f.double().cpu() - fea).norm() / fea.norm()).item(), ((o.double().cpu() - out).norm() / out.norm()).item()))
print("fea ref-vs-fp64 %.2e" % ((torch.from_numpy(gold["fea_cat"]).double() - fea.detach()).norm() / fea.norm()).item())
