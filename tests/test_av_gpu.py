"""GPU parity of the AV decoders (MelDecoderImage / MelDecoderImage2 / MelDecoder_old) and the sync / domain
discriminators (Inpainting_Dis, DomainDis) against the oracle and the goldens produced by the reference's
own modules at the native MUSICES shape (80 x 208)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import viai_oracle as O


def relerr(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


B, F_BINS, T = 2, 80, 208


def inputs():
    s = O.cf_uniform("av.s", (B, 1, F_BINS, T))
    video = O.cf_uniform("av.video", (B, 256, 1, 13), -1, 1)
    fea = O.cf_uniform("av.fea", (B, 512, 52), -1, 1)
    return s, video, fea


@pytest.mark.parametrize("variant", ["image", "image2", "old"])
def test_av_decoders(variant, golden_dir):
    from viai_amd import networks as N
    gold = np.load(golden_dir + "/av.npz")
    s, video, _ = inputs()
    E = N.MelEncoder().cuda(); E.load_state_dict(O.encoder_state())
    cls = {"image": N.MelDecoderImage, "image2": N.MelDecoderImage2, "old": N.MelDecoder_old}[variant]
    G = cls().cuda(); G.load_state_dict(O.decoder_variant_state(variant))
    feats = [f.detach() for f in E(s.cuda().view(B, F_BINS, T))]
    fake = G(feats, s.size(), video.cuda()) if variant != "old" else G(feats, s.size())
    assert tuple(fake.shape) == (B, 1, F_BINS, T)
    assert relerr(fake, gold["dec_%s.fake" % variant]) < 1e-4
    fake.mean().backward()
    key = "deconv1_1_1.weight" if variant != "old" else "deconv1_1.weight"
    g = dict(G.named_parameters())[key].grad
    assert abs(O.digest(g)[2] - gold["dec_%s.g.%s.dg" % (variant, key)][2]) < 3e-2 * gold["dec_%s.g.%s.dg" % (variant, key)][2]
    g2 = G.conv6_2.weight.grad
    assert relerr(O.digest(g2), gold["dec_%s.g.conv6_2.weight.dg" % variant]) < 1e-3
    # unused-by-forward parameters stay without gradient, as in the reference
    if variant != "old":
        assert G.deconv1_1.weight.grad is None
    # oracle agreement on all gradients (global)
    osd = O._leafify(O.decoder_variant_state(variant))
    ofe = [f.detach() for f in O.encoder_forward(O.encoder_state(), s.view(B, F_BINS, T))]
    ofake = O.decoder_variant_forward(osd, variant, ofe, s.shape, video if variant != "old" else None)
    keys = [k for k, p in G.named_parameters() if p.grad is not None and k not in ("deconv1_1.bias", "deconv1_1_1.bias", "deconv1_2.bias", "conv6_1.bias")]
    og = torch.autograd.grad(ofake.mean(), [osd[k] for k in keys])
    num = sum((dict(G.named_parameters())[k].grad.cpu().double() - o.double()).pow(2).sum().item() for k, o in zip(keys, og))
    den = sum(o.double().pow(2).sum().item() for o in og)
    assert (num / den) ** 0.5 < 2e-2


def test_init_deconv_1_1_1():
    from viai_amd import networks as N
    G = N.MelDecoderImage()
    G.init_deconv_1_1_1()
    assert torch.equal(G.deconv1_1_1.weight[:256], G.deconv1_1.weight)
    assert torch.equal(G.deconv1_1_1.weight[256:], G.deconv1_1.weight)


def test_inpainting_dis(golden_dir):
    from viai_amd import networks as N
    gold = np.load(golden_dir + "/av.npz")
    s, _, fea = inputs()
    D = N.Inpainting_Dis().cuda(); D.load_state_dict(O.inpainting_dis_state()); D.train()
    y = D(s.cuda(), fea.cuda())
    assert tuple(y.shape) == (B, 21)
    assert relerr(y, gold["inp_dis.out"]) < 1e-4
    y.mean().backward()
    for k in ("mel_conv1.weight", "mel_conv4.weight", "vid_conv1.weight", "conv.weight", "vid_bn1.weight"):
        assert relerr(O.digest(dict(D.named_parameters())[k].grad), gold["inp_dis.g.%s.dg" % k]) < 5e-3, k
    osd = O.inpainting_dis_state()
    assert relerr(D.vid_bn1.running_var, O.inpainting_dis_forward(osd, s, fea) is not None and osd["vid_bn1.running_var"]) < 1e-4


def test_domain_dis(golden_dir):
    from viai_amd import networks as N
    gold = np.load(golden_dir + "/av.npz")
    emb = O.cf_uniform("av.emb", (4, 256, 1, 13), -1, 1)
    D = N.DomainDis().cuda(); D.load_state_dict(O.domain_dis_state())
    y = D(emb.cuda())
    assert tuple(y.shape) == (4, 1)
    assert relerr(y, gold["dom_dis.out"]) < 1e-5
    y.mean().backward()
    for k in ("conv1.weight", "fc1.weight", "fc1.bias", "fc2.weight"):
        assert relerr(O.digest(dict(D.named_parameters())[k].grad), gold["dom_dis.g.%s.dg" % k]) < 1e-4, k


def test_multiscale_discriminator_is_a_composition_of_reference_discriminators():
    """cfg 4: k MelDiscriminators on the avg_pool2d(3,2,1,count_include_pad=False) pyramid == oracle composition."""
    from viai_amd import networks as N
    x = O.cf_uniform("msd.x", (2, 1, 64, 96)).requires_grad_(True)
    D = N.MultiScaleDiscriminator(num_D=3).cuda()
    sds = [O.disc_state("D%d." % i) for i in range(3)]
    for i in range(3):
        D._modules["scale%d" % i].load_state_dict(sds[i])
    xg = x.detach().cuda().requires_grad_(True)
    outs = D(xg)
    loss = sum(o.mean() for o in outs)
    loss.backward()
    cur, ref_loss = x, 0
    for i in range(3):
        ro = O.disc_forward(sds[i], cur)
        assert tuple(outs[i].shape) == tuple(ro.shape)
        assert relerr(outs[i], ro) < 1e-4, i
        ref_loss = ref_loss + ro.mean()
        cur = torch.nn.functional.avg_pool2d(cur, 3, 2, 1, count_include_pad=False)
    ref_loss.backward()
    assert relerr(xg.grad, x.grad) < 5e-3
