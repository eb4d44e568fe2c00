"""GPU parity of the module shells and the declared G+D step against the CPU
oracle and the golden vectors generated from the reference's own modules."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import viai_oracle as O

TOL_FWD = 1e-4      # forward tensors (north star: 1e-3 relative fp32)
# Gradient digests vs the reference's fp32 gradients.  Measured on MI355X (tools/grad_margins.py): worst norm error 2.0e-4 .. 3.9e-4
# (tiny), 8.1e-4 .. 2.6e-3 (cfg 1), 1.35e-3 (cfg 2); worst 64-sample-vector error 3.7e-3 .. 1.8e-2 -- the ranges are two ROUNDING
# REALISATIONS of the same kernels (static vs magnitude-derived operand scale of the fp16 split, round 3: VIAI_F16_DYNAMIC=0/1), which
# against fp64 on the tie-free input are equally accurate (cfg 1: E 2.3e-3 / 2.1e-3, G 2.2e-3 / 2.0e-3, D 7.8e-4 / 7.9e-4; CPU fp32:
# 1.5e-3 / 1.3e-3 / 7.0e-4).  Two fp32 evaluations of these gradients (oracle vs reference, both torch CPU) differ by 3e-3 .. 5e-3
# themselves, so the norm bound sits at that level; 64 strided samples of one tensor scatter several-fold more than its norm.
TOL_GRAD = 5e-3
TOL_GRAD_SAMPLES = 2.5e-2


def relerr(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def build_model(F_bins, T):
    from viai_amd.model import AudioModel, StepConfig
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = F_bins, T
    m = AudioModel(hp, device="cuda")
    m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
    return m


def named_grads(module):
    return {k: p.grad for k, p in module.named_parameters()}


SHADOWED = ("deconv1_1.bias", "deconv1_2.bias", "conv6_1.bias")


def to64(sd):
    return type(sd)((k, (v.double() if v.is_floating_point() else v.clone())) for k, v in sd.items())


# The only tensors allowed above the per-tensor bound, and only at the tiny shape (2 clips x 80 x 32): two BatchNorm-bias gradients of D, sums over 2 560 / 640
# pixels that nearly cancel.  D.bn1.bias is the worst-conditioned tensor of the step in ANY fp32 arithmetic -- tests/test_oracle_golden.py::
# test_d_bn1_bias_is_the_ill_conditioned_gradient shows CPU fp32 itself 10x further from fp64 on it than on the median D tensor -- and the f16x2 products
# (22 significand bits) and P16 storage (absolute error 2^-25 of a BOUND on the tensor) pay that conditioning with a larger constant.
TINY_ILL_CONDITIONED = {("grads_D", "bn1.bias"), ("grads_D", "norm_2.bias")}


def check_against_oracle(model, ocap, dcap, oE, oG, oD, tiny=False):
    """Parity criterion for gradients: against an fp64 run of the oracle ("truth"), the HIP path must be
    as accurate as the reference's fp32 CPU arithmetic is (factor 4 + a floor of 1.5e-3 per network / 3e-3 per tensor -- round 1
    had 5e-3 / 1e-2 there, an order of magnitude above what the kernels deliver; a 1 % systematic gradient error now fails).  Backprop through
    ~45 conv+BN(train) layers and BCE-on-probabilities amplifies fp32 rounding to 1e-3..2e-2 relative in
    ANY fp32 implementation (oracle-vs-reference differ by 3e-3 at cfg1), and the objective's gradient is
    DISCONTINUOUS: one ReLU/LeakyReLU pre-activation or one L1 residual within rounding noise of zero flips
    a derivative and moves every upstream gradient by ~1/sqrt(#elements) (3e-3 at the tiny shape; measured
    with 1e-7 input perturbations of the fp64 oracle: typical 1e-5, tail 1e-3).  So a fixed 1e-3 bound on
    whole-network gradients would test luck, not kernels; the kernels themselves are held to ~1e-6 against
    fp64 in tests/test_kernels_gpu.py.  Forward tensors and losses are held to 1e-4 here."""
    assert relerr(model.fake, dcap["fake"]) < TOL_FWD
    assert relerr(model._pred_fake_g.permute(0, 3, 1, 2), dcap["pred_fake_g"]) < 1e-3
    for idx, key, tol in ((0, "loss_d", 1e-4), (1, "loss_g", 1e-4), (3, "loss_l1", 1e-5)):
        assert abs(model.losses[idx].item() - dcap[key].item()) < tol * abs(dcap[key].item()), key
    report = {}
    for mod, grp in ((model.netD, "grads_D"), (model.Mel_Encoder, "grads_E"), (model.Mel_Decoder, "grads_G")):
        n_hip = n_o32 = den = 0.0
        for k, g in named_grads(mod).items():
            truth = dcap[grp][k]
            if truth is None:                       # convblock1.*: never reached by forward
                assert float(g.abs().max()) == 0.0, k
                continue
            if grp == "grads_G" and k in SHADOWED:  # bias in front of train-mode BN: exact gradient is 0
                assert float(g.abs().max()) < 1e-4
                continue
            e_hip, e_o32 = relerr(g, truth), relerr(ocap[grp][k], truth)
            # per tensor: 4 e_o32 + 3e-3 (measured worst e_hip - 4 e_o32: 7.2e-4, cfg 1, D.norm_2.bias) for EVERY tensor of every shape; the two named
            # near-cancelling sums above may reach 8e-3 at the tiny shape (measured 5.7e-3 / 3.4e-3).  The network-level bound below is unchanged.
            floor = 8e-3 if (tiny and (grp, k) in TINY_ILL_CONDITIONED) else 3e-3
            assert e_hip < 4 * e_o32 + floor, (grp, k, e_hip, e_o32)
            n_hip += (g.detach().cpu().double() - truth).pow(2).sum().item()
            n_o32 += (ocap[grp][k].double() - truth).pow(2).sum().item()
            den += truth.pow(2).sum().item()
        e_hip, e_o32 = (n_hip / den) ** 0.5, (n_o32 / den) ** 0.5
        report[grp] = (e_hip, e_o32)
        assert e_hip < 4 * e_o32 + 1.5e-3, (grp, e_hip, e_o32)           # measured: 7.8e-4 / 2.3e-3 / 2.2e-3 (D / E / G at cfg 1)
        assert e_hip < 8e-3, (grp, e_hip)
    for mod, osd in ((model.Mel_Encoder, oE), (model.Mel_Decoder, oG), (model.netD, oD)):
        for k, v in mod.state_dict().items():
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(osd[k]), k
            elif "running_" in k:
                assert relerr(v, osd[k]) < 1e-4, k
    return report


separated_input = O.separated_input


@pytest.mark.parametrize("shape", [(2, 80, 32), (4, 128, 128)], ids=["tiny", "cfg1"])
def test_step_no_update_matches_oracle_and_golden(shape, golden_dir):
    B, F_bins, T = shape
    name = "tiny" if T == 32 else "cfg1"
    s = O.cf_uniform("s.%s" % name, (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.%s" % name)
    model = build_model(F_bins, T)
    model.set_inputs(s, mask)
    model.forward_backward_no_update()
    torch.cuda.synchronize()
    # ---- leg 1: the reference's own outputs (golden fixtures), original inputs
    gold = np.load("%s/step_%s.npz" % (golden_dir, name))
    if name == "tiny":
        assert relerr(model.fake, gold["nu.fake"]) < TOL_FWD
        assert relerr(model._pred_fake_g.permute(0, 3, 1, 2), gold["nu.pred_fake_g"]) < 1e-3
    else:
        assert relerr(O.digest(model.fake.contiguous()), gold["nu.fake.dg"]) < TOL_FWD
    for key, idx in (("nu.loss_d", 0), ("nu.loss_g", 1), ("nu.loss_g_gan", 2), ("nu.loss_l1", 3)):
        ref = float(gold[key])
        assert abs(model.losses[idx].item() - ref) < 2e-4 * abs(ref), key
    for mod, grp in ((model.netD, "grads_D"), (model.Mel_Encoder, "grads_E"), (model.Mel_Decoder, "grads_G")):
        for k, g in named_grads(mod).items():
            gk = "nu.%s.%s.dg" % (grp, k)
            if gk not in gold.files or (grp == "grads_G" and k in SHADOWED):
                continue
            dg = O.digest(g)
            ref = gold[gk]
            # digest = [sum, abs-sum, l2, 64 samples]: compare the norms and the sample vector
            assert abs(dg[2] - ref[2]) < TOL_GRAD * ref[2], (gk, dg[2], ref[2])
            assert np.linalg.norm(dg[3:] - ref[3:]) < TOL_GRAD_SAMPLES * (np.linalg.norm(ref[3:]) + 1e-12), gk
    for mod, nm in ((model.Mel_Encoder, "E"), (model.Mel_Decoder, "G"), (model.netD, "D")):
        for k, v in mod.state_dict().items():
            if "running_" in k:
                assert relerr(v, gold["nu.state.%s.%s" % (nm, k)]) < 1e-4, k
    # ---- leg 2: oracle (fp32) and fp64 truth on tie-free inputs: gradient accuracy criterion
    s2 = separated_input(s, mask)
    model = build_model(F_bins, T)
    model.set_inputs(s2, mask)
    model.forward_backward_no_update()
    torch.cuda.synchronize()
    oE, oG, oD = O.encoder_state(), O.decoder_state(), O.disc_state()
    ocap = O.step_no_update(oE, oG, oD, s2, mask)
    dcap = O.step_no_update(to64(O.encoder_state()), to64(O.decoder_state()), to64(O.disc_state()), s2.double(), mask.double())
    print(check_against_oracle(model, ocap, dcap, oE, oG, oD, tiny=(name == "tiny")))


def test_step_no_update_matches_reference_golden_at_benchmark_size(golden_dir):
    """BASELINE.json configs[1] (16 x 256 x 256, the size bench.py times) against digests of the REFERENCE's own modules
    (tools/make_goldens.py --cfg2-only -> tests/golden/step_cfg2.npz): the generator output, the three discriminator
    outputs, the five encoder maps, d loss / d fake, the four losses, every BatchNorm running statistic and every
    parameter-gradient digest.  This is the only place the kernel INSTANCES of the benchmark (wide 8x16x256 halo tile,
    stride-2 patch kernels, row-run streaming kernels, f16x2 weight gradient ...) are compared with the reference end to end
    (networks/Inpainting_Networks.py:69-78, New_Inpainting_Networks.py:70-89, Discriminator_Networks.py:37-50)."""
    B, F_bins, T = 16, 256, 256
    s = O.cf_uniform("s.cfg2", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.cfg2")
    gold = np.load("%s/step_cfg2.npz" % golden_dir)
    assert list(gold["meta"][:3]) == [B, F_bins, T]
    model = build_model(F_bins, T)
    model.set_inputs(s, mask)
    model.forward_backward_no_update()
    torch.cuda.synchronize()

    def dg_err(t, key):
        """digest = [sum, abs-sum, l2, 64 strided samples]; returns (relative error of the norms, of the sample vector)"""
        dg, ref = O.digest(t.contiguous()), gold[key]
        e_norm = max(abs(dg[1] - ref[1]) / (abs(ref[1]) + 1e-30), abs(dg[2] - ref[2]) / (abs(ref[2]) + 1e-30))
        e_smp = np.linalg.norm(dg[3:] - ref[3:]) / (np.linalg.norm(ref[3:]) + 1e-30)
        return e_norm, e_smp
    e = dg_err(model.fake, "nu.fake.dg")
    assert max(e) < TOL_FWD, ("fake", e)
    e = dg_err(model._pred_fake_g.permute(0, 3, 1, 2), "nu.pred_fake_g.dg")
    assert max(e) < 1e-3, ("pred_fake_g", e)                      # BCE probabilities near 0: see check_against_oracle
    for key, idx in (("nu.loss_d", 0), ("nu.loss_g", 1), ("nu.loss_g_gan", 2), ("nu.loss_l1", 3)):
        ref = float(gold[key])
        assert abs(model.losses[idx].item() - ref) < 2e-4 * abs(ref), (key, model.losses[idx].item(), ref)
    for mod, nm in ((model.Mel_Encoder, "E"), (model.Mel_Decoder, "G"), (model.netD, "D")):
        for k, v in mod.state_dict().items():
            if "running_" in k:
                assert relerr(v, gold["nu.state.%s.%s" % (nm, k)]) < 1e-4, k
    worst = {}
    for mod, grp in ((model.netD, "grads_D"), (model.Mel_Encoder, "grads_E"), (model.Mel_Decoder, "grads_G")):
        for k, g in named_grads(mod).items():
            gk = "nu.%s.%s.dg" % (grp, k)
            if gk not in gold.files or (grp == "grads_G" and k in SHADOWED):
                continue
            en, es = dg_err(g, gk)
            worst[grp] = max(worst.get(grp, 0.0), en)
            # the reference's own fp32 gradients sit ~5e-3 from a second fp32 evaluation at this size (tools/make_goldens.py)
            assert en < TOL_GRAD, (gk, en)
            assert es < TOL_GRAD_SAMPLES, (gk, es)
    print("cfg2 worst gradient-norm digests vs reference:", worst)


def test_gradients_at_benchmark_size_against_fp64_truth():
    """The fp64-truth gradient leg at BASELINE.json configs[1] (16 x 256 x 256), tie-free input: every network's gradient error against an
    fp64 run of the oracle is at most 3x the error of the oracle's own fp32 CPU run (round-3 review: anchor the bound at the benchmark
    size on fp64 truth instead of widening digest tolerances).  Forward tensors and losses to 1e-4, running statistics to 1e-4
    (networks/Discriminator_Networks.py:14-36, New_Inpainting_Networks.py:70-89)."""
    B, F_bins, T = 16, 256, 256
    s = O.cf_uniform("s.cfg2", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.cfg2")
    s2 = separated_input(s, mask)
    model = build_model(F_bins, T)
    model.set_inputs(s2, mask)
    model.forward_backward_no_update()
    torch.cuda.synchronize()
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 16))       # more threads are slower on the 256-CPU boxes (bench.py)
    try:
        oE, oG, oD = O.encoder_state(), O.decoder_state(), O.disc_state()
        ocap = O.step_no_update(oE, oG, oD, s2, mask)
        dcap = O.step_no_update(to64(O.encoder_state()), to64(O.decoder_state()), to64(O.disc_state()), s2.double(), mask.double())
    finally:
        torch.set_num_threads(nthr)
    report = check_against_oracle(model, ocap, dcap, oE, oG, oD)
    print("cfg2 gradient error vs fp64 (HIP, CPU fp32):", report)
    for grp, (e_hip, e_o32) in report.items():
        assert e_hip < 3 * e_o32, (grp, e_hip, e_o32)


def test_module_api_nchw_roundtrip():
    """reference-style use: modules called with NCHW tensors, reference GANLoss-style torch loss, torch.optim.Adam."""
    from viai_amd.networks import MelDecoder, MelDiscriminator, MelEncoder
    B, F_bins, T = 2, 80, 32
    E, G, D = MelEncoder().cuda(), MelDecoder().cuda(), MelDiscriminator().cuda()
    E.load_state_dict(O.encoder_state()); G.load_state_dict(O.decoder_state()); D.load_state_dict(O.disc_state())
    s = O.cf_uniform("s.tiny", (B, 1, F_bins, T))
    sc = s.cuda()
    feats = E(sc.view(B, F_bins, T))
    assert [tuple(f.shape) for f in feats] == [(2, 32, 40, 16), (2, 64, 20, 16), (2, 128, 10, 8), (2, 256, 5, 4), (2, 256, 1, 2)]
    fake = G(feats, sc.size())
    assert tuple(fake.shape) == (B, 1, F_bins, T)
    pred = D(fake)
    loss = torch.nn.BCELoss()(pred, torch.ones_like(pred)) + 100.0 * torch.nn.functional.l1_loss(fake, sc)
    opt = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=2e-4, betas=(0.5, 0.999))
    opt.zero_grad()
    loss.backward()
    opt.step()
    # same thing on the oracle
    oE, oG, oD = O._leafify(O.encoder_state()), O._leafify(O.decoder_state()), O.disc_state()
    ofe = O.encoder_forward(oE, s.view(B, F_bins, T))
    ofake = O.decoder_forward(oG, ofe, s.shape)
    opred = O.disc_forward(oD, ofake)
    oloss = O.gan_loss(opred, True) + 100.0 * O.l1_loss(ofake, s)
    assert relerr(fake, ofake) < TOL_FWD
    assert abs(loss.item() - oloss.item()) < 1e-4 * abs(oloss.item())
    (gw,) = torch.autograd.grad(oloss, oE["conv1.weight"])
    assert relerr(E.conv1.weight.grad, gw) < TOL_GRAD


def test_eval_mode_uses_running_stats():
    from viai_amd.networks import MelDiscriminator
    D = MelDiscriminator().cuda()
    sd = O.disc_state()
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = O.cf_uniform("ev." + k, tuple(sd[k].shape), 0.5, 1.5)
        if k.endswith("running_mean"):
            sd[k] = O.cf_uniform("ev." + k, tuple(sd[k].shape), -0.2, 0.2)
    D.load_state_dict(sd)
    D.eval()
    x = O.cf_uniform("ev.x", (2, 1, 16, 32))
    y = D(x.cuda())
    ref = O.disc_forward(sd, x, training=False)
    assert relerr(y, ref) < TOL_FWD


def test_full_step_with_adam_tracks_oracle_loosely(golden_dir):
    """chained steps WITH the Adam updates: losses only, loose tolerance (sign-flip chaos)."""
    B, F_bins, T = 2, 80, 32
    s = O.cf_uniform("s.tiny", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.tiny")
    model = build_model(F_bins, T)
    gold = np.load("%s/step_tiny.npz" % golden_dir)
    for it in range(3):
        model.set_inputs(s, mask)
        model.optimize_parameters(it)
        v = model.get_loss_items()
        assert abs(v[0] - float(gold["ch.step%d.loss_d" % it])) < 3e-2 * float(gold["ch.step%d.loss_d" % it])
        assert abs(v[1] - float(gold["ch.step%d.loss_g" % it])) < 3e-2 * float(gold["ch.step%d.loss_g" % it])


def test_one_adam_step_from_synced_state_matches_oracle():
    """params after ONE full step: compare only weights whose gradient is well above the rounding noise
    (Adam's first step is lr*sign(g); noise-level gradients flip sign between ANY two implementations)."""
    B, F_bins, T = 2, 80, 32
    s = O.cf_uniform("s.tiny", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.tiny")
    model = build_model(F_bins, T)
    model.set_inputs(s, mask)
    model.optimize_parameters(0)
    torch.cuda.synchronize()
    oE, oG, oD = O.encoder_state(), O.decoder_state(), O.disc_state()
    optG, optD = O.new_optimizers(oE, oG, oD)
    cap = O.train_step(oE, oG, oD, optG, optD, s, mask)
    init = O.disc_state()
    for k, p in model.netD.named_parameters():
        g = cap["grads_D"][k]
        sure = g.abs() > 1e-3 * g.abs().max()          # gradients far from the noise floor
        d_model = (p.detach().cpu() - init[k])[sure]
        d_orc = (oD[k] - init[k])[sure]
        assert relerr(d_model, d_orc) < 2e-2, k


SHADOWED_CHAIN = ("G.deconv1_1.bias", "G.deconv1_2.bias", "G.conv6_1.bias")


def test_parameters_after_one_two_three_adam_steps_match_reference_chain(golden_dir):
    """SURVEY.md section 8 row a14, "params after 1 and 3 Adam steps": three full steps (both fused-Adam updates) on the tie-free tiny
    input; after EVERY step, EVERY parameter tensor of E, G and D (nothing filtered except the three biases in front of train-mode
    BatchNorm, whose exact gradient is zero) is compared
      (a) at 256 strided samples per tensor with the chain the REFERENCE's modules + torch.optim.Adam produced
          (tools/make_goldens.py::chain_goldens -> tests/golden/step_chain.npz), and
      (b) in full with the oracle's chain run here on the CPU.
    Bound: 1e-3 relative per network after the first step (measured CPU-vs-CPU, oracle vs reference: 6e-5 .. 2.1e-4).  Adam's update
    is lr * m / sqrt(v), i.e. +-lr wherever a gradient is at the rounding-noise level or flips sign between steps, so two correct fp32
    implementations drift apart step by step: the reference and the oracle themselves are 8e-4 apart after two steps and 2.4e-3 after
    three (stored in the fixture as `oracle_vs_reference`); the later steps are therefore held to 2.5x that measured floor."""
    B, F_bins, T = 2, 80, 32
    gold = np.load("%s/step_chain.npz" % golden_dir)
    steps = int(gold["meta"][3])
    s = O.cf_uniform("s.tiny", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.tiny")
    s2 = O.separated_input(s, mask)
    model = build_model(F_bins, T)
    oE, oG, oD = O.encoder_state(), O.decoder_state(), O.disc_state()
    optG, optD = O.new_optimizers(oE, oG, oD)
    report = []
    for it in range(1, steps + 1):
        model.set_inputs(s2, mask)
        model.optimize_parameters(it - 1)
        v = model.get_loss_items()
        O.train_step(oE, oG, oD, optG, optD, s2, mask)
        floor = gold["step%d.oracle_vs_reference" % it]
        for i, (nm, mod, osd) in enumerate((("E", model.Mel_Encoder, oE), ("G", model.Mel_Decoder, oG), ("D", model.netD, oD))):
            bound = max(1e-3, 2.5 * float(floor[i]))
            num = den = onum = oden = 0.0
            for k, t in mod.state_dict().items():
                if "num_batches" in k:
                    continue
                ref = gold["step%d.%s.%s" % (it, nm, k)]
                got = O.strided_samples(t)
                if "running_" in k:
                    # statistics, not parameters: batch means / variances of maps with as few as 4 .. 12 elements per channel at this
                    # shape respond to the parameter drift several-fold (measured 7e-3 on deconv1_1_bn.running_mean at step 3)
                    assert np.linalg.norm(got - ref) < 5 * bound * (np.linalg.norm(ref) + 1e-30), (it, nm, k)
                    continue
                if (nm + "." + k) in SHADOWED_CHAIN:
                    assert np.abs(got - ref).max() < 3 * 2e-4 * it, (it, nm, k)       # noise-driven in torch, exactly still here
                    continue
                num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
                full = t.detach().cpu().double()
                onum += (full - osd[k].double()).pow(2).sum().item(); oden += osd[k].double().pow(2).sum().item()
            e_ref, e_orc = (num / den) ** 0.5, (onum / oden) ** 0.5
            report.append((it, nm, e_ref, e_orc, bound))
            assert e_ref < bound, ("vs reference samples", it, nm, e_ref, bound)
            assert e_orc < bound, ("vs oracle, full tensors", it, nm, e_orc, bound)
        for key, idx in (("loss_d", 0), ("loss_g", 1), ("loss_l1", 3)):
            ref = float(gold["step%d.%s" % (it, key)])
            assert abs(v[idx] - ref) < (2e-4 if it == 1 else 5e-3) * abs(ref), (it, key, v[idx], ref)
    print("chain (step, net, err vs reference samples, err vs oracle, bound):", report)


def test_weights_beyond_the_f16x2_range_raise_at_the_sync_point():
    """the f16x2 weight images clamp beyond |w| = 255.9: AudioModel reads max |w| of its arenas next to the loss scalars (the one
    host sync of an iteration, train_whole_sync.py:85) and raises instead of training on clipped weights."""
    B, F_bins, T = 2, 80, 32
    model = build_model(F_bins, T)
    model.set_inputs(O.cf_uniform("s.tiny", (B, 1, F_bins, T)), O.make_mask(B, T, "mask.tiny"))
    model.optimize_parameters(0)
    model.get_loss_items()                                  # fine
    with torch.no_grad():
        model.netD.conv3.weight[3, 4, 1, 1] = 300.0
    model.weights_changed()
    model.optimize_parameters(1)
    with pytest.raises(FloatingPointError, match="diverged"):
        model.get_loss_items()


def _scaled_states(f):
    E, G, D = O.encoder_state(), O.decoder_state(), O.disc_state()
    for sd in (E, G, D):
        for k, v in sd.items():
            if k.endswith("weight") and v.dim() == 4:
                sd[k] = v * f
    return E, G, D


@pytest.mark.parametrize("use_graph", [False, True])
def test_packed_weight_cache_follows_every_kind_of_weight_update(use_graph):
    """The conv kernels read cached packed weight images (ops._packed / ops.repack).  Every way the weights can change
    must reach them: checkpoint-style loads (also under hipGraph replay, whose graphs contain no per-layer packs), the
    fused Adam step, and plain in-place torch updates."""
    from viai_amd.model import AudioModel, StepConfig
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = 80, 32
    s = O.cf_uniform("pc.s", (2, 1, 80, 32), 0, 1).cuda()
    mask = O.make_mask(2, 32, "pc.mask").cuda()

    def fresh(states):
        m = AudioModel(hp, device="cuda", use_graph=False)
        m.load_states(*states)
        m.set_inputs(s, mask)
        return m
    a = AudioModel(hp, device="cuda", use_graph=use_graph)
    a.load_states(*_scaled_states(1.0))
    a.set_inputs(s, mask)
    for i in range(3):                                   # includes graph capture when use_graph
        a.optimize_parameters(i)
    # 1) load other weights into the SAME model: the next step must use them
    a.load_states(*_scaled_states(0.7))
    ref = fresh(_scaled_states(0.7))
    a.optimize_parameters(3)
    ref.optimize_parameters(0)
    assert relerr(a.fake, ref.fake) < 1e-5
    assert abs(a.losses[0].item() - ref.losses[0].item()) < 1e-5 * abs(ref.losses[0].item())
    # 2) the fused Adam step of step 3 changed the weights: step 4's forward must see them (compare with a model that
    #    received the post-step parameters through a load)
    sdE, sdG, sdD = (dict((k, v.clone()) for k, v in m.state_dict().items()) for m in (a.Mel_Encoder, a.Mel_Decoder, a.netD))
    ref2 = fresh((sdE, sdG, sdD))
    a.optimize_parameters(4)
    ref2.optimize_parameters(0)
    assert relerr(a.fake, ref2.fake) < 1e-5
    if not use_graph:
        # 3) a plain in-place torch update bumps the tensor version and is picked up without any call
        with torch.no_grad():
            a.Mel_Decoder.conv6_2.weight.mul_(0.5)
            a.Mel_Decoder.conv6_1.weight.mul_(0.5)
        ref3 = fresh(tuple(dict((k, v.clone()) for k, v in m.state_dict().items()) for m in (a.Mel_Encoder, a.Mel_Decoder, a.netD)))
        a.forward_backward_no_update()
        ref3.forward_backward_no_update()
        assert relerr(a.fake, ref3.fake) < 1e-5


def _rccl_one_rank_child(port, marker):
    """body of test_gradient_exchange_over_rccl_is_wired_into_the_step, in its own process (see there); whatever stops it is written
    next to the marker so that the parent can report it"""
    try:
        _rccl_one_rank_body(port, marker)
    except BaseException:                            # noqa: BLE001 -- the parent only sees the exit code otherwise
        import traceback
        with open(marker + ".err", "w") as f:
            f.write(traceback.format_exc())
        os._exit(1)


def _rccl_one_rank_body(port, marker):
    import torch.distributed as dist
    from viai_amd.model import AudioModel, StepConfig
    # eager communicator (device_id): RCCL sets the communicator up inside init_process_group, on the device this rank owns, instead of lazily
    # inside the first collective of the step
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda:0"))
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = 80, 32
    s = O.cf_uniform("rc.s", (2, 1, 80, 32), 0, 1).cuda()
    mask = O.make_mask(2, 32, "rc.mask").cuda()
    outs = []
    for force, graph in ((False, False), (True, False), (True, True)):
        m = AudioModel(hp, device="cuda", use_graph=graph)
        m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
        m._force_allreduce = force
        m.set_inputs(s, mask)
        for i in range(3):
            m.optimize_parameters(i)
        m.sync_pending_update()
        torch.cuda.synchronize()
        outs.append((m.fake.detach().clone(), m.arena_D.flat.clone(), m.arena_G.flat.clone()))
    # eager: the exchange of a one-rank group is the identity and the three-stream step is deterministic -> bitwise equal
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # graph mode: the capture warm-up (two real steps) is rolled back (model._capture snapshots and restores parameters, Adam
    # state and BatchNorm buffers), so three replayed steps land where three eager steps do
    for a, b in zip(outs[0], outs[2]):
        assert torch.isfinite(b).all()
        assert relerr(b, a) < 1e-5
    m.close()
    del m
    # leave the way bench.py's ranks do: every collective of this process complete (device sync + barrier), then the communicator down BEFORE
    # the interpreter starts tearing the HIP context down.  Rounds 2 / 3 retried this child and left through os._exit because it "aborted now
    # and then in RCCL's set-up or teardown"; the abort (1 of 20 runs, tools/rccl_loop.sh) was the process-group watchdog thread querying an
    # event while THIS thread was capturing the graph-mode model's step -- illegal under the default capture mode, fixed in model._capture
    # (capture_error_mode="thread_local" while a process group is up).  No retry, no hard exit.
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    with open(marker, "w") as f:
        f.write("ok")


def test_gradient_exchange_over_rccl_is_wired_into_the_step(tmp_path):
    """one-rank RCCL group on the GPU box: the two all-reduce points of the step (ddp.py, model._allreduce) run on the
    real backend, between the side-stream joins and the Adam kernels, and leave a one-rank result unchanged.  (The
    world-size-2 semantics are covered on CPU by tests/test_ddp_gloo.py and on the GPU over gloo by tests/test_ddp_gpu.py.)
    Runs in a spawned process (one process group per interpreter) that initialises the communicator eagerly and takes it down in order; a
    non-zero exit code of the child -- an abort in RCCL's set-up or teardown included -- fails the test."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    marker = str(tmp_path / "rccl_ok")
    p = ctx.Process(target=_rccl_one_rank_child, args=(port, marker))
    p.start()
    p.join(600)
    if p.is_alive():
        p.kill()
        pytest.fail("the RCCL child did not finish")
    why = open(marker + ".err").read() if os.path.exists(marker + ".err") else "no Python exception"
    assert os.path.exists(marker) and p.exitcode == 0, "the RCCL child failed (exit code %s): %s" % (p.exitcode, why[-1500:])


def test_graph_capture_leaves_training_state_untouched():
    """`_capture()` warms up with two real train steps.  They must not leak: after the first graph-mode step the
    parameters, Adam moments / step counter and BatchNorm buffers equal those after ONE eager step, and a re-capture for a new
    input shape does not add hidden steps either."""
    from viai_amd.model import AudioModel, StepConfig
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = 80, 32
    s = O.cf_uniform("gc.s", (2, 1, 80, 32), 0, 1).cuda()
    mask = O.make_mask(2, 32, "gc.mask").cuda()
    s2 = O.cf_uniform("gc.s2", (3, 1, 80, 32), 0, 1).cuda()
    mask2 = O.make_mask(3, 32, "gc.mask2").cuda()

    def run(graph):
        m = AudioModel(hp, device="cuda", use_graph=graph)
        m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
        m.set_inputs(s, mask)
        m.optimize_parameters(0)
        m.set_inputs(s2, mask2)                      # new shape: graph mode re-captures
        m.optimize_parameters(1)
        torch.cuda.synchronize()
        bn = m.netD.norm3
        return [m.arena_G.flat.clone(), m.arena_D.flat.clone(), m.optimizer_G.exp_avg.clone(), m.optimizer_D.exp_avg_sq.clone(),
                m.optimizer_G.state.clone(), bn.running_mean.clone(), bn.num_batches_tracked.clone().double()]
    eager, graph = run(False), run(True)
    assert float(eager[4][0]) == 2.0 and float(graph[4][0]) == 2.0            # Adam step counter: two steps, not six
    assert int(eager[6]) == int(graph[6]) == 6                                # norm3 saw 3 forwards per step
    for a, b in zip(eager, graph):
        assert relerr(b, a) < 1e-5


def test_launch_plan_replays_the_eager_step_bitwise():
    """csrc/plan.hip: the step recorded once (stream capture used as a recorder) and replayed from C issues the eager step's
    kernels with the eager step's arguments on the eager step's streams, so after four steps -- with a change of input shape
    (re-record) in the middle -- parameters, Adam state, BatchNorm buffers and the loss scalars are BIT-identical to the eager
    model's, and the plan really is multi-stream (side streams noted for the library's own launches)."""
    from viai_amd.model import AudioModel, StepConfig
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = 80, 32
    s = O.cf_uniform("pl.s", (2, 1, 80, 32), 0, 1).cuda()
    mask = O.make_mask(2, 32, "pl.mask").cuda()
    s2 = O.cf_uniform("pl.s2", (3, 1, 80, 32), 0, 1).cuda()
    mask2 = O.make_mask(3, 32, "pl.mask2").cuda()

    def run(plan):
        m = AudioModel(hp, device="cuda", use_plan=plan)
        m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
        m.set_inputs(s, mask)
        m.optimize_parameters(0)
        m.optimize_parameters(1)
        info = m.plan_info() if plan else None
        m.set_inputs(s2, mask2)
        m.optimize_parameters(2)
        m.optimize_parameters(3)
        losses = m.get_loss_items()
        bn = m.netD.norm3
        return [m.arena_G.flat.clone(), m.arena_D.flat.clone(), m.optimizer_G.exp_avg.clone(), m.optimizer_D.exp_avg_sq.clone(),
                m.optimizer_G.state.clone(), bn.running_mean.clone(), bn.running_var.clone(), m.fake.clone(),
                torch.tensor(losses)], info
    (eager, _), (plan, info) = run(False), run(True)
    assert len(info) == 3
    for nodes, kernels, noted, copies, fills, streams, events, waits in info:
        assert nodes == kernels + copies + fills and nodes > 0
        assert noted >= 0.8 * kernels                          # the library's launches carry their stream
    assert info[0][5] >= 3 and info[1][5] >= 2                 # D(real) + weight-gradient side streams are kept
    for a, b in zip(eager, plan):
        assert torch.equal(a, b)


def test_three_stream_step_is_bitwise_the_single_stream_step(monkeypatch):
    """Weight gradients on a side stream and D(real) on a third stream only reorder launches in time: every
    accumulation keeps its order (event-ordered streams), so losses, gradients and parameters after three steps are
    bit-identical to the fully serial step -- with no host synchronisation between the steps."""
    from viai_amd.model import AudioModel, StepConfig
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = 80, 32
    s = O.cf_uniform("st.s", (2, 1, 80, 32), 0, 1).cuda()
    mask = O.make_mask(2, 32, "st.mask").cuda()

    def run(wgrad, dreal):
        monkeypatch.setenv("VIAI_WGRAD_STREAM", wgrad)
        monkeypatch.setenv("VIAI_DREAL_STREAM", dreal)
        m = AudioModel(hp, device="cuda", use_graph=False)
        assert (m._wgrad_stream is not None) == (wgrad == "1") and (m._dreal_stream is not None) == (dreal == "1")
        m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
        m.set_inputs(s, mask)
        for i in range(3):
            m.optimize_parameters(i)
        torch.cuda.synchronize()
        return [m.fake.detach().clone(), m.losses.clone(), m.arena_D.grad.clone(), m.arena_G.grad.clone(),
                m.arena_D.flat.clone(), m.arena_G.flat.clone()]
    serial = run("0", "0")
    for cfg in (("1", "0"), ("1", "1"), ("1", "1")):
        for a, b in zip(serial, run(*cfg)):
            assert torch.equal(a, b), cfg
