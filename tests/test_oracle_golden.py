"""CPU: the oracle restatement against the golden vectors generated from the REFERENCE's own modules
(tools/make_goldens.py), plus oracle self-consistency.  Runs everywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import viai_oracle as O


def relerr(a, b):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_closed_form_generator_is_bit_stable():
    u = O.cf_uniform("probe", (7,), 0, 1).numpy()
    # values pinned once; an integer hash, so identical on every machine
    assert u.dtype == np.float32
    assert np.all((u >= 0) & (u < 1))
    again = O.cf_uniform("probe", (7,), 0, 1).numpy()
    assert np.array_equal(u, again)
    assert not np.array_equal(u, O.cf_uniform("probe2", (7,), 0, 1).numpy())


def test_state_tables_match_reference_key_counts():
    # 30 / 113 / 25 state_dict keys
    assert len(O.encoder_state()) == 30
    assert len(O.decoder_state()) == 113
    assert len(O.disc_state()) == 25
    assert tuple(O.decoder_state()["deconv1_1.weight"].shape) == (256, 256, 3, 3)
    assert tuple(O.decoder_state()["convblock4.conv4_0.weight"].shape) == (128, 32, 3, 3)
    assert tuple(O.disc_state()["conv1.weight"].shape) == (64, 1, 1, 4)


def test_layer_goldens(golden_dir):
    g = np.load(golden_dir + "/layers.npz")
    x = O.cf_uniform("lg.x", (2, 32, 3, 8), -1, 1)
    w = O.cf_std("lg.w", (32, 32, 3, 3), 0.1)
    b = O.cf_uniform("lg.b", (32,), -0.1, 0.1)
    assert relerr(F.conv_transpose2d(x, w, b, stride=1, padding=(0, 1)), g["convT_p01"]) < 1e-6
    assert relerr(O.bilinear_ac(x, (7, 20)), g["bilinear_3x8_to_7x20"]) < 1e-6
    p = torch.tensor([[0.0, 1.0, 0.25, 0.999999, 1e-30, 0.5]])
    for lsgan, nm in ((False, "bce"), (True, "mse")):
        for real, rn in ((False, "fake"), (True, "real")):
            ref = float(g["gan_%s_%s" % (nm, rn)])
            assert abs(O.gan_loss(p, real, lsgan).item() - ref) <= 1e-6 * max(1.0, abs(ref))
    f1 = O.cf_uniform("lg.f1", (6, 256), -1, 1)
    f2 = O.cf_uniform("lg.f2", (6, 256), -1, 1)
    for mv in (False, True):
        ref = float(g["l2c_mv%d" % mv])
        assert abs(O.l2_contrastive(f1, f2, 12.0, mv).item() - ref) <= 1e-5 * abs(ref)


def test_adam_golden(golden_dir):
    g = np.load(golden_dir + "/adam.npz")
    sd = {"p": O.cf_uniform("adam.p", (4096,), -1, 1)}
    opt = O.Adam(sd)
    for t in range(3):
        gr = O.cf_uniform("adam.g%d" % t, (4096,), -1, 1) * (10.0 ** O.cf_uniform("adam.e%d" % t, (4096,), -9, 0))
        opt.step(sd, {"p": gr})
        assert relerr(sd["p"], g["p_after_%d" % (t + 1)]) < 1e-6


def test_step_no_update_matches_reference_golden_tiny(golden_dir):
    """full tensors of the tiny case: fake, feature maps, D outputs, losses, gradient digests, BN buffers."""
    g = np.load(golden_dir + "/step_tiny.npz")
    B, F_bins, T, _ = [int(v) for v in g["meta"]]
    s = O.cf_uniform("s.tiny", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.tiny")
    E, G, D = O.encoder_state(), O.decoder_state(), O.disc_state()
    cap = O.step_no_update(E, G, D, s, mask)
    for k in ("fake", "pred_fake_d", "pred_real", "pred_fake_g"):
        assert relerr(cap[k], g["nu." + k]) < 5e-5, k
    assert relerr(cap["d_fake"], g["nu.d_fake"]) < 1e-3
    for i, f in enumerate(cap["feats"]):
        assert relerr(f, g["nu.feat%d" % i]) < 5e-5
    for k in ("loss_d", "loss_g", "loss_g_gan", "loss_l1"):
        assert abs(cap[k].item() - float(g["nu." + k])) < 1e-5 * abs(float(g["nu." + k])), k
    for grp in ("grads_D", "grads_E", "grads_G"):
        for k, gr in cap[grp].items():
            key = "nu.%s.%s.dg" % (grp, k)
            if gr is None:
                assert key not in g.files
                continue
            if k in ("deconv1_1.bias", "deconv1_2.bias", "conv6_1.bias") and grp == "grads_G":
                continue
            assert relerr(O.digest(gr), g[key]) < 5e-3, key
    for sd, nm in ((E, "E"), (G, "G"), (D, "D")):
        for k, v in sd.items():
            if "running_" in k:
                assert relerr(v, g["nu.state.%s.%s" % (nm, k)]) < 1e-4, k
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(g["nu.state.%s.%s" % (nm, k)])


@pytest.mark.parametrize("name", ["cfg1"])
def test_step_no_update_matches_reference_golden_digests(name, golden_dir):
    g = np.load(golden_dir + "/step_%s.npz" % name)
    B, F_bins, T, _ = [int(v) for v in g["meta"]]
    s = O.cf_uniform("s.%s" % name, (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.%s" % name)
    cap = O.step_no_update(O.encoder_state(), O.decoder_state(), O.disc_state(), s, mask)
    assert relerr(O.digest(cap["fake"]), g["nu.fake.dg"]) < 5e-5
    for k in ("loss_d", "loss_g", "loss_l1"):
        assert abs(cap[k].item() - float(g["nu." + k])) < 1e-5 * abs(float(g["nu." + k]))
    assert relerr(O.digest(cap["grads_D"]["conv3.weight"]), g["nu.grads_D.conv3.weight.dg"]) < 5e-3


def test_chained_steps_track_reference_losses(golden_dir):
    g = np.load(golden_dir + "/step_tiny.npz")
    s = O.cf_uniform("s.tiny", (2, 1, 80, 32))
    mask = O.make_mask(2, 32, "mask.tiny")
    E, G, D = O.encoder_state(), O.decoder_state(), O.disc_state()
    optG, optD = O.new_optimizers(E, G, D)
    for it in range(3):
        cap = O.train_step(E, G, D, optG, optD, s, mask)
        for k in ("loss_d", "loss_g", "loss_l1"):
            ref = float(g["ch.step%d.%s" % (it, k)])
            assert abs(cap[k].item() - ref) < 3e-2 * abs(ref), (it, k)


def test_transposed_conv_is_flipped_conv():
    """self-consistency (SURVEY.md §4 ii): stride-1 ConvTranspose2d == Conv2d with flipped, transposed weights."""
    x = O.cf_uniform("tc.x", (2, 8, 5, 6), -1, 1)
    w = O.cf_std("tc.w", (8, 4, 3, 3), 0.2)
    a = F.conv_transpose2d(x, w, None, stride=1, padding=(0, 1))
    b = F.conv2d(x, w.flip(2, 3).transpose(0, 1), None, stride=1, padding=(2, 1))
    assert relerr(a, b) < 1e-6


def test_mask_policy():
    m = O.make_mask(8, 256, "mp")
    assert tuple(m.shape) == (8, 1, 1, 256)
    assert torch.all((m == 0) | (m == 1))
    assert torch.all(m.sum(dim=3) == 256 - 64)


def test_oracle_reproduces_reference_digests_at_benchmark_size(golden_dir):
    """BASELINE.json configs[1] (16 x 256 x 256): the oracle's forward tensors / losses / BatchNorm statistics against the digests
    the reference's modules produced (tools/make_goldens.py --cfg2-only).  ~20 s of CPU; this is also the function bench.py's
    cpu_baseline leg times."""
    g = np.load(golden_dir + "/step_cfg2.npz")
    B, F_bins, T, _ = [int(v) for v in g["meta"]]
    assert (B, F_bins, T) == (16, 256, 256)
    s = O.cf_uniform("s.cfg2", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.cfg2")
    E, G, D = O.encoder_state(), O.decoder_state(), O.disc_state()
    cap = O.step_no_update(E, G, D, s, mask)
    for k in ("fake", "pred_fake_d", "pred_real", "pred_fake_g"):
        assert relerr(O.digest(cap[k]), g["nu.%s.dg" % k]) < 5e-5, k
    for i, f in enumerate(cap["feats"]):
        assert relerr(O.digest(f), g["nu.feat%d.dg" % i]) < 5e-5
    for k in ("loss_d", "loss_g", "loss_g_gan", "loss_l1"):
        assert abs(cap[k].item() - float(g["nu." + k])) < 2e-5 * abs(float(g["nu." + k])), k
    for sd, nm in ((E, "E"), (G, "G"), (D, "D")):
        for k, v in sd.items():
            if "running_" in k:
                assert relerr(v, g["nu.state.%s.%s" % (nm, k)]) < 1e-4, k


def test_wavenet_helper_symbols_match_reference_golden(golden_dir):
    """`receptive_field_size` (wavenet_vocoder/wavenet.py:41-59), `sequence_mask` (loss_functions.py:11-21), `to_one_hot`
    (wavenet_vocoder/mixture.py:108-114): values produced by the reference's functions (tests/golden/wavenet_deep.npz)."""
    import torch
    from viai_amd import losses, wavenet
    gold = np.load(golden_dir + "/wavenet_deep.npz")
    assert [wavenet.receptive_field_size(24, 4, 3), wavenet.receptive_field_size(6, 2, 3, lambda x: 1),
            wavenet.receptive_field_size(4, 2, 2)] == list(gold["rf"]) and int(gold["rf"][0]) == 505
    with pytest.raises(AssertionError):
        wavenet.receptive_field_size(5, 2, 3)
    lens = torch.tensor([3, 5, 1])
    assert np.array_equal(losses.sequence_mask(lens).numpy(), gold["seqmask"])
    assert np.array_equal(wavenet.sequence_mask(lens, 6).numpy(), gold["seqmask6"])
    assert np.array_equal(wavenet.to_one_hot(torch.tensor([[1, 0, 3], [2, 2, 0]]), 4).numpy(), gold["onehot"])
    assert np.array_equal(wavenet.to_one_hot(torch.tensor([2, 0]), 3, 0.5).numpy(), gold["onehot_fill"])


def test_wavenet_oracle_reproduces_reference_incremental_synthesis_at_reference_depth(golden_dir):
    """24 layers / 4 stacks (dilations 1 .. 32), T = 160: the reference's `incremental_forward` (wavenet.py:237-364), teacher-forced
    over the whole length, equals the oracle's batch forward + sampler (causality) -- pins the oracle at every dilation."""
    from oracle import wavenet_oracle as W
    gold = np.load(golden_dir + "/wavenet_deep.npz")
    cfg = W.WNConfigDeep
    B, T = int(gold["meta"][0]), int(gold["meta"][1])
    assert (cfg.layers, cfg.stacks) == (24, 4) == (int(gold["meta"][2]), int(gold["meta"][3])) and T >= 130
    sd = W.wavenet_state(cfg, tag="WND.")
    c = O.cf_uniform("wnd.c", (B, cfg.cin_channels, T // 16), 0, 1)
    v1 = O.cf_uniform("wnd.v1", (B, T, 10), 1e-5, 1 - 1e-5)
    v2 = O.cf_uniform("wnd.v2", (B, T), 1e-5, 1 - 1e-5)
    xin = O.cf_uniform("wnd.xin", (B, 1, T), -1, 1)
    with torch.no_grad():
        yh = W.wavenet_forward(sd, xin, c, cfg)
    assert relerr(O.digest(yh, 256), gold["yhat_tf.dg"]) < 1e-5
    assert relerr(W.mol_sample(yh, v1, v2, -7.0), gold["gen_tf"][:, 0]) < 1e-4


def test_wavenet_ring_buffer_oracle_matches_reference_incremental_synthesis(golden_dir):
    """the O(T) ring-buffer restatement of conv.py:17-46 / wavenet.py:237-364 (bench.py's configs[4] `cpu_baseline`) against both
    reference runs of tests/golden/wavenet_deep.npz: teacher-forced and free-running after four forced samples."""
    from oracle import wavenet_oracle as W
    gold = np.load(golden_dir + "/wavenet_deep.npz")
    cfg = W.WNConfigDeep
    B, T = int(gold["meta"][0]), int(gold["meta"][1])
    sd = W.wavenet_state(cfg, tag="WND.")
    c = O.cf_uniform("wnd.c", (B, cfg.cin_channels, T // 16), 0, 1)
    v1 = O.cf_uniform("wnd.v1", (B, T, 10), 1e-5, 1 - 1e-5)
    v2 = O.cf_uniform("wnd.v2", (B, T), 1e-5, 1 - 1e-5)
    xin = O.cf_uniform("wnd.xin", (B, 1, T), -1, 1)
    assert relerr(W.incremental_forward_ring(sd, c, T, v1, v2, cfg, test_inputs=xin), gold["gen_tf"]) < 1e-4
    assert relerr(W.incremental_forward_ring(sd, c, T, v1, v2, cfg, test_inputs=xin[:, :, :4]), gold["gen_free"]) < 1e-3
    # and the small configuration's reference run (tests/golden/wavenet.npz), incl. global conditioning
    g2 = np.load(golden_dir + "/wavenet.npz")
    cg = O.cf_uniform("wn.cg", (2, W.WNConfig.cin_channels, 2), 0, 1)
    gen = W.incremental_forward_ring(W.wavenet_state(W.WNConfig), cg, 32, O.cf_uniform("wn.v1", (2, 32, 10), 1e-5, 1 - 1e-5),
                                     O.cf_uniform("wn.v2", (2, 32), 1e-5, 1 - 1e-5), W.WNConfig, test_inputs=O.cf_uniform("wn.tin", (2, 1, 4), -1, 1))
    assert relerr(gen, g2["gen"]) < 1e-4


def test_d_bn1_bias_is_the_ill_conditioned_gradient():
    """The claim behind tests/test_networks_gpu.py::TINY_ILL_CONDITIONED, shown on the CPU: at the tiny shape the gradient of D.bn1.bias is a sum over
    2 560 pixels that nearly cancels, and ANY fp32 arithmetic pays for it -- the oracle's own fp32 run is an order of magnitude further from its fp64 run
    on that tensor than on the median D tensor.  (The HIP path's f16x2 products and P16 storage pay the same conditioning with a larger constant; that is
    what the relaxed per-tensor floor of exactly this tensor covers, and nothing else may use it.)"""
    B, F_bins, T = 2, 80, 32
    s = O.cf_uniform("s.tiny", (B, 1, F_bins, T))
    mask = O.make_mask(B, T, "mask.tiny")
    s2 = O.separated_input(s, mask)

    def to64(sd):
        return type(sd)((k, (v.double() if v.is_floating_point() else v.clone())) for k, v in sd.items())
    c32 = O.step_no_update(O.encoder_state(), O.decoder_state(), O.disc_state(), s2, mask)
    c64 = O.step_no_update(to64(O.encoder_state()), to64(O.decoder_state()), to64(O.disc_state()), s2.double(), mask.double())
    err = {k: float((c32["grads_D"][k].double() - c64["grads_D"][k]).norm() / c64["grads_D"][k].norm()) for k in c64["grads_D"]}
    med = sorted(err.values())[len(err) // 2]
    assert err["bn1.bias"] == max(err.values()) and err["bn1.bias"] > 10 * med, err
