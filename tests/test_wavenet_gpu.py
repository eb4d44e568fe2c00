"""GPU parity of the WaveNet vocoder path (teacher-forced forward/backward, MoL loss + sampler, incremental
synthesis) against the oracle and the goldens produced by the reference's wavenet_vocoder package."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import viai_oracle as O
from oracle import wavenet_oracle as W


def relerr(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def build(cfg=W.WNConfig, dropout=0.0):
    from viai_amd.wavenet import WaveNet
    net = WaveNet(out_channels=cfg.out_channels, layers=cfg.layers, stacks=cfg.stacks, residual_channels=cfg.residual_channels,
                  gate_channels=cfg.gate_channels, skip_out_channels=cfg.skip_out_channels, kernel_size=cfg.kernel_size, dropout=dropout,
                  cin_channels=cfg.cin_channels, gin_channels=-1, weight_normalization=True, upsample_conditional_features=True,
                  upsample_scales=list(cfg.upsample_scales), freq_axis_kernel_size=cfg.freq_axis_kernel_size, scalar_input=True)
    sd = W.wavenet_state(cfg)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    return net.cuda()


def test_teacher_forced_forward_loss_grads_match_reference_golden(golden_dir):
    from viai_amd.wavenet import DiscretizedMixturelogisticLoss
    gold = np.load(golden_dir + "/wavenet.npz")
    cfg = W.WNConfig
    B, T = 2, 64
    x = O.cf_uniform("wn.x", (B, 1, T), -1, 1)
    c = O.cf_uniform("wn.c", (B, cfg.cin_channels, T // 16), 0, 1)
    y = O.cf_uniform("wn.y", (B, T, 1), -1, 1)
    y[0, 3, 0], y[1, 5, 0] = -1.0, 1.0
    mask = torch.ones(B, T, 1); mask[1, T - 10:] = 0
    net = build().train()
    yh = net(x.cuda(), c.cuda())
    assert tuple(yh.shape) == (B, 30, T)
    assert relerr(yh, gold["yhat"]) < 1e-4
    loss = DiscretizedMixturelogisticLoss()(yh, y.cuda(), mask=mask.cuda())
    assert abs(loss.item() - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))
    loss.backward()
    params = dict(net.named_parameters())
    for k in gold.files:
        if k.startswith("g."):
            assert relerr(params[k[2:]].grad, gold[k]) < 2e-3, k
    # fused NHWC path (no (B,C,T) detour) gives the same loss
    net.zero_grad()
    from viai_amd.wavenet import mol_loss
    l2 = mol_loss(net.forward_nhwc(x.cuda(), c.cuda()), y.cuda(), mask.cuda())
    assert abs(l2.item() - loss.item()) < 1e-6 * abs(loss.item())


def test_mol_sampler_matches_reference_golden(golden_dir):
    from viai_amd.wavenet import mol_sample
    gold = np.load(golden_dir + "/wavenet.npz")
    B, T = 2, 64
    yh = torch.from_numpy(gold["yhat"])                                   # (B,30,T)
    u1 = O.cf_uniform("wn.u1", (B, T, 10), 1e-5, 1 - 1e-5)
    u2 = O.cf_uniform("wn.u2", (B, T), 1e-5, 1 - 1e-5)
    rows = torch.nn.functional.pad(yh.transpose(1, 2), (0, 2)).reshape(B, 1, T, 32).contiguous().cuda()
    s = mol_sample(rows, u1.cuda(), u2.cuda(), -7.0).reshape(B, T)
    assert relerr(s, gold["sample"]) < 1e-5


def test_mol_loss_edge_branches_against_oracle():
    """all four likelihood branches (y<-0.999, y>0.999, cdf_delta>1e-5, tiny cdf_delta) and the log-scale clamp."""
    from viai_amd.wavenet import mol_loss
    B, T = 2, 48
    yh = O.cf_uniform("ml.yh", (B, 30, T), -2, 2)
    yh[:, 20:30, :8] = 6.0                        # huge scales -> cdf_delta <= 1e-5 branch
    yh[:, 20:30, 8:12] = -40.0                    # below log_scale_min -> clamp (zero gradient)
    y = O.cf_uniform("ml.y", (B, T, 1), -0.99, 0.99)
    y[0, 0, 0], y[0, 20, 0], y[1, 30, 0] = -1.0, 1.0, 0.9995
    mask = torch.ones(B, T, 1)
    a = yh.clone().requires_grad_(True)
    ref = W.mol_loss(a, y, mask, 65536, math.log(1e-14))
    ref.backward()
    rows = torch.nn.functional.pad(yh.transpose(1, 2), (0, 2)).reshape(B, 1, T, 32).contiguous().cuda().requires_grad_(True)
    out = mol_loss(rows, y.cuda(), mask.cuda(), 65536, math.log(1e-14))
    assert abs(out.item() - ref.item()) < 2e-5 * abs(ref.item())
    out.backward()
    g = rows.grad[..., :30].reshape(B, T, 30).transpose(1, 2)
    assert relerr(g, a.grad) < 1e-4


@pytest.mark.parametrize("use_graph", [False, True])
def test_incremental_forward_matches_reference_golden(use_graph, golden_dir):
    """free-running synthesis (4 teacher-forced samples, then fed back) with the sampler's uniforms injected:
    the reference's incremental_forward output, sample for sample."""
    gold = np.load(golden_dir + "/wavenet.npz")
    cfg = W.WNConfig
    B, Tg = 2, 32
    cg = O.cf_uniform("wn.cg", (B, cfg.cin_channels, Tg // 16), 0, 1)
    v1 = O.cf_uniform("wn.v1", (B, Tg, 10), 1e-5, 1 - 1e-5)
    v2 = O.cf_uniform("wn.v2", (B, Tg), 1e-5, 1 - 1e-5)
    tin = O.cf_uniform("wn.tin", (B, 1, 4), -1, 1)
    net = build().eval()
    gen = net.incremental_forward(None, c=cg.cuda(), g=None, T=Tg, test_inputs=tin.cuda(), softmax=False, quantize=False,
                                  log_scale_min=-7.0, uniforms=(v1, v2), use_graph=use_graph)
    assert tuple(gen.shape) == (B, 1, Tg)
    assert relerr(gen, gold["gen"]) < 2e-4, relerr(gen, gold["gen"])


def test_incremental_equals_batch_forward_under_teacher_forcing():
    """upstream invariant (SURVEY.md §4 i): with every input teacher-forced, the step-by-step logits equal forward()."""
    cfg = W.WNConfig
    B, T = 2, 48
    x = O.cf_uniform("tf.x", (B, 1, T), -1, 1)
    c = O.cf_uniform("tf.c", (B, cfg.cin_channels, T // 16), 0, 1)
    net = build().eval()
    yh = net(x.cuda(), c.cuda())                                   # (B,30,T), logits at t from inputs <= t
    xin = torch.cat((torch.zeros(B, 1, 1), x[:, :, :-1]), 2)       # incremental step t consumes x_in[t]
    yh_shift = net(xin.cuda(), c.cuda())
    _, logits = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=xin.cuda(), log_scale_min=-7.0, use_graph=True, return_logits=True)
    assert relerr(logits.transpose(1, 2), yh_shift) < 1e-4


@pytest.mark.gpu
def test_global_conditioning_matches_reference_golden(golden_dir):
    """speaker-id global conditioning (wavenet.py:198-206,284-290; modules.py:140-145,195-199): teacher-forced
    forward and incremental synthesis against the reference's outputs (tests/golden/wavenet_g.npz)."""
    from viai_amd.wavenet import WaveNet
    cfg = W.WNConfigG
    gold = np.load(os.path.join(golden_dir, "wavenet_g.npz"))
    net = WaveNet(out_channels=cfg.out_channels, layers=cfg.layers, stacks=cfg.stacks, residual_channels=cfg.residual_channels,
                  gate_channels=cfg.gate_channels, skip_out_channels=cfg.skip_out_channels, kernel_size=cfg.kernel_size, dropout=0.0,
                  cin_channels=cfg.cin_channels, gin_channels=cfg.gin_channels, n_speakers=cfg.n_speakers,
                  upsample_scales=list(cfg.upsample_scales), freq_axis_kernel_size=cfg.freq_axis_kernel_size)
    sd = W.wavenet_state(cfg)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net = net.cuda().eval()
    B, T = 2, 48
    x = O.cf_uniform("wng.x", (B, 1, T), -1, 1).cuda()
    c = O.cf_uniform("wng.c", (B, cfg.cin_channels, T // 16), 0, 1).cuda()
    g = torch.tensor([[2], [0]], dtype=torch.long).cuda()
    with torch.no_grad():
        yh = net(x, c, g)
    assert relerr(yh.cpu(), torch.from_numpy(gold["yhat"])) < 1e-4
    Tg = 32
    cg = O.cf_uniform("wng.cg", (B, cfg.cin_channels, Tg // 16), 0, 1).cuda()
    v1 = O.cf_uniform("wng.v1", (B, Tg, 10), 1e-5, 1 - 1e-5)
    v2 = O.cf_uniform("wng.v2", (B, Tg), 1e-5, 1 - 1e-5)
    tin = O.cf_uniform("wng.tin", (B, 1, 3), -1, 1)
    for use_graph in (False, True):
        gen = net.incremental_forward(None, c=cg, g=g, T=Tg, test_inputs=tin, uniforms=(v1, v2), use_graph=use_graph, log_scale_min=-7.0)
        assert tuple(gen.shape) == (B, 1, Tg)
        assert relerr(gen.cpu(), torch.from_numpy(gold["gen"])) < 2e-3, relerr(gen.cpu(), torch.from_numpy(gold["gen"]))
    # and the speaker matters
    gen2 = net.incremental_forward(None, c=cg, g=torch.tensor([[1], [1]]).cuda(), T=Tg, test_inputs=tin, uniforms=(v1, v2), log_scale_min=-7.0)
    assert relerr(gen2.cpu(), torch.from_numpy(gold["gen"])) > 1e-2


def test_reference_size_wavenet_matches_reference_digests(golden_dir):
    """BASELINE.json configs[4] model size (24 layers / 4 stacks / 512 / 512 / 256 channels, hop 256: wavenet.py:62-175 selects other
    conv tiles -- K = 1536 -- than the 64-channel test config): teacher-forced forward, MoL loss and gradient digests at B1, T = 1024
    against tests/golden/wavenet_full.npz (tools/make_goldens.py --wavenet-full-only, the reference's wavenet_vocoder package)."""
    from viai_amd.wavenet import DiscretizedMixturelogisticLoss
    gold = np.load(golden_dir + "/wavenet_full.npz")
    cfg = W.WNConfigFull
    B, T = [int(v) for v in gold["meta"]]
    x = O.cf_uniform("wnf.x", (B, 1, T), -1, 1)
    c = O.cf_uniform("wnf.c", (B, cfg.cin_channels, T // 256), 0, 1)
    y = O.cf_uniform("wnf.y", (B, T, 1), -1, 1)
    net = build(cfg).train()
    assert sum(p.numel() for p in net.parameters()) == 24737396
    yh = net(x.cuda(), c.cuda())
    assert tuple(yh.shape) == (B, 30, T)
    assert relerr(yh[:, :, -64:], gold["yhat_tail"]) < 1e-4
    assert relerr(O.digest(yh.contiguous(), 256), gold["yhat.dg"]) < 1e-4
    loss = DiscretizedMixturelogisticLoss()(yh, y.cuda(), mask=torch.ones(B, T, 1).cuda())
    assert abs(loss.item() - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))
    loss.backward()
    params = dict(net.named_parameters())
    for k in gold.files:
        if k.startswith("g."):
            dg, ref = O.digest(params[k[2:-3]].grad), gold[k]
            assert abs(dg[2] - ref[2]) < 2e-3 * ref[2], (k, dg[2], ref[2])
            assert np.linalg.norm(dg[3:] - ref[3:]) < 4e-3 * np.linalg.norm(ref[3:]), k


def test_one_hot_input_wavenet_matches_reference_golden(golden_dir):
    """`scalar_input=False` (one-hot mu-law input, 256-way logits; wavenet.py:116-119,177-235): teacher-forced forward and
    cross-entropy gradients against the reference's module (tests/golden/wavenet_onehot.npz)."""
    from viai_amd.wavenet import WaveNet
    gold = np.load(golden_dir + "/wavenet_onehot.npz")
    cfg = W.WNConfigOneHot
    net = WaveNet(out_channels=cfg.out_channels, layers=cfg.layers, stacks=cfg.stacks, residual_channels=cfg.residual_channels,
                  gate_channels=cfg.gate_channels, skip_out_channels=cfg.skip_out_channels, kernel_size=cfg.kernel_size, dropout=0.0,
                  cin_channels=cfg.cin_channels, gin_channels=-1, weight_normalization=True, upsample_conditional_features=True,
                  upsample_scales=list(cfg.upsample_scales), freq_axis_kernel_size=cfg.freq_axis_kernel_size, scalar_input=False)
    sd = W.wavenet_state(cfg)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    net = net.cuda().train()
    B, T = 2, 64
    idx = (O.cf_uniform("wno.idx", (B, T), 0, 1) * cfg.out_channels).long().clamp(max=cfg.out_channels - 1)
    x = torch.nn.functional.one_hot(idx, cfg.out_channels).float().transpose(1, 2).contiguous()
    c = O.cf_uniform("wno.c", (B, cfg.cin_channels, T // 16), 0, 1)
    tgt = (O.cf_uniform("wno.tgt", (B, T), 0, 1) * cfg.out_channels).long().clamp(max=cfg.out_channels - 1)
    yh = net(x.cuda(), c.cuda())
    assert tuple(yh.shape) == (B, cfg.out_channels, T)
    assert relerr(yh, gold["yhat"]) < 1e-4
    loss = torch.nn.functional.cross_entropy(yh, tgt.cuda())
    assert abs(loss.item() - float(gold["loss"])) < 1e-4 * float(gold["loss"])
    loss.backward()
    params = dict(net.named_parameters())
    for k in gold.files:
        if k.startswith("g."):
            assert relerr(params[k[2:]].grad, gold[k]) < 2e-3, k
    sm = net(x.cuda(), c.cuda(), softmax=True)              # the reference's `self.softmax(x, dim=1)` is a latent TypeError; this is its intent
    assert relerr(sm, torch.softmax(torch.from_numpy(gold["yhat"]), 1)) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("skip", [256, 320])
def test_synthesis_paths_agree_with_batch_forward_at_wide_layers(skip):
    """The two launch forms of a synthesis time step -- `viai_wavenet_synth_run` (host loop, time index by value; the default) and
    `viai_wavenet_synth_step` under a HIP graph (time index on the device) -- run the same kernels and must give the same logits
    bit for bit, and both must equal the teacher-forced forward().  Reference-like widths (512 residual / 512 gate channels:
    the block-per-row gate kernel with two 16-byte chunks per thread); skip = 256 takes the 16-wave head kernel, skip = 320 the
    generic one (S > 256)."""
    from viai_amd.wavenet import WaveNet
    torch.manual_seed(3)
    net = WaveNet(out_channels=30, layers=4, stacks=2, residual_channels=512, gate_channels=512, skip_out_channels=skip,
                  kernel_size=3, dropout=0.0, cin_channels=80, gin_channels=-1, upsample_conditional_features=True,
                  upsample_scales=[4, 4], scalar_input=True).cuda().eval()
    B, T = 8, 32
    x = torch.rand(B, 1, T) * 2 - 1
    c = torch.rand(B, 80, T // 16)
    xin = torch.cat((torch.zeros(B, 1, 1), x[:, :, :-1]), 2)
    with torch.no_grad():
        yh = net(xin.cuda(), c.cuda())
    out_r, log_r = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=xin.cuda(), log_scale_min=-7.0, use_graph=False, return_logits=True)
    out_g, log_g = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=xin.cuda(), log_scale_min=-7.0, use_graph=True, return_logits=True)
    assert torch.equal(log_r, log_g)
    assert relerr(log_r.transpose(1, 2), yh) < 1e-4


def build_cfg(cfg, tag):
    from viai_amd.wavenet import WaveNet
    net = WaveNet(out_channels=cfg.out_channels, layers=cfg.layers, stacks=cfg.stacks, residual_channels=cfg.residual_channels,
                  gate_channels=cfg.gate_channels, skip_out_channels=cfg.skip_out_channels, kernel_size=cfg.kernel_size, dropout=0.0,
                  cin_channels=cfg.cin_channels, gin_channels=-1, weight_normalization=True, upsample_conditional_features=True,
                  upsample_scales=list(cfg.upsample_scales), freq_axis_kernel_size=cfg.freq_axis_kernel_size, scalar_input=True)
    sd = W.wavenet_state(cfg, tag=tag)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    return net.cuda()


@pytest.mark.parametrize("use_graph", [False, True])
def test_incremental_synthesis_at_reference_depth_matches_reference_golden(use_graph, golden_dir):
    """BASELINE.json configs[4]'s layer stack -- 24 layers / 4 stacks, dilations 1 .. 32 (wavenet.py:116-131), the ring buffers of
    conv.py:17-46 -- at reduced width, T = 160 (every ring wraps at least twice), B = 2, sampler uniforms injected: the output of
    the REFERENCE's `incremental_forward` (wavenet.py:237-364; tests/golden/wavenet_deep.npz), teacher-forced over the whole length
    and free-running after four teacher-forced samples, sample for sample."""
    gold = np.load(golden_dir + "/wavenet_deep.npz")
    cfg = W.WNConfigDeep
    B, T = int(gold["meta"][0]), int(gold["meta"][1])
    assert T >= 130 and (cfg.layers, cfg.stacks) == (24, 4)
    c = O.cf_uniform("wnd.c", (B, cfg.cin_channels, T // 16), 0, 1)
    v1 = O.cf_uniform("wnd.v1", (B, T, 10), 1e-5, 1 - 1e-5)
    v2 = O.cf_uniform("wnd.v2", (B, T), 1e-5, 1 - 1e-5)
    xin = O.cf_uniform("wnd.xin", (B, 1, T), -1, 1)
    net = build_cfg(cfg, "WND.").eval()
    assert [f.conv.dilation[0] for f in net.conv_layers] == [1, 2, 4, 8, 16, 32] * 4
    gen_tf = net.incremental_forward(None, c=c.cuda(), g=None, T=T, test_inputs=xin.cuda(), softmax=False, quantize=False,
                                     log_scale_min=-7.0, uniforms=(v1, v2), use_graph=use_graph)
    assert tuple(gen_tf.shape) == (B, 1, T)
    assert relerr(gen_tf, gold["gen_tf"]) < 2e-4, relerr(gen_tf, gold["gen_tf"])
    gen_free = net.incremental_forward(None, c=c.cuda(), g=None, T=T, test_inputs=xin[:, :, :4].contiguous().cuda(), softmax=False,
                                       quantize=False, log_scale_min=-7.0, uniforms=(v1, v2), use_graph=use_graph)
    # free-running: each sample is fed back, so rounding differences compound over 156 steps through 24 layers
    assert relerr(gen_free, gold["gen_free"]) < 2e-3, relerr(gen_free, gold["gen_free"])
    # and the first 64 fed-back samples (before any compounding) at the single-step tolerance
    assert relerr(gen_free[:, :, :64], gold["gen_free"][:, :, :64]) < 2e-4


def test_incremental_equals_batch_forward_at_reference_size(monkeypatch):
    """configs[4] as benchmarked: 24 layers / 4 stacks / 512 residual + gate / 256 skip channels (24.7 M parameters), B = 8 streams,
    T = 256 = one conditioning frame at hop 256 (the dilation-32 ring of 65 slots wraps three times): with every input
    teacher-forced the step-by-step logits of all three launch forms equal the teacher-forced forward() (SURVEY.md section 4 invariant i;
    wavenet.py:268-280).  The two chain forms (C loop / captured graph) agree bit for bit; the pipelined form (round 6: one persistent launch,
    csrc/wavenet_pipe.hip -- the default at this size) sums the same products in another order and agrees with them to fp32 rounding."""
    from viai_amd import _lib
    cfg = W.WNConfigFull
    net = build_cfg(cfg, "WN.").eval()
    assert sum(p.numel() for p in net.parameters()) == 24737396
    B, T = 8, 256
    x = O.cf_uniform("wnf8.x", (B, 1, T), -1, 1)
    c = O.cf_uniform("wnf8.c", (B, cfg.cin_channels, 1), 0, 1)
    xin = torch.cat((torch.zeros(B, 1, 1), x[:, :, :-1]), 2)
    with torch.no_grad():
        yh = net(xin.cuda(), c.cuda())
    u = (O.cf_uniform("wnf8.u1", (B, T, 10), 1e-5, 1 - 1e-5), O.cf_uniform("wnf8.u2", (B, T), 1e-5, 1 - 1e-5))
    lib = _lib.load()
    ran = []
    real = lib.viai_wn_pipe_run
    monkeypatch.setattr(lib, "viai_wn_pipe_run", lambda *a: (ran.append(a[-3:-1]), real(*a))[1], raising=False)
    out_p, log_p = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=xin.cuda(), log_scale_min=-7.0, use_graph=False,
                                           return_logits=True, uniforms=u)
    assert ran == [(0, T)]                                       # the pipelined kernel took the whole call
    monkeypatch.setenv("VIAI_WN_PIPE", "0")
    out_r, log_r = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=xin.cuda(), log_scale_min=-7.0, use_graph=False,
                                           return_logits=True, uniforms=u)
    out_g, log_g = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=xin.cuda(), log_scale_min=-7.0, use_graph=True,
                                           return_logits=True, uniforms=u)
    assert len(ran) == 1
    assert torch.equal(log_r, log_g) and torch.equal(out_r, out_g)
    assert relerr(log_p, log_r) < 1e-5 and relerr(out_p, out_r) < 1e-4, (relerr(log_p, log_r), relerr(out_p, out_r))
    for lg in (log_r, log_p):
        assert relerr(lg.transpose(1, 2), yh) < 1e-4, relerr(lg.transpose(1, 2), yh)
        assert relerr(lg[:, 130:].transpose(1, 2), yh[:, :, 130:]) < 1e-4            # the steps after every ring has wrapped


def test_pipelined_synthesis_free_running_matches_the_chain_and_continues_across_launches(monkeypatch):
    """free-running synthesis (every sample feeds the next time step: the hand-off from the sampler back to stage 0) at the reference size, B = 4
    streams, in two launches (time steps [0, 100) then [100, 256): rings and tags carry over) against the chain of launches, which the reference goldens
    pin (tests above).  Autoregression amplifies rounding, so the first 48 samples are held to 1e-4 and the whole signal to 2e-3."""
    from viai_amd import _lib
    cfg = W.WNConfigFull
    net = build_cfg(cfg, "WN.").eval()
    B, T = 4, 256
    c = O.cf_uniform("wnp4.c", (B, cfg.cin_channels, 1), 0, 1)
    u = (O.cf_uniform("wnp4.u1", (B, T, 10), 1e-5, 1 - 1e-5), O.cf_uniform("wnp4.u2", (B, T), 1e-5, 1 - 1e-5))
    timing = {"warmup": 100}
    out_p = net.incremental_forward(None, c=c.cuda(), T=T, log_scale_min=-7.0, uniforms=u, timing=timing)
    assert timing.get("form") == "pipe" and timing["steps"] == T - 100
    monkeypatch.setenv("VIAI_WN_PIPE", "0")
    out_r = net.incremental_forward(None, c=c.cuda(), T=T, log_scale_min=-7.0, uniforms=u)
    assert float(out_r.abs().max()) > 0.01
    assert relerr(out_p[:, :, :48], out_r[:, :, :48]) < 1e-4, relerr(out_p[:, :, :48], out_r[:, :, :48])
    assert relerr(out_p, out_r) < 2e-3, relerr(out_p, out_r)


@pytest.mark.parametrize("B", [1, 2])
def test_pipelined_synthesis_with_few_streams_and_a_teacher_forced_prefix(B, monkeypatch):
    """fewer streams than stages in flight (B = 1: the pipeline holds ONE token, every stage waits for the whole revolution; B = 2), and a teacher-forced prefix of
    40 samples followed by free-running synthesis (the switch from `test_inputs` to the sampler's own output inside one launch): the pipelined form against the
    chain of launches, same injected uniforms."""
    cfg = W.WNConfigFull
    net = build_cfg(cfg, "WN.").eval()
    T = 256
    c = O.cf_uniform("wnp%d.c" % B, (B, cfg.cin_channels, 1), 0, 1)
    tin = O.cf_uniform("wnp%d.x" % B, (B, 1, 40), -1, 1)
    u = (O.cf_uniform("wnp%d.u1" % B, (B, T, 10), 1e-5, 1 - 1e-5), O.cf_uniform("wnp%d.u2" % B, (B, T), 1e-5, 1 - 1e-5))
    timing = {"warmup": 0}
    out_p = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=tin.cuda(), log_scale_min=-7.0, uniforms=u, timing=timing)
    assert timing.get("form") == "pipe"
    # fresh launches again: at t = 1 of a one-stream run a stage's sibling blocks may still be loading their weights when the first block asks for their
    # x(0) columns (a past tap is a wait, not an assertion: found as a one-in-two failure of this case); the arithmetic has a fixed order, so repeats are bitwise
    for _ in range(3):
        again = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=tin.cuda(), log_scale_min=-7.0, uniforms=u)
        assert torch.equal(again, out_p)
    monkeypatch.setenv("VIAI_WN_PIPE", "0")
    out_r = net.incremental_forward(None, c=c.cuda(), T=T, test_inputs=tin.cuda(), log_scale_min=-7.0, uniforms=u)
    assert tuple(out_p.shape) == (B, 1, T) and float(out_r.abs().max()) > 0.01
    assert relerr(out_p[:, :, :64], out_r[:, :, :64]) < 1e-4, relerr(out_p[:, :, :64], out_r[:, :, :64])
    assert relerr(out_p, out_r) < 2e-3, relerr(out_p, out_r)


@pytest.mark.gpu
def test_pipelined_synthesis_with_more_streams_than_the_chain_form_takes():
    """12 streams in one launch (the chain of launches is built for 1, 2, 4 or 8): the streams are independent and a stream's arithmetic does not depend on how
    many travel with it, so streams 0 - 7 and 4 - 11 must be BITWISE the two 8-stream runs on the same conditioning and uniforms."""
    cfg = W.WNConfigFull
    net = build_cfg(cfg, "WN.").eval()
    B, T = 12, 256
    c = O.cf_uniform("wn12.c", (B, cfg.cin_channels, 1), 0, 1)
    u1, u2 = O.cf_uniform("wn12.u1", (B, T, 10), 1e-5, 1 - 1e-5), O.cf_uniform("wn12.u2", (B, T), 1e-5, 1 - 1e-5)
    timing = {"warmup": 0}
    out = net.incremental_forward(None, c=c.cuda(), T=T, log_scale_min=-7.0, uniforms=(u1, u2), timing=timing)
    assert timing.get("form") == "pipe" and tuple(out.shape) == (B, 1, T) and float(out.abs().max()) > 0.01
    for lo in (0, 4):
        sl = slice(lo, lo + 8)
        part = net.incremental_forward(None, c=c[sl].cuda(), T=T, log_scale_min=-7.0, uniforms=(u1[sl], u2[sl]))
        assert torch.equal(out[sl], part), lo
