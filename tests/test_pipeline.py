"""The callers either side of the train-step path: lr schedules, EMA, retrieval metrics,
device batch assembly, inverse mel.  CPU tests pin oracle/pipeline_oracle.py and the host-side schedule functions to
the reference-generated goldens (tests/golden/pipeline.npz, tools/make_goldens.py::pipeline_goldens); GPU tests
compare the HIP kernels (through libviai_hip.so) with the oracle.  frames_prep / slice_clips / inv_mel restate
Data_loaders/audio_loader.py:185-245,471-523 and utils/audio.py:135-144; since round 2 they are PINNED too:
tests/golden/loader.npz holds what the reference's own sample_data_new / collate_fn / _denormalize / _db_to_amp / _amp_to_db /
_normalize / lws_num_frames / lws_pad_lr returned on synthetic closed-form frames and mels
(tools/make_goldens.py::loader_goldens: empty stand-ins for the packages those functions never call, a cv2 harness that
serves the synthetic frames)."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as P
from oracle import viai_oracle as O


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "pipeline.npz"))


def test_lr_schedules_match_reference_golden(gold):
    from viai_amd import lrschedule as L
    for i, s in enumerate(gold["lr_steps"].tolist()):
        for key, host, orc in (("lr_noam", L.noam_learning_rate_decay(1e-3, s, 2000), P.noam_learning_rate_decay(1e-3, s, 2000)),
                               ("lr_step", L.step_learning_rate_decay(1e-3, s, 0.98, 50000), P.step_learning_rate_decay(1e-3, s, 0.98, 50000)),
                               ("lr_cyclic", L.cyclic_cosine_annealing(1e-3, s, 200000, 5), P.cyclic_cosine_annealing(1e-3, s, 200000, 5))):
            ref = float(gold[key][i])
            assert abs(host - ref) <= 1e-13 * max(abs(ref), 1e-30), (key, s)
            assert abs(float(orc) - ref) <= 1e-13 * max(abs(ref), 1e-30), (key, s)


def test_oracle_ema_and_retrieval_match_reference_golden(gold):
    for decay in (0.9999, 0.9):
        sh = O.cf_uniform("ema.w0", (3, 50), -1, 1).numpy()
        for i in range(5):
            sh = P.ema_update(sh, O.cf_uniform("ema.x%d" % i, (3, 50), -1, 1).numpy(), decay)
        assert np.abs(sh - gold["ema_%g" % decay]).max() <= 1e-7
    clips, caps = _retrieval_inputs()
    m, ranks, top1 = P.l2_retrieval(clips, caps)
    assert np.array_equal(ranks, gold["ret_ranks"]) and np.array_equal(top1, gold["ret_top1"])
    assert np.allclose(m, gold["ret_metrics"])


def _retrieval_inputs():
    W = O.cf_uniform("ret.W", (4, 256), -1, 1).numpy()
    z = O.cf_uniform("ret.z", (96, 4), -1, 1).numpy()
    clips = (z @ W).astype(np.float32)
    caps = ((z + 0.5 * O.cf_uniform("ret.e", (96, 4), -1, 1).numpy()) @ W).astype(np.float32)
    return clips, caps


def test_oracle_batch_assembly_index_arithmetic():
    # slice_clips: every element equals the documented source index (audio_loader.py:471-475)
    T_total, D, hop, N = 120, 5, 7, 6
    c = np.arange(T_total * D, dtype=np.float32).reshape(T_total, D)
    x = np.arange(T_total * hop, dtype=np.float32)
    cb, xb = P.slice_clips(c, x, [0, 10, 24], N, hop)
    assert cb.shape == (3, D, 4 * N) and xb.shape == (3, 1, 4 * N * hop)
    assert cb[1, 2, 3] == c[3 + 40 + 3, 2] and xb[1, 0, 11] == x[(3 + 40) * hop + 11]
    assert cb[2, 0, 4 * N - 1] == 0.0 and cb[2, 0, 20] == c[3 + 96 + 20, 0]      # past the end -> zero padding
    # frames_prep: flip then crop rows by crop_x and columns by crop_y
    f = (np.arange(2 * 9 * 9 * 3) % 251).astype(np.uint8).reshape(2, 9, 9, 3)
    out = P.frames_prep(f, 5, 2, 1, True)
    assert out.shape == (2, 3, 5, 5)
    assert out[1, 2, 3, 4] == np.float32((float(f[1, 2 + 3, 9 - 1 - (1 + 4), 2]) - 127.) / 128.)
    out = P.frames_prep(f, 5, 2, 1, False)
    assert out[0, 1, 0, 0] == np.float32((float(f[0, 2, 1, 1]) - 127.) / 128.)
    # inverse mel: S = 1 -> 0 dB -> amplitude 1; S = 0 -> min_level_db
    assert np.allclose(P.inv_mel_amplitude(np.array([1.0, 0.0, 0.5, 2.0, -1.0]), -100.0), [1.0, 1e-5, 10 ** -2.5, 1.0, 1e-5])


def test_checkpoint_helpers_roundtrip_cpu(tmp_path):
    """WaveNet-style checkpoint dict (utils/model_util.py:122-148) and the tolerant copy (utils/util.py:124-144)."""
    from viai_amd import util as U
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    class Ema:
        shadow = {"0.weight": torch.full((3, 4), 7.0)}
    paths = U.save_checkpoint(m, opt, 12, 3, str(tmp_path), 1, name="wn", ema=Ema())
    assert os.path.basename(paths[0]) == "wn_checkpoint_step000000012.pth.tar"
    assert os.path.basename(paths[1]) == "checkpoint_step000000012_ema.pth"
    ck = torch.load(paths[0], weights_only=False)
    assert set(ck) == {"model", "optimizer", "global_step", "global_epoch", "global_test_step"}
    assert torch.load(paths[1], weights_only=False)["model"]["0.weight"].eq(7.0).all()
    m2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(5, 2))          # second layer mismatches
    prefixed = {"module." + k: v for k, v in ck["model"].items()}                    # DataParallel prefix
    U.copy_state_dict(prefixed, m2, strip="module.")
    assert torch.equal(m2[0].weight, m[0].weight) and torch.equal(m2[0].bias, m[0].bias)
    assert tuple(m2[1].weight.shape) == (2, 5)                                       # shape mismatch: skipped, not an error
    step, epoch, tstep = U.load_checkpoint(paths[0], m2, strip=None)
    assert (step, epoch, tstep) == (12, 1, 3)


# ------------------------------------------------------------------------------------------------ GPU (HIP kernels)
@pytest.mark.gpu
def test_ema_kernel_matches_reference_golden(gold):
    from viai_amd.losses import ExponentialMovingAverage
    for decay in (0.9999, 0.9):
        ema = ExponentialMovingAverage(decay)
        ema.register("w", O.cf_uniform("ema.w0", (3, 50), -1, 1).cuda())
        for i in range(5):
            ema.update("w", O.cf_uniform("ema.x%d" % i, (3, 50), -1, 1).cuda())
        assert np.abs(ema.shadow["w"].cpu().numpy() - gold["ema_%g" % decay]).max() <= 1e-7


@pytest.mark.gpu
def test_l2_retrieval_kernel_matches_reference_golden(gold):
    from viai_amd import util as U
    clips, caps = _retrieval_inputs()
    metrics, (ranks, top1) = U.L2retrieval(torch.from_numpy(clips).cuda(), torch.from_numpy(caps).cuda(), return_ranks=True)
    assert np.array_equal(ranks, gold["ret_ranks"]) and np.array_equal(top1, gold["ret_top1"])       # bit-exact index work
    assert np.allclose(metrics, gold["ret_metrics"])
    r, t, d = U.l2_ranks(torch.from_numpy(clips).cuda(), torch.from_numpy(caps).cuda(), return_dist=True)
    ref = np.sqrt(((caps[:, None, :].astype(np.float64) - clips[None].astype(np.float64)) ** 2).sum(-1))
    assert np.abs(d.cpu().numpy() - ref).max() <= 1e-5 * ref.max()
    # larger, random: every caption's own clip perturbed -> compare with the oracle
    g = torch.Generator().manual_seed(3)
    a = torch.randn(700, 64, generator=g)
    b = a + 0.9 * torch.randn(700, 64, generator=g)
    m2, r2, t2 = P.l2_retrieval(a.numpy(), b.numpy())
    r, t = U.l2_ranks(a.cuda(), b.cuda())
    assert np.array_equal(r.cpu().numpy(), r2) and np.array_equal(t.cpu().numpy(), t2)


@pytest.mark.gpu
@pytest.mark.parametrize("flip", [False, True])
def test_frames_prep_kernel_matches_oracle(flip):
    from viai_amd import batch
    g = torch.Generator().manual_seed(1)
    for C, S, size, cx, cy in ((3, 40, 32, 5, 8), (2, 40, 32, 0, 0), (3, 256, 224, 31, 7)):
        f = torch.randint(0, 256, (2, 3, S, S, C), generator=g, dtype=torch.uint8)
        want = P.frames_prep(f.numpy(), size, cx, cy, flip)
        got = batch.frames_prep(f.cuda(), size, cx, cy, flip, nchw=True)
        assert tuple(got.shape) == want.shape
        assert np.array_equal(got.cpu().numpy(), want)                 # (px - 127) / 128 is exact in fp32
        nhwc4 = batch.frames_prep(f.cuda(), size, cx, cy, flip)
        assert tuple(nhwc4.shape) == (6, size, size, 4) and float(nhwc4[..., C:].abs().max()) == 0.0
        assert np.array_equal(nhwc4[..., :C].permute(0, 3, 1, 2).cpu().numpy().reshape(want.shape), want)


@pytest.mark.gpu
def test_slice_clips_kernel_matches_oracle():
    from viai_amd import batch
    g = torch.Generator().manual_seed(2)
    T_total, D, hop, N = 300, 80, 256, 13
    c = torch.rand(T_total, D, generator=g)
    x = torch.rand(T_total * hop, generator=g) * 2 - 1
    starts = [0, 17, 40, 62]                          # the last clip runs past the end -> zero padding
    want_c, want_x = P.slice_clips(c.numpy(), x.numpy(), starts, N, hop)
    got_c, got_x = batch.slice_clips(c.cuda(), x.cuda(), starts, N, hop)
    assert np.array_equal(got_c.cpu().numpy(), want_c) and np.array_equal(got_x.cpu().numpy(), want_x)


@pytest.mark.gpu
def test_inv_mel_amplitude_kernel_matches_oracle():
    from viai_amd import audio
    S = torch.cat([O.cf_uniform("inv.S", (2, 80, 50), -0.2, 1.2).flatten(), torch.tensor([0.0, 1.0, 0.5])])
    got = audio.inv_mel_amplitude(S.cuda(), -100.0).cpu().numpy().astype(np.float64)
    want = P.inv_mel_amplitude(S.numpy(), -100.0)
    assert np.abs(got / want - 1).max() <= 1e-5          # fp32 exp10 vs float64 power


@pytest.mark.gpu
def test_lr_schedule_reaches_the_device_adam_state():
    from viai_amd import lrschedule as L
    from viai_amd.model import FlatArena, FusedAdam
    p = torch.nn.Parameter(torch.ones(8, device="cuda"))
    arena = FlatArena([("p", p)])
    opt = FusedAdam(arena, 1e-3, (0.9, 0.999), 1e-8)
    lr = L.apply_schedule(opt, L.noam_learning_rate_decay, 1e-3, 100, warmup_steps=2000)
    assert abs(float(opt.state[1]) - lr) <= 1e-18 and abs(lr - 1e-3 * 2000 ** 0.5 * 101 * 2000 ** -1.5) < 1e-15


# ------------------------------------------------------------------------------------------------------------------
# checkpoint interchange against what the REFERENCE writes (utils/util.py:146-162)
# ------------------------------------------------------------------------------------------------------------------

def _describe(v):
    """same structure walk as tools/make_goldens.py::checkpoint_structure_golden"""
    from collections import OrderedDict
    if torch.is_tensor(v):
        return {"tensor": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
    if isinstance(v, dict):
        return {"dict": [[k if isinstance(k, str) else int(k), _describe(x)] for k, x in v.items()], "ordered": isinstance(v, OrderedDict)}
    if isinstance(v, (list, tuple)):
        return {type(v).__name__: [_describe(x) for x in v]}
    return {type(v).__name__: v}


def _ref_structure(golden_dir):
    import json
    with open(os.path.join(golden_dir, "checkpoint_structure.json")) as f:
        return json.load(f)


def test_optimizer_state_dict_has_the_reference_layout_cpu(golden_dir):
    """FusedAdam.state_dict() over the flat arena == the torch.optim.Adam state_dict inside a checkpoint written by the reference's
    save_inpainting_checkpoint: same param_group keys and values, same per-parameter entries (none for the never-reached
    convblock1.*), same tensor shapes / dtypes, `step` as a 0-d float tensor.  (CPU: only the host-side container logic.)"""
    from viai_amd import networks as N
    from viai_amd.model import FlatArena, FusedAdam
    ref = dict(_ref_structure(golden_dir)["checkpoint"]["dict"])
    E, G, D = N.MelEncoder(), N.MelDecoder(), N.MelDiscriminator()
    g_named = [("E." + n, p) for n, p in E.named_parameters()] + [("G." + n, p) for n, p in G.named_parameters()]
    arena = FlatArena(g_named)
    dead = [i for i, n in enumerate(arena.names) if n.startswith("G.convblock1.")]
    opt = FusedAdam(arena, 2e-4, (0.5, 0.999), 1e-8, dead)
    assert opt.state_dict()["state"] == {}                                   # before the first step torch has no state either
    opt.state[0] = 1.0
    assert _describe(opt.state_dict()) == ref["optimizer_G"]
    optD = FusedAdam(FlatArena([("D." + n, p) for n, p in D.named_parameters()]), 2e-4, (0.5, 0.999), 1e-8)
    optD.state[0] = 1.0
    assert _describe(optD.state_dict()) == ref["optimizer_D"]
    for mod, key in ((E, "Mel_Encoder"), (G, "Mel_Decoder"), (D, "netD")):
        assert _describe(mod.state_dict()) == ref[key], key
    # and the reverse direction: a torch.optim.Adam accepts it
    tE, tG = N.MelEncoder(), N.MelDecoder()
    topt = torch.optim.Adam(list(tE.parameters()) + list(tG.parameters()), lr=1e-3)
    topt.load_state_dict(opt.state_dict())
    assert topt.param_groups[0]["lr"] == 2e-4 and len(topt.state) == len(opt.state_dict()["state"])


@pytest.mark.gpu
def test_checkpoint_file_has_the_reference_structure_and_reference_style_files_load(tmp_path, golden_dir):
    """AudioModel.save_inpainting_checkpoint after one step writes the SAME structure (top-level key order, file name, module
    state_dict keys / shapes / dtypes, optimizer layout, python scalar types) as the reference's save_inpainting_checkpoint did
    for its own modules (tests/golden/checkpoint_structure.json); a checkpoint assembled the reference's way (torch modules'
    state_dicts + torch.optim.Adam.state_dict()) loads and continues bit-identically to the model it was taken from."""
    from viai_amd.model import AudioModel, StepConfig
    ref = _ref_structure(golden_dir)
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths, hp.name = 80, 32, "viai_golden"
    s = O.cf_uniform("s.tiny", (2, 1, 80, 32))
    mask = O.make_mask(2, 32, "mask.tiny")
    a = AudioModel(hp, device="cuda")
    a.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
    a.set_inputs(s, mask)
    a.optimize_parameters(0)
    path = a.save_inpainting_checkpoint(7, 3, str(tmp_path), 2)
    assert os.path.basename(path) == ref["file_name"]
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert _describe(ck) == ref["checkpoint"]
    # reference-style file: plain torch containers.  torch.optim.Adam round-trips our optimizer state, then writes it its own way
    import collections
    tparams = [torch.nn.Parameter(p.detach().cpu().clone()) for p in a.arena_G.params]
    topt = torch.optim.Adam(tparams, lr=1e-3)
    topt.load_state_dict(ck["optimizer_G"])
    ck2 = collections.OrderedDict(ck)
    ck2["optimizer_G"] = topt.state_dict()
    p2 = os.path.join(str(tmp_path), "ref_style.pth.tar")
    torch.save(dict(ck2), p2)
    b = AudioModel(hp, device="cuda")
    assert b.load_inpainting_checkpoint(p2) == (7, 2, 3)
    b.set_inputs(s, mask)
    a.optimize_parameters(1)
    b.optimize_parameters(1)
    torch.cuda.synchronize()
    assert torch.equal(a.arena_G.flat, b.arena_G.flat) and torch.equal(a.arena_D.flat, b.arena_D.flat)


# ------------------------------------------------------------------------------------------------------------------
# batch assembly and the audio helpers against what the REFERENCE's loader / audio functions returned (loader.npz)
# ------------------------------------------------------------------------------------------------------------------

def _loader_frame(kind, idx, gray, R=256):
    """the synthetic frame tools/make_goldens.py's cv2 harness served for `<kind>/<idx>.jpg`"""
    t = O.cf_uniform("ld.%s.%d" % (kind, idx), (R, R) if gray else (R, R, 3), 0, 256).numpy()
    return np.clip(np.floor(t), 0, 255).astype(np.uint8)


def _loader_clip_frames(lg, ln, i):
    item = int(lg["starts"][ln]) + i + 1
    rgb = _loader_frame("image_crop", item, False)[..., ::-1].copy()                    # cv2.cvtColor(BGR -> RGB)
    fl = np.stack((_loader_frame("flow_x_crop", item, True), _loader_frame("flow_y_crop", item, True)), -1)
    return rgb, fl


def _loader_utterances(lg, use=52, hop=256):
    mels, wavs = [], []
    for u in range(2):
        T_mel = 3 + 4 * (int(lg["starts"].max()) + use) + 5 + 11 * u
        mels.append(O.cf_uniform("ld.c%d" % u, (T_mel, 80), 0, 1))
        wavs.append(O.cf_uniform("ld.x%d" % u, (T_mel * hop,), -1, 1))
    return mels, wavs


def test_oracle_batch_assembly_and_audio_helpers_match_reference_loader_golden(golden_dir):
    from oracle import audio_oracle as AO
    lg = np.load(os.path.join(golden_dir, "loader.npz"))
    cx, cy, flip = [int(v) for v in lg["crop_flip"]]
    assert flip == 1 and (cx, cy) != (0, 0)                       # the fixture exercises flip and a non-trivial crop
    rgb, _ = _loader_clip_frames(lg, 0, 0)
    assert np.abs(P.frames_prep(rgb, 224, cx, cy, flip) - lg["video_first"]).max() < 1e-6
    _, fl = _loader_clip_frames(lg, len(lg["starts"]) - 1, 51)
    assert np.abs(P.frames_prep(fl, 224, cx, cy, flip) - lg["flow_last"]).max() < 1e-6
    mels, wavs = _loader_utterances(lg)
    cs, xs = zip(*[P.slice_clips(m.numpy(), w.numpy(), lg["starts"].tolist(), 52, 256) for m, w in zip(mels, wavs)])
    c_all, x_all = np.concatenate(cs), np.concatenate(xs)
    assert np.array_equal(c_all[1], lg["collate.c_clip1"])
    assert np.allclose(O.digest(torch.from_numpy(c_all), 256), lg["collate.c.dg"], rtol=1e-12)
    assert np.allclose(O.digest(torch.from_numpy(x_all), 256), lg["collate.x.dg"], rtol=1e-12)
    S = O.cf_uniform("ld.S", (80, 64), -0.2, 1.2).numpy()
    assert np.allclose(P.inv_mel_amplitude(S, float(lg["min_level_db"])), lg["inv_mel"], rtol=1e-12)
    amp = 10.0 ** O.cf_uniform("ld.amp", (80, 64), -7, 1).double().numpy()
    mn, ref = float(lg["min_level_db"]), float(lg["ref_level_db"])
    assert np.allclose(AO.normalize(AO.amp_to_db(amp, mn) - ref, mn), lg["amp_to_db_norm"], rtol=1e-12, atol=1e-15)
    for row in lg["lws_table"].tolist():
        length = row[0]
        assert AO.lws_num_frames(length, 1024, 256) == row[1] and tuple(AO.lws_pad_lr(length, 1024, 256)) == tuple(row[2:4])
        assert AO.lws_num_frames(length, 1024, 320) == row[4] and tuple(AO.lws_pad_lr(length, 1024, 320)) == tuple(row[5:7])
    from viai_amd import audio as A                               # host mirror in the product package (plain arithmetic, no GPU)
    for row in lg["lws_table"].tolist():
        assert A.lws_num_frames(row[0], 1024, 256) == row[1] and tuple(A.lws_pad_lr(row[0], 1024, 256)) == tuple(row[2:4])


@pytest.mark.gpu
def test_batch_assembly_kernels_match_reference_loader_golden(golden_dir):
    """viai_frames_prep / viai_slice_clips / viai_mel_denorm_amp against the outputs of the reference's sample_data_new / collate_fn /
    _db_to_amp(_denormalize(.)) on the same synthetic frames, mels and waveforms (tests/golden/loader.npz)."""
    from viai_amd import audio, batch
    lg = np.load(os.path.join(golden_dir, "loader.npz"))
    cx, cy, flip = [int(v) for v in lg["crop_flip"]]
    starts = lg["starts"].tolist()
    vids, flows = [], []
    for ln in range(len(starts)):
        fr = [_loader_clip_frames(lg, ln, i) for i in range(52)]
        vids.append(np.stack([f[0] for f in fr])); flows.append(np.stack([f[1] for f in fr]))
    video = batch.frames_prep(torch.from_numpy(np.stack(vids)).cuda(), 224, cx, cy, bool(flip), nchw=True)
    flow = batch.frames_prep(torch.from_numpy(np.stack(flows)).cuda(), 224, cx, cy, bool(flip), nchw=True)
    assert tuple(video.shape) == (2, 52, 3, 224, 224) and tuple(flow.shape) == (2, 52, 2, 224, 224)
    assert np.array_equal(video[0, 0].cpu().numpy(), lg["video_first"]) and np.array_equal(flow[-1, -1].cpu().numpy(), lg["flow_last"])
    assert np.allclose(O.digest(video, 256), lg["video.dg"], rtol=1e-12) and np.allclose(O.digest(flow, 256), lg["flow.dg"], rtol=1e-12)
    mels, wavs = _loader_utterances(lg)
    cs, xs = zip(*[batch.slice_clips(m.cuda(), w.cuda(), starts, 52, 256) for m, w in zip(mels, wavs)])
    c_all, x_all = torch.cat(cs), torch.cat(xs)
    assert tuple(c_all.shape) == (4, 80, 208) and tuple(x_all.shape) == (4, 1, 208 * 256)
    assert np.array_equal(c_all[1].cpu().numpy(), lg["collate.c_clip1"])
    assert np.allclose(O.digest(c_all, 256), lg["collate.c.dg"], rtol=1e-12) and np.allclose(O.digest(x_all, 256), lg["collate.x.dg"], rtol=1e-12)
    S = O.cf_uniform("ld.S", (80, 64), -0.2, 1.2)
    got = audio.inv_mel_amplitude(S.cuda(), float(lg["min_level_db"])).cpu().numpy().astype(np.float64)
    assert np.abs(got / lg["inv_mel"] - 1).max() <= 2e-5
