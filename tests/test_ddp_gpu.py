"""GPU, world size 2: the PRODUCT train step (real kernels, three streams, the bucketed / overlapped gradient exchange of
model.AudioModel) in two processes that share cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the driver
runs the RCCL path on a multi-GPU node).  Checks the data-parallel contract of SURVEY.md section 8e:

* after the exchange every rank holds the SUM of the per-rank gradient arenas (the 1/N is folded into the Adam kernel), for the
  D step and for the G step -- against two single-process runs on each rank's clips;
* after three full steps (both Adam updates per step, G exchange + Adam(E,G) deferred behind the next step's D(real) branch)
  the replicas' parameters, Adam moments and losses-by-rank behave: parameters bit-identical across ranks, BatchNorm running
  statistics per replica (the reference has no SyncBN).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (2, 80, 32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(norm="bn"):
    from collections import OrderedDict
    from oracle import viai_oracle as O
    from viai_amd.model import AudioModel, StepConfig
    hp = StepConfig()
    hp.cin_channels, hp.max_mel_lengths = SHAPE[1], SHAPE[2]
    if norm == "in":                                     # MelDecoder(hp) honours hp.normlayer (New_Inpainting_Networks.py:49)
        hp.normlayer = torch.nn.InstanceNorm2d
    m = AudioModel(hp, device="cuda:0")
    G = O.decoder_state()
    if norm == "in":                                     # InstanceNorm decoder: no norm parameters / buffers, biased convs (closed form: same on every rank)
        G = OrderedDict((k, G[k] if (k in G and G[k].shape == v.shape) else O.cf_std("ddpgpu.G." + k, tuple(v.shape), 0.05))
                        for k, v in m.Mel_Decoder.state_dict().items())
        assert any(isinstance(x, torch.nn.InstanceNorm2d) for x in m.Mel_Decoder.modules())
        assert m._early_G == [] and m._late_G == [(0, m.arena_G.size)]       # per-sample backward calls: no early buckets
    m.load_states(O.encoder_state(), G, O.disc_state())
    return m


def _clips(rank):
    from viai_amd import synth
    B, F, T = SHAPE
    return synth.mel_batch(B, F, T, "ddpgpu.s", rank).cuda(), synth.time_mask(B, T, "ddpgpu.mask", rank).cuda()


def _worker(rank, world, port, out_dir, norm):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from viai_amd import ddp
    r, _, w = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.cuda.set_device(0)
    m = _model(norm)
    assert m.world == 2 and m._exchanging() and m._overlapped()
    s, mask = _clips(rank)
    m.set_inputs(s, mask)
    m.forward_backward_no_update()                      # exchange runs, Adam does not: arenas hold the all-reduced SUMS
    torch.cuda.synchronize()
    out = {"gD_sum": m.arena_D.grad.cpu().clone(), "gG_sum": m.arena_G.grad.cpu().clone()}
    m = _model(norm)
    m.set_inputs(s, mask)
    for i in range(3):
        m.optimize_parameters(i)
    assert m._g_update_pending                           # the last G update is still riding the exchange stream ...
    m.sync_pending_update()                              # ... until somebody needs it
    torch.cuda.synchronize()
    out.update(pG=m.arena_G.flat.cpu().clone(), pD=m.arena_D.flat.cpu().clone(), mG=m.optimizer_G.exp_avg.cpu().clone(),
               vD=m.optimizer_D.exp_avg_sq.cpu().clone(), rm=m.netD.norm3.running_mean.cpu().clone(),
               losses=m.losses.cpu().clone(), stepG=m.optimizer_G.state.cpu().clone())
    torch.save(out, os.path.join(out_dir, "r%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("norm", ["bn", "in"])
def test_two_process_product_step_exchanges_sums_and_keeps_replicas_identical(tmp_path, norm):
    """norm = "in": the decoder built with nn.InstanceNorm2d, whose layers run the conv kernel once per sample with the same weight --
    a gradient-ready hook armed for ONE backward invocation of the trigger layer would reduce a bucket the other samples are still
    accumulating into; such models exchange whole arenas after the last weight gradient (model._plan_exchange)."""
    import sys
    sys.path.insert(0, ROOT)
    port = _free_port()
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, port, str(tmp_path), norm), nprocs=2, join=True)
    a = torch.load(tmp_path / "r0.pt")
    b = torch.load(tmp_path / "r1.pt")
    # every rank holds the same reduced gradients and ends with the same parameters / moments, bit for bit
    for k in ("gD_sum", "gG_sum", "pG", "pD", "mG", "vD", "stepG"):
        assert torch.equal(a[k], b[k]), k
    assert float(a["stepG"][0]) == 3.0
    assert not torch.equal(a["rm"], b["rm"])             # BatchNorm statistics stay per replica
    assert not torch.equal(a["losses"], b["losses"])     # each rank reports the losses of its own clips
    # the reduced gradient == sum of the two single-rank gradient arenas (same kernels, one process, no group)
    sums = None
    for rank in range(2):
        m = _model(norm)
        assert m.world == 1 and not m._exchanging()
        m.set_inputs(*_clips(rank))
        m.forward_backward_no_update()
        torch.cuda.synchronize()
        g = (m.arena_D.grad.cpu().double(), m.arena_G.grad.cpu().double())
        sums = g if sums is None else (sums[0] + g[0], sums[1] + g[1])
        del m
    for got, want, name in ((a["gD_sum"], sums[0], "D"), (a["gG_sum"], sums[1], "G")):
        err = ((got.double() - want).norm() / want.norm()).item()
        assert err < 1e-6, (name, err)                   # one fp32 addition per element
