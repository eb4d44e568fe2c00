"""CPU, world_size 2 over gloo: the data-parallel exchange (one all-reduce per flat gradient arena, 1/N folded
into the optimizer) reproduces "N replicas + mean-reduced gradients == one replica on the N-times batch"
for a BN-free model (BatchNorm statistics stay per replica by design)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from viai_amd import ddp, synth
    from viai_amd.model import FlatArena
    r, _, w = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # deliberately different init per rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    arena = FlatArena(list(net.named_parameters()))
    ddp.broadcast_arena(arena.flat, src=0)              # -> identical parameters
    x = synth.uniform("ddp.x", (4, 6), -1, 1, ) if False else synth.mel_batch(4, 2, 3, "ddp.x", rank).reshape(4, 6)
    arena.zero_grad()
    net(x).pow(2).mean().backward()
    # the product step exchanges the arena in BUCKETS: in-place sum-all-reduce of contiguous views (model._reduce_range)
    half = arena.size // 2
    ddp.all_reduce_sum_(arena.grad[half:])
    ddp.all_reduce_sum_(arena.grad[:half])
    arena.grad.div_(w)
    torch.save({"flat": arena.flat.clone(), "grad": arena.grad.clone(), "x": x}, os.path.join(out_dir, "r%d.pt" % rank))
    t = ddp.barrier_max_ms(float(rank + 1))
    assert t == float(world)
    # the divergence flag of model.get_loss_items is collective: rank 1's bad maximum (a NaN mapped to +inf) reaches rank 0, so both raise in the same call
    flag = torch.tensor([1.0 + rank, float("inf") if rank == 1 else 3.0])
    ddp.all_reduce_max_(flag)
    assert flag.tolist() == [float(world), float("inf")]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_big_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "r0.pt")
    b = torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["flat"], b["flat"])            # broadcast made the replicas identical
    assert torch.equal(a["grad"], b["grad"])            # every rank holds the same reduced gradient
    assert not torch.equal(a["x"], b["x"])              # rank-keyed synthetic data
    # single process on the concatenated batch
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from viai_amd.model import FlatArena
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    arena = FlatArena(list(net.named_parameters()))
    arena.flat.copy_(a["flat"])
    arena.zero_grad()
    net(torch.cat((a["x"], b["x"]))).pow(2).mean().backward()
    assert torch.allclose(arena.grad, a["grad"], rtol=1e-5, atol=1e-7)
