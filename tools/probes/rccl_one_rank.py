#!/usr/bin/env python3
"""How often does a one-rank RCCL process group abort in set-up / teardown on this stack, and which recipe makes it boring?

    python tools/probes/rccl_one_rank.py [runs per variant, default 20]

Every run is a fresh interpreter (the failure is a process abort).  Variants:
  lazy        init_process_group("nccl") without device_id (communicator created at the first collective), destroy at the end
  eager       init_process_group("nccl", device_id=cuda:0) (communicator created inside init)
  eager_sync  eager + torch.cuda.synchronize() + barrier before destroy_process_group()
  model       eager_sync around three AudioModel steps with the forced exchange (what tests/test_networks_gpu.py runs)
Prints per variant: ok / failed counts, exit codes and the stderr tail of the first failure."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
variant = %(variant)r
import torch, torch.distributed as dist
kw = {}
if variant != "lazy":
    torch.cuda.set_device(0)
    kw["device_id"] = torch.device("cuda:0")
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, **kw)
t = torch.ones(1 << 20, device="cuda")
for _ in range(3):
    dist.all_reduce(t)
if variant == "model":
    from oracle import viai_oracle as O
    from viai_amd.model import AudioModel, StepConfig
    hp = StepConfig(); hp.cin_channels, hp.max_mel_lengths = 80, 32
    m = AudioModel(hp, device="cuda")
    m.load_states(O.encoder_state(), O.decoder_state(), O.disc_state())
    m._force_allreduce = True
    m.set_inputs(O.cf_uniform("rc.s", (2, 1, 80, 32), 0, 1).cuda(), O.make_mask(2, 32, "rc.mask").cuda())
    for i in range(3):
        m.optimize_parameters(i)
    m.sync_pending_update()
    m.close()
if variant in ("eager_sync", "model"):
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
dist.destroy_process_group()
print("CHILD_OK", float(t[0]))
"""


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lazy", "eager", "eager_sync", "model"]
    for variant in variants:
        ok, bad, first = 0, [], None
        for _ in range(runs):
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), NCCL_DEBUG="WARN")
            r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "variant": variant, "port": port}], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            if r.returncode == 0 and "CHILD_OK" in r.stdout:
                ok += 1
            else:
                bad.append(r.returncode)
                if first is None:
                    first = (r.stdout[-500:], r.stderr[-2500:])
        print("variant %-10s ok %d / %d   exit codes of failures: %s" % (variant, ok, runs, bad))
        if first is not None:
            print("  first failure stdout tail:", first[0])
            print("  first failure stderr tail:", first[1])
        sys.stdout.flush()


if __name__ == "__main__":
    main()
