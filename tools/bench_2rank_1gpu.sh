#!/bin/bash
# bench.py's N = 2 path on ONE GPU: two processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one device).
# Exercises the launch contract, the bucketed exchange, comm_ms_exposed and the rank-0 roofline pass; the numbers are not a measurement.
cd "$(dirname "$0")/.."
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29611} WORLD_SIZE=2 LOCAL_RANK=0 VIAI_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=1 python bench.py --gpus 2 --steps 8 --warmup 2 "$@" > /tmp/bench_rank1.log 2>&1 &
RANK=0 python bench.py --gpus 2 --steps 8 --warmup 2 "$@"
wait
