O=gpurun_out/r04_r; mkdir -p $O
ok=0; for i in $(seq 1 20); do python -m pytest tests/test_networks_gpu.py::test_gradient_exchange_over_rccl_is_wired_into_the_step -q -x > $O/rccl_$i.log 2>&1 && ok=$((ok+1)); done; echo "rccl one-rank test: $ok / 20 passed" | tee $O/rccl_loop.txt
python -m pytest tests/test_bench_gpu.py tests/test_ddp_gpu.py -q 2>&1 | tail -3
