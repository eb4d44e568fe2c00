mkdir -p gpurun_out/r05_j
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_networks_gpu.py tests/test_p16_gpu.py -x -q 2>&1 | tail -6 | tee gpurun_out/r05_j/tests.txt
tools/prof_stats.sh r05_j/step --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>&1 | grep -E 'cin1|stats_cov' | head
