mkdir -p gpurun_out/r05_j
timeout 2500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r05_j/gputest_tail.txt
