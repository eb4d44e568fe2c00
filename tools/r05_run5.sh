mkdir -p gpurun_out/r05_g
for s in "16 256 128 64 128" "16 128 64 128 256" "16 64 128 64 128"; do tools/layer_stats.sh s2 --shape $s --stride 2 --bn --p16 2>&1 | head -5 | tee -a gpurun_out/r05_g/s2.txt; done
timeout 900 python -m pytest tests/test_p16_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/r05_g/tests.txt
