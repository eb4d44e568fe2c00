set -x
mkdir -p gpurun_out/r05_a
timeout 900 python -m pytest tests/test_p16_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r05_a/p16_tests.txt
cat gpurun_out/r05_a/p16_tests.txt
tools/r05_ab.sh gpurun_out/r05_a VIAI_HALO_DMA
for s in "16 256 128 64 128" "16 128 64 128 256"; do tools/layer_stats.sh s2 --shape $s --stride 2 --bn --p16 2>&1 | tee -a gpurun_out/r05_a/s2.txt; done
