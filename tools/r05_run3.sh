mkdir -p gpurun_out/r05_c
timeout 900 python -m pytest tests/test_p16_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r05_c/p16_tests.txt
for v in 0 1; do
for s in "16 256 128 64 128" "16 128 64 128 256" "16 64 128 64 128"; do VIAI_HALO_DMA=$v tools/layer_stats.sh s2_$v --shape $s --stride 2 --bn --p16 2>&1 | head -5 | tee -a gpurun_out/r05_c/s2.txt; done
done
