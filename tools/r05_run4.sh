mkdir -p gpurun_out/r05_d
timeout 900 python -m pytest tests/test_p16_gpu.py -x -q -k "presplit_operands" 2>&1 | tail -5 | tee gpurun_out/r05_d/p16_tests.txt
for f in tools/probes/s2_p*.bin; do echo == $f; $f 16 256 128 64 128 20 | grep -E "per launch|consumer stage  [1234567]|loader   stage  [1234567]|realtime"; done
tools/probes/s2_dma_bench.bin 16 256 128 64 128 50
tools/probes/s2_dma_bench.bin 16 128 64 128 256 50
tools/probes/s2_dma_bench.bin 16 64 128 64 128 50
