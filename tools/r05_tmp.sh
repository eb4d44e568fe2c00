timeout 900 python -m pytest tests/test_p16_gpu.py -q -k "presplit_operands" 2>&1 | grep -E "^E  |passed|failed" | head -12
tools/probes/s2_dma_prof.bin 16 64 32 256 512 10 1 | grep -E "per launch|consumer stage  [01] "
tools/probes/s2_dma_prof.bin 16 256 128 64 128 10 2 | grep -E "per launch|consumer stage  [01] "
tools/probes/s2_dma_prof.bin 16 128 64 128 256 10 2 | grep -E "per launch"
