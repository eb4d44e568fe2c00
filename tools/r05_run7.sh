timeout 900 python -m pytest tests/test_p16_gpu.py -q -k "discriminator_and_decoder or presplit_operands" 2>&1 | grep -E "^E  |passed|failed" | head -12
VIAI_HALO_DMA=0 timeout 900 python -m pytest -x -q -p no:cacheprovider "tests/test_networks_gpu.py::test_step_no_update_matches_oracle_and_golden" "tests/test_p16_gpu.py::test_forward_and_data_gradient_on_presplit_operands_are_bitwise_the_fp32_input_kernels" 2>&1 | grep -E "^E  |passed|failed" | head -8
tools/probes/s2_dma_prof.bin 16 64 32 256 512 10 1 | grep -E "per launch|consumer stage (1[4-7])"
tools/probes/s2_dma_prof.bin 16 256 128 64 128 10 2 | grep -E "per launch|consumer stage  ([2-5])"
tools/layer_stats.sh dconv3 --shape 16 64 32 256 512 --bn --p16 2>&1 | head -3
VIAI_HALO_DMA=0 tools/layer_stats.sh dconv3_off --shape 16 64 32 256 512 --bn --p16 2>&1 | head -3
tools/layer_stats.sh s2 --shape 16 256 128 64 128 --stride 2 --bn --p16 2>&1 | head -4
