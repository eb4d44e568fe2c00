# weight-gradient grid sizes against the step time, one box: bash tools/sweep_wgrad_blocks.sh  (defaults: 192 / 192 / 128 blocks for wide / stride-2 / narrow)
O=gpurun_out/r04_j; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_$tag.log 2>&1; python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith(chr(123)):
        d=json.loads(l); print(sys.argv[2], d['ms_per_step'])
" $O/b_$tag.log $tag; }
run base A=1
run w_128 VIAI_WGRAD_PATCH_BLOCKS=128
run w_160 VIAI_WGRAD_PATCH_BLOCKS=160
run w_256 VIAI_WGRAD_PATCH_BLOCKS=256
run base2 A=1
run s2_128 VIAI_WGRAD_PATCH_BLOCKS_S2=128
run s2_256 VIAI_WGRAD_PATCH_BLOCKS_S2=256
run n_96 VIAI_WGRAD_PATCH_BLOCKS_NARROW=96
run n_192 VIAI_WGRAD_PATCH_BLOCKS_NARROW=192
run base3 A=1
run all128 VIAI_WGRAD_PATCH_BLOCKS=128 VIAI_WGRAD_PATCH_BLOCKS_S2=128 VIAI_WGRAD_PATCH_BLOCKS_NARROW=96
run all256 VIAI_WGRAD_PATCH_BLOCKS=256 VIAI_WGRAD_PATCH_BLOCKS_S2=256 VIAI_WGRAD_PATCH_BLOCKS_NARROW=256
run base4 A=1
