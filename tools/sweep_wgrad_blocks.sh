O=gpurun_out/r04_j; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/b_$tag.log 2>&1; python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith(chr(123)):
        d=json.loads(l); print(sys.argv[2], d['ms_per_step'])
" $O/b_$tag.log $tag; }
run base A=1
run s2_256 VIAI_WGRAD_PATCH_BLOCKS_S2=256
run s2_128 VIAI_WGRAD_PATCH_BLOCKS_S2=128
run n_256 VIAI_WGRAD_PATCH_BLOCKS_NARROW=256
run n_192 VIAI_WGRAD_PATCH_BLOCKS_NARROW=192
run w_256 VIAI_WGRAD_PATCH_BLOCKS=256
run w_128 VIAI_WGRAD_PATCH_BLOCKS=128
run base2 A=1
run all256 VIAI_WGRAD_PATCH_BLOCKS=256 VIAI_WGRAD_PATCH_BLOCKS_S2=256 VIAI_WGRAD_PATCH_BLOCKS_NARROW=256
run plan A=1
