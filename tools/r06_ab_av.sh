# same-box alternating pairs of the vision-infused step: BatchNorm partials of the linear-tile conv kernel per 128 pixels (VIAI_LIN_STAT_MERGE=0) / merged per block
run() { python bench.py --config av --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2 3 4; do a=$(VIAI_LIN_STAT_MERGE=0 run); b=$(run); echo "pair $r: per-128-pixel partials $a   merged per block $b"; done
