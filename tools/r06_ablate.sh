#!/bin/bash
# alternating same-box runs: tools/r06_ablate.sh <outfile> [pairs]
out=$1; n=${2:-3}
for i in $(seq 1 $n); do for w in none wgrad; do
  l=$(python tools/ablate_step.py $w --no-cpu-baseline --no-roofline --no-extra --steps 30 --warmup 8 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$w $l" | tee -a "$out"
done; done
