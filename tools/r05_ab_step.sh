#!/bin/bash
# same-box alternating pairs of the default bench with one switch off / on:  tools/r05_ab_step.sh <outfile> <ENVVAR> [pairs]
out=$1; var=$2; n=${3:-4}
for i in $(seq 1 $n); do for v in 0 1; do
  l=$(env $var=$v python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | grep '^{"metric"' | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$var=$v $l" | tee -a "$out"
done; done
