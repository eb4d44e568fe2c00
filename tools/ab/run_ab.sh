#!/bin/bash
# same-box alternation of library builds: tools/ab/run_ab.sh "A C" rounds  (copies tools/ab/lib<X>.so over the in-tree library)
L=vision-infused-audio-inpainter-viai_amd/libviai_hip.so
cp $L /tmp/lib_orig.so
for i in $(seq 1 ${2:-3}); do for v in $1; do
  cp tools/ab/lib$v.so $L
  r=$(python bench.py --config wavenet --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  echo "$v $r"
done; done
cp /tmp/lib_orig.so $L
