#!/bin/bash
# single-stream timeline of one steady-state step: bench.py --graph under rocprofv3 --kernel-trace, summarised by tools/trace_step.py
#   tools/trace_graph.sh <out.txt> [extra bench args]        env passes through (VIAI_* switches)
outf=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/trace_graph_$$
rm -rf "$d"; mkdir -p "$d"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python "$root/bench.py" --graph --steps 12 --warmup 3 --no-cpu-baseline --no-roofline "$@" > "$d/log.txt" 2>&1)
f=$(find "$d" -name "*kernel_trace.csv" | head -1)
if [ -z "$f" ]; then tail -30 "$d/log.txt"; exit 1; fi
python "$root/tools/trace_step.py" "$f" > "$outf"
head -1 "$outf"
