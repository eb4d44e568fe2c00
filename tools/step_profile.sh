#!/bin/bash
# One round's evidence of the default step in one go: three-stream kernel stats + queue report, single-stream timeline.   tools/step_profile.sh <outdir> <tag>
out=$1; tag=$2; root=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$out"
d=/tmp/step_profile_$$; rm -rf "$d"; mkdir -p "$d"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python "$root/bench.py" --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > "$d/log.txt" 2>&1)
f=$(find "$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats_bench.csv"
python "$root/tools/step_queues.py" "$d" > "$out/${tag}_queues.txt" 2>&1
"$root/tools/trace_graph.sh" "$out/${tag}_graph_step_trace.txt"
head -30 "$out/${tag}_queues.txt"
