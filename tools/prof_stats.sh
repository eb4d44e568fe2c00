#!/bin/bash
# rocprofv3 kernel stats of a bench run: tools/prof_stats.sh <tag> [bench args]; env passes through.  Writes gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/prof_$tag
rm -rf "$out"; mkdir -p "$out" "$root/gpurun_out"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python "$root/bench.py" "$@" > "$out/log.txt" 2>&1)
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then tail -30 "$out/log.txt"; exit 1; fi
cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
grep '^{"metric"' "$out/log.txt" > "$root/gpurun_out/${tag}_profiled_bench_line.json"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    print("%-100s %6s %9.1f us %5.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
