#!/bin/bash
# Kernel-trace statistics of tools/profile_layer.py (one conv layer, forward + backward) under rocprofv3.
#   tools/layer_stats.sh <tag> [profile_layer.py arguments]      env passes through (VIAI_* switches)
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/layer_stats_$tag
rm -rf "$out"; mkdir -p "$out"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python "$root/tools/profile_layer.py" "$@" > "$out/log.txt" 2>&1)
python - "$out" "$tag" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs:
    print(open(sys.argv[1] + "/log.txt").read()[-2000:]); sys.exit(1)
print("==", sys.argv[2])
for r in list(csv.DictReader(open(fs[0])))[:10]:
    print("%-90s %4s %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
