#!/bin/bash
# Effective shader clock of one layer's kernels: GRBM_GUI_ACTIVE / wall time per dispatch (MI355X_MICROARCH.md, "DVFS give-back"),
# plus the kernel-trace-only durations of the same command for comparison.    tools/pmc_clock.sh <out.txt> [profile_layer.py arguments]
# PMC_SCRIPT=tools/probes/stft_time.py tools/pmc_clock.sh <out.txt> 1024 256    -- another script instead of profile_layer.py
out=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/pmc_clock_$$
rm -rf "$d"; mkdir -p "$d"
(cd /tmp && TMPDIR=/tmp rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$d/pmc" -o out -- python "$root/${PMC_SCRIPT:-tools/profile_layer.py}" "$@" > "$d/log1.txt" 2>&1) || tail -5 "$d/log1.txt"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d "$d/plain" -o out -- python "$root/${PMC_SCRIPT:-tools/profile_layer.py}" "$@" > "$d/log2.txt" 2>&1) || tail -5 "$d/log2.txt"
python - "$d" > "$out" <<'PY'
import collections, csv, glob, os, sys
d = sys.argv[1]
def durations(sub):
    r = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            r[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return r
pm, pl = durations("pmc"), durations("plain")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        cnt[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("%-100s %9s %9s %12s %8s %12s" % ("kernel", "us(pmc)", "us(plain)", "GUI_ACTIVE", "GHz", "WAVE_CYC x4"))
for k in sorted(pm, key=lambda k: -sum(pm[k])):
    if "at::native" in k or k not in cnt: continue
    us = sorted(pm[k])[len(pm[k]) // 2]
    up = sorted(pl[k])[len(pl[k]) // 2] if k in pl else float("nan")
    ga = cnt[k].get("GRBM_GUI_ACTIVE", [0]); ga = sum(ga) / len(ga)
    wc = cnt[k].get("SQ_WAVE_CYCLES", [0]); wc = 4 * sum(wc) / len(wc)
    print("%-100s %9.1f %9.1f %12.0f %8.3f %12.0f" % (k[:100], us, up, ga, ga / us / 1e3, wc))
PY
cat "$out"
