#!/bin/bash
# Time-weighted MFMA-busy fraction of the conv kernels of the default train step: ONE rocprofv3 --pmc pass of bench.py (single-stream --graph replay, so
# every dispatch has the chip to itself) with SQ_VALU_MFMA_BUSY_CYCLES (= MFMA instructions x their pipe cycles, summed over the SIMDs) and GRBM_GUI_ACTIVE
# (shader-engine busy cycles, summed over the 8 XCDs).  mfma_busy(kernel) = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
#   tools/pmc_step.sh <out.json>
out=$1
root=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/pmc_step_$$; rm -rf "$d"; mkdir -p "$d"
(cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$d" -o out -- python "$root/bench.py" --graph --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > "$d/log.txt" 2>&1) || tail -5 "$d/log.txt"
python - "$d" "$out" <<'PY'
import collections, csv, glob, json, os, re, sys
d, out = sys.argv[1], sys.argv[2]
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[r["Kernel_Name"]] += 1
def short(k): return re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
rows = {}
tot_busy = tot_cyc = conv_busy = conv_cyc = 0.0
for k, c in cnt.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0          # SIMD-cycles the dispatches of this kernel had
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if cyc <= 0: continue
    tot_busy += busy; tot_cyc += cyc
    if busy > 0: conv_busy += busy; conv_cyc += cyc
    rows[short(k)] = {"dispatches": n[k], "mfma_busy": round(busy / cyc, 4), "share_of_shader_cycles": 0.0, "_cyc": cyc}
for v in rows.values(): v["share_of_shader_cycles"] = round(v.pop("_cyc") / tot_cyc, 4)
res = {"mfma_busy_conv_kernels_time_weighted": round(conv_busy / conv_cyc, 4), "mfma_busy_whole_step": round(tot_busy / tot_cyc, 4),
       "note": "cycle-weighted over every dispatch of the profiled steps; conv kernels = the kernels that issue MFMAs; cycles are shader cycles (the clock is NOT constant: MFMA-heavy kernels run at 1.3 - 1.7 GHz under the power limit)",
       "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["share_of_shader_cycles"])[:40])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("mfma_busy_conv_kernels_time_weighted", "mfma_busy_whole_step")}))
for k, v in list(res["kernels"].items())[:12]: print("%-72s %s" % (k, v))
PY
