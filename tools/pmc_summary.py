#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, as written by the recipe in DESIGN.md section 3.3)
into one JSON: per kernel, the average of every counter over its dispatches.

    python tools/pmc_summary.py gpurun_out/r01d_pmc profiles/r01_d_pmc_dconv3.json

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads, so `hbm_bytes` below is 2 * FETCH_SIZE + WRITE_SIZE (x 1024); both are L2
memory-side request counters and therefore INCLUDE Infinity-Cache hits (they bound HBM traffic from above)."""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if "at::native" in k:
        continue
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["dispatches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    out[k] = d
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "kernels:", len(out))
