#!/usr/bin/env python3
"""GPU occupancy of a rocprofv3 kernel trace: wall span, time with >= 1 kernel resident, idle gaps, mean concurrency.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 20 --warmup 5
    python tools/trace_busy.py out [skip_fraction]

Only the last (1 - skip_fraction) of the trace is analysed (default 0.5: the timed steps, not the warm-up).
With --steps MARKER A B the trace is cut into steps at every start of a kernel whose name contains MARKER (one that runs once
per step, e.g. mask_mul_kernel) and steps A..B-1 are analysed."""
import csv
import glob
import sys

src = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 0.5
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
ev.sort()
if "--steps" in sys.argv:
    i = sys.argv.index("--steps")
    marker, A, B = sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3])
    marks = [e[0] for e in ev if marker in e[2]]
    print("marker occurrences:", len(marks))
    ev = [e for e in ev if marks[A] <= e[0] < marks[B]]
    print("per-step span (ms):", " ".join("%.2f" % ((marks[k + 1] - marks[k]) / 1e6) for k in range(A, B)))
else:
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    cut = t0 + (t1 - t0) * skip
    ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0
tot = sum(e[1] - e[0] for e in ev)
cs, ce = ev[0][0], ev[0][1]
gaps = []
for s, e, n, q in ev[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, n))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = t1 - t0
print("kernels %d  span %.2f ms  busy %.2f ms (%.1f%%)  idle %.2f ms  kernel-time %.2f ms  mean concurrency %.2f"
      % (len(ev), span / 1e6, busy / 1e6, 100 * busy / span, (span - busy) / 1e6, tot / 1e6, tot / busy))
gaps.sort(reverse=True)
import collections
by = collections.Counter()
for g, n in gaps:
    by[n[:70]] += g
print("idle time by the kernel that ended the gap:")
for n, g in by.most_common(12):
    print("  %8.3f ms  %s" % (g / 1e6, n))
