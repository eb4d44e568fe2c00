#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 kernel trace (bench.py --graph: one stream, so every duration is the kernel alone), in launch order.
    python tools/trace_step.py trace.csv [min_us]"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Grid_Size_X', 0) or 0)))
rows.sort()
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
ends = adam[1::2]
a, b = ends[4] + 1, ends[5] + 1
step = rows[a:b]
print(len(step), 'kernels; wall %.1f us; kernel sum %.1f us' % ((step[-1][1] - step[0][0]) / 1e3, sum(r[1] - r[0] for r in step) / 1e3))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)[:64]
for s, e, n, g in step:
    if (e - s) / 1e3 >= mn:
        print('%8.1f  %7.1f us  grid %8d  %s' % ((s - step[0][0]) / 1e3, (e - s) / 1e3, g, short(n)))
