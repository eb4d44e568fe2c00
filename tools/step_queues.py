#!/usr/bin/env python3
"""Per hardware queue: kernel time by class and the idle gaps of a steady-state train step, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline
    python tools/step_queues.py out [first_step] [n_steps]

Steps are cut at the second adam_kernel of each step.  The queue with the most kernel time is the main chain."""
import collections
import csv
import glob
import re
import sys

src = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[3]]
ends = adam[1::2]
a, b = ends[first] + 1, ends[first + nst] + 1
step = rows[a:b]
t0, t1 = min(r[0] for r in step), max(r[1] for r in step)
print("%d steps: wall %.3f ms/step, %d kernels/step, kernel-time sum %.3f ms/step" % (nst, (t1 - t0) / 1e6 / nst, len(step) // nst, sum(r[1] - r[0] for r in step) / 1e6 / nst))


def cls(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    for k, v in (("bn_", "batchnorm"), ("cin1_bn_bwd", "batchnorm(cin1)"), ("wgrad_reduce", "wgrad_reduce"), ("wgrad", "wgrad"), ("cin1", "direct conv"), ("cout1", "direct conv"),
                 ("conv_", "mfma conv fwd/dgrad"), ("bilinear", "resample"), ("avgpool", "resample"), ("pack_", "pack"), ("adam", "adam"), ("loss", "loss"),
                 ("elementwise", "torch elementwise"), ("copyBuffer", "copies"), ("mask_mul", "mask"), ("act_bwd", "act_bwd"), ("colsum", "colsum"), ("absmax", "absmax")):
        if k in n:
            return v
    return n[:30]


qs = collections.defaultdict(list)
for r in step:
    qs[r[2]].append(r)
for q, ev in sorted(qs.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
    tot = sum(e[1] - e[0] for e in ev)
    by = collections.Counter()
    for s, e, _, n in ev:
        by[cls(n)] += e - s
    gaps = collections.Counter()
    idle = 0
    for (s0, e0, _, n0), (s1, e1, _, n1) in zip(ev, ev[1:]):
        if s1 > e0:
            idle += s1 - e0
            gaps[cls(n0) + " -> " + cls(n1)] += s1 - e0
    print("\nqueue %d: %d kernels/step, kernel time %.3f ms/step, gaps between its own kernels %.3f ms/step" % (q, len(ev) // nst, tot / 1e6 / nst, idle / 1e6 / nst))
    for k, v in by.most_common(12):
        print("    %-28s %7.3f ms/step" % (k, v / 1e6 / nst))
    print("  largest gap classes:")
    for k, v in gaps.most_common(8):
        print("    %-50s %7.3f ms/step" % (k, v / 1e6 / nst))

    # the largest individual gaps of the busiest queue, first analysed step: who waited for whom
    if q == max(qs, key=lambda k: sum(e[1] - e[0] for e in qs[k])):
        one = [e for e in ev if e[0] < rows[ends[first + 1]][1]]
        big = sorted(((s1 - e0, n0, n1, s1) for (s0, e0, _, n0), (s1, e1, _, n1) in zip(one, one[1:]) if s1 > e0), reverse=True)[:14]
        print("  largest single gaps (first step): us, after -> before, what ran on the other queues meanwhile")
        for gap, n0, n1, s1 in big:
            other = [r for r in step if r[2] != q and r[0] < s1 and r[1] > s1 - gap]
            names = ", ".join(sorted({re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", r[3]))[:28] for r in other}))[:110]
            short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", n))[:40]
            print("    %6.1f  %-40s -> %-40s | %s" % (gap / 1e3, short(n0), short(n1), names))

# the tail of the first analysed step: what separates the last kernels of the backward chain from the optimizer (us relative to the step's end)
if "--tail" in sys.argv:
    n_tail = int(sys.argv[sys.argv.index("--tail") + 1])
    one = rows[ends[first] + 1:ends[first + 1] + 1]
    tend = max(r[1] for r in one)
    print("\nlast %d kernels of a step (start, end in us before the step's end; queue; name):" % n_tail)
    for s, e, q, n in sorted(one, key=lambda r: r[0])[-n_tail:]:
        print("  %8.1f %8.1f  q%-2d %s" % ((s - tend) / 1e3, (e - tend) / 1e3, q, re.sub(r"\(anonymous namespace\)::", "", n)[:110]))
    print("first 12 kernels of the next step:")
    nxt = rows[ends[first + 1] + 1:ends[first + 1] + 13]
    for s, e, q, n in nxt:
        print("  %8.1f %8.1f  q%-2d %s" % ((s - tend) / 1e3, (e - tend) / 1e3, q, re.sub(r"\(anonymous namespace\)::", "", n)[:110]))
