"""text timeline of one steady-state train step from a rocprofv3 kernel trace: per hardware queue, which kernels ran when"""
import csv, sys, re
path, which = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
# step boundaries: the Cin=1 stride-(1,2) D.conv1 forward on the real clip is the first conv of a step; use adam_kernel pairs instead
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[3]]
# two adam per step: the second one ends the step
ends = adam[1::2]
a, b = ends[which] + 1, ends[which + 1] + 1
step = rows[a:b]
t0 = min(r[0] for r in step)
print("step wall %.3f ms, %d kernels, kernel-time sum %.3f ms" % ((max(r[1] for r in step) - t0) / 1e6, len(step), sum(r[1] - r[0] for r in step) / 1e6))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]
qs = sorted(set(r[2] for r in step))
print("queues", qs)
for s, e, q, n in step:
    col = qs.index(q)
    print("%8.1f %8.1f %7.1f  %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, " " * (col * 46), short(n)))
