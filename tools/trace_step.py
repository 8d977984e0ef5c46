"""One step of a workload from a rocprofv3 kernel trace, in dispatch order: consecutive dispatches of the same kernel are
folded into one line (count, total and mean duration, span on the device's clock).  The step = the dispatches between the
last two k_scatter_init launches (one per refactorisation).
usage: trace_step.py <kernel_trace.csv> [marker kernel, default k_scatter_init]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_scatter_init"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", name)
    return (m.group(1) if m else name)[:44]
marks = [i for i, e in enumerate(ev) if marker in e[2]]
if len(marks) < 2:
    sys.exit("marker kernel seen %d times" % len(marks))
a, b = marks[-2], marks[-1]
step = ev[a:b]
print("step: %d dispatches, span %.3f ms, inside kernels %.3f ms" % (len(step), (step[-1][1] - step[0][0]) / 1e6, sum(e - s for s, e, _ in step) / 1e6))
runs = []
for s, e, n in step:
    k = short(n)
    if runs and runs[-1][0] == k:
        runs[-1][1] += 1
        runs[-1][2] += e - s
        runs[-1][4] = e
    else:
        runs.append([k, 1, e - s, s, e])
t0 = step[0][0]
for k, n, d, s, e in runs:
    print("%9.1f us  %-44s x%-4d total %8.1f us  mean %7.1f us  span %8.1f us" % ((s - t0) / 1e3, k, n, d / 1e3, d / n / 1e3, (e - s) / 1e3))
# every dispatch of the step with the idle time since the previous dispatch ended (what a kernel boundary costs here)
print("dispatch by dispatch: start (us), duration (us), gap since the previous dispatch ended (us)")
prev_end = None
for s, e, n in step:
    print("%9.1f  %8.1f  %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, short(n)))
    prev_end = e
