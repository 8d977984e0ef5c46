"""Where the time of a step goes that is not inside kernels: from a rocprofv3 kernel trace, the idle time on the device
between consecutive dispatches, attributed to the dispatch that FOLLOWS the gap and summed per kernel name -- plus the
largest gaps (host synchronisations show up as the long ones).
usage: trace_gaps.py <kernel_trace.csv> [skip fraction at the head, default 0.3]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
ev = ev[int(len(ev) * skip):]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(lambda: [0, 0.0])
big = []
prev_end = ev[0][1]
for s, e, name in ev[1:]:
    g = s - prev_end
    if g > 0:
        m = re.search(r"(k_\w+|__amd_\w+)", name)
        short = (m.group(1) if m else name)[:40]
        gaps[short][0] += 1
        gaps[short][1] += g
        big.append((g, short))
    prev_end = max(prev_end, e)
print("dispatches %d, span %.2f ms, inside kernels %.2f ms (%.1f%%), idle %.2f ms" % (len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
print("idle time by the kernel that follows the gap:")
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:20]:
    print("   %-42s n %6d  mean %7.2f us  total %8.3f ms" % (k, n, t / n / 1e3, t / 1e6))
big.sort(reverse=True)
print("largest gaps (us):", ", ".join("%.0f before %s" % (g / 1e3, k) for g, k in big[:12]))
hist = collections.Counter()
for g, _ in big:
    hist[min(int(g / 1000), 20)] += 1
print("gap histogram (us -> count):", ", ".join("%s%d: %d" % (">=" if k == 20 else "", k, v) for k, v in sorted(hist.items())))
