"""Per-dispatch view of a rocprofv3 kernel trace: for one kernel (substring of its name) the launches grouped by
grid size, with count / mean / total duration.  usage: trace_summary.py <kernel_trace.csv> <name substring> [...]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for pat in sys.argv[2:]:
    sel = [r for r in rows if pat in r["Kernel_Name"]]
    groups = collections.defaultdict(list)
    for r in sel:
        wg = (int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"]))
        g = (int(r["Grid_Size_X"]) // wg[0], int(r["Grid_Size_Y"]) // wg[1], int(r["Grid_Size_Z"]) // wg[2])
        groups[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in groups.values())
    print("%s: %d launches, %.2f ms" % (pat, len(sel), tot / 1e3))
    for g, v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print("   grid %-16s wgs %6d  n %5d  mean %8.1f us  total %8.2f ms" % (g, g[0] * g[1] * g[2], len(v), sum(v) / len(v), sum(v) / 1e3))
