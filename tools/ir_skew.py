#!/usr/bin/env python3
"""Per-phase statistics of the time stamps EVERY workgroup of a k_bundle_ir launch leaves with CHIP_IR_DEBUG=2
(csrc/capi.cpp: fused_enqueue writes the last launch's stamps to CHIP_IR_DEBUG_FILE, default /tmp/chip_ir_stamps.bin).
usage:  CHIP_IR_DEBUG=2 python bench.py --no-extras --steps 1 --warmup 1 ... ; python tools/ir_skew.py [file]"""
import collections
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/chip_ir_stamps.bin"
raw = open(path, "rb").read()
G = int(np.frombuffer(raw[:4], dtype=np.int32)[0])
t = np.frombuffer(raw[4:], dtype=np.int64).reshape(G, 32)
hw = t[:, 0]
st = t[:, 1:].astype(np.float64)
n = int(np.min(np.sum(st > 0, axis=1)))
st = st[:, :n] * 0.01  # 100 MHz -> us
t0 = st[:, 0].min()
print("workgroups %d, stamps %d, launch span %.1f us" % (G, n, st[:, n - 1].max() - t0))
print("stamp   arrival time since the first workgroup started (us): min / median / p90 / max   | duration of the phase before it: min / median / p90 / max")
for i in range(n):
    a = st[:, i] - t0
    d = st[:, i] - st[:, i - 1] if i else a
    print("%3d    %7.1f %7.1f %7.1f %7.1f   | %7.1f %7.1f %7.1f %7.1f" % (i, a.min(), np.median(a), np.percentile(a, 90), a.max(),
                                                                        d.min(), np.median(d), np.percentile(d, 90), d.max()))
xcc = (hw >> 32) & 0xf
hid = hw & 0xffffffff
cu = (hid >> 8) & 0xf
sh = (hid >> 12) & 0x1
se = (hid >> 13) & 0x7
key = [(int(x), int(s), int(h), int(c)) for x, s, h, c in zip(xcc, se, sh, cu)]
per = collections.Counter(key)
print("distinct (xcc, se, sh, cu): %d; workgroups per CU: min %d max %d" % (len(per), min(per.values()), max(per.values())))
# compute time = sum of the phases that are not waits is unknown here; use the arrival at the LAST compute stamp
work = np.zeros(G)
for i in range(1, n):
    d = st[:, i] - st[:, i - 1]
    work += np.minimum(d, np.median(d) * 3)
xs_ = sorted(set(xcc.tolist()))
print("per-XCC median duration of every phase (us), one row per stamp; last column: spread of the XCC medians")
for i in range(1, n):
    d = st[:, i] - st[:, i - 1]
    med = [float(np.median(d[xcc == x])) for x in xs_]
    print("%3d   " % i + " ".join("%6.1f" % m for m in med) + "   | %5.1f" % (max(med) - min(med)))
print("per-XCC median ARRIVAL at every stamp (us since the first start)")
for i in range(n):
    a = st[:, i] - t0
    print("%3d   " % i + " ".join("%6.1f" % float(np.median(a[xcc == x])) for x in xs_))
print("per-XCC mean start %s" % np.round([np.mean(st[xcc == x, 0] - t0) for x in sorted(set(xcc.tolist()))], 1))
late = np.argsort(-(st[:, 2] - t0))[:10]
print("latest 10 workgroups at stamp 2 (end of the first forward sweep): id, xcc, se, sh, cu, start, arrival")
for b in late:
    print("   %5d  xcc %d se %d sh %d cu %2d   start %6.1f   stamp2 %6.1f   wgs on its CU %d"
          % (b, xcc[b], se[b], sh[b], cu[b], st[b, 0] - t0, st[b, 2] - t0, per[key[b]]))
by_cu_load = collections.defaultdict(list)
for b in range(G):
    by_cu_load[per[key[b]]].append(st[b, 2] - st[b, 0])
for k_ in sorted(by_cu_load):
    print("workgroups sharing a CU with %d in total: %4d, staging + first forward sweep median %.1f us" % (k_, len(by_cu_load[k_]), np.median(by_cu_load[k_])))
