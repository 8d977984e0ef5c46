#!/usr/bin/env python3
"""Host-only census: relaxed chain supernodes among the TOP nodes of a config (columns j -> parent(j),
merged when j is the heaviest child of its parent), their padding (explicit zeros needed to give all
columns of a supernode the structure of its last column) and their share of the factor flops.

usage: python tools/census_supernodes.py [c2] [c5] [--small]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import __graft_entry__ as g
import __graft_entry__ as _graft_entry
_graft_entry.load_package()
import clarabel_rs_amd.synthetic as problems


def census(name, pr, hip):
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"], settings=hip.Settings.default(device=hip.DEVICE_HOST_ONLY))
    et, Lp, Li, lv = ks.symbolic()
    del Li
    N = ks.N
    NF = ks.NF
    cnt = np.diff(Lp).astype(np.int64)
    flops = cnt.astype(np.float64) ** 2
    tot = flops.sum()
    # heaviest child of every node
    best = np.full(N, -1, dtype=np.int64)
    bestc = np.full(N, -1, dtype=np.int64)
    for j in range(N):
        p = et[j]
        if p >= 0 and cnt[j] > bestc[p]:
            bestc[p] = cnt[j]
            best[p] = j
    # chains: start at nodes that are not the heaviest child of their parent (or roots), walk DOWN via best[]
    is_link = np.zeros(N, dtype=bool)  # j linked to parent
    for p in range(N):
        j = best[p]
        if j >= NF and p >= NF:
            is_link[j] = True
    heads = [j for j in range(NF, N) if not is_link[j]]  # top of each chain (the LAST column)
    rows = []
    for h in heads:
        chain = [h]
        j = best[h]
        while j >= NF and is_link[j]:
            chain.append(j)
            j = best[j]
        chain.reverse()  # first eliminated first
        w = len(chain)
        if w < 16:
            continue
        c = cnt[chain]
        last = c[-1]
        padded = last + (w - 1 - np.arange(w))
        zeros = int((padded - c).sum())
        assert (padded >= c).all()
        rows.append((w, int(c.sum()), zeros, float((c.astype(float) ** 2).sum()), int(last),
                     float((padded.astype(float) ** 2).sum())))
    rows.sort(reverse=True)
    print("%s: N=%d NF=%d top=%d nnzL=%d total flops %.3g" % (name, N, NF, N - NF, int(Lp[-1]), tot))
    if not rows:
        print("   no chains of width >= 16")
        return
    r = np.array(rows, dtype=float)
    print("   chains >= 16: %d, columns %d (%.1f%% of top), nnz %d, zeros to add %d (+%.1f%% of nnzL), flops share %.1f%%, padded flops %.3g"
          % (len(rows), int(r[:, 0].sum()), 100 * r[:, 0].sum() / max(1, N - NF), int(r[:, 1].sum()), int(r[:, 2].sum()),
             100 * r[:, 2].sum() / Lp[-1], 100 * r[:, 3].sum() / tot, r[:, 5].sum()))
    for row in rows[:8]:
        print("   w=%d nnz=%d zeros=%d (%.0f%%) below-last=%d" % (row[0], row[1], row[2], 100.0 * row[2] / row[1], row[4]))
    ws = r[:, 0]
    print("   width histogram: >=1024: %d, 256-1023: %d, 64-255: %d, 16-63: %d"
          % ((ws >= 1024).sum(), ((ws >= 256) & (ws < 1024)).sum(), ((ws >= 64) & (ws < 256)).sum(), (ws < 64).sum()))


def main():
    hip = g.load_package()
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c2", "c5"]
    small = "--small" in sys.argv
    if "c2" in which:
        pr = problems.random_qp(10000, 20000, band=30) if small else problems.random_qp(100000, 200000, band=50, seed=1)
        census("C2", pr, hip)
    if "c5" in which:
        nc, dim = (8, 20) if small else (200, 50)
        pr = problems.chordal_sdp(nc, dim, 10 if not small else 4, nc, 51 if not small else 9, seed=5)
        census("C5", pr, hip)


if __name__ == "__main__":
    main()
