#!/usr/bin/env python3
"""Host-only census (no GPU needed): how much of the numeric factorisation of a config sits in the
structurally dense trailing triangle of L (the last T columns, every row below the diagonal
present)?  Decides whether a dense right-looking tail factorisation pays.

usage: python tools/census_dense_tail.py [c2] [c5] [--small]"""
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
    N = ks.N
    cnt = np.diff(Lp).astype(np.int64)
    full = (N - 1 - np.arange(N)).astype(np.int64)
    dense = cnt == full
    # the tail: largest suffix of columns that are all dense
    nd = np.nonzero(~dense)[0]
    T0 = int(nd[-1]) + 1 if len(nd) else 0
    T = N - T0
    flops = cnt.astype(np.float64) ** 2  # left/right-looking update flops of column j ~ cnt_j^2
    tot = flops.sum()
    # flops of the updates that stay INSIDE the tail: column j in the tail contributes cnt_j^2
    tail = flops[T0:].sum()
    print("%s: N=%d nnzL=%d  dense tail T=%d (T0=%d)  tail nnz=%d (%.1f%%)  tail flops share=%.1f%%  total flops=%.3g"
          % (name, N, int(Lp[-1]), T, T0, int(cnt[T0:].sum()), 100.0 * cnt[T0:].sum() / max(1, Lp[-1]),
             100.0 * tail / tot, tot))
    # relaxed tails: allow a fraction of explicit zeros
    for T1 in (2 * T, 3 * T, 4 * T, 6 * T, 8 * T):
        if T1 <= 0 or T1 > N:
            continue
        s = N - T1
        inside = 0
        # entries of columns s.. all lie in rows > s by construction
        inside = int(cnt[s:].sum())
        fullnz = T1 * (T1 - 1) // 2
        print("   relaxed T=%d: fill %.1f%% of the triangle, flops share %.1f%%, padded dense flops %.3g"
              % (T1, 100.0 * inside / fullnz, 100.0 * flops[s:].sum() / tot, T1 ** 3 / 3.0))


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
    if "c3" in which:
        census("C3", problems.portfolio_socp(), hip)


if __name__ == "__main__":
    main()
