#!/usr/bin/env python3
"""Full-size runs of the non-headline BASELINE configs (2, 4, 5) on one MI355X: timing of the
same step bench.py times (1 update + 3 solves, refinement r = 1) and parity -- against the CPU
oracle where that finishes in reasonable time, otherwise the refined residual against an
independent scipy SpMV of the unregularised K.  Prints one JSON line per config.

usage: python tools/scale_check.py [c2] [c4] [c5] [--small]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp

import __graft_entry__ as g
import __graft_entry__ as _graft_entry
_graft_entry.load_package()
import clarabel_rs_amd.synthetic as problems


def run(name, pr, hip, use_oracle, steps=5, hs=None):
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    st = hip.Settings.default(iterative_refinement_max_iter=1, iterative_refinement_reltol=0.0,
                              iterative_refinement_abstol=0.0, use_graph=1 if "--graph" in sys.argv else 0)
    t0 = time.time()
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"], settings=st)
    t_setup = time.time() - t0
    info = ks.linear_solver_info()
    rng = np.random.default_rng(0)
    rhs = [(rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])) for _ in range(3)]
    x, z = np.zeros(pr["n"]), np.zeros(pr["m"])

    def step(timed=None):
        t = time.perf_counter()
        assert ks.update_scaling(pr["s"], pr["z"])
        assert ks.update(hs)
        ks.synchronize()
        tu = time.perf_counter() - t
        t = time.perf_counter()
        for rx, rz in rhs:
            ks.setrhs(rx, rz)
            assert ks.solve(x, z)
        ts = time.perf_counter() - t
        if timed is not None:
            timed.append((tu, ts))

    step()
    tt = []
    for _ in range(steps):
        step(tt)
    tu = float(np.median([a for a, _ in tt]))
    ts = float(np.median([b for _, b in tt]))
    out = {"config": name, "n": pr["n"], "m": pr["m"], "N": ks.N, "nnzK": ks.nnzK, "nnzL": int(info.nnzL),
           "tree_depth": int(info.n_levels), "setup_s": round(t_setup, 2),
           "update_ms(host-ptr API, incl. H2D of s,z)": round(1e3 * tu, 3),
           "3_solves_ms(host-ptr API, incl. H2D/D2H)": round(1e3 * ts, 3),
           "iter_per_s(host-ptr API)": round(1.0 / (tu + ts), 2), "ir_rounds": int(ks.linear_solver_info().last_ir_iterations)}
    # parity
    got = np.concatenate([x, z])
    K = ks.kkt_matrix()
    vals = ks.values()
    Ku = sp.csc_matrix((vals, K.rowval.astype(np.int64), K.colptr.astype(np.int64)), shape=(ks.N, ks.N))
    Kf = Ku + sp.triu(Ku, 1).T
    b = np.concatenate([rhs[-1][0], rhs[-1][1], np.zeros(ks.p)])
    ok, xf = ks.solve_full(b)
    r = b - Kf @ xf
    out["residual_inf_rel"] = float(np.max(np.abs(r)) / max(1.0, np.max(np.abs(b))))
    if use_oracle:
        from oracle import oracle as orc
        ost = orc.Settings.default()
        ost.ir_max_iter, ost.ir_reltol, ost.ir_abstol = 1, 0.0, 0.0
        cones = orc.Cones(pr["cones"])
        cones.update_scaling(pr["s"], pr["z"])
        t0 = time.time()
        ko = orc.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones, settings=ost, perm=ks.perm)
        t1 = time.perf_counter()
        assert ko.update(hs)
        tou = time.perf_counter() - t1
        t1 = time.perf_counter()
        for rx, rz in rhs:
            ko.setrhs(rx, rz)
            okk, xo, zo = ko.solve()
        tos = time.perf_counter() - t1
        ref = np.concatenate([xo, zo])
        out["rel_err_vs_oracle"] = float(np.max(np.abs(got - ref)) / max(1.0, np.max(np.abs(ref))))
        out["oracle_update_ms"] = round(1e3 * tou, 1)
        out["oracle_3_solves_ms"] = round(1e3 * tos, 1)
        out["oracle_iter_per_s_1core"] = round(1.0 / (tou + tos), 3)
    print(json.dumps(out), flush=True)


def main():
    hip = g.load_package()
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c2", "c4"]
    small = "--small" in sys.argv
    if "c2" in which:
        pr = problems.random_qp(10000, 20000, band=30) if small else problems.random_qp(100000, 200000, band=50, seed=1)
        run("C2 random sparse QP", pr, hip, True)
    if "c4" in which:
        pr = problems.batched_socp(32, 2000, 2) if small else problems.batched_socp(1024, 2000, 2, seed=100)
        run("C4 batched %d x SOCP(n=2000)" % (32 if small else 1024), pr, hip, True)
    if "c5m" in which:  # a quarter of config 5 (host setup ~15 s instead of ~60 s): kernel iteration
        pr = problems.chordal_sdp(50, 50, 10, 50, 51, seed=5, with_hs=False)
        run("C5m chordal SDP 50 x PSD(50) + 50 x SOC (PSD scalings and Hs on the device)", pr, hip, False, steps=2)
    if "c5" in which:
        nc, dim = (8, 20) if small else (200, 50)
        pr = problems.chordal_sdp(nc, dim, 10 if not small else 4, nc, 51 if not small else 9, seed=5, with_hs=small)
        run("C5 chordal SDP %d x PSD(%d) + %d x SOC (PSD scalings and Hs on the device)" % (nc, dim, nc), pr, hip, small, steps=2)


if __name__ == "__main__":
    main()
