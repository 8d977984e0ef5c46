#!/usr/bin/env python3
"""Generates tests/golden/c5_full_oracle.npz: the CPU oracle's answers on BASELINE config 5 at its FULL size
(200 x PSDTriangleCone(50) cliques with overlap 10 + 200 x SecondOrderCone(51); N = 277 345).

The scalar oracle (oracle/qdldl_oracle.c, the restatement of src/qdldl/qdldl.rs:469-768) needs ~10 CPU-minutes
for this factorisation, far beyond what a GPU parity test may spend, so its answers are committed as a fixture:

  * the oracle runs the reference sequence  update (Hs scatter + static regularisation + refactor)  ->
    setrhs -> solve (+ iterative refinement, DEFAULT settings)  for NRHS seeded right-hand sides;
  * Hs blocks of the PSD cones: oracle/psd_numpy restatement of psdtrianglecone.rs:144-204,467-509
    (synthetic.psd_scaling_Hs), SOC blocks from the C oracle's cones;
  * permutation: the product's own host analysis (HOST_ONLY handle: no GPU needed) -- the solution of the
    refined solve is ordering independent up to rounding, the test compares at 1e-8;
  * stored: the NRHS post-refinement solutions (x, z) in full, the static regulariser, the number of
    dynamically regularised pivots, positive inertia, the refinement rounds taken, max|K.nzval| and a seeded
    65536-entry sample of the oracle's K.nzval after the update (the device's fused Hs write is compared
    with it), and the right-hand-side seed.

Run from the repo root (no GPU needed):   python tests/golden/make_c5_fixture.py [ncliques]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

RHS_SEED = 20260925
NRHS = 2
NSAMPLE = 65536


def positive_inertia(orc, ko):
    """positive pivots of the oracle's last factorisation (qdldl.rs:653-655)"""
    import ctypes as C
    lib = orc.lib()
    lib.orc_kktsolver_ldl.restype = C.c_void_p
    return int(lib.orc_qdldl_positive_inertia(C.c_void_p(lib.orc_kktsolver_ldl(ko._h))))


def main():
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out = os.path.join(ROOT, "tests", "golden", "c5_full_oracle.npz" if nc == 200 else "c5_%d_oracle.npz" % nc)
    hip = graft.load_package()
    from oracle import oracle as orc
    import clarabel_rs_amd.synthetic as problems
    t0 = time.time()
    pr = problems.chordal_sdp(nc, 50, 10, nc, 51, seed=5, with_hs=True)
    print("generated: n=%d m=%d (%.1f s)" % (pr["n"], pr["m"], time.time() - t0), flush=True)
    st = hip.Settings.default(device=hip.DEVICE_HOST_ONLY)
    ks = hip.HipKKTSolver(hip.CscMatrix(pr["n"], pr["n"], *pr["P"]), hip.CscMatrix(pr["m"], pr["n"], *pr["A"]),
                          pr["cones"], pr["m"], pr["n"], settings=st)
    perm = np.asarray(ks.perm).copy()
    print("host analysis: N=%d (%.1f s)" % (ks.N, time.time() - t0), flush=True)
    cones = orc.Cones(pr["cones"])
    assert cones.update_scaling(pr["s"], pr["z"])
    ko = orc.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones, perm=perm)
    t1 = time.time()
    assert ko.update(pr["hsblocks"])
    print("oracle update + refactor: %.1f s" % (time.time() - t1), flush=True)
    rng = np.random.default_rng(RHS_SEED)
    sols, rounds = [], []
    for k in range(NRHS):
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ko.setrhs(rx, rz)
        t1 = time.time()
        ok, x, z = ko.solve()
        assert ok
        sols.append(np.concatenate([x, z]))
        rounds.append(int(ko.last_ir_iters))
        print("oracle solve %d: %.1f s" % (k, time.time() - t1), flush=True)
    nzval = np.asarray(ko.kkt.nzval)
    srng = np.random.default_rng(RHS_SEED + 1)
    sample_idx = np.sort(srng.choice(nzval.size, size=min(NSAMPLE, nzval.size), replace=False))
    np.savez_compressed(
        out, ncliques=nc, n=pr["n"], m=pr["m"], N=ks.N, rhs_seed=RHS_SEED, nrhs=NRHS,
        solutions=np.stack(sols), regularizer=float(ko.regularizer),
        regularize_count=int(ko.ldl_regularize_count()),
        positive_inertia=positive_inertia(orc, ko),
        ir_rounds=np.asarray(rounds), k_absmax=float(np.max(np.abs(nzval))), k_nnz=int(nzval.size),
        k_sample_idx=sample_idx, k_sample_val=nzval[sample_idx])
    print("wrote %s (%.1f MB), total %.1f s" % (out, os.path.getsize(out) / 1e6, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
