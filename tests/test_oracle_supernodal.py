"""oracle/ldl_sn.c -- the supernodal multi-threaded host comparator quoted as `cpu_baseline_mt` of bench.py's c2 / c5
lines -- against the scalar oracle (the restatement of src/qdldl/qdldl.rs): same matrix, same permutation up to a
postorder of the elimination tree, same pivot rule.  CPU only."""
import numpy as np
import pytest

from tests import problems


def _kkt_case(hip, oracle, pr, hs=None):
    """the oracle's factorisation under the PRODUCT's permutation (host analysis only: no GPU)"""
    st = hip.Settings.default(device=hip.DEVICE_HOST_ONLY)
    hk = hip.HipKKTSolver(hip.CscMatrix(pr["n"], pr["n"], *pr["P"]), hip.CscMatrix(pr["m"], pr["n"], *pr["A"]),
                          pr["cones"], pr["m"], pr["n"], settings=st)
    cones = oracle.Cones(pr["cones"])
    assert cones.update_scaling(pr["s"], pr["z"])
    ko = oracle.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones, perm=hk.perm)
    assert ko.update(hs)
    import ctypes as C
    L = oracle.lib()
    L.orc_kktsolver_ldl.restype = C.c_void_p
    f = C.c_void_p(L.orc_kktsolver_ldl(ko._h))
    P_I64, P_F64 = C.POINTER(C.c_int64), C.POINTER(C.c_double)
    L.orc_qdldl_perm.restype = P_I64
    L.orc_qdldl_D.restype = P_F64
    perm = np.ctypeslib.as_array(L.orc_qdldl_perm(f), shape=(ko.N,)).copy()
    D_perm = np.ctypeslib.as_array(L.orc_qdldl_D(f), shape=(ko.N,)).copy()
    D = np.empty(ko.N)
    D[perm] = D_perm  # original numbering
    return ko, perm, D


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("which", ["banded_qp", "chordal_sdp", "socp"])
def test_supernodal_comparator_reproduces_the_oracle(hip, oracle, which, threads):
    from oracle import ldl_sn
    if which == "banded_qp":
        pr, hs = problems.random_qp(3000, 6000, band=20, seed=1), None
    elif which == "chordal_sdp":
        pr = problems.chordal_sdp(5, 12, 3, 4, 7, seed=5)
        hs = pr["hsblocks"]
    else:
        pr, hs = problems.portfolio_socp(6, 40, seed=3), None
    ko, perm, D_o = _kkt_case(hip, oracle, pr, hs)
    Kp, Ki, Kx = np.asarray(ko.kkt.colptr), np.asarray(ko.kkt.rowval), np.asarray(ko.kkt.nzval).copy()
    N = ko.N
    signs = np.asarray(ko.dsigns).astype(np.int8)
    # the values the oracle factored: K with the static shift +-eps on its diagonal (directldlkktsolver.rs:217-264)
    cols = np.repeat(np.arange(N), np.diff(Kp))
    dpos = np.nonzero(Ki == cols)[0]
    Kreg = Kx.copy()
    Kreg[dpos] += ko.regularizer * signs[cols[dpos]]
    sn = ldl_sn.LdlSN(N, Kp, Ki, perm, threads=threads)
    assert sorted(sn.perm.tolist()) == list(range(N))
    assert 0 < sn.nsn <= N and sn.panel_entries >= sn.nnzL + N
    ok, nreg = sn.factor(Kreg, signs, ko.settings.dynamic_reg_eps, ko.settings.dynamic_reg_delta)
    assert ok and nreg == ko.ldl_regularize_count()
    # the pivots (a different, but fixed, order of summation: agreement to rounding times the growth of the factorisation)
    assert np.max(np.abs(sn.D() - D_o) / np.abs(D_o)) <= 1e-6
    assert (np.sign(sn.D()) == np.sign(D_o)).all()
    # solve + one refinement round against the oracle's refined solution of the same right-hand side
    rng = np.random.default_rng(2)
    b = rng.standard_normal(N)
    ok_ref, x_ref = ko.solve_full(b)
    assert ok_ref
    x = b.copy()
    sn.solve(x)
    e = np.empty(N)
    sn.residual(Kx, x, b, e)  # (against the UNregularised K, as the reference's refinement does)
    sn.solve(e)
    x += e
    sn.residual(Kx, x, b, e)
    sn.solve(e)
    x += e
    assert np.max(np.abs(x - x_ref)) <= 1e-8 * max(1.0, np.max(np.abs(x_ref)))
    # the residual routine itself against scipy
    import scipy.sparse as sp
    Ku = sp.csc_matrix((Kx, Ki, Kp), shape=(N, N))
    Kf = Ku + sp.triu(Ku, 1).T
    sn.residual(Kx, x, b, e)
    assert np.max(np.abs(e - (b - Kf @ x))) <= 1e-12 * max(1.0, np.max(np.abs(b)))


def test_supernodal_comparator_amalgamates_narrow_supernodes(hip, oracle):
    """relaxed amalgamation: fewer, wider supernodes, same pivots"""
    from oracle import ldl_sn
    pr = problems.random_qp(3000, 6000, band=20, seed=1)
    ko, perm, D_o = _kkt_case(hip, oracle, pr)
    Kp, Ki, Kx = np.asarray(ko.kkt.colptr), np.asarray(ko.kkt.rowval), np.asarray(ko.kkt.nzval).copy()
    N = ko.N
    signs = np.asarray(ko.dsigns).astype(np.int8)
    cols = np.repeat(np.arange(N), np.diff(Kp))
    dpos = np.nonzero(Ki == cols)[0]
    Kx[dpos] += ko.regularizer * signs[cols[dpos]]
    a = ldl_sn.LdlSN(N, Kp, Ki, perm, threads=2, relax=0.0)
    b = ldl_sn.LdlSN(N, Kp, Ki, perm, threads=2, relax=0.3)
    assert b.nsn < a.nsn and b.panel_entries > a.panel_entries and a.nnzL == b.nnzL
    for s in (a, b):
        ok, _ = s.factor(Kx, signs, ko.settings.dynamic_reg_eps, ko.settings.dynamic_reg_delta)
        assert ok
        assert np.max(np.abs(s.D() - D_o) / np.abs(D_o)) <= 1e-6
