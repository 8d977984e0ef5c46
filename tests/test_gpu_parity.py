"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerance (BASELINE.json north_star / SURVEY.md 8c): 1e-8 relative on the
post-refinement KKT solution with the SAME permutation injected into the oracle; LDL-level
known-answer tests keep the reference's own 1e-8 / 1e-10 absolute bounds."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from tests import problems

pytestmark = pytest.mark.gpu

TOL = 1e-8


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))


# ---- L1: DirectLDLSolver known-answer tests ---------------------------------------
def mat4(hip):
    return hip.CscMatrix(4, 4, [0, 1, 3, 6, 8], [0, 0, 1, 0, 1, 2, 2, 3], [8., -3., 8., 2., -1., 8., -1., 1.])


@pytest.mark.parametrize("perm", [None, [0, 1, 2, 3], [3, 0, 2, 1]])
def test_solve_basic(hip, perm):
    # qdldl/test.rs:194-230
    st = hip.Settings.default(dynamic_regularization_eps=1e-12, dynamic_regularization_delta=1e-7)
    f = hip.HipDirectLDLSolver(mat4(hip), [1, 1, 1, 1], st, perm=perm)
    x = np.zeros(4)
    with pytest.raises(hip.ChipError) as e:  # logical-only until the first refactor (test.rs:232-247)
        f.solve(None, x, np.array([20.0, -22.0, 32.0, -7.0]))
    assert e.value.code == hip.ERR_NOT_FACTORED
    assert f.refactor()
    b = np.array([20.0, -22.0, 32.0, -7.0])
    f.solve(None, x, b)
    assert np.max(np.abs(x - np.array([1., -2., 3., -4.]))) <= 1e-8
    assert list(b) == [20.0, -22.0, 32.0, -7.0]


def test_faer_kat(hip):
    # ldlsolvers/faer_ldl.rs:352-404
    K = hip.CscMatrix(6, 6, [0, 1, 2, 4, 6, 8, 10], [0, 1, 0, 2, 1, 3, 0, 4, 1, 5],
                      [1.0, 2.0, 1.0, -1.0, 1.0, -2.0, -1.0, -3.0, -1.0, -4.0])
    f = hip.HipDirectLDLSolver(K, [1, 1, -1, -1, -1, -1])
    assert f.refactor()
    x = np.zeros(6)
    b = np.array([1., 2., 3., 4., 5., 6.])
    f.solve(None, x, b)
    xs = np.array([1.0, 0.9090909090909091, -2.0, -1.5454545454545454, -2.0, -1.7272727272727275])
    assert np.max(np.abs(x - xs)) < 1e-10
    f.update_values([9], [-10.0])
    assert f.refactor()
    f.solve(None, x, b)
    xs = np.array([1.0, 1.3076923076923077, -2.0, -1.346153846153846, -2.0, -0.7307692307692306])
    assert np.max(np.abs(x - xs)) < 1e-10
    f.offset_values([1, 2], 3., [1, -1])
    f.scale_values([1, 2], 2.)
    assert f.refactor()
    info = f.linear_solver_info()
    assert info.nnzA == 10 and info.positive_inertia == 2 and info.direct == 1 and info.threads == 1


def test_zero_pivot_regularised(hip, oracle):
    # qdldl/test.rs:266-283 with regularisation ON (as the KKT adapter always has it,
    # ldlsolvers/qdldl.rs:38): the zero pivot is replaced by delta*sign, same as the oracle
    K = mat4(hip)
    K.nzval[-1] = 0.0
    f = hip.HipDirectLDLSolver(K, [1, 1, 1, 1], perm=[3, 0, 1, 2])
    assert f.refactor()
    o = oracle.QDLDL(4, K.colptr, K.rowval, K.nzval, perm=f.perm, Dsigns=[1, 1, 1, 1], logical=True,
                     regularize_eps=1e-13, regularize_delta=2e-7)
    assert o.refactor()
    assert f.linear_solver_info().regularize_count == o.regularize_count >= 1
    x = np.zeros(4)
    b = np.array([1., 2., 3., 4.])
    f.solve(None, x, b)
    assert relerr(x, o.solve(b)) <= 1e-9


def test_nonfinite_values_fail_refactor(hip):
    K = mat4(hip)
    f = hip.HipDirectLDLSolver(K, [1, 1, 1, 1])
    f.update_values([0], [np.nan])
    assert f.refactor() is False  # Dinv.is_finite() == false, ldlsolvers/qdldl.rs:105


def _rand_quasidef(rng, n1, n2, density):
    import scipy.sparse as sp
    rs = np.random.RandomState(int(rng.integers(1 << 30)))
    B = sp.random(n2, n1, density=density, random_state=rs, format="csc")
    H = sp.random(n1, n1, density=density / 2, random_state=rs, format="csc")
    H = H @ H.T + sp.diags(rng.uniform(0.5, 2.0, n1))
    G = sp.diags(rng.uniform(0.5, 2.0, n2))
    K = sp.bmat([[H, B.T], [B, -G]], format="csc")
    K = sp.triu(K, format="csc")
    K.sort_indices()
    ds = np.array([1] * n1 + [-1] * n2, dtype=np.int8)
    return K, ds


def test_ldl_dense_front_beyond_lds(hip, oracle):
    """one fully dense quasidefinite matrix of order 2300: the leading columns of L are longer
    than the factor kernel's LDS column buffer (2048) -> flattened update with L2 atomics"""
    rng = np.random.default_rng(3)
    n1, n2 = 2200, 100
    M = rng.standard_normal((n1, 40))
    A = rng.standard_normal((n2, n1))
    K = np.block([[M @ M.T + n1 * np.eye(n1), A.T], [A, -(n2 * np.eye(n2))]])
    K = sp.triu(sp.csc_matrix(K), format="csc")
    K.sort_indices()
    ds = np.array([1] * n1 + [-1] * n2, dtype=np.int8)
    Kc = hip.CscMatrix.from_scipy(K)
    f = hip.HipDirectLDLSolver(Kc, ds)
    assert f.refactor()
    o = oracle.QDLDL(n1 + n2, Kc.colptr, Kc.rowval, Kc.nzval, perm=f.perm, Dsigns=ds, logical=True,
                     regularize_eps=1e-13, regularize_delta=2e-7)
    assert o.refactor()
    Lp, Li, Lx, D, Dinv = f.factors()
    assert np.diff(Lp).max() > 2048
    assert np.array_equal(Li, o.Li) and relerr(D, o.D) <= 1e-10 and relerr(Lx, o.Lx) <= 1e-9
    b = rng.standard_normal(n1 + n2)
    x = np.zeros(n1 + n2)
    f.solve(None, x, b)
    assert relerr(x, o.solve(b)) <= TOL


@pytest.mark.parametrize("n1,n2,density,seed", [(30, 20, 0.1, 0), (400, 300, 0.01, 1), (2000, 3000, 0.002, 2)])
def test_ldl_random_quasidefinite(hip, oracle, n1, n2, density, seed):
    """factor + solve of random sparse quasidefinite matrices (general fill: exercises the
    workgroup-per-column kernels), factors compared entry by entry with the oracle."""
    rng = np.random.default_rng(seed)
    K, ds = _rand_quasidef(rng, n1, n2, density)
    n = n1 + n2
    Kc = hip.CscMatrix.from_scipy(K)
    f = hip.HipDirectLDLSolver(Kc, ds)
    assert f.refactor()
    o = oracle.QDLDL(n, Kc.colptr, Kc.rowval, Kc.nzval, perm=f.perm, Dsigns=ds, logical=True,
                     regularize_eps=1e-13, regularize_delta=2e-7)
    assert o.refactor()
    Lp, Li, Lx, D, Dinv = f.factors()
    if not f.supernodes():
        assert np.array_equal(Lp, o.Lp) and np.array_equal(Li, o.Li)
        assert relerr(Lx, o.Lx) <= 1e-9
    else:  # chain supernodes pad the pattern with explicit zeros: compare as matrices
        La = sp.csc_matrix((Lx, Li, Lp), shape=(n, n))
        Lo = sp.csc_matrix((o.Lx, o.Li, o.Lp), shape=(n, n))
        assert abs(La - Lo).max() <= 1e-9 * max(1.0, np.abs(o.Lx).max())
        pat = sp.csc_matrix((np.ones(len(Li)), Li, Lp), shape=(n, n)) - sp.csc_matrix((np.ones(len(o.Li)), o.Li, o.Lp), shape=(n, n))
        assert pat.data.min(initial=0.0) >= 0.0
    assert relerr(D, o.D) <= 1e-10
    assert f.linear_solver_info().positive_inertia == o.positive_inertia == n1
    b = rng.standard_normal(n)
    x = np.zeros(n)
    f.solve(None, x, b)
    assert relerr(x, o.solve(b)) <= TOL


# ---- L2: KKTSolver on the configs ---------------------------------------------------
def _solvers(hip, oracle, pr, settings=None, hs=None, late=False):
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"], settings=settings)
    cones = oracle.Cones(pr["cones"])
    ost = oracle.Settings.default()
    if settings is not None:
        ost.ir_max_iter = settings.iterative_refinement_max_iter
        ost.ir_reltol = settings.iterative_refinement_reltol
        ost.ir_abstol = settings.iterative_refinement_abstol
        ost.ir_enable = settings.iterative_refinement_enable
        ost.ir_stop_ratio = settings.iterative_refinement_stop_ratio
        ost.static_reg_enable = settings.static_regularization_enable
    ko = oracle.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones, settings=ost, perm=ks.perm)
    return ks, ko, cones


def _check_update_and_solve(hip, oracle, pr, hs=None, nrhs=2, settings=None, tol=TOL):
    ks, ko, cones = _solvers(hip, oracle, pr, settings)
    assert ks.update_scaling(pr["s"], pr["z"])
    assert cones.update_scaling(pr["s"], pr["z"])
    assert ks.update(hs)
    assert ko.update(hs)
    # the device copy of K.nzval after the fused Hs / sparse-cone update == the oracle's K
    assert relerr(ks.values(), ko.kkt.nzval) <= 1e-13
    assert abs(ks.linear_solver_info().last_regularizer - ko.regularizer) <= 1e-20 + 1e-12 * ko.regularizer
    rng = np.random.default_rng(42)
    for _ in range(nrhs):
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ks.setrhs(rx, rz)
        ko.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        ok, xo, zo = ko.solve()
        assert ok
        assert relerr(np.concatenate([x, z]), np.concatenate([xo, zo])) <= tol
    return ks, ko


def test_c1_basic_qp(hip, oracle):
    _check_update_and_solve(hip, oracle, problems.basic_qp())


@pytest.mark.parametrize("late", [False, True])
def test_c2_random_qp(hip, oracle, late):
    _check_update_and_solve(hip, oracle, problems.random_qp(3000, 6000, band=20, seed=1, late=late))


@pytest.mark.parametrize("which", ["band20", "band50", "band50_late", "chordal_sdp", "no_supernodes"])
def test_bundle_sweeps_entry_parallel_against_row_form(hip, oracle, which, monkeypatch):
    """systems with a level-scheduled top: the stand-alone bundle sweeps entry-parallel (bundle_solve.hip:
    k_bundle_sweep_flat -- one stream of (row, column, value) batches per bundle, an LDS atomic per entry, the top rows'
    share of the backward sweep in a flat prologue) against the oracle, and against the row- / column-per-thread form of
    the same handle shape (CHIP_NO_BUNDLE_FLAT_SWEEP): refined solutions agree to rounding, unrefined ones to the accuracy
    of one LDL' solve"""
    hs = None
    if which == "band20":
        pr = problems.random_qp(3000, 6000, band=20, seed=1)
    elif which == "chordal_sdp":
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
        hs = pr["hsblocks"]
    else:
        pr = problems.random_qp(20000, 40000, band=50, seed=1, late=(which == "band50_late"))
        if which == "no_supernodes":
            monkeypatch.setenv("CHIP_NO_SNODE", "1")
    ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=3)
    monkeypatch.setenv("CHIP_NO_BUNDLE_FLAT_SWEEP", "1")
    ks0, _ = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1)
    monkeypatch.delenv("CHIP_NO_BUNDLE_FLAT_SWEEP")
    rng = np.random.default_rng(23)
    for _ in range(2):
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        sols = []
        for k in (ks0, ks):
            assert k.update_scaling(pr["s"], pr["z"]) and k.update(hs)
            k.setrhs(rx, rz)
            x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
            assert k.solve(x, z)
            sols.append(np.concatenate([x, z]))
        assert relerr(sols[1], sols[0]) <= 1e-9
    st = hip.Settings.default(iterative_refinement_enable=0)
    _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1, settings=st, tol=1e-5 if "late" in which else 1e-6)


def test_c2_solve_sequence_as_hipgraph(hip, oracle):
    """settings.use_graph: the launch sequence of the LDL' solves replayed as hipGraphs -- same
    results as direct launches (several right-hand sides, so every graph is replayed)"""
    st = hip.Settings.default(use_graph=1)
    _check_update_and_solve(hip, oracle, problems.random_qp(3000, 6000, band=20, seed=1), nrhs=4, settings=st)


@pytest.mark.parametrize("snodes", [True, False])
def test_c2_tall_top_blocked_substitution(hip, oracle, snodes, monkeypatch):
    """a banded QP whose elimination tree has a tall top (~2200 sequential levels, ~6800 rows): with
    chain supernodes (default) the top runs as dense trapezoids; without them (CHIP_NO_SNODE) the
    substitutions run block by block with inverted 128-row diagonal blocks (k_topblk_*) and the heavy
    columns through the chunked column kernel"""
    if not snodes:
        monkeypatch.setenv("CHIP_NO_SNODE", "1")
    pr = problems.random_qp(20000, 40000, band=50, seed=1)
    ks, ko = _check_update_and_solve(hip, oracle, pr, nrhs=2)
    assert ks.N - ks.NF >= 4 * 128
    assert (len(ks.supernodes()) > 0) == snodes


@pytest.mark.parametrize("late", [False, True])
def test_c3_portfolio_socp(hip, oracle, late):
    ks, ko = _check_update_and_solve(hip, oracle, problems.portfolio_socp(12, 300, seed=3, late=late))
    assert ks.linear_solver_info().n_levels <= 8


def test_c3_several_dense_top_rows(hip, oracle):
    """three dense coupling rows (budget + two more equality rows over all x): the top of the
    elimination tree has k = 3 nodes, folded into the bundle kernels with a 3 x 3 finishing step"""
    import scipy.sparse as sp2
    pr = problems.portfolio_socp(12, 600, seed=21)
    n, m = pr["n"], pr["m"]
    rng = np.random.default_rng(2)
    A = sp2.csc_matrix((pr["A"][2], pr["A"][1], pr["A"][0]), shape=(m, n))
    extra = sp2.csc_matrix(rng.uniform(0.5, 1.5, (2, n)))
    A2 = sp2.vstack([A, extra], format="csc")
    A2.sort_indices()
    pr2 = dict(n=n, m=m + 2, P=pr["P"], A=problems._csc(A2), cones=list(pr["cones"]) + [(0, 2)],
               s=np.concatenate([pr["s"], np.zeros(2)]), z=np.concatenate([pr["z"], np.zeros(2)]))
    ks, ko = _check_update_and_solve(hip, oracle, pr2, nrhs=2)
    assert ks.N - ks.NF == 3


def test_c3_dense_soc_blocks(hip, oracle):
    # SOC(4): dense Hs block path (socone.rs:224-245)
    _check_update_and_solve(hip, oracle, problems.portfolio_socp(7, 3, seed=4))


def test_c3_big_rows_split_kernels(hip, oracle):
    """budget row longer than the B-chunk threshold (16384): exercises the chunked
    atomics kernels of factor / forward solve / symv"""
    _check_update_and_solve(hip, oracle, problems.portfolio_socp(40, 500, seed=9), nrhs=1)


def test_c4_batched(hip, oracle):
    _check_update_and_solve(hip, oracle, problems.batched_socp(16, 200, 2, seed=100))


def test_c5_chordal_sdp_host_hs(hip, oracle):
    pr = problems.chordal_sdp(6, 6, 2, 3, 7, seed=5)
    _check_update_and_solve(hip, oracle, pr, hs=pr["hsblocks"])


@pytest.mark.parametrize("dim", [3, 8, 21, 50, 72])
def test_c5_psd_scaling_on_device(hip, oracle, dim):
    """PSDTriangleCone update_scaling on the device (psdtrianglecone.rs:144-204: chol, chol, SVD,
    R R', skron) -- the Hs blocks are NOT handed over by the host; the oracle side uses the numpy
    restatement tests/problems.psd_scaling_Hs for the same (S, Z)"""
    # dim 50 = BASELINE config 5's clique size; one clique there (the generator's 4*ncliques x
    # variables would otherwise fuse four 1275-wide blocks into a single dense front)
    # (72 > 64: the work matrices of the cone kernels no longer fit LDS and live in HBM scratch)
    pr = problems.chordal_sdp(1 if dim >= 50 else 4, dim, min(3, dim - 1), 2, 7, seed=dim)
    ks, ko, cones = _solvers(hip, oracle, pr)
    assert ks.update_scaling(pr["s"], pr["z"]) and cones.update_scaling(pr["s"], pr["z"])
    assert ks.update()              # no hsblocks: computed on the device
    assert ko.update(pr["hsblocks"])
    assert relerr(ks.values(), ko.kkt.nzval) <= 1e-11
    rng = np.random.default_rng(1)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    ks.setrhs(rx, rz)
    ko.setrhs(rx, rz)
    x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
    assert ks.solve(x, z)
    ok, xo, zo = ko.solve()
    assert ok and relerr(np.concatenate([x, z]), np.concatenate([xo, zo])) <= TOL


@pytest.mark.parametrize("dim", [8, 21, 50])
def test_psd_hs_written_row_by_row_of_the_value_store(hip, oracle, dim, monkeypatch):
    """when every PSD cone's Hs block is one of the dense diagonal blocks of the top, the cones write Hs in the order the
    entries have in the device's value store (k_psd_write_hs_rows: coalesced stores, no index array) instead of
    scattering through mapHs: K against the oracle's, and against the scattered form (CHIP_NO_PSD_ROWS)"""
    pr = problems.chordal_sdp(1 if dim >= 50 else 4, dim, min(3, dim - 1), 2, 7, seed=dim)
    monkeypatch.setenv("CHIP_DENSE_SYMV_MIN", "1")
    vals = {}
    for form in ("rows", "CHIP_NO_PSD_ROWS"):
        if form != "rows":
            monkeypatch.setenv(form, "1")
        ks, ko, cones = _solvers(hip, oracle, pr)
        if dim >= 50:  # (a single clique: its block is in the top as a whole; the smaller instances may keep cone rows in
            #            bundles.  The counter tells the structure; CHIP_NO_PSD_ROWS only selects the launch.)
            assert hip.debug_counter(ks, "psd_hs_row_blocks") > 0
        assert ks.update_scaling(pr["s"], pr["z"]) and cones.update_scaling(pr["s"], pr["z"])
        assert ks.update()
        assert ko.update(pr["hsblocks"])
        assert relerr(ks.values(), ko.kkt.nzval) <= 1e-11
        vals[form] = ks.values().copy()
        rng = np.random.default_rng(1)
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ks.setrhs(rx, rz)
        ko.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        ok, xo, zo = ko.solve()
        assert ok and relerr(np.concatenate([x, z]), np.concatenate([xo, zo])) <= TOL
    assert np.array_equal(vals["rows"], vals["CHIP_NO_PSD_ROWS"])  # (the same arithmetic per entry)


@pytest.mark.parametrize("which", ["one_clique_50", "four_cliques_40"])
def test_psd_hs_written_into_the_factor_storage_directly(hip, oracle, which, monkeypatch):
    """PSD blocks that are dense diagonal blocks of the top AND contiguous row by row in L's panels are written into K and
    into L by the one kernel (k_psd_write_hs_rows with Lx; Engine::hs_direct_begin clears the fill-in range ahead of it);
    the refactor then scatters only the REST of K's top entries (k_scatter_rest) instead of reading 1.6e8 entries back
    (config 5).  First update against the oracle; a second update at another scaling point (its refactor must not see
    the first one's factor in L) against the scattered form (CHIP_NO_HS_DIRECT): same K, solutions equal to rounding."""
    pr = problems.chordal_sdp(1, 50, 3, 2, 7, seed=50) if which == "one_clique_50" else problems.chordal_sdp(4, 40, 24, 2, 9, seed=11)
    monkeypatch.setenv("CHIP_DENSE_SYMV_MIN", "1")
    monkeypatch.setenv("CHIP_FILL_RANGE_MIN", "0")  # (the top's fill-in cleared as a range whatever its share: what the direct form needs)
    res = {}
    for form in ("direct", "CHIP_NO_HS_DIRECT"):
        if form != "direct":
            monkeypatch.setenv(form, "1")
        ks, ko, cones = _solvers(hip, oracle, pr)
        out = []
        for it in range(2):
            s_, z_ = pr["s"] * (1.0 + 0.25 * it), pr["z"] / (1.0 + 0.5 * it)
            assert ks.update_scaling(s_, z_)
            assert ks.update()
            rng = np.random.default_rng(1 + it)
            rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
            ks.setrhs(rx, rz)
            x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
            assert ks.solve(x, z)
            out.append((ks.values().copy(), np.concatenate([x, z])))
            if it == 0:
                assert cones.update_scaling(s_, z_) and ko.update(pr["hsblocks"])
                assert relerr(ks.values(), ko.kkt.nzval) <= 1e-11
                ko.setrhs(rx, rz)
                ok, xo, zo = ko.solve()
                assert ok and relerr(out[0][1], np.concatenate([xo, zo])) <= TOL
        n_direct = hip.debug_counter(ks, "hs_direct_refactors")
        assert (n_direct == 2) == (form == "direct"), (form, n_direct)
        # scaling + update as ONE enqueue (chip_kkt_update_scaled_enqueue): the clear of L's fill-in range runs on the
        # second stream beside the scaling kernels (Engine::hs_direct_prefill_async), twice in a row
        for it in range(2, 4):
            s_, z_ = pr["s"] * (1.0 + 0.25 * it), pr["z"] / (1.0 + 0.5 * it)
            s_d, z_d = hip.DeviceArray(s_), hip.DeviceArray(z_)
            ks.update_scaled_enqueue(s_d.ptr, z_d.ptr)
            rng = np.random.default_rng(1 + it)
            rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
            rxd, rzd, lhs = hip.DeviceArray(rx), hip.DeviceArray(rz), hip.DeviceArray(pr["n"] + pr["m"])
            ks.setrhs_dev(rxd.ptr, rzd.ptr)
            ks.solve_dev_enqueue(lhs.ptr, lhs.ptr + 8 * pr["n"])
            uok, sok = ks.collect()
            assert uok and sok == [True]
            out.append((ks.values().copy(), lhs.numpy().copy()))
        assert (hip.debug_counter(ks, "hs_direct_refactors") == 4) == (form == "direct")
        res[form] = out
    for it in range(4):
        assert np.array_equal(res["direct"][it][0], res["CHIP_NO_HS_DIRECT"][it][0])  # K: the same arithmetic per entry
        assert relerr(res["direct"][it][1], res["CHIP_NO_HS_DIRECT"][it][1]) <= 1e-10


def test_psd_scaling_failure_reported(hip):
    pr = problems.chordal_sdp(2, 4, 2, 1, 6, seed=2)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    s = pr["s"].copy()
    s[0] = -1.0  # S[0,0] < 0: Cholesky fails -> update_scaling false (psdtrianglecone.rs:165-169)
    assert ks.update_scaling(s, pr["z"]) is False


@pytest.mark.parametrize("strategy", [0, 1])
def test_mixed_conic_exp_pow_on_device(hip, oracle, strategy):
    """Exponential / Power cone scalings computed ON THE DEVICE (expcone.rs:106-124,
    powcone.rs:99-117, nonsymmetric_common.rs:53-143), both scaling strategies"""
    pr = problems.mixed_conic(seed=11)
    ks, ko, cones = _solvers(hip, oracle, pr)
    mu = 0.43
    assert ks.update_scaling(pr["s"], pr["z"], mu, strategy)
    assert cones.update_scaling(pr["s"], pr["z"], mu, strategy)
    assert ks.update() and ko.update()
    assert relerr(ks.values(), ko.kkt.nzval) <= 1e-12
    rng = np.random.default_rng(2)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    ks.setrhs(rx, rz)
    ko.setrhs(rx, rz)
    x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
    assert ks.solve(x, z)
    ok, xo, zo = ko.solve()
    assert ok and relerr(np.concatenate([x, z]), np.concatenate([xo, zo])) <= TOL
    # mul_Hs on the device against the oracle
    xt = hip.DeviceArray(rz)
    yt = hip.DeviceArray(pr["m"])
    ks.mul_Hs_dev(yt.ptr, xt.ptr)
    ks.synchronize()
    assert relerr(yt.numpy(), cones.mul_Hs(rz)) <= 1e-12


def test_refactor_sequence_and_update_PA(hip, oracle):
    """several IPM-like iterations on one handle (values change, pattern fixed) + update_P/A
    (directldlkktsolver.rs:191-197)"""
    pr = problems.random_qp(500, 1000, band=10, seed=21)
    ks, ko, cones = _solvers(hip, oracle, pr)
    rng = np.random.default_rng(3)
    for it in range(3):
        s = rng.uniform(0.1, 3.0, pr["m"]) * 10.0 ** (-it)
        z = rng.uniform(0.1, 3.0, pr["m"])
        assert ks.update_scaling(s, z) and cones.update_scaling(s, z)
        if it == 1:
            Px = pr["P"][2] * 1.5
            Ax = pr["A"][2] * 0.5
            ks.update_P(Px)
            ks.update_A(Ax)
            L = oracle.lib()
            # oracle: same value overwrite through the maps
            km = ko.kkt
            v = km.nzval
            v[km.map("P", len(Px))] = Px
            v[km.map("A", len(Ax))] = Ax
            km.set_nzval(v)
            import ctypes as C
            ldl = L.orc_kktsolver_ldl(ko._h)
            idx = np.concatenate([km.map("P", len(Px)), km.map("A", len(Ax))]).astype(np.int64)
            vals = np.concatenate([Px, Ax])
            L.orc_qdldl_update_values(C.c_void_p(ldl), idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                      vals.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(len(idx)))
        assert ks.update() and ko.update()
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ks.setrhs(rx, rz)
        ko.setrhs(rx, rz)
        x, zz = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, zz)
        ok, xo, zo = ko.solve()
        assert relerr(np.concatenate([x, zz]), np.concatenate([xo, zo])) <= TOL


@pytest.mark.parametrize("which", ["qp", "arrow", "supernodes"])
def test_l1_fast_path_registered_sets_and_device_refinement(hip, oracle, which):
    """the strict drop-in's fast path (chip_ldl_register_index / *_values_id / chip_ldl_solve_refined / chip_ldl_pin_buffer):
    the call sequence DirectLDLKKTSolver makes per interior-point iteration -- update_values on the changed entries
    (directldlkktsolver.rs:143), the static regulariser on (offset_values, :245), refactor, the regulariser off (:255-261),
    solve with refinement (:266-321) -- through registered index sets and the device-resident refinement, through the plain
    calls, and mixed on one handle: solutions against the oracle's KKTSolver on the same K, three iterations"""
    if which == "qp":
        pr = problems.random_qp(500, 1000, band=10, seed=21)
    elif which == "arrow":
        pr = problems.portfolio_socp(6, 120, seed=3)
    else:
        pr = problems.random_qp(3000, 6000, band=30, seed=2)
    ks, ko, cones = _solvers(hip, oracle, pr)
    rng = np.random.default_rng(8)
    scal = [(pr["s"] * (1.0 + 0.4 * k), pr["z"] / (1.0 + 0.3 * k)) for k in range(4)]
    vals = []
    for s_, z_ in scal[:2]:
        assert ks.update_scaling(s_, z_) and ks.update()
        vals.append(ks.values().copy())
    changed = np.nonzero(vals[0] != vals[1])[0].astype(np.int64)  # (the Hs blocks and the sparse cones' u / v / D entries)
    assert 0 < len(changed) < ks.nnzK
    K = ks.kkt_matrix()
    maps = ks.maps()
    dsigns, dfull = maps["dsigns"], maps["diag_full"]
    N = ks.N
    mk = lambda: hip.HipDirectLDLSolver(hip.CscMatrix(N, N, K.colptr, K.rowval, vals[1]), dsigns, hip.Settings.default(), perm=ks.perm)
    fast, plain = mk(), mk()
    id_upd = fast.register_index(changed)
    id_diag = fast.register_index(dfull, signs=dsigns)
    xbuf, bbuf = np.zeros(N), np.zeros(N)
    fast.pin_buffer(xbuf)
    fast.pin_buffer(bbuf)
    for it, (s_, z_) in enumerate(scal):
        assert ks.update_scaling(s_, z_) and ks.update()
        assert cones.update_scaling(s_, z_) and ko.update()
        v = ks.values()
        eps = float(ks.linear_solver_info().last_regularizer)
        b = np.concatenate([rng.standard_normal(pr["n"] + pr["m"]), np.zeros(N - pr["n"] - pr["m"])])
        okr, xr = ko.solve_full(b)
        assert okr
        # -- registered sets (iteration 2: the update through the PLAIN call on the same handle, index shipped)
        if it == 2:
            fast.update_values(changed, v[changed])
        else:
            fast.update_values_id(id_upd, v[changed])
        fast.offset_values_id(id_diag, eps)
        assert fast.refactor()
        fast.offset_values_id(id_diag, -eps)
        bbuf[:] = b
        ok, rounds = fast.solve_refined(xbuf, bbuf)
        assert ok and relerr(xbuf, xr) <= TOL, (it, rounds)
        # -- the same through the plain calls + the device-resident refinement
        plain.update_values(changed, v[changed])
        plain.offset_values(dfull, eps, dsigns)
        assert plain.refactor()
        plain.offset_values(dfull, -eps, dsigns)
        x2 = np.zeros(N)
        ok2, _ = plain.solve_refined(x2, b)
        assert ok2 and relerr(x2, xr) <= TOL
        # -- and the refinement's settings: one round fixed / none (the raw LDL' solve of the regularised system)
        st1 = hip.Settings.default(iterative_refinement_max_iter=1, iterative_refinement_reltol=0.0, iterative_refinement_abstol=0.0)
        ok3, r3 = fast.solve_refined(xbuf, bbuf, st1)
        assert ok3 and r3 == 1 and relerr(xbuf, xr) <= 1e-7
        x4 = np.zeros(N)
        fast.solve(None, x4, b)
        st0 = hip.Settings.default(iterative_refinement_enable=0)
        ok5, r5 = fast.solve_refined(xbuf, bbuf, st0)
        assert ok5 and r5 == 0 and relerr(xbuf, x4) <= 1e-13


@pytest.mark.parametrize("which", ["qp_supernodes", "chordal_sdp", "chordal_sdp_wide", "arrow", "forest", "general"])
def test_paired_solves_match_separate_solves(hip, oracle, which, monkeypatch):
    """chip_kkt_solve2_dev_enqueue: the two independent solves of an interior-point iteration (constant right-hand side,
    affine direction) as one call -- on level-scheduled systems enqueued on two streams and overlapped -- against the
    oracle's two solves and against two separate calls (bitwise on the deterministic kernels, to rounding where sweeps
    use atomics); three iterations, default refinement and the refinement switched off; CHIP_NO_SOLVE_PAIR: one after
    the other"""
    hs = None
    if which == "qp_supernodes":
        pr = problems.random_qp(20000, 40000, band=50, seed=1)
    elif which == "chordal_sdp":
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
        hs = pr["hsblocks"]
    elif which == "chordal_sdp_wide":
        # (cliques of 40: chain supernodes several hundred columns wide -- the pipelined substitution k_snode_tri, whose
        # paired form streams a panel ONCE for both right-hand sides: k_snode_tri<., 2>, asserted below)
        pr = problems.chordal_sdp(5, 40, 8, 5, 21, seed=5)
        hs = pr["hsblocks"]
    elif which == "arrow":
        pr = problems.portfolio_socp(12, 300, seed=3)
    elif which == "forest":
        pr = problems.blockdiag([problems.portfolio_socp(2 + (i % 3), 120 + 40 * (i % 4), seed=100 + i) for i in range(14)])
    else:
        pr = problems.mixed_conic(nexp=20, npow=10, nsoc=3, socdim=9, nn=30, seed=11)
    n, m = pr["n"], pr["m"]
    # (CHIP_NO_PAIR_LOCKSTEP: the pair as two independent chains of launches on two streams, no two-vector launches)
    for env, st in ((None, None), ("CHIP_NO_SOLVE_PAIR", None), ("CHIP_NO_PAIR_LOCKSTEP", None), (None, hip.Settings.default(iterative_refinement_enable=0))):
        if env:
            monkeypatch.setenv(env, "1")
        ks, ko, cones = _solvers(hip, oracle, pr, settings=st)
        if env:
            monkeypatch.delenv(env)
        rng = np.random.default_rng(4)
        outs = [hip.DeviceArray(n + m) for _ in range(4)]
        for it in range(3):
            # (host-supplied Hs blocks belong to the problem's own (s, z): that instance keeps its scaling point)
            s_, z_ = (pr["s"], pr["z"]) if hs is not None else (pr["s"] * (1.0 + 0.3 * it), pr["z"] / (1.0 + 0.2 * it))
            assert ks.update_scaling(s_, z_) and ks.update(hs)
            assert cones.update_scaling(s_, z_) and ko.update(hs)
            assert relerr(ks.values(), ko.kkt.nzval) <= 1e-13
            rhs = [(rng.standard_normal(n), rng.standard_normal(m)) for _ in range(2)]
            dev = [(hip.DeviceArray(a), hip.DeviceArray(b)) for a, b in rhs]
            ks.solve2_dev_enqueue(dev[0][0].ptr, dev[0][1].ptr, outs[0].ptr, outs[0].ptr + 8 * n,
                                  dev[1][0].ptr, dev[1][1].ptr, outs[1].ptr, outs[1].ptr + 8 * n)
            uok, sok = ks.collect()
            assert uok and sok == [True, True]
            if which == "chordal_sdp_wide":  # the two-right-hand-side launches ran exactly when the pair is allowed to share them
                assert (hip.debug_counter(ks, "tri2_launches") > 0) == (env is None), (env, hip.debug_counter(ks, "tri2_launches"))
                # ... and the residuals' products with the dense diagonal blocks of the top were one launch per pair (k_dblk_symv<2>)
                assert hip.debug_counter(ks, "dblk_blocks") > 0
                assert (hip.debug_counter(ks, "dblk2_launches") > 0) == (env is None and st is None)
            for k in range(2):  # the same two solves as separate calls on the same handle
                ks.setrhs_dev(dev[k][0].ptr, dev[k][1].ptr)
                ks.solve_dev_enqueue(outs[2 + k].ptr, outs[2 + k].ptr + 8 * n)
            uok, sok = ks.collect()
            assert uok and sok == [True, True]
            for k in range(2):
                ko.setrhs(*rhs[k])
                ok, xo, zo = ko.solve()
                assert ok
                ref = np.concatenate([xo, zo])
                tol = TOL if st is None else 1e-5
                assert relerr(outs[k].numpy(), ref) <= tol, (which, env, it, k)
                assert relerr(outs[k].numpy(), outs[2 + k].numpy()) <= (1e-10 if st is None else 1e-7)


def test_ir_fixed_one_round(hip, oracle):
    """the benchmark's refinement setting: max_iter=1, tolerances 0 => exactly one extra
    round, same on both sides"""
    st = hip.Settings.default(iterative_refinement_max_iter=1, iterative_refinement_reltol=0.0,
                              iterative_refinement_abstol=0.0)
    ks, ko = _check_update_and_solve(hip, oracle, problems.portfolio_socp(6, 100, seed=5), settings=st, tol=1e-7)
    assert ks.linear_solver_info().last_ir_iterations == 1 and ko.last_ir_iters == 1


@pytest.mark.parametrize("case", ["default", "loose_tol", "three_rounds", "no_refinement", "aliased", "flags0", "old_kernel"])
def test_fused_solve_iterates_on_chip(hip, oracle, case, monkeypatch):
    """k_bundle_irs (round 6: one bundle per workgroup, the iterates on chip) through every way it can end: the default
    settings; a tolerance that round 0 already meets (the verdict arrives at the middle barrier of a speculative round 1 and
    the result is x0 FROM THE REGISTERS); three forced rounds (candidates that later rounds build on go through xa / xb, x0
    leaves the registers at round 1); refinement off (one round, the candidate straight to the result vectors); result
    vectors that ALIAS the right-hand side (no write ahead of the verdict); the kernel's experiment bits off
    (CHIP_IRS_FLAGS=0); and the same problems on k_bundle_ir (CHIP_NO_IR_SF) -- all against the oracle with the same
    settings, device vectors in and out, and the refinement rounds taken must equal the oracle's"""
    pr = problems.portfolio_socp(6, 700, seed=11)
    kw = {}
    if case == "loose_tol":
        kw = dict(iterative_refinement_reltol=1e-3, iterative_refinement_abstol=1e-3)
    elif case == "three_rounds":
        kw = dict(iterative_refinement_max_iter=3, iterative_refinement_reltol=0.0, iterative_refinement_abstol=0.0,
                  iterative_refinement_stop_ratio=0.0)
    elif case == "no_refinement":
        kw = dict(iterative_refinement_enable=0)
    if case == "flags0":
        monkeypatch.setenv("CHIP_IRS_FLAGS", "0")
    if case == "old_kernel":
        monkeypatch.setenv("CHIP_NO_IR_SF", "1")
    st = hip.Settings.default(**kw) if kw else None
    ks, ko, cones = _solvers(hip, oracle, pr, settings=st)
    assert bool(ks.step_kernels() & 4) == (case != "old_kernel")
    n, m = pr["n"], pr["m"]
    rng = np.random.default_rng(3)
    for it in range(2):
        s_, z_ = pr["s"] * (1.0 + 0.3 * it), pr["z"] / (1.0 + 0.2 * it)
        assert ks.update_scaling(s_, z_) and ks.update() and cones.update_scaling(s_, z_) and ko.update()
        rx, rz = rng.standard_normal(n), rng.standard_normal(m)
        ko.setrhs(rx, rz)
        ok, xo, zo = ko.solve()
        assert ok
        ref = np.concatenate([xo, zo])
        if case == "aliased":  # one buffer for (rhsx, rhsz) and (lhsx, lhsz)
            buf = hip.DeviceArray(np.concatenate([rx, rz]))
            ks.setrhs_dev(buf.ptr, buf.ptr + 8 * n)
            assert ks.solve_dev(buf.ptr, buf.ptr + 8 * n)
            got = buf.numpy()
        else:
            d_rx, d_rz, out = hip.DeviceArray(rx), hip.DeviceArray(rz), hip.DeviceArray(n + m)
            ks.setrhs_dev(d_rx.ptr, d_rz.ptr)
            assert ks.solve_dev(out.ptr, out.ptr + 8 * n)
            got = out.numpy()
            assert np.array_equal(d_rx.numpy(), rx) and np.array_equal(d_rz.numpy(), rz)  # (the lent vectors are only read)
        tol = 1e-5 if case in ("no_refinement", "loose_tol") else TOL
        assert relerr(got, ref) <= tol, (case, it, relerr(got, ref))
        assert ks.linear_solver_info().last_ir_iterations == ko.last_ir_iters, (case, ks.linear_solver_info().last_ir_iterations, ko.last_ir_iters)
        if case == "three_rounds":
            assert ko.last_ir_iters == 3
        if case == "loose_tol":
            assert ko.last_ir_iters == 0
    assert ks.fused_fallbacks() == 0


@pytest.mark.parametrize("which", ["arrow", "forest"])
def test_repeated_solve_on_fused_handle(hip, oracle, which):
    """solve() twice on the same setrhs() (the reference keeps self.b, directldlkktsolver.rs:168-175), and
    setrhs() followed by solve_full() with no solve in between, on handles whose solve is the fused persistent
    launch (k_bundle_ir keeps the folded top's right-hand side in LDS: bp must still end up complete)"""
    pr = problems.portfolio_socp(12, 300, seed=3) if which == "arrow" else problems.batched_socp(16, 200, 2, seed=100)
    ks, ko, cones = _solvers(hip, oracle, pr)
    assert ks.update_scaling(pr["s"], pr["z"]) and cones.update_scaling(pr["s"], pr["z"])
    assert ks.update() and ko.update()
    rng = np.random.default_rng(7)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    ks.setrhs(rx, rz)
    ko.setrhs(rx, rz)
    ok, xo, zo = ko.solve()
    assert ok
    ref = np.concatenate([xo, zo])
    x1, z1 = np.zeros(pr["n"]), np.zeros(pr["m"])
    x2, z2 = np.zeros(pr["n"]), np.zeros(pr["m"])
    assert ks.solve(x1, z1)
    assert ks.solve(x2, z2)  # same right-hand side again, no setrhs in between
    assert relerr(np.concatenate([x1, z1]), ref) <= TOL
    assert relerr(np.concatenate([x2, z2]), ref) <= TOL
    # setrhs (noted, borrowed pointers) then solve_full with a different full-N right-hand side
    b = rng.standard_normal(ks.N)
    ks.setrhs(rx, rz)
    okf, xf = ks.solve_full(b)
    okr, xr = ko.solve_full(b)
    assert okf and okr
    assert relerr(xf, xr) <= TOL
    # and the handle still solves the next ordinary right-hand side
    ks.setrhs(rz[:pr["n"]] if pr["m"] >= pr["n"] else rx, rz)
    ko.setrhs(rz[:pr["n"]] if pr["m"] >= pr["n"] else rx, rz)
    ok, xo, zo = ko.solve()
    assert ks.solve(x1, z1) and ok
    assert relerr(np.concatenate([x1, z1]), np.concatenate([xo, zo])) <= TOL


def _enqueue_three(hip, ks, pr, rng):
    rhs = [(hip.DeviceArray(rng.standard_normal(pr["n"])), hip.DeviceArray(rng.standard_normal(pr["m"]))) for _ in range(3)]
    lhs = [hip.DeviceArray(pr["n"] + pr["m"]) for _ in range(3)]
    for (rx, rz), l in zip(rhs, lhs):
        ks.setrhs_dev(rx.ptr, rz.ptr)
        ks.solve_dev_enqueue(l.ptr, l.ptr + 8 * pr["n"])
    return rhs, lhs


def test_fused_solve_with_coresident_kernel(hip, oracle):
    """The sharded path lets RCCL's ring kernels run on a second stream UNDER the next step's persistent
    k_bundle_ir launches, whose grid was sized for an idle device.  A co-resident kernel that holds CU slots for a
    while (32 / 256 workgroups x 1024 threads with 64 KB of LDS each, 3 ms) must only delay the grid barrier, never
    break it: a config-4-share handle (a forest of 6007-node trees, one workgroup per tree) solves three enqueued
    right-hand sides correctly while the spinner is resident, without taking the fallback path."""
    pr = problems.batched_socp(128, 2000, 2, seed=100)
    ks, ko, cones = _solvers(hip, oracle, pr)
    assert ks.update_scaling(pr["s"], pr["z"]) and cones.update_scaling(pr["s"], pr["z"])
    assert ks.update() and ko.update()
    rng = np.random.default_rng(11)
    for spin_blocks in (32, 256):
        hip.debug_spin(0, spin_blocks, 1024, 65536, 3000.0)
        rhs, lhs = _enqueue_three(hip, ks, pr, rng)
        uok, sok = ks.collect()
        assert uok and list(sok) == [True, True, True]
        hip.debug_spin(0, 0)
        for (rx, rz), l in zip(rhs, lhs):
            ko.setrhs(rx.numpy(), rz.numpy())
            ok, xo, zo = ko.solve()
            assert ok and relerr(l.numpy(), np.concatenate([xo, zo])) <= TOL
    assert ks.fused_fallbacks() == 0


def test_fused_solve_timeout_falls_back(hip, oracle, monkeypatch):
    """a fused launch whose workgroups are NOT all resident (forced: CHIP_IR_TEST_DROP makes the last workgroup
    leave at once, so every grid barrier times out after ~2 s): the solve is repeated on the one-kernel-per-phase
    path -- synchronous solve and enqueue / collect alike -- instead of reporting a failure"""
    monkeypatch.setenv("CHIP_IR_TEST_DROP", "1")
    pr = problems.portfolio_socp(12, 300, seed=3)
    ks, ko, cones = _solvers(hip, oracle, pr)
    assert ks.update_scaling(pr["s"], pr["z"]) and cones.update_scaling(pr["s"], pr["z"])
    assert ks.update() and ko.update()
    rng = np.random.default_rng(12)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    ks.setrhs(rx, rz)
    ko.setrhs(rx, rz)
    x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
    assert ks.solve(x, z)
    ok, xo, zo = ko.solve()
    assert ok and relerr(np.concatenate([x, z]), np.concatenate([xo, zo])) <= TOL
    assert ks.fused_fallbacks() == 1
    rhs, lhs = _enqueue_three(hip, ks, pr, rng)
    uok, sok = ks.collect()
    assert uok and list(sok) == [True, True, True]
    # every one of them is reported as repeated at collect time: consumers of its lhs must be re-issued
    assert ks.repeated_solves == [0, 1, 2]
    for (drx, drz), l in zip(rhs, lhs):
        ko.setrhs(drx.numpy(), drz.numpy())
        ok, xo, zo = ko.solve()
        assert ok and relerr(l.numpy(), np.concatenate([xo, zo])) <= TOL
    assert ks.fused_fallbacks() == 4


def test_soc_scaling_failure_reported(hip):
    pr = problems.portfolio_socp(2, 10, seed=1)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    s = pr["s"].copy()
    s[1 + pr["n"]] = 0.0  # first SOC: s0 = 0 -> not interior -> update_scaling false (socone.rs:149-151)
    assert ks.update_scaling(s, pr["z"]) is False


def test_mul_Hs_identity(hip, oracle):
    """Hs z = s for the NT scaling, on the device (compositecone.rs:259-264)"""
    pr = problems.portfolio_socp(5, 40, seed=8)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    assert ks.update_scaling(pr["s"], pr["z"])
    zt = hip.DeviceArray(pr["z"])
    yt = hip.DeviceArray(pr["m"])
    ks.mul_Hs_dev(yt.ptr, zt.ptr)
    ks.synchronize()
    assert relerr(yt.numpy(), pr["s"]) <= 1e-10
    cones = oracle.Cones(pr["cones"])
    cones.update_scaling(pr["s"], pr["z"])
    assert relerr(yt.numpy(), cones.mul_Hs(pr["z"])) <= 1e-12


@pytest.mark.parametrize("late", [False, True])
def test_cone_step_operations(hip, oracle, late):
    """affine_ds / combined_ds_shift / ds_from_dz_offset / step_length / margins of the composite
    cone (Zero + Nonnegative + sparse and dense SecondOrder cones) on the device vs the oracle"""
    pr = problems.portfolio_socp(6, 257, seed=13, late=late)
    extra = problems.portfolio_socp(3, 3, seed=14)  # SOC(4): dense-form cones
    pr = problems.blockdiag([pr, extra])
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    cones = oracle.Cones(pr["cones"])
    s, z = pr["s"], pr["z"]
    assert ks.update_scaling(s, z) and cones.update_scaling(s, z)
    m = pr["m"]
    rng = np.random.default_rng(5)
    dz, dsv = 0.3 * rng.standard_normal(m), 0.3 * rng.standard_normal(m)
    D = hip.DeviceArray
    # affine_ds
    out, d_s, d_z, d_ds = D(m), D(s), D(z), D(dsv)  # keep references: .ptr of a temporary dangles
    ks.affine_ds_dev(out.ptr, d_s.ptr)
    ks.synchronize()
    assert relerr(out.numpy(), cones.affine_ds(s)) <= 1e-13
    # combined_ds_shift (steps overwritten by W dz, W^-1 ds)
    sh, tz, ts = D(m), D(dz), D(dsv)
    ks.combined_ds_shift_dev(sh.ptr, tz.ptr, ts.ptr, 0.37)
    ks.synchronize()
    osh, oz, os_ = cones.combined_ds_shift(dz, dsv, 0.37)
    assert relerr(tz.numpy(), oz) <= 1e-12 and relerr(ts.numpy(), os_) <= 1e-12
    assert relerr(sh.numpy(), osh) <= 1e-11
    # ds_from_dz_offset
    o2 = D(m)
    ks.ds_from_dz_offset_dev(o2.ptr, d_ds.ptr, d_z.ptr)
    ks.synchronize()
    assert relerr(o2.numpy(), cones.ds_from_dz_offset(dsv, z)) <= 1e-11
    # step_length: several directions incl. ones that leave the cones quickly
    for scale in (0.05, 1.0, 30.0):
        t1, t2 = D(scale * dz), D(scale * dsv)
        a_dev = ks.step_length_dev(t1.ptr, t2.ptr, d_z.ptr, d_s.ptr, 1.0)
        a_ref = cones.step_length(scale * dz, scale * dsv, z, s, 1.0)
        assert abs(a_dev - a_ref) <= 1e-12 * max(1.0, a_ref)
    # margins
    zz = z + 0.5 * rng.standard_normal(m)
    d_zz = D(zz)
    a_dev, b_dev = ks.margins_dev(d_zz.ptr)
    a_ref, b_ref = cones.margins(zz)
    assert abs(a_dev - a_ref) <= 1e-12 * max(1.0, abs(a_ref)) and abs(b_dev - b_ref) <= 1e-10 * max(1.0, b_ref)


@pytest.mark.parametrize("strategy", [0, 1])
def test_nonsymmetric_cone_step_operations(hip, oracle, strategy):
    """Exponential / Power cones mixed with Zero / NN / SOC: affine_ds, combined_ds_shift (3rd-order
    correction, expcone.rs:254-308 / powcone.rs:260-337), ds_from_dz_offset, step_length (backtracking
    after the symmetric cones, compositecone.rs:300-340), compute_barrier, unit_initialization"""
    pr = problems.mixed_conic(nexp=300, npow=300, seed=21)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    cones = oracle.Cones(pr["cones"])
    s, z, m = pr["s"], pr["z"], pr["m"]
    assert ks.update_scaling(s, z, 0.31, strategy) and cones.update_scaling(s, z, 0.31, strategy)
    rng = np.random.default_rng(6)
    dz, dsv = 0.05 * rng.standard_normal(m), 0.05 * rng.standard_normal(m)
    D = hip.DeviceArray
    out, d_s, d_z, d_ds, d_dz = D(m), D(s), D(z), D(dsv), D(dz)
    ks.affine_ds_dev(out.ptr, d_s.ptr)
    assert relerr(out.numpy(), cones.affine_ds(s)) <= 1e-13
    sh, tz, ts = D(m), D(dz), D(dsv)
    ks.combined_ds_shift_dev(sh.ptr, tz.ptr, ts.ptr, 0.37)
    osh, oz, os_ = cones.combined_ds_shift(dz, dsv, 0.37)
    assert relerr(tz.numpy(), oz) <= 1e-12 and relerr(ts.numpy(), os_) <= 1e-12
    assert relerr(sh.numpy(), osh) <= 1e-9
    o2 = D(m)
    ks.ds_from_dz_offset_dev(o2.ptr, d_ds.ptr, d_z.ptr)
    assert relerr(o2.numpy(), cones.ds_from_dz_offset(dsv, z)) <= 1e-11
    for scale in (0.2, 3.0, 60.0):
        t1, t2 = D(scale * dz), D(scale * dsv)
        a_dev = ks.step_length_dev(t1.ptr, t2.ptr, d_z.ptr, d_s.ptr, 1.0)
        a_ref = cones.step_length(scale * dz, scale * dsv, z, s, 1.0)
        assert abs(a_dev - a_ref) <= 1e-12 * max(1.0, a_ref)
        assert a_dev < 1.0
    for alpha in (0.0, 0.4):
        b_dev = ks.compute_barrier_dev(d_z.ptr, d_s.ptr, d_dz.ptr, d_ds.ptr, alpha)
        b_ref = cones.compute_barrier(z, s, dz, dsv, alpha)
        assert np.isfinite(b_ref) and abs(b_dev - b_ref) <= 1e-10 * max(1.0, abs(b_ref))
    uz, us = D(m), D(m)
    ks.unit_initialization_dev(uz.ptr, us.ptr)
    rz, rs = np.zeros(m), np.zeros(m)
    cones.unit_initialization(rz, rs)
    assert np.array_equal(uz.numpy(), rz) and np.array_equal(us.numpy(), rs)


def test_psd_large_cone_hs_on_device(hip):
    """PSDTriangleCone(128): update_scaling (two Cholesky factors, Jacobi SVD, R R') and Hs = skron(R R') on the
    device with the work matrices in HBM scratch -- the 3.4e7 Hs entries written into K against the numpy
    restatement, and the identity Hs z = s through mul_Hs (the oracle's scalar LDL' of an 8256-wide dense
    front would take minutes, so no solve here: test_c5_psd_scaling_on_device[72] has it)"""
    dim = 128
    pr = problems.chordal_sdp(1, dim, 3, 1, 7, seed=dim, with_hs=False)
    ks = hip.HipKKTSolver(hip.CscMatrix(pr["n"], pr["n"], *pr["P"]), hip.CscMatrix(pr["m"], pr["n"], *pr["A"]),
                          pr["cones"], pr["m"], pr["n"])
    assert ks.update_scaling(pr["s"], pr["z"]) and ks.update()
    numel = dim * (dim + 1) // 2
    S = np.zeros((dim, dim))
    Z = np.zeros((dim, dim))
    r, c = np.tril_indices(dim)
    w = np.where(r == c, 1.0, 1.0 / np.sqrt(2.0))
    S[r, c] = S[c, r] = pr["s"][:numel] * w
    Z[r, c] = Z[c, r] = pr["z"][:numel] * w
    hs_ref = problems.psd_scaling_Hs(S, Z)
    mapHs = ks.maps()["Hsblocks"][:len(hs_ref)]
    assert relerr(-ks.values()[mapHs], hs_ref) <= 1e-10
    y, d_z = hip.DeviceArray(pr["m"]), hip.DeviceArray(pr["z"])
    ks.mul_Hs_dev(y.ptr, d_z.ptr)
    ks.synchronize()
    assert relerr(y.numpy()[:numel], pr["s"][:numel]) <= 1e-9  # NT scaling: Hs z = s


@pytest.mark.parametrize("dim", [3, 8, 21, 50, 96])
def test_psd_cone_operations(hip, oracle, dim):
    """PSDTriangleCone operations either side of the solve on the device (one workgroup per cone:
    GEMMs with R / Rinv, Jacobi eigenvalues, Cholesky log-det) against the numpy restatement
    oracle/psd_numpy.py, in a composite with Nonnegative and SecondOrder cones"""
    from oracle import psd_numpy
    pr = problems.chordal_sdp(1 if dim >= 50 else 3, dim, min(3, dim - 1), 2, 7, seed=40 + dim, with_hs=dim < 64)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    cones = psd_numpy.MixedCones(oracle, pr["cones"])
    s, z, m = pr["s"], pr["z"], pr["m"]
    assert ks.update_scaling(s, z) and cones.update_scaling(s, z)
    rng = np.random.default_rng(dim)
    # a symmetric-cone-friendly direction: small relative to the interior point
    dz, dsv = 0.1 * rng.standard_normal(m), 0.1 * rng.standard_normal(m)
    D = hip.DeviceArray
    d_s, d_z, d_ds, d_dz = D(s), D(z), D(dsv), D(dz)
    x = rng.standard_normal(m)
    y, d_x = D(m), D(x)
    ks.mul_Hs_dev(y.ptr, d_x.ptr)
    assert relerr(y.numpy(), cones.mul_Hs(x)) <= 1e-11
    out = D(m)
    ks.affine_ds_dev(out.ptr, d_s.ptr)
    assert relerr(out.numpy(), cones.affine_ds(s)) <= 1e-11
    sh, tz, ts = D(m), D(dz), D(dsv)
    ks.combined_ds_shift_dev(sh.ptr, tz.ptr, ts.ptr, 0.37)
    osh, oz, os_ = cones.combined_ds_shift(dz, dsv, 0.37)
    # scaled-space vectors: comparable thanks to the shared SVD conventions (descending, signed)
    assert relerr(tz.numpy(), oz) <= 1e-9 and relerr(ts.numpy(), os_) <= 1e-9
    assert relerr(sh.numpy(), osh) <= 1e-9
    # the invariant chain: (affine_ds + shift) mapped back by ds_from_dz_offset
    rhs_s = cones.affine_ds(s) + osh
    o2, d_rs = D(m), D(rhs_s)
    ks.ds_from_dz_offset_dev(o2.ptr, d_rs.ptr, d_z.ptr)
    assert relerr(o2.numpy(), cones.ds_from_dz_offset(rhs_s, z)) <= 1e-9
    for scale in (0.05, 1.0, 40.0):
        t1, t2 = D(scale * dz), D(scale * dsv)
        a_dev = ks.step_length_dev(t1.ptr, t2.ptr, d_z.ptr, d_s.ptr, 1.0)
        a_ref = cones.step_length(scale * dz, scale * dsv, z, s, 1.0)
        assert abs(a_dev - a_ref) <= 1e-10 * max(1.0, a_ref)
    zz = z + 0.7 * rng.standard_normal(m)
    d_zz = D(zz)
    a_dev, b_dev = ks.margins_dev(d_zz.ptr)
    a_ref, b_ref = cones.margins(zz)
    assert abs(a_dev - a_ref) <= 1e-10 * max(1.0, abs(a_ref)) and abs(b_dev - b_ref) <= 1e-10 * max(1.0, b_ref)
    for alpha in (0.0, 0.3):
        b_dev = ks.compute_barrier_dev(d_z.ptr, d_s.ptr, d_dz.ptr, d_ds.ptr, alpha)
        b_ref = cones.compute_barrier(z, s, dz, dsv, alpha)
        assert np.isfinite(b_ref) and abs(b_dev - b_ref) <= 1e-10 * max(1.0, abs(b_ref))
    sh2 = D(zz)
    ks.scaled_unit_shift_dev(sh2.ptr, 0.25, True)
    ref = zz.copy()
    cones.scaled_unit_shift(ref, 0.25, True)
    assert np.array_equal(sh2.numpy(), ref)
    uz, us = D(m), D(m)
    ks.unit_initialization_dev(uz.ptr, us.ptr)
    rz, rs = np.zeros(m), np.zeros(m)
    cones.unit_initialization(rz, rs)
    assert np.array_equal(uz.numpy(), rz) and np.array_equal(us.numpy(), rs)


def test_genpow_cone_on_device(hip, oracle):
    """GenPowerCone (genpowcone.rs): dual scaling, Hs diagonal + the [q, r, p] sparse expansion written
    into K (datamaps.rs:322-343), mul_Hs, step operations, barrier; vs the oracle, plus a KKT solve"""
    rng = np.random.default_rng(77)
    cones, s_parts, z_parts = [(1, 5)], [rng.uniform(0.5, 2, 5)], [rng.uniform(0.5, 2, 5)]
    for d1, d2 in ((2, 1), (3, 2), (6, 4), (40, 25)):
        a = rng.uniform(0.2, 1.0, d1)
        a /= a.sum()
        a[-1] = 1.0 - a[:-1].sum()
        cones.append((5, d1, d2, list(a)))
        wdir = rng.standard_normal(d2)
        for parts in (s_parts, z_parts):  # interior of K and K*: u > 0, ||w|| well below prod u^alpha
            u = rng.uniform(0.8, 2.0, d1)
            # same w direction in s and z: the reference's primal gradient scales its w part with the
            # cone's r vector (built from z, genpowcone.rs:427), so only then is the barrier finite
            w = wdir * (0.3 * np.prod((u / a) ** a) / max(np.linalg.norm(wdir), 1e-9))
            parts.append(np.concatenate([u, w]))
    s, z = np.concatenate(s_parts), np.concatenate(z_parts)
    m = len(s)
    n = 12
    import scipy.sparse as sp2
    A = sp2.random(m, n, density=0.3, random_state=np.random.RandomState(3), format="csc") + \
        sp2.csc_matrix((np.ones(n), (np.arange(n), np.arange(n))), shape=(m, n))
    P = sp2.diags(rng.uniform(0.5, 1.5, n)).tocsc()
    pr = dict(n=n, m=m, P=problems._csc(P), A=problems._csc(A), cones=cones, s=s, z=z)
    ks, ko, oc = _solvers(hip, oracle, pr)
    mu = 0.43
    assert ks.update_scaling(s, z, mu, 1) and oc.update_scaling(s, z, mu, 1)
    assert ks.update() and ko.update()
    assert relerr(ks.values(), ko.kkt.nzval) <= 1e-12
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    ks.setrhs(rx, rz)
    ko.setrhs(rx, rz)
    x, zz = np.zeros(n), np.zeros(m)
    assert ks.solve(x, zz)
    ok, xo, zo = ko.solve()
    assert ok and relerr(np.concatenate([x, zz]), np.concatenate([xo, zo])) <= TOL
    D = hip.DeviceArray
    v = rng.standard_normal(m)
    y, d_v = D(m), D(v)
    ks.mul_Hs_dev(y.ptr, d_v.ptr)
    assert relerr(y.numpy(), oc.mul_Hs(v)) <= 1e-12
    dz, dsv = 0.05 * rng.standard_normal(m), 0.05 * rng.standard_normal(m)
    d_s, d_z, d_ds, d_dz = D(s), D(z), D(dsv), D(dz)
    out = D(m)
    ks.affine_ds_dev(out.ptr, d_s.ptr)
    assert relerr(out.numpy(), oc.affine_ds(s)) <= 1e-13
    sh, tz, ts = D(m), D(dz), D(dsv)
    ks.combined_ds_shift_dev(sh.ptr, tz.ptr, ts.ptr, 0.37)
    osh, _, _ = oc.combined_ds_shift(dz, dsv, 0.37)
    assert relerr(sh.numpy(), osh) <= 1e-12
    for scale in (0.5, 8.0, 100.0):
        t1, t2 = D(scale * dz), D(scale * dsv)
        a_dev = ks.step_length_dev(t1.ptr, t2.ptr, d_z.ptr, d_s.ptr, 1.0)
        a_ref = oc.step_length(scale * dz, scale * dsv, z, s, 1.0)
        assert abs(a_dev - a_ref) <= 1e-12 * max(1.0, a_ref)
    zero = D(m)
    b_dev = ks.compute_barrier_dev(d_z.ptr, d_s.ptr, zero.ptr, zero.ptr, 0.0)
    b_ref = oc.compute_barrier(z, s, np.zeros(m), np.zeros(m), 0.0)
    # NB with the reference's primal gradient (w part scaled by the cone's r vector, genpowcone.rs:427)
    # -g(s) is generally outside the dual cone, so the primal barrier is -inf on both sides
    assert b_dev == b_ref or abs(b_dev - b_ref) <= 1e-9 * max(1.0, abs(b_ref))
    b_dev = ks.compute_barrier_dev(d_z.ptr, d_s.ptr, d_dz.ptr, d_ds.ptr, 0.3)
    b_ref = oc.compute_barrier(z, s, dz, dsv, 0.3)
    assert b_dev == b_ref or abs(b_dev - b_ref) <= 1e-9 * max(1.0, abs(b_ref))
    uz, us = D(m), D(m)
    ks.unit_initialization_dev(uz.ptr, us.ptr)
    rz2, rs2 = np.zeros(m), np.zeros(m)
    oc.unit_initialization(rz2, rs2)
    assert relerr(uz.numpy(), rz2) <= 1e-15 and relerr(us.numpy(), rs2) <= 1e-15


def test_genpow_barrier_finite_case(hip, oracle):
    """GenPowerCone with dim2 = 0 (pure power part): the primal gradient has no w part, -g(s) is dual
    feasible and the barrier is finite -- checks both barrier halves against the oracle"""
    rng = np.random.default_rng(8)
    cones = [(5, 3, 0, [0.2, 0.3, 0.5]), (5, 2, 0, [0.6, 0.4]), (1, 2)]
    m, n = 7, 3
    s, z = rng.uniform(0.5, 2.0, m), rng.uniform(0.5, 2.0, m)
    import scipy.sparse as sp2
    A = sp2.csc_matrix(rng.standard_normal((m, n)))
    P = sp2.identity(n, format="csc")
    pr = dict(n=n, m=m, P=problems._csc(P), A=problems._csc(A), cones=cones, s=s, z=z)
    ks, ko, oc = _solvers(hip, oracle, pr)
    assert ks.update_scaling(s, z, 0.7, 1) and oc.update_scaling(s, z, 0.7, 1)
    assert ks.update() and ko.update()
    assert relerr(ks.values(), ko.kkt.nzval) <= 1e-12
    D = hip.DeviceArray
    dz, dsv = 0.1 * rng.standard_normal(m), 0.1 * rng.standard_normal(m)
    d_s, d_z, d_ds, d_dz = D(s), D(z), D(dsv), D(dz)
    for alpha in (0.0, 0.5):
        b_dev = ks.compute_barrier_dev(d_z.ptr, d_s.ptr, d_dz.ptr, d_ds.ptr, alpha)
        b_ref = oc.compute_barrier(z, s, dz, dsv, alpha)
        assert np.isfinite(b_ref) and abs(b_dev - b_ref) <= 1e-10 * max(1.0, abs(b_ref))


def test_degenerate_cone_dimensions(hip, oracle):
    """zero-length and one-dimensional cones (tests/basic_sdp.rs test_sdp_empty_cone, NonnegativeConeT(0)):
    nothing to launch for them, everything else unaffected"""
    from oracle import psd_numpy
    cones = [(1, 0), (6, 0), (6, 1), (2, 2), (1, 3), (0, 0), (6, 2)]
    rng = np.random.default_rng(4)
    S2, Z2 = np.array([[2.0, 0.3], [0.3, 1.5]]), np.array([[1.2, -0.2], [-0.2, 0.9]])
    s = np.concatenate([[1.7], [2.0, 0.5], rng.uniform(0.5, 2, 3), psd_numpy.mat_to_svec(S2)])
    z = np.concatenate([[0.6], [1.5, -0.4], rng.uniform(0.5, 2, 3), psd_numpy.mat_to_svec(Z2)])
    m, n = len(s), 3
    import scipy.sparse as sp2
    A = sp2.csc_matrix(rng.standard_normal((m, n)))
    P = sp2.identity(n, format="csc")
    pr = dict(n=n, m=m, P=problems._csc(P), A=problems._csc(A), cones=cones)
    Pm, Am = hip.CscMatrix(n, n, *pr["P"]), hip.CscMatrix(m, n, *pr["A"])
    ks = hip.HipKKTSolver(Pm, Am, cones, m, n)
    mc = psd_numpy.MixedCones(oracle, cones)
    ko = oracle.KKTSolver(n, m, pr["P"], pr["A"], mc.c, perm=ks.perm)
    assert ks.update_scaling(s, z) and mc.update_scaling(s, z)
    assert ks.update() and ko.update(mc.get_Hs())
    assert relerr(ks.values(), ko.kkt.nzval) <= 1e-12
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    ks.setrhs(rx, rz)
    ko.setrhs(rx, rz)
    x, zz = np.zeros(n), np.zeros(m)
    assert ks.solve(x, zz)
    ok, xo, zo = ko.solve()
    assert ok and relerr(np.concatenate([x, zz]), np.concatenate([xo, zo])) <= TOL
    D = hip.DeviceArray
    v = rng.standard_normal(m)
    y, d_v = D(m), D(v)
    ks.mul_Hs_dev(y.ptr, d_v.ptr)
    assert relerr(y.numpy(), mc.mul_Hs(v)) <= 1e-12
    d_z = D(z)
    a_dev, b_dev = ks.margins_dev(d_z.ptr)
    a_ref, b_ref = mc.margins(z)
    assert abs(a_dev - a_ref) <= 1e-12 and abs(b_dev - b_ref) <= 1e-12 * max(1.0, b_ref)


def test_full_scale_properties_c3(hip):
    """BASELINE config 3 at full size (n = 10^6): too big for the oracle in seconds, so check
    size-independent properties: residual of the refined solution against an independent
    scipy SpMV of the UNregularised K, and linearity of the solve."""
    import scipy.sparse as sp
    pr = problems.portfolio_socp(1000, 1000, seed=3)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    assert ks.update_scaling(pr["s"], pr["z"])
    assert ks.update()
    K = ks.kkt_matrix()
    vals = ks.values()
    Ku = sp.csc_matrix((vals, K.rowval.astype(np.int64), K.colptr.astype(np.int64)), shape=(ks.N, ks.N))
    Kfull = Ku + sp.triu(Ku, 1).T
    rng = np.random.default_rng(1)
    b1, b2 = rng.standard_normal(ks.N), rng.standard_normal(ks.N)
    ok1, x1 = ks.solve_full(b1)
    ok2, x2 = ks.solve_full(b2)
    ok3, x3 = ks.solve_full(2.0 * b1 - 3.0 * b2)
    assert ok1 and ok2 and ok3
    for b, x in ((b1, x1), (b2, x2)):
        r = b - Kfull @ x
        assert np.max(np.abs(r)) <= 1e-8 * max(1.0, np.max(np.abs(b)))
    assert relerr(x3, 2.0 * x1 - 3.0 * x2) <= 1e-7
    info = ks.linear_solver_info()
    assert info.n == 3003001 and info.positive_inertia == pr["n"] + 1000


# ---- full BASELINE sizes against the oracle (north_star: "solution within 1e-8 relative of reference") ----
def _full_scale_parity(hip, oracle, pr, hs_dev=True, nrhs=3, kvals_tol=1e-13):
    """update (scaling + fused Hs + static regularisation + refactor) and `nrhs` solves with the DEFAULT
    refinement settings at a BASELINE.json size, post-refinement solution against the oracle with the same
    permutation; also the device copy of K.nzval and the static regulariser"""
    ks, ko, cones = _solvers(hip, oracle, pr)
    assert ks.update_scaling(pr["s"], pr["z"]) and cones.update_scaling(pr["s"], pr["z"])
    assert ks.update(None if hs_dev else pr.get("hsblocks"))
    assert ko.update(pr.get("hsblocks"))
    assert relerr(ks.values(), ko.kkt.nzval) <= kvals_tol
    assert abs(ks.linear_solver_info().last_regularizer - ko.regularizer) <= 1e-20 + 1e-12 * ko.regularizer
    rng = np.random.default_rng(1234)
    worst = 0.0
    for _ in range(nrhs):
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ks.setrhs(rx, rz)
        ko.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        ok, xo, zo = ko.solve()
        assert ok
        worst = max(worst, relerr(np.concatenate([x, z]), np.concatenate([xo, zo])))
    assert worst <= TOL, "rel. err vs oracle %g" % worst
    return ks, ko


def test_full_scale_parity_c3(hip, oracle):
    """BASELINE config 3 (the bench workload) at its full size, n = 10^6, N = 3 003 001"""
    ks, ko = _full_scale_parity(hip, oracle, problems.portfolio_socp(1000, 1000, seed=3))
    assert ks.N == 3003001
    assert ks.linear_solver_info().regularize_count == ko.ldl_regularize_count()


def test_full_scale_parity_c2(hip, oracle):
    """BASELINE config 2 at its full size: random sparse QP n = 10^5, m = 2 x 10^5 (chain supernodes)"""
    ks, _ = _full_scale_parity(hip, oracle, problems.random_qp(100000, 200000, band=50, seed=1), nrhs=2)
    assert ks.N == 300000


def test_full_scale_parity_c4(hip, oracle):
    """BASELINE config 4 at its full size: 1024 independent SOCPs of n = 2000 (a forest of 1024 trees)"""
    ks, _ = _full_scale_parity(hip, oracle, problems.batched_socp(1024, 2000, 2, seed=100), nrhs=2)
    assert ks.N == 1024 * 6007 and ks.N == ks.NF


def test_full_scale_parity_c4_share_grouped_fold(hip, oracle):
    """one GPU's share of BASELINE config 4 at N = 8 (128 of the 1024 trees) at FULL size: the forest is cut into several
    bundles per tree and every tree's top (its last few columns) is folded into the kernels of ITS bundles -- grouped
    fold -- and both the factorisation and the solve run as the register-resident step kernels (k_gstep_factor,
    k_gstep_solve): the kernels the 8-GPU bound of DESIGN.md section 7 rests on.  step_kernels() == 3 proves it is THOSE
    kernels this full-size parity holds for (not k_bundle_ir<.., true> + the three-launch factorisation, their fallback)."""
    ks, ko = _full_scale_parity(hip, oracle, problems.batched_socp(128, 2000, 2, seed=100), nrhs=2)
    wm = ks.work_model()
    assert wm["fold_groups"] == 128 and wm["n_bundles"] > 512 and wm["fused_threads"] == 256
    assert ks.step_kernels() == 3
    assert ks.fused_fallbacks() == 0
    assert ks.linear_solver_info().regularize_count == ko.ldl_regularize_count()
    # the one-kernel-per-phase path on the same handle (the fallback of the fused launch) sees an ordinary top
    rng = np.random.default_rng(3)
    b = rng.standard_normal(ks.N)
    okf, xf = ks.solve_full(b)
    okr, xr = ko.solve_full(b)
    assert okf and okr and relerr(xf, xr) <= TOL


@pytest.mark.parametrize("late", [False, True])
def test_grouped_fold_ragged_forest(hip, oracle, late, monkeypatch):
    """grouped fold on a forest of UNEQUAL trees (tops of different sizes, one tree small enough to stay whole),
    benign and late-iterate scalings, with and without refinement"""
    parts = [problems.portfolio_socp(2 + (i % 3), 120 + 40 * (i % 4), seed=100 + i, late=late) for i in range(14)]
    parts.append(problems.portfolio_socp(1, 20, seed=300, late=late))
    pr = problems.blockdiag(parts)
    ks, ko = _check_update_and_solve(hip, oracle, pr, nrhs=3)
    assert ks.work_model()["fold_groups"] >= 10
    assert ks.step_kernels() == 3  # (solve and factorisation of the grouped fold: the register-resident step kernels)
    st = hip.Settings.default(iterative_refinement_enable=0)
    _check_update_and_solve(hip, oracle, pr, nrhs=1, settings=st)
    st = hip.Settings.default(iterative_refinement_max_iter=1, iterative_refinement_reltol=0.0, iterative_refinement_abstol=0.0)
    _check_update_and_solve(hip, oracle, pr, nrhs=2, settings=st)
    st = hip.Settings.default(iterative_refinement_max_iter=0)
    _check_update_and_solve(hip, oracle, pr, nrhs=1, settings=st, tol=1e-5)
    # the same through k_bundle_ir's grouped form + the three-launch factorisation (the step kernels switched off)
    monkeypatch.setenv("CHIP_NO_STEP_KERNEL", "1")
    ks1, _ = _check_update_and_solve(hip, oracle, pr, nrhs=2)
    assert ks1.work_model()["fold_groups"] >= 10 and ks1.step_kernels() == 0
    monkeypatch.delenv("CHIP_NO_STEP_KERNEL")
    monkeypatch.setenv("CHIP_NO_GROUPFOLD", "1")
    ks2, _ = _check_update_and_solve(hip, oracle, pr, nrhs=1)
    assert ks2.work_model()["fold_groups"] == 0


@pytest.mark.parametrize("which", ["arrow", "forest", "qp", "expcone"])
def test_update_scaled_enqueue_matches_the_two_calls(hip, oracle, which):
    """chip_kkt_update_scaled_enqueue = cones.update_scaling + the KKT update as one enqueue (core/solver.rs:334-352): the
    fused cone launch and -- where the bundle factorisation takes over the preparation work (an arrow with one top
    column: config 3's shape; a grouped fold) -- the launch sequence without eps / scatter / pivot launches must give
    the same K values, the same static regulariser, the same pivots' bookkeeping and solutions as the two calls, twice
    in a row (the slotted maxima alternate between two sets), and match the oracle"""
    if which == "arrow":
        pr = problems.portfolio_socp(12, 300, seed=3)
    elif which == "forest":
        parts = [problems.portfolio_socp(2 + (i % 3), 120 + 40 * (i % 4), seed=100 + i) for i in range(14)]
        pr = problems.blockdiag(parts)
    elif which == "qp":
        pr = problems.random_qp(3000, 6000, band=20, seed=1)
    else:
        pr = problems.mixed_conic(nexp=20, npow=10, nsoc=3, socdim=9, nn=30, seed=11)  # (other cone kinds: the two calls as they are)
    ks, ko, cones = _solvers(hip, oracle, pr)
    ks2, _, _ = _solvers(hip, oracle, pr)
    rng = np.random.default_rng(9)
    s_d, z_d = hip.DeviceArray(pr["s"]), hip.DeviceArray(pr["z"])
    for rep in range(3):
        s = pr["s"] * (1.0 + 0.1 * rep)
        z = pr["z"] * (1.0 + 0.05 * rep)
        s_d.copy_from(s)
        z_d.copy_from(z)
        ks.update_scaled_enqueue(s_d.ptr, z_d.ptr)
        uok, sok = ks.collect()
        assert uok and sok == []
        assert ks2.update_scaling(s, z) and ks2.update()
        assert cones.update_scaling(s, z) and ko.update()
        assert relerr(ks.values(), ks2.values()) == 0.0
        assert relerr(ks.values(), ko.kkt.nzval) <= 1e-13
        i1, i2 = ks.linear_solver_info(), ks2.linear_solver_info()
        assert i1.last_regularizer == i2.last_regularizer
        assert abs(i1.last_regularizer - ko.regularizer) <= 1e-20 + 1e-12 * ko.regularizer
        assert i1.regularize_count == i2.regularize_count and i1.positive_inertia == i2.positive_inertia
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        x, zz, x2, z2 = np.zeros(pr["n"]), np.zeros(pr["m"]), np.zeros(pr["n"]), np.zeros(pr["m"])
        ks.setrhs(rx, rz)
        assert ks.solve(x, zz)
        ks2.setrhs(rx, rz)
        assert ks2.solve(x2, z2)
        ko.setrhs(rx, rz)
        ok, xo, zo = ko.solve()
        assert ok
        assert relerr(np.concatenate([x, zz]), np.concatenate([xo, zo])) <= TOL
        assert relerr(np.concatenate([x, zz]), np.concatenate([x2, z2])) <= 1e-10


@pytest.mark.parametrize("which", ["arrow", "forest"])
def test_fast_and_plain_refactor_alternate_on_one_handle(hip, oracle, which):
    """chip_kkt_update_scaled_enqueue (fast preparation: eps from the slotted maxima inside the factor kernel, the two
    sets of slots alternating) and the plain update_scaling + update calls MIXED on one handle: after an odd number of
    fast refactors the cone kernels write set 1, and the plain path must reduce (and clear) THAT set -- the static
    regulariser of every refactor equals the oracle's (directldlkktsolver.rs:324-329), whichever path ran before"""
    if which == "arrow":
        pr = problems.portfolio_socp(12, 300, seed=3)
    else:
        parts = [problems.portfolio_socp(2 + (i % 3), 120 + 40 * (i % 4), seed=100 + i) for i in range(14)]
        pr = problems.blockdiag(parts)
    ks, ko, cones = _solvers(hip, oracle, pr)
    s_d, z_d = hip.DeviceArray(pr["s"]), hip.DeviceArray(pr["z"])
    rng = np.random.default_rng(5)
    # fast, plain, fast, fast, plain, plain, fast: every transition, on both parities
    for rep, fast in enumerate([True, False, True, True, False, False, True]):
        s = pr["s"] * (1.0 + 0.3 * rep)  # (the diagonal maxima differ from refactor to refactor: a stale set would show)
        z = pr["z"] * (1.0 + 0.2 * rep) / (1.0 + 0.5 * (rep % 2))
        if fast:
            s_d.copy_from(s)
            z_d.copy_from(z)
            ks.update_scaled_enqueue(s_d.ptr, z_d.ptr)
            uok, sok = ks.collect()
            assert uok and sok == []
        else:
            assert ks.update_scaling(s, z) and ks.update()
        assert cones.update_scaling(s, z) and ko.update()
        info = ks.linear_solver_info()
        assert abs(info.last_regularizer - ko.regularizer) <= 1e-20 + 1e-12 * ko.regularizer, (rep, fast)
        assert info.regularize_count == ko.ldl_regularize_count()
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        x, zz = np.zeros(pr["n"]), np.zeros(pr["m"])
        ks.setrhs(rx, rz)
        assert ks.solve(x, zz)
        ko.setrhs(rx, rz)
        ok, xo, zo = ko.solve()
        assert ok and relerr(np.concatenate([x, zz]), np.concatenate([xo, zo])) <= TOL


def test_parity_c5_24_cliques(hip, oracle):
    """BASELINE config 5's shape with 24 cliques of PSD(50) + 24 sparse SOC(51) -- the largest instance the
    scalar oracle factors in well under a minute (the full 200-clique instance takes it ~10 minutes);
    PSD scalings and Hs = skron(R R') computed on the device, the oracle receives the numpy restatement's
    Hs blocks for the same (S, Z)"""
    pr = problems.chordal_sdp(24, 50, 10, 24, 51, seed=5)
    ks, _ = _full_scale_parity(hip, oracle, pr, hs_dev=True, nrhs=2, kvals_tol=1e-11)
    assert len(ks.supernodes()) >= 24


# ---- BASELINE config 5 at its FULL size (200 x PSD(50) + 200 x SOC(51), N = 277 345) ------------------------
# The scalar oracle needs ~10 CPU-minutes for this factorisation: its answers are the committed fixture
# tests/golden/c5_full_oracle.npz (generator: tests/golden/make_c5_fixture.py, same problem generator and seed).
C5_FIXTURE = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "c5_full_oracle.npz")


@pytest.fixture(scope="module")
def c5_full(hip):
    """the full-size handle, built once (host analysis ~17 s), scaled and factored on the device"""
    pr = problems.chordal_sdp(200, 50, 10, 200, 51, seed=5, with_hs=False)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    assert ks.update_scaling(pr["s"], pr["z"])
    assert ks.update()
    yield pr, ks
    del ks


def test_full_scale_parity_c5(hip, c5_full):
    """post-refinement solutions (default refinement settings) against the oracle's, <= 1e-8 relative; the device's
    fused Hs write (PSD scalings, skron) against a seeded sample of the oracle's K.nzval; static regulariser,
    dynamically regularised pivots and inertia equal"""
    pr, ks = c5_full
    fx = np.load(C5_FIXTURE)
    assert int(fx["N"]) == ks.N == 277345 and int(fx["n"]) == pr["n"] and int(fx["m"]) == pr["m"]
    assert len(ks.supernodes()) >= 200
    kv = ks.values()
    assert int(fx["k_nnz"]) == len(kv)
    kerr = np.max(np.abs(kv[fx["k_sample_idx"]] - fx["k_sample_val"])) / max(1.0, float(fx["k_absmax"]))
    assert kerr <= 1e-11, "K.nzval sample: %g" % kerr
    info = ks.linear_solver_info()
    assert abs(info.last_regularizer - float(fx["regularizer"])) <= 1e-20 + 1e-12 * float(fx["regularizer"])
    assert info.regularize_count == int(fx["regularize_count"])
    assert info.positive_inertia == int(fx["positive_inertia"])
    rng = np.random.default_rng(int(fx["rhs_seed"]))
    for k in range(int(fx["nrhs"])):
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ks.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        err = relerr(np.concatenate([x, z]), fx["solutions"][k])
        assert err <= TOL, "rhs %d: rel. err vs oracle fixture %g" % (k, err)


def test_full_scale_properties_c5(hip, c5_full):
    """size-independent checks at the full size: residual of the refined solution against an independent scipy SpMV
    of the UNregularised K, linearity of the solve, inertia = n + #sparse SOC"""
    pr, ks = c5_full
    K = ks.kkt_matrix()
    vals = ks.values()
    Ku = sp.csc_matrix((vals, K.rowval.astype(np.int64), K.colptr.astype(np.int64)), shape=(ks.N, ks.N))
    rng = np.random.default_rng(1)
    b1, b2 = rng.standard_normal(ks.N), rng.standard_normal(ks.N)
    ok1, x1 = ks.solve_full(b1)
    ok2, x2 = ks.solve_full(b2)
    ok3, x3 = ks.solve_full(2.0 * b1 - 3.0 * b2)
    assert ok1 and ok2 and ok3
    for b, x in ((b1, x1), (b2, x2)):
        r = b - (Ku @ x + Ku.T @ x - Ku.diagonal() * x)
        assert np.max(np.abs(r)) <= 1e-8 * max(1.0, np.max(np.abs(b)), np.max(np.abs(x)))
    assert relerr(x3, 2.0 * x1 - 3.0 * x2) <= 1e-7
    assert ks.linear_solver_info().positive_inertia == pr["n"] + 200


def test_run_to_run_spread_c5_full(hip, c5_full):
    """four refactor + solve passes over the same values at the full size, where the 200 supernodes' update matrices
    meet in fp64 atomics (k_snode_extend): spread of the refined solution <= 1e-12, pivot-rule outcome identical"""
    pr, ks = c5_full
    rng = np.random.default_rng(5)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    sols, counts = [], []
    for _ in range(4):
        assert ks.update_scaling(pr["s"], pr["z"])
        assert ks.update()
        info = ks.linear_solver_info()
        counts.append((info.regularize_count, info.positive_inertia))
        ks.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        sols.append(np.concatenate([x, z]))
    assert len(set(counts)) == 1, counts
    spread = max(relerr(s_, sols[0]) for s_ in sols[1:])
    assert spread <= 1e-12, "run-to-run spread %g" % spread


@pytest.mark.parametrize("which", ["c3", "c2_supernodes", "c5_supernodes"])
def test_run_to_run_spread(hip, which):
    """The factorisation accumulates through fp64 atomics (LDS adds inside the bundles, slotted global adds
    for folded top rows, split-k supernode updates), so two refactor + solve passes over the same values
    need not agree bitwise (the reference is bit-deterministic).  Bound the spread of the refined solution
    and require the pivot-rule outcome (regularize_count, inertia) to be identical."""
    if which == "c3":
        pr = problems.portfolio_socp(200, 500, seed=3, late=True)
    elif which == "c2_supernodes":
        pr = problems.random_qp(20000, 40000, band=50, seed=1, late=True)
    else:
        pr = problems.chordal_sdp(6, 30, 6, 6, 21, seed=7)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
    rng = np.random.default_rng(5)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    sols, counts = [], []
    for _ in range(4):
        assert ks.update_scaling(pr["s"], pr["z"])
        assert ks.update()
        info = ks.linear_solver_info()
        counts.append((info.regularize_count, info.positive_inertia))
        ks.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        sols.append(np.concatenate([x, z]))
    assert len(set(counts)) == 1, counts
    spread = max(relerr(s, sols[0]) for s in sols[1:])
    assert spread <= 1e-12, "run-to-run spread %g" % spread


@pytest.mark.parametrize("which", ["c3", "c2_supernodes", "c5_supernodes"])
def test_threaded_analysis_same_handle(hip, which, monkeypatch):
    """the host analysis splits its big passes over std::threads by destination ownership (symbolic.cpp,
    amd_order.cpp): every array must come out the same for any thread count.  Forced on at a small size
    (CHIP_HOST_PAR_MIN=0), compared with the single-thread analysis through the device: same permutation,
    same symbolic factorisation, same pivots and a solution within the run-to-run spread."""
    if which == "c3":
        pr = problems.portfolio_socp(60, 200, seed=3, late=True)
    elif which == "c2_supernodes":
        pr = problems.random_qp(6000, 12000, band=30, seed=1, late=True)
    else:
        pr = problems.chordal_sdp(5, 20, 5, 4, 11, seed=7)
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    rng = np.random.default_rng(5)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    out = []
    for threads in (1, 5):
        monkeypatch.setenv("CHIP_HOST_THREADS", str(threads))
        monkeypatch.setenv("CHIP_HOST_PAR_MIN", "0")
        ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
        assert ks.update_scaling(pr["s"], pr["z"])
        assert ks.update()
        info = ks.linear_solver_info()
        ks.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        out.append((np.asarray(ks.perm).copy(), [np.asarray(a).copy() for a in ks.symbolic()],
                    (info.regularize_count, info.positive_inertia, info.nnzL), np.concatenate([x, z])))
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    assert out[0][2] == out[1][2]
    assert relerr(out[1][3], out[0][3]) <= 1e-12


@pytest.mark.parametrize("which", ["arrow", "forest", "forest_big", "general"])
def test_async_enqueue_collect(hip, oracle, which):
    """chip_kkt_update_enqueue / solve_dev_enqueue / collect: a whole iteration's KKT work enqueued without
    a host synchronisation, verdicts collected once -- same solutions as the synchronous calls.  "arrow" and
    "forest" run the fused solve + refinement launch with the decisions taken on the device (default
    refinement settings: data-dependent round counts), "general" (a banded QP with a tall top) takes the
    host-controlled path behind the same entry points."""
    pr = {"arrow": lambda: problems.portfolio_socp(20, 300, seed=3, late=True),
          "forest": lambda: problems.batched_socp(24, 300, 2, seed=100),
          # (few 6007-node trees, config 4's share of one GPU of eight: the 1024-thread variant of the fused launch)
          "forest_big": lambda: problems.batched_socp(6, 2000, 2, seed=100),
          "general": lambda: problems.random_qp(3000, 6000, band=20, seed=1)}[which]()
    ks, ko, cones = _solvers(hip, oracle, pr)
    rng = np.random.default_rng(9)
    rhs = [(rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])) for _ in range(3)]
    d_s, d_z = hip.DeviceArray(pr["s"]), hip.DeviceArray(pr["z"])
    d_rhs = [(hip.DeviceArray(a), hip.DeviceArray(b)) for a, b in rhs]
    d_lhs = [hip.DeviceArray(pr["n"] + pr["m"]) for _ in rhs]
    for _ in range(2):  # twice: the control blocks must be reusable without host help
        ks.update_scaling_dev(d_s.ptr, d_z.ptr)
        ks.update_enqueue()
        for (a, b), l in zip(d_rhs, d_lhs):
            ks.setrhs_dev(a.ptr, b.ptr)
            ks.solve_dev_enqueue(l.ptr, l.ptr + 8 * pr["n"])
        uok, sok = ks.collect()
        assert uok and sok == [True, True, True]
    assert cones.update_scaling(pr["s"], pr["z"]) and ko.update()
    for (a, b), l in zip(rhs, d_lhs):
        ko.setrhs(a, b)
        ok, xo, zo = ko.solve()
        assert ok and relerr(l.numpy(), np.concatenate([xo, zo])) <= TOL
    # a non-finite right-hand side: the solve's verdict is false, the handle stays usable
    bad = hip.DeviceArray(np.full(pr["n"], np.nan))
    ks.setrhs_dev(bad.ptr, d_rhs[0][1].ptr)
    ks.solve_dev_enqueue(d_lhs[0].ptr, d_lhs[0].ptr + 8 * pr["n"])
    ks.setrhs_dev(d_rhs[1][0].ptr, d_rhs[1][1].ptr)
    ks.solve_dev_enqueue(d_lhs[1].ptr, d_lhs[1].ptr + 8 * pr["n"])
    uok, sok = ks.collect()
    assert uok and sok == [False, True]


# ---- L3: DefaultKKTSystem / DefaultResiduals on the device ----------------------------
def _l3_pair(hip, oracle, pr, seed=0):
    rng = np.random.default_rng(seed)
    n, m = pr["n"], pr["m"]
    q, b = rng.standard_normal(n), rng.standard_normal(m)
    P = hip.CscMatrix(n, n, *pr["P"])
    A = hip.CscMatrix(m, n, *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], m, n)
    sysd = hip.HipKKTSystem(ks, P, A, q, b)
    cones = oracle.Cones(pr["cones"])
    ko = oracle.KKTSolver(n, m, pr["P"], pr["A"], cones, perm=ks.perm)
    syso = oracle.KKTSystem(ko, cones, n, m, pr["P"], pr["A"], q, b)
    return ks, sysd, cones, ko, syso, rng


def _dvars(hip, v):
    d = hip.DeviceVariables(len(v.x), len(v.s))
    d.x.copy_from(v.x)
    d.s.copy_from(v.s)
    d.z.copy_from(v.z)
    d.tau, d.kappa = v.tau, v.kappa
    return d


@pytest.mark.parametrize("which", ["socp", "qp"])
def test_l3_kktsystem_and_residuals(hip, oracle, which):
    """chip_kktsystem_{update,solve,solve_initial_point} + chip_residuals_update against the
    oracle restatement of default/kktsystem.rs:108-259 and default/residuals.rs:69-111"""
    pr = problems.portfolio_socp(12, 300, seed=3) if which == "socp" else problems.random_qp(800, 1500, band=12, seed=4)
    ks, sysd, cones, ko, syso, rng = _l3_pair(hip, oracle, pr)
    n, m = pr["n"], pr["m"]
    assert ks.update_scaling(pr["s"], pr["z"]) and cones.update_scaling(pr["s"], pr["z"])
    assert sysd.update() and syso.update()
    # variables at the scaling point; rhs = a random (well-scaled) direction request
    v = oracle.Variables(n, m)
    v.x, v.s, v.z = rng.standard_normal(n), pr["s"].copy(), pr["z"].copy()
    v.tau, v.kappa = 1.3, 0.7
    rhs = oracle.Variables(n, m)
    rhs.x, rhs.z, rhs.s = rng.standard_normal(n), rng.standard_normal(m), cones.affine_ds(pr["s"])
    rhs.tau, rhs.kappa = 0.4, -0.2
    for direction in (hip.STEP_AFFINE, hip.STEP_COMBINED):
        lo = oracle.Variables(n, m)
        assert syso.solve(lo, rhs, v, direction)
        ld, dr, dv = hip.DeviceVariables(n, m), _dvars(hip, rhs), _dvars(hip, v)
        assert sysd.solve(ld, dr, dv, direction)
        assert abs(ld.tau - lo.tau) <= TOL * max(1.0, abs(lo.tau))
        assert abs(ld.kappa - lo.kappa) <= TOL * max(1.0, abs(lo.kappa))
        for a, c in ((ld.x, lo.x), (ld.z, lo.z), (ld.s, lo.s)):
            assert relerr(a.numpy(), c) <= TOL
    # residuals
    ro = syso.residuals(v)
    D = hip.DeviceArray
    rx, rz, rxi, rzi, Px = D(n), D(m), D(n), D(m), D(n)
    rd = sysd.residuals_update(_dvars(hip, v), rx, rz, rxi, rzi, Px)
    for k in ("rtau", "dot_qx", "dot_bz", "dot_sz", "dot_xPx"):
        assert abs(rd[k] - ro[k]) <= 1e-10 * max(1.0, abs(ro[k]))
    for a, k in ((rx, "rx"), (rz, "rz"), (rxi, "rx_inf"), (rzi, "rz_inf"), (Px, "Px")):
        assert relerr(a.numpy(), ro[k]) <= 1e-12
    # initial point (LP branch for the SOCP: P == 0; QP branch otherwise)
    vo, vd = oracle.Variables(n, m), hip.DeviceVariables(n, m)
    assert syso.solve_initial_point(vo) and sysd.solve_initial_point(vd)
    for a, c in ((vd.x, vo.x), (vd.z, vo.z), (vd.s, vo.s)):
        assert relerr(a.numpy(), c) <= TOL


def test_l3_update_data(hip, oracle):
    """data_updating.rs:98-133 through chip_kktsystem_update_data: new P / A values on the same patterns
    and new q / b reach the KKT matrix, the SpMV copies and the RHS algebra"""
    pr = problems.random_qp(400, 700, band=10, seed=9)
    ks, sysd, cones, ko, syso, rng = _l3_pair(hip, oracle, pr)
    n, m = pr["n"], pr["m"]
    P2 = (pr["P"][0], pr["P"][1], pr["P"][2] * 1.7)
    A2 = (pr["A"][0], pr["A"][1], pr["A"][2] * rng.uniform(0.5, 1.5, len(pr["A"][2])))
    q2, b2 = rng.standard_normal(n), rng.standard_normal(m)
    sysd.update_data(P=P2[2], A=A2[2], q=q2, b=b2)
    cones2 = oracle.Cones(pr["cones"])
    ko2 = oracle.KKTSolver(n, m, P2, A2, cones2, perm=ks.perm)
    syso2 = oracle.KKTSystem(ko2, cones2, n, m, P2, A2, q2, b2)
    assert ks.update_scaling(pr["s"], pr["z"]) and cones2.update_scaling(pr["s"], pr["z"])
    assert sysd.update() and syso2.update()
    assert relerr(ks.values(), ko2.kkt.nzval) <= 1e-13
    v = oracle.Variables(n, m)
    v.x, v.s, v.z, v.tau, v.kappa = rng.standard_normal(n), pr["s"].copy(), pr["z"].copy(), 0.9, 1.1
    ro = syso2.residuals(v)
    D = hip.DeviceArray
    rx, rz, rxi, rzi, Px = D(n), D(m), D(n), D(m), D(n)
    rd = sysd.residuals_update(_dvars(hip, v), rx, rz, rxi, rzi, Px)
    assert abs(rd["rtau"] - ro["rtau"]) <= 1e-10 * max(1.0, abs(ro["rtau"]))
    assert relerr(rx.numpy(), ro["rx"]) <= 1e-12 and relerr(rz.numpy(), ro["rz"]) <= 1e-12
    rhs = oracle.Variables(n, m)
    rhs.x, rhs.z, rhs.s, rhs.tau, rhs.kappa = ro["rx"], ro["rz"], cones2.affine_ds(pr["s"]), ro["rtau"], 0.99
    lo, ld = oracle.Variables(n, m), hip.DeviceVariables(n, m)
    assert syso2.solve(lo, rhs, v, hip.STEP_AFFINE) and sysd.solve(ld, _dvars(hip, rhs), _dvars(hip, v), hip.STEP_AFFINE)
    assert abs(ld.tau - lo.tau) <= TOL * max(1.0, abs(lo.tau))
    assert relerr(ld.x.numpy(), lo.x) <= TOL and relerr(ld.z.numpy(), lo.z) <= TOL


@pytest.mark.parametrize("name", ["basic_qp", "basic_lp", "basic_socp", "basic_expcone", "basic_powcone", "basic_sdp", "basic_genpowcone", "basic_unconstrained",
                                  "basic_eq_constrained"])
def test_e2e_reference_answers_on_device(hip, oracle, name):
    """the reference's end-to-end known answers (tests/basic_qp.rs:100-117, basic_lp.rs:27-44,
    basic_socp.rs:54-70, basic_expcone.rs:38-56, basic_powcone.rs:4-47, basic_sdp.rs:29-57, basic_genpowcone.rs:4-55) reached with every L1-L3 operation on the device, and the same
    trajectory as the oracle-backed loop"""
    from tests import e2e_problems as E
    from tests import ipm_driver as ipm
    pr = getattr(E, name)()
    args = (pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    td, to = [], []
    out = ipm.solve(ipm.HipBackend(hip, *args), pr["cones"], pr["q"], pr["b"], trace=td)
    ref = ipm.solve(ipm.OracleBackend(oracle, *args), pr["cones"], pr["q"], pr["b"], trace=to)
    assert out["status"] == "Solved"
    if pr["x"] is not None:
        assert np.linalg.norm(out["x"] - np.array(pr["x"])) <= pr["tol"]
    assert abs(out["obj_val"] - pr["obj"]) <= pr["tol"]
    assert out["iterations"] == ref["iterations"]
    for a, c in zip(td, to):  # (mu, alpha, sigma, res_primal, res_dual, gap): same path
        if c[0] > 1e-6:  # near convergence the iterates amplify rounding differences
            assert abs(a[0] - c[0]) <= 1e-6 * c[0]
            assert abs(a[1] - c[1]) <= 1e-6


def test_e2e_sparse_soc_on_device(hip):
    from tests import e2e_problems as E
    from tests import ipm_driver as ipm
    pr = E.basic_socp(sparse_soc=True)
    be = ipm.HipBackend(hip, pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    assert ipm.solve(be, pr["cones"], pr["q"], pr["b"])["status"] == "Solved"


def test_json_fixture_hs35_on_device(hip):
    import os
    from tests import ipm_driver as ipm
    from tests import json_problem
    pr = json_problem.load(os.path.join(os.path.dirname(__file__), "golden", "hs35_reference.json"))  # the reference's own file
    be = ipm.HipBackend(hip, pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    out = ipm.solve(be, pr["cones"], pr["q"], pr["b"])
    assert out["status"] == "Solved"
    assert np.linalg.norm(out["x"] - np.array([4.0 / 3.0, 7.0 / 9.0, 4.0 / 9.0])) <= 1e-6
    assert abs(out["obj_val"] + 9.0 - 1.0 / 9.0) <= 1e-6


@pytest.mark.parametrize("min_switch", [0.1, 0.999])
def test_e2e_mixed_conic_on_device(hip, oracle, min_switch):
    """tests/mixed_conic.rs:4-45 with every operation on the device, both scaling strategies (the
    second forces the dual scaling + barrier backtracking), same iteration count as the oracle loop"""
    from tests import e2e_problems as E
    from tests import ipm_driver as ipm
    pr = E.mixed_conic()
    args = (pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    out = ipm.solve(ipm.HipBackend(hip, *args), pr["cones"], pr["q"], pr["b"], min_switch_step_length=min_switch)
    ref = ipm.solve(ipm.OracleBackend(oracle, *args), pr["cones"], pr["q"], pr["b"], min_switch_step_length=min_switch)
    assert out["status"] == "Solved" and abs(out["obj_val"]) <= 1e-8 and np.linalg.norm(out["x"]) <= 1e-6
    if min_switch < 0.5:
        assert out["iterations"] == ref["iterations"]
    else:
        # the optimum x = 0 is the apex of all five cones and dx is rounding noise (~1e-15): whether a
        # trial point of the barrier backtracking is inside a cone (finite barrier) or outside (NaN) is
        # decided by that noise, so the two loops take different -- equally valid -- paths.  Every
        # operation agrees when fed the same inputs (checked by shadowing one loop with the other).
        assert out["iterations"] <= 30 and ref["iterations"] <= 30


# ---- DefaultVariables on the device (default/variables.rs:58-261) ---------------------------
@pytest.mark.parametrize("which", ["socp", "mixed"])
def test_variables_operations_on_device(hip, oracle, which):
    """chip_variables_* against the same algebra composed on the host from the oracle's cone
    operations (variables.rs:63-239), plus chip_vec_norms / chip_kkt_degree"""
    from tests import ipm_driver as ipm
    pr = problems.portfolio_socp(6, 120, seed=5) if which == "socp" else problems.mixed_conic()
    ks, sysd, cones, ko, syso, rng = _l3_pair(hip, oracle, pr, seed=3)
    n, m = pr["n"], pr["m"]
    strategy = 0
    assert ks.update_scaling(pr["s"], pr["z"], 0.7, strategy) and cones.update_scaling(pr["s"], pr["z"], 0.7, strategy)
    assert ks.degree() == cones.degree
    v = oracle.Variables(n, m)
    v.x, v.s, v.z, v.tau, v.kappa = rng.standard_normal(n), pr["s"].copy(), pr["z"].copy(), 1.2, 0.8
    res = syso.residuals(v)
    D = hip.DeviceArray
    dv = _dvars(hip, v)
    rx, rz = D(res["rx"]), D(res["rz"])
    deg = cones.degree
    mu = (res["dot_sz"] + v.tau * v.kappa) / (deg + 1)
    assert abs(sysd.calc_mu(dv, res["dot_sz"]) - mu) <= 1e-15 * abs(mu)
    # affine rhs
    d = hip.DeviceVariables(n, m)
    sysd.affine_step_rhs(d, rx, rz, res["rtau"], dv)
    aff_s = cones.affine_ds(v.s)
    assert relerr(d.x.numpy(), res["rx"]) == 0 and relerr(d.z.numpy(), res["rz"]) == 0
    assert relerr(d.s.numpy(), aff_s) <= 1e-13
    assert d.tau == res["rtau"] and d.kappa == v.tau * v.kappa
    # a step (small, so that the point stays interior) and the combined rhs
    step = oracle.Variables(n, m)
    step.x, step.z, step.s = rng.standard_normal(n), 0.6 * rng.standard_normal(m) * np.abs(v.z).mean(), \
        0.6 * rng.standard_normal(m) * np.abs(v.s).mean()
    step.tau, step.kappa = -0.3, 0.2
    sigma, mm = 0.3, 0.9
    dstep = _dvars(hip, step)
    sysd.combined_step_rhs(d, rx, rz, res["rtau"], dv, dstep, sigma, mu, mm)
    shift, sz, ss = cones.combined_ds_shift(mm * step.z, step.s, sigma * mu)
    assert relerr(d.x.numpy(), (1 - sigma) * res["rx"]) <= 1e-15
    assert relerr(d.z.numpy(), (1 - sigma) * res["rz"]) <= 1e-15
    assert relerr(d.s.numpy(), aff_s + shift) <= 1e-11
    assert abs(d.tau - (1 - sigma) * res["rtau"]) <= 1e-15 * max(1, abs(res["rtau"]))
    assert abs(d.kappa - (-sigma * mu + mm * step.tau * step.kappa + v.tau * v.kappa)) <= 1e-15
    assert relerr(dstep.z.numpy(), sz) <= 1e-11 and relerr(dstep.s.numpy(), ss) <= 1e-11  # overwritten as the reference
    # step length, barrier, add_step, rescale, norms on a fresh copy of the step
    dstep = _dvars(hip, step)
    for direction, frac in ((hip.STEP_AFFINE, 0.99), (hip.STEP_COMBINED, 0.99)):
        a_ref = min(-v.tau / step.tau, 1.0)
        a_ref = cones.step_length(step.z, step.s, v.z, v.s, a_ref)
        if direction == hip.STEP_COMBINED:
            a_ref *= frac
        assert abs(sysd.calc_step_length(dv, dstep, direction, frac) - a_ref) <= 1e-9
    alpha = 0.5 * a_ref
    ctau, ckap = v.tau + alpha * step.tau, v.kappa + alpha * step.kappa
    mu_a = (float(np.dot(v.s + alpha * step.s, v.z + alpha * step.z)) + ctau * ckap) / (deg + 1)
    bar = (deg + 1) * np.log(mu_a) - np.log(ctau) - np.log(ckap) + cones.compute_barrier(v.z, v.s, step.z, step.s, alpha)
    assert abs(sysd.barrier(dv, dstep, alpha) - bar) <= 1e-8 * max(1.0, abs(bar))
    sysd.add_step(dv, dstep, alpha)
    assert relerr(dv.x.numpy(), v.x + alpha * step.x) <= 1e-15 and relerr(dv.z.numpy(), v.z + alpha * step.z) <= 1e-15
    assert relerr(dv.s.numpy(), v.s + alpha * step.s) <= 1e-15
    assert abs(dv.tau - ctau) <= 1e-15 and abs(dv.kappa - ckap) <= 1e-15
    nx, nz, nrx = sysd.vec_norms(dv.x, dv.z, rx)
    assert abs(nx - np.linalg.norm(v.x + alpha * step.x)) <= 1e-12 * nx and abs(nrx - np.linalg.norm(res["rx"])) <= 1e-12 * nrx
    assert abs(nz - np.linalg.norm(v.z + alpha * step.z)) <= 1e-12 * nz
    assert sysd.vec_norms() == []
    x_before, t_before = dv.x.numpy(), (dv.tau, dv.kappa)
    sysd.rescale(dv)
    sc = max(t_before)
    assert relerr(dv.x.numpy(), x_before / sc) <= 1e-15 and abs(dv.tau - t_before[0] / sc) <= 1e-15
    assert max(dv.tau, dv.kappa) == 1.0
    # initialisations
    if which == "socp":
        w = oracle.Variables(n, m)
        w.s, w.z = rng.standard_normal(m), rng.standard_normal(m)
        dw = _dvars(hip, w)
        sysd.symmetric_initialization(dw)
        be = ipm.OracleBackend(oracle, n, m, pr["P"], pr["A"], np.zeros(n), np.zeros(m), pr["cones"])
        ipm._shift_to_cone_interior(be, w.s, True)
        ipm._shift_to_cone_interior(be, w.z, False)
        assert relerr(dw.s.numpy(), w.s) <= 1e-12 and relerr(dw.z.numpy(), w.z) <= 1e-12
        assert dw.tau == 1.0 and dw.kappa == 1.0
    else:
        dw = hip.DeviceVariables(n, m)
        dw.x.copy_from(rng.standard_normal(n))
        sysd.unit_initialization(dw)
        zz, ss2 = np.zeros(m), np.zeros(m)
        cones.unit_initialization(zz, ss2)
        assert relerr(dw.z.numpy(), zz) <= 1e-15 and relerr(dw.s.numpy(), ss2) <= 1e-15
        assert not dw.x.numpy().any() and dw.tau == 1.0 and dw.kappa == 1.0


@pytest.mark.parametrize("name", ["basic_qp", "basic_lp", "basic_socp", "basic_expcone", "basic_powcone", "basic_sdp",
                                  "basic_genpowcone", "basic_eq_constrained"])
def test_device_resident_ipm_loop(hip, oracle, name):
    """tests/ipm_device.py: the whole iteration (residuals, scaling, KKT update, both step right-hand
    sides, both solves, step lengths, add_step) with all vectors resident in HBM reaches the
    reference's answers along the same path as the oracle-backed loop"""
    from tests import e2e_problems as E
    from tests import ipm_device, ipm_driver as ipm
    pr = getattr(E, name)()
    args = (pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    td, to = [], []
    out = ipm_device.solve_device(hip, *args, trace=td)
    ref = ipm.solve(ipm.OracleBackend(oracle, *args), pr["cones"], pr["q"], pr["b"], trace=to)
    assert out["status"] == "Solved"
    if pr["x"] is not None:
        assert np.linalg.norm(out["x"] - np.array(pr["x"])) <= pr["tol"]
    assert abs(out["obj_val"] - pr["obj"]) <= pr["tol"]
    assert out["iterations"] == ref["iterations"]
    for a, c in zip(td, to):
        if c[0] > 1e-6:
            assert abs(a[0] - c[0]) <= 1e-6 * c[0]
            assert abs(a[1] - c[1]) <= 1e-6


def test_device_resident_ipm_portfolio(hip, oracle):
    """a feasible portfolio SOCP (config-3 shape, small) solved by the device-resident loop and by the
    oracle-backed loop: same iteration count, same optimum"""
    from tests import ipm_device, ipm_driver as ipm
    pr = problems.portfolio_problem(8, 60, seed=2)
    args = (pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    out = ipm_device.solve_device(hip, *args)
    ref = ipm.solve(ipm.OracleBackend(oracle, *args), pr["cones"], pr["q"], pr["b"])
    assert out["status"] == "Solved" and ref["status"] == "Solved"
    assert out["iterations"] == ref["iterations"]
    assert abs(out["obj_val"] - ref["obj_val"]) <= 1e-7 * max(1.0, abs(ref["obj_val"]))
    assert np.linalg.norm(out["x"] - ref["x"]) <= 1e-5 * max(1.0, np.linalg.norm(ref["x"]))


# ---- chain supernodes: dense trapezoids factored on the matrix cores --------------------------
@pytest.mark.parametrize("which", ["banded_qp", "chordal_sdp", "wide_psd"])
def test_chain_supernodes_factor_parity(hip, oracle, which, monkeypatch):
    """k_factor_snode (padded chain supernodes of the top, v_mfma_f64_16x16x4_f64 block updates) against
    the oracle and against the column-by-column path (CHIP_NO_SNODE): same solution, same pivots, the
    padded entries of L exactly zero"""
    if which == "banded_qp":
        pr, hs = problems.random_qp(20000, 40000, band=50, seed=1), None
    elif which == "chordal_sdp":
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
        hs = pr["hsblocks"]
    else:  # one supernode wider than a k-chunk of the kernel (w > 384 + 32)
        pr = problems.chordal_sdp(2, 40, 6, 1, 5, seed=7)
        hs = pr["hsblocks"]
    ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=2)
    sns = ks.supernodes()
    assert len(sns) > 0
    if which == "wide_psd":
        assert max(len(c) for c in sns) > 416
    # the same factorisation without supernodes, through the L1 handle on the same K and permutation
    K = ks.kkt_matrix()
    vals = ks.values()
    Kc = hip.CscMatrix(ks.N, ks.N, K.colptr, K.rowval, vals)
    dsigns = ks.maps()["dsigns"]
    st = hip.Settings.default()
    fa = hip.HipDirectLDLSolver(Kc, dsigns, st, perm=ks.perm)
    monkeypatch.setenv("CHIP_NO_SNODE", "1")
    fb = hip.HipDirectLDLSolver(Kc, dsigns, st, perm=ks.perm)
    monkeypatch.delenv("CHIP_NO_SNODE")
    assert len(fa.supernodes()) > 0 and fb.supernodes() == []
    fa.refactor()
    fb.refactor()
    Lpa, Lia, Lxa, Da, _ = fa.factors()
    Lpb, Lib, Lxb, Db, _ = fb.factors()
    assert relerr(Da, Db) <= 1e-9
    # entry by entry: common entries agree, padded entries are exact zeros
    import scipy.sparse as sp
    N = ks.N
    La = sp.csc_matrix((Lxa, Lia, Lpa), shape=(N, N))
    Lb = sp.csc_matrix((Lxb, Lib, Lpb), shape=(N, N))
    diff = (La - Lb).tocoo()
    assert np.max(np.abs(diff.data), initial=0.0) <= 1e-9 * max(1.0, np.max(np.abs(Lxb)))
    pad = sp.csc_matrix((np.ones(len(Lia)), Lia, Lpa), shape=(N, N)) - sp.csc_matrix((np.ones(len(Lib)), Lib, Lpb), shape=(N, N))
    pad.eliminate_zeros()
    assert (pad.data > 0).all()  # a superset
    if which != "wide_psd":  # (that block is dense already: nothing to pad)
        assert pad.nnz > 0
        assert np.abs(np.asarray(La[pad.nonzero()])).max() == 0.0


@pytest.mark.parametrize("form", ["CHIP_NO_SNODE_PANEL", "CHIP_NO_PANEL_MFMA", "CHIP_NO_PANEL_DIAG_MFMA", "CHIP_SN_PANEL_SLOTS",
                                  "CHIP_NO_PANEL_OVERLAP", "CHIP_NO_PANEL_OVERLAP+CHIP_SN_PANEL_SLOTS", "CHIP_NO_PANEL_UNIFORM",
                                  "CHIP_NO_FACTOR_OVERLAP"])
@pytest.mark.parametrize("which", ["banded_qp", "chordal_sdp"])
def test_chain_supernodes_fallback_forms(hip, oracle, which, form, monkeypatch):
    """the block column of a supernode has three older forms behind switches -- separate k_snode_diag / k_snode_rows
    launches, and the scalar forms of the panel kernel's two phases -- and, in launches with many supernodes, a form in
    which a workgroup walks several groups of 256 rows (CHIP_SN_PANEL_SLOTS=1 forces it here: one workgroup per
    supernode); CHIP_NO_PANEL_OVERLAP: the panel kernel whose block factorisation and rows run one after the other
    (k_snode_panel) instead of overlapped by two teams of waves (k_snode_panel2, the default); CHIP_NO_FACTOR_OVERLAP: the
    bundle columns' contributions into the supernode members all ahead of the chain of block columns instead of beside it
    on the second stream: each against the oracle, and its factor against the default form's (same pivots, entries
    within rounding)"""
    if which == "banded_qp":
        pr, hs = problems.random_qp(20000, 40000, band=50, seed=1), None
    else:
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
        hs = pr["hsblocks"]
    ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1)
    assert len(ks.supernodes()) > 0
    K = ks.kkt_matrix()
    Kc = hip.CscMatrix(ks.N, ks.N, K.colptr, K.rowval, ks.values())
    dsigns = ks.maps()["dsigns"]
    fa = hip.HipDirectLDLSolver(Kc, dsigns, hip.Settings.default(), perm=ks.perm)
    fa.refactor()
    _, _, Lxa, Da, _ = fa.factors()
    for name in form.split("+"):
        monkeypatch.setenv(name, "1")
    ks2, _ = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1)
    fb = hip.HipDirectLDLSolver(Kc, dsigns, hip.Settings.default(), perm=ks.perm)
    fb.refactor()
    _, _, Lxb, Db, _ = fb.factors()
    assert relerr(Da, Db) <= 1e-10
    assert np.max(np.abs(Lxa - Lxb)) <= 1e-10 * max(1.0, np.max(np.abs(Lxa)))
    if form == "CHIP_NO_PANEL_UNIFORM" and os.environ.get("CHIP_DETERMINISTIC"):
        # (the two forms of the block factorisation do the same operations in the same order: with the k-split atomics
        # of the update tiles out of the way the factors are bitwise equal)
        assert np.array_equal(Da, Db) and np.array_equal(Lxa, Lxb)


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("emit", ["atomics", "assembled"])
def test_ancestor_updates_in_wide_tiles(hip, oracle, emit, waves, monkeypatch):
    """k_snode_extend_wide (64 x 256 / 128 x 256 tiles: four / eight waves of 16 rows x 256 columns, the staged operand double buffered) against
    the 256 x 64 tiles of k_snode_extend on supernodes with more than 256 rows of B -- a last column block narrower than
    256, row groups that end inside a tile, tiles on and above the diagonal: each against the oracle, their factors
    against each other; with both ways the tiles leave (fp64 atomics / stores into the per-target-column assembly).
    CHIP_SN_WIDE_MIN_COUNT=1: on levels of any size (default: from eight supernodes on)."""
    pr = problems.chordal_sdp(4, 40, 24, 2, 9, seed=11)  # (supernodes of 820 columns with 316 / 616 rows of B)
    hs = pr["hsblocks"]
    monkeypatch.setenv("CHIP_SN_WIDE_MIN_COUNT", "1")
    monkeypatch.setenv("CHIP_SN_WIDE_WAVES", str(waves))
    monkeypatch.setenv("CHIP_NO_EXTEND_ASM" if emit == "atomics" else "CHIP_EXTEND_ASM_MIN", "1" if emit == "atomics" else "2")
    factors = {}
    for form in ("wide", "CHIP_NO_SN_WIDE"):
        if form != "wide":
            monkeypatch.setenv(form, "1")
        ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1)
        sns = ks.supernodes()
        et, Lp, Li, lv = ks.symbolic()
        cnt = np.diff(Lp)
        assert max(int(cnt[c[-1]]) for c in sns) > 256  # rows of B of the widest update: more than one wide column block
        K = ks.kkt_matrix()
        Kc = hip.CscMatrix(ks.N, ks.N, K.colptr, K.rowval, ks.values())
        f = hip.HipDirectLDLSolver(Kc, ks.maps()["dsigns"], hip.Settings.default(), perm=ks.perm)
        f.refactor()
        _, _, Lx, D, _ = f.factors()
        factors[form] = (Lx, D)
    (Lxa, Da), (Lxb, Db) = factors["wide"], factors["CHIP_NO_SN_WIDE"]
    assert relerr(Da, Db) <= 1e-10
    assert np.max(np.abs(Lxa - Lxb)) <= 1e-10 * max(1.0, np.max(np.abs(Lxa)))


@pytest.mark.parametrize("which", ["banded_qp", "chordal_sdp", "chordal_sdp_long_columns"])
def test_ancestor_updates_assembled_per_target_column(hip, oracle, which, monkeypatch):
    """a unit level's update matrices written to a private buffer and summed per target column (k_snode_assemble, a
    fixed order of summation) against the fp64 atomics of k_snode_extend: each against the oracle, their factors
    against each other.  CHIP_EXTEND_ASM_MIN=2 assembles every level with two supernodes or more (default: four)."""
    if which == "banded_qp":
        pr, hs = problems.random_qp(20000, 40000, band=50, seed=1), None
    elif which == "chordal_sdp":
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
        hs = pr["hsblocks"]
    else:  # target columns longer than the kernel's LDS window (4096 rows; here cut to 100)
        pr = problems.chordal_sdp(12, 36, 8, 4, 9, seed=3)
        hs = pr["hsblocks"]
        monkeypatch.setenv("CHIP_SN_ASM_CAP", "100")
    factors = {}
    for form in ("CHIP_NO_EXTEND_ASM", "CHIP_EXTEND_ASM_MIN"):
        monkeypatch.setenv(form, "2" if form == "CHIP_EXTEND_ASM_MIN" else "1")
        ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1)
        assert len(ks.supernodes()) > 1
        K = ks.kkt_matrix()
        Kc = hip.CscMatrix(ks.N, ks.N, K.colptr, K.rowval, ks.values())
        f = hip.HipDirectLDLSolver(Kc, ks.maps()["dsigns"], hip.Settings.default(), perm=ks.perm)
        f.refactor()
        Lp, _, Lx, D, _ = f.factors()
        factors[form] = (Lx, D, int(np.max(np.diff(Lp))))
        monkeypatch.delenv(form)
    (Lxa, Da, _), (Lxb, Db, longest) = factors["CHIP_NO_EXTEND_ASM"], factors["CHIP_EXTEND_ASM_MIN"]
    assert relerr(Da, Db) <= 1e-10
    assert np.max(np.abs(Lxa - Lxb)) <= 1e-10 * max(1.0, np.max(np.abs(Lxa)))
    if which == "chordal_sdp_long_columns":
        assert longest > 300


@pytest.mark.parametrize("which", ["chordal_sdp", "wide_psd", "three_tiles"])
def test_dense_blocks_of_the_top_in_the_residual(hip, oracle, which, monkeypatch):
    """the Hs blocks of PSD cones whose rows sit in the top: the refinement residual multiplies them from K's values
    directly (k_dblk_symv, every entry read once) and the full-row copy S of the top rows leaves them out -- the refined
    solutions against the oracle and against the form without the mechanism
    (CHIP_NO_DENSE_SYMV).  CHIP_DENSE_SYMV_MIN=1: the mechanism also for these small totals (default: 2^20 entries)."""
    if which == "chordal_sdp":
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
    elif which == "wide_psd":
        pr = problems.chordal_sdp(2, 40, 6, 1, 5, seed=7)
    else:  # blocks of 1275: three row tiles of the kernel, the last one ragged
        pr = problems.chordal_sdp(3, 50, 10, 3, 9, seed=2)
    monkeypatch.setenv("CHIP_NO_DENSE_SYMV", "1")
    ks0, _ = _check_update_and_solve(hip, oracle, pr, hs=pr["hsblocks"], nrhs=1)
    assert hip.debug_counter(ks0, "dense_blocks") == 0
    monkeypatch.delenv("CHIP_NO_DENSE_SYMV")
    monkeypatch.setenv("CHIP_DENSE_SYMV_MIN", "1")
    ks, ko = _check_update_and_solve(hip, oracle, pr, hs=pr["hsblocks"], nrhs=3)
    nb, rows = hip.debug_counter(ks, "dense_blocks"), hip.debug_counter(ks, "dense_block_rows")
    assert nb >= 2 and rows >= 64 * nb
    assert hip.debug_counter(ks, "nnzS") < hip.debug_counter(ks0, "nnzS") - rows * 60
    # the same right-hand side through both handles: the refined solutions agree to rounding
    rng = np.random.default_rng(3)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    sols = []
    for k in (ks0, ks):
        k.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert k.solve(x, z)
        sols.append(np.concatenate([x, z]))
    assert relerr(sols[1], sols[0]) <= 1e-12


@pytest.mark.parametrize("which", ["banded_qp", "chordal_sdp"])
def test_chain_supernodes_solve_without_lds_rows(hip, oracle, which, monkeypatch):
    """the substitutions through supernodes whose rows of B exceed the LDS budget (forced here by
    CHIP_SN_XB_CAP): global atomics / loads instead of the LDS copy, same solution (the block-by-block substitution
    kernels: CHIP_NO_SNODE_G keeps these supernodes off the one-pass matrices)"""
    monkeypatch.setenv("CHIP_SN_XB_CAP", "16")
    monkeypatch.setenv("CHIP_NO_SNODE_G", "1")
    if which == "banded_qp":
        pr, hs = problems.random_qp(20000, 40000, band=50, seed=1), None
    else:
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
        hs = pr["hsblocks"]
    ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=2)
    assert len(ks.supernodes()) > 0


@pytest.mark.parametrize("which", ["banded_qp", "banded_qp_late", "chordal_sdp", "wide_psd", "mixed_widths"])
def test_supernode_substitution_matrices(hip, oracle, which, monkeypatch):
    """substitutions through supernodes of moderate width in ONE pass over G = [I; L_B] T^-1 (snode_g.hip: k_snode_ginv
    per refactor, k_snode_gfwd / k_snode_gbwd per sweep) against the oracle and against the block-by-block substitution
    of the same factors (CHIP_NO_SNODE_G): benign and late-iterate scalings, a supernode wider than one staged chunk of
    the build (w > 416), and a handle on which only the narrow levels take the path (CHIP_SN_G_MAXW)"""
    hs = None
    if which == "banded_qp":
        pr = problems.random_qp(20000, 40000, band=50, seed=1)
    elif which == "banded_qp_late":
        pr = problems.random_qp(20000, 40000, band=50, seed=1, late=True)
    elif which == "chordal_sdp":
        pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)
        hs = pr["hsblocks"]
    elif which == "wide_psd":
        pr = problems.chordal_sdp(2, 40, 6, 1, 5, seed=7)
        hs = pr["hsblocks"]
    else:
        pr = problems.random_qp(20000, 40000, band=50, seed=1)
        monkeypatch.setenv("CHIP_SN_G_MAXW", "250")
    ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=3)
    ng, nl = hip.debug_counter(ks, "g_levels"), hip.debug_counter(ks, "sn_levels")
    assert ng >= 1 and nl >= ng
    if which == "mixed_widths":
        assert ng < nl  # (some levels keep the block-by-block substitution)
    elif which != "wide_psd":
        assert ng == nl
    # the same handle shape without the matrices: refined solutions agree to rounding (both handles refactor a second
    # time first: the matrices are rebuilt by every refactor)
    monkeypatch.setenv("CHIP_NO_SNODE_G", "1")
    ks0, _ = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1)
    assert hip.debug_counter(ks0, "g_levels") == 0
    monkeypatch.delenv("CHIP_NO_SNODE_G")
    rng = np.random.default_rng(11)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    sols = []
    for k in (ks0, ks):
        assert k.update_scaling(pr["s"], pr["z"]) and k.update(hs)
        k.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert k.solve(x, z)
        sols.append(np.concatenate([x, z]))
    assert relerr(sols[1], sols[0]) <= 1e-9
    # backward error BEFORE refinement of the two forms (what decides whether a hard iterate costs one refinement round
    # or three): ||b - K x||inf / (||K||inf ||x||inf + ||b||inf) of ONE LDL' solve, K = the device's unregularised values.
    # G = [I; L_B] T^-1 holds an explicit triangular inverse: it may lose digits where T is ill conditioned, and the bound
    # below is where that would show.
    Kc = ks.kkt_matrix()
    Ku = sp.csc_matrix((ks.values(), Kc.rowval.astype(np.int64), Kc.colptr.astype(np.int64)), shape=(ks.N, ks.N))  # (the CURRENT values)
    Kf = Ku + sp.triu(Ku, 1).T
    knorm = float(abs(Kf).sum(axis=1).max())
    bfull = np.concatenate([rx, rz, np.zeros(ks.N - pr["n"] - pr["m"])])
    st0 = hip.Settings.default(iterative_refinement_enable=0)
    berr = []
    for k in (ks0, ks):
        k.set_settings(st0)
        ok, xf = k.solve_full(bfull)
        assert ok
        berr.append(float(np.max(np.abs(bfull - Kf @ xf)) / (knorm * np.max(np.abs(xf)) + np.max(np.abs(bfull)))))
        k.set_settings(hip.Settings.default())
    print("backward error before refinement [%s]: block substitution %.2e, one-pass matrices %.2e" % (which, berr[0], berr[1]))
    assert berr[1] <= max(64.0 * berr[0], 1e-13), berr
    # the row gathers over non-member columns in their own launches instead of inside the supernodes' launches
    monkeypatch.setenv("CHIP_NO_SWEEP_MERGE", "1")
    _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=2)
    monkeypatch.delenv("CHIP_NO_SWEEP_MERGE")
    # without refinement the two forms still agree to the accuracy of one LDL' solve
    st = hip.Settings.default(iterative_refinement_enable=0)
    _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1, settings=st, tol=1e-5 if "late" in which else 1e-6)


@pytest.mark.parametrize("which,grid", [("banded_qp", 0), ("banded_qp", 1), ("banded_qp", 3), ("banded_qp_late", 0),
                                        ("narrow_band", 2), ("mixed_widths", 0), ("banded_qp_graph", 0)])
def test_persistent_sweeps_over_runs_of_unit_levels(hip, oracle, which, grid, monkeypatch):
    """a run of consecutive unit levels on the one-pass matrices is ONE persistent launch per sweep (snode_g.hip:
    k_snode_gsweep, a grid barrier between the levels, the vector read and written at the coherence point): solutions
    against the oracle, and against the launch-per-level form of the same handle shape (CHIP_NO_SWEEP_PERSIST) -- the two
    run the same arithmetic on the supernodes' blocks; also with grids of 1, 2 and 3 workgroups (every workgroup walks
    several tasks per level), at a late iterate's scaling and on a handle whose wide levels split the sweep into several
    runs.  (The chordal SDPs of this suite alternate supernode levels with levels of ordinary columns -- no run of two;
    BASELINE config 5, whose narrow levels form runs, is checked against the oracle by bench.py at full size.)"""
    hs = None
    if which == "narrow_band":
        pr = problems.random_qp(12000, 24000, band=30, seed=4)
    else:
        pr = problems.random_qp(20000, 40000, band=50, seed=1, late=(which == "banded_qp_late"))
        if which == "mixed_widths":
            monkeypatch.setenv("CHIP_SN_G_MAXW", "250")
    if grid:
        monkeypatch.setenv("CHIP_GSWEEP_GRID", str(grid))
    # (banded_qp_graph: the solve sequence captured and replayed as hipGraphs -- the barrier's words are back at zero when
    # a persistent launch ends, so a replay meets them as the first launch did)
    st = hip.Settings.default(use_graph=1) if which == "banded_qp_graph" else None
    ks, ko = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=4 if st else 3, settings=st)
    runs, lv = hip.debug_counter(ks, "gsweep_runs"), hip.debug_counter(ks, "gsweep_levels")
    assert runs >= 2 and lv >= 2 * runs                      # (a forward and a backward run at least)
    assert hip.debug_counter(ks, "gsweep_launches") >= runs  # the solves above went through them
    monkeypatch.setenv("CHIP_NO_SWEEP_PERSIST", "1")
    ks0, _ = _check_update_and_solve(hip, oracle, pr, hs=hs, nrhs=1)
    assert hip.debug_counter(ks0, "gsweep_runs") == 0 and hip.debug_counter(ks0, "gsweep_launches") == 0
    monkeypatch.delenv("CHIP_NO_SWEEP_PERSIST")
    rng = np.random.default_rng(17)
    for _ in range(2):
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        sols = []
        for k in (ks0, ks):
            assert k.update_scaling(pr["s"], pr["z"]) and k.update(hs)
            k.setrhs(rx, rz)
            x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
            assert k.solve(x, z)
            sols.append(np.concatenate([x, z]))
        assert relerr(sols[1], sols[0]) <= 1e-10


def test_persistent_sweep_timeout_recovers(hip, oracle):
    """a persistent sweep whose level barrier cannot complete (CHIP_GS_TEST_DROP: the launch behaves as if it were not
    co-resident) times out, raises the solve's non-finite flag and leaves the barrier words dirty: the solve is repeated
    on the per-level launches, the words are cleared, and the handle keeps to the per-level launches afterwards -- every
    later solve is right (round 5's advisor finding: stale words poisoned every later persistent sweep)"""
    pr = problems.random_qp(12000, 24000, band=30, seed=4)
    ks, ko = _check_update_and_solve(hip, oracle, pr, nrhs=1)
    assert hip.debug_counter(ks, "gsweep_launches") > 0 and hip.debug_counter(ks, "gsweep_recoveries") == 0
    rng = np.random.default_rng(5)
    try:
        hip.debug_set_switch("CHIP_GS_TEST_DROP", 1)
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ks.setrhs(rx, rz)
        ko.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)  # (timed out inside, repeated without the persistent launches)
        ok, xo, zo = ko.solve()
        assert ok and relerr(np.concatenate([x, z]), np.concatenate([xo, zo])) <= TOL
        assert hip.debug_counter(ks, "gsweep_recoveries") == 1
    finally:
        hip.debug_set_switch("CHIP_GS_TEST_DROP", None)
    n0 = hip.debug_counter(ks, "gsweep_launches")
    for _ in range(2):  # later solves: per-level launches only, right answers
        rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
        ks.setrhs(rx, rz)
        ko.setrhs(rx, rz)
        x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
        assert ks.solve(x, z)
        ok, xo, zo = ko.solve()
        assert ok and relerr(np.concatenate([x, z]), np.concatenate([xo, zo])) <= TOL
    assert hip.debug_counter(ks, "gsweep_launches") == n0


@pytest.mark.parametrize("seed", range(10))
def test_ldl_structure_fuzz_with_supernodes(hip, oracle, seed):
    """random quasidefinite matrices built to produce all kinds of tops -- banded parts (long chains),
    dense blocks (wide supernodes), random couplings (side subtrees hanging off chains, several unit
    levels, narrow last blocks) -- factored and solved through chain supernodes when the analysis finds
    them: D and the solution against the oracle with the same permutation"""
    rng = np.random.default_rng(1000 + seed)
    n1 = int(rng.integers(300, 1500))
    n2 = int(rng.integers(300, 2500))
    n = n1 + n2
    rs = np.random.RandomState(seed)
    band = int(rng.integers(2, 40))
    diags = [rng.standard_normal(n1 - k) * 0.3 for k in range(1, band)]
    H = sp.diags(diags, list(range(1, band)), shape=(n1, n1), format="csc")
    H = H + H.T + sp.diags(rng.uniform(2.0, 4.0, n1) * band)
    B = sp.random(n2, n1, density=float(rng.uniform(0.5, 4.0)) / n1, random_state=rs, format="csc")
    nblk = int(rng.integers(0, 4))  # dense blocks in the (2,2) part: PSD-like cone blocks
    G = sp.diags(rng.uniform(0.5, 2.0, n2)).tolil()
    pos = 0
    for _ in range(nblk):
        w = int(rng.integers(20, 180))
        if pos + w > n2:
            break
        M = rng.standard_normal((w, w))
        G[pos:pos + w, pos:pos + w] = M @ M.T / w + np.eye(w)
        pos += w + int(rng.integers(0, 50))
    K = sp.triu(sp.bmat([[H, B.T], [B, -G.tocsc()]], format="csc"), format="csc")
    K.sort_indices()
    ds = np.array([1] * n1 + [-1] * n2, dtype=np.int8)
    Kc = hip.CscMatrix.from_scipy(K)
    f = hip.HipDirectLDLSolver(Kc, ds)
    assert f.refactor()
    o = oracle.QDLDL(n, Kc.colptr, Kc.rowval, Kc.nzval, perm=f.perm, Dsigns=ds, logical=True,
                     regularize_eps=1e-13, regularize_delta=2e-7)
    assert o.refactor()
    Lp, Li, Lx, D, Dinv = f.factors()
    assert relerr(D, o.D) <= 1e-9
    assert f.linear_solver_info().positive_inertia == o.positive_inertia == n1
    for _ in range(2):
        b = rng.standard_normal(n)
        x = np.zeros(n)
        f.solve(None, x, b)
        assert relerr(x, o.solve(b)) <= 1e-7


def _arrow_of_bands(rng, nblocks, nrows_couple):
    """several banded blocks of different widths joined by a few dense-ish coupling rows: the tops of
    the blocks are separate chains that merge into one (supernodes on several unit levels, ancestor
    updates between them)"""
    blocks, sizes = [], []
    for _ in range(nblocks):
        nb = int(rng.integers(400, 1200))
        band = int(rng.integers(8, 40))
        diags = [rng.standard_normal(nb - k) * 0.3 for k in range(1, band)]
        H = sp.diags(diags, list(range(1, band)), shape=(nb, nb), format="csc")
        blocks.append(H + H.T + sp.diags(rng.uniform(2.0, 4.0, nb) * band))
        sizes.append(nb)
    n1 = sum(sizes)
    H = sp.block_diag(blocks, format="csc")
    n2 = nrows_couple
    B = sp.random(n2, n1, density=0.05, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csc")
    G = sp.diags(rng.uniform(0.5, 2.0, n2))
    K = sp.triu(sp.bmat([[H, B.T], [B, -G]], format="csc"), format="csc")
    K.sort_indices()
    return K, np.array([1] * n1 + [-1] * n2, dtype=np.int8), n1


@pytest.mark.parametrize("seed", range(4))
def test_ldl_supernode_trees(hip, oracle, seed):
    rng = np.random.default_rng(77 + seed)
    K, ds, n1 = _arrow_of_bands(rng, int(rng.integers(3, 6)), int(rng.integers(40, 400)))
    n = K.shape[0]
    Kc = hip.CscMatrix.from_scipy(K)
    f = hip.HipDirectLDLSolver(Kc, ds)
    assert len(f.supernodes()) >= 2
    assert f.refactor()
    o = oracle.QDLDL(n, Kc.colptr, Kc.rowval, Kc.nzval, perm=f.perm, Dsigns=ds, logical=True,
                     regularize_eps=1e-13, regularize_delta=2e-7)
    assert o.refactor()
    Lp, Li, Lx, D, Dinv = f.factors()
    assert relerr(D, o.D) <= 1e-9
    b = rng.standard_normal(n)
    x = np.zeros(n)
    f.solve(None, x, b)
    assert relerr(x, o.solve(b)) <= 1e-7
