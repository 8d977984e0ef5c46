"""CPU-side (-m "not gpu") tests of the host logic: C-ABI exports, AMD ordering,
KKT assembly + LDLDataMap against the oracle, symbolic analysis against the
oracle's etree / column counts under the same permutation."""
import re
import os

import numpy as np
import pytest

from tests import problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(hip):
    L = hip.lib()
    for name, least in (("clarabel_hip.h", 35), ("clarabel_hip_testing.h", 2)):  # (the in-tree build has the test hooks)
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
        syms = sorted(set(re.findall(r"\b(chip_[a-z_A-Z0-9]+)\s*\(", hdr)))
        assert len(syms) >= least
        for s in syms:
            assert hasattr(L, s), s


def test_shipped_library_exports_the_abi_and_no_test_hooks(hip):
    """the library as it ships (make ship: TESTING=0, libclarabel_hip_ship.so): every symbol of include/clarabel_hip.h,
    none of include/clarabel_hip_testing.h, and a call that needs no GPU (chip_settings_default) answers like the test
    build's"""
    import ctypes as C
    path = hip.SHIP_LIB_PATH
    assert os.path.exists(path), "run __graft_entry__.build() (make ship)"
    L = C.CDLL(path)

    def syms(name):
        hdr = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", name)).read(), flags=re.S)
        return sorted(set(re.findall(r"\b(chip_[a-z_A-Z0-9]+)\s*\(", hdr)))
    for s_ in syms("clarabel_hip.h"):
        assert hasattr(L, s_), s_
    for s_ in syms("clarabel_hip_testing.h"):
        assert not hasattr(L, s_), "test hook %s in the shipped library" % s_
    a, b = hip.Settings(), hip.Settings()
    L.chip_settings_default(C.byref(a))
    hip.lib().chip_settings_default(C.byref(b))
    assert bytes(a) == bytes(b)


@pytest.mark.gpu
def test_shipped_library_runs_the_smoke_problem():
    """one small KKT update + solve on cuda:0 THROUGH the shipped library (a fresh interpreter with CLARABEL_HIP_LIB set),
    checked against the oracle like __graft_entry__.smoke()"""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import __graft_entry__ as g; pkg = g.load_package(); "
            "assert pkg.LIB_PATH.endswith('libclarabel_hip_ship.so'); assert not hasattr(pkg.lib(), 'chip_debug_set_switch'); "
            "g.smoke(); print('ship smoke ok')") % ROOT
    env = dict(os.environ, CLARABEL_HIP_LIB=os.path.join(ROOT, "clarabel.rs_amd", "libclarabel_hip_ship.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ship smoke ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_switches_are_parsed_once_and_settable_by_name(hip, monkeypatch):
    """the CHIP_* diagnostic switches: read from the environment when a handle is created (csrc/switches.hpp), set or
    cleared by name through the test hook, unknown names refused; no launch loop calls getenv"""
    with pytest.raises(Exception):
        hip.debug_set_switch("CHIP_NO_SUCH_SWITCH", "1")
    pr = problems.random_qp(3000, 6000, band=30, seed=2)
    assert len(_mk(hip, pr).supernodes()) > 0
    hip.debug_set_switch("CHIP_NO_SNODE", "1")
    try:
        assert len(_mk(hip, pr).supernodes()) == 0
    finally:
        hip.debug_set_switch("CHIP_NO_SNODE")
    assert len(_mk(hip, pr).supernodes()) > 0
    monkeypatch.setenv("CHIP_NO_SNODE", "1")  # (the environment is read when a handle is created)
    assert len(_mk(hip, pr).supernodes()) == 0
    monkeypatch.delenv("CHIP_NO_SNODE")
    csrc = os.path.join(ROOT, "clarabel.rs_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".hpp")) and f != "switches.cpp":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_dense_blocks_of_the_top_leave_the_full_rows(hip, monkeypatch):
    """host analysis only: the Hs blocks of PSD cones whose rows were ordered together are recognised as dense diagonal
    blocks of the top (symbolic.cpp) and their off-diagonal entries leave S, the full-row copy of the top rows the
    residual reads -- every other entry stays.  (The numeric side: tests/test_gpu_parity.py.)"""
    pr = problems.chordal_sdp(8, 20, 4, 8, 9, seed=5)  # 8 cliques of PSD(20): svec blocks of 210 rows
    monkeypatch.setenv("CHIP_NO_DENSE_SYMV", "1")
    full = hip.debug_counter(_mk(hip, pr), "nnzS")
    monkeypatch.delenv("CHIP_NO_DENSE_SYMV")
    monkeypatch.setenv("CHIP_DENSE_SYMV_MIN", "1")  # (the default leaves totals below 2^20 entries alone)
    ks = _mk(hip, pr)
    left = hip.debug_counter(ks, "nnzS")
    # S holds both triangles of the top rows: a block of m rows that leaves takes m (m - 1) entries with it.  The blocks
    # found are the parts of the cones' 210 svec rows that sit in the top next to each other (~190 of them per cone)
    taken = full - left
    assert taken % 2 == 0 and taken >= 8 * 150 * 149, (full, left)
    assert left < full / 4
    monkeypatch.delenv("CHIP_DENSE_SYMV_MIN")
    assert hip.debug_counter(_mk(hip, pr), "nnzS") == full  # (below the default threshold: nothing moves)


def test_every_object_depends_on_every_header():
    """struct Switches and the kernel views are shared by all translation units: an object that is not rebuilt when a
    header changes reads the wrong fields without any diagnostic (it happened: a switch read through a stale layout).
    Every compile rule of csrc/Makefile must list $(HEADERS)."""
    mk = open(os.path.join(ROOT, "clarabel.rs_amd", "csrc", "Makefile")).read()
    assert "HEADERS  = $(wildcard *.hpp)" in mk
    rules = [ln for ln in mk.splitlines() if ln and not ln.startswith(("\t", "#")) and ".o:" in ln]
    assert len(rules) >= 6
    for ln in rules:
        assert "$(HEADERS)" in ln, ln


def test_settings_defaults(hip):
    s = hip.Settings.default()
    # settings.rs:139-181
    assert s.static_regularization_enable == 1 and s.static_regularization_constant == 1e-8
    assert s.static_regularization_proportional == np.finfo(float).eps ** 2
    assert s.dynamic_regularization_eps == 1e-13 and s.dynamic_regularization_delta == 2e-7
    assert s.iterative_refinement_reltol == 1e-13 and s.iterative_refinement_abstol == 1e-12
    assert s.iterative_refinement_max_iter == 10 and s.iterative_refinement_stop_ratio == 5.0
    assert s.amd_dense_scale == 1.5


def _host_only(hip):
    return hip.Settings.default(device=hip.DEVICE_HOST_ONLY)


def test_no_cpu_fallback(hip):
    """numeric entry points refuse to run without a GPU instead of falling back"""
    n, Ap, Ai, Ax = 4, [0, 1, 3, 6, 8], [0, 0, 1, 0, 1, 2, 2, 3], [8., -3., 8., 2., -1., 8., -1., 1.]
    K = hip.CscMatrix(n, n, Ap, Ai, Ax)
    f = hip.HipDirectLDLSolver(K, [1] * 4, _host_only(hip))
    with pytest.raises(hip.ChipError) as e:
        f.refactor()
    assert e.value.code == hip.ERR_NO_DEVICE
    if hip.device_count() == 0:
        with pytest.raises(hip.ChipError) as e:
            hip.HipDirectLDLSolver(K, [1] * 4)
        assert e.value.code == hip.ERR_NO_DEVICE


def test_structure_errors(hip):
    # qdldl.rs:213-228 / test.rs:285-318
    st = _host_only(hip)
    K = hip.CscMatrix(3, 3, [0, 3, 6, 9], [0, 1, 2] * 3, [1., 2., 1., 3., 3., 4., 5., 6., 7.])
    with pytest.raises(hip.ChipError) as e:
        hip.HipDirectLDLSolver(K, [1] * 3, st)
    assert e.value.code == hip.ERR_NOT_TRIU
    K = hip.CscMatrix(3, 3, [0, 1, 1, 4], [0, 0, 1, 2], [1., 5., 6., 7.])
    with pytest.raises(hip.ChipError) as e:
        hip.HipDirectLDLSolver(K, [1] * 3, st)
    assert e.value.code == hip.ERR_EMPTY_COLUMN
    K = hip.CscMatrix(4, 4, [0, 1, 3, 6, 8], [0, 0, 1, 0, 1, 2, 2, 3], [8., -3., 8., 2., -1., 8., -1., 1.])
    for bad in ([3, 0, 2, 0], [4, 0, 2, 1]):  # test.rs:37-47
        with pytest.raises(hip.ChipError) as e:
            hip.HipDirectLDLSolver(K, [1] * 4, st, perm=bad)
        assert e.value.code == hip.ERR_BAD_PERM


def test_amd_4x4(hip):
    """qdldl/test.rs:123-129 pins crate amd to perm=[3,0,1,2] on this matrix.  Our AMD is an
    independent implementation: require a valid permutation with the SAME fill (zero here)."""
    Ap, Ai = [0, 1, 3, 6, 8], [0, 0, 1, 0, 1, 2, 2, 3]
    perm, iperm, _ = hip.amd_order(4, Ap, Ai)
    assert sorted(perm) == [0, 1, 2, 3] and all(iperm[perm[k]] == k for k in range(4))
    # node 3 has degree 1: any minimum-degree ordering starts with it, as the reference's does
    assert perm[0] == 3


def _fill(oracle, n, Ap, Ai, perm):
    f = oracle.QDLDL(n, Ap, Ai, np.ones(len(Ai)), perm=perm, logical=True)
    return f.nnzL


def _rand_sym(rng, n, density):
    import scipy.sparse as sp
    M = sp.random(n, n, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csc")
    M = sp.triu(M + M.T, 1) + sp.eye(n)
    M = M.tocsc()
    M.sort_indices()
    return M.indptr.astype(np.int64), M.indices.astype(np.int64)


@pytest.mark.parametrize("n,density", [(60, 0.08), (300, 0.02), (1500, 0.003)])
def test_amd_quality_random(hip, oracle, n, density):
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    rng = np.random.default_rng(n)
    Ap, Ai = _rand_sym(rng, n, density)
    perm, iperm, info = hip.amd_order(n, Ap, Ai)
    assert sorted(perm) == list(range(n))
    fill_amd = _fill(oracle, n, Ap, Ai, perm)
    fill_nat = _fill(oracle, n, Ap, Ai, np.arange(n))
    M = sp.csc_matrix((np.ones(len(Ai)), Ai, Ap), shape=(n, n))
    rcm = reverse_cuthill_mckee((M + M.T).tocsr(), symmetric_mode=True)
    fill_rcm = _fill(oracle, n, Ap, Ai, rcm)
    assert fill_amd <= fill_nat and fill_amd <= 1.05 * fill_rcm
    # amd::Info.lnz analogue is an upper bound on (and close to) the true fill
    assert info[0] >= fill_amd * 0.999 and info[0] <= 1.5 * fill_amd + n


@pytest.mark.parametrize("n,density", [(200, 0.03), (800, 0.01), (500, 0.08), (3000, 0.002)])
def test_amd_rescan_skip_is_exact(hip, n, density, monkeypatch):
    """amd_order.cpp skips the rescan of a member's variable list when the member was in the previous pivot's
    element and nothing that touches its list has happened since; with CHIP_AMD_RESCAN every member rescans
    (the textbook degree update): the two orderings must be identical"""
    rng = np.random.default_rng(n)
    Ap, Ai = _rand_sym(rng, n, density)
    p1, ip1, _ = hip.amd_order(n, Ap, Ai)
    monkeypatch.setenv("CHIP_AMD_RESCAN", "1")
    p2, ip2, _ = hip.amd_order(n, Ap, Ai)
    assert np.array_equal(p1, p2) and np.array_equal(ip1, ip2)


def test_amd_arrow_and_dense_rows(hip, oracle):
    """an arrow matrix: the dense row/column must be ordered last => zero fill"""
    n = 2000
    cols = np.arange(n)
    Ap = np.concatenate([[0], np.cumsum(np.where(cols == 0, 1, 2))]).astype(np.int64)
    Ai = []
    for c in range(n):
        Ai += [0, c] if c > 0 else [0]
    perm, _, _ = hip.amd_order(n, Ap, np.array(Ai, dtype=np.int64))
    assert perm[-1] == 0
    assert _fill(oracle, n, Ap, Ai, perm) == n - 1


CASES = {
    "basic_qp": lambda: problems.basic_qp(),
    "qp_small": lambda: problems.random_qp(300, 600, band=10, seed=1),
    "socp_small": lambda: problems.portfolio_socp(4, 12, seed=3),
    "socp_dense_soc": lambda: dict(problems.portfolio_socp(3, 3, seed=4)),  # SOC(4): dense Hs blocks
    "batched": lambda: problems.batched_socp(5, 20, 2, seed=100),
    "sdp": lambda: problems.chordal_sdp(4, 4, 2, 2, 6, seed=5),
}


def _mk(hip, pr, settings=None, perm=None):
    P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
    A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
    return hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"], settings=settings or hip.Settings.default(
        device=hip.DEVICE_HOST_ONLY), perm=perm)


@pytest.mark.parametrize("name", sorted(CASES))
def test_kkt_assembly_matches_oracle(hip, oracle, name):
    """same CSC arrays and the same LDLDataMap as kkt_assembly.rs / the oracle restatement"""
    pr = CASES[name]()
    ks = _mk(hip, pr)
    cones = oracle.Cones(pr["cones"])
    Ko = oracle.assemble_kkt(pr["n"], pr["m"], pr["P"], pr["A"], cones, "triu")
    K = ks.kkt_matrix()
    assert ks.N == Ko.N and ks.nnzK == Ko.nnz and ks.p == cones.pdim and ks.nHs == cones.nblockvals
    assert np.array_equal(K.colptr.astype(np.int64), Ko.colptr)
    assert np.array_equal(K.rowval.astype(np.int64), Ko.rowval)
    assert np.array_equal(K.nzval, Ko.nzval)
    mp = ks.maps()
    assert np.array_equal(mp["P"], Ko.map("P", len(pr["P"][1])))
    assert np.array_equal(mp["A"], Ko.map("A", len(pr["A"][1])))
    assert np.array_equal(mp["Hsblocks"], Ko.map("Hs", cones.nblockvals))
    assert np.array_equal(mp["diagP"], Ko.map("diagP", pr["n"]))
    assert np.array_equal(mp["diag_full"], Ko.map("diag_full", Ko.N))
    kso = oracle.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones)
    assert np.array_equal(mp["dsigns"], kso.dsigns)


@pytest.mark.parametrize("name", sorted(CASES))
def test_symbolic_matches_oracle(hip, oracle, name):
    """etree, column counts and the pattern of L agree with the reference algorithm
    (qdldl.rs:433-464 + logical _factor) run under the engine's final permutation; the
    level-major re-sort is a topological order of the SAME tree with the SAME fill."""
    pr = CASES[name]()
    ks = _mk(hip, pr)
    K = ks.kkt_matrix()
    perm = ks.perm
    N = ks.N
    assert sorted(perm) == list(range(N))
    f = oracle.QDLDL(N, K.colptr, K.rowval, K.nzval, perm=perm, logical=True)
    et, Lp, Li, lv = ks.symbolic()
    assert np.array_equal(et, f.etree)  # -1 == QDLDL_UNKNOWN
    assert np.array_equal(np.diff(Lp), f.Lnz)
    assert np.array_equal(Lp, f.Lp) and np.array_equal(Li, f.Li)
    info = ks.linear_solver_info()
    assert info.nnzL == f.nnzL and info.nnzA == f.nnzA and info.name == b"hip"
    # level sets: parents strictly above children
    level = lv
    assert len(level) == N and info.n_levels == level.max() + 1
    for j in range(N):
        if et[j] >= 0:
            assert level[et[j]] > level[j] and et[j] > j
    # fill never exceeds that of the raw AMD order: the level-major re-sort is an equivalent reordering,
    # and re-sequencing the members of a chain (symbolic.cpp "shallower tree, same fill") can only drop
    # fill edges the raw order created among them
    p0, _, _ = hip.amd_order(N, K.colptr, K.rowval)
    f0 = oracle.QDLDL(N, K.colptr, K.rowval, K.nzval, perm=p0, logical=True)
    assert f.nnzL <= f0.nnzL
    lv0 = np.zeros(N, dtype=np.int64)
    for j in range(N):
        if f0.etree[j] >= 0:
            lv0[f0.etree[j]] = max(lv0[f0.etree[j]], lv0[j] + 1)
    assert level.max() <= lv0.max()


@pytest.mark.parametrize("name", sorted(CASES))
def test_amd_rescan_skip_is_exact_on_kkt(hip, name, monkeypatch):
    pr = CASES[name]()
    k1 = _mk(hip, pr)
    monkeypatch.setenv("CHIP_AMD_RESCAN", "1")
    k2 = _mk(hip, pr)
    assert np.array_equal(np.asarray(k1.perm), np.asarray(k2.perm))


@pytest.mark.parametrize("name", sorted(CASES))
def test_threaded_analysis_same_symbolic(hip, name, monkeypatch):
    """the big analysis passes run on several std::threads, each owning a range of destination keys and
    scanning the source in order: the result must not depend on the thread count (forced on for these
    small cases with CHIP_HOST_PAR_MIN=0)"""
    pr = CASES[name]()
    got = []
    for threads in (1, 3, 8):
        monkeypatch.setenv("CHIP_HOST_THREADS", str(threads))
        monkeypatch.setenv("CHIP_HOST_PAR_MIN", "0")
        ks = _mk(hip, pr)
        info = ks.linear_solver_info()
        got.append((np.asarray(ks.perm).copy(), [np.asarray(a).copy() for a in ks.symbolic()], info.nnzL, info.n_levels))
    for g in got[1:]:
        assert np.array_equal(g[0], got[0][0])
        for a, b in zip(g[1], got[0][1]):
            assert np.array_equal(a, b)
        assert g[2:] == got[0][2:]


@pytest.mark.parametrize("which", ["identical", "ragged"])
def test_amd_by_connected_components(hip, which, monkeypatch):
    """block-diagonal KKT systems (BASELINE config 4): minimum degree runs once per DISTINCT component pattern
    (identical blocks share one ordering, distinct ones are ordered in parallel) instead of once over the whole
    graph -- a valid permutation with the same fill and the same tree depth as the whole-graph ordering"""
    if which == "identical":
        pr = problems.batched_socp(12, 60, 2, seed=100)
    else:
        pr = problems.blockdiag([problems.portfolio_socp(2, 10 + 3 * (i % 4), seed=100 + i) for i in range(9)])
    k1 = _mk(hip, pr)
    monkeypatch.setenv("CHIP_NO_COMPONENTS", "1")
    k2 = _mk(hip, pr)
    p1, p2 = np.asarray(k1.perm), np.asarray(k2.perm)
    assert sorted(p1.tolist()) == list(range(k1.N))
    i1, i2 = k1.linear_solver_info(), k2.linear_solver_info()
    assert i1.nnzL == i2.nnzL and i1.n_levels == i2.n_levels
    if which == "identical":   # every block gets the same relative order
        K = k1.kkt_matrix()
        import scipy.sparse as sp
        from scipy.sparse.csgraph import connected_components
        G = sp.csc_matrix((np.ones(len(K.rowval)), K.rowval.astype(np.int64), K.colptr.astype(np.int64)), shape=(k1.N, k1.N))
        nc, lab = connected_components(G + G.T, directed=False)
        assert nc == 12


def test_user_perm_respected_up_to_level_sort(hip, oracle, monkeypatch):
    # (chain supernodes pad their columns with explicit zeros, which nnzL counts: off for the exact comparison)
    monkeypatch.setenv("CHIP_NO_SNODE", "1")
    pr = problems.random_qp(120, 240, band=6, seed=11)
    ks0 = _mk(hip, pr)
    N = ks0.N
    rng = np.random.default_rng(0)
    user = rng.permutation(N)
    ks = _mk(hip, pr, perm=user)
    K = ks.kkt_matrix()
    f_user = oracle.QDLDL(N, K.colptr, K.rowval, K.nzval, perm=user, logical=True)
    assert ks.linear_solver_info().nnzL == f_user.nnzL


def test_portfolio_structure_is_shallow(hip):
    """the block-arrow KKT of config 3 must come out with a handful of levels and ~1x fill
    (SURVEY.md 7 'hard parts'): that is what makes the level-scheduled GPU path viable."""
    pr = problems.portfolio_socp(20, 50, seed=3)
    ks = _mk(hip, pr)
    info = ks.linear_solver_info()
    assert info.n_levels <= 6  # (8 before the chain re-sequencing of symbolic.cpp: u, v pivoted before the last cone rows)
    assert info.nnzL <= 1.2 * info.nnzA


def test_auto_select_rule(hip):
    """ldl_auto_select (ldlsolvers/auto.rs:62-87): flops / nnz(L) < 40 -> simplicial ('qdldl'), else
    supernodal ('faer'); fed with the statistics of our own AMD ordering"""
    s = hip.Settings.default()
    assert s.linesearch_backtrack_step == 0.8 and s.min_terminate_step_length == 1e-4  # settings.rs:96-104
    pr = problems.portfolio_socp(20, 50, seed=3)      # block arrow: almost no fill
    info = _mk(hip, pr).linear_solver_info()
    assert hip.auto_select(info.amd_lnz, info.amd_ndiv, info.amd_nmultsubs_ldl) == "qdldl"
    pr = problems.chordal_sdp(2, 30, 3, 1, 6, seed=1)  # dense 465-wide PSD blocks
    info = _mk(hip, pr).linear_solver_info()
    assert hip.auto_select(info.amd_lnz, info.amd_ndiv, info.amd_nmultsubs_ldl) == "faer"
    assert hip.auto_select(100.0, 1000.0, 2999.0) == "qdldl" and hip.auto_select(100.0, 1000.0, 3000.0) == "faer"


def test_chain_supernodes_pad_to_dense_trapezoids(hip, monkeypatch):
    """symbolic.cpp chain supernodes: each is a parent chain of top columns; every member's structure
    is exactly [later members..., struct(last)...] (ascending), and the padded pattern is a superset of
    the unpadded one (CHIP_NO_SNODE) with the same elimination tree and permutation"""
    from tests import problems
    st = _host_only(hip)
    found = 0
    for pr in (problems.random_qp(3000, 6000, band=30, seed=2), problems.chordal_sdp(8, 12, 3, 4, 7, seed=5)):
        P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
        A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
        ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"], settings=st)
        et, Lp, Li, lv = ks.symbolic()
        sns = ks.supernodes()
        monkeypatch.setenv("CHIP_NO_SNODE", "1")
        k0 = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"], settings=st)
        monkeypatch.delenv("CHIP_NO_SNODE")
        et0, Lp0, Li0, _ = k0.symbolic()
        assert k0.supernodes() == []
        assert (ks.perm == k0.perm).all() and (et == et0).all()
        for j in range(ks.N):  # superset, column by column
            assert set(Li0[Lp0[j]:Lp0[j + 1]]) <= set(Li[Lp[j]:Lp[j + 1]])
        seen = set()
        for cols in sns:
            found += 1
            w, last = len(cols), cols[-1]
            assert w >= 16 and cols[0] >= ks.NF
            below = Li[Lp[last]:Lp[last + 1]]
            for t in range(w):
                assert cols[t] not in seen
                seen.add(int(cols[t]))
                if t + 1 < w:
                    assert et[cols[t]] == cols[t + 1]
                assert list(Li[Lp[cols[t]]:Lp[cols[t] + 1]]) == list(cols[t + 1:]) + list(below)
        if not sns:
            assert (Lp == Lp0).all()
    assert found > 0


def test_integration_md_matches_header():
    """INTEGRATION.md holds the Rust side of the boundary as SOURCE (no toolchain here to compile it): every
    #[repr(C)] struct and every extern "C" prototype in it is checked against include/clarabel_hip.h --
    field order / types, function names, parameter counts and types -- so the document cannot drift."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "clarabel_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)

    def c_type(t):
        t = " ".join(t.replace("*", " * ").split())
        const = t.startswith("const ")
        t = t[6:] if const else t
        stars = t.count("*")
        base = t.replace("*", "").strip()
        base = {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "int8_t": "i8", "uint8_t": "u8", "double": "f64",
                "char": "c_char", "void": "void", "chip_settings": "ChipSettings", "chip_info": "ChipInfo"}.get(base, base)
        if base.startswith("chip_") or base == "void":
            base = "c_void"  # opaque handles
        out = base
        for k in range(stars):
            out = ("*const " if (const and k == 0) else "*mut ") + out
        return out

    def rust_type(t):
        return " ".join(t.split())

    # --- structs
    for rs_name, c_name in (("ChipSettings", "chip_settings"), ("ChipInfo", "chip_info")):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % rs_name, md, flags=re.S).group(1)
        body = re.sub(r"//[^\n]*", "", body)
        rs_fields = [(a.strip(), rust_type(b)) for a, b in re.findall(r"(\w+)\s*:\s*([^,]+),", body)]
        cbody = re.search(r"typedef struct \{([^}]*)\} %s;" % c_name, hdr).group(1)
        c_fields = []
        for decl in cbody.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"([\w ]+?)\s+([\w\[\], ]+)$", decl)
            ty, names = m.group(1), m.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                if arr:
                    c_fields.append((arr.group(1), "[%s; %s]" % (c_type(ty), arr.group(2))))
                else:
                    c_fields.append((nm, c_type(ty)))
        assert rs_fields == c_fields, (rs_name, rs_fields, c_fields)
    # --- prototypes
    protos = {}
    for m in re.finditer(r"(?:int32_t|void|const char \*)\s*(chip_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        ret = "i32" if hdr[m.start():m.start() + 7] == "int32_t" else "()"
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.match(r"(.*?)(\w+)\[(\w*)\]$", a)  # T name[N] decays to a pointer
                if arr:
                    params.append(c_type(arr.group(1).strip() + " *"))
                else:
                    params.append(c_type(re.match(r"(.*?)(\w+)$", a).group(1).strip()))
        protos[name] = (params, ret)
    seen = 0
    for block in re.findall(r'extern "C" \{(.*?)\n\}', md, flags=re.S):
        block = re.sub(r"//[^\n]*", "", block)
        for m in re.finditer(r"fn (chip_\w+)\s*\((.*?)\)\s*(->\s*i32)?\s*;", block, flags=re.S):
            name, args = m.group(1), " ".join(m.group(2).split())
            params = [rust_type(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
            assert name in protos, "INTEGRATION.md binds %s which the header does not declare" % name
            assert (params, "i32" if m.group(3) else "()") == protos[name], (name, params, protos[name])
            seen += 1
    assert seen >= 35
