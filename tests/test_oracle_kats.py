"""Pins the CPU oracle against every known-answer test the reference holds for
the KKT path.  Citations are relative to /root/reference/src."""
import numpy as np
import pytest

UNKNOWN = -1


def mat4():
    # qdldl/test.rs:5-21
    return 4, [0, 1, 3, 6, 8], [0, 0, 1, 0, 1, 2, 2, 3], [8., -3., 8., 2., -1., 8., -1., 1.]


def test_invperm(oracle):
    # qdldl/test.rs:31-47
    oracle.invperm([3, 0, 2, 1])
    with pytest.raises(ValueError):
        oracle.invperm([3, 0, 2, 0])
    with pytest.raises(ValueError):
        oracle.invperm([4, 0, 2, 1])


def test_permute(oracle):
    # qdldl/test.rs:49-61
    perm = [3, 0, 2, 1]
    b = [1., 2., 3., 4.]
    x = oracle.permute(b, perm)
    assert list(x) == [4., 1., 3., 2.]
    assert list(oracle.ipermute(x, perm)) == b


def test_solve_from_factors(oracle):
    # qdldl/test.rs:63-100 (exact equality, as in the reference)
    Lp = [0, 2, 4, 5, 5]
    Li = [1, 2, 2, 3, 3]
    Lx = [1., 2., 1., 7., -3.]
    dinv = [0.25, -1.0, -0.5, 1.0]
    x = [-3., 2., 1., 4.]
    assert list(oracle.lsolve(Lp, Li, Lx, [-3., -1., -3., 15.])) == x
    assert list(oracle.ltsolve(Lp, Li, Lx, [1., 31., -11., 4.])) == x
    assert list(oracle.solve_factors(Lp, Li, Lx, dinv, [4., -27., -1., -279.])) == x


def test_etree(oracle):
    # qdldl/test.rs:102-121
    n, Ap, Ai, _ = mat4()
    et, _ = oracle.etree(n, Ap, Ai)
    assert list(et) == [1, 2, 3, UNKNOWN]


def test_permute_symmetric(oracle):
    # qdldl/test.rs:131-164
    n, Ap, Ai, Ax = mat4()
    Pc, Pr, Pv, mp = oracle.permute_symmetric(n, Ap, Ai, Ax, [0, 1, 2, 3])
    assert list(Pc) == Ap and list(Pr) == Ai and list(Pv) == Ax
    assert list(mp) == list(range(len(Ax)))
    Ax2 = [float(i + 1) for i in range(len(Ax))]
    iperm = oracle.invperm([2, 3, 0, 1])
    Pc, Pr, Pv, _ = oracle.permute_symmetric(n, Ap, Ai, Ax2, iperm)
    assert list(Pc) == [0, 1, 3, 5, 8]
    assert list(Pr) == [0, 0, 1, 2, 0, 2, 3, 0]
    assert list(Pv) == [6.0, 7.0, 8.0, 1.0, 4.0, 2.0, 3.0, 5.0]


@pytest.mark.parametrize("perm", [[0, 1, 2, 3], [3, 0, 1, 2], [3, 0, 2, 1]])
def test_solve_basic(oracle, perm):
    # qdldl/test.rs:194-230 ([3,0,1,2] is the AMD answer of test_amd :123-129)
    n, Ap, Ai, Ax = mat4()
    f = oracle.QDLDL(n, Ap, Ai, Ax, perm=perm)
    x = f.solve([20.0, -22.0, 32.0, -7.0])
    assert np.max(np.abs(x - np.array([1., -2., 3., -4.]))) <= 1e-8


def test_solve_logical(oracle):
    # qdldl/test.rs:232-264
    n, Ap, Ai, Ax = mat4()
    f = oracle.QDLDL(n, Ap, Ai, Ax, logical=True)
    with pytest.raises(RuntimeError):
        f.solve([20.0, -22.0, 32.0, -7.0])
    assert f.refactor()
    x = f.solve([20.0, -22.0, 32.0, -7.0])
    assert np.max(np.abs(x - np.array([1., -2., 3., -4.]))) <= 1e-8


def test_bad_numeric_pivot(oracle):
    # qdldl/test.rs:266-283
    n, Ap, Ai, Ax = mat4()
    A0 = list(Ax)
    A0[0] = 0.
    with pytest.raises(ValueError, match="ZeroPivot"):
        oracle.QDLDL(n, Ap, Ai, A0, regularize_enable=False)
    A1 = list(Ax)
    A1[-1] = 0.
    # the reference runs this case under its AMD ordering, perm=[3,0,1,2] (test_amd, :123-129),
    # which makes the zeroed A[3,3] the first pivot
    with pytest.raises(ValueError, match="ZeroPivot"):
        oracle.QDLDL(n, Ap, Ai, A1, perm=[3, 0, 1, 2], regularize_enable=False)


def test_structure_errors(oracle):
    # qdldl/test.rs:285-318 (dense 3x3 => not triu; zero column)
    Ap = [0, 3, 6, 9]
    Ai = [0, 1, 2] * 3
    Ax = [1., 2., 1., 3., 3., 4., 5., 6., 7.]
    with pytest.raises(ValueError, match="NotUpperTriangular"):
        oracle.QDLDL(3, Ap, Ai, Ax, logical=True)
    # [1 0 5; 0 0 6; 1 0 7] in CSC: col1 empty (and col0 has a tril entry -> triu check first
    # in the reference: check order is square, triu, empty col (qdldl.rs:213-228))
    Ap = [0, 1, 1, 4]
    Ai = [0, 0, 1, 2]
    Ax = [1., 5., 6., 7.]
    with pytest.raises(ValueError, match="EmptyColumn"):
        oracle.QDLDL(3, Ap, Ai, Ax, logical=True)


def test_faer_kat(oracle):
    # ldlsolvers/faer_ldl.rs:352-404 -- boundary KAT shared by all engines
    colptr = [0, 1, 2, 4, 6, 8, 10]
    rowval = [0, 1, 0, 2, 1, 3, 0, 4, 1, 5]
    nzval = [1.0, 2.0, 1.0, -1.0, 1.0, -2.0, -1.0, -3.0, -1.0, -4.0]
    ds = [1, 1, -1, -1, -1, -1]
    b = [1., 2., 3., 4., 5., 6.]
    for perm in ([0, 1, 2, 3, 4, 5], [5, 3, 1, 4, 2, 0]):
        f = oracle.QDLDL(6, colptr, rowval, nzval, perm=perm, Dsigns=ds, logical=True,
                         regularize_eps=1e-13, regularize_delta=2e-7)
        tri = f.triuA
        mp = f.AtoPAPt
        assert all(nzval[i] == tri[2][mp[i]] for i in range(len(nzval)))
        assert f.refactor()
        x = f.solve(b)
        xs = [1.0, 0.9090909090909091, -2.0, -1.5454545454545454, -2.0, -1.7272727272727275]
        assert np.max(np.abs(x - xs)) < 1e-10
        f.update_values([9], [-10.0])
        assert f.refactor()
        x = f.solve(b)
        xs = [1.0, 1.3076923076923077, -2.0, -1.346153846153846, -2.0, -0.7307692307692306]
        assert np.max(np.abs(x - xs)) < 1e-10
        f.offset_values([1, 2], 3., [1, -1])
        f.scale_values([1, 2], 2.)
        tri = f.triuA
        assert tri[2][mp[1]] == (2.0 + 3.) * 2 and tri[2][mp[2]] == (1.0 - 3.) * 2


def test_symv_quadform(oracle):
    # algebra/tests/matrix.rs:4-14,250-286
    Ap, Ai, Ax = [0, 1, 3, 6, 8], [0, 0, 1, 0, 1, 2, 2, 3], [4., -3., 8., 7., -1., 2., -3., 1.]
    x = [1., 2., -3., -4.]
    y = [0., 1., -1., 2.]
    out = oracle.symv(4, Ap, Ai, Ax, y, x, -2., 3.)
    assert list(out) == [46.0, -29.0, -25.0, -4.0]
    # tril storage of the same matrix (A.t())
    Lp, Li, Lx = [0, 3, 5, 7, 8], [0, 1, 2, 1, 2, 2, 3, 3], [4., -3., 7., 8., -1., 2., -3., 1.]
    out = oracle.symv(4, Lp, Li, Lx, y, x, -2., 3.)
    assert list(out) == [46.0, -29.0, -25.0, -4.0]
    assert oracle.quad_form_triu(4, Ap, Ai, Ax, y, x) == 15.


def test_norms(oracle):
    # algebra/tests/vector.rs:103-180 (norm_inf NaN propagation; stable norm)
    assert oracle.norm_inf([1., -5., 3.]) == 5.
    assert np.isnan(oracle.norm_inf([1., np.nan, 3.]))
    assert oracle.norm_inf([1., np.inf, 3.]) == np.inf
    assert oracle.norm2([3., 4.]) == 5.
    assert abs(oracle.norm2([1e200, 1e200]) - np.sqrt(2) * 1e200) < 1e186
    assert oracle.norm2([0., 0.]) == 0.


# ---- KKT assembly: kkt_assembly.rs:185-355 ------------------------------------
def _dense_to_csc(M):
    M = np.asarray(M, dtype=float)
    colptr, rowval, nzval = [0], [], []
    for j in range(M.shape[1]):
        for i in range(M.shape[0]):
            if M[i, j] != 0:
                rowval.append(i)
                nzval.append(M[i, j])
        colptr.append(len(rowval))
    return colptr, rowval, nzval


def _csc_to_dense(N, colptr, rowval, nzval):
    M = np.zeros((N, N))
    for j in range(N):
        for p in range(colptr[j], colptr[j + 1]):
            M[rowval[p], j] += nzval[p]
    return M


P3 = [[1., 2., 4.], [0., 3., 5.], [0., 0., 6.]]
A63 = [[7., 0., 8.], [0., 9., 10.], [1., 2., 3.], [7., 0., 8.], [0., 9., 10.], [1., 2., 3.]]


def _assemble(oracle, specs, shape):
    cones = oracle.Cones(specs)
    K = oracle.assemble_kkt(3, 6, _dense_to_csc(P3), _dense_to_csc(A63), cones, shape)
    return K, cones


def _expected(shape, lower_right, extra=None):
    P = np.array(P3)
    A = np.array(A63)
    N = 9 if extra is None else 11
    M = np.zeros((N, N))
    M[:3, :3] = P
    M[:3, 3:9] = A.T
    M[3:9, 3:9] = lower_right
    if extra is not None:
        M[3:9, 9] = 2.
        M[3:9, 10] = 3.
        M[9, 9] = 4.
        M[10, 10] = 4.
    return M if shape == "triu" else M.T


@pytest.mark.parametrize("shape", ["triu", "tril"])
def test_kkt_assembly_nncone(oracle, shape):
    K, _ = _assemble(oracle, [(oracle.CONE_NONNEG, 6)], shape)
    v = K.nzval
    v[K.map("Hs", 6)] = -1.
    got = _csc_to_dense(9, K.colptr, K.rowval, v)
    assert np.array_equal(got, _expected(shape, -np.eye(6)))
    # exact CSC pattern == CscMatrix::from(dense) pattern (sorted rows, no explicit zeros dropped
    # except that structural entries are all nonzero here)
    cp, rv, nv = _dense_to_csc(_expected(shape, -np.eye(6)))
    assert list(K.colptr) == cp and list(K.rowval) == rv and list(v) == nv


@pytest.mark.parametrize("shape", ["triu", "tril"])
def test_kkt_assembly_expcones(oracle, shape):
    K, _ = _assemble(oracle, [(oracle.CONE_EXP, 3), (oracle.CONE_EXP, 3)], shape)
    v = K.nzval
    v[K.map("Hs", 12)] = -1.
    blk = -np.triu(np.ones((3, 3)))
    lr = np.zeros((6, 6))
    lr[:3, :3] = blk
    lr[3:, 3:] = blk
    exp = _expected(shape, lr)
    cp, rv, nv = _dense_to_csc(exp)
    assert list(K.colptr) == cp and list(K.rowval) == rv and list(v) == nv


@pytest.mark.parametrize("shape", ["triu", "tril"])
def test_kkt_assembly_socone(oracle, shape):
    K, cones = _assemble(oracle, [(oracle.CONE_SOC, 6)], shape)
    assert K.nsparse == 1 and cones.pdim == 2
    v = K.nzval
    v[K.map("v", 6, 0)] = 2.
    v[K.map("u", 6, 0)] = 3.
    v[K.map("D", 2, 0)] = 4.
    v[K.map("Hs", 6)] = -1.
    exp = _expected(shape, -np.eye(6), extra=True)
    cp, rv, nv = _dense_to_csc(exp)
    assert list(K.colptr) == cp and list(K.rowval) == rv and list(v) == nv
    # diag_full: last (triu) / first (tril) entry of each column
    df = K.map("diag_full", 11)
    assert all(K.rowval[df[j]] == j for j in range(11))


def test_kkt_missing_diag_and_signs(oracle):
    # P with a missing diagonal entry -> structural zero inserted (kkt_assembly.rs:120-121)
    P = [[1., 2., 0.], [0., 0., 5.], [0., 0., 0.]]
    cones = oracle.Cones([(oracle.CONE_ZERO, 1), (oracle.CONE_NONNEG, 2), (oracle.CONE_SOC, 3)])
    K = oracle.assemble_kkt(3, 6, _dense_to_csc(P), _dense_to_csc(A63), cones, "triu")
    df = K.map("diag_full", K.N)
    assert K.N == 9 and all(K.rowval[df[j]] == j for j in range(9))
    # nnz formula kkt_assembly.rs:38-44: nnzP(3) + n(3) - diagP(1) + nnzA(14) + Hs(1+2+6) + 0 + 0
    assert K.nnz == 3 + 3 - 1 + 14 + 9
    ks = oracle.KKTSolver(3, 6, _dense_to_csc(P), _dense_to_csc(A63), cones)
    assert list(ks.dsigns) == [1, 1, 1] + [-1] * 6


# ---- cones: NT identities (no KATs in the reference) ----------------
def _rand_soc(rng, n):
    v = rng.standard_normal(n)
    v[0] = np.linalg.norm(v[1:]) + rng.uniform(0.1, 2.0)
    return v


@pytest.mark.parametrize("dim", [3, 4, 5, 50])
def test_soc_scaling_identities(oracle, dim):
    rng = np.random.default_rng(dim)
    s, z = _rand_soc(rng, dim), _rand_soc(rng, dim)
    cones = oracle.Cones([(oracle.CONE_SOC, dim)])
    assert cones.update_scaling(s, z)
    # Hs z = s for the NT scaling (W'W z = s)
    assert np.allclose(cones.mul_Hs(z), s, rtol=1e-10, atol=1e-12)
    st = cones.state(0)
    w, eta = st["w"], st["eta"]
    J = np.diag([1.] + [-1.] * (dim - 1))
    H = eta ** 2 * (2 * np.outer(w, w) - J)
    Hs = cones.get_Hs()
    if dim <= 4:
        # dense packed triu, column major (csc/utils.rs:183-200)
        k = 0
        for col in range(dim):
            for row in range(col + 1):
                assert abs(Hs[k] - H[row, col]) < 1e-10 * max(1, abs(H[row, col]))
                k += 1
    else:
        # sparse form: H = eta^2 (D + u u' - v v')  (socone.rs:187-223)
        D = np.diag(Hs) / eta ** 2
        u, v = st["u"], st["v"]
        assert np.allclose(eta ** 2 * (D + np.outer(u, u) - np.outer(v, v)), H, rtol=1e-9, atol=1e-11)


def test_nn_scaling(oracle):
    rng = np.random.default_rng(0)
    s, z = rng.uniform(0.1, 3, 7), rng.uniform(0.1, 3, 7)
    cones = oracle.Cones([(oracle.CONE_NONNEG, 7)])
    cones.update_scaling(s, z)
    assert np.allclose(cones.get_Hs(), s / z, rtol=1e-15)
    assert np.allclose(cones.mul_Hs(z), s, rtol=1e-15)


# ---- full KKT solver: sparse-SOC expansion == dense Hs ---------------------------
def test_kktsolver_matches_dense(oracle):
    rng = np.random.default_rng(5)
    n, dims = 6, [3, 2, 7, 4]
    specs = [(oracle.CONE_ZERO, dims[0]), (oracle.CONE_NONNEG, dims[1]), (oracle.CONE_SOC, dims[2]),
             (oracle.CONE_SOC, dims[3])]
    m = sum(dims)
    A = rng.standard_normal((m, n)) * (rng.uniform(size=(m, n)) < 0.6)
    Pd = rng.standard_normal((n, n))
    Pd = np.triu(Pd @ Pd.T + n * np.eye(n))
    s = np.concatenate([np.zeros(3), rng.uniform(0.5, 2, 2), _rand_soc(rng, 7), _rand_soc(rng, 4)])
    z = np.concatenate([np.zeros(3), rng.uniform(0.5, 2, 2), _rand_soc(rng, 7), _rand_soc(rng, 4)])
    cones = oracle.Cones(specs)
    assert cones.update_scaling(s, z)
    ks = oracle.KKTSolver(n, m, _dense_to_csc(Pd), _dense_to_csc(A), cones)
    assert ks.p == 2 and ks.N == n + m + 2
    assert ks.update()
    # dense reference: K = [P A'; A -H] with H built from mul_Hs columns
    H = np.zeros((m, m))
    for j in range(m):
        e = np.zeros(m)
        e[j] = 1.
        H[:, j] = cones.mul_Hs(e)
    Pfull = Pd + np.triu(Pd, 1).T
    Kd = np.block([[Pfull, A.T], [A, -H]])
    eps = ks.regularizer
    Kreg = Kd + np.diag([eps] * n + [-eps] * m)
    rhsx, rhsz = rng.standard_normal(n), rng.standard_normal(m)
    ks.setrhs(rhsx, rhsz)
    ok, x, zz = ks.solve()
    assert ok
    ref = np.linalg.solve(Kd + np.diag([0.] * n + [-1e-30] * 3 + [0.] * (m - 3)) if False else Kreg, np.concatenate([rhsx, rhsz]))
    # IR is against the UNregularised K (directldlkktsolver.rs:255-261); zero-cone rows make Kd
    # nonsingular here since A has full column rank on those rows w.h.p. -> compare to Kd solve
    ref0 = np.linalg.solve(Kd, np.concatenate([rhsx, rhsz]))
    got = np.concatenate([x, zz])
    assert np.max(np.abs(got - ref0)) <= 1e-8 * max(1, np.max(np.abs(ref0)))
    assert ref.shape == got.shape


# ---- Exponential / Power cones (no Hs KATs in the reference; identities instead) ----------
def test_wright_omega(oracle):
    # expcone.rs:461-472
    import ctypes as C
    L = oracle.lib()
    for z in [1e-7, 1e-5, 1e-3, 1e-1, 1e1, 1e3, 1e5, 1e7, 1e9]:
        y = L.orc_wright_omega(C.c_double(z))
        assert abs(z - (y + np.log(y))) / z < 1e-9


def _dual_barrier(tag, z, alpha):
    if tag == 3:  # f*(z) = -log(z1 - z0 - z0 log(z2/-z0)) - log(-z0) - log(z2)
        l = np.log(-z[2] / z[0])
        return -np.log(-z[0] * l - z[0] + z[1]) - np.log(-z[0]) - np.log(z[2])
    phi = (z[0] / alpha) ** (2 * alpha) * (z[1] / (1 - alpha)) ** (2 - 2 * alpha)
    return -np.log(phi - z[2] ** 2) - (1 - alpha) * np.log(z[0]) - alpha * np.log(z[1])


@pytest.mark.parametrize("tag,alpha", [(3, 0.5), (4, 0.6), (4, 0.1)])
def test_nonsymmetric_cone_scalings(oracle, tag, alpha):
    from tests import problems
    rng = np.random.default_rng(int(tag * 10 + alpha * 100))
    s, z = problems.exp_interior(rng) if tag == 3 else problems.pow_interior(rng, alpha)
    cones = oracle.Cones([(tag, 3, 0, alpha)])
    assert cones.update_scaling(s, z, 0.37, 0)
    st = cones.state(0)
    # grad / H_dual are the gradient and Hessian of the dual barrier (finite differences)
    h = 1e-6
    g_fd = np.array([(_dual_barrier(tag, z + h * e, alpha) - _dual_barrier(tag, z - h * e, alpha)) / (2 * h)
                     for e in np.eye(3)])
    assert np.allclose(st["grad3"], g_fd, rtol=1e-6, atol=1e-8)
    Hd = st["Hdual"]
    H = np.array([[Hd[0], Hd[1], Hd[3]], [Hd[1], Hd[2], Hd[4]], [Hd[3], Hd[4], Hd[5]]])
    H_fd = np.zeros((3, 3))
    for j, e in enumerate(np.eye(3)):
        cp, cm = oracle.Cones([(tag, 3, 0, alpha)]), oracle.Cones([(tag, 3, 0, alpha)])
        cp.update_scaling(s, z + h * e, 1.0, 1)
        cm.update_scaling(s, z - h * e, 1.0, 1)
        H_fd[:, j] = (cp.state(0)["grad3"] - cm.state(0)["grad3"]) / (2 * h)
    assert np.allclose(H, H_fd, rtol=1e-5, atol=1e-7)
    # <grad f*(z), z> = -3 (logarithmic homogeneity, degree 3)
    assert abs(st["grad3"] @ z + 3.0) < 1e-10
    # primal-dual scaling satisfies the secant equation Hs z = s (nonsymmetric_common.rs:131-138)
    assert np.allclose(cones.mul_Hs(z), s, rtol=1e-10, atol=1e-12)
    # dual scaling: Hs = mu * H_dual
    cd = oracle.Cones([(tag, 3, 0, alpha)])
    cd.update_scaling(s, z, 0.37, 1)
    assert np.allclose(cd.state(0)["Hs3"], 0.37 * Hd, rtol=1e-14)
    assert np.allclose(cd.get_Hs(), 0.37 * Hd, rtol=1e-14)


def test_cone_step_ops_identities(oracle):
    """no reference KATs for these either: identities of the Jordan algebra / NT scaling.
    (symmetric_common.rs:53-96, socone.rs:258-302)"""
    from tests import problems
    pr = problems.portfolio_socp(3, 9, seed=21)
    cones = oracle.Cones(pr["cones"])
    s, z = pr["s"], pr["z"]
    assert cones.update_scaling(s, z)
    m = pr["m"]
    # affine_ds = lambda o lambda, and for NT scalings  <lambda, lambda> = <s, z> per cone
    ads = cones.affine_ds(s)
    off = 0
    for (tag, dim) in [(c[0], c[1]) for c in pr["cones"]]:
        if tag == oracle.CONE_NONNEG:
            assert np.allclose(ads[off:off + dim], s[off:off + dim] * z[off:off + dim], rtol=1e-12)
        elif tag == oracle.CONE_SOC:
            assert abs(ads[off] - s[off:off + dim] @ z[off:off + dim]) <= 1e-10 * abs(ads[off])
        off += dim
    # Delta s offset with ds = affine_ds equals s:  W'(lambda \ (lambda o lambda)) = W' lambda = s
    assert np.allclose(cones.ds_from_dz_offset(ads, z), s, rtol=1e-9, atol=1e-11)
    # combined shift with zero steps is -sigma*mu*e
    sh, _, _ = cones.combined_ds_shift(np.zeros(m), np.zeros(m), 0.25)
    e = np.zeros(m)
    off = 0
    for c in pr["cones"]:
        if c[0] == oracle.CONE_NONNEG:
            e[off:off + c[1]] = 1.0
        elif c[0] == oracle.CONE_SOC:
            e[off] = 1.0
        off += c[1]
    assert np.allclose(sh, -0.25 * e, atol=1e-15)
    # step length: the boundary is hit exactly at alpha (some cone residual ~ 0), interior before
    rng = np.random.default_rng(3)
    dz, ds = rng.standard_normal(m), rng.standard_normal(m)
    a = cones.step_length(dz, ds, z, s, 1e6)
    assert 0 < a < 1e6
    am, _ = cones.margins(z + 0.999 * a * dz)
    ams, _ = cones.margins(s + 0.999 * a * ds)
    assert min(am, ams) > 0
    am2, _ = cones.margins(z + 1.001 * a * dz)
    ams2, _ = cones.margins(s + 1.001 * a * ds)
    assert min(am2, ams2) < 0


def test_genpow_cone_identities(oracle):
    """GenPowerCone (genpowcone.rs): no unit KATs in the reference, so cross-checks that do not go
    through the restatement's formulas: grad = gradient of the dual barrier f*(z) (finite
    differences of barrier_dual), <grad, z> = -(dim1 + 1), Hs = mu * Hessian (finite differences
    of grad), and the KKT solve through the rank-3 sparse expansion [q, r, p] (datamaps.rs:227-343)
    equal to a dense solve with the unexpanded Hs"""
    rng = np.random.default_rng(12)
    d1, d2 = 4, 3
    a = rng.uniform(0.3, 1.0, d1)
    a /= a.sum()
    a[-1] = 1.0 - a[:-1].sum()
    cones_spec = [(oracle.CONE_GENPOW, d1, d2, list(a))]
    m = d1 + d2

    def interior():
        u = rng.uniform(0.8, 2.0, d1)
        w = rng.standard_normal(d2)
        w *= 0.4 * np.prod((u / a) ** a) / np.linalg.norm(w)
        return np.concatenate([u, w])

    z, s = interior(), interior()
    mu = 0.37

    def grad_at(zz):
        c = oracle.Cones(cones_spec)
        assert c.update_scaling(s, zz, mu, 1)
        sh, _, _ = c.combined_ds_shift(np.zeros(m), np.zeros(m), 1.0)  # shift = grad * sigma_mu
        return sh

    def fstar(zz):  # dual barrier (genpowcone.rs:333-356) written out independently
        phi = np.prod((zz[:d1] / a) ** (2 * a))
        return -np.log(phi - zz[d1:] @ zz[d1:]) - np.sum((1 - a) * np.log(zz[:d1]))

    g = grad_at(z)
    h = 1e-6
    gfd = np.array([(fstar(z + h * e) - fstar(z - h * e)) / (2 * h) for e in np.eye(m)])
    assert np.max(np.abs(g - gfd)) <= 1e-7 * max(1.0, np.max(np.abs(g)))
    assert abs(g @ z + (d1 + 1)) <= 1e-12
    cones = oracle.Cones(cones_spec)
    assert cones.update_scaling(s, z, mu, 1)
    Hs = np.column_stack([cones.mul_Hs(e) for e in np.eye(m)])
    Hfd = np.column_stack([(grad_at(z + h * e) - grad_at(z - h * e)) / (2 * h) for e in np.eye(m)])
    assert np.allclose(Hs, Hs.T, atol=1e-12)
    assert np.max(np.abs(Hs - mu * Hfd)) <= 1e-6 * np.max(np.abs(Hs))
    # sparse expansion == dense Hs in the KKT solve
    n = 3
    A = rng.standard_normal((m, n))
    Pd = np.diag(rng.uniform(1.0, 2.0, n))
    st = oracle.Settings.default()
    st.static_reg_enable = 0
    ks = oracle.KKTSolver(n, m, _dense_to_csc(np.triu(Pd)), _dense_to_csc(A), cones, settings=st)
    assert ks.p == 3
    assert ks.update()
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    ks.setrhs(rx, rz)
    ok, x, zz = ks.solve()
    Kd = np.block([[Pd, A.T], [A, -Hs]])
    ref = np.linalg.solve(Kd, np.concatenate([rx, rz]))
    assert ok and np.max(np.abs(np.concatenate([x, zz]) - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("n", [2, 5, 17])
def test_psd_numpy_oracle_identities(oracle, n):
    """oracle/psd_numpy.py (PSDTriangleCone): Nesterov-Todd identities Hs z = s and
    lambda = W z = W^-T s (diagonal), W^-1 = inverse of W, Hs = skron(R R') consistent with mul_Hs,
    step length = distance to the boundary, margins / barrier against direct eigenvalues"""
    from oracle import psd_numpy as PN
    rng = np.random.default_rng(n)

    def rand_pd():
        G = rng.standard_normal((n, n))
        return G @ G.T + n * np.eye(n)

    S, Z = rand_pd(), rand_pd()
    s, z = PN.mat_to_svec(S), PN.mat_to_svec(Z)
    assert np.allclose(PN.svec_to_mat(s, n), S) and abs(s @ z - np.trace(S @ Z)) <= 1e-10 * abs(s @ z)
    c = PN.PSDCone(n)
    assert c.update_scaling(s, z)
    assert np.max(np.abs(c.mul_Hs(z) - s)) <= 1e-10 * np.max(np.abs(s))
    lam1, lam2 = c.mul_W(False, z), c.mul_Winv(True, s)
    L = np.diag(c.lam)
    assert np.max(np.abs(PN.svec_to_mat(lam1, n) - L)) <= 1e-9 * c.lam.max()
    assert np.max(np.abs(PN.svec_to_mat(lam2, n) - L)) <= 1e-9 * c.lam.max()
    assert np.all(np.diff(c.lam) <= 0)  # descending, the convention shared with the device
    v = rng.standard_normal(c.numel)
    assert np.max(np.abs(c.mul_Winv(False, c.mul_W(False, v)) - v)) <= 1e-9 * np.max(np.abs(v))
    # Hs block (packed triu) == matrix of mul_Hs
    H = np.column_stack([c.mul_Hs(e) for e in np.eye(c.numel)])
    r, cc = np.tril_indices(c.numel)
    assert np.max(np.abs(c.get_Hs() - H[cc, r])) <= 1e-10 * np.max(np.abs(H))
    # step length: z + alpha dz hits the boundary of the cone exactly at alpha
    dz = PN.mat_to_svec(-rand_pd())
    ds = np.zeros(c.numel)
    alpha = c.step_length(dz, ds, 1e9)
    emin = np.linalg.eigvalsh(PN.svec_to_mat(z + alpha * dz, n)).min()
    assert abs(emin) <= 1e-7 * np.linalg.eigvalsh(Z).max()
    a, b = c.margins(z)
    ev = np.linalg.eigvalsh(Z)
    assert abs(a - ev.min()) <= 1e-10 * ev.max() and abs(b - ev.sum()) <= 1e-10 * ev.sum()
    bar = c.compute_barrier(z, s, np.zeros(c.numel), np.zeros(c.numel), 0.0)
    assert abs(bar + np.log(np.linalg.det(Z)) + np.log(np.linalg.det(S))) <= 1e-8 * abs(bar)


# ---- the reference's dense known-answer tests, pinned on oracle/psd_numpy.py ------------------------------------
def test_psd_svec_conversions_kat():
    """algebra/dense/matrix_math.rs:389-424 (test_svec_conversions): <svec X, svec Y> = <X, Y>, the round trip, and
    the packed order itself -- upper triangle column by column, off-diagonals times sqrt 2 (:165-205)"""
    from oracle import psd_numpy as PN
    X = np.array([[1., 3., -2.], [3., -4., 7.], [-2., 7., 5.]])
    Y = np.array([[2., 5., -4.], [5., 6., 2.], [-4., 2., -3.]])
    x, y = PN.mat_to_svec(X), PN.mat_to_svec(Y)
    assert abs(x @ y - np.sum(X * Y)) < 1e-12
    assert np.max(np.abs(PN.svec_to_mat(x, 3) - X)) < 1e-12
    r2 = np.sqrt(2.0)
    assert np.allclose(x, [1., 3. * r2, -4., -2. * r2, 7. * r2, 5.], rtol=0, atol=1e-15)
    # triangular_index(k) = k (k + 3) / 2 is the position of diagonal k (scalarmath.rs:28-32)
    assert list(PN.PSDCone(3).diag_idx) == [0, 2, 5]


CHOL_KATS = [  # algebra/dense/blas/cholesky.rs:308-378 = svd.rs:367-437: (S, X, B = S X)
    (np.array([[4., 1.], [1., 3.]]), np.array([[2., 3.], [1., 2.]]), np.array([[9., 14.], [5., 9.]])),
    (np.array([[8., -2., 4.], [-2., 12., 2.], [4., 2., 6.]]), np.array([[1., 2.], [3., 4.], [5., 6.]]),
     np.array([[22., 32.], [44., 56.], [40., 52.]])),
    (np.array([[10., 2., 3., 1.], [2., 8., 0., 3.], [3., 0., 6., 2.], [1., 3., 2., 9.]]),
     np.array([[1., 2.], [2., 3.], [3., 1.], [4., 2.]]), np.array([[27., 31.], [30., 34.], [29., 16.], [49., 31.]])),
]


@pytest.mark.parametrize("S,X,B", CHOL_KATS)
def test_psd_dense_engine_kats(S, X, B):
    """the LAPACK calls the numpy restatement stands on, against the reference's vectors for its own engines:
    Cholesky (cholesky.rs:380-419: L L' = S to 1e-8, solve to 1e-12), SVD (svd.rs:439-465 solve to 1e-10; :528-575
    singular values descending, U S V' reconstructs A to 1e-10)"""
    L = np.linalg.cholesky(S)   # what PSDCone.update_scaling / logdet_barrier call
    assert np.max(np.abs(L @ L.T - S)) < 1e-8 and np.allclose(L, np.tril(L))
    assert np.max(np.abs(np.linalg.solve(L.T, np.linalg.solve(L, B)) - X)) <= 1e-12
    U, sg, Vt = np.linalg.svd(S)   # what PSDCone.update_scaling calls
    assert np.all(np.diff(sg) <= 0)
    assert np.max(np.abs((U * sg) @ Vt - S)) < 1e-10
    assert np.max(np.abs(Vt.T @ ((U.T @ B) / sg[:, None]) - X)) < 1e-10


def test_psd_dense_engine_kats_rectangular_and_logdet_and_eig():
    from oracle import psd_numpy as PN
    # svd.rs:492-509 (2x4 and 4x2 factor data): descending, reconstruction
    A24 = np.array([[10., 2., 3., 1.], [2., 8., 0., 3.]])
    for A in (A24, A24.T.copy()):
        U, sg, Vt = np.linalg.svd(A, full_matrices=False)
        assert np.all(np.diff(sg) <= 0) and np.max(np.abs((U * sg) @ Vt - A)) < 1e-10
    # cholesky.rs:421-441: logdet of [[8,-2,4],[-2,12,2],[4,2,6]] = 5.69035945432406, through the cone's barrier
    S = np.array([[8., -2., 4.], [-2., 12., 2.], [4., 2., 6.]])
    c = PN.PSDCone(3)
    assert abs(c.logdet_barrier(PN.mat_to_svec(S), np.zeros(6), 0.0) - 5.69035945432406) < 1e-10
    # syevr.rs:284-318: eigenvalues of the 4x4 matrix are [-1, -1, 8, 9] (1e-6), through the cone's margins
    E = np.array([[3., 2., 4., 0.], [2., 0., 2., 0.], [4., 2., 3., 0.], [0., 0., 0., 9.]])
    assert np.max(np.abs(np.linalg.eigvalsh(E) - np.array([-1., -1., 8., 9.]))) < 1e-6
    amin, apos = PN.PSDCone(4).margins(PN.mat_to_svec(E))
    assert abs(amin + 1.0) < 1e-6 and abs(apos - 17.0) < 1e-6


def test_psd_skron_against_its_definition():
    """psdtrianglecone.rs:467-509 (skron) has no test in the reference: pinned by its definition instead --
    skron(B) svec(X) = svec(B X B') for symmetric X (the operator the KKT block Hs = skron(R R') must be), entry by
    entry, with the sqrt 2 rules of the svec basis"""
    from oracle import psd_numpy as PN
    rng = np.random.default_rng(4)
    for n in (2, 3, 6):
        G = rng.standard_normal((n, n))
        Bm = G @ G.T + np.eye(n)
        numel = n * (n + 1) // 2
        packed = PN.skron_triu(Bm)
        H = np.zeros((numel, numel))
        r, c = np.tril_indices(numel)   # packed triu, column major == row-major lower of the transpose
        H[c, r] = packed
        H[r, c] = packed
        for _ in range(3):
            Xs = rng.standard_normal((n, n))
            Xs = Xs + Xs.T
            assert np.max(np.abs(H @ PN.mat_to_svec(Xs) - PN.mat_to_svec(Bm @ Xs @ Bm.T))) <= 1e-11 * np.max(np.abs(H))


@pytest.mark.parametrize("threads", [1, 3])
def test_mt_comparator_reproduces_the_oracle(oracle, threads):
    """oracle/ldl_mt.c (bench.py's cpu_baseline_mt: the qdldl column algorithm on OpenMP threads) against the scalar
    oracle on the same permuted, regularised matrix: factors entry by entry, the solve, and the residual"""
    from oracle import ldl_mt
    from tests import problems
    pr = problems.portfolio_socp(6, 40, seed=3)
    cones = oracle.Cones(pr["cones"])
    assert cones.update_scaling(pr["s"], pr["z"])
    ko = oracle.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones)
    assert ko.update()
    L = oracle.lib()
    L.orc_kktsolver_ldl.restype = ldl_mt.C.c_void_p
    f = ldl_mt.C.c_void_p(L.orc_kktsolver_ldl(ko._h))
    # the engine's own permutation: recover it from a solve of unit vectors is overkill -- the KKT solver was built
    # without one, so the oracle's AMD stand-in chose it; read it back through the public accessor
    L.orc_qdldl_perm.restype = ldl_mt.P_I64
    perm = np.ctypeslib.as_array(L.orc_qdldl_perm(f), shape=(ko.N,)).copy()
    mt = ldl_mt.LdlMT(oracle, ko, perm, threads)
    Ax = mt.values()
    ok, reg = mt.factor(Ax, ko.settings.dynamic_reg_eps, ko.settings.dynamic_reg_delta)
    assert ok and reg == ko.ldl_regularize_count()
    L.orc_qdldl_Lx.restype = ldl_mt.P_F64
    L.orc_qdldl_D.restype = ldl_mt.P_F64
    Lx_o = np.ctypeslib.as_array(L.orc_qdldl_Lx(f), shape=(mt.nnzL,))
    D_o = np.ctypeslib.as_array(L.orc_qdldl_D(f), shape=(mt.n,))
    assert np.max(np.abs(mt.Lx()[:mt.nnzL] - Lx_o)) <= 1e-12 * max(1.0, np.max(np.abs(Lx_o)))
    # (a different summation order: pivots formed by cancellation -- late-iterate scalings -- agree to fewer digits)
    assert np.max(np.abs(mt.D() - D_o) / np.abs(D_o)) <= 1e-6
    rng = np.random.default_rng(1)
    b = rng.standard_normal(mt.n)
    x = b.copy()
    mt.solve(x)
    e = np.empty(mt.n)
    mt.residual(Ax, x, b, e)   # residual against the matrix that was factored: tiny
    assert np.max(np.abs(e)) <= 1e-6 * max(1.0, np.max(np.abs(x)))  # (raw LDL' solve with +-1e-8 pivots, no refinement)
