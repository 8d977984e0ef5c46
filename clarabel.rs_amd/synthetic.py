"""Seeded synthetic generators for the five BASELINE.json configs.
Pure numpy/scipy; used by tests/, tools/ and bench.py (host-side input generation only).  Every generator returns a dict
  n, m, P=(colptr,rowval,nzval) [triu], A=(colptr,rowval,nzval), cones=[(tag,dim[,dim2])],
  s, z  (a strictly interior primal/dual pair, consistent with the cones)
"""
import numpy as np
import scipy.sparse as sp

ZERO, NN, SOC, EXP, POW, GENPOW, PSD = range(7)


def _csc(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    return (M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(np.float64))


def _interior(rng, cones, late=False):
    """random strictly interior (s, z); `late` mimics a late IPM iterate:
    NN entries with s/z spanning 1e-6..1e6."""
    s_parts, z_parts = [], []
    for c in cones:
        tag, dim = c[0], c[1]
        if tag == ZERO:
            s_parts.append(np.zeros(dim))
            z_parts.append(np.zeros(dim))
        elif tag == NN:
            if late:
                r = 10.0 ** rng.uniform(-3, 3, dim)
                mu = 10.0 ** rng.uniform(-4, -2, dim)
                s_parts.append(np.sqrt(mu) * r)
                z_parts.append(np.sqrt(mu) / r)
            else:
                s_parts.append(rng.uniform(0.3, 3.0, dim))
                z_parts.append(rng.uniform(0.3, 3.0, dim))
        elif tag == SOC:
            for parts in (s_parts, z_parts):
                v = rng.standard_normal(dim)
                v[0] = np.linalg.norm(v[1:]) * (1.0 + (rng.uniform(1e-3, 1e-1) if late else rng.uniform(0.1, 1.0))) + 1e-3
                parts.append(v)
        else:
            numel = 3 if tag in (EXP, POW) else dim * (dim + 1) // 2
            s_parts.append(np.zeros(numel))
            z_parts.append(np.zeros(numel))
    return np.concatenate(s_parts) if s_parts else np.zeros(0), np.concatenate(z_parts) if z_parts else np.zeros(0)


def basic_qp():
    """C1: tests/basic_qp.rs:16-42 verbatim (n=2, m=6, NN(3)+NN(3))."""
    P = np.array([[4., 1.], [0., 2.]])  # triu of [4 1;1 2]
    A0 = np.array([[1., 1.], [1., 0.], [0., 1.]])
    A = np.vstack([-A0, A0])
    cones = [(NN, 3), (NN, 3)]
    return dict(n=2, m=6, P=_csc(P), A=_csc(A), cones=cones, q=np.array([1., 1.]),
                b=np.array([-1., 0., 0., 1., 0.7, 0.7]), s=np.ones(6), z=np.ones(6))


def random_qp(n=100000, m=200000, band=50, seed=1, late=False):
    """C2: band-limited random sparse QP, one NonnegativeCone(m)."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(m), 5)
    center = (np.arange(m) * n) // m
    cols = (np.repeat(center, 5) + rng.integers(-band, band + 1, 5 * m)) % n
    vals = rng.standard_normal(5 * m)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(m, n)).tocsc()
    A.sum_duplicates()
    rng2 = np.random.default_rng(seed + 1)
    d = rng2.uniform(1.0, 2.0, n)
    oc = np.repeat(np.arange(n), 2)
    orow = oc - rng2.integers(1, band + 1, 2 * n)
    keep = orow >= 0
    ov = 0.1 * rng2.standard_normal(2 * n)
    P = sp.coo_matrix((np.concatenate([d, ov[keep]]), (np.concatenate([np.arange(n), orow[keep]]),
                                                       np.concatenate([np.arange(n), oc[keep]]))), shape=(n, n)).tocsc()
    P.sum_duplicates()
    cones = [(NN, m)]
    s, z = _interior(rng, cones, late)
    return dict(n=n, m=m, P=_csc(P), A=_csc(A), cones=cones, s=s, z=z)


def portfolio_socp(nblocks=1000, blocksize=1000, seed=3, late=False):
    """C3: n = nblocks*blocksize; cones [Zero(1) budget, NN(n), nblocks x SOC(blocksize+1)];
    P = 0; KKT is block-arrow: per SOC two dense columns (u, v), one dense budget row."""
    rng = np.random.default_rng(seed)
    n = nblocks * blocksize
    dim = blocksize + 1
    m = 1 + n + nblocks * dim
    # rows: 0 budget (1'x), 1..n: -I, then per block: row0 empty, rows 1..blocksize = -diag(d)
    d = rng.uniform(0.5, 1.5, n)
    j = np.arange(n)
    soc_rows = 1 + n + (j // blocksize) * dim + 1 + (j % blocksize)
    rows = np.concatenate([np.zeros(n, dtype=np.int64), 1 + j, soc_rows])
    cols = np.concatenate([j, j, j])
    vals = np.concatenate([np.ones(n), -np.ones(n), -d])
    A = sp.coo_matrix((vals, (rows, cols)), shape=(m, n)).tocsc()
    A.sort_indices()
    P = sp.csc_matrix((n, n))
    cones = [(ZERO, 1), (NN, n)] + [(SOC, dim)] * nblocks
    s, z = _interior(rng, cones, late)
    return dict(n=n, m=m, P=_csc(P), A=_csc(A), cones=cones, s=s, z=z)


def portfolio_problem(nblocks=1000, blocksize=1000, seed=3):
    """a complete, strictly feasible and bounded problem on the config-3 pattern (portfolio_socp):
    maximise mu'x  s.t.  1'x = nblocks,  x >= 0,  ||d_k o x_k|| <= gamma_k per block (risk budgets)."""
    pr = portfolio_socp(nblocks, blocksize, seed)
    rng = np.random.default_rng(seed + 1000)
    n, dim = pr["n"], blocksize + 1
    q = -rng.uniform(0.0, 1.0, n)
    b = np.zeros(pr["m"])
    b[0] = float(nblocks)
    gamma = rng.uniform(1.5, 3.0, nblocks) / np.sqrt(blocksize)
    b[1 + n + dim * np.arange(nblocks)] = gamma
    pr.update(q=q, b=b)
    return pr


def batched_socp(nbatch=1024, n_b=2000, blocks_per=2, seed=100, late=False):
    """C4: `nbatch` independent copies of a small C3-pattern SOCP, concatenated block
    diagonally (csc/block_concatenate.rs:22) -> elimination forest with nbatch roots."""
    parts = [portfolio_socp(blocks_per, n_b // blocks_per, seed + i, late) for i in range(nbatch)]
    return blockdiag(parts)


def blockdiag(parts):
    n = sum(p["n"] for p in parts)
    m = sum(p["m"] for p in parts)
    A = sp.block_diag([sp.csc_matrix((p["A"][2], p["A"][1], p["A"][0]), shape=(p["m"], p["n"])) for p in parts],
                      format="csc")
    P = sp.block_diag([sp.csc_matrix((p["P"][2], p["P"][1], p["P"][0]), shape=(p["n"], p["n"])) for p in parts],
                      format="csc")
    cones = [c for p in parts for c in p["cones"]]
    return dict(n=n, m=m, P=_csc(P), A=_csc(A), cones=cones, s=np.concatenate([p["s"] for p in parts]),
                z=np.concatenate([p["z"] for p in parts]), part_n=[p["n"] for p in parts],
                part_m=[p["m"] for p in parts])


def _svec(M):
    """svec of a symmetric matrix: packed triu, column major, off-diagonals * sqrt(2)
    (src/algebra/dense/matrix_math.rs:165-205)."""
    k = M.shape[0]
    r, c = np.tril_indices(k)  # row-major lower == column-major upper of a symmetric matrix
    return M[r, c] * np.where(r == c, 1.0, np.sqrt(2.0))


def psd_scaling_Hs(S, Z):
    """Nesterov-Todd Hs = (R R') (x)_s (R R') for a PSDTriangleCone (psdtrianglecone.rs:144-204,
    467-509), built with dense numpy; returns the packed-triu (column-major) Hs block that
    get_Hs would hand to the KKT update (dense/types.rs:187-201)."""
    L1 = np.linalg.cholesky(S)
    L2 = np.linalg.cholesky(Z)
    U, sig, Vt = np.linalg.svd(L2.T @ L1)
    R = L1 @ Vt.T @ np.diag(sig ** -0.5)
    B = R @ R.T
    k = S.shape[0]
    numel = k * (k + 1) // 2
    # column a of H = svec(B E_a B'), E_a the a-th svec basis matrix; vectorised over a
    ci, cj = np.tril_indices(k)          # basis index a <-> (i, j) = (cj, ci) with i <= j
    i_idx, j_idx = cj, ci
    scale_a = np.where(i_idx == j_idx, 1.0, 1.0 / np.sqrt(2.0))
    # (B E B')[p, q] = s * (B[p,i] B[q,j] + B[p,j] B[q,i]) (off-diagonal), B[p,i] B[q,i] (diagonal)
    rp, rq = np.tril_indices(k)
    p_idx, q_idx = rq, rp                # output svec entry (p, q), p <= q
    wgt = np.where(p_idx == q_idx, 1.0, np.sqrt(2.0))
    T1 = B[p_idx][:, i_idx] * B[q_idx][:, j_idx]
    T2 = B[p_idx][:, j_idx] * B[q_idx][:, i_idx]
    H = np.where(i_idx == j_idx, T1, (T1 + T2) * scale_a) * wgt[:, None]
    H = 0.5 * (H + H.T)
    r, c = np.tril_indices(numel)
    return H[r, c]


def chordal_sdp(ncliques=6, dim=6, overlap=2, nsoc=3, socdim=7, seed=5, with_hs=True):
    """C5 (small scale): the post-decomposition shape of chordal/decomp/augment_compact.rs:31-75:
    a chain of PSDTriangleCone(dim) cliques whose overlapping svec entries are tied by +1/-1
    columns, plus sparse-form SOCs.  Hs blocks of the PSD cones are supplied by the host
    (psd_scaling_Hs) -- exactly what chip_kkt_update(hsblocks) consumes."""
    rng = np.random.default_rng(seed)
    numel = dim * (dim + 1) // 2
    ov = overlap * (overlap + 1) // 2
    n_orig = 4 * ncliques
    n = n_orig + (ncliques - 1) * ov
    m = ncliques * numel + nsoc * socdim
    rows, cols, vals = [], [], []
    for r in range(ncliques * numel):
        for c in rng.choice(n_orig, size=3, replace=False):
            rows.append(r)
            cols.append(int(c))
            vals.append(rng.standard_normal())
    # overlap coupling: leading `ov` svec entries of clique k+1 tied to trailing of clique k
    for k in range(ncliques - 1):
        for t in range(ov):
            col = n_orig + k * ov + t
            rows += [k * numel + numel - ov + t, (k + 1) * numel + t]
            cols += [col, col]
            vals += [1.0, -1.0]
    base = ncliques * numel
    for r in range(nsoc * socdim):
        for c in rng.choice(n_orig, size=2, replace=False):
            rows.append(base + r)
            cols.append(int(c))
            vals.append(rng.standard_normal())
    A = sp.coo_matrix((vals, (rows, cols)), shape=(m, n)).tocsc()
    A.sum_duplicates()
    P = sp.csc_matrix((n, n))
    cones = [(PSD, dim)] * ncliques + [(SOC, socdim)] * nsoc
    s, z = _interior(rng, cones)
    hs = []
    nscal = min(ncliques, 16)  # distinct scalings, cycled (the numpy oracle costs ~0.6 s per 50x50 cone)
    cache = []
    for k in range(ncliques):
        if k < nscal:
            G1 = rng.standard_normal((dim, dim))
            G2 = rng.standard_normal((dim, dim))
            S, Z = G1 @ G1.T + dim * np.eye(dim), G2 @ G2.T + dim * np.eye(dim)
            cache.append((_svec(S), _svec(Z), psd_scaling_Hs(S, Z) if with_hs else None))
        sv, zv, h = cache[k % nscal]
        s[k * numel:(k + 1) * numel] = sv
        z[k * numel:(k + 1) * numel] = zv
        hs.append(h)
    # (with_hs=False: large cones whose numel x numel Hs block the caller does not need on the host)
    hs_full = np.concatenate(hs + [np.zeros(socdim)] * nsoc) if with_hs else None
    return dict(n=n, m=m, P=_csc(P), A=_csc(A), cones=cones, s=s, z=z, hsblocks=hs_full)


def exp_interior(rng, scale=0.05):
    """a strictly interior primal/dual pair of the exponential cone near the central point of
    expcone.rs:94-100 (unit_initialization)"""
    c = np.array([-1.051383945322714, 0.556409619469370, 1.258967884768947])
    return c * (1.0 + scale * rng.standard_normal(3)), c * (1.0 + scale * rng.standard_normal(3))


def pow_interior(rng, alpha):
    """interior of K_pow(alpha) = {s0^a s1^(1-a) >= |s2|} and of its dual
    {(z0/a)^a (z1/(1-a))^(1-a) >= |z2|}"""
    s = np.array([rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), 0.0])
    s[2] = rng.uniform(-0.8, 0.8) * s[0] ** alpha * s[1] ** (1 - alpha)
    z = np.array([rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), 0.0])
    z[2] = rng.uniform(-0.8, 0.8) * (z[0] / alpha) ** alpha * (z[1] / (1 - alpha)) ** (1 - alpha)
    return s, z


def mixed_conic(nexp=40, npow=30, nsoc=5, socdim=9, nn=50, seed=11):
    """all device-held cone kinds in one problem (cf. tests/mixed_conic.rs): Zero, Nonnegative,
    SecondOrder (sparse and dense), Exponential, Power; A random sparse, P diagonal-dominant"""
    rng = np.random.default_rng(seed)
    cones, s_parts, z_parts = [(ZERO, 3), (NN, nn)], [np.zeros(3), rng.uniform(0.3, 3, nn)], \
        [np.zeros(3), rng.uniform(0.3, 3, nn)]
    for i in range(nsoc):
        d = socdim if i % 2 == 0 else 3
        cones.append((SOC, d))
        for parts in (s_parts, z_parts):
            v = rng.standard_normal(d)
            v[0] = np.linalg.norm(v[1:]) * 1.5 + 0.1
            parts.append(v)
    for _ in range(nexp):
        cones.append((EXP, 3))
        s, z = exp_interior(rng)
        s_parts.append(s)
        z_parts.append(z)
    for _ in range(npow):
        a = float(rng.uniform(0.15, 0.85))
        cones.append((POW, 3, 0, a))
        s, z = pow_interior(rng, a)
        s_parts.append(s)
        z_parts.append(z)
    s, z = np.concatenate(s_parts), np.concatenate(z_parts)
    m = len(s)
    n = max(8, m // 3)
    A = sp.random(m, n, density=min(1.0, 4.0 / n), random_state=np.random.RandomState(seed), format="csc")
    A = A + sp.csc_matrix((np.ones(min(m, n)), (np.arange(min(m, n)), np.arange(min(m, n)))), shape=(m, n))
    P = sp.diags(rng.uniform(0.5, 1.5, n)).tocsc()
    return dict(n=n, m=m, P=_csc(P), A=_csc(A), cones=cones, s=s, z=z)
