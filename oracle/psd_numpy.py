"""Test infrastructure only (see oracle/__init__.py): numpy restatement of PSDTriangleCone
(src/solver/core/cones/psdtrianglecone.rs:8-509) with LAPACK-backed numpy.linalg in place of
the reference's dense engines (algebra/dense/blas/*), plus `MixedCones`, a composite cone
(compositecone.rs:11-352) that delegates Zero / Nonnegative / SecondOrder / Exponential / Power
cones to the C oracle and PSD cones to this module, and `KKTSystemPy`, the DefaultKKTSystem
algebra (default/kktsystem.rs:108-259) written over such a composite.

Conventions the reference leaves to LAPACK are fixed here so that results are comparable across
implementations: singular values descending, and every right singular vector signed so that its
largest-magnitude entry is positive.  `parity unpinned` applies to those two choices only; all
quantities that reach the KKT system (Hs, mul_Hs, ds_from_dz_offset of the combined rhs, step
lengths, margins, barriers) are invariant under them, and the reference's end-to-end answer
tests/basic_sdp.rs:29-57 pins the whole chain (tests/test_e2e_oracle.py).
"""
import numpy as np

SQRT2 = np.sqrt(2.0)


def svec_to_mat(x, n):
    # dense/matrix_math.rs:165-183
    M = np.zeros((n, n))
    r, c = np.tril_indices(n)  # column-major upper == row-major lower of the transpose
    vals = np.where(r == c, x, x / SQRT2)
    M[c, r] = vals
    M[r, c] = vals
    return M


def mat_to_svec(M):
    # dense/matrix_math.rs:186-205
    n = M.shape[0]
    r, c = np.tril_indices(n)
    return np.where(r == c, M[c, r], (M[c, r] + M[r, c]) / SQRT2)


def skron_triu(B):
    """packed triu (column major) of B (x)_s B, psdtrianglecone.rs:467-509"""
    k = B.shape[0]
    numel = k * (k + 1) // 2
    ci, cj = np.tril_indices(k)
    i_idx, j_idx = cj, ci
    scale_a = np.where(i_idx == j_idx, 1.0, 1.0 / SQRT2)
    rp, rq = np.tril_indices(k)
    p_idx, q_idx = rq, rp
    wgt = np.where(p_idx == q_idx, 1.0, SQRT2)
    T1 = B[p_idx][:, i_idx] * B[q_idx][:, j_idx]
    T2 = B[p_idx][:, j_idx] * B[q_idx][:, i_idx]
    H = np.where(i_idx == j_idx, T1, (T1 + T2) * scale_a) * wgt[:, None]
    H = 0.5 * (H + H.T)
    r, c = np.tril_indices(numel)
    return H[r, c]


class PSDCone:
    """PSDTriangleCone(n): numel = n(n+1)/2, degree n"""

    def __init__(self, n):
        self.n = n
        self.numel = n * (n + 1) // 2
        self.R = np.eye(n)
        self.Rinv = np.eye(n)
        self.lam = np.ones(n)
        self.lisqrt = np.ones(n)
        self.diag_idx = np.array([k * (k + 1) // 2 + k for k in range(n)], dtype=np.int64)  # triangular_index

    # psdtrianglecone.rs:144-204
    def update_scaling(self, s, z):
        n = self.n
        if n == 0:  # bail early on a zero-length cone (:151-154)
            return True
        S, Z = svec_to_mat(s, n), svec_to_mat(z, n)
        try:
            L1, L2 = np.linalg.cholesky(S), np.linalg.cholesky(Z)
        except np.linalg.LinAlgError:
            return False
        U, sig, Vt = np.linalg.svd(L2.T @ L1)
        V = Vt.T
        sgn = np.sign(V[np.argmax(np.abs(V), axis=0), np.arange(n)])
        sgn[sgn == 0] = 1.0
        V, U = V * sgn, U * sgn
        self.lam = sig.copy()
        self.lisqrt = 1.0 / np.sqrt(sig)
        self.R = (L1 @ V) * self.lisqrt
        self.Rinv = (self.lisqrt[:, None] * U.T) @ L2.T
        return True

    def get_Hs(self):
        return skron_triu(self.R @ self.R.T)

    # :308-396 mul_Wx_inner
    def _mul(self, Rx, transpose, x):
        X = svec_to_mat(x, self.n)
        Y = Rx @ X @ Rx.T if transpose else Rx.T @ X @ Rx
        return mat_to_svec(Y)

    def mul_W(self, transpose, x):
        return self._mul(self.R, transpose, x)

    def mul_Winv(self, transpose, x):
        return self._mul(self.Rinv, transpose, x)

    def mul_Hs(self, x):  # :214-218
        return self.mul_W(True, self.mul_W(False, x))

    def affine_ds(self):  # :220-225
        ds = np.zeros(self.numel)
        ds[self.diag_idx] = self.lam * self.lam
        return ds

    def circ_op(self, y, z):  # :406-420
        Y, Z = svec_to_mat(y, self.n), svec_to_mat(z, self.n)
        return mat_to_svec(0.5 * (Y @ Z + Z @ Y))

    def lambda_inv_circ_op(self, z):  # :284-299
        Z = svec_to_mat(z, self.n)
        return mat_to_svec(2.0 * Z / (self.lam[:, None] + self.lam[None, :]))

    def combined_ds_shift(self, step_z, step_s, sigma_mu):  # symmetric_common.rs:53-84
        wz = self.mul_W(False, step_z)
        ws = self.mul_Winv(True, step_s)
        shift = self.circ_op(ws, wz)
        shift[self.diag_idx] += -sigma_mu
        return shift, wz, ws

    def ds_from_dz_offset(self, ds):  # symmetric_common.rs:89-95
        return self.mul_W(True, self.lambda_inv_circ_op(ds))

    def _step_component(self, d, amax):  # :437-463
        if self.n == 0:
            return amax
        D = svec_to_mat(d, self.n) * self.lisqrt[:, None] * self.lisqrt[None, :]
        gamma = np.linalg.eigvalsh(D).min()
        return min(-1.0 / gamma, amax) if gamma < 0 else amax

    def step_length(self, dz, ds, amax):  # :235-279
        az = self._step_component(self.mul_W(False, dz), amax)
        as_ = self._step_component(self.mul_Winv(True, ds), amax)
        return min(az, as_)

    def margins(self, z):  # :104-121
        if self.n == 0:
            return 1.7976931348623157e308, 0.0
        e = np.linalg.eigvalsh(svec_to_mat(z, self.n))
        return float(e.min()), float(np.maximum(e, 0.0).sum())

    def scaled_unit_shift(self, z, alpha):  # :123-129, in place
        z[self.diag_idx] += alpha

    def logdet_barrier(self, x, dx, alpha):  # :289-303
        if self.n == 0:
            return 0.0
        Q = svec_to_mat(x + alpha * dx, self.n)
        try:
            L = np.linalg.cholesky(Q)
        except np.linalg.LinAlgError:
            return np.inf
        return 2.0 * float(np.sum(np.log(np.diag(L))))

    def compute_barrier(self, z, s, dz, ds, alpha):  # :281-286
        return -self.logdet_barrier(z, dz, alpha) - self.logdet_barrier(s, ds, alpha)


PSD_TAG = 6


class MixedCones:
    """CompositeCone over the C oracle's cones + numpy PSD cones.  `specs` as for oracle.Cones."""

    def __init__(self, oracle_mod, specs):
        self.specs = [tuple(c) for c in specs]
        self.c = oracle_mod.Cones(specs)  # PSD entries are inert there (layout and degree only)
        self.psd = []  # (offset, block offset, PSDCone)
        pos = blk = 0
        for sp_ in self.specs:
            tag, dim = sp_[0], sp_[1]
            if tag == PSD_TAG:
                cone = PSDCone(dim)
                self.psd.append((pos, blk, cone))
                pos += cone.numel
                blk += cone.numel * (cone.numel + 1) // 2
            else:
                ne = 3 if tag in (3, 4) else dim
                pos += ne
                if tag == 2 and dim <= 4:
                    blk += ne * (ne + 1) // 2
                elif tag in (3, 4):
                    blk += 6
                else:
                    blk += ne
        self.m = pos
        self.nblockvals = blk
        assert blk == self.c.nblockvals
        self.degree = self.c.degree
        self.is_symmetric = self.c.is_symmetric
        self._h = self.c._h

    def _sl(self, off, cone):
        return slice(off, off + cone.numel)

    def update_scaling(self, s, z, mu=1.0, strategy=0):
        ok = self.c.update_scaling(s, z, mu, strategy)
        for off, _, cone in self.psd:
            ok = cone.update_scaling(s[self._sl(off, cone)], z[self._sl(off, cone)]) and ok
        return ok

    def get_Hs(self):
        Hs = self.c.get_Hs()
        for _, blk, cone in self.psd:
            h = cone.get_Hs()
            Hs[blk:blk + len(h)] = h
        return Hs

    def mul_Hs(self, x):
        y = self.c.mul_Hs(x)
        for off, _, cone in self.psd:
            y[self._sl(off, cone)] = cone.mul_Hs(x[self._sl(off, cone)])
        return y

    def affine_ds(self, s):
        ds = self.c.affine_ds(s)
        for off, _, cone in self.psd:
            ds[self._sl(off, cone)] = cone.affine_ds()
        return ds

    def combined_ds_shift(self, step_z, step_s, sigma_mu):
        shift, wz, ws = self.c.combined_ds_shift(step_z, step_s, sigma_mu)
        for off, _, cone in self.psd:
            sl = self._sl(off, cone)
            shift[sl], wz[sl], ws[sl] = cone.combined_ds_shift(step_z[sl], step_s[sl], sigma_mu)
        return shift, wz, ws

    def ds_from_dz_offset(self, ds, z):
        out = self.c.ds_from_dz_offset(ds, z)
        for off, _, cone in self.psd:
            out[self._sl(off, cone)] = cone.ds_from_dz_offset(ds[self._sl(off, cone)])
        return out

    def step_length(self, dz, ds, z, s, alpha_max=1.0):
        # compositecone.rs:300-340: symmetric cones (PSD among them) first.  The C oracle handles
        # its symmetric cones, then its nonsymmetric ones; PSD cones are inserted in between by
        # running the C call on the PSD-reduced alpha (monotone: min commutes for symmetric cones)
        a = alpha_max
        for off, _, cone in self.psd:
            sl = self._sl(off, cone)
            a = min(a, cone.step_length(dz[sl], ds[sl], a))
        return self.c.step_length(dz, ds, z, s, a)

    def margins(self, z):
        a, b = self.c.margins(z)
        for off, _, cone in self.psd:
            ca, cb = cone.margins(z[self._sl(off, cone)])
            a, b = min(a, ca), b + cb
        return a, b

    def scaled_unit_shift(self, z, alpha, primal_cone):
        self.c.scaled_unit_shift(z, alpha, primal_cone)
        for off, _, cone in self.psd:
            z[off + cone.diag_idx] += alpha

    def unit_initialization(self, z, s):
        self.c.unit_initialization(z, s)
        for off, _, cone in self.psd:
            z[self._sl(off, cone)] = 0.0
            s[self._sl(off, cone)] = 0.0
            z[off + cone.diag_idx] = 1.0
            s[off + cone.diag_idx] = 1.0

    def compute_barrier(self, z, s, dz, ds, alpha):
        b = self.c.compute_barrier(z, s, dz, ds, alpha)
        for off, _, cone in self.psd:
            sl = self._sl(off, cone)
            b += cone.compute_barrier(z[sl], s[sl], dz[sl], ds[sl], alpha)
        return b

    def identity_vector(self):
        """e of the composite cone (input to update_scaling that reproduces set_identity_scaling)"""
        e = np.zeros(self.m)
        self.unit_initialization(e, np.zeros(self.m))
        return e


class KKTSystemPy:
    """DefaultKKTSystem (default/kktsystem.rs:108-259) + DefaultResiduals::update over
    oracle.KKTSolver and MixedCones (Hs of PSD cones handed to the solver as an override)."""

    def __init__(self, oracle_mod, kktsolver, cones, n, m, P, A, q, b):
        import scipy.sparse as sp
        self.ks, self.cones, self.n, self.m = kktsolver, cones, n, m
        Pu = sp.csc_matrix((np.asarray(P[2], float), np.asarray(P[1]), np.asarray(P[0])), shape=(n, n))
        self.P = Pu + sp.triu(Pu, 1).T
        self.A = sp.csc_matrix((np.asarray(A[2], float), np.asarray(A[1]), np.asarray(A[0])), shape=(m, n))
        self.q, self.b = np.asarray(q, float), np.asarray(b, float)
        self.nnzP = len(P[2])
        self.x2, self.z2 = np.zeros(n), np.zeros(m)

    def update(self):
        if not self.ks.update(self.cones.get_Hs()):
            return False
        self.ks.setrhs(-self.q, self.b)
        ok, self.x2, self.z2 = self.ks.solve()
        return ok

    def solve(self, lhs, rhs, variables, direction):
        cterm = variables.s.copy() if direction == 0 else self.cones.ds_from_dz_offset(rhs.s, variables.z)
        self.ks.setrhs(rhs.x, cterm - rhs.z)
        ok, x1, z1 = self.ks.solve()
        if not ok:
            return False
        P, q, b, x2, z2 = self.P, self.q, self.b, self.x2, self.z2
        xi = variables.x / variables.tau
        tau_num = rhs.tau - rhs.kappa / variables.tau + q @ x1 + b @ z1 + 2.0 * (xi @ (P @ x1))
        d = xi - x2
        tau_den = variables.kappa / variables.tau - q @ x2 - b @ z2 + (d @ (P @ d)) - (x2 @ (P @ x2))
        lhs.tau = tau_num / tau_den
        lhs.x[:] = x1 + lhs.tau * x2
        lhs.z[:] = z1 + lhs.tau * z2
        lhs.s[:] = -(self.cones.mul_Hs(lhs.z) + cterm)
        lhs.kappa = -(rhs.kappa + variables.kappa * lhs.tau) / variables.tau
        return True

    def solve_initial_point(self, variables):
        if self.nnzP == 0:
            self.ks.setrhs(np.zeros(self.n), self.b)
            ok, x, s = self.ks.solve()
            variables.x[:], variables.s[:] = x, -s
            if not ok:
                return ok
            self.ks.setrhs(-self.q, np.zeros(self.m))
            ok, _, z = self.ks.solve()
            variables.z[:] = z
            return ok
        self.ks.setrhs(-self.q, self.b)
        ok, x, z = self.ks.solve()
        variables.x[:], variables.z[:], variables.s[:] = x, z, -z
        return ok

    def residuals(self, v):
        Px = self.P @ v.x
        rx_inf = -(self.A.T @ v.z)
        rz_inf = self.A @ v.x + v.s
        qx, bz, sz, xPx = self.q @ v.x, self.b @ v.z, v.s @ v.z, v.x @ Px
        return dict(rx=rx_inf - Px - v.tau * self.q, rz=rz_inf - v.tau * self.b, rx_inf=rx_inf, rz_inf=rz_inf, Px=Px,
                    rtau=qx + bz + v.kappa + xPx / v.tau, dot_qx=qx, dot_bz=bz, dot_sz=sz, dot_xPx=xPx)
