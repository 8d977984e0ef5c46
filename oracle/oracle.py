"""ctypes binding of the CPU oracle (oracle/*.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (clarabel.rs_amd) never does.

The oracle restates the reference's qdldl engine and DirectLDLKKTSolver
(see the headers of qdldl_oracle.c / kkt_oracle.c for file:line citations).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64 = np.int64
f64 = np.float64
P_I64 = C.POINTER(C.c_int64)
P_F64 = C.POINTER(C.c_double)
P_I8 = C.POINTER(C.c_int8)
P_I32 = C.POINTER(C.c_int32)

ERR_NAMES = {0: "ok", 1: "IncompatibleDimension", 2: "EmptyColumn", 3: "NotUpperTriangular",
             4: "ZeroPivot", 5: "InvalidPermutation", 6: "SymbolicOnly"}

CONE_ZERO, CONE_NONNEG, CONE_SOC, CONE_EXP, CONE_POW, CONE_GENPOW, CONE_PSDTRI = range(7)


def build(force=False):
    """liboracle.so (portable); with ORACLE_NATIVE=1 in the environment (set by bench.py's cpu_baseline
    leg before the first use) a -march=native build made on THIS machine is preferred for timing."""
    srcs = [os.path.join(_HERE, f) for f in ("qdldl_oracle.c", "kkt_oracle.c")]
    if os.environ.get("ORACLE_NATIVE") == "1":
        so = os.path.join(_HERE, "_native", "liboracle_native.so")
        try:
            if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
                subprocess.check_call(["make", "-C", _HERE, "-B", "native"], stdout=subprocess.DEVNULL)
            return so
        except Exception:
            pass  # no compiler on this box: fall back to the prebuilt portable library
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def is_native():
    return _LIB is not None and "native" in getattr(_LIB, "_name", "")


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_qdldl_new.restype = C.c_int
        for name in ("n", "nnzL", "nnzA", "positive_inertia", "regularize_count"):
            getattr(L, "orc_qdldl_" + name).restype = C.c_int64
        for name in ("Lp", "Li", "etree", "Lnz", "Ap", "Ai", "AtoPAPt"):
            getattr(L, "orc_qdldl_" + name).restype = P_I64
        for name in ("Lx", "D", "Dinv", "Ax"):
            getattr(L, "orc_qdldl_" + name).restype = P_F64
        L.orc_norm_inf.restype = C.c_double
        L.orc_norm2.restype = C.c_double
        L.orc_quad_form_triu.restype = C.c_double
        L.orc_cones_new.restype = C.c_void_p
        L.orc_cones_new_ex.restype = C.c_void_p
        L.orc_cones_step_length.restype = C.c_double
        L.orc_wright_omega.restype = C.c_double
        for name in ("Hs3", "Hdual", "grad3"):
            getattr(L, "orc_cone_" + name).restype = P_F64
        for name in ("numel", "nblockvals", "pdim"):
            getattr(L, "orc_cones_" + name).restype = C.c_int64
        L.orc_cone_eta.restype = C.c_double
        L.orc_cone_d.restype = C.c_double
        for name in ("w", "lambda", "u", "v"):
            getattr(L, "orc_cone_" + name).restype = P_F64
        L.orc_assemble_kkt.restype = C.c_void_p
        L.orc_kktmat_dim.restype = C.c_int64
        L.orc_kktmat_nnz.restype = C.c_int64
        L.orc_kktmat_nsparse.restype = C.c_int64
        for name in ("colptr", "rowval", "map_P", "map_A", "map_Hs", "map_diagP", "map_diag_full",
                     "map_u", "map_v", "map_q", "map_D"):
            getattr(L, "orc_kktmat_" + name).restype = P_I64
        L.orc_kktmat_nzval.restype = P_F64
        L.orc_kktsolver_new.restype = C.c_void_p
        L.orc_kktsolver_dim.restype = C.c_int64
        L.orc_kktsolver_pdim.restype = C.c_int64
        L.orc_kktsolver_x.restype = P_F64
        L.orc_kktsolver_b.restype = P_F64
        L.orc_kktsolver_kktmat.restype = C.c_void_p
        L.orc_kktsolver_ldl.restype = C.c_void_p
        L.orc_kktsolver_dsigns.restype = P_I8
        L.orc_kktsolver_last_ir_iters.restype = C.c_int32
        L.orc_kktsolver_regularizer.restype = C.c_double
        L.orc_kktsystem_new.restype = C.c_void_p
        L.orc_dot.restype = C.c_double
        L.orc_cones_degree.restype = C.c_int64
        L.orc_cones_compute_barrier.restype = C.c_double
    return _LIB


def _pi(a):
    return a.ctypes.data_as(P_I64)


def _pf(a):
    return a.ctypes.data_as(P_F64)


def _ai(a):
    return np.ascontiguousarray(a, dtype=i64)


def _af(a):
    return np.ascontiguousarray(a, dtype=f64)


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class Settings(C.Structure):
    """orc_settings / CoreSettings subset (settings.rs:139-181)."""
    _fields_ = [("static_reg_enable", C.c_int), ("static_reg_constant", C.c_double),
                ("static_reg_proportional", C.c_double), ("dynamic_reg_enable", C.c_int),
                ("dynamic_reg_eps", C.c_double), ("dynamic_reg_delta", C.c_double),
                ("ir_enable", C.c_int), ("ir_reltol", C.c_double), ("ir_abstol", C.c_double),
                ("ir_max_iter", C.c_int32), ("ir_stop_ratio", C.c_double)]

    @staticmethod
    def default():
        s = Settings()
        lib().orc_settings_default(C.byref(s))
        return s


# ----------------------------------------------------------------------------
# free functions
# ----------------------------------------------------------------------------
def invperm(p):
    p = _ai(p)
    ip = np.zeros_like(p)
    rc = lib().orc_invperm(C.c_int64(len(p)), _pi(p), _pi(ip))
    if rc:
        raise ValueError(ERR_NAMES[rc])
    return ip


def permute(b, p):
    b, p = _af(b), _ai(p)
    x = np.zeros_like(b)
    lib().orc_permute(C.c_int64(len(p)), _pf(x), _pf(b), _pi(p))
    return x


def ipermute(b, p):
    b, p = _af(b), _ai(p)
    x = np.zeros_like(b)
    lib().orc_ipermute(C.c_int64(len(p)), _pf(x), _pf(b), _pi(p))
    return x


def permute_symmetric(n, Ap, Ai, Ax, iperm):
    Ap, Ai, Ax, iperm = _ai(Ap), _ai(Ai), _af(Ax), _ai(iperm)
    nnz = int(Ap[n])
    Pc = np.zeros(n + 1, dtype=i64)
    Pr = np.zeros(nnz, dtype=i64)
    Pv = np.zeros(nnz, dtype=f64)
    mp = np.zeros(nnz, dtype=i64)
    lib().orc_permute_symmetric(C.c_int64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pi(iperm), _pi(Pc),
                                _pi(Pr), _pf(Pv), _pi(mp))
    return Pc, Pr, Pv, mp


def etree(n, Ap, Ai):
    Ap, Ai = _ai(Ap), _ai(Ai)
    work = np.zeros(max(n, 1), dtype=i64)
    Lnz = np.zeros(max(n, 1), dtype=i64)
    et = np.zeros(max(n, 1), dtype=i64)
    lib().orc_etree(C.c_int64(n), _pi(Ap), _pi(Ai), _pi(work), _pi(Lnz), _pi(et))
    return et[:n], Lnz[:n]


def lsolve(Lp, Li, Lx, x):
    Lp, Li, Lx, x = _ai(Lp), _ai(Li), _af(Lx), _af(x).copy()
    lib().orc_lsolve(C.c_int64(len(x)), _pi(Lp), _pi(Li), _pf(Lx), _pf(x))
    return x


def ltsolve(Lp, Li, Lx, x):
    Lp, Li, Lx, x = _ai(Lp), _ai(Li), _af(Lx), _af(x).copy()
    lib().orc_ltsolve(C.c_int64(len(x)), _pi(Lp), _pi(Li), _pf(Lx), _pf(x))
    return x


def solve_factors(Lp, Li, Lx, Dinv, b):
    Lp, Li, Lx, Dinv, b = _ai(Lp), _ai(Li), _af(Lx), _af(Dinv), _af(b).copy()
    lib().orc_solve_factors(C.c_int64(len(b)), _pi(Lp), _pi(Li), _pf(Lx), _pf(Dinv), _pf(b))
    return b


def symv(n, Ap, Ai, Ax, y, x, a, b):
    Ap, Ai, Ax, y, x = _ai(Ap), _ai(Ai), _af(Ax), _af(y).copy(), _af(x)
    lib().orc_symv(C.c_int64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pf(y), _pf(x), C.c_double(a), C.c_double(b))
    return y


def quad_form_triu(n, Ap, Ai, Ax, y, x):
    Ap, Ai, Ax, y, x = _ai(Ap), _ai(Ai), _af(Ax), _af(y), _af(x)
    return lib().orc_quad_form_triu(C.c_int64(n), _pi(Ap), _pi(Ai), _pf(Ax), _pf(y), _pf(x))


def norm_inf(v):
    v = _af(v)
    return lib().orc_norm_inf(_pf(v), C.c_int64(len(v)))


def norm2(v):
    v = _af(v)
    return lib().orc_norm2(_pf(v), C.c_int64(len(v)))


# ----------------------------------------------------------------------------
# QDLDLFactorisation
# ----------------------------------------------------------------------------
class QDLDL:
    """QDLDLFactorisation (qdldl.rs:72-211) with an explicit permutation."""

    def __init__(self, n, Ap, Ai, Ax, perm=None, Dsigns=None, logical=False, regularize_enable=True,
                 regularize_eps=1e-12, regularize_delta=1e-7, m=None):
        Ap, Ai, Ax = _ai(Ap), _ai(Ai), _af(Ax)
        perm = _ai(np.arange(n) if perm is None else perm)
        self._h = C.c_void_p()
        ds = None
        if Dsigns is not None:
            ds = np.ascontiguousarray(Dsigns, dtype=np.int8)
        rc = lib().orc_qdldl_new(C.byref(self._h), C.c_int64(n if m is None else m), C.c_int64(n), _pi(Ap),
                                 _pi(Ai), _pf(Ax), _pi(perm),
                                 ds.ctypes.data_as(P_I8) if ds is not None else None,
                                 C.c_int(int(logical)), C.c_int(int(regularize_enable)),
                                 C.c_double(regularize_eps), C.c_double(regularize_delta))
        if rc:
            self._h = None
            raise ValueError(ERR_NAMES[rc])
        self.n = n
        self.perm = perm

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_qdldl_free(self._h)
            self._h = None

    def solve(self, b):
        b = _af(b).copy()
        rc = lib().orc_qdldl_solve(self._h, _pf(b))
        if rc:
            raise RuntimeError(ERR_NAMES[rc])
        return b

    def update_values(self, idx, vals):
        idx, vals = _ai(idx), _af(vals)
        lib().orc_qdldl_update_values(self._h, _pi(idx), _pf(vals), C.c_int64(len(idx)))

    def scale_values(self, idx, s):
        idx = _ai(idx)
        lib().orc_qdldl_scale_values(self._h, _pi(idx), C.c_double(s), C.c_int64(len(idx)))

    def offset_values(self, idx, off, signs):
        idx = _ai(idx)
        signs = np.ascontiguousarray(signs, dtype=np.int8)
        lib().orc_qdldl_offset_values(self._h, _pi(idx), C.c_double(off), signs.ctypes.data_as(P_I8),
                                      C.c_int64(len(idx)))

    def refactor(self):
        rc = lib().orc_qdldl_refactor(self._h)
        if rc:
            raise RuntimeError(ERR_NAMES[rc])
        return bool(lib().orc_qdldl_dinv_is_finite(self._h))

    @property
    def nnzL(self):
        return lib().orc_qdldl_nnzL(self._h)

    @property
    def nnzA(self):
        return lib().orc_qdldl_nnzA(self._h)

    @property
    def positive_inertia(self):
        return lib().orc_qdldl_positive_inertia(self._h)

    @property
    def regularize_count(self):
        return lib().orc_qdldl_regularize_count(self._h)

    def _arr(self, name, n, dtype):
        return _view(getattr(lib(), "orc_qdldl_" + name)(self._h), n, dtype)

    @property
    def Lp(self):
        return self._arr("Lp", self.n + 1, i64)

    @property
    def Li(self):
        return self._arr("Li", self.nnzL, i64)

    @property
    def Lx(self):
        return self._arr("Lx", self.nnzL, f64)

    @property
    def D(self):
        return self._arr("D", self.n, f64)

    @property
    def Dinv(self):
        return self._arr("Dinv", self.n, f64)

    @property
    def etree(self):
        return self._arr("etree", self.n, i64)

    @property
    def Lnz(self):
        return self._arr("Lnz", self.n, i64)

    @property
    def triuA(self):
        nnz = self.nnzA
        return (self._arr("Ap", self.n + 1, i64), self._arr("Ai", nnz, i64), self._arr("Ax", nnz, f64))

    @property
    def AtoPAPt(self):
        return self._arr("AtoPAPt", self.nnzA, i64)


# ----------------------------------------------------------------------------
# cones + KKT
# ----------------------------------------------------------------------------
class Cones:
    """CompositeCone subset (compositecone.rs:11-128).  `specs` is a list of
    (tag, dim), (tag, dim, dim2) or (tag, dim, dim2, alpha) [alpha: PowerConeT exponent]."""

    def __init__(self, specs):
        self.specs = [(tuple(s) + (0, 0, 0.5)[len(s) - 1:])[:4] if len(s) < 4 else tuple(s) for s in specs]
        tags = np.array([s[0] for s in self.specs], dtype=np.int32)
        dims = np.array([s[1] for s in self.specs], dtype=i64)
        dims2 = np.array([s[2] for s in self.specs], dtype=i64)
        alphas = np.array([0.5 if s[0] == CONE_GENPOW else s[3] for s in self.specs], dtype=f64)
        self._h = C.c_void_p(lib().orc_cones_new_ex(C.c_int64(len(self.specs)), tags.ctypes.data_as(P_I32),
                                                    _pi(dims), _pi(dims2), _pf(alphas)))
        self.numel = lib().orc_cones_numel(self._h)
        self.nblockvals = lib().orc_cones_nblockvals(self._h)
        self.pdim = lib().orc_cones_pdim(self._h)
        for i, sp in enumerate(self.specs):  # GenPowerConeT(alpha, dim2): (5, len(alpha), dim2, alpha)
            if sp[0] == CONE_GENPOW:
                a = _af(sp[3])
                assert len(a) == sp[1] and abs(a.sum() - 1.0) < 1e-12 and (a > 0).all()
                lib().orc_cones_set_genpow_alpha(self._h, C.c_int64(i), _pf(a))

    @property
    def allows_primal_dual_scaling(self):
        return bool(lib().orc_cones_allows_primal_dual_scaling(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_cones_free(self._h)
            self._h = None

    def update_scaling(self, s, z, mu=1.0, strategy=0):
        """Cone::update_scaling(s, z, mu, strategy); strategy 0 = PrimalDual, 1 = Dual"""
        s, z = _af(s), _af(z)
        return bool(lib().orc_cones_update_scaling_ex(self._h, _pf(s), _pf(z), C.c_double(mu), C.c_int(strategy)))

    def get_Hs(self, init=None):
        Hs = np.zeros(self.nblockvals) if init is None else _af(init).copy()
        lib().orc_cones_get_Hs(self._h, _pf(Hs))
        return Hs

    def mul_Hs(self, x):
        x = _af(x)
        y = np.zeros_like(x)
        lib().orc_cones_mul_Hs(self._h, _pf(y), _pf(x))
        return y

    def affine_ds(self, s):
        s = _af(s)
        ds = np.zeros_like(s)
        lib().orc_cones_affine_ds(self._h, _pf(ds), _pf(s))
        return ds

    def combined_ds_shift(self, step_z, step_s, sigma_mu):
        """returns (shift, step_z', step_s') -- the reference overwrites the steps in place"""
        dz, dsv = _af(step_z).copy(), _af(step_s).copy()
        shift = np.zeros_like(dz)
        lib().orc_cones_combined_ds_shift(self._h, _pf(shift), _pf(dz), _pf(dsv), C.c_double(sigma_mu))
        return shift, dz, dsv

    def ds_from_dz_offset(self, ds, z):
        ds, z = _af(ds), _af(z)
        out = np.zeros_like(ds)
        lib().orc_cones_ds_from_dz_offset(self._h, _pf(out), _pf(ds), _pf(z))
        return out

    def step_length(self, dz, ds, z, s, alpha_max=1.0):
        dz, ds, z, s = _af(dz), _af(ds), _af(z), _af(s)
        return lib().orc_cones_step_length(self._h, _pf(dz), _pf(ds), _pf(z), _pf(s), C.c_double(alpha_max))

    def compute_barrier(self, z, s, dz, ds, alpha):
        z, s, dz, ds = _af(z), _af(s), _af(dz), _af(ds)
        return lib().orc_cones_compute_barrier(self._h, _pf(z), _pf(s), _pf(dz), _pf(ds), C.c_double(alpha))

    def unit_initialization(self, z, s):
        """in place on float64 arrays"""
        lib().orc_cones_unit_initialization(self._h, _pf(z), _pf(s))

    @property
    def is_symmetric(self):
        return bool(lib().orc_cones_is_symmetric(self._h))

    def scaled_unit_shift(self, z, alpha, primal_cone):
        """in place on the float64 array z"""
        lib().orc_cones_scaled_unit_shift(self._h, _pf(z), C.c_double(alpha), C.c_int(1 if primal_cone else 0))

    @property
    def degree(self):
        return int(lib().orc_cones_degree(self._h))

    def margins(self, z):
        z = _af(z)
        a, b = C.c_double(0), C.c_double(0)
        lib().orc_cones_margins(self._h, _pf(z), C.byref(a), C.byref(b))
        return a.value, b.value

    def numel_of(self, i):
        tag, dim, dim2 = self.specs[i][:3]
        if tag in (CONE_EXP, CONE_POW):
            return 3
        if tag == CONE_PSDTRI:
            return dim * (dim + 1) // 2
        if tag == CONE_GENPOW:
            return dim + dim2
        return dim

    def state(self, i):
        L = lib()
        n = self.numel_of(i)
        out = {"eta": L.orc_cone_eta(self._h, C.c_int64(i)), "d": L.orc_cone_d(self._h, C.c_int64(i))}
        for name in ("w", "lambda", "u", "v"):
            p = getattr(L, "orc_cone_" + name)(self._h, C.c_int64(i))
            out[name] = _view(p, n, f64) if p else None
        out["Hs3"] = _view(L.orc_cone_Hs3(self._h, C.c_int64(i)), 6, f64)
        out["Hdual"] = _view(L.orc_cone_Hdual(self._h, C.c_int64(i)), 6, f64)
        out["grad3"] = _view(L.orc_cone_grad3(self._h, C.c_int64(i)), 3, f64)
        return out


class KKTMatrix:
    """assemble_kkt_matrix result (kkt_assembly.rs:20-52)."""

    def __init__(self, h, owned=True):
        self._h = C.c_void_p(h)
        self._owned = owned
        L = lib()
        self.N = L.orc_kktmat_dim(self._h)
        self.nnz = L.orc_kktmat_nnz(self._h)

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_h", None):
            lib().orc_kktmat_free(self._h)
            self._h = None

    @property
    def colptr(self):
        return _view(lib().orc_kktmat_colptr(self._h), self.N + 1, i64)

    @property
    def rowval(self):
        return _view(lib().orc_kktmat_rowval(self._h), self.nnz, i64)

    @property
    def nzval(self):
        return _view(lib().orc_kktmat_nzval(self._h), self.nnz, f64)

    def set_nzval(self, v):
        v = _af(v)
        assert len(v) == self.nnz
        C.memmove(lib().orc_kktmat_nzval(self._h), v.ctypes.data, v.nbytes)

    def map(self, name, n, i=None):
        f = getattr(lib(), "orc_kktmat_map_" + name)
        p = f(self._h) if i is None else f(self._h, C.c_int64(i))
        return _view(p, n, i64)

    @property
    def nsparse(self):
        return lib().orc_kktmat_nsparse(self._h)


def assemble_kkt(n, m, P, A, cones, shape="triu"):
    """P, A: (colptr,rowval,nzval) CSC triples."""
    Pp, Pi, Px = _ai(P[0]), _ai(P[1]), _af(P[2])
    Ap, Ai, Ax = _ai(A[0]), _ai(A[1]), _af(A[2])
    h = lib().orc_assemble_kkt(C.c_int64(n), C.c_int64(m), _pi(Pp), _pi(Pi), _pf(Px), _pi(Ap), _pi(Ai),
                               _pf(Ax), cones._h, C.c_int(0 if shape == "triu" else 1))
    return KKTMatrix(h)


class KKTSolver:
    """DirectLDLKKTSolver (directldlkktsolver.rs:18-405) over the qdldl engine."""

    def __init__(self, n, m, P, A, cones, settings=None, perm=None):
        self.settings = settings or Settings.default()
        self.cones = cones
        Pp, Pi, Px = _ai(P[0]), _ai(P[1]), _af(P[2])
        Ap, Ai, Ax = _ai(A[0]), _ai(A[1]), _af(A[2])
        err = C.c_int(0)
        pp = None
        if perm is not None:
            perm = _ai(perm)
            pp = _pi(perm)
        h = lib().orc_kktsolver_new(C.c_int64(n), C.c_int64(m), _pi(Pp), _pi(Pi), _pf(Px), _pi(Ap), _pi(Ai),
                                    _pf(Ax), cones._h, C.byref(self.settings), pp, C.byref(err))
        if not h:
            raise ValueError(ERR_NAMES.get(err.value, str(err.value)))
        self._h = C.c_void_p(h)
        self.n, self.m = n, m
        self.N = lib().orc_kktsolver_dim(self._h)
        self.p = lib().orc_kktsolver_pdim(self._h)
        self.kkt = KKTMatrix(lib().orc_kktsolver_kktmat(self._h), owned=False)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kktsolver_free(self._h)
            self._h = None

    @property
    def dsigns(self):
        return _view(lib().orc_kktsolver_dsigns(self._h), self.N, np.int8)

    def update(self, hs_override=None):
        ho = None
        if hs_override is not None:
            hs_override = _af(hs_override)
            ho = _pf(hs_override)
        return bool(lib().orc_kktsolver_update(self._h, C.byref(self.settings), ho))

    def setrhs(self, rhsx, rhsz):
        rhsx, rhsz = _af(rhsx), _af(rhsz)
        lib().orc_kktsolver_setrhs(self._h, _pf(rhsx), _pf(rhsz))

    def solve(self):
        x = np.zeros(self.n)
        z = np.zeros(self.m)
        ok = bool(lib().orc_kktsolver_solve(self._h, C.byref(self.settings), _pf(x), _pf(z)))
        return ok, x, z

    def solve_full(self, b):
        b = _af(b)
        x = np.zeros(self.N)
        ok = bool(lib().orc_kktsolver_solve_full(self._h, C.byref(self.settings), _pf(b), _pf(x)))
        return ok, x

    @property
    def last_ir_iters(self):
        return lib().orc_kktsolver_last_ir_iters(self._h)

    @property
    def regularizer(self):
        return lib().orc_kktsolver_regularizer(self._h)

    def ldl_regularize_count(self):
        """dynamic-regularisation hits of the last refactor (qdldl.rs:110-112)"""
        lib().orc_kktsolver_ldl.restype = C.c_void_p
        return int(lib().orc_qdldl_regularize_count(C.c_void_p(lib().orc_kktsolver_ldl(self._h))))


class Variables:
    """DefaultVariables (default/variables.rs:12-36): x[n], s[m], z[m], tau, kappa (host arrays)"""

    def __init__(self, n, m):
        self.x, self.s, self.z = np.zeros(n), np.zeros(m), np.zeros(m)
        self.tau, self.kappa = 1.0, 1.0


class KKTSystem:
    """DefaultKKTSystem (default/kktsystem.rs:16-292) + DefaultResiduals.update (default/residuals.rs:69-111)
    over an existing KKTSolver; q, b, P (triu), A as the solver data."""

    def __init__(self, kktsolver, cones, n, m, P, A, q, b):
        self.ks, self.cones, self.n, self.m = kktsolver, cones, n, m
        Pp, Pi, Px = _ai(P[0]), _ai(P[1]), _af(P[2])
        Ap, Ai, Ax = _ai(A[0]), _ai(A[1]), _af(A[2])
        q, b = _af(q), _af(b)
        self._h = C.c_void_p(lib().orc_kktsystem_new(kktsolver._h, cones._h, C.c_int64(n), C.c_int64(m), _pi(Pp),
                                                     _pi(Pi), _pf(Px), _pi(Ap), _pi(Ai), _pf(Ax), _pf(q), _pf(b)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kktsystem_free(self._h)
            self._h = None

    def update(self, hs_override=None):
        ho = None
        if hs_override is not None:
            hs_override = _af(hs_override)
            ho = _pf(hs_override)
        return bool(lib().orc_kktsystem_update(self._h, C.byref(self.ks.settings), ho))

    def solve(self, lhs, rhs, variables, step_direction):
        """step_direction: 0 = affine, 1 = combined; fills lhs (Variables) -> bool"""
        tk = np.zeros(2)
        ok = bool(lib().orc_kktsystem_solve(
            self._h, C.byref(self.ks.settings), _pf(lhs.x), _pf(lhs.z), _pf(lhs.s), _pf(tk), _pf(_af(rhs.x)),
            _pf(_af(rhs.z)), _pf(_af(rhs.s)), C.c_double(rhs.tau), C.c_double(rhs.kappa), _pf(_af(variables.x)),
            _pf(_af(variables.z)), _pf(_af(variables.s)), C.c_double(variables.tau), C.c_double(variables.kappa),
            C.c_int(step_direction)))
        if ok:
            lhs.tau, lhs.kappa = float(tk[0]), float(tk[1])
        return ok

    def solve_initial_point(self, variables):
        return bool(lib().orc_kktsystem_solve_initial_point(self._h, C.byref(self.ks.settings), _pf(variables.x),
                                                            _pf(variables.s), _pf(variables.z)))

    def residuals(self, variables):
        """-> dict(rx, rz, rx_inf, rz_inf, Px, rtau, dot_qx, dot_bz, dot_sz, dot_xPx)"""
        n, m = self.n, self.m
        rx, rz, rxi, rzi, Px, o = np.zeros(n), np.zeros(m), np.zeros(n), np.zeros(m), np.zeros(n), np.zeros(5)
        lib().orc_residuals_update(self._h, _pf(_af(variables.x)), _pf(_af(variables.z)), _pf(_af(variables.s)),
                                   C.c_double(variables.tau), C.c_double(variables.kappa), _pf(rx), _pf(rz),
                                   _pf(rxi), _pf(rzi), _pf(Px), _pf(o))
        return dict(rx=rx, rz=rz, rx_inf=rxi, rz_inf=rzi, Px=Px, rtau=o[0], dot_qx=o[1], dot_bz=o[2],
                    dot_sz=o[3], dot_xPx=o[4])
