"""clarabel.rs_amd -- host-side mirror (Python/ctypes) of the MI355X-native KKT
backend's C ABI (include/clarabel_hip.h).

The reference's host language is Rust (absent from this image), so the mirror of
its operator interface is written in Python over the C ABI, keeping the
reference's names, argument meaning and error behaviour:

  HipDirectLDLSolver   <->  trait DirectLDLSolver<f64>
                            (src/solver/core/kktsolvers/direct/quasidef/mod.rs:14-26,
                             reference engine ldlsolvers/qdldl.rs:18-107)
  HipKKTSolver         <->  trait KKTSolver<f64> / DirectLDLKKTSolver
                            (src/solver/core/kktsolvers/mod.rs:7-18,
                             quasidef/directldlkktsolver.rs:18-405)

There is NO CPU fallback: every numeric call runs hand-written HIP kernels from
csrc/ through libclarabel_hip.so, and the import fails loudly when that library
is missing.  (The CPU oracle under /oracle is test infrastructure and is never
imported from here.)

Because the directory name contains a dot it cannot be imported with a plain
`import`; use `__graft_entry__.load_package()` (or tests/conftest.py's `hip`
fixture), which loads it under the module name `clarabel_rs_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CLARABEL_HIP_LIB: another build of the same ABI -- tests load the library as it ships, libclarabel_hip_ship.so, through it)
LIB_PATH = os.environ.get("CLARABEL_HIP_LIB") or os.path.join(_HERE, "libclarabel_hip.so")
SHIP_LIB_PATH = os.path.join(_HERE, "libclarabel_hip_ship.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "clarabel_hip.h")

u64 = np.uint64
f64 = np.float64
P_U64 = C.POINTER(C.c_uint64)
P_I64 = C.POINTER(C.c_int64)
P_F64 = C.POINTER(C.c_double)
P_I8 = C.POINTER(C.c_int8)
P_I32 = C.POINTER(C.c_int32)

# chip_status
OK, ERR_DIM, ERR_EMPTY_COLUMN, ERR_NOT_TRIU, ERR_ZERO_PIVOT, ERR_BAD_PERM = 0, -1, -2, -3, -4, -5
ERR_NOT_FACTORED, ERR_NO_DEVICE, ERR_HIP, ERR_ARG, ERR_UNSUPPORTED = -6, -7, -8, -9, -10
STATUS_NAMES = {0: "ok", -1: "IncompatibleDimension", -2: "EmptyColumn", -3: "NotUpperTriangular",
                -4: "ZeroPivot", -5: "InvalidPermutation", -6: "NotFactored", -7: "NoDevice", -8: "HipError",
                -9: "BadArgument", -10: "Unsupported"}
DEVICE_HOST_ONLY = -2

# SupportedConeT tags
ZeroConeT, NonnegativeConeT, SecondOrderConeT, ExponentialConeT, PowerConeT, GenPowerConeT, PSDTriangleConeT = range(7)

# profile families of chip_kkt_profile
PF_NONE, PF_SYMV_T, PF_BWD_T, PF_FWD_T, PF_FACTOR_T = range(5)


class ChipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = ""
        try:
            msg = lib().chip_last_error().decode()
        except Exception:
            pass
        super().__init__("%s: %s (%d) %s" % (where, STATUS_NAMES.get(code, "?"), code, msg))


class Settings(C.Structure):
    """chip_settings == the CoreSettings fields the path consumes
    (src/solver/implementations/default/settings.rs:126-181) + engine knobs."""
    _fields_ = [("static_regularization_enable", C.c_int32), ("static_regularization_constant", C.c_double),
                ("static_regularization_proportional", C.c_double), ("dynamic_regularization_enable", C.c_int32),
                ("dynamic_regularization_eps", C.c_double), ("dynamic_regularization_delta", C.c_double),
                ("iterative_refinement_enable", C.c_int32), ("iterative_refinement_reltol", C.c_double),
                ("iterative_refinement_abstol", C.c_double), ("iterative_refinement_max_iter", C.c_int32),
                ("iterative_refinement_stop_ratio", C.c_double), ("device", C.c_int32),
                ("amd_dense_scale", C.c_double), ("use_graph", C.c_int32), ("reserved0", C.c_int32),
                ("linesearch_backtrack_step", C.c_double), ("min_terminate_step_length", C.c_double),
                ("reserved", C.c_int32 * 2)]

    @staticmethod
    def default(**kw):
        s = Settings()
        lib().chip_settings_default(C.byref(s))
        for k, v in kw.items():
            setattr(s, k, v)
        return s


class Info(C.Structure):
    """chip_info == LinearSolverInfo (src/solver/core/kktsolvers/mod.rs:27-38) + factor statistics."""
    _fields_ = [("name", C.c_char * 16), ("threads", C.c_int64), ("direct", C.c_int32), ("nnzA", C.c_int64),
                ("nnzL", C.c_int64), ("positive_inertia", C.c_int64), ("regularize_count", C.c_int64),
                ("n", C.c_int64), ("n_levels", C.c_int64), ("amd_lnz", C.c_double), ("amd_ndiv", C.c_double),
                ("amd_nmultsubs_ldl", C.c_double), ("last_ir_iterations", C.c_int32),
                ("last_regularizer", C.c_double)]


_LIB = None


def build(verbose=False, testing=True):
    """Compile csrc/ into libclarabel_hip.so for gfx950 (hipcc cross-compiles without a GPU).  testing: with the
    hooks of include/clarabel_hip_testing.h (what the test suite needs; the Makefile's own default leaves them out)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "TESTING=%d" % (1 if testing else 0)]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def build_ship(verbose=False):
    """the library as it ships (TESTING=0: no test hooks) beside the test build: libclarabel_hip_ship.so"""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "ship"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return SHIP_LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: the HIP extension is mandatory (no CPU fallback). "
                              "Run `python -c 'import __graft_entry__ as g; g.build()'`." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.chip_last_error.restype = C.c_char_p
        L.chip_kkt_stream.restype = C.c_void_p
        _LIB = L
    return _LIB


def device_count():
    return int(lib().chip_device_count())


def _u(a):
    return np.ascontiguousarray(a, dtype=u64)


def _f(a):
    return np.ascontiguousarray(a, dtype=f64)


def _pu(a):
    return a.ctypes.data_as(P_U64)


def _pf(a):
    return a.ctypes.data_as(P_F64)


def _check(rc, where):
    if rc < 0:
        raise ChipError(rc, where)
    return rc


def _status(rc, where):
    """1 / 0 -> bool (the reference's is_success), negative -> ChipError"""
    return bool(_check(rc, where))


def auto_select(lnz, n_div, n_mult_subs_ldl):
    """ldl_auto_select's rule (ldlsolvers/auto.rs:62-87): 'qdldl' (simplicial) or 'faer' (supernodal)"""
    return "faer" if lib().chip_auto_select(C.c_double(lnz), C.c_double(n_div), C.c_double(n_mult_subs_ldl)) else "qdldl"


def amd_order(n, colptr, rowval, dense_scale=1.5):
    """AMD ordering of the symmetric matrix with upper triangle (colptr,rowval).
    Replaces `amd::order` at src/qdldl/qdldl.rs:905-917.  Returns (perm, iperm, info3)."""
    colptr, rowval = _u(colptr), _u(rowval)
    perm = np.zeros(n, dtype=u64)
    iperm = np.zeros(n, dtype=u64)
    info = np.zeros(3)
    _check(lib().chip_amd_order(C.c_int64(n), _pu(colptr), _pu(rowval), C.c_double(dense_scale), _pu(perm),
                                _pu(iperm), _pf(info)), "chip_amd_order")
    return perm.astype(np.int64), iperm.astype(np.int64), info


def _supernodes(fn, h):
    cnt = C.c_int64()
    _check(fn(h, C.byref(cnt), None, None), "get_supernodes")
    if cnt.value == 0:
        return []
    ptr = np.zeros(cnt.value + 1, dtype=u64)
    _check(fn(h, C.byref(cnt), _pu(ptr), None), "get_supernodes")
    cols = np.zeros(max(int(ptr[-1]), 1), dtype=u64)
    _check(fn(h, C.byref(cnt), _pu(ptr), _pu(cols)), "get_supernodes")
    return [cols[int(ptr[i]):int(ptr[i + 1])].astype(np.int64) for i in range(cnt.value)]


class CscMatrix:
    """CscMatrix<f64> (src/algebra/csc/core.rs:45-60): m, n, colptr, rowval, nzval."""

    def __init__(self, m, n, colptr, rowval, nzval):
        self.m, self.n = int(m), int(n)
        self.colptr = _u(colptr)
        self.rowval = _u(rowval)
        self.nzval = _f(nzval)
        assert len(self.colptr) == self.n + 1 and len(self.rowval) == len(self.nzval)

    @property
    def nnz(self):
        return int(self.colptr[-1])

    @staticmethod
    def from_scipy(M):
        M = M.tocsc()
        M.sort_indices()
        return CscMatrix(M.shape[0], M.shape[1], M.indptr, M.indices, M.data)


class HipDirectLDLSolver:
    """`direct_solve_method = "hip"`: DirectLDLSolver<f64> on the MI355X.

    new(KKT, Dsigns, settings, perm)  -- ldlsolvers/config.rs:21-22
    required_matrix_shape() = Triu     -- quasidef/mod.rs:14-16"""

    def __init__(self, KKT, Dsigns, settings=None, perm=None):
        assert KKT.m == KKT.n, "KKT matrix is not square"  # ldlsolvers/qdldl.rs:24
        self.settings = settings or Settings.default()
        self.n = KKT.n
        self._h = C.c_void_p()
        ds = np.ascontiguousarray(Dsigns, dtype=np.int8)
        pp = None
        if perm is not None:
            perm = _u(perm)
            pp = _pu(perm)
        _check(lib().chip_ldl_create(C.byref(self._h), C.c_int64(self.n), _pu(KKT.colptr), _pu(KKT.rowval),
                                     _pf(KKT.nzval), ds.ctypes.data_as(P_I8), pp, C.byref(self.settings)),
               "chip_ldl_create")

    @staticmethod
    def required_matrix_shape():
        return "triu"

    def __del__(self):
        if getattr(self, "_h", None):
            lib().chip_ldl_destroy(self._h)
            self._h = None

    def update_values(self, index, values):
        index, values = _u(index), _f(values)
        _check(lib().chip_ldl_update_values(self._h, _pu(index), _pf(values), C.c_int64(len(index))), "update_values")

    def scale_values(self, index, scale):
        index = _u(index)
        _check(lib().chip_ldl_scale_values(self._h, _pu(index), C.c_double(scale), C.c_int64(len(index))),
               "scale_values")

    def offset_values(self, index, offset, signs):
        index = _u(index)
        signs = np.ascontiguousarray(signs, dtype=np.int8)
        assert len(index) == len(signs)  # qdldl.rs:167
        _check(lib().chip_ldl_offset_values(self._h, _pu(index), C.c_double(offset), signs.ctypes.data_as(P_I8),
                                            C.c_int64(len(index))), "offset_values")

    def set_values(self, nzval):
        nzval = _f(nzval)
        _check(lib().chip_ldl_set_values(self._h, _pf(nzval)), "set_values")

    # ---- fast path of the strict drop-in (include/clarabel_hip.h: chip_ldl_register_index ...) ----
    def register_index(self, index, signs=None):
        index = _u(index)
        sg = None if signs is None else np.ascontiguousarray(signs, dtype=np.int8)
        out = C.c_int32()
        _check(lib().chip_ldl_register_index(self._h, _pu(index), C.c_int64(len(index)),
                                             None if sg is None else sg.ctypes.data_as(C.POINTER(C.c_int8)), C.byref(out)),
               "ldl_register_index")
        return int(out.value)

    def update_values_id(self, set_id, values):
        values = _f(values)
        _check(lib().chip_ldl_update_values_id(self._h, C.c_int32(set_id), _pf(values)), "ldl_update_values_id")

    def scale_values_id(self, set_id, scale):
        _check(lib().chip_ldl_scale_values_id(self._h, C.c_int32(set_id), C.c_double(scale)), "ldl_scale_values_id")

    def offset_values_id(self, set_id, offset):
        _check(lib().chip_ldl_offset_values_id(self._h, C.c_int32(set_id), C.c_double(offset)), "ldl_offset_values_id")

    def pin_buffer(self, array):
        _check(lib().chip_ldl_pin_buffer(self._h, C.c_void_p(array.ctypes.data), C.c_uint64(array.nbytes)), "ldl_pin_buffer")

    def solve_refined(self, x, b, settings=None):
        """solve + device-resident refinement (directldlkktsolver.rs:266-321); returns (ok, rounds)"""
        its = C.c_int32()
        rc = lib().chip_ldl_solve_refined(self._h, _pf(x), _pf(_f(b)), None if settings is None else C.byref(settings), C.byref(its))
        return bool(_status(rc, "ldl_solve_refined")), int(its.value)

    def refactor(self, kkt=None):
        """-> bool (all Dinv finite), ldlsolvers/qdldl.rs:98-106"""
        return bool(_check(lib().chip_ldl_refactor(self._h), "refactor"))

    def solve(self, kkt, x, b):
        """x <- K^-1 b; b untouched (ldlsolvers/qdldl.rs:91-96)"""
        b = _f(b)
        assert x.dtype == f64 and x.flags.c_contiguous and len(x) == self.n and len(b) == self.n
        _check(lib().chip_ldl_solve(self._h, _pf(x), _pf(b)), "solve")

    def solve_dev(self, x_ptr, b_ptr):
        _check(lib().chip_ldl_solve_dev(self._h, C.c_void_p(x_ptr), C.c_void_p(b_ptr)), "solve_dev")

    def linear_solver_info(self):
        info = Info()
        _check(lib().chip_ldl_info(self._h, C.byref(info)), "info")
        return info

    @property
    def perm(self):
        p = np.zeros(self.n, dtype=u64)
        _check(lib().chip_ldl_get_perm(self._h, _pu(p)), "get_perm")
        return p.astype(np.int64)

    def symbolic(self):
        info = self.linear_solver_info()
        et = np.zeros(self.n, dtype=u64)
        Lp = np.zeros(self.n + 1, dtype=u64)
        Li = np.zeros(max(info.nnzL, 1), dtype=u64)
        lv = np.zeros(max(self.n, 1), dtype=u64)
        _check(lib().chip_ldl_get_symbolic(self._h, _pu(et), _pu(Lp), _pu(Li), _pu(lv)), "get_symbolic")
        et = et.astype(np.int64)  # UINT64_MAX -> -1
        return et, Lp.astype(np.int64), Li[:info.nnzL].astype(np.int64), lv[:self.n].astype(np.int64)

    def supernodes(self):
        return _supernodes(lib().chip_ldl_get_supernodes, self._h)

    def factors(self):
        info = self.linear_solver_info()
        Lp = np.zeros(self.n + 1, dtype=u64)
        Li = np.zeros(max(info.nnzL, 1), dtype=u64)
        Lx = np.zeros(max(info.nnzL, 1))
        D = np.zeros(self.n)
        Dinv = np.zeros(self.n)
        _check(lib().chip_ldl_get_factors(self._h, _pu(Lp), _pu(Li), _pf(Lx), _pf(D), _pf(Dinv)), "get_factors")
        return Lp.astype(np.int64), Li[:info.nnzL].astype(np.int64), Lx[:info.nnzL], D, Dinv


class HipKKTSolver:
    """KKTSolver<f64> (kktsolvers/mod.rs:7-18) == DirectLDLKKTSolver on the device.

    new(P, A, cones, m, n, settings) -- directldlkktsolver.rs:60-118.
    `cones`: list of (tag, dim), (tag, dim, dim2) or (tag, dim, dim2, alpha) SupportedConeT
    descriptors (alpha = PowerConeT exponent)."""

    def __init__(self, P, A, cones, m, n, settings=None, perm=None):
        assert P.n == n and A.n == n and A.m == m
        self.settings = settings or Settings.default()
        self.cones = [(tuple(c) + (0, 0, 0.5)[len(c) - 1:])[:4] if len(c) < 4 else tuple(c) for c in cones]
        tags = np.array([c[0] for c in self.cones], dtype=np.int32)
        dims = np.array([c[1] for c in self.cones], dtype=np.int64)
        dims2 = np.array([c[2] for c in self.cones], dtype=np.int64)
        alphas = np.array([0.5 if c[0] == 5 else c[3] for c in self.cones], dtype=np.float64)
        self._h = C.c_void_p()
        pp = None
        if perm is not None:
            perm = _u(perm)
            pp = _pu(perm)
        _check(lib().chip_kkt_create(C.byref(self._h), C.c_int64(n), C.c_int64(m), _pu(P.colptr), _pu(P.rowval),
                                     _pf(P.nzval), _pu(A.colptr), _pu(A.rowval), _pf(A.nzval),
                                     C.c_int64(len(self.cones)), tags.ctypes.data_as(P_I32),
                                     dims.ctypes.data_as(P_I64), dims2.ctypes.data_as(P_I64), _pf(alphas),
                                     C.byref(self.settings), pp), "chip_kkt_create")
        if self.settings.device != DEVICE_HOST_ONLY:
            for i, c in enumerate(self.cones):  # GenPowerConeT(alpha, dim2) == (5, len(alpha), dim2, alpha)
                if c[0] == 5:
                    a = _f(c[3])
                    assert len(a) == c[1]
                    _check(lib().chip_kkt_set_genpow_alpha(self._h, C.c_int64(i), _pf(a)), "set_genpow_alpha")
        d = (C.c_int64 * 8)()
        lib().chip_kkt_dims(self._h, d)
        self.n, self.m, self.p, self.N, self.nnzK, self.nHs, self.NF, self.nnzU = [int(v) for v in d]
        self.nnzP, self.nnzA = P.nnz, A.nnz

    def __del__(self):
        if getattr(self, "_h", None):
            lib().chip_kkt_destroy(self._h)
            self._h = None

    # -- layout introspection (host) ------------------------------------------
    def kkt_matrix(self):
        cp = np.zeros(self.N + 1, dtype=u64)
        rv = np.zeros(max(self.nnzK, 1), dtype=u64)
        nz = np.zeros(max(self.nnzK, 1))
        _check(lib().chip_kkt_get_matrix(self._h, _pu(cp), _pu(rv), _pf(nz)), "get_matrix")
        return CscMatrix(self.N, self.N, cp, rv[:self.nnzK], nz[:self.nnzK])

    def maps(self):
        mP = np.zeros(max(self.nnzP, 1), dtype=u64)
        mA = np.zeros(max(self.nnzA, 1), dtype=u64)
        mH = np.zeros(max(self.nHs, 1), dtype=u64)
        dP = np.zeros(max(self.n, 1), dtype=u64)
        dF = np.zeros(max(self.N, 1), dtype=u64)
        ds = np.zeros(max(self.N, 1), dtype=np.int8)
        _check(lib().chip_kkt_get_map(self._h, _pu(mP), _pu(mA), _pu(mH), _pu(dP), _pu(dF),
                                      ds.ctypes.data_as(P_I8)), "get_map")
        return {"P": mP[:self.nnzP].astype(np.int64), "A": mA[:self.nnzA].astype(np.int64),
                "Hsblocks": mH[:self.nHs].astype(np.int64), "diagP": dP[:self.n].astype(np.int64),
                "diag_full": dF[:self.N].astype(np.int64), "dsigns": ds[:self.N]}

    @property
    def perm(self):
        p = np.zeros(self.N, dtype=u64)
        _check(lib().chip_kkt_get_perm(self._h, _pu(p)), "get_perm")
        return p.astype(np.int64)

    def symbolic(self):
        info = self.linear_solver_info()
        et = np.zeros(self.N, dtype=u64)
        Lp = np.zeros(self.N + 1, dtype=u64)
        Li = np.zeros(max(info.nnzL, 1), dtype=u64)
        lv = np.zeros(max(self.N, 1), dtype=u64)
        _check(lib().chip_kkt_get_symbolic(self._h, _pu(et), _pu(Lp), _pu(Li), _pu(lv)), "get_symbolic")
        return et.astype(np.int64), Lp.astype(np.int64), Li[:info.nnzL].astype(np.int64), lv[:self.N].astype(np.int64)

    def supernodes(self):
        """[(columns ascending, permuted numbering), ...] of the chain supernodes"""
        return _supernodes(lib().chip_kkt_get_supernodes, self._h)

    def values(self):
        nz = np.zeros(max(self.nnzK, 1))
        _check(lib().chip_kkt_get_values(self._h, _pf(nz)), "get_values")
        return nz[:self.nnzK]

    # -- the KKTSolver trait -----------------------------------------------------
    def update_scaling(self, s, z, mu=1.0, strategy=0):
        """cones.update_scaling(s, z, mu, strategy) for the device-held cones -> bool
        (strategy 0 = PrimalDual, 1 = Dual; mu only matters for Exp/Pow cones under Dual)"""
        s, z = _f(s), _f(z)
        assert len(s) == self.m and len(z) == self.m
        return bool(_check(lib().chip_kkt_update_scaling(self._h, _pf(s), _pf(z), C.c_double(mu),
                                                         C.c_int32(strategy)), "update_scaling"))

    def update_scaling_dev(self, s_ptr, z_ptr, mu=1.0, strategy=0):
        return bool(_check(lib().chip_kkt_update_scaling_dev(self._h, C.c_void_p(s_ptr), C.c_void_p(z_ptr),
                                                             C.c_double(mu), C.c_int32(strategy)),
                           "update_scaling_dev"))

    def update(self, hsblocks=None):
        """KKTSolver::update(cones, settings) -> bool"""
        hp = None
        if hsblocks is not None:
            hsblocks = _f(hsblocks)
            assert len(hsblocks) == self.nHs
            hp = _pf(hsblocks)
        return bool(_check(lib().chip_kkt_update(self._h, hp), "update"))

    def setrhs(self, rhsx, rhsz):
        rhsx, rhsz = _f(rhsx), _f(rhsz)
        assert len(rhsx) == self.n and len(rhsz) == self.m
        _check(lib().chip_kkt_setrhs(self._h, _pf(rhsx), _pf(rhsz)), "setrhs")

    def setrhs_dev(self, x_ptr, z_ptr):
        _check(lib().chip_kkt_setrhs_dev(self._h, C.c_void_p(x_ptr), C.c_void_p(z_ptr)), "setrhs_dev")

    def solve(self, lhsx=None, lhsz=None):
        """KKTSolver::solve(lhsx, lhsz, settings) -> bool"""
        px = _pf(lhsx) if lhsx is not None else None
        pz = _pf(lhsz) if lhsz is not None else None
        return bool(_check(lib().chip_kkt_solve(self._h, px, pz), "solve"))

    def solve_dev(self, x_ptr, z_ptr):
        return bool(_check(lib().chip_kkt_solve_dev(self._h, C.c_void_p(x_ptr) if x_ptr else None,
                                                    C.c_void_p(z_ptr) if z_ptr else None), "solve_dev"))

    def solve_full(self, b):
        b = _f(b)
        assert len(b) == self.N
        x = np.zeros(self.N)
        ok = bool(_check(lib().chip_kkt_solve_full(self._h, _pf(x), _pf(b)), "solve_full"))
        return ok, x

    def update_P(self, Pnzval):
        Pnzval = _f(Pnzval)
        assert len(Pnzval) == self.nnzP
        _check(lib().chip_kkt_update_P(self._h, _pf(Pnzval)), "update_P")

    def update_A(self, Anzval):
        Anzval = _f(Anzval)
        assert len(Anzval) == self.nnzA
        _check(lib().chip_kkt_update_A(self._h, _pf(Anzval)), "update_A")

    def mul_Hs_dev(self, y_ptr, x_ptr):
        _check(lib().chip_kkt_mul_Hs_dev(self._h, C.c_void_p(y_ptr), C.c_void_p(x_ptr)), "mul_Hs_dev")

    # -- CompositeCone operations either side of the solve (device pointers, m doubles) -----
    def affine_ds_dev(self, ds_ptr, s_ptr):
        _check(lib().chip_kkt_affine_ds_dev(self._h, C.c_void_p(ds_ptr), C.c_void_p(s_ptr)), "affine_ds")

    def combined_ds_shift_dev(self, shift_ptr, step_z_ptr, step_s_ptr, sigma_mu):
        _check(lib().chip_kkt_combined_ds_shift_dev(self._h, C.c_void_p(shift_ptr), C.c_void_p(step_z_ptr),
                                                    C.c_void_p(step_s_ptr), C.c_double(sigma_mu)), "combined_ds_shift")

    def ds_from_dz_offset_dev(self, out_ptr, ds_ptr, z_ptr):
        _check(lib().chip_kkt_ds_from_dz_offset_dev(self._h, C.c_void_p(out_ptr), C.c_void_p(ds_ptr),
                                                    C.c_void_p(z_ptr)), "ds_from_dz_offset")

    def step_length_dev(self, dz_ptr, ds_ptr, z_ptr, s_ptr, alpha_max=1.0):
        a = C.c_double(0)
        _check(lib().chip_kkt_step_length_dev(self._h, C.c_void_p(dz_ptr), C.c_void_p(ds_ptr), C.c_void_p(z_ptr),
                                              C.c_void_p(s_ptr), C.c_double(alpha_max), C.byref(a)), "step_length")
        return a.value

    def margins_dev(self, z_ptr):
        a, b = C.c_double(0), C.c_double(0)
        _check(lib().chip_kkt_margins_dev(self._h, C.c_void_p(z_ptr), C.byref(a), C.byref(b)), "margins")
        return a.value, b.value

    def scaled_unit_shift_dev(self, z_ptr, alpha, primal_cone):
        _check(lib().chip_kkt_scaled_unit_shift_dev(self._h, C.c_void_p(z_ptr), C.c_double(alpha),
                                                    C.c_int32(1 if primal_cone else 0)), "scaled_unit_shift")

    def unit_initialization_dev(self, z_ptr, s_ptr):
        _check(lib().chip_kkt_unit_initialization_dev(self._h, C.c_void_p(z_ptr), C.c_void_p(s_ptr)),
               "unit_initialization")

    def compute_barrier_dev(self, z_ptr, s_ptr, dz_ptr, ds_ptr, alpha):
        out = C.c_double(0)
        _check(lib().chip_kkt_compute_barrier_dev(self._h, C.c_void_p(z_ptr), C.c_void_p(s_ptr),
                                                  C.c_void_p(dz_ptr), C.c_void_p(ds_ptr), C.c_double(alpha),
                                                  C.byref(out)), "compute_barrier")
        return out.value

    def degree(self):
        d = C.c_int64()
        _check(lib().chip_kkt_degree(self._h, C.byref(d)), "degree")
        return d.value

    def linear_solver_info(self):
        info = Info()
        _check(lib().chip_kkt_info(self._h, C.byref(info)), "info")
        return info

    def synchronize(self):
        _check(lib().chip_kkt_synchronize(self._h), "synchronize")

    # -- asynchronous variants: enqueue a whole iteration, collect the verdicts once ------------------
    def update_enqueue(self, hsblocks=None):
        hb = None if hsblocks is None else _f(hsblocks)
        _check(lib().chip_kkt_update_enqueue(self._h, None if hb is None else _pf(hb)), "update_enqueue")

    def update_scaled_enqueue(self, s_ptr, z_ptr, mu=1.0, strategy=0, hsblocks=None):
        """cones.update_scaling + the KKT update as ONE enqueue (core/solver.rs:334-352), verdicts with collect()"""
        hb = None if hsblocks is None else _f(hsblocks)
        _check(lib().chip_kkt_update_scaled_enqueue(self._h, C.c_void_p(s_ptr), C.c_void_p(z_ptr), C.c_double(mu),
                                                    C.c_int32(strategy), None if hb is None else _pf(hb)),
               "update_scaled_enqueue")

    def solve_dev_enqueue(self, x_ptr, z_ptr):
        _check(lib().chip_kkt_solve_dev_enqueue(self._h, C.c_void_p(x_ptr), C.c_void_p(z_ptr)), "solve_dev_enqueue")

    def solve2_dev_enqueue(self, rxa, rza, lxa, lza, rxb, rzb, lxb, lzb):
        """two independent solves as one call (chip_kkt_solve2_dev_enqueue): device pointers of the two right-hand sides
        and of the two results"""
        args = [C.c_void_p(int(p)) for p in (rxa, rza, lxa, lza, rxb, rzb, lxb, lzb)]
        _check(lib().chip_kkt_solve2_dev_enqueue(self._h, *args), "solve2_dev_enqueue")

    def collect(self):
        """-> (update_ok, [solve_ok, ...]) of everything enqueued since the last collect; one synchronisation.
        self.repeated_solves: indices (into that list) of solves whose fused launch timed out and that collect repeated
        on the one-kernel-per-phase path -- their lhs held garbage until now: device work enqueued behind them that
        consumed it (an all-gather, a dependent right-hand side) must be re-issued"""
        uok, n = C.c_int32(1), C.c_int32(0)
        sok = (C.c_int32 * 16)()
        _check(lib().chip_kkt_collect(self._h, C.byref(uok), C.byref(n), sok), "collect")
        cnt = min(n.value, 16)
        self.repeated_solves = [i for i in range(cnt) if sok[i] == 2]
        return bool(uok.value), [bool(sok[i]) for i in range(cnt)]

    def set_settings(self, settings):
        """the reference passes `settings` to update() / solve() on every call (kktsolvers/mod.rs:7-18)"""
        _check(lib().chip_kkt_set_settings(self._h, C.byref(settings)), "set_settings")
        self.settings = settings

    def scaling_ok(self):
        return bool(_status(lib().chip_kkt_scaling_ok(self._h), "scaling_ok"))

    def profile(self, family):
        _check(lib().chip_kkt_profile(self._h, C.c_int32(family)), "profile")

    def profile_read(self):
        out = (C.c_double * 8)()
        _check(lib().chip_kkt_profile_read(self._h, out), "profile_read")
        return {"launches": int(out[0]), "ms": float(out[1]), "family": int(out[2])}

    def work_model(self):
        """work of the chain-supernode kernels per refactor / per sweep (chip_kkt_work_model)"""
        out = (C.c_double * 8)()
        _check(lib().chip_kkt_work_model(self._h, out), "work_model")
        return {"sn_update_flops": float(out[0]), "sn_panel_entries": float(out[1]), "sn_extend_flops": float(out[2]),
                "sn_diag_rows_flops": float(out[3]), "n_supernodes": int(out[4]), "fold_groups": int(out[5]),
                "n_bundles": int(out[6]), "fused_threads": int(out[7])}

    def sweep_model(self):
        """how the substitutions run through the chain supernodes (chip_kkt_sweep_model)"""
        out = (C.c_double * 4)()
        _check(lib().chip_kkt_sweep_model(self._h, out), "sweep_model")
        return {"g_doubles": float(out[0]), "g_levels": int(out[1]), "sn_levels": int(out[2]), "g_build_launches": int(out[3])}

    def fused_fallbacks(self):
        return int(lib().chip_kkt_fused_fallbacks(self._h))

    def step_kernels(self):
        """bit 0: the fused solve is k_gstep_solve, bit 1: the bundle factorisation is k_gstep_factor (grouped fold)"""
        return int(lib().chip_kkt_step_kernels(self._h))


class CVars(C.Structure):
    """chip_vars: DefaultVariables (default/variables.rs:12-36) with device pointers"""
    _fields_ = [("x", C.c_void_p), ("z", C.c_void_p), ("s", C.c_void_p), ("tau", C.c_double),
                ("kappa", C.c_double)]


class DeviceVariables:
    """x[n], s[m], z[m] in HBM + tau, kappa on the host (DefaultVariables::new, variables.rs:41-50)"""

    def __init__(self, n, m):
        self.n, self.m = n, m
        self.x, self.s, self.z = DeviceArray(n), DeviceArray(m), DeviceArray(m)
        self.tau, self.kappa = 1.0, 1.0

    def cvars(self):
        return CVars(self.x.ptr, self.z.ptr, self.s.ptr, self.tau, self.kappa)


STEP_AFFINE, STEP_COMBINED = 0, 1


class HipKKTSystem:
    """DefaultKKTSystem (default/kktsystem.rs:16-292) + DefaultResiduals::update
    (default/residuals.rs:69-111), device resident, over an existing HipKKTSolver."""

    def __init__(self, kktsolver, P, A, q, b):
        self.ks = kktsolver
        self.n, self.m = kktsolver.n, kktsolver.m
        self._h = C.c_void_p()
        q, b = _f(q), _f(b)
        _check(lib().chip_kktsystem_create(C.byref(self._h), kktsolver._h, _pu(P.colptr), _pu(P.rowval),
                                           _pf(P.nzval), _pu(A.colptr), _pu(A.rowval), _pf(A.nzval), _pf(q),
                                           _pf(b)), "chip_kktsystem_create")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().chip_kktsystem_destroy(self._h)
            self._h = C.c_void_p()

    def update(self):
        return _status(lib().chip_kktsystem_update(self._h), "kktsystem_update")

    def solve(self, lhs, rhs, variables, step_direction):
        """lhs, rhs, variables: DeviceVariables; lhs.tau / lhs.kappa are written -> bool"""
        cl, cr, cv = lhs.cvars(), rhs.cvars(), variables.cvars()
        ok = _status(lib().chip_kktsystem_solve(self._h, C.byref(cl), C.byref(cr), C.byref(cv),
                                                C.c_int32(step_direction)), "kktsystem_solve")
        if ok:
            lhs.tau, lhs.kappa = cl.tau, cl.kappa
        return ok

    def solve_initial_point(self, variables):
        cv = variables.cvars()
        return _status(lib().chip_kktsystem_solve_initial_point(self._h, C.byref(cv)), "solve_initial_point")

    def residuals_update(self, variables, rx, rz, rx_inf, rz_inf, Px, norms=False):
        """device outputs (DeviceArray) + dict of the host scalars of residuals.rs:103-110; with
        norms=True also 'norms' = (||x||, ||z||, ||s||, ||rz||, ||rx||) from the same synchronisation"""
        cv = variables.cvars()
        o = (C.c_double * 5)()
        nrm = (C.c_double * 5)() if norms else None
        _check(lib().chip_residuals_update_norms(self._h, C.byref(cv), C.c_void_p(rx.ptr), C.c_void_p(rz.ptr),
                                                 C.c_void_p(rx_inf.ptr), C.c_void_p(rz_inf.ptr), C.c_void_p(Px.ptr),
                                                 o, nrm), "residuals_update")
        out = dict(rtau=o[0], dot_qx=o[1], dot_bz=o[2], dot_sz=o[3], dot_xPx=o[4])
        if norms:
            out["norms"] = tuple(nrm)
        return out

    # ---- DefaultVariables on the device (default/variables.rs:58-261) -------------------------
    @staticmethod
    def _p(a):
        return C.c_void_p(a if isinstance(a, int) else a.ptr)

    def calc_mu(self, variables, dot_sz):
        cv, o = variables.cvars(), C.c_double()
        _check(lib().chip_variables_calc_mu(self._h, C.byref(cv), C.c_double(dot_sz), C.byref(o)), "calc_mu")
        return o.value

    def affine_step_rhs(self, d, rx, rz, rtau, variables):
        cd, cv = d.cvars(), variables.cvars()
        _check(lib().chip_variables_affine_step_rhs(self._h, C.byref(cd), self._p(rx), self._p(rz),
                                                    C.c_double(rtau), C.byref(cv)), "affine_step_rhs")
        d.tau, d.kappa = cd.tau, cd.kappa

    def combined_step_rhs(self, d, rx, rz, rtau, variables, step, sigma, mu, m):
        cd, cv, cs = d.cvars(), variables.cvars(), step.cvars()
        _check(lib().chip_variables_combined_step_rhs(self._h, C.byref(cd), self._p(rx), self._p(rz),
                                                      C.c_double(rtau), C.byref(cv), C.byref(cs),
                                                      C.c_double(sigma), C.c_double(mu), C.c_double(m)),
               "combined_step_rhs")
        d.tau, d.kappa = cd.tau, cd.kappa

    def calc_step_length(self, variables, step, step_direction, max_step_fraction=0.99):
        cv, cs, o = variables.cvars(), step.cvars(), C.c_double()
        _check(lib().chip_variables_calc_step_length(self._h, C.byref(cv), C.byref(cs), C.c_int32(step_direction),
                                                     C.c_double(max_step_fraction), C.byref(o)), "calc_step_length")
        return o.value

    def add_step(self, variables, step, alpha):
        cv, cs = variables.cvars(), step.cvars()
        _check(lib().chip_variables_add_step(self._h, C.byref(cv), C.byref(cs), C.c_double(alpha)), "add_step")
        variables.tau, variables.kappa = cv.tau, cv.kappa

    def symmetric_initialization(self, variables):
        cv = variables.cvars()
        _check(lib().chip_variables_symmetric_initialization(self._h, C.byref(cv)), "symmetric_initialization")
        variables.tau, variables.kappa = cv.tau, cv.kappa

    def unit_initialization(self, variables):
        cv = variables.cvars()
        _check(lib().chip_variables_unit_initialization(self._h, C.byref(cv)), "unit_initialization")
        variables.tau, variables.kappa = cv.tau, cv.kappa

    def barrier(self, variables, step, alpha):
        cv, cs, o = variables.cvars(), step.cvars(), C.c_double()
        _check(lib().chip_variables_barrier(self._h, C.byref(cv), C.byref(cs), C.c_double(alpha), C.byref(o)),
               "barrier")
        return o.value

    def rescale(self, variables):
        cv = variables.cvars()
        _check(lib().chip_variables_rescale(self._h, C.byref(cv)), "rescale")
        variables.tau, variables.kappa = cv.tau, cv.kappa

    def vec_norms(self, *arrays):
        """Euclidean norms of up to 8 DeviceArrays, one host synchronisation"""
        k = len(arrays)
        ptrs = (C.c_void_p * max(k, 1))(*[a.ptr for a in arrays])
        lens = (C.c_int64 * max(k, 1))(*[a.n for a in arrays])
        out = (C.c_double * max(k, 1))()
        _check(lib().chip_vec_norms(self._h, C.c_int32(k), ptrs, lens, out), "vec_norms")
        return [out[i] for i in range(k)]

    def update_data(self, P=None, A=None, q=None, b=None):
        args = [None if v is None else _f(v) for v in (P, A, q, b)]
        _check(lib().chip_kktsystem_update_data(self._h, *[None if v is None else _pf(v) for v in args]),
               "kktsystem_update_data")


# ---------------------------------------------------------------------------
# sharded path: RCCL communicator of the C ABI (csrc/comm.cpp)
# ---------------------------------------------------------------------------
COMM_ID_BYTES = 128


def comm_unique_id():
    """rank 0: the rendezvous token to hand to the other ranks out of band (bytes)"""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().chip_comm_get_unique_id(buf), "chip_comm_get_unique_id")
    return bytes(buf)


class Comm:
    """one per process / GPU; collective construction over all ranks"""

    def __init__(self, unique_id, world, rank, device=-1):
        assert len(unique_id) == COMM_ID_BYTES
        self._h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().chip_comm_create(C.byref(self._h), buf, C.c_int32(world), C.c_int32(rank), C.c_int32(device)),
               "chip_comm_create")
        self.world, self.rank = world, rank

    def __del__(self):
        if getattr(self, "_h", None):
            lib().chip_comm_destroy(self._h)
            self._h = None

    def attach(self, kkt):
        _check(lib().chip_kkt_attach_comm(kkt._h, self._h), "attach_comm")

    def allgather_step(self, kkt, send_ptr, recv_ptr, counts):
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        assert len(cnt) == self.world
        _check(lib().chip_kkt_allgather_step(kkt._h, self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr),
                                             cnt.ctypes.data_as(P_I64)), "allgather_step")

    def wait(self, kkt):
        """kkt's stream waits on the device for the last all-gather"""
        _check(lib().chip_kkt_wait_comm(kkt._h, self._h), "wait_comm")

    def debug_spin(self, blocks, threads=256, usec=60.0):
        """test hook: a spinner on the communicator's stream behind the last collective (chip_comm_debug_spin)"""
        lib().chip_comm_debug_spin.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double]
        _check(lib().chip_comm_debug_spin(self._h, blocks, threads, float(usec)), "comm_debug_spin")

    def synchronize(self):
        _check(lib().chip_comm_synchronize(self._h), "comm_synchronize")

    def allreduce(self, vals, op="sum"):
        v = np.ascontiguousarray(np.atleast_1d(vals), dtype=np.float64).copy()
        _check(lib().chip_comm_allreduce(self._h, _pf(v), C.c_int32(len(v)), C.c_int32({"sum": 0, "min": 1, "max": 2}[op])),
               "comm_allreduce")
        return v

    def barrier(self):
        self.allreduce([0.0])


# ---------------------------------------------------------------------------
# raw HBM buffers without torch (tests / single-GPU bench plumbing): thin ctypes
# calls into the SAME libamdhip64 instance the extension is linked against.
# NB when torch is used in the process, import torch BEFORE this package so that
# both share torch's bundled HIP runtime (two runtimes cannot both own the GPU).
# ---------------------------------------------------------------------------
_HIPRT = None


def _hiprt():
    global _HIPRT
    if _HIPRT is None:
        lib()
        _HIPRT = C.CDLL("libamdhip64.so.7")
    return _HIPRT


class DeviceArray:
    """n fp64 values in HBM on the current device"""

    def __init__(self, n_or_array):
        rt = _hiprt()
        host = None
        if not np.isscalar(n_or_array):
            host = _f(n_or_array)
            n = len(host)
        else:
            n = int(n_or_array)
        self.n = n
        self._p = C.c_void_p()
        rc = rt.hipMalloc(C.byref(self._p), C.c_size_t(max(n, 1) * 8))
        if rc != 0:
            raise RuntimeError("hipMalloc failed: %d" % rc)
        if host is not None:
            self.copy_from(host)
        else:
            rt.hipMemset(self._p, 0, C.c_size_t(max(n, 1) * 8))

    @classmethod
    def view(cls, ptr, n):
        """non-owning view of n fp64 values at device address `ptr` (never freed here)"""
        a = cls.__new__(cls)
        a.n = int(n)
        a._p = C.c_void_p(ptr)
        a._borrowed = True
        return a

    @property
    def ptr(self):
        return self._p.value

    def copy_from(self, host):
        host = _f(host)
        assert len(host) == self.n
        rc = _hiprt().hipMemcpy(self._p, host.ctypes.data_as(C.c_void_p), C.c_size_t(self.n * 8), C.c_int(1))
        if rc != 0:
            raise RuntimeError("hipMemcpy H2D failed: %d" % rc)

    def numpy(self):
        out = np.zeros(self.n)
        rc = _hiprt().hipMemcpy(out.ctypes.data_as(C.c_void_p), self._p, C.c_size_t(self.n * 8), C.c_int(2))
        if rc != 0:
            raise RuntimeError("hipMemcpy D2H failed: %d" % rc)
        return out

    def __del__(self):
        if getattr(self, "_borrowed", False):
            return
        if getattr(self, "_p", None) and self._p.value:
            try:
                _hiprt().hipFree(self._p)
            except Exception:
                pass
            self._p = C.c_void_p()


def debug_spin(device, blocks, threads=256, lds_bytes=0, usec=1000.0):
    """test hook (include/clarabel_hip_testing.h; CHIP_TESTING builds): a co-resident kernel that only spins;
    blocks = 0 waits for the spinners"""
    lib().chip_debug_spin.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double]
    _check(lib().chip_debug_spin(device, blocks, threads, lds_bytes, float(usec)), "debug_spin")


def debug_set_switch(name, value=None):
    """test hook: set (value given) or clear one CHIP_* diagnostic switch and re-parse the switch table
    (csrc/switches.hpp) -- for handles that already exist; the environment is read whenever a handle is created"""
    lib().chip_debug_set_switch.argtypes = [C.c_char_p, C.c_char_p]
    _check(lib().chip_debug_set_switch(name.encode(), None if value is None else str(value).encode()), "debug_set_switch")


def debug_counter(kkt, name):
    """test hook: one structural figure of a HipKKTSolver by name (include/clarabel_hip_testing.h: chip_debug_counter)"""
    out = C.c_double(0.0)
    lib().chip_debug_counter.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
    _check(lib().chip_debug_counter(kkt._h, name.encode(), C.byref(out)), "debug_counter")
    return out.value


def set_device(ordinal):
    """hipSetDevice for this thread (one process per GPU: the rank's local device)"""
    rc = _hiprt().hipSetDevice(C.c_int(int(ordinal)))
    if rc != 0:
        raise RuntimeError("hipSetDevice(%d) failed: %d" % (ordinal, rc))


def device_synchronize():
    _hiprt().hipDeviceSynchronize()
