"""Test / measurement infrastructure only: ctypes binding of oracle/ldl_mt.c -- a MULTI-THREADED host comparator
(the columns of one elimination-tree level of the reference's left-looking LDL', qdldl.rs:469-669, on OpenMP
threads; level-scheduled solves).  It is NOT the reference and NOT its faer engine; bench.py quotes it as
`cpu_baseline_mt` with kind "port-mt".  Built on the box it runs on (gcc -fopenmp -march=native)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
P_I64 = C.POINTER(C.c_int64)
P_F64 = C.POINTER(C.c_double)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_native", "libldlmt.so")
        src = os.path.join(_HERE, "ldl_mt.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-fPIC", "-std=c11", "-D_GNU_SOURCE", "-fopenmp", "-march=native",
                                   "-fno-fast-math", "-ffp-contract=off", "-shared", "-o", so, src, "-lm"])
        L = C.CDLL(so)
        L.orc_mt_new.restype = C.c_void_p
        L.orc_mt_Lx.restype = P_F64
        L.orc_mt_D.restype = P_F64
        _LIB = L
    return _LIB


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class LdlMT:
    """factor / solve / residual of the permuted KKT matrix the oracle's QDLDL engine holds (same pattern of L, same
    elimination tree, same regularised values, same pivot rule)"""

    def __init__(self, orc, ko, perm, threads):
        L = orc.lib()
        L.orc_kktsolver_ldl.restype = C.c_void_p
        f = C.c_void_p(L.orc_kktsolver_ldl(ko._h))
        n = int(L.orc_qdldl_n(f))
        nnzL, nnzA = int(L.orc_qdldl_nnzL(f)), int(L.orc_qdldl_nnzA(f))

        def arr(name, cnt, ctype, dt):
            fn = getattr(L, "orc_qdldl_" + name)
            fn.restype = C.POINTER(ctype)
            return np.ctypeslib.as_array(fn(f), shape=(cnt,)).astype(dt, copy=True)
        self.n = n
        self.Lp, self.Li = arr("Lp", n + 1, C.c_int64, np.int64), arr("Li", max(nnzL, 1), C.c_int64, np.int64)
        self.etree = arr("etree", n, C.c_int64, np.int64)
        self.Ap, self.Ai = arr("Ap", n + 1, C.c_int64, np.int64), arr("Ai", nnzA, C.c_int64, np.int64)
        self._f, self._orc, self._nnzA = f, L, nnzA
        self.perm = _i(perm)
        self.signs = np.ascontiguousarray(np.asarray(ko.dsigns)[self.perm], dtype=np.int8)
        self.threads = int(threads)
        self._h = C.c_void_p(lib().orc_mt_new(C.c_int64(n), self.Lp.ctypes.data_as(P_I64), self.Li.ctypes.data_as(P_I64),
                                              self.etree.ctypes.data_as(P_I64), self.Ap.ctypes.data_as(P_I64),
                                              self.Ai.ctypes.data_as(P_I64), C.c_int(self.threads)))
        self.nnzL = nnzL

    def values(self):
        """the engine's current (regularised) permuted values"""
        fn = self._orc.orc_qdldl_Ax
        fn.restype = P_F64
        return np.ctypeslib.as_array(fn(self._f), shape=(self._nnzA,)).copy()

    def factor(self, Ax, eps, delta):
        rc = C.c_int64(0)
        bad = lib().orc_mt_factor(self._h, Ax.ctypes.data_as(P_F64), self.signs.ctypes.data_as(C.POINTER(C.c_int8)),
                                  C.c_double(eps), C.c_double(delta), C.byref(rc))
        return bad == 0, int(rc.value)

    def solve(self, x):
        lib().orc_mt_solve(self._h, x.ctypes.data_as(P_F64))

    def residual(self, Ax, x, b, y):
        lib().orc_mt_residual(self._h, self.Ap.ctypes.data_as(P_I64), self.Ai.ctypes.data_as(P_I64), Ax.ctypes.data_as(P_F64),
                              x.ctypes.data_as(P_F64), b.ctypes.data_as(P_F64), y.ctypes.data_as(P_F64))

    def Lx(self):
        return np.ctypeslib.as_array(lib().orc_mt_Lx(self._h), shape=(max(self.nnzL, 1),))

    def D(self):
        return np.ctypeslib.as_array(lib().orc_mt_D(self._h), shape=(self.n,))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_mt_free(self._h)
            self._h = None
