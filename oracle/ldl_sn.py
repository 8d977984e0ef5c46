"""Test / measurement infrastructure only: ctypes binding of oracle/ldl_sn.c -- a SUPERNODAL multi-threaded host
comparator (multifrontal LDL' with relaxed amalgamation, OpenMP over the assembly tree and inside the dense updates),
for the systems with dense fronts on which the reference would pick faer's supernodal engine instead of QDLDL
(ldlsolvers/auto.rs:60-88).  It is NOT the reference and NOT faer; bench.py quotes it as `cpu_baseline_mt` of the
c2 / c5 lines with kind "port-supernodal".  Built on the box it runs on (gcc -O3 -march=native -fopenmp)."""
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
        so = os.path.join(_HERE, "_native", "libldlsn.so")
        src = os.path.join(_HERE, "ldl_sn.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-fPIC", "-std=c11", "-D_GNU_SOURCE", "-fopenmp", "-march=native",
                                   "-shared", "-o", so, src, "-lm"])
        L = C.CDLL(so)
        L.sn_new.restype = C.c_void_p
        L.sn_D.restype = P_F64
        L.sn_perm.restype = P_I64
        L.sn_flops.restype = C.c_double
        for name in ("sn_nsn", "sn_nnzL", "sn_panel_entries", "sn_levels"):
            getattr(L, name).restype = C.c_int64
        _LIB = L
    return _LIB


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class LdlSN:
    """analysis of the upper triangle (colptr, rowval) of a symmetric quasidefinite matrix under the permutation `perm`
    (perm[new] = old; re-postordered inside), then factor(values, signs, eps, delta) / solve(x) / residual(...)"""

    def __init__(self, n, colptr, rowval, perm, threads=1, relax=0.15):
        self.n = int(n)
        self.Ap, self.Ai, self._perm_in = _i(colptr), _i(rowval), _i(perm)
        self.threads = int(threads)
        self._h = C.c_void_p(lib().sn_new(C.c_int64(self.n), self.Ap.ctypes.data_as(P_I64), self.Ai.ctypes.data_as(P_I64),
                                          self._perm_in.ctypes.data_as(P_I64), C.c_int(self.threads), C.c_double(relax)))
        L = lib()
        self.nsn, self.nnzL = int(L.sn_nsn(self._h)), int(L.sn_nnzL(self._h))
        self.panel_entries, self.levels = int(L.sn_panel_entries(self._h)), int(L.sn_levels(self._h))
        self.flops = float(L.sn_flops(self._h))
        self.perm = np.ctypeslib.as_array(L.sn_perm(self._h), shape=(max(self.n, 1),))[:self.n].copy()

    def factor(self, Ax, signs, eps, delta):
        Ax = np.ascontiguousarray(Ax, dtype=np.float64)
        sg = np.ascontiguousarray(signs, dtype=np.int8)
        nreg = C.c_int64(0)
        bad = lib().sn_factor(self._h, Ax.ctypes.data_as(P_F64), sg.ctypes.data_as(C.POINTER(C.c_int8)), C.c_double(eps),
                              C.c_double(delta), C.byref(nreg))
        return bad == 0, int(nreg.value)

    def solve(self, x):
        """x <- K^-1 x (original numbering, in place)"""
        assert x.dtype == np.float64 and x.flags.c_contiguous
        lib().sn_solve(self._h, x.ctypes.data_as(P_F64))

    def residual(self, Ax, x, b, y):
        lib().sn_residual(self._h, self.Ap.ctypes.data_as(P_I64), self.Ai.ctypes.data_as(P_I64), Ax.ctypes.data_as(P_F64),
                          x.ctypes.data_as(P_F64), b.ctypes.data_as(P_F64), y.ctypes.data_as(P_F64))

    def D(self):
        """the pivots in the ORIGINAL numbering"""
        d = np.ctypeslib.as_array(lib().sn_D(self._h), shape=(max(self.n, 1),))[:self.n]
        out = np.empty(self.n)
        out[self.perm] = d
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sn_free(self._h)
            self._h = None
