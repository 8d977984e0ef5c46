"""The reference's own end-to-end fixtures (data copied as numbers, not code):
tests/basic_qp.rs:6-42, tests/basic_lp.rs:5-25, tests/basic_socp.rs:5-52 with their asserted
solutions (basic_qp.rs:100-117, basic_lp.rs:27-44, basic_socp.rs:54-70)."""
import numpy as np
import scipy.sparse as sp

ZERO, NN, SOC, EXP, POW, GENPOW, PSD = 0, 1, 2, 3, 4, 5, 6


def _csc(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    return (M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(np.float64))


def _triu(P):
    return _csc(sp.triu(sp.csc_matrix(P), format="csc"))


def basic_qp():
    P = np.array([[4.0, 1.0], [1.0, 2.0]])
    A0 = np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
    A = np.vstack([-A0, A0])
    return dict(n=2, m=6, P=_triu(P), A=_csc(A), q=[1.0, 1.0], b=[-1.0, 0.0, 0.0, 1.0, 0.7, 0.7],
                cones=[(NN, 3), (NN, 3)], x=[0.3, 0.7], obj=1.8800000298331538, tol=1e-6)


def basic_lp():
    I3 = np.eye(3)
    A = 2.0 * np.vstack([I3, -I3])
    return dict(n=3, m=6, P=_csc(sp.csc_matrix((3, 3))), A=_csc(A), q=[3.0, -2.0, 1.0], b=[1.0] * 6,
                cones=[(NN, 3), (NN, 3)], x=[-0.5, 0.5, -0.5], obj=-3.0, tol=1e-8)


def basic_socp(sparse_soc=False):
    P = np.array([[1.4652521089139698, 0.6137176286085666, -1.1527861771130112],
                  [0.6137176286085666, 2.219109946678485, -1.4400420548730628],
                  [-1.1527861771130112, -1.4400420548730628, 1.6014483534926371]])
    I3 = np.eye(3)
    A = np.vstack([2.0 * I3, -2.0 * I3, I3])
    cones = [(NN, 3), (SOC, 6)] if sparse_soc else [(NN, 3), (NN, 3), (SOC, 3)]
    return dict(n=3, m=9, P=_triu(P), A=_csc(A), q=[0.1, -2.0, 1.0], b=[1.0] * 6 + [0.0] * 3, cones=cones,
                x=None if sparse_soc else [-0.5, 0.435603, -0.245459], obj=None if sparse_soc else -8.4590e-01,
                tol=1e-4)


def basic_expcone():
    # tests/basic_expcone.rs:5-36 and :38-56: max x s.t. y exp(x/y) <= z, y == 1, z == exp(5)
    A = np.vstack([-np.eye(3), np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])])
    return dict(n=3, m=5, P=_csc(sp.csc_matrix((3, 3))), A=_csc(A), q=[-1.0, 0.0, 0.0],
                b=[0.0, 0.0, 0.0, 1.0, float(np.exp(5.0))], cones=[(EXP, 3), (ZERO, 2)],
                x=[5.0, 1.0, float(np.exp(5.0))], obj=-5.0, tol=1e-6)


def basic_powcone():
    # tests/basic_powcone.rs:4-47: max x1^0.6 y^0.4 + x2^0.1 s.t. x1 + 2y + 3x2 == 3
    A2 = np.array([[1.0, 2.0, 0.0, 3.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 1.0, 0.0]])
    A = np.vstack([-np.eye(6), A2])
    return dict(n=6, m=8, P=_csc(sp.csc_matrix((6, 6))), A=_csc(A), q=[0.0, 0.0, -1.0, 0.0, 0.0, -1.0],
                b=[0.0] * 6 + [3.0, 1.0], cones=[(POW, 3, 0, 0.6), (POW, 3, 0, 0.1), (ZERO, 2)], x=None,
                obj=-1.8458, tol=1e-3)


def basic_sdp():
    # tests/basic_sdp.rs:6-40: min 0.5 x'x s.t. mat(b - x) PSD (3 x 3, svec coordinates)
    I6 = sp.identity(6, format="csc")
    return dict(n=6, m=6, P=_triu(I6), A=_csc(I6), q=[0.0] * 6, b=[-3.0, 1.0, 4.0, 1.0, 2.0, 5.0],
                cones=[(PSD, 3)],
                x=[-3.0729833267361095, 0.3696004167288786, -0.022226685581313674, 0.31441213129613066,
                   -0.026739700851545107, -0.016084530571308823], obj=4.840076866013861, tol=1e-6)


def basic_genpowcone():
    # tests/basic_genpowcone.rs:4-55: the power-cone problem with GenPowerConeT([0.6, 0.4], 1), ([0.1, 0.9], 1)
    pr = basic_powcone()
    pr["cones"] = [(GENPOW, 2, 1, [0.6, 0.4]), (GENPOW, 2, 1, [0.1, 0.9]), (ZERO, 2)]
    return pr


def basic_unconstrained():
    # tests/basic_unconstrained.rs:4-17: min 0.5 x'x + c'x, no constraints, no cones; x = -c
    return dict(n=3, m=0, P=_triu(sp.identity(3, format="csc")), A=_csc(sp.csc_matrix((0, 3))),
                q=[1.0, 2.0, -3.0], b=[], cones=[], x=[-1.0, -2.0, 3.0], obj=-7.0, tol=1e-6)


def basic_eq_constrained():
    # tests/basic_eq_constrained.rs:34-47: min 0.5 x'x s.t. x2 + x3 = 2, x2 - x3 = 0 (ZeroConeT(2))
    A = np.array([[0.0, 1.0, 1.0], [0.0, 1.0, -1.0]])
    return dict(n=3, m=2, P=_triu(sp.identity(3, format="csc")), A=_csc(A), q=[0.0, 0.0, 0.0], b=[2.0, 0.0],
                cones=[(ZERO, 2)], x=[0.0, 1.0, 1.0], obj=1.0, tol=1e-6)


def mixed_conic():
    # tests/mixed_conic.rs:4-45: x in R^3 stacked into Zero(3), NN(3), SOC(3), Pow(0.5), Exp; optimum 0
    I3 = np.eye(3)
    A = np.vstack([I3] * 5)
    return dict(n=3, m=15, P=_triu(sp.identity(3, format="csc")), A=_csc(A), q=[1.0, 1.0, 1.0], b=[0.0] * 15,
                cones=[(ZERO, 3), (NN, 3), (SOC, 3), (POW, 3, 0, 0.5), (EXP, 3)], x=[0.0, 0.0, 0.0], obj=0.0,
                tol=1e-8)
