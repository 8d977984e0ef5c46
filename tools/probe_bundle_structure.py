import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
import clarabel_rs_amd.synthetic as problems
pr = problems.portfolio_socp(4, 1000, seed=3)
P = pkg.CscMatrix(pr["n"], pr["n"], *pr["P"]); A = pkg.CscMatrix(pr["m"], pr["n"], *pr["A"])
ks = pkg.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"])
et, Lp, Li, lv = ks.symbolic()
N = ks.N
print("N", N, "nnzL", len(Li), "nnzK", ks.nnzK)
perm = ks.perm
n, m = pr["n"], pr["m"]
def kind(o):
    if o < n: return "x"
    if o == n: return "zb"
    if o < n + 1 + n: return "znn"
    if o < n + m: return "zsoc"
    return "uv"
import collections
cl = np.diff(Lp)
lvl = lv
for L in range(int(lvl.max()) + 1):
    idx = np.where(lvl == L)[0]
    kinds = collections.Counter(kind(perm[i]) for i in idx)
    print("level", L, "nodes", len(idx), dict(kinds), "col len hist", dict(collections.Counter(cl[idx].tolist())))
# first 3010 nodes of permuted order
print("first bundle kinds by position:", [(i, kind(perm[i])) for i in (0, 1, 999, 1000, 1001, 1999, 2000, 2001, 2002, 3000, 3001, 3002, 3003)])
