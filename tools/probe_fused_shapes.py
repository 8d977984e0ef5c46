import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
import clarabel_rs_amd.synthetic as problems
def mk(pr):
    return pkg.HipKKTSolver(pkg.CscMatrix(pr["n"], pr["n"], *pr["P"]), pkg.CscMatrix(pr["m"], pr["n"], *pr["A"]), pr["cones"], pr["m"], pr["n"])
for name, pr in (("blockdiag 6 x socp(1,700)", problems.blockdiag([problems.portfolio_socp(1, 700, seed=i) for i in range(6)])),
                 ("socp(6,700) no budget? n/a", problems.portfolio_socp(6, 700, seed=1)),
                 ("blockdiag 3 x socp(2,600)", problems.blockdiag([problems.portfolio_socp(2, 600, seed=i) for i in range(3)])),
                 ("blockdiag 8 x socp(1,200)", problems.blockdiag([problems.portfolio_socp(1, 200, seed=i) for i in range(8)])),
                 ("blockdiag 8 x qp(300,500)", problems.blockdiag([problems.random_qp(300, 500, band=5, seed=i) for i in range(8)]))):
    ks = mk(pr)
    wm = ks.work_model()
    print(name, "N", ks.N, "NF", ks.NF, "step_kernels", ks.step_kernels(), "groups", wm["fold_groups"], "bundles", wm["n_bundles"], "fused_threads", wm["fused_threads"])
