"""Wall-clock of the host-side analysis (ordering + symbolic factorisation + maps) per phase.
usage: CHIP_TIMING=1 python tools/setup_timing.py c5m|c5|c4|c3|c2   (CHIP_HOST_THREADS=T to pin the thread count)"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

hip = g.load_package()
syn = importlib.import_module("clarabel_rs_amd.synthetic")
which = sys.argv[1] if len(sys.argv) > 1 else "c5m"
t = time.time()
if which in ("c5", "c5m"):
    nc = 200 if which == "c5" else 50
    pr = syn.chordal_sdp(nc, 50, 10, nc, 51, seed=5, with_hs=False)
elif which == "c3":
    pr = syn.portfolio_socp(1000, 1000, seed=3)
elif which == "c4":
    pr = syn.batched_socp(1024, 2000, 2, seed=100)
elif which == "c2":
    pr = syn.random_qp(100000, 200000, band=50, seed=1)
else:
    raise SystemExit("unknown workload " + which)
print("generate %.2fs" % (time.time() - t), flush=True)
P = hip.CscMatrix(pr["n"], pr["n"], *pr["P"])
A = hip.CscMatrix(pr["m"], pr["n"], *pr["A"])
t = time.time()
ks = hip.HipKKTSolver(P, A, pr["cones"], pr["m"], pr["n"], settings=hip.Settings.default(device=hip.DEVICE_HOST_ONLY))
print("host setup %.2fs  (threads: %s)" % (time.time() - t, os.environ.get("CHIP_HOST_THREADS", "default")), flush=True)
info = ks.linear_solver_info()
print("N", ks.N, "nnzL", info.nnzL, "levels", info.n_levels)
