"""Host side of the setup of BASELINE config 5 (chordal SDP with <ncliques> cliques, default 200) without touching a GPU
(device = HOST_ONLY): KKT assembly, ordering, symbolic analysis.  CHIP_TIMING=1 prints the stages.
usage: CHIP_TIMING=1 python tools/host_setup_timing.py [ncliques]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g
hip = g.load_package()
import clarabel_rs_amd.synthetic as problems
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0=time.time(); pr = problems.chordal_sdp(nc, 50, 10, nc, 51, seed=5, with_hs=False); print('gen', time.time()-t0)
n, m = pr['n'], pr['m']
st = hip.Settings.default(device=hip.DEVICE_HOST_ONLY)
t0=time.time()
ks = hip.HipKKTSolver(hip.CscMatrix(n, n, *pr["P"]), hip.CscMatrix(m, n, *pr["A"]), pr["cones"], m, n, settings=st)
print('setup', time.time()-t0, 'N', ks.N)

