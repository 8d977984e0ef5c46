#!/usr/bin/env python3
"""Whole interior-point iterations with every vector resident in HBM (tests/ipm_device.py: L2 + L3 +
chip_variables_* of the C ABI; only scalars cross the boundary) on a complete, feasible problem of
the config-3 shape (tests/problems.py:portfolio_problem).  Measures what SURVEY 8(f) rows 2-3 add
around the KKT work that bench.py times: residuals, cone scaling, step right-hand sides, step
lengths, variable updates.  Prints one JSON line.

usage: python tools/ipm_scale.py [nblocks blocksize] [--phases] [--max-iter K]
  --phases   bracket every phase with stream synchronisations and report ms per phase per iteration
             (slower overall: the run without it is the throughput number)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import __graft_entry__ as g
from tests import ipm_device
import __graft_entry__ as _graft_entry
_graft_entry.load_package()
import clarabel_rs_amd.synthetic as problems


def main():
    import torch  # noqa: F401  (shares its HIP runtime with the extension; must be imported first)
    hip = g.load_package()
    nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
    nb, bs = (nums + [1000, 1000])[:2] if len(nums) >= 2 else (1000, 1000)
    max_iter = 200
    if "--max-iter" in sys.argv:
        max_iter = int(sys.argv[sys.argv.index("--max-iter") + 1])
    t0 = time.time()
    pr = problems.portfolio_problem(nb, bs, seed=3)
    t_gen = time.time() - t0
    args = (pr["n"], pr["m"], pr["P"], pr["A"], pr["q"], pr["b"], pr["cones"])
    out = dict(workload="portfolio problem %d x SOC(%d), n=%d, m=%d" % (nb, bs + 1, pr["n"], pr["m"]),
               generate_s=round(t_gen, 2))
    t0 = time.time()
    r = ipm_device.solve_device(hip, *args, max_iter=max_iter, fetch=True)
    out["total_s_incl_setup"] = round(time.time() - t0, 2)
    x = r["x"]
    out.update(status=r["status"], iterations=r["iterations"], loop_ms=round(1e3 * r["loop_s"], 2),
               ms_per_iteration=round(1e3 * r["loop_s"] / max(1, r["iterations"]), 3),
               iterations_per_s=round(r["iterations"] / r["loop_s"], 1), obj_val=r["obj_val"],
               res_primal=r["res_primal"], res_dual=r["res_dual"], gap_abs=r["gap_abs"],
               budget_error=abs(float(x.sum()) - nb), min_x=float(x.min()),
               kkt_dim=int(r["info"].n), nnzL=int(r["info"].nnzL))
    # independent feasibility check on the host (numpy only)
    import scipy.sparse as sp
    A = sp.csc_matrix((pr["A"][2], pr["A"][1], pr["A"][0]), shape=(pr["m"], pr["n"]))
    s = pr["b"] - A @ x
    n = pr["n"]
    soc = s[1 + n:].reshape(nb, bs + 1)
    out["host_check"] = dict(eq=abs(s[0]), nn_min=float(s[1:1 + n].min()),
                             soc_min_margin=float((soc[:, 0] - np.linalg.norm(soc[:, 1:], axis=1)).min()),
                             obj=float(pr["q"] @ x))
    if "--phases" in sys.argv:
        timing = {}
        r2 = ipm_device.solve_device(hip, *args, max_iter=max_iter, timing=timing, fetch=False)
        it = max(1, r2["iterations"])
        out["phase_ms_per_iteration"] = {k: round(1e3 * v / it, 3) for k, v in sorted(timing.items())}
        out["phase_run_ms_per_iteration"] = round(1e3 * r2["loop_s"] / it, 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
