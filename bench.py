#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec of KKT factor+solve work (BASELINE.json metric).

One "step" = the KKT work of one interior-point iteration (SURVEY.md 8d;
reference call sites core/solver.rs:351,361,398):
    1 x update   : cone scaling (update_scaling) + fused Hs / sparse-cone value
                   update + static regularisation + numeric LDL' refactor
    3 x solve    : LDL' solve + iterative refinement fixed at r = 1 extra round
                   (max_iter = 1, tolerances 0)  ->  6 LDL' solves + 6 symv
All inputs (s, z, right-hand sides) are resident in HBM when the timed region
starts; outputs stay in HBM.

N = 1 : BASELINE config 3, portfolio SOCP n = 10^6 (1000 x SOC(1001)).
N > 1 : weak scaling -- every rank owns one n = 10^6 shard of a block-diagonal
        problem (N independent portfolio blocks; the elimination forest has N
        roots, so factor/solve need no exchange); after each of the 3 solves the
        step direction is all-gathered over RCCL/xGMI so every rank holds the
        full (dx, dz).  value = N shards * steps / time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch  # imported BEFORE the extension so both share one HIP runtime
import torch.distributed as dist

import __graft_entry__ as graft
from tests import problems

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def baseline_metric():
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "IPM iterations/sec (KKT factor+solve) at n=10^6 SOCP, 1/2/4/8 GPUs"


def algorithmic_bytes(N, nnzK, nnzL, nnzHs, m):
    """SURVEY.md 8(d) byte model (i32 indices, fp64 values), per unit of work."""
    B_update = 12 * nnzHs + 24 * N + 16 * m
    B_factor = 12 * nnzK + 12 * nnzL + 17 * N
    B_solve = 24 * nnzL + 40 * N
    B_symv = 12 * nnzK + 24 * N
    return dict(update=B_update, factor=B_factor, solve=B_solve, symv=B_symv,
                iter=B_update + B_factor + 6 * B_solve + 6 * B_symv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nblocks", type=int, default=1000, help="SOC blocks per shard (default: config 3)")
    ap.add_argument("--blocksize", type=int, default=1000)
    ap.add_argument("--cpu-steps", type=int, default=-1, help="oracle steps for cpu_baseline (-1 auto, 0 off)")
    ap.add_argument("--profile-family", type=int, default=1,
                    help="kernel family timed with hipEvents for the roofline (1 = k_bundle_symv)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch N > 1 as: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    # one process per GPU; CHIP_BENCH_BACKEND=gloo + several ranks on one GPU is a plumbing test mode
    backend = os.environ.get("CHIP_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    hip = graft.load_package()

    # ---- this rank's shard ----------------------------------------------------
    pr = problems.portfolio_socp(args.nblocks, args.blocksize, seed=3 + rank)
    n, m = pr["n"], pr["m"]
    st = hip.Settings.default(iterative_refinement_max_iter=1, iterative_refinement_reltol=0.0,
                              iterative_refinement_abstol=0.0, device=local_rank)
    t0 = time.time()
    P = hip.CscMatrix(n, n, *pr["P"])
    A = hip.CscMatrix(m, n, *pr["A"])
    ks = hip.HipKKTSolver(P, A, pr["cones"], m, n, settings=st)
    t_setup = time.time() - t0
    info = ks.linear_solver_info()
    dev = torch.device("cuda", local_rank)
    rng = np.random.default_rng(1234 + rank)
    s_d = torch.tensor(pr["s"], device=dev)
    z_d = torch.tensor(pr["z"], device=dev)
    rhs = [(torch.tensor(rng.standard_normal(n), device=dev), torch.tensor(rng.standard_normal(m), device=dev))
           for _ in range(3)]
    gdev = dev if backend == "nccl" else torch.device("cpu")
    # one (lhs, gathered) pair per solve of a step: the all-gather of solve k is asynchronous and
    # overlaps the compute of the following solves; its buffers are only reused one step later
    lhs = [torch.zeros(n + m, device=dev, dtype=torch.float64) for _ in range(3)]
    gathered = [torch.zeros(world * (n + m), device=gdev, dtype=torch.float64) if world > 1 else None
                for _ in range(3)]
    works = [None, None, None]
    torch.cuda.synchronize()

    def step():
        ks.update_scaling_dev(s_d.data_ptr(), z_d.data_ptr())
        if not ks.update():
            raise RuntimeError("KKT update failed")
        for k, (rx, rz) in enumerate(rhs):
            if works[k] is not None:
                # the gather that used these buffers one step ago: wait() only orders torch's current
                # stream behind it, the engine launches on its own stream -> wait on the host as well
                works[k].wait()
                if backend == "nccl":
                    torch.cuda.current_stream().synchronize()
                works[k] = None
            ks.setrhs_dev(rx.data_ptr(), rz.data_ptr())
            if not ks.solve_dev(lhs[k].data_ptr(), lhs[k].data_ptr() + 8 * n):
                raise RuntimeError("KKT solve failed")
            if world > 1:
                # every rank ends up with the full step direction (dx, dz) of the block-diagonal
                # problem: RCCL all-gather over xGMI (7 x 24 MB received per rank at 8 GPUs),
                # launched once this rank's solve is complete and left running behind the next solve
                ks.synchronize()
                works[k] = dist.all_gather_into_tensor(gathered[k], lhs[k] if backend == "nccl" else lhs[k].cpu(),
                                                       async_op=True)

    def sync_all():
        ks.synchronize()
        for k in range(3):
            if works[k] is not None:
                works[k].wait()
                works[k] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    ks.profile(args.profile_family)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    prof = ks.profile_read()
    ks.profile(0)
    if world > 1:
        t = torch.tensor([elapsed], device=gdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ir = ks.linear_solver_info().last_ir_iterations

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.steps / elapsed
        Bm = algorithmic_bytes(ks.N, ks.nnzK, info.nnzL, ks.nHs, m)
        # family 1 = the dominant kernel: residual of all bundle rows, K stored once (U):
        # 12 B per streamed K entry + 24 B per row (x, b read; e written)
        fam_bytes = {1: 12 * ks.nnzU + 24 * ks.NF}.get(args.profile_family)
        fam_name = {1: "k_bundle_symv (residual e = b - Kx over the %d bundle rows, ||e||inf folded in)" % ks.NF,
                    2: "k_gather_merged<1> (BWD top levels)", 3: "k_gather_merged<0> (FWD top levels)",
                    4: "k_factor_T"}.get(args.profile_family, "?")
        # HBM bytes per launch from the PMC passes (tools/pmc_traffic.sh -> tools/pmc_summarize.py ->
        # profiles/*_pmc_traffic.json; FETCH_SIZE x2 + WRITE_SIZE, see that script).  PMC counters
        # cannot be collected from inside the timed run, so the committed summary of the same
        # command is quoted; null when the workload differs from the profiled one.
        traffic = None
        try:
            import glob
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
            if pj and args.nblocks == 1000 and args.blocksize == 1000 and args.profile_family == 1:
                traffic = json.load(open(pj[-1]))["kernels"]["k_bundle_symv"]["hbm_bytes"]
        except Exception:
            traffic = None
        roof = None
        if prof["launches"] > 0 and fam_bytes:
            avg_ms = prof["ms"] / prof["launches"]
            ach = fam_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": fam_name,
                    "launches": prof["launches"], "avg_launch_us": round(1e3 * avg_ms, 2),
                    "algorithmic_bytes_per_launch": fam_bytes,
                    "whole_step": {"algorithmic_bytes": Bm["iter"],
                                   "achieved_GBs": round(Bm["iter"] / (ms_per_step * 1e-3) / 1e9, 1),
                                   "frac": round(Bm["iter"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        cpu = None
        if world == 1 and args.cpu_steps != 0:
            cpu = cpu_baseline(pr, ks, args)
        out = {
            "metric": baseline_metric(),
            "value": round(value, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "portfolio SOCP (BASELINE config 3): n=%d, Zero(1)+NN(%d)+%d x SOC(%d) per shard, "
                                   "%d shard(s) block-diagonal" % (n, n, args.nblocks, args.blocksize + 1, world),
                       "kkt_dim": ks.N, "nnz_triu_K": ks.nnzK, "nnz_L": int(info.nnzL), "etree_levels": int(info.n_levels),
                       "per_step": "1 update(scaling+Hs+static reg+refactor) + 3 solves x (LDL solve + 1 IR round)",
                       "ir_rounds": int(ir), "setup_s": round(t_setup, 2),
                       "collective": "all_gather(step direction) x3/step over RCCL" if world > 1 else "none"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(pr, ks, args):
    """the CPU oracle (C restatement of the reference qdldl path, 1 thread -- the reference
    engine reports threads: 1, ldlsolvers/qdldl.rs:68) on the SAME workload and permutation,
    bounded to ~10-30 s of CPU work."""
    from oracle import oracle as orc
    ost = orc.Settings.default()
    ost.ir_max_iter = 1
    ost.ir_reltol = 0.0
    ost.ir_abstol = 0.0
    cones = orc.Cones(pr["cones"])
    ko = orc.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones, settings=ost, perm=ks.perm)
    rng = np.random.default_rng(99)
    rhs = [(rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])) for _ in range(3)]

    def step():
        cones.update_scaling(pr["s"], pr["z"])
        assert ko.update()
        for rx, rz in rhs:
            ko.setrhs(rx, rz)
            ok, _, _ = ko.solve()
            assert ok

    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter() - t0
    nsteps = args.cpu_steps if args.cpu_steps > 0 else max(2, min(200, int(15.0 / max(t1, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    el = time.perf_counter() - t0
    return {"value": round(nsteps / el, 4), "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "%d full steps of the same workload (n=%d) on 1 host core, oracle/ C restatement of "
                      "src/qdldl + DirectLDLKKTSolver, same permutation; host has %d cores"
                      % (nsteps, pr["n"], os.cpu_count() or 0)}


if __name__ == "__main__":
    main()
