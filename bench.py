#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec of KKT factor+solve work (BASELINE.json metric).

One "step" = the KKT work of one interior-point iteration (SURVEY.md 8d;
reference call sites core/solver.rs:351,361,398):
    1 x update   : cone scaling (update_scaling) + fused Hs / sparse-cone value
                   update + static regularisation + numeric LDL' refactor
    3 x solve    : LDL' solve + iterative refinement fixed at r = 1 extra round
                   (max_iter = 1, tolerances 0)  ->  6 LDL' solves + 6 symv
All inputs (s, z, right-hand sides) are resident in HBM when the timed region
starts; outputs stay in HBM.  No PyTorch: device buffers, streams and the RCCL
exchange all go through the C ABI (include/clarabel_hip.h).

--gpus 1 (default): BASELINE config 3, portfolio SOCP n = 10^6 (1000 x SOC(1001)) --
        the configuration the metric is quoted on -- on one MI355X; the JSON line also carries
        `parity` (solutions against the CPU oracle on the same inputs), `cpu_baseline`
        (the oracle, 1 core), `cpu_baseline_mt` (oracle/ldl_mt.c: the same column algorithm on OpenMP
        threads, labelled non-reference)
        and `batched_c4` (BASELINE config 4 whole on this GPU: the N = 1 point of the
        strong-scaling curve below).
--gpus N > 1 (one process per GPU; launched by torch.distributed.run -- only its env vars are
        used -- or, when WORLD_SIZE is not set, by bench.py itself as N subprocesses):
        BASELINE config 4, 1024 independent SOCPs of n = 2000, sharded by
        whole elimination trees over the ranks (clarabel.rs_amd/sharding.py: 1024/N blocks
        each) -- STRONG scaling: the total problem is fixed.  Factor / solves / refinement
        need no exchange; the iteration's step direction -- the solution of the LAST of the
        three solves, the combined direction the iterate moves along (the first two solves
        feed block-local updates and scalars only) -- is all-gathered with RCCL over xGMI
        (chip_kkt_allgather_step: ordered behind the solve by an event, running on its own
        stream behind the next step's update and first two solves; --gather-every-solve
        gathers all three solutions instead).  value = steps / time of the whole 1024-block
        problem.
        Every rank checks its shard against the CPU oracle on the same inputs and permutation
        and the ranks reduce the error (max) -> `parity` in the N > 1 line; the gathered step
        direction is compared bit for bit with the ranks' local solutions (checksums of every
        segment all-reduced).
--workload c3|c4 forces the workload (c4 at N = 1 = the whole batched problem).
--workload c2|c5|c5m: the other BASELINE configs (systems whose top runs as chain supernodes), with
        their own `roofline` (c2: HBM, the pipelined supernode substitution k_snode_tri; c5 / c5m: the f64
        matrix cores, k_snode_update, flops from the supernode geometry), `cpu_baseline` (the oracle, bounded;
        c5 at quarter size), `cpu_baseline_mt` (oracle/ldl_sn.c: a supernodal multifrontal LDL' on OpenMP threads --
        the kind of engine the reference itself would pick for these systems, labelled "port-supernodal, NOT faer")
        and `parity` (c2 / c5m: the oracle live; c5: tests/golden/c5_full_oracle.npz).  The default line carries
        compact `c2` / `c5` objects with the same fields.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

import __graft_entry__ as graft

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0  # what a plain copy reaches on this part (same guide): `frac_of_achievable` beside `frac`
MFMA_F64_PEAK_TFLOPS = 78.6  # dense fp64 matrix-core peak
TOL = 1e-8             # north_star: "solution within 1e-8 relative of reference"


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


def bench_settings(hip, device):
    # (BENCH_USE_GRAPH=1: the launch sequence of an LDL' solve replayed as a hipGraph -- an experiment switch of the tools)
    return hip.Settings.default(iterative_refinement_max_iter=1, iterative_refinement_reltol=0.0,
                                iterative_refinement_abstol=0.0, device=device,
                                use_graph=1 if os.environ.get("BENCH_USE_GRAPH") else 0)


class Workload:
    """one rank's KKT system, resident on its GPU, and the step that is timed"""

    def __init__(self, hip, pr, device, rank, nrhs=3):
        self.hip, self.pr = hip, pr
        n, m = pr["n"], pr["m"]
        t0 = time.time()
        self.ks = hip.HipKKTSolver(hip.CscMatrix(n, n, *pr["P"]), hip.CscMatrix(m, n, *pr["A"]), pr["cones"], m, n,
                                   settings=bench_settings(hip, device))
        self.t_setup = time.time() - t0
        rng = np.random.default_rng(1234 + rank)
        self.rhs_host = [(rng.standard_normal(n), rng.standard_normal(m)) for _ in range(nrhs)]
        self.s_d, self.z_d = hip.DeviceArray(pr["s"]), hip.DeviceArray(pr["z"])
        self.rhs = [(hip.DeviceArray(rx), hip.DeviceArray(rz)) for rx, rz in self.rhs_host]
        self.lhs = [hip.DeviceArray(n + m) for _ in range(nrhs)]
        self.n, self.m = n, m
        self.gather_every_solve = False
        self.repeats = 0   # solves repeated at collect time after a fused launch timed out (chip_kkt_collect: 2)
        self.coresident = None  # (blocks, usec, device): see --coresident
        # the first two of an iteration's three solves do not depend on each other (the constant right-hand side of
        # kktsystem.rs:108-125 and the affine direction of core/solver.rs:351-361): they go to the device as ONE call,
        # chip_kkt_solve2_dev_enqueue; the third (the combined direction) depends on the affine result.  False: three calls.
        self.pair = nrhs == 3

    def step(self, comm=None, gathered=None, counts=None):
        # one interior-point iteration's KKT work is ENQUEUED as a whole; the reference's bools (update:
        # pivots finite and cones interior; solve: refinement residuals finite) are produced on the device
        # and collected with one synchronisation at the end of the step
        ks = self.ks
        # (cones.update_scaling + kktsystem.update's KKT part, core/solver.rs:334-352, as one enqueue)
        ks.update_scaled_enqueue(self.s_d.ptr, self.z_d.ptr)
        last = len(self.rhs) - 1
        first = 0
        if self.pair and not (comm is not None and self.gather_every_solve):
            (ra, za), (rb, zb) = self.rhs[0], self.rhs[1]
            ks.solve2_dev_enqueue(ra.ptr, za.ptr, self.lhs[0].ptr, self.lhs[0].ptr + 8 * self.n,
                                  rb.ptr, zb.ptr, self.lhs[1].ptr, self.lhs[1].ptr + 8 * self.n)
            first = 2
        for k, (rx, rz) in enumerate(self.rhs):
            if k < first:
                continue
            exchange = comm is not None and (self.gather_every_solve or k == last)
            if exchange:
                # lhs[k] / gathered[k] were handed to the all-gather one step ago: this stream waits
                # for it ON THE DEVICE (no host synchronisation) before overwriting them
                comm.wait(ks)
            ks.setrhs_dev(rx.ptr, rz.ptr)
            ks.solve_dev_enqueue(self.lhs[k].ptr, self.lhs[k].ptr + 8 * self.n)
            if exchange:
                # every rank ends up with the full step direction (dx, dz): RCCL all-gather over xGMI,
                # enqueued behind this solve by an event and left running behind the next solves
                comm.allgather_step(ks, self.lhs[k].ptr, gathered[k].ptr, counts)
            if k == last and self.coresident:
                if exchange and hasattr(comm, "debug_spin"):
                    # on the communicator's stream, behind the all-gather: the time RCCL's kernel would hold CUs at N = 8
                    comm.debug_spin(self.coresident[0], 256, self.coresident[1])
                else:  # (no communicator: a foreign kernel on a stream of its own, with no ordering at all)
                    self.hip.debug_spin(self.coresident[2], self.coresident[0], 256, 0, self.coresident[1])
        uok, sok = ks.collect()
        if not uok or len(sok) != len(self.rhs) or not all(sok):
            raise RuntimeError("KKT step failed: update %s, solves %s" % (uok, sok))
        if ks.repeated_solves:
            # a fused launch timed out (its workgroups were not all resident) and collect repeated the solve: what was
            # enqueued behind it consumed garbage.  The exchange is re-issued for the affected solves; the step is no
            # longer a clean measurement and the line says so.
            self.repeats += len(ks.repeated_solves)
            if comm is not None:
                for k in ks.repeated_solves:
                    if self.gather_every_solve or k == last:
                        comm.wait(ks)
                        comm.allgather_step(ks, self.lhs[k].ptr, gathered[k].ptr, counts)

    def run(self, steps, warmup, profile_family=0, comm=None, gathered=None, counts=None, events_in_timed_region=True,
            sequential_pass=True):
        """W warm-up steps, then K timed steps bracketed by synchronisations.  The dominant kernel's launch durations come from
        hipEvent pairs on the engine's stream around each of its launches (chip_kkt_profile): inside the timed region when the
        events_in_timed_region, in a SECOND pass of the same steps right after it otherwise (the default of every workload
        since the end of round 5): an event pair costs ~6 us on the stream -- 2 ms of an 18 ms step of config 2 (324 pairs), and
        33 us of config 3's 0.78 ms step (3 pairs: 1282 it/s with them in the timed region, 1337 without, same box) -- which would
        be charged to `value`.  The pass with the pairs is on the line too (`roofline_events_pass`)."""
        def sync():
            self.ks.synchronize()
            if comm is not None:
                comm.synchronize()
                comm.barrier()
            self.hip.device_synchronize()
        for _ in range(warmup):
            self.step(comm, gathered, counts)
        sync()
        self.ks.profile(profile_family if events_in_timed_region else 0)
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(steps):
            self.step(comm, gathered, counts)   # (ends with collect(): one synchronisation per step)
            marks.append(time.perf_counter())
        sync()
        elapsed = time.perf_counter() - t0
        d = np.diff(np.asarray(marks)) * 1e3
        self.step_ms = {"min": round(float(d.min()), 4), "median": round(float(np.median(d)), 4),
                        "max": round(float(d.max()), 4)} if len(d) else None
        self.events_pass = None
        self.sequential = None
        if not events_in_timed_region and self.pair and sequential_pass:
            # (level-scheduled systems: the paired call overlaps two chains of launches on two streams; the same steps with three
            # separate solve calls, for the line's other policy)
            self.pair = False
            for _ in range(min(2, warmup)):
                self.step(comm, gathered, counts)
            sync()
            t1 = time.perf_counter()
            for _ in range(steps):
                self.step(comm, gathered, counts)
            sync()
            self.sequential = {"policy": "1 update + 3 separate solve calls", "steps": steps,
                               "ms_per_step": round(1e3 * (time.perf_counter() - t1) / steps, 4)}
            self.pair = True
        if not events_in_timed_region and profile_family:
            self.ks.profile(profile_family)
            t1 = time.perf_counter()
            for _ in range(steps):
                self.step(comm, gathered, counts)
            sync()
            self.events_pass = {"steps": steps, "ms_per_step_with_events": round(1e3 * (time.perf_counter() - t1) / steps, 4)}
        prof = self.ks.profile_read()
        self.ks.profile(0)
        self.own_elapsed = elapsed  # (this rank's clock; the line's value uses the maximum over the ranks)
        if comm is not None:
            elapsed = float(comm.allreduce([elapsed], "max")[0])
        return elapsed, prof

    def solutions(self):
        self.ks.synchronize()
        return [a.numpy() for a in self.lhs]


def profile_commit(name):
    """the commit that put profiles/<name> into the tree (so that a quoted PMC figure can be matched to the code it was taken
    on); None outside a git checkout (the GPU box gets a snapshot without .git)"""
    if not name:
        return None
    try:
        import subprocess
        out = subprocess.run(["git", "-C", ROOT, "log", "-n", "1", "--format=%h %cs", "--", os.path.join("profiles", name)],
                             capture_output=True, text=True, timeout=10)
        return out.stdout.strip() or None
    except Exception:
        return None


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


def sweep_roofline(ks, wm, prof, steps, nsolves):
    """HBM roofline object of the sweeps through the chain supernodes (profile family 11): supernodes on the one-pass
    matrices stream G = [I; L_B] T^-1 once per sweep and unit level (8 B per entry: dense, no index; k_snode_gfwd /
    k_snode_gbwd), the others their trapezoid of L (k_snode_tri; 12 B per entry of SURVEY 8(d)'s B_solve, the model is
    kept).  achieved = algorithmic bytes of all launches of the family / their total duration."""
    sm = ks.sweep_model()
    sweeps = 2 * nsolves
    if sm["g_levels"] > 0 and sm["g_levels"] == sm["sn_levels"]:
        per_sweep = 8.0 * sm["g_doubles"]
        kernel = ("k_snode_gsweep / k_snode_gfwd / k_snode_gbwd (one pass over G = [I; L_B] T^-1 of every supernode, forward and "
                  "backward sweeps; %d unit levels -- a run of consecutive levels is one persistent launch, a level with more "
                  "blocks than two rounds of its grid a launch of its own; G built once per refactor by k_snode_ginv)" % sm["g_levels"])
        short = "k_snode_gsweep"
    else:
        per_sweep = 12.0 * wm["sn_panel_entries"]
        kernel = ("k_snode_tri (pipelined substitution through the wide chain supernodes of one unit level, forward and "
                  "backward sweeps)%s" % ("; %d of %d levels on the one-pass matrices" % (sm["g_levels"], sm["sn_levels"]) if sm["g_levels"] else ""))
        short = "k_snode_tri"
    tot_bytes = per_sweep * sweeps * steps
    avg_ms = prof["ms"] / prof["launches"]
    ach = tot_bytes / prof["launches"] / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "kernel_short": short, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_us": round(1e3 * avg_ms, 2),
            "algorithmic_bytes_per_launch": round(tot_bytes / prof["launches"])}


def oracle_leg(w, args, time_it=True):
    """The CPU oracle (C restatement of the reference qdldl path + DirectLDLKKTSolver, 1 thread -- the
    reference engine reports threads: 1, ldlsolvers/qdldl.rs:68) on the SAME workload, permutation and
    right-hand sides: (a) parity of the GPU solutions, under the bench's fixed r = 1 refinement and under
    the default refinement settings; (b) cpu_baseline, bounded to ~10-30 s of CPU work."""
    os.environ["ORACLE_NATIVE"] = "1"  # -march=native build made on this box, see oracle/Makefile
    from oracle import oracle as orc
    pr, ks, hip = w.pr, w.ks, w.hip
    ost = orc.Settings.default()
    ost.ir_max_iter, ost.ir_reltol, ost.ir_abstol = 1, 0.0, 0.0
    cones = orc.Cones(pr["cones"])
    ko = orc.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones, settings=ost, perm=ks.perm)

    def step(keep=None):
        cones.update_scaling(pr["s"], pr["z"])
        # (PSD cones: the oracle takes the numpy restatement's Hs blocks for the same (S, Z), oracle/psd_numpy.py)
        assert ko.update(pr.get("hsblocks"))
        for rx, rz in w.rhs_host:
            ko.setrhs(rx, rz)
            ok, x, z = ko.solve()
            assert ok
            if keep is not None:
                keep.append(np.concatenate([x, z]))

    ref_r1 = []
    t0 = time.perf_counter()
    step(ref_r1)
    t1 = time.perf_counter() - t0
    oracle_leg.last_step_s = t1  # (the compact lines of configs 2 / 5 quote this one step as their cpu_baseline)
    got_r1 = w.solutions()
    err_r1 = max(relerr(g, r) for g, r in zip(got_r1, ref_r1))
    # default refinement (reltol 1e-13, abstol 1e-12, <= 10 rounds) on both sides, same factorisation
    dflt = hip.Settings.default(device=ks.settings.device)
    ks.set_settings(dflt)
    ko.settings = orc.Settings.default()
    err_d, rounds = 0.0, []
    for k, (rx, rz) in enumerate(w.rhs):
        ks.setrhs_dev(rx.ptr, rz.ptr)
        assert ks.solve_dev(w.lhs[k].ptr, w.lhs[k].ptr + 8 * w.n)
        rounds.append(int(ks.linear_solver_info().last_ir_iterations))
        ko.setrhs(*w.rhs_host[k])
        ok, x, z = ko.solve()
        assert ok
        ks.synchronize()
        err_d = max(err_d, relerr(w.lhs[k].numpy(), np.concatenate([x, z])))
    ks.set_settings(bench_settings(hip, ks.settings.device))
    ko.settings = ost
    parity = {"rel_err_vs_oracle": err_d, "tol": TOL, "ok": bool(err_d <= TOL),
              "what": "max over the %d solves of ||x_gpu - x_oracle||inf / max(1, ||x_oracle||inf), post-refinement, "
                      "default refinement settings (rounds taken on the GPU: %s), same inputs and permutation"
                      % (len(w.rhs), rounds),
              "rel_err_fixed_r1": err_r1,
              "fixed_r1": "the same for the solutions of the timed configuration (exactly one refinement round)"}
    cpu = None
    if time_it:
        nsteps = args.cpu_steps if args.cpu_steps > 0 else max(1 if t1 > 20.0 else 2, min(200, int(15.0 / max(t1, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        el = time.perf_counter() - t0
        cpu = {"value": round(nsteps / el, 4), "unit": "iterations/s", "cores": 1, "kind": "port",
               "sample": "%d full steps of the same workload (N=%d) on 1 of the host's %d cores, oracle/ C restatement "
                         "of src/qdldl + DirectLDLKKTSolver (%s), same permutation and right-hand sides"
                         % (nsteps, ks.N, os.cpu_count() or 0,
                            "-O3 -march=native" if orc.is_native() else "-O3 -march=x86-64-v3")}
    return parity, cpu, ko


def mt_leg(w, ko):
    """labelled NON-reference comparator: what the host's cores do on the SAME algorithm -- oracle/ldl_mt.c, the
    left-looking column LDL' of qdldl.rs:469-669 with the columns of an elimination-tree level in parallel on OpenMP
    threads, level-scheduled solves, a threaded residual; same pattern, permutation, regularised values and pivot
    rule as the oracle, whose factors and solution it must reproduce.  The reference's own multi-threaded engine
    (faer, ldlsolvers/faer_ldl.rs) needs a Rust toolchain; it is the reference's choice only for systems with dense
    fronts (auto.rs:60-88: configs 2 / 5), for which this supernode-free code says nothing."""
    try:
        from oracle import oracle as orc
        from oracle import ldl_mt
        ncpu = os.cpu_count() or 1
        best = None
        N = ko.N
        rng = np.random.default_rng(5)
        bs = [rng.standard_normal(N) for _ in range(3)]
        # reference solution of the first right-hand side: the oracle's own factorisation (already done by the caller)
        ok_ref, x_ref = ko.solve_full(bs[0])
        Kp, Ki, Kx = np.asarray(ko.kkt.colptr), np.asarray(ko.kkt.rowval), np.asarray(ko.kkt.nzval)
        t_leg = time.perf_counter()
        prev = None
        for threads in [t for t in (1, 4, 16, 32) if t <= ncpu]:
            # (bounded: more threads are tried only while they pay and the leg stays within ~20 s -- on the GPU box's
            # 256-core host the 64- and 256-thread passes once took minutes and lost to 16 threads)
            if prev is not None and (time.perf_counter() - t_leg > 20.0 or best is not prev):
                break
            mt = ldl_mt.LdlMT(orc, ko, w.ks.perm, threads)
            Ax = mt.values()
            perm = mt.perm
            # the UNregularised permuted values for the residual (directldlkktsolver.rs:255-261): the engine's values
            # minus the static shift +-eps on the diagonal
            cols = np.repeat(np.arange(N), np.diff(mt.Ap))
            dpos = np.nonzero(mt.Ai == cols)[0]
            Ar = Ax.copy()
            Ar[dpos] -= ko.regularizer * mt.signs[cols[dpos]]
            deps, ddelta = ko.settings.dynamic_reg_eps, ko.settings.dynamic_reg_delta
            mt.factor(Ax, deps, ddelta)  # warm-up (thread pool, page faults)
            t0 = time.perf_counter()
            okf, _ = mt.factor(Ax, deps, ddelta)
            xs = []
            for b in bs:
                bp = np.ascontiguousarray(b[perm])
                x = bp.copy()
                mt.solve(x)
                e = np.empty(N)
                mt.residual(Ar, x, bp, e)   # (one refinement round, as in the timed GPU step)
                mt.solve(e)
                x += e
                mt.residual(Ar, x, bp, e)
                xs.append(x)
            el = time.perf_counter() - t0
            x0 = np.empty(N)
            x0[perm] = xs[0]
            err = float(np.max(np.abs(x0 - x_ref)) / max(1.0, np.max(np.abs(x_ref)))) if ok_ref else None
            cand = {"value": round(1.0 / el, 4), "threads": threads, "rel_err_vs_oracle_solution": err, "factor_ok": bool(okf)}
            if best is None or cand["value"] > best["value"]:
                best = cand
            prev = cand
            del mt
        return {"value": best["value"], "unit": "iterations/s", "cores": best["threads"],
                "kind": "port-mt (oracle/ldl_mt.c: the qdldl column algorithm with the columns of an elimination-tree level "
                        "on OpenMP threads; NOT the reference and NOT its faer engine, which needs a Rust toolchain)",
                "sample": "1 step (factor + 3 x (solve, residual, solve, residual)) at N=%d for each of several thread counts "
                          "up to the host's %d cores, best kept; its solution against the oracle's: %s"
                          % (N, ncpu, "%.1e" % best["rel_err_vs_oracle_solution"] if best["rel_err_vs_oracle_solution"] is not None else "n/a")}
    except Exception as ex:  # a comparator, never a reason to lose the bench line
        return {"value": None, "error": repr(ex)[:300]}


def sn_leg(w, hip, budget_s=60.0):
    """labelled NON-reference comparator for the systems with dense fronts (configs 2 / 5), where the reference itself
    would take faer's supernodal LDL' instead of QDLDL (ldlsolvers/auto.rs:60-88): oracle/ldl_sn.c, a multifrontal
    supernodal LDL' with relaxed amalgamation on OpenMP threads -- "port-supernodal, NOT faer" (faer needs a Rust
    toolchain).  Same K (the values the device holds + the static shift), same signs, same pivot rule, the product's
    permutation re-postordered; one step = factor + 3 x (solve, residual, solve, residual), best of several thread
    counts; its solution against the device's for the same right-hand side."""
    try:
        from oracle import ldl_sn
        ks = w.ks
        ncpu = os.cpu_count() or 1
        N = ks.N
        K = ks.kkt_matrix()
        Kp, Ki = np.asarray(K.colptr).astype(np.int64), np.asarray(K.rowval).astype(np.int64)
        Kx = np.ascontiguousarray(ks.values())
        signs = np.asarray(ks.maps()["dsigns"]).astype(np.int8)
        info = ks.linear_solver_info()
        cols = np.repeat(np.arange(N), np.diff(Kp))
        dpos = np.nonzero(Ki == cols)[0]
        Kreg = Kx.copy()
        Kreg[dpos] += info.last_regularizer * signs[cols[dpos]]
        del cols
        st = ks.settings
        rng = np.random.default_rng(5)
        nm = ks.n + ks.m  # (the sparse second-order cones' expansion variables follow: right-hand side 0, kktsolver setrhs)
        bs = [np.concatenate([rng.standard_normal(nm), np.zeros(N - nm)]) for _ in range(3)]
        # the device's solution of the first right-hand side (default refinement)
        ks.setrhs(bs[0][:ks.n], bs[0][ks.n:nm])
        xg, zg = np.zeros(ks.n), np.zeros(ks.m)
        ok_dev = ks.solve(xg, zg)
        x_dev = np.concatenate([xg, zg])
        best, prev, sn = None, None, None
        t_leg = time.perf_counter()
        t_an = None
        for threads in [t for t in (1, 16, 64) if t <= ncpu] or [1]:
            if prev is not None and (time.perf_counter() - t_leg > budget_s or best is not prev):
                break  # (bounded: more threads only while they pay)
            t0 = time.perf_counter()
            del sn
            sn = ldl_sn.LdlSN(N, Kp, Ki, np.asarray(ks.perm), threads=threads)
            if t_an is None:
                t_an = time.perf_counter() - t0
            sn.factor(Kreg, signs, st.dynamic_regularization_eps, st.dynamic_regularization_delta)  # warm-up (pages, thread pool)
            t0 = time.perf_counter()
            okf, nreg = sn.factor(Kreg, signs, st.dynamic_regularization_eps, st.dynamic_regularization_delta)
            xs = []
            e = np.empty(N)
            for b in bs:
                x = b.copy()
                sn.solve(x)
                sn.residual(Kx, x, b, e)   # (one refinement round, as in the timed GPU step)
                sn.solve(e)
                x += e
                sn.residual(Kx, x, b, e)
                xs.append(x)
            el = time.perf_counter() - t0
            err = float(np.max(np.abs(xs[0][:nm] - x_dev)) / max(1.0, np.max(np.abs(x_dev)))) if ok_dev else None
            cand = {"value": round(1.0 / el, 4), "threads": threads, "err": err, "ok": bool(okf), "nreg": nreg}
            if best is None or cand["value"] > best["value"]:
                best = cand
            prev = cand
        out = {"value": best["value"], "unit": "iterations/s", "cores": best["threads"],
               "kind": "port-supernodal, NOT faer (oracle/ldl_sn.c: multifrontal LDL' with relaxed amalgamation, OpenMP over the "
                       "assembly tree and inside the dense updates; the reference would run faer here, which needs a Rust toolchain)",
               "sample": "1 step (factor + 3 x (solve, residual, solve, residual)) at N=%d for thread counts of {1, 16, 64} up to "
                         "the host's %d cores, best kept; %d supernodes, %.3g flops per factorisation, analysis %.1f s (not timed); "
                         "its solution against the device's: %s"
                         % (N, ncpu, sn.nsn, sn.flops, t_an or 0.0, "%.1e" % best["err"] if best["err"] is not None else "n/a"),
               "regularize_count": [int(best["nreg"]), int(info.regularize_count)]}
        del sn
        return out
    except Exception as ex:  # a comparator, never a reason to lose the bench line
        import traceback
        return {"value": None, "error": repr(ex)[:200], "where": traceback.format_exc().strip().splitlines()[-3:]}


def fixture_parity_c5(w, hip):
    """BASELINE config 5 at FULL size: the scalar oracle needs ~10 CPU-minutes for this factorisation, so its
    answers are a committed fixture (tests/golden/c5_full_oracle.npz, made by tests/golden/make_c5_fixture.py:
    same generator and seed, default refinement settings, seeded right-hand sides)."""
    path = os.path.join(ROOT, "tests", "golden", "c5_full_oracle.npz")
    if not os.path.exists(path):
        return None
    fx = np.load(path)
    ks = w.ks
    if int(fx["N"]) != ks.N or int(fx["n"]) != w.n or int(fx["m"]) != w.m:
        return None
    ks.set_settings(hip.Settings.default(device=ks.settings.device))
    rng = np.random.default_rng(int(fx["rhs_seed"]))
    err, rounds = 0.0, []
    lhs = hip.DeviceArray(w.n + w.m)
    for k in range(int(fx["nrhs"])):
        rx, rz = hip.DeviceArray(rng.standard_normal(w.n)), hip.DeviceArray(rng.standard_normal(w.m))
        ks.setrhs_dev(rx.ptr, rz.ptr)
        assert ks.solve_dev(lhs.ptr, lhs.ptr + 8 * w.n)
        ks.synchronize()
        rounds.append(int(ks.linear_solver_info().last_ir_iterations))
        err = max(err, relerr(lhs.numpy(), fx["solutions"][k]))
    kv = ks.values()
    kerr = float(np.max(np.abs(kv[fx["k_sample_idx"]] - fx["k_sample_val"])) / max(1.0, float(fx["k_absmax"])))
    info = ks.linear_solver_info()
    ks.set_settings(bench_settings(hip, ks.settings.device))
    return {"rel_err_vs_oracle": err, "tol": TOL, "ok": bool(err <= TOL),
            "what": "max over %d solves of ||x_gpu - x_oracle||inf / max(1, ||x_oracle||inf), post-refinement, default "
                    "refinement settings (rounds on the GPU: %s, in the oracle: %s); oracle answers from the committed "
                    "fixture tests/golden/c5_full_oracle.npz" % (int(fx["nrhs"]), rounds, fx["ir_rounds"].tolist()),
            "kkt_values_rel_err_sample": kerr,
            "regularize_count": [int(info.regularize_count), int(fx["regularize_count"])],
            "positive_inertia": [int(info.positive_inertia), int(fx["positive_inertia"])]}


def quarter_c5_cpu_baseline(hip, problems, args):
    """cpu_baseline of the config-5 line: ONE step of the oracle on the quarter-size instance (50 cliques; the
    full-size factorisation takes the scalar oracle ~10 minutes), permutation from the product's host analysis"""
    os.environ["ORACLE_NATIVE"] = "1"
    from oracle import oracle as orc
    pr = problems.chordal_sdp(50, 50, 10, 50, 51, seed=5, with_hs=True)
    st = hip.Settings.default(device=hip.DEVICE_HOST_ONLY)
    hk = hip.HipKKTSolver(hip.CscMatrix(pr["n"], pr["n"], *pr["P"]), hip.CscMatrix(pr["m"], pr["n"], *pr["A"]),
                          pr["cones"], pr["m"], pr["n"], settings=st)
    ost = orc.Settings.default()
    ost.ir_max_iter, ost.ir_reltol, ost.ir_abstol = 1, 0.0, 0.0
    cones = orc.Cones(pr["cones"])
    ko = orc.KKTSolver(pr["n"], pr["m"], pr["P"], pr["A"], cones, settings=ost, perm=hk.perm)
    rng = np.random.default_rng(1234)
    rhs = [(rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])) for _ in range(3)]
    t0 = time.perf_counter()
    cones.update_scaling(pr["s"], pr["z"])
    assert ko.update(pr["hsblocks"])
    for rx, rz in rhs:
        ko.setrhs(rx, rz)
        assert ko.solve()[0]
    el = time.perf_counter() - t0
    return {"value": round(1.0 / el, 5), "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "1 full step of the QUARTER-size instance (50 x PSD(50) + 50 x SOC(51), N=%d; the full size takes "
                      "the scalar oracle ~10 minutes per factorisation) on 1 of the host's %d cores, oracle/ C restatement "
                      "of src/qdldl + DirectLDLKKTSolver, Hs blocks from oracle/psd_numpy" % (ko.N, os.cpu_count() or 0)}


def extra_workload(hip, problems, which, args, device):
    """compact object of the default N = 1 line for BASELINE configs 2 / 5: the same Workload.step (1 update + 3 solves,
    one refinement round each), device resident, with its roofline figure, parity and host setup time -- so that the
    driver's own run sees every config, not only config 3 (bench.py --workload c2 / c5 print the full lines)"""
    try:
        if which == "c2":
            pr = problems.random_qp(100000, 200000, band=50, seed=1)
            desc, fam = "random sparse QP (BASELINE config 2): n=100000, m=200000, Nonnegative cone", 11
        else:
            pr = problems.chordal_sdp(200, 50, 10, 200, 51, seed=5, with_hs=False)
            desc, fam = "chordal SDP (BASELINE config 5): 200 x PSD(50) cliques with overlap 10 + 200 x SOC(51)", 7
        w = Workload(hip, pr, device, 0)
        steps, warm = max(3, min(args.steps, 10)), max(1, min(args.warmup, 2))
        el, prof = w.run(steps, warm, fam, events_in_timed_region=False)
        ms = 1e3 * el / steps
        step_ms0, events_pass0, sequential0 = w.step_ms, w.events_pass, w.sequential  # (of THIS run: later passes overwrite them)
        ks = w.ks
        info = ks.linear_solver_info()
        wm = ks.work_model()
        Bm = algorithmic_bytes(ks.N, ks.nnzK, info.nnzL, ks.nHs, w.m)
        roof = None
        if prof["launches"] > 0 and which == "c2" and wm["sn_panel_entries"] > 0:
            roof = sweep_roofline(ks, wm, prof, steps, len(w.rhs) * 2)
            roof.update({"launches_per_step": round(prof["launches"] / steps, 1), "kernel_ms_per_step": round(prof["ms"] / steps, 3)})
        elif prof["launches"] > 0 and which == "c5" and wm["sn_update_flops"] > 0:
            per_launch = wm["sn_update_flops"] * steps / prof["launches"]
            avg_ms = prof["ms"] / prof["launches"]
            ach = per_launch / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "k_snode_update", "achieved": round(ach, 2), "peak": MFMA_F64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / MFMA_F64_PEAK_TFLOPS, 4),
                    "launches_per_step": round(prof["launches"] / steps, 1), "kernel_ms_per_step": round(prof["ms"] / steps, 3)}
        whole = round(Bm["iter"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        chain = None
        if which == "c2" and wm["sn_update_flops"] > 0:
            # the factorisation's chain of block columns -- the LARGEST share of this step (the sweeps above are the second):
            # every block column of every unit level is a k_snode_update launch (left-looking update of the 64 columns, f64
            # matrix cores) followed by a k_snode_panel2 launch (the block's LDL' + the rows below it); two short passes
            # with event pairs around those launches.  flops from the supernode geometry (chip_kkt_work_model).
            fams = {}
            for f_, nm in ((7, "k_snode_update"), (8, "k_snode_panel2")):
                _, pf = w.run(3, 1, f_, events_in_timed_region=True, sequential_pass=False)
                if pf["launches"] > 0:
                    fams[nm] = {"launches_per_step": round(pf["launches"] / 3.0, 1), "avg_launch_us": round(1e3 * pf["ms"] / pf["launches"], 2),
                                "ms_per_step": round(pf["ms"] / 3.0, 3)}
            if fams:
                tot_ms = sum(v_["ms_per_step"] for v_ in fams.values())
                flops = wm["sn_update_flops"] + wm["sn_diag_rows_flops"]
                chain = {"bound": "mfma", "kernel": "k_snode_update + k_snode_panel2 (one pair per block column of a unit level, each waiting for the one before: a chain of dependent launches)",
                         "families": fams, "ms_per_step": round(tot_ms, 3), "flops_per_refactor": flops,
                         "achieved": round(flops / (tot_ms * 1e-3) / 1e12, 3), "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / (tot_ms * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, 4),
                         "what": "latency of %d dependent launches, not arithmetic: %.0f MFLOP per launch"
                                 % (int(sum(v_["launches_per_step"] for v_ in fams.values())),
                                    flops / max(1.0, sum(v_["launches_per_step"] for v_ in fams.values())) / 1e6)}
        cpu = None
        if which == "c2":
            parity, _, ko = oracle_leg(w, args, time_it=False)
            del ko
            t1 = getattr(oracle_leg, "last_step_s", None)
            if t1:
                cpu = {"value": round(1.0 / t1, 4), "unit": "iterations/s", "cores": 1, "kind": "port",
                       "sample": "1 full step of the same workload on 1 of the host's %d cores (the oracle's parity step, timed)"
                                 % (os.cpu_count() or 0)}
        else:
            parity = fixture_parity_c5(w, hip)
        out = {"workload": desc, "value": round(steps / el, 3), "unit": "iterations/s", "ms_per_step": round(ms, 4),
               "steps": steps, "step_ms": step_ms0, "roofline_events_pass": events_pass0,
               "per_step": "1 update + (2 paired + 1) solves, one refinement round each", "other_solve_policy": sequential0, "kkt_dim": ks.N, "nnz_L": int(info.nnzL), "setup_s": round(w.t_setup, 2),
               "roofline": roof, "roofline_factor_chain": chain, "whole_step_frac_of_hbm_peak": whole,
               "parity": None if parity is None else {k: parity[k] for k in ("rel_err_vs_oracle", "tol", "ok") if k in parity},
               "cpu_baseline": cpu, "cpu_baseline_mt": None if args.cpu_steps == 0 else sn_leg(w, hip)}
        del w
        return out
    except Exception as ex:  # an extra, never a reason to lose the bench line
        return {"error": repr(ex)[:300]}


def l1_dropin_leg(hip, w, args):
    """BASELINE config 3 through the STRICT drop-in boundary (L1, trait DirectLDLSolver): host buffers in, host buffers
    out, exactly the calls directldlkktsolver.rs:143,174,253,297 make per interior-point iteration -- the values of K
    handed over (the update_values stream of :143 / :245 as one chip_ldl_set_values: the engine reads them at refactor,
    the Pardiso pattern of SURVEY a26), refactor(), then solve(x, b) for each of the 3 solves and for each refinement
    round's correction (3 x (1 + 1) = 6 with the bench's fixed r = 1; the residuals in between are the CALLER's host work
    in the reference and are not timed).  Every byte crosses PCIe: that is what this row measures."""
    try:
        ks = w.ks
        K = ks.kkt_matrix()
        vals = np.ascontiguousarray(ks.values())
        maps = ks.maps()
        dsigns = maps["dsigns"]
        eps = float(ks.linear_solver_info().last_regularizer)
        reg = vals.copy()   # the caller applies the static regulariser before refactor (:217-250)
        dfull = np.asarray(maps["diag_full"], dtype=np.int64)
        reg[dfull] += eps * np.asarray(dsigns, dtype=np.float64)
        N = ks.N
        t0 = time.time()
        f = hip.HipDirectLDLSolver(hip.CscMatrix(N, N, K.colptr, K.rowval, reg), dsigns, hip.Settings.default(device=ks.settings.device),
                                   perm=ks.perm)
        t_setup = time.time() - t0
        rng = np.random.default_rng(77)
        bs = [rng.standard_normal(N) for _ in range(3)]
        xs = [np.zeros(N) for _ in range(6)]
        nsolve = 6

        def step():
            f.set_values(reg)
            if not f.refactor():
                raise RuntimeError("L1 refactor failed")
            for k in range(nsolve):
                f.solve(None, xs[k], bs[k % 3])

        steps = max(3, min(args.steps, 10))
        step()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        el = time.perf_counter() - t0
        # check: backward error of the raw LDL' solve against the same regularised K (scipy SpMV of the symmetric matrix),
        # and the solution against the oracle's factorisation with the same permutation and pivot rule
        import scipy.sparse as sp
        Ku = sp.csc_matrix((reg, np.asarray(K.rowval, dtype=np.int64), np.asarray(K.colptr, dtype=np.int64)), shape=(N, N))
        Kf = Ku + sp.triu(Ku, 1).T
        res = bs[0] - Kf @ xs[0]
        berr = float(np.max(np.abs(res)) / (float(np.max(np.abs(Kf).sum(axis=1))) * float(np.max(np.abs(xs[0]))) + float(np.max(np.abs(bs[0])))))
        from oracle import oracle as orc
        st0 = hip.Settings.default()
        fo = orc.QDLDL(N, K.colptr, K.rowval, reg, perm=ks.perm, Dsigns=dsigns, regularize_eps=st0.dynamic_regularization_eps,
                       regularize_delta=st0.dynamic_regularization_delta)
        xo = fo.solve(bs[0])
        err = relerr(xs[0], xo)
        bytes_step = 8 * len(reg) + nsolve * 2 * 8 * N
        del fo
        # ---- the same iteration through the FAST path of the boundary (chip_ldl_register_index / *_values_id /
        # chip_ldl_solve_refined / chip_ldl_pin_buffer): the entries that change between two scalings as one registered set,
        # diag_full + Dsigns as another; per step update_values on the set, regulariser on, refactor, regulariser off
        # (directldlkktsolver.rs:143, 245, 255-261) and 3 solves with ONE refinement round each on the device (:266-321)
        fast = None
        try:
            ks.update_scaling(w.pr["s"] * 1.25, w.pr["z"] * 0.8)
            ks.update()
            vals2 = np.ascontiguousarray(ks.values())
            ks.update_scaling(w.pr["s"], w.pr["z"])
            ks.update()
            changed = np.nonzero(vals != vals2)[0].astype(np.int64)
            del vals2
            id_upd = f.register_index(changed)
            id_diag = f.register_index(dfull, signs=dsigns)
            newv = np.ascontiguousarray(vals[changed])
            xf = [np.zeros(N) for _ in range(3)]
            for a in [newv] + bs + xf:
                f.pin_buffer(a)
            st1 = hip.Settings.default(iterative_refinement_max_iter=1, iterative_refinement_reltol=0.0, iterative_refinement_abstol=0.0,
                                       device=ks.settings.device)

            def fstep():
                f.update_values_id(id_upd, newv)
                f.offset_values_id(id_diag, eps)
                if not f.refactor():
                    raise RuntimeError("L1 refactor failed")
                f.offset_values_id(id_diag, -eps)
                for k in range(3):
                    ok, _ = f.solve_refined(xf[k], bs[k], st1)
                    if not ok:
                        raise RuntimeError("L1 refined solve failed")

            f.set_values(vals)   # (the unregularised values: the fast path applies the regulariser itself)
            fstep()
            t0 = time.perf_counter()
            for _ in range(steps):
                fstep()
            elf = time.perf_counter() - t0
            # against the device-resident L2 solve of the same right-hand side (same refinement setting)
            nm = w.n + w.m
            ks.set_settings(st1)
            ks.setrhs(bs[0][:w.n], bs[0][w.n:nm])
            xg, zg = np.zeros(w.n), np.zeros(w.m)
            okd = ks.solve(xg, zg)
            ks.set_settings(bench_settings(hip, ks.settings.device))
            bfull = np.concatenate([bs[0][:nm], np.zeros(N - nm)])
            okf, _ = f.solve_refined(xf[0], bfull, st1)
            errf = relerr(xf[0][:nm], np.concatenate([xg, zg])) if (okd and okf) else None
            fbytes = 8 * len(changed) + 3 * 2 * 8 * N
            fast = {"what": "the same iteration through registered index sets (%d changed entries of K + diag_full), the device-resident "
                            "refinement (chip_ldl_solve_refined, one round) and page-locked caller buffers: per step 1 x update_values, "
                            "2 x offset_values, 1 refactor, 3 refined solves" % len(changed),
                    "value": round(steps / elf, 3), "unit": "iterations/s", "ms_per_step": round(1e3 * elf / steps, 3),
                    "pcie_bytes_per_step": int(fbytes), "pcie_GBs": round(fbytes / (elf / steps) / 1e9, 1),
                    "rel_err_vs_L2_solution": errf, "ok": bool(errf is not None and errf <= TOL)}
        except Exception as ex:
            fast = {"error": repr(ex)[:300]}
        del f
        return {"what": "config 3 through chip_ldl_set_values / chip_ldl_refactor / chip_ldl_solve with HOST buffers (pageable numpy "
                        "arrays): per step 1 refactor + 6 solves, every operand over PCIe",
                "fast_path": fast,
                "value": round(steps / el, 3), "unit": "iterations/s", "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
                "pcie_bytes_per_step": int(bytes_step), "pcie_GBs": round(bytes_step / (el / steps) / 1e9, 1),
                "setup_s": round(t_setup, 2),
                "backward_error": berr, "rel_err_vs_oracle_ldl_solve": err, "ok": bool(berr <= 1e-10),
                "note": "raw LDL' solve of the statically regularised K (no refinement: that is the caller's loop at this boundary): "
                        "backward error ||b - K x||inf / (||K||inf ||x||inf + ||b||inf) by an independent SpMV, and the solution "
                        "against the oracle's factorisation of the same matrix with the same permutation (the pivots of the "
                        "regularised zero rows are +-1e-8: forward errors of 1e-7 between two elimination orders are rounding)"}
    except Exception as ex:
        return {"error": repr(ex)[:300]}


def launch_ranks(args):
    """stdlib launcher for --gpus N > 1 when no launcher set WORLD_SIZE: N subprocesses of this script, one per GPU,
    with the environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    raise SystemExit(rc)


class FileComm:
    """--fake-comm: the exchange interface of hip.Comm over files in /tmp, every rank on GPU 0 -- a PLUMBING CHECK of
    the N > 1 code of this script (sharding, parity reduction, checksums) on a single-GPU box, where RCCL refuses two
    ranks on one device.  Never a measurement: the line it produces says so."""

    def __init__(self, hip, world, rank):
        self.hip, self.world, self.rank, self.seq = hip, world, rank, 0
        self.base = "/tmp/chip_fakecomm_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())

    def _exchange(self, arr):
        self.seq += 1
        mine = "%s_%d_%d.npy" % (self.base, self.seq, self.rank)
        np.save(mine + ".tmp.npy", arr)
        os.replace(mine + ".tmp.npy", mine)
        out = []
        for r in range(self.world):
            path = "%s_%d_%d.npy" % (self.base, self.seq, r)
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 600:
                    raise RuntimeError("fake comm: rank %d missing" % r)
                time.sleep(0.002)
            out.append(np.load(path))
        return out

    def attach(self, ks):
        pass

    def wait(self, ks):
        pass

    def synchronize(self):
        pass

    def allgather_step(self, ks, send_ptr, recv_ptr, counts):
        ks.synchronize()
        parts = self._exchange(self.hip.DeviceArray.view(send_ptr, int(counts[self.rank])).numpy())
        self.hip.DeviceArray.view(recv_ptr, int(sum(counts))).copy_from(np.concatenate(parts))

    def allreduce(self, vals, op="sum"):
        parts = self._exchange(np.atleast_1d(np.asarray(vals, dtype=np.float64)))
        f = {"sum": np.sum, "max": np.max, "min": np.min}[op]
        return f(np.stack(parts), axis=0)

    def barrier(self):
        self.allreduce([0.0])


def rendezvous_id(hip, rank, world):
    """the 128-byte RCCL token from rank 0 to the others through a file in /tmp (single node; the
    launcher's pid makes the name unique per run).  Setup only, never in the timed region."""
    path = "/tmp/chip_comm_id_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    if rank == 0:
        tok = hip.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(tok)
        os.replace(path + ".tmp", path)
        return tok
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 300:
            raise RuntimeError("rank %d: no RCCL token from rank 0" % rank)
        time.sleep(0.05)
    return open(path, "rb").read()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["auto", "c2", "c3", "c4", "c5", "c5m"], default="auto",
                    help="auto: c3 at N = 1, c4 at N > 1; c2 / c5 / c5m (a quarter of c5): the other BASELINE configs, "
                         "device resident, N = 1 only")
    ap.add_argument("--nblocks", type=int, default=1000, help="c3: SOC blocks (default: config 3)")
    ap.add_argument("--blocksize", type=int, default=1000)
    ap.add_argument("--nbatch", type=int, default=1024, help="c4: independent SOCPs (default: config 4)")
    ap.add_argument("--force-comm", action="store_true",
                    help="c4 at N = 1: run the all-gather path with a one-rank RCCL communicator (plumbing check)")
    ap.add_argument("--coresident", default="",
                    help="BLOCKS:USEC -- rehearsal of a rank of the sharded run on ONE GPU: every step, where the all-gather of the step "
                         "direction is enqueued, a kernel of BLOCKS workgroups (256 threads) that holds its CUs for USEC microseconds is "
                         "launched on a stream of its own (what RCCL's ring kernel does next to the persistent launches); use with "
                         "--workload c4 --nbatch 128 --force-comm.  Needs the library's test hooks (chip_debug_spin).")
    ap.add_argument("--fake-comm", action="store_true",
                    help="N > 1 on ONE GPU with a file-based stand-in for the RCCL exchange: plumbing check of this script's "
                         "N > 1 code (never a measurement)")
    ap.add_argument("--gather-every-solve", action="store_true",
                    help="N > 1: all-gather the solution of each of the 3 solves, not only the step direction")
    ap.add_argument("--cpu-steps", type=int, default=-1, help="oracle steps for cpu_baseline (-1 auto, 0 off)")
    ap.add_argument("--no-extras", action="store_true", help="skip parity / cpu legs / batched_c4 (profiling runs)")
    ap.add_argument("--profile-family", type=int, default=5,
                    help="kernel family timed with hipEvents for the roofline (5 = k_bundle_ir, the fused solve + "
                         "refinement launch; 6 = k_bundle_factor; 1 = k_bundle_symv on the one-kernel-per-phase path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        launch_ranks(args)  # (does not return)
    hip = graft.load_package()
    import clarabel_rs_amd.synthetic as problems
    import clarabel_rs_amd.sharding as sharding
    ndev = hip.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = 0 if args.fake_comm else local_rank % ndev
    hip.set_device(device)
    workload = args.workload if args.workload != "auto" else ("c3" if world == 1 else "c4")
    if world > 1 and workload != "c4":
        raise SystemExit("N > 1 runs the sharded batched workload (c4)")
    if workload in ("c2", "c5", "c5m") and args.profile_family == 5:
        # (no fused launch for systems with a level-scheduled top) c2: the pipelined supernode substitution
        # (most of the step's kernel time); c5: the MFMA update tiles of the supernode factorisation
        args.profile_family = 11 if workload == "c2" else 7

    comm = gathered = counts = None
    if workload == "c3":
        pr = problems.portfolio_socp(args.nblocks, args.blocksize, seed=3)
        desc = ("portfolio SOCP (BASELINE config 3): n=%d, Zero(1)+NN(%d)+%d x SOC(%d)"
                % (pr["n"], pr["n"], args.nblocks, args.blocksize + 1))
    elif workload == "c2":
        pr = problems.random_qp(100000, 200000, band=50, seed=1)
        desc = "random sparse QP (BASELINE config 2): n=100000, m=200000, Nonnegative cone"
    elif workload in ("c5", "c5m"):
        nc = 200 if workload == "c5" else 50
        # (quarter size: the oracle runs live and needs the numpy restatement's Hs blocks)
        pr = problems.chordal_sdp(nc, 50, 10, nc, 51, seed=5, with_hs=(nc != 200 and not args.no_extras))
        desc = ("chordal SDP (BASELINE config 5%s): %d x PSD(50) cliques with overlap 10 + %d x SOC(51); PSD scalings, "
                "Hs = skron(R R') and all cone state on the device" % ("" if nc == 200 else ", quarter size", nc, nc))
    else:
        # whole elimination trees per rank, balanced by the blocks' factor work (identical blocks here)
        ranges = sharding.partition_blocks(np.ones(args.nbatch), world)
        b0, b1 = ranges[rank]
        pr = problems.batched_socp(b1 - b0, 2000, 2, seed=100 + b0)
        desc = ("batched SOCP (BASELINE config 4): %d independent SOCPs of n=2000 (2 x SOC(1001) + budget row each), "
                "%d per GPU" % (args.nbatch, b1 - b0))
        counts = [(e - b) * (pr["n"] + pr["m"]) // (b1 - b0) for b, e in ranges]
    w = Workload(hip, pr, device, rank)
    w.gather_every_solve = bool(args.gather_every_solve)
    if args.coresident:
        cb, cu = args.coresident.split(":")
        w.coresident = (int(cb), float(cu), device)
    info = w.ks.linear_solver_info()
    t_setup = w.t_setup
    if world > 1 or (args.force_comm and workload == "c4"):
        # (--force-comm: the exchange path with a one-rank communicator -- plumbing check on a single-GPU box)
        comm = FileComm(hip, world, rank) if args.fake_comm else hip.Comm(rendezvous_id(hip, rank, world), world, rank, device)
        comm.attach(w.ks)
        gathered = [hip.DeviceArray(int(sum(counts))) for _ in range(3)]
    elapsed, prof = w.run(args.steps, args.warmup, args.profile_family, comm, gathered, counts,
                          events_in_timed_region=False, sequential_pass=workload in ("c2", "c5", "c5m"))
    own_elapsed_main, own_repeats_main = getattr(w, "own_elapsed", elapsed), w.repeats  # (this rank's clock of the run `value` comes from)
    ir = w.ks.linear_solver_info().last_ir_iterations
    # N > 1: the same steps once more with the OTHER exchange policy, so that the driver's curve can be read either way
    # (SURVEY 8(e) says one all-gather per solve; the default gathers the step direction only)
    other_policy = None
    if comm is not None and world > 1:
        w.gather_every_solve = not w.gather_every_solve
        el2, _ = w.run(args.steps, args.warmup, 0, comm, gathered, counts)
        other_policy = {"policy": "3 x all-gather per step (every solve's solution)" if w.gather_every_solve else
                                  "1 x all-gather per step (the step direction: the last solve's solution)",
                        "value": round(args.steps / el2, 3), "unit": "iterations/s", "ms_per_step": round(1e3 * el2 / args.steps, 4)}
        w.gather_every_solve = not w.gather_every_solve

    # ---- every rank's own figures on the line (N > 1, or the one-rank rehearsal): its ms per step on its own clock, the
    #      solves it had to repeat, its fall-backs, and what ONE exchange costs when nothing else runs -- so that the first
    #      real run on several GPUs can be read: a slow rank, a rank whose persistent launches lost co-residency, or the
    #      collective itself
    per_rank = None
    if comm is not None:
        last = len(w.rhs) - 1
        w.ks.synchronize()
        comm.synchronize()
        comm.barrier()
        reps = 20
        t1 = time.perf_counter()
        for _ in range(reps):
            comm.wait(w.ks)
            comm.allgather_step(w.ks, w.lhs[last].ptr, gathered[last].ptr, counts)
        comm.synchronize()
        w.ks.synchronize()
        exch_us = 1e6 * (time.perf_counter() - t1) / reps
        mine = np.zeros(4 * world)
        mine[4 * rank:4 * rank + 4] = [1e3 * own_elapsed_main / args.steps, float(own_repeats_main), float(w.ks.fused_fallbacks()), exch_us]
        allv = np.asarray(comm.allreduce(mine.tolist(), "sum")).reshape(world, 4)
        per_rank = {"ms_per_step": [round(float(v), 4) for v in allv[:, 0]],
                    "fused_launch_repeats": [int(v) for v in allv[:, 1]],
                    "fused_fallbacks": [int(v) for v in allv[:, 2]],
                    "exchange_alone_us": [round(float(v), 1) for v in allv[:, 3]],
                    "what": "per rank, in rank order: ms per step on the rank's own clock (value uses the maximum), solves "
                            "repeated after a fused launch lost co-residency, solves that fell back to the per-phase path, and the "
                            "all-gather of the step direction alone (%d back-to-back exchanges of %d doubles, host clock)"
                            % (reps, int(sum(counts)))}

    # ---- N > 1: every rank checks ITS shard against the oracle; the ranks reduce the verdict ----------------
    parity_sharded = None
    if world > 1 and not args.no_extras:
        # (a) bit-for-bit: the gathered step direction against every rank's local solution -- checksums (sum and
        # sum of |.|) of each rank's segment of MY gathered copy against the owner's, all-reduced
        comm.wait(w.ks)
        comm.synchronize()
        w.ks.synchronize()
        last = len(w.lhs) - 1
        g = gathered[last].numpy()
        own = w.lhs[last].numpy()
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        mine = np.zeros(2 * world)
        mine[2 * rank], mine[2 * rank + 1] = float(np.sum(own)), float(np.sum(np.abs(own)))
        truth = comm.allreduce(mine.tolist(), "sum")   # (one non-zero contribution per entry: exact)
        seen = np.array([f(g[offs[r]:offs[r + 1]]) for r in range(world) for f in (np.sum, lambda a: np.sum(np.abs(a)))])
        seg_diff = float(np.max(np.abs(seen - np.asarray(truth))))
        own_equal = bool(np.array_equal(g[offs[rank]:offs[rank + 1]], own))
        # (b) this rank's solutions against the oracle on the same shard, permutation and right-hand sides
        par_r, _, _ = oracle_leg(w, args, time_it=False)
        red = comm.allreduce([par_r["rel_err_vs_oracle"], par_r["rel_err_fixed_r1"], seg_diff, 0.0 if own_equal else 1.0], "max")
        parity_sharded = {"rel_err_vs_oracle": red[0], "tol": TOL, "ok": bool(red[0] <= TOL and red[2] == 0.0 and red[3] == 0.0),
                          "what": "max over the %d ranks (RCCL all-reduce) of each rank's max over its 3 solves of "
                                  "||x_gpu - x_oracle||inf / max(1, ||x_oracle||inf) on its own shard, post-refinement, "
                                  "default refinement settings, same inputs and permutation" % world,
                          "rel_err_fixed_r1": red[1],
                          "gathered_vs_local": {"own_slice_bit_equal_on_every_rank": bool(red[3] == 0.0),
                                                "max_abs_diff_of_segment_checksums": red[2],
                                                "what": "every rank's segment of the all-gathered step direction (sum and "
                                                        "sum of |.|) against the owner's local solution"}}
    if rank == 0:
        ks, m = w.ks, w.m
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.steps / elapsed
        # byte model of the WHOLE problem (all ranks): every rank holds 1/world of it
        Bm = algorithmic_bytes(ks.N * world, ks.nnzK * world, info.nnzL * world, ks.nHs * world, m * world)
        # family 1 = residual of all bundle rows, K stored once (U): 12 B per streamed K entry + 24 B per row
        # family 5 = the fused launch: (r + 1) LDL' solves + (r + 1) residuals of SURVEY 8(d)'s per-unit figures
        Bu = algorithmic_bytes(ks.N, ks.nnzK, info.nnzL, ks.nHs, m)
        wm = ks.work_model()
        fam = args.profile_family
        nsolves = len(w.rhs) * (int(ir) + 1)
        # per-launch work of the profiled family; families 7 / 11: TOTAL work of the family per step
        fam_bytes = {1: 12 * ks.nnzU + 24 * ks.NF, 5: (int(ir) + 1) * (Bu["solve"] + Bu["symv"]),
                     6: Bu["factor"]}.get(fam)
        fam_name = {1: "k_bundle_symv (residual e = b - Kx over the %d bundle rows, ||e||inf folded in)" % ks.NF,
                    2: "k_gather_merged<1> (BWD top levels)", 3: "k_gather_merged<0> (FWD top levels)",
                    4: "k_factor_T",
                    5: "%s (one launch = setrhs + %d x (LDL' solve + residual) + refinement decisions + getlhs; "
                       "algorithmic bytes = %d x (B_solve + B_symv))"
                       % ("k_gstep_solve" if ks.step_kernels() & 1 else ("k_bundle_irs" if ks.step_kernels() & 4 else "k_bundle_ir"),
                          int(ir) + 1, int(ir) + 1),
                    6: "k_bundle_factor (numeric LDL' of all bundle columns)",
                    7: "k_snode_update (left-looking update of a 64-column block of every supernode of a unit level: "
                       "16 x 64 tiles of v_mfma_f64_16x16x4_f64)",
                    8: "k_snode_diag", 9: "k_snode_rows", 10: "k_snode_extend",
                    11: "k_snode_tri (pipelined substitution through the wide chain supernodes of one unit level, "
                        "forward and backward sweeps)",
                    12: "k_gather_merged (supernode substitution path)"}.get(fam, "?")
        # HBM bytes per launch from the PMC passes (tools/pmc_traffic.sh -> tools/pmc_summarize.py ->
        # profiles/*_pmc_traffic.json; FETCH_SIZE x2 + WRITE_SIZE, see that script).  PMC counters cannot be
        # collected from inside the timed run: the committed summary of the same command is QUOTED (with the
        # file it comes from) -- null when the workload differs from the profiled one.
        traffic = traffic_src = traffic_commit = None
        try:
            import glob
            suffix = {"c3": "", "c2": "_c2", "c5": "_c5"}.get(workload)
            if workload == "c4" and world == 1 and args.nbatch in (128, 256):
                suffix = "_c4_%d" % args.nbatch
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic%s.json" % suffix))) if suffix is not None else []
            knames = {1: ["k_bundle_symv"], 5: ["k_gstep_solve" if ks.step_kernels() & 1 else ("k_bundle_irs" if ks.step_kernels() & 4 else "k_bundle_ir")],
                      6: ["k_bundle_factor"],
                      7: ["k_snode_update"],
                      11: ["k_snode_gsweep", "k_snode_gfwd", "k_snode_gbwd"] if ks.sweep_model()["g_levels"] else ["k_snode_tri"]}.get(fam)
            full_size = (workload == "c3" and args.nblocks == 1000 and args.blocksize == 1000) or workload in ("c2", "c5", "c4")
            if pj and knames and full_size:
                kk = json.load(open(pj[-1]))["kernels"]
                have = [k for k in knames if k in kk]  # (launch-weighted mean over the family's kernels, per launch)
                traffic = round(sum(kk[k]["hbm_bytes"] * kk[k].get("launches_sampled", 1) for k in have) /
                                sum(kk[k].get("launches_sampled", 1) for k in have))
                traffic_src = os.path.basename(pj[-1])
                traffic_commit = json.load(open(pj[-1])).get("taken_at_commit")
        except Exception:
            traffic = None
        whole = {"algorithmic_bytes": Bm["iter"],
                 "achieved_GBs": round(Bm["iter"] / (ms_per_step * 1e-3) / 1e9, 1),
                 "frac": round(Bm["iter"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world, 4)}
        roof = None
        if prof["launches"] > 0 and fam_bytes:
            avg_ms = prof["ms"] / prof["launches"]
            ach = fam_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "frac_of_achievable": round(ach / HBM_ACHIEVABLE_GBS, 4),
                    "traffic": traffic, "traffic_profiled_in": traffic_src,
                    "traffic_profiled_at": traffic_commit or profile_commit(traffic_src),
                    "kernel": fam_name,
                    "launches": prof["launches"], "avg_launch_us": round(1e3 * avg_ms, 2),
                    "algorithmic_bytes_per_launch": fam_bytes, "whole_step": whole}
        elif prof["launches"] > 0 and fam == 7 and wm["sn_update_flops"] > 0:
            # f64 matrix cores: multiply-add flops of the update tiles from the supernode geometry (2 per
            # multiply-add; rows at or below the block only), one refactor per step
            per_launch = wm["sn_update_flops"] * args.steps / prof["launches"]
            avg_ms = prof["ms"] / prof["launches"]
            ach = per_launch / (avg_ms * 1e-3) / 1e12
            f_dense = wm["sn_update_flops"] + wm["sn_extend_flops"] + wm["sn_diag_rows_flops"]
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F64_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_profiled_in": traffic_src,
                    "kernel": fam_name,
                    "launches": prof["launches"], "avg_launch_us": round(1e3 * avg_ms, 2),
                    "flops_per_launch": round(per_launch), "flops_per_refactor_update_tiles": wm["sn_update_flops"],
                    "kernel_ms_per_step": round(prof["ms"] / args.steps, 3),
                    "whole_step": dict(whole, dense_flops_per_step=f_dense,
                                       dense_TFLOPs_over_step=round(f_dense / (ms_per_step * 1e-3) / 1e12, 2),
                                       dense_frac_of_mfma_peak=round(f_dense / (ms_per_step * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, 4))}
        elif prof["launches"] > 0 and fam == 11 and wm["sn_panel_entries"] > 0:
            roof = sweep_roofline(ks, wm, prof, args.steps, nsolves)
            roof.update({"traffic": traffic, "traffic_profiled_in": traffic_src, "launches": prof["launches"],
                         "kernel_ms_per_step": round(prof["ms"] / args.steps, 3), "whole_step": whole})
        elif prof["launches"] > 0:
            roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": fam_name, "launches": prof["launches"],
                    "avg_launch_us": round(1e3 * prof["ms"] / prof["launches"], 2),
                    "kernel_ms_per_step": round(prof["ms"] / args.steps, 3), "whole_step": whole}
        parity = cpu = cpu_mt = c4 = None
        extras = {}
        step_ms = getattr(w, "step_ms", None)
        w_repeats = w.repeats  # (solves repeated at collect time after a fused launch timed out: 0 in a clean run)
        w_events_pass = getattr(w, "events_pass", None)
        w_sequential = getattr(w, "sequential", None)
        w_fallbacks = int(w.ks.fused_fallbacks())
        if world > 1:
            parity = parity_sharded
        elif not args.no_extras and workload == "c5":
            parity = fixture_parity_c5(w, hip)
            if args.cpu_steps != 0:
                cpu = quarter_c5_cpu_baseline(hip, problems, args)
                cpu_mt = sn_leg(w, hip)
        elif not args.no_extras:
            parity, cpu, ko = oracle_leg(w, args, time_it=args.cpu_steps != 0)
            if args.cpu_steps != 0 and workload in ("c3", "c4"):
                cpu_mt = mt_leg(w, ko)
            elif args.cpu_steps != 0 and workload == "c2":
                cpu_mt = sn_leg(w, hip)
            del ko
            if workload == "c3" and args.workload == "auto":
                extras["l1_dropin"] = l1_dropin_leg(hip, w, args)
            if workload == "c3" and args.workload == "auto":
                # the N = 1 point of the sharded workload's strong-scaling curve: config 4 whole on this GPU
                del w
                pr4 = problems.batched_socp(args.nbatch, 2000, 2, seed=100)
                w4 = Workload(hip, pr4, device, 0)
                el4, _ = w4.run(args.steps, args.warmup)
                i4 = w4.ks.linear_solver_info()
                B4 = algorithmic_bytes(w4.ks.N, w4.ks.nnzK, i4.nnzL, w4.ks.nHs, w4.m)
                par4, _, _ = oracle_leg(w4, args, time_it=False)
                c4 = {"workload": "batched SOCP (BASELINE config 4): %d independent SOCPs of n=2000, whole problem on 1 GPU, "
                                  "device resident" % args.nbatch,
                      "value": round(args.steps / el4, 3), "unit": "iterations/s", "ms_per_step": round(1e3 * el4 / args.steps, 4),
                      "step_ms": w4.step_ms,
                      "kkt_dim": w4.ks.N, "nnz_L": int(i4.nnzL), "setup_s": round(w4.t_setup, 2),
                      "whole_step_frac_of_hbm_peak": round(B4["iter"] / (el4 / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                      "parity": par4,
                      "note": "the --gpus N > 1 lines run THIS workload sharded (strong scaling); this is their N = 1 base"}
                del w4
                # the other BASELINE configs and the strict drop-in boundary, compact (driver-visible in the default line)
                extras["c2"] = extra_workload(hip, problems, "c2", args, device)
                extras["c5"] = extra_workload(hip, problems, "c5", args, device)
        out = {
            "metric": baseline_metric(),
            "value": round(value, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "kkt_dim": ks.N * world if workload == "c4" else ks.N,
                       "kkt_dim_per_gpu": ks.N, "nnz_triu_K": ks.nnzK, "nnz_L": int(info.nnzL),
                       "etree_levels": int(info.n_levels),
                       "per_step": "1 update(scaling+Hs+static reg+refactor) + (2 paired + 1) solves x (LDL solve + 1 IR round): the constant "
                                   "right-hand side and the affine direction are independent and go to the device as one call "
                                   "(chip_kkt_solve2_dev_enqueue); the combined direction depends on the affine result",
                       "other_solve_policy": w_sequential,
                       "ir_rounds": int(ir), "setup_s": round(t_setup, 2),
                       "gpus_on_problem": int(info.threads) if world == 1 else world,
                       "collective": "FILE-BASED STAND-IN on one GPU (--fake-comm): a plumbing check, NOT a measurement" if args.fake_comm else
                                     ("%s per step, native RCCL (ncclAllGather fp64, %d doubles) on its own stream, event-ordered"
                                      % ("3 x all-gather (every solve's solution)" if args.gather_every_solve else
                                         "1 x all-gather of the step direction (the last solve's solution)",
                                         int(sum(counts)))) if comm is not None else "none"},
            "step_ms": step_ms,
            "timing": "value / ms_per_step: K steps between two synchronisations WITHOUT the hipEvent pairs of the roofline "
                      "measurement on the stream; roofline: the same K steps once more, right after, with a pair around every launch "
                      "of the dominant kernel (roofline_events_pass.ms_per_step_with_events is that pass)",
            "roofline_events_pass": w_events_pass,
            "roofline": roof, "parity": parity, "cpu_baseline": cpu, "cpu_baseline_mt": cpu_mt, "batched_c4": c4,
            "c2": extras.get("c2"), "c5": extras.get("c5"), "l1_dropin": extras.get("l1_dropin"),
            "other_exchange_policy": other_policy,
            "fused_launch_repeats": int(w_repeats),
            "per_rank": per_rank,
            "rehearsal": None if not args.coresident else {
                "what": "one rank of the sharded run rehearsed on one GPU: the exchange path with a one-rank RCCL communicator "
                        "(the all-gather enqueued every step behind the last solve, overlapping the next step) plus a co-resident "
                        "kernel of %s workgroups x 256 threads spinning for %s us per step on its own stream -- what RCCL's ring "
                        "kernel occupies next to the persistent launches at N = 8" % tuple(args.coresident.split(":")),
                "ms_per_step": round(ms_per_step, 4), "fused_launch_repeats": int(w_repeats),
                "fused_fallbacks": w_fallbacks},
        }
        print(json.dumps(out))
        sys.stdout.flush()
    # the JSON line stays the LAST line of stdout: RCCL prints a version banner to file descriptor 1 when a communicator
    # goes away (at interpreter exit) -- everything written to it from here on goes to stderr
    try:
        sys.stdout.flush()
        os.dup2(2, 1)
    except OSError:
        pass
    if comm is not None:
        comm.synchronize()
        comm.barrier()


if __name__ == "__main__":
    main()
