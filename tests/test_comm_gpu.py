"""-m gpu: the RCCL exchange layer of the C ABI (csrc/comm.cpp).  A single-GPU box can only form a
1-rank communicator (RCCL refuses two ranks on one device), which still exercises the whole call path:
token, ncclCommInitRank, event ordering between the engine's stream and the communicator's stream, the
all-gather and the scalar all-reduce.  With >= 2 GPUs the sharded step direction is compared with the
unsharded solve."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from tests import problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve_shard(hip, pr, rx, rz):
    ks = hip.HipKKTSolver(hip.CscMatrix(pr["n"], pr["n"], *pr["P"]), hip.CscMatrix(pr["m"], pr["n"], *pr["A"]),
                          pr["cones"], pr["m"], pr["n"])
    assert ks.update_scaling(pr["s"], pr["z"]) and ks.update()
    d_rx, d_rz = hip.DeviceArray(rx), hip.DeviceArray(rz)
    lhs = hip.DeviceArray(pr["n"] + pr["m"])
    ks.setrhs_dev(d_rx.ptr, d_rz.ptr)
    assert ks.solve_dev(lhs.ptr, lhs.ptr + 8 * pr["n"])
    return ks, lhs


def test_comm_single_rank_allgather_and_allreduce(hip):
    pr = problems.batched_socp(8, 200, 2, seed=100)
    rng = np.random.default_rng(3)
    rx, rz = rng.standard_normal(pr["n"]), rng.standard_normal(pr["m"])
    ks, lhs = _solve_shard(hip, pr, rx, rz)
    comm = hip.Comm(hip.comm_unique_id(), 1, 0)
    comm.attach(ks)
    assert ks.linear_solver_info().threads == 1
    out = hip.DeviceArray(pr["n"] + pr["m"])
    comm.allgather_step(ks, lhs.ptr, out.ptr, [pr["n"] + pr["m"]])   # enqueued behind the solve by an event
    comm.wait(ks)
    comm.synchronize()
    ks.synchronize()
    assert np.array_equal(out.numpy(), lhs.numpy())
    assert np.array_equal(comm.allreduce([1.5, -2.0], "sum"), [1.5, -2.0])
    assert np.array_equal(comm.allreduce([3.0], "max"), [3.0])
    comm.barrier()


def test_rank_rehearsal_exchange_beside_the_step_kernels(hip, oracle):
    """One rank of the 8-GPU sharded run rehearsed on one GPU: a 128-tree share (the register-resident step kernels, whose
    grid fills the chip exactly), five iterations; behind every iteration's last solve the all-gather of the step direction
    is enqueued on the communicator's stream, FOLLOWED by a kernel that holds 16 workgroups for 60 us on that stream (the
    time RCCL's ring kernel occupies CUs when eight ranks exchange).  The exchange runs beside the next iteration's cone
    update and factorisation; the next persistent solve launch waits for it on the device -- a foreign wave that is
    resident while such a launch starts fragments the register file and strands some of its workgroups (measured: every
    solve then ran into its wait budget and was repeated).  No repeats, no fallbacks, solutions against the oracle."""
    pr = problems.batched_socp(128, 2000, 2, seed=100)
    n, m = pr["n"], pr["m"]
    ks = hip.HipKKTSolver(hip.CscMatrix(n, n, *pr["P"]), hip.CscMatrix(m, n, *pr["A"]), pr["cones"], m, n)
    assert ks.step_kernels() == 3
    comm = hip.Comm(hip.comm_unique_id(), 1, 0)
    comm.attach(ks)
    rng = np.random.default_rng(5)
    s_d, z_d = hip.DeviceArray(pr["s"]), hip.DeviceArray(pr["z"])
    rhs = [(rng.standard_normal(n), rng.standard_normal(m)) for _ in range(3)]
    dev = [(hip.DeviceArray(a), hip.DeviceArray(b)) for a, b in rhs]
    lhs = [hip.DeviceArray(n + m) for _ in range(3)]
    gathered = hip.DeviceArray(n + m)
    for it in range(5):
        ks.update_scaled_enqueue(s_d.ptr, z_d.ptr)
        for k in range(3):
            if k == 2:
                comm.wait(ks)  # (the buffers of the previous iteration's exchange are about to be overwritten)
            ks.setrhs_dev(dev[k][0].ptr, dev[k][1].ptr)
            ks.solve_dev_enqueue(lhs[k].ptr, lhs[k].ptr + 8 * n)
        comm.allgather_step(ks, lhs[2].ptr, gathered.ptr, [n + m])
        comm.debug_spin(16, 256, 60.0)
        uok, sok = ks.collect()
        assert uok and sok == [True, True, True] and not ks.repeated_solves, (it, sok, ks.repeated_solves)
    comm.wait(ks)
    comm.synchronize()
    ks.synchronize()
    assert ks.fused_fallbacks() == 0
    assert np.array_equal(gathered.numpy(), lhs[2].numpy())
    cones = oracle.Cones(pr["cones"])
    assert cones.update_scaling(pr["s"], pr["z"])
    ko = oracle.KKTSolver(n, m, pr["P"], pr["A"], cones, perm=ks.perm)
    assert ko.update()
    for k in range(3):
        ko.setrhs(*rhs[k])
        ok, xo, zo = ko.solve()
        ref = np.concatenate([xo, zo])
        assert ok and float(np.max(np.abs(lhs[k].numpy() - ref)) / max(1.0, np.max(np.abs(ref)))) <= 1e-8


def _rank_main(rank, world, token, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    hip = g.load_package()
    from tests import problems as P
    import clarabel_rs_amd.sharding as sharding
    hip.set_device(rank)
    parts = [P.portfolio_socp(2, 40 + 10 * (i % 3), seed=100 + i) for i in range(6)]   # ragged blocks
    ranges = sharding.partition_blocks([p["n"] + p["m"] for p in parts], world)
    layout = sharding.ShardLayout([p["n"] for p in parts], [p["m"] for p in parts], ranges)
    b, e = ranges[rank]
    mine = P.blockdiag(parts[b:e])
    rng = np.random.default_rng(7)
    gx, gz = rng.standard_normal(layout.n), rng.standard_normal(layout.m)
    x0, z0 = sum(layout.n_rank[:rank]), sum(layout.m_rank[:rank])
    ks, lhs = _solve_shard(hip, mine, gx[x0:x0 + mine["n"]], gz[z0:z0 + mine["m"]])
    comm = hip.Comm(token, world, rank, rank)
    comm.attach(ks)
    out = hip.DeviceArray(sum(layout.len_rank))
    comm.allgather_step(ks, lhs.ptr, out.ptr, layout.len_rank)   # ragged counts: group of broadcasts
    comm.synchronize()
    got = out.numpy()[layout.global_index_packed()]
    nrm = comm.allreduce([float(np.max(np.abs(got)))], "max")[0]
    threads = ks.linear_solver_info().threads
    if rank == 0:
        whole = P.blockdiag(parts)
        ksw, lw = _solve_shard(hip, whole, gx, gz)
        ksw.synchronize()
        q.put((float(np.max(np.abs(got - lw.numpy()))), nrm, threads))
    comm.barrier()


def test_sharded_step_direction_two_gpus(hip):
    if hip.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    token = hip.comm_unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, token, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, nrm, threads = q.get(timeout=300)
    for p in procs:
        p.join(60)
    assert err <= 1e-9 * max(1.0, nrm) and threads == 2


def test_bench_sharded_code_path_with_file_comm(hip):
    """bench.py's N > 1 code (stdlib launcher, sharding by whole trees, per-rank oracle parity reduced over the ranks,
    checksums of the gathered step direction) exercised on ONE GPU: two ranks on device 0 with the file-based
    stand-in for the RCCL exchange (--fake-comm) -- RCCL itself refuses two ranks on one device."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fake-comm", "--nbatch", "8",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and "STAND-IN" in line["config"]["collective"]
    par = line["parity"]
    assert par["ok"] and par["rel_err_vs_oracle"] <= 1e-8
    assert par["gathered_vs_local"]["own_slice_bit_equal_on_every_rank"]
    assert par["gathered_vs_local"]["max_abs_diff_of_segment_checksums"] == 0.0
    pr_ = line["per_rank"]  # (every rank's own clock, repeats, fall-backs and the exchange alone: what a real N > 1 run is read by)
    assert len(pr_["ms_per_step"]) == 2 and all(v > 0 for v in pr_["ms_per_step"]) and all(v > 0 for v in pr_["exchange_alone_us"])
    assert pr_["fused_launch_repeats"] == [0, 0] and max(pr_["ms_per_step"]) <= line["ms_per_step"] * 1.01 + 1e-3  # (rounded to four decimals on the line)


def test_sharded_scalars_of_an_iteration_through_allreduce(hip):
    """The only numbers of an interior-point iteration that cross ranks when a block-diagonal problem is sharded by whole
    blocks: the two sums of the step in tau (default/kktsystem.rs:175-186: q'x1 + b'z1 + 2 xi'P x1, and
    q'x2 + b'z2 - |xi - x2|_P^2 + |x2|_P^2) and the step length, a minimum over the cones (compositecone.rs:300-340).
    Two shards of a four-block QP are solved by their own handles; every shard's partial sums go through
    chip_comm_allreduce (a 1-rank communicator: RCCL refuses two ranks on one device -- the call sequence and the
    reduction operators are what is pinned, the second rank's contribution is added on the host as the collective
    would) and the results are compared with the unsharded problem's."""
    import scipy.sparse as sp
    parts = [problems.random_qp(300, 600, band=10, seed=40 + i) for i in range(4)]
    whole = problems.blockdiag(parts)
    shards = [problems.blockdiag(parts[:2]), problems.blockdiag(parts[2:])]
    rng = np.random.default_rng(11)
    n, m = whole["n"], whole["m"]
    q, b = rng.standard_normal(n), rng.standard_normal(m)
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    xvar, tau, kappa, rtau, rkappa = rng.standard_normal(n), 1.3, 0.7, 0.4, -0.2
    dz_, ds_ = rng.standard_normal(m), rng.standard_normal(m)

    def Pfull(pr):
        Pu = sp.csc_matrix((pr["P"][2], pr["P"][1], pr["P"][0]), shape=(pr["n"], pr["n"]))
        return Pu + sp.triu(Pu, 1).T

    def shard_values(pr, sl_n, sl_m):
        """one shard: the two solves, its partial sums and its step length"""
        ks = hip.HipKKTSolver(hip.CscMatrix(pr["n"], pr["n"], *pr["P"]), hip.CscMatrix(pr["m"], pr["n"], *pr["A"]),
                              pr["cones"], pr["m"], pr["n"])
        assert ks.update_scaling(pr["s"], pr["z"]) and ks.update()
        sols = []
        for fx, fz in ((rx[sl_n], rz[sl_m]), (-q[sl_n], b[sl_m])):  # (x1, z1) and the constant part (x2, z2), :108-125
            ks.setrhs(fx, fz)
            x, z = np.zeros(pr["n"]), np.zeros(pr["m"])
            assert ks.solve(x, z)
            sols.append((x, z))
        (x1, z1), (x2, z2) = sols
        P = Pfull(pr)
        xi = xvar[sl_n] / tau
        num = q[sl_n] @ x1 + b[sl_m] @ z1 + 2.0 * (xi @ (P @ x1))
        den = -(q[sl_n] @ x2) - (b[sl_m] @ z2) + (xi - x2) @ (P @ (xi - x2)) - x2 @ (P @ x2)
        D = hip.DeviceArray
        d_dz, d_ds, d_z, d_s = D(dz_[sl_m]), D(ds_[sl_m]), D(pr["z"]), D(pr["s"])  # (kept alive across the call)
        alpha = ks.step_length_dev(d_dz.ptr, d_ds.ptr, d_z.ptr, d_s.ptr, 1.0)
        return np.array([num, den]), alpha, np.concatenate([x1, z1])

    truth_sums, truth_alpha, truth_sol = shard_values(whole, slice(0, n), slice(0, m))
    comm = hip.Comm(hip.comm_unique_id(), 1, 0)
    sums, alpha, off_n, off_m, sols = np.zeros(2), np.inf, 0, 0, []
    for pr in shards:
        part, a, sol = shard_values(pr, slice(off_n, off_n + pr["n"]), slice(off_m, off_m + pr["m"]))
        sums += np.asarray(comm.allreduce(part.tolist(), "sum"))   # this rank's share through the collective
        alpha = min(alpha, comm.allreduce([a], "min")[0])
        sols.append((sol[:pr["n"]], sol[pr["n"]:]))
        off_n += pr["n"]
        off_m += pr["m"]
    dtau = (rtau - rkappa / tau + sums[0]) / (kappa / tau + sums[1])
    dtau_truth = (rtau - rkappa / tau + truth_sums[0]) / (kappa / tau + truth_sums[1])
    assert abs(dtau - dtau_truth) <= 1e-9 * max(1.0, abs(dtau_truth))
    assert abs(alpha - truth_alpha) <= 1e-12 * max(1.0, abs(truth_alpha))
    gathered = np.concatenate([s_[0] for s_ in sols] + [s_[1] for s_ in sols])  # (the all-gathered step direction)
    assert np.max(np.abs(gathered - truth_sol)) <= 1e-8 * max(1.0, np.max(np.abs(truth_sol)))
