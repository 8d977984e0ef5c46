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
