"""N > 1 path on CPU: world_size-2 gloo run of the block sharding + all-gather of the step
direction (layout: clarabel.rs_amd/sharding.py; gloo all-gather: tests/gloo_gather.py).  The per-rank numeric engine is the ORACLE here
(there is no GPU in this container; the HIP engine has no CPU fallback) -- what is under
test is the partition, the global index layout and the collective, checked against the
unsharded oracle solve of the whole block-diagonal problem."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nbatch, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    from oracle import oracle as orc
    from tests import problems
    from tests.gloo_gather import all_gather_step
    pkg = g.load_package()
    sharding = __import__("importlib").import_module("clarabel_rs_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parts = [problems.portfolio_socp(2, 6 + (i % 3), seed=100 + i) for i in range(nbatch)]
        weights = [p["n"] + p["m"] for p in parts]
        ranges = sharding.partition_blocks(weights, world)
        layout = sharding.ShardLayout([p["n"] for p in parts], [p["m"] for p in parts], ranges)
        b, e = ranges[rank]
        mine = problems.blockdiag(parts[b:e])
        cones = orc.Cones(mine["cones"])
        cones.update_scaling(mine["s"], mine["z"])
        ks = orc.KKTSolver(mine["n"], mine["m"], mine["P"], mine["A"], cones)
        assert ks.update()
        # the same global right-hand side on every rank, sliced by the layout
        rng = np.random.default_rng(7)
        gx, gz = rng.standard_normal(layout.n), rng.standard_normal(layout.m)
        x0 = sum(layout.n_rank[:rank])
        z0 = sum(layout.m_rank[:rank])
        ks.setrhs(gx[x0:x0 + mine["n"]], gz[z0:z0 + mine["m"]])
        ok, x, z = ks.solve()
        assert ok
        local = torch.tensor(np.concatenate([x, z]))
        full = all_gather_step(local, layout, dist).numpy()
        if rank == 0:
            whole = problems.blockdiag(parts)
            cw = orc.Cones(whole["cones"])
            cw.update_scaling(whole["s"], whole["z"])
            kw = orc.KKTSolver(whole["n"], whole["m"], whole["P"], whole["A"], cw)
            assert kw.update()
            kw.setrhs(gx, gz)
            okw, xw, zw = kw.solve()
            ref = np.concatenate([xw, zw])
            q.put(("ok", float(np.max(np.abs(full - ref)) / max(1.0, np.max(np.abs(ref)))), ranges))
    except Exception as ex:  # pragma: no cover
        if rank == 0:
            q.put(("err", repr(ex), None))
        raise
    finally:
        dist.destroy_process_group()
    del pkg


@pytest.mark.parametrize("nbatch", [5, 8])
def test_sharded_blocks_gloo_world2(nbatch):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nbatch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    tag, err, ranges = q.get(timeout=10)
    assert tag == "ok", err
    assert err <= 1e-9
    assert ranges[0][0] == 0 and ranges[-1][1] == nbatch and all(b < e for b, e in ranges)


def test_partition_blocks_balance():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    sharding = __import__("importlib").import_module("clarabel_rs_amd.sharding")
    w = np.ones(1024)
    r = sharding.partition_blocks(w, 8)
    assert [e - b for b, e in r] == [128] * 8  # config 4: 128 blocks per GPU
    r = sharding.partition_blocks([5, 1, 1, 1, 1, 1], 2)
    assert r == [(0, 1), (1, 6)]
    r = sharding.partition_blocks([1, 1, 1], 8)  # fewer blocks than ranks: trailing ranks empty
    assert r[0] == (0, 1) and r[-1][1] == 3
    lay = sharding.ShardLayout([2, 3, 4], [5, 6, 7], [(0, 2), (2, 3)])
    gi = lay.global_index()
    assert lay.maxlen == 16 and len(gi) == 27 and len(set(gi.tolist())) == 27
