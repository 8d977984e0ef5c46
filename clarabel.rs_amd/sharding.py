"""Sharding of block-diagonal KKT systems over the GPUs of one node.

The KKT path shards across connected components of the elimination forest: a problem
made of independent blocks (BASELINE config 4: 1024 independent SOCPs; or N portfolio
shards) is split by whole blocks over ranks, one process per GPU.  Factorisation and
triangular solves of a rank's blocks need no exchange; the only collective is an
all-gather of the step direction (native RCCL from C++: csrc/comm.cpp, chip_kkt_allgather_step;
the CPU tests exercise this layout with a gloo all-gather that lives in tests/gloo_gather.py).

Only layout + collective plumbing lives here; the numeric work is done by whatever
`solver` object the caller provides (the product uses HipKKTSolver).
"""
import numpy as np


def partition_blocks(weights, world):
    """contiguous split of blocks 0..len(weights)-1 over `world` ranks, balanced by weight
    (e.g. sum c_j^2 of each block's factor); returns [(begin, end)] per rank."""
    weights = np.asarray(weights, dtype=np.float64)
    nb = len(weights)
    if world <= 0:
        raise ValueError("world must be positive")
    csum = np.concatenate([[0.0], np.cumsum(weights)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(csum, target, side="left"))
        # keep every rank non-empty when there are enough blocks
        k = max(k, bounds[-1] + (1 if nb >= world else 0))
        k = min(k, nb - (world - r) if nb >= world else nb)
        bounds.append(k)
    bounds.append(nb)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


class ShardLayout:
    """where each rank's (x_r, z_r) lands in the global (x, z) of the block-diagonal problem"""

    def __init__(self, part_n, part_m, ranges):
        self.part_n = list(part_n)
        self.part_m = list(part_m)
        self.ranges = list(ranges)
        self.world = len(ranges)
        self.n_rank = [int(sum(self.part_n[b:e])) for b, e in ranges]
        self.m_rank = [int(sum(self.part_m[b:e])) for b, e in ranges]
        self.n = int(sum(self.part_n))
        self.m = int(sum(self.part_m))
        self.len_rank = [a + b for a, b in zip(self.n_rank, self.m_rank)]
        self.maxlen = max(self.len_rank) if self.len_rank else 0

    def global_index(self):
        """index array g such that global[(x|z)] = gathered_padded[g], where gathered_padded is
        the all-gather of every rank's [x_r, z_r] padded to `maxlen`"""
        xs, zs = [], []
        for r in range(self.world):
            base = r * self.maxlen
            xs.append(base + np.arange(self.n_rank[r]))
            zs.append(base + self.n_rank[r] + np.arange(self.m_rank[r]))
        return np.concatenate(xs + zs) if xs else np.zeros(0, dtype=np.int64)


def _global_index_packed(self):
    """the same for an UNPADDED gather (chip_kkt_allgather_step with ragged counts): rank r's
    [x_r, z_r] starts at sum(len_rank[:r])"""
    xs, zs, base = [], [], 0
    for r in range(self.world):
        xs.append(base + np.arange(self.n_rank[r]))
        zs.append(base + self.n_rank[r] + np.arange(self.m_rank[r]))
        base += self.len_rank[r]
    return np.concatenate(xs + zs) if xs else np.zeros(0, dtype=np.int64)


ShardLayout.global_index_packed = _global_index_packed
