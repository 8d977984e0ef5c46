"""Test-only helper (CPU, gloo): the all-gather of the sharded step direction with torch.distributed, used by
tests/test_multigpu_gloo.py to exercise clarabel.rs_amd/sharding.py's partition and global index layout without
a GPU.  The product's exchange step is native RCCL (csrc/comm.cpp) and never imports torch."""


def all_gather_step(local_lhs, layout, dist, out=None, index=None):
    """all-gather of the local [dx_r, dz_r] (torch tensor, length <= layout.maxlen) into the
    global [dx, dz].  `dist` is torch.distributed (initialised by the caller)."""
    import torch
    pad = local_lhs
    if local_lhs.numel() != layout.maxlen:
        pad = torch.zeros(layout.maxlen, dtype=local_lhs.dtype, device=local_lhs.device)
        pad[:local_lhs.numel()] = local_lhs
    gathered = torch.empty(layout.world * layout.maxlen, dtype=pad.dtype, device=pad.device)
    dist.all_gather_into_tensor(gathered, pad)
    if index is None:
        index = torch.as_tensor(layout.global_index(), device=pad.device)
    res = gathered[index]
    if out is not None:
        out.copy_(res)
        return out
    return res
