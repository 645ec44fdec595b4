"""Multi-GPU layer: one process per GPU, sensors (views) sharded across ranks, one
RCCL all-reduce of the per-voxel gradient buffers per optimisation step.

The reference is single-device.  Rays are independent given a
read-only grid and every cross-ray interaction is additive, so the path shards
over views with a full grid replica per GPU; the only exchange step is the sum
of dL/dsdf (plus any other per-voxel gradient, e.g. an albedo volume) -- fused
into a single flat bucket so xGMI sees one large collective instead of several
small ones.  Works with any torch.distributed backend (`nccl` = RCCL on ROCm;
`gloo` for the CPU tests of the host logic).
"""
import torch


def view_shard(n_views, rank, world):
    """Balanced contiguous shard of view indices for `rank` (first `n_views % world`
    ranks get one extra).  Every view is owned by exactly one rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n_views, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def strided_view_shard(indices, rank, world):
    """Shard of an already chosen per-iteration view batch (e.g. the reference's strided
    `get_sensor_iterator` batch, python/opt_configs.py:57-66): round-robin so that each
    rank's views are spread around the ring."""
    return list(indices)[rank::world]


def all_reduce_gradients(tensors, group=None):
    """Sums the given gradient tensors over all ranks in place with ONE collective."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensors
    tensors = list(tensors)
    if len(tensors) == 1 and tensors[0].is_contiguous():
        dist.all_reduce(tensors[0], op=dist.ReduceOp.SUM, group=group)
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors


def broadcast_parameters(tensors, src=0, group=None):
    """Makes every replica start from rank `src`'s parameters (grid, textures)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
    return tensors
