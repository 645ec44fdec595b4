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
import math

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


def work_partition(n_views, film_rows, world):
    """Work units of one step for `world` ranks: a list (one entry per rank) of lists of (view, row0, row1) with
    film-block row windows [row0, row1) of a film block with `film_rows` (= H + 4) rows.

    world divides n_views : whole views, dealt round-robin over the ring (every rank sees equally expensive views).
    otherwise             : every view is cut into t = world / gcd(n_views, world) row windows ("pixel tiles within a view
                            when N > views-per-iteration") and the n_views * t windows are dealt round-robin --
                            12 views on 8 ranks: 24 half-views, 3 per rank, instead of the 2/1 imbalance of whole views."""
    if world < 1 or n_views < 1:
        raise ValueError("bad n_views / world")
    # smallest number of windows per view that makes the units divide evenly over the ranks (1 when world divides n_views)
    tiles = world // math.gcd(n_views, world)
    edges = [round(k * film_rows / tiles) for k in range(tiles + 1)]
    units = [(v, edges[k], edges[k + 1]) for v in range(n_views) for k in range(tiles)]
    return [units[r::world] for r in range(world)]


def rank_windows(units):
    """Groups a rank's (view, row0, row1) units by row window: [((row0, row1), [views...]), ...] -- one library call per
    window renders all of the rank's views that share it."""
    out = {}
    for v, r0, r1 in units:
        out.setdefault((r0, r1), []).append(v)
    return sorted(out.items())


def all_reduce_sum(t, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def render_step(ops, n_views, W, H, rank, world, loss_grad, grad_grid, group=None):
    """One differentiable render of `n_views` views shared by `world` ranks, split by views and -- when world does not
    divide them -- by pixel tiles (work_partition).  `ops` supplies the four film-level operators of the renderer for a
    list of views and a row window (dsdf.render_film / develop / GradSweep on a GPU; the oracle's in the CPU tests):
        ops.film(views, rows)                      -> primal film blocks (len(views), H+4, W+4, C) of the window's samples
        ops.develop(film_total)                    -> images (n, H, W, 3)
        ops.sweep(views, rows)                     -> (gradient-pass film blocks, handle)
        ops.backward(handle, film_total, grad_image, grad_grid)   accumulates dL/dsdf of the window's samples
    Exchange steps: sum of the primal films, sum of the gradient-pass films (only views that are actually split would need
    them; all views are reduced here for simplicity: 2 x n_views x 2 MiB at 512^2), and the ONE sum of dL/dsdf.
    loss_grad(images) -> dL/d(images).  Returns the images (identical on every rank)."""
    import torch
    units = work_partition(n_views, H + 4, world)[rank]
    wins = rank_windows(units)
    film = None
    for rows, views in wins:
        f = ops.film(views, rows)
        if film is None:
            film = torch.zeros((n_views,) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
        film[views] += f
    if film is None:
        film = ops.empty_film(n_views)
    all_reduce_sum(film, group)
    images = ops.develop(film)
    grad_images = loss_grad(images)
    film_g = torch.zeros_like(film)
    handles = []
    for rows, views in wins:
        f, h = ops.sweep(views, rows)
        film_g[views] += f
        handles.append((views, h))
    all_reduce_sum(film_g, group)
    for views, h in handles:
        ops.backward(h, film_g[views].contiguous(), grad_images[views].contiguous(), grad_grid)
    all_reduce_gradients([grad_grid], group)
    return images


class HipOps:
    """The film-level operators of render_step on the HIP path (dsdf.render_film / develop / GradSweep)."""

    def __init__(self, grid, sensors, spp, spp_grad, seeds, seeds_grad, integrator=0, **kw):
        from . import renderer
        self.r = renderer
        self.grid, self.sensors, self.spp, self.spp_grad = grid, list(sensors), int(spp), int(spp_grad)
        self.seeds, self.seeds_grad, self.integrator, self.kw = list(seeds), list(seeds_grad), integrator, kw
        self.W, self.H = self.sensors[0].film_size()

    def empty_film(self, n):
        return self.r.new_film(n, self.W, self.H, self.integrator, self.grid.device)

    def film(self, views, rows):
        f = self.empty_film(len(views))
        return self.r.render_film(self.grid, [self.sensors[v] for v in views], self.spp, f, rows,
                                  seeds=[self.seeds[v] for v in views], integrator=self.integrator, **self.kw)

    def develop(self, film):
        return self.r.develop(film, self.W, self.H, self.integrator)

    def sweep(self, views, rows):
        h = self.r.GradSweep(self.grid, [self.sensors[v] for v in views], self.spp_grad, rows,
                             seeds=[self.seeds_grad[v] for v in views], integrator=self.integrator, **self.kw)
        return h.sweep(self.empty_film(len(views))), h

    def backward(self, handle, film_total, grad_image, grad_grid):
        handle.backward(film_total, grad_image, grad_grid)


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
