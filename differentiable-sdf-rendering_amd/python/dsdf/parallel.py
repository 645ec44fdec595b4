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


def cost_partition(cost, world):
    """COST-AWARE work units (round 6).  `cost`: (n_views, film_rows) array of predicted work per film-block
    row of every view (row_costs() below: from the per-pixel proofs of the previous step -- the pixels whose samples are traced
    dominate, pixels that are proven away cost their film weights, far pixels nothing).  The uniform deal of work_partition() left
    the ranks of the bench scene 1.36 x (N = 4) and 1.64 x (N = 8) apart (profiles/r05last_bench.json) because views and upper /
    lower halves of views differ that much in silhouette length.

    Shape of the result = shape of work_partition(): with G = gcd(n_views, world) the views form G groups of n_views / G views,
    every group is cut into t = world / G row windows, rank g * t + k renders window k of the views of group g -- so a rank still
    makes ONE library call per pass (all its units share a row window).  What changes:
      * the views are dealt to the groups by predicted cost (largest first, each to the lightest group that has room), not round-robin;
      * a group's windows are cut where its summed row costs reach k / t of its total, not at k / t of the rows.
    Deterministic in `cost` (every rank computes the same partition from the same all-reduced matrix)."""
    import numpy as np
    cost = np.asarray(cost, np.float64)
    if cost.ndim != 2 or world < 1:
        raise ValueError("cost must be (n_views, film_rows)")
    n_views, rows = cost.shape
    G = math.gcd(n_views, world)
    tiles, per_group = world // G, n_views // G
    if rows < tiles:
        raise ValueError("more row windows than film rows")
    cost = np.where(np.isfinite(cost) & (cost > 0), cost, 0.0)
    vc = cost.sum(1)
    groups, sums = [[] for _ in range(G)], [0.0] * G
    for v in sorted(range(n_views), key=lambda i: (-vc[i], i)):
        g = min((g for g in range(G) if len(groups[g]) < per_group), key=lambda g: (sums[g], g))
        groups[g].append(v); sums[g] += vc[v]
    out = []
    for g in range(G):
        views = sorted(groups[g])
        h = cost[views].sum(0) + 1e-12 * max(vc.sum(), 1.0) / rows          # (a floor: an all-zero histogram is cut evenly)
        c = np.concatenate([[0.0], np.cumsum(h)])
        edges = [0]
        for k in range(1, tiles):
            e = int(np.searchsorted(c, c[-1] * k / tiles))
            # the row whose inclusion crosses the target goes to the side that leaves the smaller error
            if e > 0 and abs(c[e - 1] - c[-1] * k / tiles) <= abs(c[e] - c[-1] * k / tiles):
                e -= 1
            e = max(edges[-1] + 1, min(e, rows - (tiles - k)))              # every window keeps at least one row
            edges.append(e)
        edges.append(rows)
        for k in range(tiles):
            out.append([(v, edges[k], edges[k + 1]) for v in views])
    return out


def row_costs(flags, n_views, Wb, Hb, spp, spp_grad, silhouette=True):
    """Predicted work per film-block row from the per-pixel proof flags of a step (csrc/dsdf_proof.h: bit 0 / 1 = every sample misses
    in the primal / gradient pass, 2 / 3 = ... and so does every pixel within +-4, 4 = every sample hits, 5 = deep inside), as a
    (n_views, Hb) float64 tensor on the flags' device.  Weights = wave instructions per 64-sample chunk of the bench scene
    (profiles/valu_model.json: a traced primal chunk ~ 24 wave iterations x 176 + 1454, a proven one ~ 500, a traced chunk of the
    sweep ~ 20 x 759): only their RATIOS matter."""
    import torch
    f = flags[:n_views * Wb * Hb].view(n_views, Hb, Wb)
    empty, empty_g, far, far_g, hit, deep = ((f & b) != 0 for b in (1, 2, 4, 8, 16, 32))
    listed_p = ~far & ~(deep if silhouette else torch.zeros_like(deep))
    proven_p = listed_p & (empty | (hit if silhouette else torch.zeros_like(hit)))
    traced_p = listed_p & ~proven_p
    listed_g = ~far_g
    traced_g = listed_g & ~empty_g
    cp, cg = spp / 64.0, spp_grad / 64.0
    c = cp * (5600.0 * traced_p.sum(2, dtype=torch.float64) + 500.0 * proven_p.sum(2, dtype=torch.float64)) + \
        cg * (15000.0 * traced_g.sum(2, dtype=torch.float64) + 500.0 * (listed_g & ~traced_g).sum(2, dtype=torch.float64))
    return c


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


def _call_loss_grad(loss_grad, images, views):
    """loss_grad(images) for callables written against ALL views, loss_grad(images, views) for those that take the view
    indices of the images they are handed."""
    import inspect
    try:
        two = len(inspect.signature(loss_grad).parameters) >= 2
    except (TypeError, ValueError):
        two = False
    return loss_grad(images, views) if two else loss_grad(images)


def render_step(ops, n_views, W, H, rank, world, loss_grad, grad_grid, group=None, extra_grads=(), gather_images=True,
                async_reduce=False, force_reduce=False, partition=None):
    """One differentiable render of `n_views` views shared by `world` ranks, split by views and -- when world does not
    divide them -- by pixel tiles (work_partition).  `ops` supplies the four film-level operators of the renderer for a
    list of views and a row window (dsdf.render_film / develop / GradSweep on a GPU; the oracle's in the CPU tests):
        ops.film(views, rows)                      -> primal film blocks (len(views), H+4, W+4, C) of the window's samples
        ops.develop(film_total)                    -> images (n, H, W, 3)
        ops.sweep(views, rows)                     -> (gradient-pass film blocks, handle)
        ops.backward(handle, film_total, grad_image, grad_grid)   accumulates dL/dsdf of the window's samples
      optional: ops.begin_sweeps() / ops.end_sweeps() bracket the sweep calls (the HIP operators run them on a side stream
      beside the primal films: they do not depend on the image gradient).
    Exchange steps: when views are SPLIT into row windows, the sum of the primal films and the sum of the gradient-pass
    films (2 x n_views x 2 MiB at 512^2); whole views need neither (a view's film lives on one rank).  Always: ONE sum of
    dL/dsdf and of `extra_grads` (e.g. the albedo gradient of sdf_direct_reparam) in a single bucket -- with
    async_reduce=True it is issued non-blocking and its work handle returned, so that it overlaps whatever the caller
    enqueues next (wait before reading the gradients).
    loss_grad(images[, views]) -> dL/d(images).  Returns the images: of all views on every rank (whole views:
    all-gathered when gather_images, else only this rank's rows are filled), or (images, work) with async_reduce."""
    import torch
    import torch.distributed as dist
    # (partition: the units of ALL ranks, e.g. cost_partition(...) of the previous step's row costs; default: the uniform deal)
    if partition is not None and len(partition) != world:
        raise ValueError("partition must list the units of every rank")
    units = (partition if partition is not None else work_partition(n_views, H + 4, world))[rank]
    wins = rank_windows(units)
    split = world // math.gcd(n_views, world) > 1
    mine = sorted({v for v, _, _ in units})
    # ---- per row window: the gradient-pass sweep (independent of the primal images; HIP: a side stream) and the primal film of
    # the same views, bracketed so that the two passes share ONE per-pixel proof (ops.begin_group / end_group: the sweep writes the
    # flags, the film waits for them -- which also puts the film's small kernels in front of the sweep's persistent workers; without
    # the bracket they were found waiting 2.7 ms for a wave slot, profiles/r05_ab.md)
    if hasattr(ops, 'begin_sweeps'):
        ops.begin_sweeps()
    swept, film = [], None
    for rows, views in wins:
        if hasattr(ops, 'begin_group'):
            ops.begin_group(views)
        try:
            fg, h = ops.sweep(views, rows)
            swept.append((views, fg, h))                  # (the films are summed after join_sweeps(): they may still be in flight)
            f = ops.film(views, rows)
        finally:
            if hasattr(ops, 'end_group'):
                ops.end_group()
        if film is None:
            film = torch.zeros((n_views,) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
        film[views] += f
    if hasattr(ops, 'end_sweeps'):
        ops.end_sweeps()
    if film is None:
        film = ops.empty_film(n_views)
    if split:
        all_reduce_sum(film, group)
        images = ops.develop(film)
        grad_images = _call_loss_grad(loss_grad, images, list(range(n_views)))
    else:
        images = torch.zeros((n_views, H, W, 3), dtype=film.dtype, device=film.device)
        if mine:
            images[mine] = ops.develop(film[mine].contiguous())
        if gather_images and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            all_reduce_sum(images, group)                  # every view is non-zero on exactly one rank
            grad_images = _call_loss_grad(loss_grad, images, list(range(n_views)))
        else:
            gi = _call_loss_grad(loss_grad, images[mine] if _wants_views(loss_grad) else images, mine)
            grad_images = torch.zeros_like(images)
            if mine:
                grad_images[mine] = gi if gi.shape[0] == len(mine) else gi[mine]
    if hasattr(ops, 'join_sweeps'):
        ops.join_sweeps()
    film_g, handles = torch.zeros_like(film), []
    for views, f, h in swept:
        film_g[views] += f
        handles.append((views, h))
    if split:
        all_reduce_sum(film_g, group)
    for views, h in handles:
        ops.backward(h, film_g[views].contiguous(), grad_images[views].contiguous(), grad_grid)
    work = all_reduce_gradients([grad_grid] + list(extra_grads), group, async_op=async_reduce, force=force_reduce)
    return (images, work) if async_reduce else images


class CostTracker:
    """Row costs of all views as the ranks measure them, one step behind: after a step every rank writes the rows it has proofs
    for (ops.row_costs()) and how long its step took, ONE all-reduce (25 KB at 12 x 516) gives every rank the same matrix and the
    same vector of rank times, and partition() is cost_partition() of it -- the uniform deal until the first measurement.

    TIME FEEDBACK.  The proxy of row_costs() counts traced and proven chunks; what a rank's step takes also contains the latency of
    its longest rays (the tail kernels: ~4 ms whatever the shard, profiles/r06_scaling.md), so equal predicted cost left the emulated
    ranks of the bench scene 1.23 x (N = 4) / 1.55 x (N = 8) apart.  Every update therefore multiplies the rows a rank rendered by
    (its time / the mean time) ^ feedback (1.0; clipped to [0.5, 2] per step): rows in slow shards get heavier, the next cut gives that shard
    fewer of them.  Static geometry converges in two or three steps; a moving one is followed with one step of delay."""

    def __init__(self, n_views, rows, world, device=None, feedback=1.0):
        import numpy as np
        self.n_views, self.rows, self.world = n_views, rows, world
        self.device = device                        # where the collective's buffer lives (RCCL reduces device tensors)
        self.feedback = feedback
        self.cost = None
        self.corr = np.ones((n_views, rows))
        self.times = None
        self._last = None                           # the partition the last step was rendered with
        self._np = np

    def partition(self):
        if self.cost is None:
            self._last = work_partition(self.n_views, self.rows, self.world)
        else:
            self._last = cost_partition(self.cost * self.corr, self.world)
        return self._last

    def update(self, views, costs, rank=0, step_ms=None, group=None):
        """views: the views this rank has row costs for; costs: (len(views), rows) tensor; step_ms: how long this rank's last step
        took (None / 0: no time feedback).  Collective (every rank calls it)."""
        import torch
        import torch.distributed as dist
        np = self._np
        dev = self.device if self.device is not None else (costs.device if hasattr(costs, 'device') else 'cpu')
        m = torch.zeros(2 * self.n_views * self.rows + self.world, dtype=torch.float64, device=dev)
        planes = m[:2 * self.n_views * self.rows].view(2, self.n_views, self.rows)
        if len(views) and costs is not None:
            planes[0, list(views)] = torch.as_tensor(costs, dtype=torch.float64, device=dev)
            planes[1, list(views)] = 1.0
        if step_ms:
            m[2 * self.n_views * self.rows + rank] = float(step_ms)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
        m = m.cpu().numpy()
        pl = m[:2 * self.n_views * self.rows].reshape(2, self.n_views, self.rows)
        seen = pl[1] > 0
        new = np.where(seen, pl[0] / np.maximum(pl[1], 1.0), 0.0)
        if self.cost is None:
            fill = new[seen].mean() if seen.any() else 1.0
            self.cost = np.where(seen, new, fill)
        else:
            self.cost = np.where(seen, new, self.cost)
        self.apply_times(m[2 * self.n_views * self.rows:])
        return self.cost

    def apply_times(self, times):
        """The feedback step on its own (the single-GPU emulation of bench.py feeds it the emulated ranks' times)."""
        np = self._np
        t = np.asarray(times, np.float64)
        self.times = t
        if self.feedback and self._last is not None and len(t) == self.world and (t > 0).all():
            mean = t.mean()
            for r, units in enumerate(self._last):
                f = float(np.clip((t[r] / mean) ** self.feedback, 0.5, 2.0))
                for v, r0, r1 in units:
                    self.corr[v, r0:r1] *= f
            self.corr /= self.corr.mean()


def _wants_views(loss_grad):
    import inspect
    try:
        return len(inspect.signature(loss_grad).parameters) >= 2
    except (TypeError, ValueError):
        return False


class HipOps:
    """The film-level operators of render_step on the HIP path (dsdf.render_film / develop / GradSweep).  Sweeps run on a
    side stream beside the primal films (two-stream schedule of dsdf.render_step); their workspaces (the backward queue
    lives in them between sweep and backward) are kept and re-used by the next step's sweeps."""

    def __init__(self, grid, sensors, spp, spp_grad, seeds, seeds_grad, integrator=0, two_streams=True, **kw):
        import torch
        from . import renderer
        self.r = renderer
        self.grid, self.sensors, self.spp, self.spp_grad = grid, list(sensors), int(spp), int(spp_grad)
        self.integrator, self.kw = integrator, kw
        self.kw_film = {k: v for k, v in kw.items() if k != 'grad_albedo'}
        self.W, self.H = self.sensors[0].film_size()
        self.set_seeds(seeds, seeds_grad)
        self._ws, self._next_ws = [], 0
        self._side = torch.cuda.Stream(grid.device, priority=-1) if two_streams else None
        self._in_sweeps = False

    def set_seeds(self, seeds, seeds_grad):
        self.seeds, self.seeds_grad = list(seeds), list(seeds_grad)

    def empty_film(self, n):
        return self.r.new_film(n, self.W, self.H, self.integrator, self.grid.device)

    def film(self, views, rows):
        f = self.empty_film(len(views))
        return self.r.render_film(self.grid, [self.sensors[v] for v in views], self.spp, f, rows,
                                  seeds=[self.seeds[v] for v in views], integrator=self.integrator, **self.kw_film)

    def develop(self, film):
        return self.r.develop(film, self.W, self.H, self.integrator)

    # ---- one per-pixel proof for the sweep and the film of a window group (dsdf_share_pixel_skip, as dsdf.step_begin does)
    def begin_group(self, views):
        import torch
        from . import _lib
        dev = self.grid.device
        n = len(views) * (self.W + 4) * (self.H + 4)
        self._group_views = list(views)
        # ONE flags buffer PER window group of a step: group k + 1's sweep writes its flags on the side stream while group
        # k's film kernels may still read theirs on the main stream -- nothing orders that writer after that reader
        bufs = self.__dict__.setdefault('_flag_bufs', [])
        gi = getattr(self, '_group_i', 0)
        self._group_i = gi + 1
        while len(bufs) <= gi:
            bufs.append(None)
        if bufs[gi] is None or bufs[gi].numel() < n:
            bufs[gi] = torch.empty(n, dtype=torch.uint8, device=dev)
        self._flags = bufs[gi]
        self._share_lib = self.grid.lib(self.r._needs_extended(self.kw.get('shading')))
        with torch.cuda.device(dev):
            _lib.check(self._share_lib.dsdf_share_pixel_skip(self.r._ptr(self._flags), self._flags.numel()))

    def end_group(self):
        import torch
        with torch.cuda.device(self.grid.device):
            self._share_lib.dsdf_share_pixel_skip(None, 0)
        self._flag_views = getattr(self, '_group_views', None)

    def row_costs(self):
        """(views, costs): the views of the LAST window group and their predicted work per film-block row (parallel.row_costs of the
        group's proof flags -- a proof covers all rows of its views, whatever the window)."""
        views = getattr(self, '_flag_views', None)
        if not views or getattr(self, '_flags', None) is None:
            return [], None
        from .renderer import DSDF_SILHOUETTE
        sil = self.integrator in (DSDF_SILHOUETTE, 'sdf_silhouette_reparam')
        return list(views), row_costs(self._flags, len(views), self.W + 4, self.H + 4, self.spp, self.spp_grad, sil)

    # ---- sweeps: on the side stream, between begin_sweeps() and end_sweeps(); joined before the backward
    def begin_sweeps(self):
        import torch
        self._next_ws = 0
        self._group_i = 0
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.grid.device))
            self._in_sweeps = True

    def end_sweeps(self):
        self._in_sweeps = False

    def join_sweeps(self):
        import torch
        if self._side is not None:
            torch.cuda.current_stream(self.grid.device).wait_stream(self._side)

    def sweep(self, views, rows):
        import contextlib
        import torch
        ctx = torch.cuda.stream(self._side) if (self._side is not None and self._in_sweeps) else contextlib.nullcontext()
        with ctx:
            lend = self._ws[self._next_ws] if self._next_ws < len(self._ws) else None
            h = self.r.GradSweep(self.grid, [self.sensors[v] for v in views], self.spp_grad, rows,
                                 seeds=[self.seeds_grad[v] for v in views], integrator=self.integrator, workspace=lend, **self.kw)
            if h.ws is not None:
                if self._next_ws < len(self._ws):
                    self._ws[self._next_ws] = h.ws
                else:
                    self._ws.append(h.ws)
            self._next_ws += 1
            film = h.sweep(self.empty_film(len(views)))
        if self._side is not None:
            main = torch.cuda.current_stream(self.grid.device)
            film.record_stream(main)
            h.record_stream(main)
        return film, h

    def backward(self, handle, film_total, grad_image, grad_grid):
        handle.backward(film_total, grad_image, grad_grid)


class GradBucket:
    """Gradient tensors as VIEWS into one persistent flat buffer: the all-reduce of a step then moves that buffer as it is --
    no per-step torch.cat into a fresh bucket and no copy back (round 3 built a new flat tensor every step: 256 MiB at
    BASELINE's configs[4]).  `views[i]` has shapes[i]; zero_() clears all of them with one memset."""

    def __init__(self, shapes, device, dtype=None):
        import math
        dtype = dtype or torch.float32
        sizes = [int(math.prod(s)) for s in shapes]
        self.flat = torch.zeros(sum(sizes), dtype=dtype, device=device)
        self.views, off = [], 0
        for s, n in zip(shapes, sizes):
            self.views.append(self.flat[off:off + n].view(*s))
            off += n

    def zero_(self):
        self.flat.zero_()
        return self

    def all_reduce(self, group=None, async_op=False, force=False):
        return all_reduce_gradients([self.flat], group, async_op=async_op, force=force)


def _as_one_buffer(tensors):
    """The flat tensor that covers `tensors` when they are consecutive contiguous views of ONE storage (a GradBucket), else None."""
    t0 = tensors[0]
    if not all(t.is_contiguous() and t.dtype == t0.dtype and t.device == t0.device for t in tensors):
        return None
    try:
        base = t0.untyped_storage().data_ptr()
        if any(t.untyped_storage().data_ptr() != base for t in tensors):
            return None
    except Exception:
        return None
    off = t0.storage_offset()
    for t in tensors:
        if t.storage_offset() != off:
            return None
        off += t.numel()
    return torch.as_strided(t0, (off - t0.storage_offset(),), (1,), t0.storage_offset())


_BIG = 1 << 20            # elements: a tensor this large is reduced in place on its own (bandwidth-bound either way)
_small_buckets = {}       # (device, dtype, numel) -> persistent flat buffer for the small tensors of a call


def all_reduce_gradients(tensors, group=None, async_op=False, force=False):
    """Sums the given gradient tensors over all ranks in place.  Tensors that are consecutive views of one buffer (GradBucket)
    go out as that buffer -- one collective, no copy.  Otherwise every large tensor (>= 2^20 elements) is reduced in place by
    its own collective and the small ones share ONE persistent bucket (copied in and out; no per-step allocation).  async_op:
    the collectives are issued non-blocking and a handle with .wait() is returned (None when there is nothing to reduce).
    force: issue them even in a world of one rank (exercises the transport on a single GPU)."""
    import torch.distributed as dist
    tensors = list(tensors)
    if not tensors or not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return None if async_op else tensors
    one = _as_one_buffer(tensors)
    if one is not None:
        w = dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return _BucketWork([w], None) if async_op else tensors
    works, small = [], []
    for t in tensors:
        if t.numel() >= _BIG and t.is_contiguous():
            works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
        else:
            small.append(t)
    unpacks = []
    by_dtype = {}
    for t in small:                                       # (one bucket per dtype: nothing is cast)
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, group_ts in by_dtype.items():
        n = sum(t.numel() for t in group_ts)
        if async_op:
            # a non-blocking reduce owns its bucket until wait(): overlapping calls must not share one
            flat = torch.empty(n, dtype=dtype, device=group_ts[0].device)
        else:
            key = (group_ts[0].device, dtype, n)
            flat = _small_buckets.get(key)
            if flat is None:
                flat = _small_buckets[key] = torch.empty(n, dtype=dtype, device=group_ts[0].device)
        off = 0
        for t in group_ts:
            flat[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op))

        def _unpack(flat=flat, group_ts=group_ts):
            o = 0
            for t in group_ts:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()
        unpacks.append(_unpack)
    unpack = (lambda: [u() for u in unpacks]) if unpacks else None
    if not async_op:
        if unpack:
            unpack()
        return tensors
    return _BucketWork(works, unpack)


class _BucketWork:
    def __init__(self, works, unpack):
        self.works, self.unpack = works, unpack

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()
        if self.unpack:
            self.unpack()


def broadcast_parameters(tensors, src=0, group=None):
    """Makes every replica start from rank `src`'s parameters (grid, textures)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
    return tensors
