"""N>1 host logic on CPU (gloo, world_size 2): view sharding + the single bucketed
gradient all-reduce reproduce the single-process result.  The per-view gradients
come from the oracle here (the HIP renderer needs a GPU); what is under test is
dsdf/parallel.py."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    for p in (os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    import sdf_oracle as O
    from dsdf import parallel
    n_views, R, W, H, spp = 3, 16, 8, 8, 4
    grid = O.sphere_grid(R)
    gen = torch.Generator().manual_seed(5)
    offs = [torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float64) for _ in range(n_views)]
    gis = [torch.randn(H, W, 3, generator=gen, dtype=torch.float64) for _ in range(n_views)]
    origins = O.regular_camera_origins(n_views)

    def view_grad(i):
        return O.render_backward(O.Grid3d(grid), O.Camera(origins[i]), W, H, spp, offs[i], gis[i], O.SILHOUETTE)

    mine = parallel.view_shard(n_views, rank, world)
    g = torch.zeros(R, R, R, dtype=torch.float64)
    tex = torch.full((4, 4), float(rank + 1), dtype=torch.float64)        # a second gradient buffer in the same bucket
    for i in mine:
        g += view_grad(i)
    parallel.all_reduce_gradients([g, tex])
    if rank == 0:
        full = sum(view_grad(i) for i in range(n_views))
        out.put((float((g - full).abs().max()), float(full.abs().max()), float(tex[0, 0]), mine))
    # GradBucket: the tensors are views of ONE persistent buffer -> reduced as that buffer (no cat, no copy back); and the
    # split policy of a plain list: a large tensor in place by its own collective, the small ones through a cached bucket
    bucket = parallel.GradBucket([(R, R, R), (4, 4, 3)], 'cpu', torch.float64)
    bucket.views[0].copy_(torch.full((R, R, R), float(rank + 1), dtype=torch.float64))
    bucket.views[1].fill_(10.0 * (rank + 1))
    flat_ptr = bucket.flat.data_ptr()
    assert parallel._as_one_buffer(bucket.views) is not None and parallel._as_one_buffer([g, tex]) is None
    work = parallel.all_reduce_gradients(bucket.views, async_op=True)
    work.wait()
    tot = sum(range(1, world + 1))
    assert bucket.flat.data_ptr() == flat_ptr and float(bucket.views[0][1, 2, 3]) == tot and float(bucket.views[1][0, 0, 0]) == 10.0 * tot
    big = torch.full(((1 << 20) + 5,), float(rank + 1))
    small_a, small_b = torch.full((7,), float(rank)), torch.full((2, 3), 2.0 * rank)
    n_cached = len(parallel._small_buckets)
    for _ in range(2):                                            # twice: the second call re-uses the cached small bucket
        big.fill_(float(rank + 1)); small_a.fill_(float(rank)); small_b.fill_(2.0 * rank)
        parallel.all_reduce_gradients([big, small_a, small_b])
        assert float(big[-1]) == tot and float(small_a[0]) == tot - world and float(small_b[1, 2]) == 2.0 * (tot - world)
    assert len(parallel._small_buckets) == n_cached + 1
    w = torch.full((3,), float(rank))
    parallel.broadcast_parameters([w], src=1)
    assert float(w[0]) == 1.0
    dist.barrier()
    dist.destroy_process_group()


def _tile_worker(rank, world, port, out, n_views, variant=False, cost_aware=False):
    """render_step (dsdf/parallel.py) with the ORACLE's film-level operators: the split by views and by pixel tiles, the two
    film sums and the gradient sum reproduce the single-process image and gradient."""
    for p in (os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    import sdf_oracle as O
    from dsdf import parallel
    R, W, H, spp = 16, 10, 8, 4
    grid = O.sphere_grid(R)
    gen = torch.Generator().manual_seed(7)
    offs = [torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float64) for _ in range(n_views)]
    tgt = torch.rand(n_views, H, W, 3, generator=gen, dtype=torch.float64)
    cams = [O.Camera(o) for o in O.regular_camera_origins(max(n_views, 3))[:n_views]]
    integ = O.SIMPLE_SHADING

    class Ops:                                     # the four film-level operators, from the oracle (C = 4: r, g, b, weight)
        def film(self, views, rows):
            with torch.no_grad():
                return torch.stack([O.render(O.Grid3d(grid), cams[v], W, H, spp, offs[v], integ, rows=rows, return_block=True) for v in views])

        def empty_film(self, n):
            return torch.zeros(n, H + 4, W + 4, 4, dtype=torch.float64)

        def develop(self, film):
            return torch.stack([O.develop(f.reshape(-1), W, H) for f in film])

        def sweep(self, views, rows):
            leaf = grid.clone().requires_grad_(True)
            blocks = torch.stack([O.render(O.Grid3d(leaf), cams[v], W, H, spp, offs[v], integ, rows=rows, return_block=True) for v in views])
            return blocks.detach(), (leaf, blocks)

        def backward(self, handle, film_total, grad_image, grad_grid):
            leaf, mine = handle
            # the window's samples against the film of ALL samples: total = mine (attached) + the others' (constants)
            total = mine + (film_total - mine.detach())
            img = torch.stack([O.develop(f.reshape(-1), W, H) for f in total])
            if img.requires_grad:
                (img * grad_image).sum().backward()
                grad_grid += leaf.grad

    g = torch.zeros(R, R, R, dtype=torch.float64)
    part = None
    if cost_aware:
        # the COST-AWARE split (parallel.cost_partition) of a skewed cost matrix: uneven windows, views dealt by cost -- and through the
        # CostTracker's collective update, so that both ranks derive the partition from the same all-reduced matrix
        import numpy as np
        full = np.outer(np.linspace(1.0, 3.0, n_views), np.linspace(0.1, 4.0, H + 4) ** 2)
        tr = parallel.CostTracker(n_views, H + 4, world)
        assert tr.partition() == parallel.work_partition(n_views, H + 4, world)
        mine0 = sorted({v for v, _, _ in tr.partition()[rank]})
        tr.update(mine0, torch.from_numpy(full[mine0]))
        part = tr.partition()
        assert part == parallel.cost_partition(full, world)
    if cost_aware:
        images = parallel.render_step(Ops(), n_views, W, H, rank, world, lambda im: 2.0 * (im - tgt), g, partition=part)
    elif variant:
        # whole views without the image gather, loss gradient per owned view, a second gradient tensor in the bucket, and the
        # non-blocking reduce
        extra = torch.full((5,), float(rank + 1), dtype=torch.float64)
        images, work = parallel.render_step(Ops(), n_views, W, H, rank, world, lambda im, views: 2.0 * (im - tgt[views]), g,
                                            extra_grads=[extra], gather_images=False, async_reduce=True)
        assert work is not None
        work.wait()
        assert float(extra[0]) == 3.0
        mine = [v for v, _, _ in parallel.work_partition(n_views, H + 4, world)[rank]]
        others = [v for v in range(n_views) if v not in mine]
        assert float(images[others].abs().max()) == 0.0 if others else True
        full = images.clone()
        dist.all_reduce(full)
        images = full
    else:
        images = parallel.render_step(Ops(), n_views, W, H, rank, world, lambda im: 2.0 * (im - tgt), g)
    if rank == 0:
        leaf = grid.clone().requires_grad_(True)
        ref = torch.stack([O.render(O.Grid3d(leaf), cams[v], W, H, spp, offs[v], integ) for v in range(n_views)])
        ((ref - tgt) ** 2).sum().backward()
        out.put((float((images - ref.detach()).abs().max()), float((g - leaf.grad).abs().max()), float(leaf.grad.abs().max()),
                 part[0] if part is not None else parallel.work_partition(n_views, H + 4, world)[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_views', [1, 3, 2])
def test_tile_split_equals_single_process(n_views):
    """world 2: one view -> two row windows of the same view; three views -> 6 half-views; two views -> whole views."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * n_views
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, out, n_views)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    err_img, err_g, mag, units = out.get(timeout=10)
    assert mag > 0 and err_img < 1e-12 and err_g <= 1e-10 * max(mag, 1.0), (err_img, err_g, mag)
    if n_views == 1:
        assert units == [(0, 0, 6)]                       # rank 0 renders rows [0, 6) of the 12-row film block of view 0
    if n_views == 2:
        assert units == [(0, 0, 12)]


@pytest.mark.parametrize('n_views', [1, 3, 4])
def test_cost_aware_split_equals_single_process(n_views):
    """world 2 with parallel.cost_partition of a skewed cost matrix (uneven row windows for 1 and 3 views, whole views dealt by cost
    for 4): image and gradient equal the single-process result, as with the uniform deal."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 300 + 7 * n_views
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, out, n_views, False, True)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    err_img, err_g, mag, units = out.get(timeout=10)
    assert mag > 0 and err_img < 1e-12 and err_g <= 1e-10 * max(mag, 1.0), (err_img, err_g, mag)
    if n_views in (1, 3):
        assert all(r0 == 0 and r1 > 6 for _, r0, r1 in units), units        # the rows are cheap at the top: rank 0's window is the larger one
    else:
        assert sorted(v for v, _, _ in units) == [0, 3] and all((r0, r1) == (0, 12) for _, r0, r1 in units), units   # LPT: {3, 0} / {2, 1}


def test_cost_partition():
    sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'))
    import numpy as np
    from dsdf import parallel
    rng = np.random.default_rng(1)
    for n in (1, 3, 12, 48):
        for world in (1, 2, 4, 5, 8):
            if n * 516 < world:
                continue
            cost = rng.random((n, 516)) * np.linspace(0.2, 2.0, 516)[None] * (0.5 + rng.random((n, 1)))
            P, U = parallel.cost_partition(cost, world), parallel.work_partition(n, 516, world)
            assert len(P) == world and [len(u) for u in P] == [len(u) for u in U]        # the same shape as the uniform deal
            cover = {}
            for u in P:
                assert len({(r0, r1) for _, r0, r1 in u}) == 1                            # ONE row window per rank
                for v, r0, r1 in u:
                    cover.setdefault(v, []).append((r0, r1))
            for v in range(n):                                                            # every view: windows tile [0, 516) exactly once
                w = sorted(cover[v])
                assert w[0][0] == 0 and w[-1][1] == 516 and all(a[1] == b[0] and a[0] < a[1] for a, b in zip(w, w[1:] + [(516, 517)]))
            load = lambda part: [sum(cost[v, a:b].sum() for v, a, b in u) for u in part]
            lp, lu = load(P), load(U)
            assert max(lp) / min(lp) <= max(1.05 * max(lu) / min(lu), 1.10), (n, world, lp, lu)   # as good as the uniform deal, or within 10 %
    cost = np.outer(np.ones(12), np.linspace(0.0, 1.0, 516))                              # all work in the lower rows
    l8 = [sum(cost[v, a:b].sum() for v, a, b in u) for u in parallel.cost_partition(cost, 8)]
    assert max(l8) / min(l8) < 1.02
    assert parallel.cost_partition(np.zeros((3, 12)), 2) == parallel.work_partition(3, 12, 2)   # no information: the even cut


def test_whole_views_async_bucket_no_gather():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 91
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, out, 4, True)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    err_img, err_g, mag, units = out.get(timeout=10)
    assert mag > 0 and err_img < 1e-12 and err_g <= 1e-10 * max(mag, 1.0), (err_img, err_g, mag)
    assert len(units) == 2


def test_work_partition():
    sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'))
    from dsdf import parallel
    for n in (1, 3, 12, 48):
        for world in (1, 2, 4, 5, 8):
            P = parallel.work_partition(n, 516, world)
            assert len(P) == world and len({len(u) for u in P}) == 1            # perfectly balanced unit counts
            cover = {}
            for u in P:
                for v, r0, r1 in u:
                    cover.setdefault(v, []).append((r0, r1))
            for v in range(n):                                                   # every view: windows tile [0, 516) exactly once
                w = sorted(cover[v])
                assert w[0][0] == 0 and w[-1][1] == 516 and all(a[1] == b[0] for a, b in zip(w, w[1:]))
    assert [len(u) for u in parallel.work_partition(12, 516, 8)] == [3] * 8     # 24 half-views instead of 2/1 whole views


def test_view_shard_partition():
    sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'))
    from dsdf import parallel
    for n in (1, 5, 12, 48):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                seen += parallel.view_shard(n, r, world)
            assert seen == list(range(n))
            sizes = [len(parallel.view_shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.strided_view_shard([0, 2, 4, 6, 8, 10], 1, 2) == [2, 6, 10]
    with pytest.raises(ValueError):
        parallel.view_shard(4, 2, 2)


def test_sharded_gradient_equals_single_process():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    err, mag, tex, mine = out.get(timeout=10)
    assert mag > 0 and err <= 1e-12 * max(mag, 1.0)
    assert tex == 3.0 and mine == [0, 1]
