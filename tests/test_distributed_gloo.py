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
    w = torch.full((3,), float(rank))
    parallel.broadcast_parameters([w], src=1)
    assert float(w[0]) == 1.0
    dist.barrier()
    dist.destroy_process_group()


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
