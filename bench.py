#!/usr/bin/env python3
"""bench.py -- differentiable renders/s of the hot path on MI355X.

One "step" = one differentiable render in the reference's sense (python/shape_opt.py:77-83 per view): for each of the
12 sensors a primal render at spp_primal (no AD) and a gradient pass at spp_grad that accumulates dL/dsdf, on a 256^3
SDF at 512^2 (BASELINE.json configs[2], the configuration the metric is quoted on; it fits one GPU).  Default spp = the
reference's 256 / 64 (python/configs.py:16,19).  Inputs are synthetic, seeded and resident in HBM before the timed region.

Multi-GPU (launched by torch.distributed.run, one rank per GPU).  Default `--scaling strong`: the metric's 12 views are
partitioned over the ranks (dsdf.parallel.view_shard; with more ranks than a divisor of 12 the shards differ by one view),
every rank renders its views, then the per-voxel gradient grid is summed with ONE RCCL all-reduce (the path's only
exchange step); value = steps / time means the same at every N.  `--scaling weak` (12 views PER rank, a 12*N ring) is
kept under a different metric string.

The JSON line carries, besides the contract's fields:
  roofline     -- VALU-issue roofline of the dominant kernel (the primal k_render_pass): the path is not HBM-bound (taps
                  are LDS/L1-resident: measured HBM traffic is ~1 % of the algorithmic tap bytes), it issues vector ALU
                  instructions.  achieved = wave-level VALU instructions per launch / launch time; peak = 1024 SIMDs x
                  2.4 GHz / 2 clk per wave64 instruction (MI355X_MICROARCH.md: SIMD-32, `v_fma_f32` 2 cyc).  The
                  instruction count is LIVE: lock-step wave iterations counted by the kernel itself (stats[7]) x the
                  per-iteration VALU instruction counts of the shipped ISA, calibrated against SQ_INSTS_VALU of the
                  committed PMC pass (profiles/, tag in `calibration`).  The HBM-equivalent algorithmic-bytes figure of
                  SURVEY 8(d) is kept as a secondary field (`hbm_equivalent`); `traffic` is null here (PMC counters cannot
                  be read in-process; the measured bytes live in profiles/).
  low_spp      -- the same step at 4/1 spp (the sampling rate at which north_star's >= 50 renders/s target and its 60 %
                  roofline coincide), timed in the same process after the headline region.
  cpu_baseline -- oracle/dsdf_oracle.c (fp32 build) on the host cores, bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'))

import torch

# VALU instructions per lock-step wave iteration / per wave, from the ISA of the shipped kernels
# (hipcc -S, instruction histogram of the loop bodies; cross-checked against SQ_INSTS_VALU: profiles/r02_sq.json)
VALU_MODEL = {
    'calibration': 'profiles/r02_sq.json',
    'primal': {'per_wave_step': 209.0, 'per_traced_wave': 600.0, 'per_wave': 430.0},
}
VALU_PEAK = 1024 * 2.4e9 / 2.0        # wave64 VALU instructions / s: 256 CUs x 4 SIMD-32, 2 clk per instruction


def synth_grid(res, device, n=32, seed=0):
    """Seeded union of spheres/tori clipped by the box SDF (SURVEY 8d synthetic inputs;
    same recipe as variables.py:161-166, 185-187 for the box clip).  Built on the device."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lin = torch.linspace(0, 1, res, device=device)
    z, y, x = torch.meshgrid(lin, lin, lin, indexing='ij')
    sd = torch.full((res, res, res), 1e9, device=device)
    for k in range(n):
        c = rng.uniform(0.3, 0.7, 3)
        if k % 2 == 0:
            r = rng.uniform(0.05, 0.12)
            sd = torch.minimum(sd, torch.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - r)
        else:
            R, r = rng.uniform(0.08, 0.16), rng.uniform(0.015, 0.035)
            q = [x - c[0], y - c[1], z - c[2]]
            ax = k % 3
            o = [a for a in range(3) if a != ax]
            ring = torch.sqrt(q[o[0]] ** 2 + q[o[1]] ** 2) - R
            sd = torch.minimum(sd, torch.sqrt(ring ** 2 + q[ax] ** 2) - r)
    lin2 = torch.linspace(-0.5, 0.5, res, device=device)
    z, y, x = torch.meshgrid(lin2, lin2, lin2, indexing='ij')
    q = torch.stack([x.abs(), y.abs(), z.abs()], -1) - 0.49
    box = torch.linalg.norm(q.clamp(min=0), dim=-1) + q.max(-1).values.clamp(max=0) - 0.01
    return torch.maximum(sd, box).contiguous()


def cpu_baseline(args, target_seconds=15.0):
    """Oracle ('port': oracle/dsdf_oracle.c, plain C + OpenMP, fp32 build) timed on the host cores on a
    bounded sample of the same workload: the same 256^3 grid, ONE sensor of the ring, the full
    512^2 film, at reduced spp chosen so that the sample takes ~15 s (probe at 4/1 spp first);
    scaled linearly in samples-per-pixel to the 256/64 spp, 12-view job."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import c_oracle
    lib = c_oracle.load()
    g = synth_grid(args.res, 'cpu').numpy()
    import dsdf
    s = dsdf.get_regular_cameras(args.views, resx=args.img, resy=args.img)[0]
    left, up, d = s.frame()
    cam = np.concatenate([s.origin, left, up, d, [math.tan(math.radians(s.fov) * 0.5), 0, 0, 0]]).astype(np.float32)
    W = H = args.img
    rng = np.random.default_rng(0)
    gi = (rng.standard_normal((H, W, 3)) / (H * W * 3)).astype(np.float32)
    integ = 0 if 'silhouette' in args.integrator else 1
    direct = args.integrator == 'sdf_direct_reparam'
    albedo = (rng.random((args.res, args.res, args.res, 3), dtype=np.float32) * 0.6 + 0.2) if direct else None

    def run(spp_p, spp_g):
        op = rng.random(((W + 4) * (H + 4) * spp_p, 2), dtype=np.float32)
        og = rng.random(((W + 4) * (H + 4) * spp_g, 2), dtype=np.float32)
        t0 = time.time()
        if direct:
            c_oracle.render_direct(lib, g, cam, W, H, spp_p, op, rng.random(op.shape, dtype=np.float32), albedo)
            t1 = time.time()
            c_oracle.render_direct_backward(lib, g, cam, W, H, spp_g, og, rng.random(og.shape, dtype=np.float32), albedo, gi)
        else:
            c_oracle.render(lib, g, cam, W, H, spp_p, op, integ)
            t1 = time.time()
            c_oracle.render_backward(lib, g, cam, W, H, spp_g, og, gi, integ)
        return t1 - t0, time.time() - t1

    tp, tg = run(4, 1)
    k = int(max(1, min(args.spp_grad, target_seconds / max(tp + tg, 1e-3))))
    spp_g, spp_p = k, 4 * k
    if k > 1:
        tp, tg = run(spp_p, spp_g)
    else:
        spp_p, spp_g = 4, 1
    t_view = tp * args.spp_primal / spp_p + tg * args.spp_grad / spp_g
    return {"value": 1.0 / (t_view * args.views), "unit": "renders/s", "cores": lib.o_num_threads(), "kind": "port",
            "sample": f"oracle/dsdf_oracle.c (C + OpenMP, fp32), 1 of {args.views} sensors, {W}x{H} film, spp {spp_p}/{spp_g}, "
                      f"{args.res}^3 grid; {tp + tg:.1f}s measured, scaled linearly in spp and views"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--img', type=int, default=512)
    ap.add_argument('--views', type=int, default=12)
    ap.add_argument('--spp-primal', type=int, default=256)
    ap.add_argument('--spp-grad', type=int, default=64)
    ap.add_argument('--integrator', default='sdf_silhouette_reparam')
    ap.add_argument('--scaling', choices=('strong', 'weak'), default='strong')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-low-spp', action='store_true')
    ap.add_argument('--overlap', type=int, default=1, help='1: primal pass and gradient sweep on two HIP streams (dsdf.render_step)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    import dsdf
    from dsdf import parallel
    dsdf.load()
    # (dry run of the multi-process path on a box with fewer GPUs than ranks: BENCH_SHARE_GPU=1 maps the ranks onto the
    # GPUs that exist, BENCH_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on one device -- never a measurement)
    share = os.environ.get('BENCH_SHARE_GPU') == '1'
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    # (BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank, e.g. under `torchrun --nproc-per-node 1`)
    if world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    data = synth_grid(args.res, dev)
    grid = dsdf.SdfGrid(data)
    target = dsdf.SdfGrid(synth_grid(args.res, dev, seed=1))
    tiled = False
    if args.scaling == 'strong':
        # the metric's 12 views, partitioned: every rank renders its shard of the SAME ring (round-robin, so that all
        # ranks see equally expensive views); the job is the same at every N.  When N does not divide the views (12 on 8
        # GPUs) the views are cut into pixel-row windows (parallel.work_partition: 24 half-views, 3 per rank) and the film
        # blocks are summed across ranks before develop / before the backward (parallel.render_step)
        ring = dsdf.get_regular_cameras(args.views, resx=args.img, resy=args.img)
        tiled = args.views % world != 0
        mine = list(range(args.views)) if tiled else parallel.strided_view_shard(list(range(args.views)), rank, world)
    else:
        # weak scaling: a ring of views*world sensors, `views` per rank
        ring = dsdf.get_regular_cameras(args.views * world, resx=args.img, resy=args.img)
        mine = list(range(args.views * world))[rank::world]
    sensors = [ring[i] for i in mine]
    nv = len(sensors)
    grad = torch.zeros_like(data)
    # BASELINE.json C5 (--integrator sdf_direct_reparam): a 3-channel albedo volume of the grid's resolution is optimised too
    shade, shade_g = {}, {}
    if args.integrator == 'sdf_direct_reparam':
        albedo = torch.rand(args.res, args.res, args.res, 3, device=dev) * 0.6 + 0.2
        shade = {'shading': dsdf.Shading(albedo, 1.0, hide_emitters=False)}
        shade_g = dict(shade, grad_albedo=torch.zeros_like(albedo))
    # target images (outside the timed region) -> L1 image gradient sign(img - target)/(H*W*3)
    tgt = torch.cat([dsdf.render_forward(target, s, 64, seeds=[1000 + i]) for i, s in zip(mine, sensors)]) if nv else None
    scale = 1.0 / (args.img * args.img * 3)

    ev = lambda: torch.cuda.Event(enable_timing=True)

    def make_step(spp_p, spp_g, prim_ms, grad_ms):
        def step(it, timed):
            # one launch traces all views of this rank's shard (primal), one launch the gradient pass
            grad.zero_()
            if tiled:
                seeds = [(it * args.views + i) * 2 for i in range(args.views)]
                ops = parallel.HipOps(grid, ring, spp_p, spp_g, seeds, [s + 1 for s in seeds], args.integrator)
                parallel.render_step(ops, args.views, args.img, args.img, rank, world, lambda im: torch.sign(im - tgt) * scale, grad)
                return
            if nv:
                seeds = [(it * args.views + i) * 2 for i in mine]
                e0, e1, e2 = ev(), ev(), ev()
                if args.overlap:
                    # primal render and the forward sweep of the gradient pass on two HIP streams (dsdf.render_step)
                    e0.record()
                    dsdf.render_step(grid, sensors, spp_p, spp_g, lambda im: torch.sign(im - tgt) * scale, grad, seeds,
                                     [s + 1 for s in seeds], integrator=args.integrator, shading=shade.get('shading'),
                                     grad_albedo=shade_g.get('grad_albedo'))
                    e1.record(); e2.record()
                else:
                    e0.record()
                    img = dsdf.render_forward(grid, sensors, spp_p, seeds=seeds, integrator=args.integrator, **shade)
                    e1.record()
                    gi = torch.sign(img - tgt) * scale
                    dsdf.render_backward(grid, sensors, spp_g, gi, grad_grid=grad, seeds=[s + 1 for s in seeds],
                                         integrator=args.integrator, **shade_g)
                    e2.record()
                if timed:
                    prim_ms.append((e0, e1)); grad_ms.append((e1, e2))
            if dist is not None:
                parallel.all_reduce_gradients([grad] + ([shade_g['grad_albedo']] if shade_g else []))
        return step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(step, warmup, steps):
        for w in range(warmup):
            step(w, False)
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k, True)
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        return elapsed

    prim_ms, grad_ms = [], []
    elapsed = timed_run(make_step(args.spp_primal, args.spp_grad, prim_ms, grad_ms), args.warmup, args.steps)
    # dL/dsdf of the last timed step (summed over ranks): must agree at every N up to the order of the float atomics
    grad_l1 = float(grad.double().abs().sum())
    if args.overlap and nv and not tiled:
        # the roofline needs the dominant kernel's OWN launch time: in the timed region above it shares the chip with the
        # gradient sweep of the other stream, so a few launches are timed alone (HIP events, same process, same inputs)
        prim_ms, grad_ms = [], []
        ov, args.overlap = args.overlap, 0
        probe = make_step(args.spp_primal, args.spp_grad, prim_ms, grad_ms)
        probe(args.warmup + args.steps, False)                 # (untimed: the stand-alone calls allocate their own workspaces)
        for k in range(3):
            probe(args.warmup + args.steps + 1 + k, True)
        torch.cuda.synchronize()
        args.overlap = ov

    low = None
    if not args.no_low_spp:
        lp, lg = [], []
        lsteps = max(args.steps, 20)
        lel = timed_run(make_step(4, 1, lp, lg), 2, lsteps)
        if args.overlap:
            lp, lg = [], []
        low = {"value": lsteps / lel if args.scaling == 'strong' else world * lsteps / lel, "unit": "renders/s", "steps": lsteps,
               "ms_per_step": 1e3 * lel / lsteps,
               "config": {"workload": f"{args.res}^3 SDF, {args.views} views x {args.img}^2, {args.integrator}, spp primal/grad 4/1 "
                                      f"(north_star's >= 50 renders/s point, SURVEY F10)"},
               "overlap": bool(args.overlap)}
        if lp:
            low["primal_ms_per_launch"] = sum(a.elapsed_time(b) for a, b in lp) / len(lp)
            low["grad_ms_per_launch"] = sum(a.elapsed_time(b) for a, b in lg) / len(lg)

    # per-launch statistics (untimed; same launch shape as the timed ones)
    out_cfg, roof = {}, None
    if nv and not tiled:
        st_p, st_g = dsdf.new_stats(dev), dsdf.new_stats(dev)
        dsdf.render_forward(grid, sensors, args.spp_primal, seeds=list(range(nv)), integrator=args.integrator, stats=st_p, **shade)
        dsdf.render_backward(grid, sensors, args.spp_grad, torch.ones(nv, args.img, args.img, 3, device=dev) * scale,
                             grad_grid=torch.zeros_like(data), seeds=list(range(50, 50 + nv)), integrator=args.integrator, **shade_g,
                             stats=st_g)
        sp, sg = dsdf.stats_dict(st_p), dsdf.stats_dict(st_g)
        prim = [a.elapsed_time(b) for a, b in prim_ms]
        gradt = [a.elapsed_time(b) for a, b in grad_ms]
        prim_avg = sum(prim) / len(prim)
        # VALU-issue roofline of the primal launch (k_render_pass<false,true,false>): wave-level VALU instructions
        m = VALU_MODEL['primal']
        total_lanes = nv * (args.img + 4) ** 2 * args.spp_primal
        waves = sp['lanes'] / 64.0                        # generated samples (pixels that are not proven far), in waves
        traced_waves = sp['bbox_lanes'] / 64.0            # lanes that enter the trace loop (after the empty-space proof)
        valu = m['per_wave_step'] * sp['wave_steps'] + m['per_traced_wave'] * traced_waves + m['per_wave'] * waves
        achieved = valu / (prim_avg * 1e-3)
        # SURVEY 8(d) HBM-equivalent figure, kept as a secondary field: 64 fp32 taps per cubic evaluation, film RMW, one grid read
        evals = sp['steps'] + sp['refine_steps']
        alg_bytes = 256.0 * evals + 16 * 2 * 8.0 * sp['lanes'] + 4.0 * args.res ** 3
        roof = {"bound": "valu", "kernel": "k_render_items<primal>", "achieved": achieved / 1e9, "peak": VALU_PEAK / 1e9,
                "unit": "G wave-instr/s", "frac": achieved / VALU_PEAK, "traffic": None,
                "valu_insts_per_launch": valu, "wave_steps_per_launch": sp['wave_steps'], "avg_launch_ms": prim_avg,
                "launch_time_from": "3 launches timed alone after the timed region" if args.overlap else "the timed region",
                "lane_utilisation": evals / max(64.0 * sp['wave_steps'], 1.0), "calibration": VALU_MODEL['calibration'],
                "hbm_equivalent": {"algorithmic_bytes_per_launch": alg_bytes, "GBps": alg_bytes / (prim_avg * 1e-3) / 1e9,
                                   "frac_of_8TBps": alg_bytes / (prim_avg * 1e-3) / 8e12,
                                   "note": "SURVEY 8(d) tap-byte model; taps are LDS/L1-resident, so this is not a bound "
                                           "(measured HBM bytes: profiles/)"}}
        out_cfg = {"mean_steps_per_bbox_lane": sp['steps'] / max(sp['bbox_lanes'], 1),
                   "hit_fraction": sp['hits'] / total_lanes, "traced_fraction": sp['bbox_lanes'] / total_lanes,
                   "generated_fraction": sp['lanes'] / total_lanes,
                   "backward_queue_fraction": sg['queue_len'] / max(nv * (args.img + 4) ** 2 * args.spp_grad, 1),
                   "primal_ms_per_launch": prim_avg, "grad_ms_per_launch": sum(gradt) / len(gradt)}

    if rank == 0:
        strong = args.scaling == 'strong'
        out = {
            "metric": "diff-renders/sec (fwd+bwd) 256^3 SDF, 512^2, 12 views" if strong else
                      "weak-scaling diff-renders/sec (fwd+bwd) 256^3 SDF, 512^2, 12 views PER GPU",
            "value": (1 if strong else world) * args.steps / elapsed, "unit": "renders/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": f"no-tex-12-hq sizes: {args.res}^3 SDF, {args.views} views x {args.img}^2, "
                                        f"{args.integrator}, spp primal/grad {args.spp_primal}/{args.spp_grad} "
                                        f"(reference semantics, configs.py:16,19)",
                            "views_total": args.views if strong else args.views * world, "views_this_rank": nv,
                            "partition": ("pixel-row windows of views: %d units per rank" % len(parallel.work_partition(args.views, args.img + 4, world)[0])) if tiled else "whole views",
                            "spp_primal": args.spp_primal, "spp_grad": args.spp_grad, "grad_l1_last_step": grad_l1,
                            "schedule": "primal pass and gradient sweep on two HIP streams (dsdf.render_step)" if args.overlap and not tiled
                            else "sequential launches"}, **out_cfg),
            "roofline": roof,
        }
        if low is not None:
            out["low_spp"] = low
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
