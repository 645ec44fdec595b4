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
  roofline     -- VALU-issue roofline of the dominant kernel (the primal k_render_items): the path is not HBM-bound (taps
                  are LDS/L1-resident: measured HBM traffic is ~1 % of the algorithmic tap bytes), it issues vector ALU
                  instructions.  achieved = wave-level VALU instructions per launch / launch time; peak = 1024 SIMDs x
                  2.4 GHz / 2 clk per wave64 instruction (MI355X_MICROARCH.md: SIMD-32, `v_fma_f32` 2 cyc).  The
                  instruction count is LIVE x MEASURED: lock-step wave iterations counted by the kernel itself in this
                  process (stats slot 7) x the VALU instructions per wave iteration of the SAME kernel measured with
                  rocprofv3 --pmc (SQ_INSTS_VALU / wave iterations of tools/pmc_workload.py), read from
                  profiles/valu_model.json (written by profiles/summarize_pmc.py -- no literals here).
                  frac_lane_weighted = frac x lane utilisation (active lanes per VALU instruction);
                  useful_flop_frac   = spline FLOPs (168 per lane-evaluation: 84 FMA) / time / 157.3 TFLOP/s fp32 vector peak;
                  traffic            = measured HBM bytes per launch of that kernel from the same PMC run (FETCH_SIZE /
                                       WRITE_SIZE passes; provenance in `traffic_from`), null when the file is absent.
                  The HBM-equivalent algorithmic-bytes figure of SURVEY 8(d) is kept as a secondary field.
  low_spp      -- the same step at 4/1 spp (north_star's >= 50 renders/s point).
  direct       -- BASELINE.json configs[4] sizes on ONE GPU: sdf_direct_reparam, 256^3 + 256^3 x 3 albedo, 12 x 512^2, 256/64 spp.
  opt_iteration-- one whole optimiser iteration at the headline sizes (python/shape_opt.py:75-105): 6 of 12 views
                  (opt_configs.py:245), multiscale-L1 loss, Laplacian regulariser, gradient scrub, Adam, box constraint,
                  redistancing, texture refresh -- the pieces timed one by one with HIP events.
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

_T0 = time.time()


def mark(msg):
    """Progress marker on stderr (flushed): where a stalled run was when it stopped."""
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


VALU_PEAK = 1024 * 2.4e9 / 2.0        # wave64 VALU instructions / s: 256 CUs x 4 SIMD-32, 2 clk per instruction
FP32_VECTOR_PEAK = 157.3e12           # MI355X_MICROARCH.md: fp32 vector peak = 1024 SIMD-32 x 32 lanes x 2 flop (FMA) x 2.4 GHz (scalar v_fma_f32 at full rate; not a packed-math figure)
SPLINE_FLOP_PER_EVAL = 168.0          # 64 + 16 + 4 FMAs of one value-only tricubic lookup


def load_valu_model():
    """profiles/valu_model.json: VALU instructions per lock-step wave iteration of the shipped kernels, measured with
    rocprofv3 --pmc on tools/pmc_workload.py and written by profiles/summarize_pmc.py."""
    path = os.path.join(ROOT, 'profiles', 'valu_model.json')
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


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


def side_roofline(model_key, kernel_name, kernel_ms, stats):
    """VALU-issue roofline of the primal render kernel of a side block (low_spp, direct): measured VALU instructions per wave
    iteration of THAT kernel (profiles/valu_model.json, written by the PMC passes of tools/pmc_workload.py --low --direct) x the
    wave iterations counted live, over the kernel's own launch time (dsdf_kernel_timing_arm / _read)."""
    model = load_valu_model()
    if not model or model_key not in model or not kernel_ms:
        return None
    m = model[model_key]
    valu = m['valu_per_wave_step'] * stats['wave_steps']
    achieved = valu / (kernel_ms * 1e-3)
    evals = stats['steps'] + stats['refine_steps']
    lane_util = evals / max(64.0 * stats['wave_steps'], 1.0)
    return {"bound": "valu", "kernel": kernel_name, "unit": "G wave-instr/s", "peak": VALU_PEAK / 1e9, "achieved": achieved / 1e9,
            "frac": achieved / VALU_PEAK, "frac_lane_weighted": achieved / VALU_PEAK * lane_util, "lane_utilisation": lane_util,
            "avg_launch_ms": kernel_ms, "wave_steps_per_launch": stats['wave_steps'], "valu_per_wave_step": m['valu_per_wave_step'],
            "traffic": m.get('hbm_bytes_per_launch'), "calibration": f"profiles/valu_model.json[{model_key}] (tag {model.get('tag')})"}


def direct_block(args, dev, grid, sensors, steps=4):
    """BASELINE.json configs[4] sizes on one GPU: sdf_direct_reparam with a 256^3 x 3 albedo volume, the same two-stream step."""
    import dsdf
    torch.manual_seed(20240)                                             # (a fixed volume: the checksums below compare across runs)
    albedo = torch.rand(args.res, args.res, args.res, 3, device=dev) * 0.6 + 0.2
    sh = dsdf.Shading(albedo, 1.0, hide_emitters=False)
    galb = torch.zeros_like(albedo)
    grad = torch.zeros(args.res, args.res, args.res, device=dev)
    nv = len(sensors)
    tgt = dsdf.render_forward(grid, sensors, 64, seeds=[2000 + i for i in range(nv)], integrator='sdf_direct_reparam', shading=sh)
    scale = 1.0 / (args.img * args.img * 3)

    def step(it):
        grad.zero_(); galb.zero_()
        seeds = [(it * nv + i) * 2 for i in range(nv)]
        dsdf.render_step(grid, sensors, args.spp_primal, args.spp_grad, lambda im: torch.sign(im - tgt) * scale, grad, seeds,
                         [x + 1 for x in seeds], integrator='sdf_direct_reparam', shading=sh, grad_albedo=galb)
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(1 + k)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # (the stand-alone calls allocate their own workspaces -- 23 GB for the gradient pass of 12 views: once, untimed)
    img = dsdf.render_forward(grid, sensors, args.spp_primal, seeds=list(range(nv)), integrator='sdf_direct_reparam', shading=sh)
    dsdf.render_backward(grid, sensors, args.spp_grad, torch.sign(img - tgt) * scale, grad_grid=grad, seeds=list(range(50, 50 + nv)),
                         integrator='sdf_direct_reparam', shading=sh, grad_albedo=galb)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    dsdf.kernel_timing_arm()
    img = dsdf.render_forward(grid, sensors, args.spp_primal, seeds=list(range(nv)), integrator='sdf_direct_reparam', shading=sh)
    kern_ms = dsdf.kernel_timing_read()
    e1.record()
    dsdf.render_backward(grid, sensors, args.spp_grad, torch.sign(img - tgt) * scale, grad_grid=grad, seeds=list(range(50, 50 + nv)),
                         integrator='sdf_direct_reparam', shading=sh, grad_albedo=galb)
    e2.record()
    torch.cuda.synchronize()
    out = {"value": steps / el, "unit": "renders/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
           "primal_ms_per_launch": e0.elapsed_time(e1), "grad_ms_per_launch": e1.elapsed_time(e2),
           "grad_l1": float(grad.double().abs().sum()), "grad_albedo_l1": float(galb.double().abs().sum()),
           "config": {"workload": f"diffuse-12-hqq sizes (BASELINE.json configs[4]) on ONE GPU: {args.res}^3 SDF + {args.res}^3 x 3 albedo, "
                                  f"{nv} views x {args.img}^2, sdf_direct_reparam (emitter sampling), spp {args.spp_primal}/{args.spp_grad}"}}
    st = dsdf.new_stats(dev)
    dsdf.render_forward(grid, sensors, args.spp_primal, seeds=list(range(nv)), integrator='sdf_direct_reparam', shading=sh, stats=st)
    out["roofline"] = side_roofline('direct_primal', 'k_render_items_store<false, false> (march of the primary rays of the wavefront primal of sdf_direct_reparam)', kern_ms,
                                    dsdf.stats_dict(st))
    del albedo, galb, grad, tgt
    return out



def scaling_prediction_block(args, dev, grid, target, ring, t1_ms, steps=3):
    """PREDICTED strong-scaling curve from ONE GPU (no xGMI is measured here): for N = 2, 4, 8 every rank's shard of the N-way
    split -- the partition a real run uses (parallel.strided_view_shard for whole views, parallel.work_partition / render_step
    with pixel-row windows when N does not divide the views) -- is rendered ALONE on this GPU with the code path that rank
    would execute, and timed.  A step of the N-GPU job takes max over ranks of that time; the one exchange (a ring all-reduce
    of dL/dsdf, 2 (N - 1) / N x S bytes per GPU over one ~153 GB/s xGMI link per direction) is issued
    non-blocking and overlaps the next step, so it only shows where it exceeds the compute; split views add two film
    all-reduces (2 x views x 2 MiB) that are NOT overlapped.  `--emulate-rank r/N` runs a single (r, N)."""
    import dsdf
    from dsdf import parallel
    scale = 1.0 / (args.img * args.img * 3)
    S = 4.0 * args.res ** 3
    link = 153e9
    grad = torch.zeros(args.res, args.res, args.res, device=dev)
    tgt_all = torch.cat([dsdf.render_forward(target, s, 64, seeds=[1000 + i]) for i, s in enumerate(ring)])

    def probe_costs():
        # the row costs a real run's ranks would have all-reduced after the previous step: the proofs of ALL views, once
        seeds0 = [i * 2 for i in range(args.views)]
        ops = parallel.HipOps(grid, ring, args.spp_primal, args.spp_grad, seeds0, [x + 1 for x in seeds0], args.integrator, two_streams=False)
        views = list(range(args.views))
        ops.begin_group(views)
        try:
            ops.sweep(views, (0, args.img + 4))
        finally:
            ops.end_group()
        _, c = ops.row_costs()
        torch.cuda.synchronize()
        dsdf.release_workspaces()
        return c.cpu().numpy()

    cost = probe_costs() if args.partition == 'cost' else None
    tracker = {}

    def rank_ms(r, N):
        split = args.views % N != 0
        if cost is not None:
            part = tracker[N]._last
            seeds0 = [i * 2 for i in range(args.views)]
            ops = parallel.HipOps(grid, ring, args.spp_primal, args.spp_grad, seeds0, [x + 1 for x in seeds0], args.integrator,
                                  two_streams=bool(args.overlap))

            def step(it):
                seeds = [(it * args.views + i) * 2 for i in range(args.views)]
                ops.set_seeds(seeds, [x + 1 for x in seeds])
                grad.zero_()
                parallel.render_step(ops, args.views, args.img, args.img, r, N, lambda im, views: torch.sign(im - tgt_all[views]) * scale,
                                     grad, gather_images=False, partition=part)
                ops.row_costs()                                       # (a real run's ranks compute these every step)
        elif not split:
            mine = parallel.strided_view_shard(list(range(args.views)), r, N)
            sensors = [ring[i] for i in mine]
            tgt = tgt_all[mine]

            def step(it):
                seeds = [(it * args.views + i) * 2 for i in mine]
                grad.zero_()
                dsdf.render_step(grid, sensors, args.spp_primal, args.spp_grad, lambda im: torch.sign(im - tgt) * scale, grad, seeds,
                                 [x + 1 for x in seeds], integrator=args.integrator)
        else:
            seeds0 = [i * 2 for i in range(args.views)]
            ops = parallel.HipOps(grid, ring, args.spp_primal, args.spp_grad, seeds0, [x + 1 for x in seeds0], args.integrator,
                                  two_streams=bool(args.overlap))

            def step(it):
                seeds = [(it * args.views + i) * 2 for i in range(args.views)]
                ops.set_seeds(seeds, [x + 1 for x in seeds])
                grad.zero_()
                parallel.render_step(ops, args.views, args.img, args.img, r, N, lambda im, views: torch.sign(im - tgt_all[views]) * scale,
                                     grad, gather_images=False)
        step(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(1 + k)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        dsdf.release_workspaces()
        return ms

    if args.emulate_rank:
        r, N = (int(x) for x in args.emulate_rank.split('/'))
        if cost is not None:
            tracker[N] = parallel.CostTracker(args.views, args.img + 4, N)
            tracker[N].cost = cost
            tracker[N].partition()
        return {"emulated_rank": r, "world": N, "ms_per_step": rank_ms(r, N)}
    rows = []
    for N in (2, 4, 8):
        mark(f'scaling_prediction N={N}')
        first = None
        if cost is not None:
            # the ranks' partition as a real run reaches it: the proxy's cut first, then `rounds` steps of time feedback
            # (parallel.CostTracker.apply_times: the rows of slow shards get heavier)
            tracker[N] = parallel.CostTracker(args.views, args.img + 4, N)
            tracker[N].cost = cost
            for rnd in range(3):
                tracker[N].partition()
                per = [rank_ms(r, N) for r in range(N)]
                if first is None:
                    first = per
                if rnd < 2:
                    tracker[N].apply_times(per)
        else:
            per = [rank_ms(r, N) for r in range(N)]
        split = args.views % N != 0
        ar = 1e3 * 2.0 * (N - 1) / N * S / link
        film = 1e3 * 2.0 * (2.0 * (N - 1) / N * args.views * (args.img + 4) ** 2 * 2 * 4 / link) if split else 0.0
        comp = max(per)
        pred = max(comp + film, ar)
        rows.append({"n_gpus": N, "partition": ("cost-aware " if cost is not None else "") + ("pixel-row windows" if split else "whole views"),
                     "max_over_min_rank_ms": max(per) / min(per), "rank_ms": [round(x, 3) for x in per],
                     "rank_ms_before_time_feedback": None if first is None else [round(x, 3) for x in first],
                     "compute_ms_max_over_ranks": comp, "allreduce_ms_model": ar, "film_exchange_ms_model": film,
                     "predicted_ms_per_step": pred, "predicted_speedup": t1_ms / pred, "predicted_efficiency": t1_ms / pred / N})
    return {"label": "PREDICTED from single-GPU shard timings + a link model; no xGMI / RCCL transfer between two GPUs was measured",
            "one_gpu_ms_per_step": t1_ms, "allreduce_model": "ring, 2 (N - 1) / N x 4 R^3 bytes per GPU at 153 GB/s per link direction, overlapped "
            "with the next step's rendering (bench.py issues it non-blocking on alternating gradient buffers)", "rows": rows}


def opt_iteration_block(args, dev, data0, ring, iters=6):
    """One whole optimiser iteration (python/shape_opt.py:75-105) at the headline sizes, with the host modules the CLI uses:
    a batch of 6 of the 12 views (opt_configs.py:245) rendered through the autograd op (primal 256 spp, gradient pass 64 spp),
    multiscale-L1 loss (losses.py:33-42) per view / batch_size, Laplacian regulariser (regularizations.py:5-25,
    weight 1e-5 as `no-tex-12`), gradient scrub (variables.py:193-199), Adam (mi.ad.Adam conventions), learning-rate
    schedule + box constraint + redistancing (variables.py:168-190), texture refresh.  Every piece is bracketed by HIP
    events; the iteration time is wall-clock over `iters` iterations."""
    import dsdf
    import losses
    import regularizations
    import variables
    key = 'SamplingIntegrator.sdf.data'
    opt = variables.Adam(lr=2e-3)
    var = variables.SdfVariable(key, args.res, upsample_iter=None, regularizer=regularizations.eval_discrete_laplacian_reg,
                                regularizer_weight=1e-5)
    var.initialize(opt)
    opt[key] = variables.atleast_4d(data0.clone())
    grid = dsdf.SdfGrid(opt[key])
    batch = 6
    tgt_grid = dsdf.SdfGrid(synth_grid(args.res, dev, seed=1))
    refs = dsdf.render_forward(tgt_grid, ring, 64, seeds=[3000 + i for i in range(len(ring))])
    names = ('render_primal', 'loss_and_gradient_pass', 'regulariser', 'scrub', 'adam', 'box_redistance', 'texture_refresh')
    acc = {n: 0.0 for n in names}
    pending = []

    def iteration(i, timed):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        idx = [(i * batch + j) % len(ring) for j in range(batch)]
        ev[0].record()
        imgs = dsdf.render(opt[key], grid, [ring[j] for j in idx], spp=args.spp_primal, seed=17 * i, spp_grad=args.spp_grad,
                           seed_grad=17 * i + 7)
        ev[1].record()
        loss = sum(losses.multiscale_l1(imgs[j], refs[v]) for j, v in enumerate(idx)) / batch
        loss.backward()
        ev[2].record()
        reg = var.eval_regularizer(opt, None, i)
        if isinstance(reg, torch.Tensor) and reg.requires_grad:
            reg.backward()
        ev[3].record()
        var.validate_gradient(opt, i)
        ev[4].record()
        opt.step()
        ev[5].record()
        var.validate(opt, i)
        ev[6].record()
        grid.update(opt[key])
        ev[7].record()
        if timed:
            pending.append(ev)

    iteration(0, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        iteration(1 + i, True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for ev in pending:
        for k, n in enumerate(names):
            acc[n] += ev[k].elapsed_time(ev[k + 1])
    parts = {n: acc[n] / iters for n in names}
    return {"value": iters / el, "unit": "iterations/s", "iterations": iters, "ms_per_iteration": 1e3 * el / iters,
            "ms_parts": parts, "ms_parts_sum": sum(parts.values()),
            "config": {"workload": f"{args.res}^3 SDF, batch of {batch} of {len(ring)} views x {args.img}^2, sdf_silhouette_reparam, spp "
                                   f"{args.spp_primal}/{args.spp_grad}, multiscale-L1 + Laplacian (1e-5), Adam, box constraint, redistance"}}


def supervise(argv):
    """Single-GPU runs are SUPERVISED: the measurement runs in a child process (`--child`), this process only relays its JSON lines.
    A kernel that never returns cannot be interrupted from inside its own process (the host thread sits in hipStreamSynchronize),
    and round 5's driver run was lost exactly so: 1800 s at 100 % GPU, nothing printed (profiles/r06_bench_stall.md).  Rules:
      * the child prints the headline line right after the timed region and the enriched line after every side block; every line
        is relayed at once (flushed), so whatever happens later the last line on stdout is a complete record;
      * no line within BENCH_HEADLINE_S (300 s) -> the child is killed (its GPU queues die with it) and started ONCE more;
      * a headline but then silence for BENCH_BLOCK_S (150 s; the longest block, the CPU baseline, takes ~40 s) -> the child is killed,
        the last line is re-printed with "aborted": <reason>, exit code 0: the headline was measured.
    Multi-rank runs (torch.distributed.run) are not supervised: killing one rank of a collective helps nobody."""
    import subprocess
    import threading
    import queue
    head_s = float(os.environ.get('BENCH_HEADLINE_S', 300))
    block_s = float(os.environ.get('BENCH_BLOCK_S', 150))
    last = None
    for attempt in (1, 2):
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv) + ['--child'], stdout=subprocess.PIPE, text=True, bufsize=1)
        q = queue.Queue()

        def pump(f=child.stdout, q=q):            # (q bound NOW: the pump of a killed first child ends after the second attempt has made its queue)
            for line in f:
                q.put(line)
            q.put(None)
        threading.Thread(target=pump, daemon=True).start()
        reason = None
        while True:
            try:
                line = q.get(timeout=head_s if last is None else block_s)
            except queue.Empty:
                reason = (f"no headline within {head_s:.0f} s" if last is None else f"no progress for {block_s:.0f} s after the headline") + f" (attempt {attempt})"
                break
            if line is None:
                break
            line = line.rstrip('\n')
            if line.startswith('{'):
                last = line
            print(line, flush=True)
        if reason is None:
            rc = child.wait()
            if rc == 0 and last is not None:
                return 0
            reason = f"child exited with code {rc} (attempt {attempt})"
        else:
            child.kill()
            child.wait()
        print(f"[bench supervisor] {reason}", file=sys.stderr, flush=True)
        if last is not None:
            try:
                rec = json.loads(last)
                rec["aborted"] = reason
                print(json.dumps(rec), flush=True)
            except ValueError:
                print(last, flush=True)
            return 0
    return 1


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
    ap.add_argument('--no-direct', action='store_true')
    ap.add_argument('--no-opt-iteration', action='store_true')
    ap.add_argument('--no-scaling-prediction', action='store_true')
    ap.add_argument('--emulate-rank', default='', help="r/N: time rank r's shard of an N-GPU strong-scaling run alone on this GPU (scaling_prediction)")
    ap.add_argument('--overlap', type=int, default=1, help='1: primal pass and gradient sweep on two HIP streams (dsdf.render_step)')
    ap.add_argument('--partition', choices=('cost', 'uniform'), default='cost',
                    help="multi-rank strong scaling: 'cost' = parallel.cost_partition of the row costs the ranks measured in the previous step "
                         "(per-pixel proof flags, one 25 KB all-reduce per step); 'uniform' = round-robin views / even row windows")
    ap.add_argument('--child', action='store_true', help='(internal) the measuring process of a supervised single-GPU run')
    args = ap.parse_args()
    if int(os.environ.get('WORLD_SIZE', 1)) == 1 and not args.child and os.environ.get('BENCH_INPROCESS') != '1':
        sys.exit(supervise(sys.argv[1:]))
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(int(os.environ.get('BENCH_TRACEBACK_EVERY', 120)), repeat=True, file=sys.stderr)

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
    dist, backend = None, None
    # (BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank, e.g. under `torchrun --nproc-per-node 1`)
    if world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    mark('library loaded; building grids')
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
        # (the cost-aware partition re-deals views and row windows from step to step: every multi-rank run goes through the film-level
        # operators of parallel.render_step then, also when the ranks divide the views)
        tiled = args.views % world != 0 or os.environ.get('BENCH_FORCE_TILED') == '1' or (world > 1 and args.partition == 'cost')
        mine = list(range(args.views)) if tiled else parallel.strided_view_shard(list(range(args.views)), rank, world)
    else:
        # weak scaling: a ring of views*world sensors, `views` per rank
        ring = dsdf.get_regular_cameras(args.views * world, resx=args.img, resy=args.img)
        mine = list(range(args.views * world))[rank::world]
    sensors = [ring[i] for i in mine]
    nv = len(sensors)
    # two gradient buffers, used alternately: the RCCL all-reduce of step i is issued non-blocking and overlaps the rendering of
    # step i + 1, which accumulates into the other buffer
    # (each a parallel.GradBucket: dL/dsdf and, for C5, dL/d(albedo) are views into ONE persistent flat buffer, so the all-reduce
    # moves that buffer as it is -- no per-step torch.cat / copy-back)
    # BASELINE.json C5 (--integrator sdf_direct_reparam): a 3-channel albedo volume of the grid's resolution is optimised too
    shade, galbs = {}, [None, None]
    shapes = [tuple(data.shape)]
    if args.integrator == 'sdf_direct_reparam':
        torch.manual_seed(20240)                                             # (a fixed volume: the checksums below compare across runs)
        albedo = torch.rand(args.res, args.res, args.res, 3, device=dev) * 0.6 + 0.2
        shade = {'shading': dsdf.Shading(albedo, 1.0, hide_emitters=False)}
        shapes.append(tuple(albedo.shape))
    buckets = [parallel.GradBucket(shapes, dev), parallel.GradBucket(shapes, dev)]
    grads = [b.views[0] for b in buckets]
    if len(shapes) > 1:
        galbs = [b.views[1] for b in buckets]
    # target images (outside the timed region) -> L1 image gradient sign(img - target)/(H*W*3)
    tgt = torch.cat([dsdf.render_forward(target, s, 64, seeds=[1000 + i]) for i, s in zip(mine, sensors)]) if nv else None
    scale = 1.0 / (args.img * args.img * 3)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    hip_ops = {}
    trackers = {}                               # (spp pair) -> parallel.CostTracker: the row costs all ranks agreed on after the last step
    step_events = {}                            # (spp pair) -> HIP events around this rank's last step
    pending = [None, None]                      # work handles of the all-reduces in flight, per gradient buffer

    def make_step(spp_p, spp_g, prim_ms, grad_ms):
        def step(it, timed):
            # one launch traces all views of this rank's shard (primal), one launch the gradient pass
            b = it & 1
            if pending[b] is not None:          # the reduce that last used this buffer (two steps ago)
                pending[b].wait(); pending[b] = None
            grad, galb = grads[b], galbs[b]
            grad.zero_()
            if galb is not None:
                galb.zero_()
            extra = [galb] if galb is not None else []
            if tiled:
                seeds = [(it * args.views + i) * 2 for i in range(args.views)]
                key = (spp_p, spp_g)
                if key not in hip_ops:          # (kept across steps: its sweep workspaces and side stream are re-used)
                    hip_ops[key] = parallel.HipOps(grid, ring, spp_p, spp_g, seeds, [x + 1 for x in seeds], args.integrator,
                                                   two_streams=bool(args.overlap), **shade)
                ops = hip_ops[key]
                ops.set_seeds(seeds, [x + 1 for x in seeds])
                ops.kw = dict(shade, **({'grad_albedo': galb} if galb is not None else {}))
                part = None
                if args.partition == 'cost':
                    if key not in trackers:
                        trackers[key] = parallel.CostTracker(args.views, args.img + 4, world, device=dev)
                    part = trackers[key].partition()
                t0, t1 = ev(), ev()
                t0.record()
                _, work = parallel.render_step(ops, args.views, args.img, args.img, rank, world,
                                               lambda im, views: torch.sign(im - tgt[views]) * scale, grad, extra_grads=extra,
                                               gather_images=False, async_reduce=True, force_reduce=dist is not None, partition=part)
                t1.record()
                pending[b] = work
                if part is not None:
                    # what this step's proofs say about the work per row, and how long the PREVIOUS step took on this rank (its events
                    # have completed; this step's have not) -> the next step's partition.  One small collective.
                    prev = step_events.get(key)
                    ms = prev[0].elapsed_time(prev[1]) if prev is not None and prev[1].query() else 0.0
                    step_events[key] = (t0, t1)
                    trackers[key].update(*ops.row_costs(), rank=rank, step_ms=ms)
                return
            if nv:
                seeds = [(it * args.views + i) * 2 for i in mine]
                e0, e1, e2 = ev(), ev(), ev()
                if args.overlap:
                    # primal render and the forward sweep of the gradient pass on two HIP streams (dsdf.render_step)
                    e0.record()
                    dsdf.render_step(grid, sensors, spp_p, spp_g, lambda im: torch.sign(im - tgt) * scale, grad, seeds,
                                     [x + 1 for x in seeds], integrator=args.integrator, shading=shade.get('shading'),
                                     grad_albedo=galb)
                    e1.record(); e2.record()
                else:
                    e0.record()
                    img = dsdf.render_forward(grid, sensors, spp_p, seeds=seeds, integrator=args.integrator, **shade)
                    e1.record()
                    gi = torch.sign(img - tgt) * scale
                    dsdf.render_backward(grid, sensors, spp_g, gi, grad_grid=grad, seeds=[x + 1 for x in seeds],
                                         integrator=args.integrator, grad_albedo=galb, **shade)
                    e2.record()
                if timed:
                    prim_ms.append((e0, e1)); grad_ms.append((e1, e2))
            if dist is not None:
                pending[b] = parallel.all_reduce_gradients([grad] + extra, async_op=True, force=True)
        return step

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait(); pending[b] = None

    def barrier():
        drain()
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
    mark('timed region: warmup + steps')
    elapsed = timed_run(make_step(args.spp_primal, args.spp_grad, prim_ms, grad_ms), args.warmup, args.steps)
    # dL/dsdf of the last timed step (summed over ranks): must agree at every N up to the order of the float atomics
    grad_l1 = float(grads[(args.warmup + args.steps - 1) & 1].double().abs().sum())
    kern_p, kern_g = [], []
    mark(f'timed region done: {1e3 * elapsed / args.steps:.2f} ms/step; probe launches')
    if nv and not tiled:
        # the roofline needs the dominant kernel's OWN launch time: in the timed region above it shares the chip with the
        # gradient sweep of the other stream, so a few calls are timed alone afterwards, same process, same inputs -- the whole
        # call with HIP events on the launch stream (prim_ms / grad_ms), the RENDER KERNEL inside it with the library's
        # measurement hook (dsdf_kernel_timing_arm / _read: two HIP events around k_render_items on the same stream)
        prim_ms, grad_ms = [], []
        ov, args.overlap = args.overlap, 0
        dist_saved, dist = dist, None
        probe = make_step(args.spp_primal, args.spp_grad, prim_ms, grad_ms)
        probe(args.warmup + args.steps, False)                 # (untimed: the stand-alone calls allocate their own workspaces)
        for k in range(3):
            probe(args.warmup + args.steps + 1 + k, True)
        torch.cuda.synchronize()
        gi0 = torch.ones(nv, args.img, args.img, 3, device=dev) * scale
        for k in range(3):
            seeds = [(7 * k + i) * 2 for i in range(nv)]
            dsdf.kernel_timing_arm()
            dsdf.render_forward(grid, sensors, args.spp_primal, seeds=seeds, integrator=args.integrator, **shade)
            kern_p.append(dsdf.kernel_timing_read())
            dsdf.kernel_timing_arm()
            dsdf.render_backward(grid, sensors, args.spp_grad, gi0, grad_grid=grads[0], seeds=[x + 1 for x in seeds], integrator=args.integrator,
                                 grad_albedo=galbs[0], **shade)
            kern_g.append(dsdf.kernel_timing_read())
        args.overlap, dist = ov, dist_saved

    # per-launch statistics (untimed; same launch shape as the timed ones)
    out_cfg, roof = {}, None
    mark('per-launch statistics')
    if nv and not tiled:
        st_p, st_g = dsdf.new_stats(dev), dsdf.new_stats(dev)
        galb = galbs[0]
        dsdf.render_forward(grid, sensors, args.spp_primal, seeds=list(range(nv)), integrator=args.integrator, stats=st_p, **shade)
        dsdf.render_backward(grid, sensors, args.spp_grad, torch.ones(nv, args.img, args.img, 3, device=dev) * scale,
                             grad_grid=torch.zeros_like(data), seeds=list(range(50, 50 + nv)), integrator=args.integrator,
                             grad_albedo=galb, stats=st_g, **shade)
        sp, sg = dsdf.stats_dict(st_p), dsdf.stats_dict(st_g)
        prim = [a.elapsed_time(b) for a, b in prim_ms]
        gradt = [a.elapsed_time(b) for a, b in grad_ms]
        prim_avg = sum(prim) / len(prim)
        kern_avg = sum(kern_p) / len(kern_p)                          # the render kernel alone (measurement hook)
        total_lanes = nv * (args.img + 4) ** 2 * args.spp_primal
        # VALU-issue roofline of the primal launch: measured VALU instructions per wave iteration x live wave iterations
        model = load_valu_model()
        evals = sp['steps'] + sp['refine_steps']                     # lane-evaluations of the render kernel (the tail kernel's: tail_steps)
        lane_util = evals / max(64.0 * sp['wave_steps'], 1.0)
        alg_bytes = 256.0 * evals + 16 * 2 * 8.0 * sp['lanes'] + 4.0 * args.res ** 3
        roof = {"bound": "valu", "kernel": "k_render_items<false, false, false> (primal render kernel)", "unit": "G wave-instr/s", "peak": VALU_PEAK / 1e9,
                "achieved": None, "frac": None, "frac_lane_weighted": None, "traffic": None,
                "wave_steps_per_launch": sp['wave_steps'], "avg_launch_ms": kern_avg, "call_ms": prim_avg,
                "launch_time_from": "3 launches after the timed region, alone on the chip: HIP events recorded by the library on the launch stream "
                                    "directly around k_render_items (dsdf_kernel_timing_arm / _read); call_ms = the whole dsdf_render_forward call "
                                    "(pixel-skip + list build + render kernel + tail kernel + develop) by events around the call",
                "lane_utilisation": lane_util,
                "useful_flop_frac": SPLINE_FLOP_PER_EVAL * evals / (kern_avg * 1e-3) / FP32_VECTOR_PEAK,
                "sweep_kernel_ms": sum(kern_g) / len(kern_g),
                "tail": {"rays": sp['tail_rays'], "lane_steps": sp['tail_steps'], "wave_steps": sp['tail_wave_steps']},
                "hbm_equivalent": {"algorithmic_bytes_per_launch": alg_bytes, "GBps": alg_bytes / (kern_avg * 1e-3) / 1e9,
                                   "frac_of_8TBps": alg_bytes / (kern_avg * 1e-3) / 8e12,
                                   "note": "SURVEY 8(d) tap-byte model; taps are LDS/L1-resident, so this is not a bound"}}
        if model and 'primal' in model:
            m = model['primal']
            if 'valu_per_wave_iteration' in m:
                # (the hit proof removes the march of whole chunks but not their set-up / film code: VALU = a x wave iterations + b x chunks of
                # 64 generated samples, both coefficients measured -- profiles/summarize_pmc.py)
                valu = m['valu_per_wave_iteration'] * sp['wave_steps'] + m['valu_per_chunk'] * sp['lanes'] / 64.0
            else:
                valu = m['valu_per_wave_step'] * sp['wave_steps']
            achieved = valu / (kern_avg * 1e-3)
            roof.update({"achieved": achieved / 1e9, "frac": achieved / VALU_PEAK, "frac_lane_weighted": achieved / VALU_PEAK * lane_util,
                         "valu_insts_per_launch": valu, "valu_per_wave_step": valu / max(sp['wave_steps'], 1),
                         "valu_per_wave_iteration": m.get('valu_per_wave_iteration'), "valu_per_chunk": m.get('valu_per_chunk'),
                         "chunks_per_launch": sp['lanes'] / 64.0,
                         "calibration": f"profiles/valu_model.json (tag {model.get('tag')}): SQ_INSTS_VALU {m.get('valu_insts_per_launch')} / "
                                        f"{m.get('wave_steps_per_launch')} wave iterations of tools/pmc_workload.py",
                         "traffic": m.get('hbm_bytes_per_launch'), "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate PMC passes)",
                         "traffic_from": model.get('source')})
        # the gradient sweep has its own VALU-issue figure (since the hit proof halved the primal march the two render kernels take the same time)
        if model and 'sweep' in model and kern_g:
            ms = model['sweep']
            kg = sum(kern_g) / len(kern_g)
            g_evals = sg['steps'] + sg['refine_steps']
            g_util = g_evals / max(64.0 * sg['wave_steps'], 1.0)
            g_valu = ms['valu_per_wave_step'] * sg['wave_steps']
            sweep = {"bound": "valu", "kernel": "k_render_items<true, false, false> (gradient sweep)", "unit": "G wave-instr/s", "peak": VALU_PEAK / 1e9,
                     "achieved": g_valu / (kg * 1e-3) / 1e9, "frac": g_valu / (kg * 1e-3) / VALU_PEAK,
                     "frac_lane_weighted": g_valu / (kg * 1e-3) / VALU_PEAK * g_util, "traffic": ms.get('hbm_bytes_per_launch'),
                     "avg_launch_ms": kg, "wave_steps_per_launch": sg['wave_steps'], "lane_utilisation": g_util,
                     "valu_insts_per_launch": g_valu, "valu_per_wave_step": ms['valu_per_wave_step'],
                     "calibration": f"profiles/valu_model.json[sweep] (tag {model.get('tag')})"}
            # (`roofline` stays the primal render kernel, as in every round; the sweep -- as long as the primal kernel since the hit
            # proof -- is listed beside it, and both together as one figure: their instructions over their summed launch times)
            roof["other_kernel"] = sweep
            if roof.get("valu_insts_per_launch"):
                roof["render_kernels_combined_frac"] = (roof["valu_insts_per_launch"] + g_valu) / ((kern_avg + kg) * 1e-3) / VALU_PEAK
        out_cfg = {"mean_steps_per_bbox_lane": (sp['steps'] + sp['tail_steps']) / max(sp['bbox_lanes'], 1),
                   "hit_fraction": sp['hits'] / total_lanes, "traced_fraction": sp['bbox_lanes'] / total_lanes,
                   "generated_fraction": sp['lanes'] / total_lanes, "handed_off_fraction": sp['tail_rays'] / max(sp['bbox_lanes'], 1),
                   "backward_queue_fraction": sg['queue_len'] / max(nv * (args.img + 4) ** 2 * args.spp_grad, 1),
                   "primal_ms_per_launch": prim_avg, "grad_ms_per_launch": sum(gradt) / len(gradt)}

    def low_spp_block():
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
        if nv and not tiled:
            lk = []
            for k in range(3):
                dsdf.kernel_timing_arm()
                dsdf.render_forward(grid, sensors, 4, seeds=[(11 * k + i) * 2 for i in range(nv)], integrator=args.integrator, **shade)
                lk.append(dsdf.kernel_timing_read())
            lst = dsdf.new_stats(dev)
            dsdf.render_forward(grid, sensors, 4, seeds=list(range(nv)), integrator=args.integrator, stats=lst, **shade)
            low["roofline"] = side_roofline('low_primal', 'k_render_pass<false, false> (general pass, 4 spp)', sum(lk) / len(lk), dsdf.stats_dict(lst))
        return low

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
                        "partition": (("cost-aware (parallel.cost_partition of the previous step's row costs): " if args.partition == 'cost' and world > 1 else "uniform: ") +
                                      "%d units per rank" % len(parallel.work_partition(args.views, args.img + 4, world)[0])) if tiled else "whole views",
                        "spp_primal": args.spp_primal, "spp_grad": args.spp_grad, "grad_l1_last_step": grad_l1,
                        "dist_backend": backend,
                        "gradient_exchange": None if dist is None else "one non-blocking all-reduce of dL/dsdf per step, overlapped with the next step's rendering (two gradient buffers)",
                        "schedule": "primal pass and gradient sweep on two HIP streams" if args.overlap else "sequential launches"}, **out_cfg),
        "roofline": roof,
    }

    # THE HEADLINE FIRST (a stall in a side block must not lose the measurement).  The line is printed and flushed
    # here, right after the timed region and its probe launches; every side block that completes re-prints the enriched line (the
    # driver parses the last one).  A side block that raises is recorded as {"error": ...}; one that would start after the
    # wall-clock budget is recorded as {"skipped": "budget"}; a block that never returns is the supervisor's business (main()).
    def emit():
        if rank == 0:
            out["bench_wall_s"] = round(time.time() - _T0, 1)
            print(json.dumps(out), flush=True)

    emit()
    budget = float(os.environ.get('BENCH_BUDGET_S', 150))

    def side(name, fn, enabled=True):
        if not enabled:
            return
        if dist is None and time.time() - _T0 > budget:      # (one rank only: ranks must not disagree about a collective block)
            out[name] = {"skipped": "budget", "budget_s": budget}
            mark(f'{name}: skipped (wall-clock budget of {budget:.0f} s spent)')
        else:
            mark(f'{name} block')
            t0 = time.time()
            try:
                res = fn()
                if res is None:
                    return
                out[name] = res
            except Exception as e:                                   # noqa: BLE001 -- the headline above must survive any side block
                import traceback
                traceback.print_exc()
                out[name] = {"error": f"{type(e).__name__}: {e}"}
                try:
                    drain(); torch.cuda.synchronize()
                except Exception:                                    # noqa: BLE001
                    pass
            mark(f'{name} block done in {time.time() - t0:.1f} s')
        emit()

    solo = world == 1 and dist is None
    side('cpu_baseline', lambda: cpu_baseline(args), rank == 0 and world == 1 and not args.no_cpu_baseline)
    side('low_spp', low_spp_block, not args.no_low_spp)

    def run_direct():
        dsdf.release_workspaces()
        return direct_block(args, dev, grid, sensors)

    def run_opt():
        dsdf.release_workspaces()
        return opt_iteration_block(args, dev, data, ring)

    def run_scaling():
        dsdf.release_workspaces()
        return scaling_prediction_block(args, dev, grid, target, ring, 1e3 * elapsed / args.steps)

    silhouette = args.integrator == 'sdf_silhouette_reparam'
    side('direct', run_direct, solo and silhouette and not args.no_direct)
    side('opt_iteration', run_opt, solo and silhouette and not args.no_opt_iteration)
    side('scaling_prediction', run_scaling, solo and strong and bool(args.emulate_rank or not args.no_scaling_prediction))
    mark('done')
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
