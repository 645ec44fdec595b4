#!/usr/bin/env python3
"""Offline study (CPU, test infrastructure: uses oracle/ and tests/harness): how much marching would a FINE empty-space proof
remove?  The shipped empty-space proof bounds the field from below by dilated 4^3-block minima (a 12^3-voxel window), so a pixel
only proves empty when its rays stay ~6-8 voxels away from the surface; the hit proof got a second stage on full-resolution window
maxima in round 4 (6^3 window).  This replays the mirror image -- window MINIMA over the taps [c - 2, c + 3]^3 of every B-spline cell --
on one 512^2 view of the bench scene and weighs the pixels it would settle by the steps the fp32 C oracle needs for their rays.

    python tools/sim_proofs.py [view] [spp]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')):
    sys.path.insert(0, p)
import c_oracle
import sdf_oracle as O
import __graft_entry__ as g
from bench import synth_grid
from conftest import HostHarness

PX_EMPTY, PX_EMPTY_G, PX_HIT = 1, 2, 16


def window_min(grid, lo=-2, hi=3):
    """per cell c: min over the taps [c + lo, c + hi] per axis (clamped), separable"""
    a = grid
    for ax in range(3):
        n = a.shape[ax]
        idx = np.arange(n)
        out = np.full_like(a, np.inf)
        for o in range(lo, hi + 1):
            out = np.minimum(out, np.take(a, np.clip(idx + o, 0, n - 1), axis=ax))
        a = out
    return a


def main():
    view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    res, W = 256, 512
    Wb = W + 4
    grid = synth_grid(res, 'cpu').numpy()
    camo = O.Camera(O.regular_camera_origins(12)[view]).rounded()
    cam = camo.params()
    hh = HostHarness(g.build_harness())
    t0 = time.time()
    flags, info = hh.pixel_proof(grid, cam, W, W)
    print(f'shipped proofs: {time.time() - t0:.1f}s, steps {info}', flush=True)
    empty, empty_g, hit = (flags & PX_EMPTY) != 0, (flags & PX_EMPTY_G) != 0, (flags & PX_HIT) != 0
    print(f'pixels: empty {empty.mean():.3f} empty_g {empty_g.mean():.3f} hit-proven {hit.mean():.3f} '
          f'primal-traced {(~empty & ~hit).mean():.3f} sweep-traced {(~empty_g).mean():.3f}')

    # fine empty-space bound along the centre rays of the pixels that are not EMPTY_G
    Fmin = window_min(grid)
    prm = hh.params
    py, px = np.nonzero(~empty_g)
    pos = np.stack([px - 2 + 0.5, py - 2 + 0.5], -1).astype(np.float64)
    o, d, maxt = camo.sample_ray(torch.from_numpy(pos), W, W)
    o, d = o.numpy(), d.numpy()
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    lo, hi = -prm.bbox_delta - 0.02, 1 + prm.bbox_delta + 0.02
    with np.errstate(divide='ignore', invalid='ignore'):
        ta, tb = (lo - o) / d, (hi - o) / d
    tn, tf = np.minimum(ta, tb).max(1), np.maximum(ta, tb).min(1)
    ok = tf >= np.maximum(tn, 0)
    tn = np.maximum(tn, 0)
    fstep = float(info[3])          # the step of the fine hit proof: lateral deviation + half a step within one voxel
    nst = int(np.ceil((tf - tn)[ok].max() / fstep)) + 2
    m = np.full(len(px), np.inf)
    for k in range(nst):
        t = np.minimum(tn + k * fstep, tf)
        x = o + t[:, None] * d
        c = np.clip(np.floor(x * res - 0.5).astype(np.int64), 0, res - 1)
        m = np.minimum(m, Fmin[c[:, 2], c[:, 1], c[:, 0]])
    t1 = tf
    thr_p = 2 * prm.trace_eps * np.maximum(t1, 1) + 1e-5
    thr_g = prm.edge_eps * (t1 + 0.1) * 1.05 + 1e-4
    f_empty = np.zeros_like(empty); f_empty_g = np.zeros_like(empty)
    f_empty[py, px] = ok & (m > thr_p)
    f_empty_g[py, px] = ok & (m > np.maximum(thr_p, thr_g))
    empty2, empty_g2 = empty | f_empty, empty_g | f_empty_g
    print(f'with the fine bound: empty {empty2.mean():.3f} empty_g {empty_g2.mean():.3f} '
          f'primal-traced {(~empty2 & ~hit).mean():.3f} sweep-traced {(~empty_g2).mean():.3f}', flush=True)

    # steps of the rays of every pixel that is traced today
    lib = c_oracle.load(False)
    rng = np.random.default_rng(0)

    def trace(mask, diff):
        py, px = np.nonzero(mask)
        steps = np.zeros((len(px), spp), np.int32)
        need = np.zeros((len(px), spp), bool)
        B = 20000
        for a in range(0, len(px), B):
            b = min(len(px), a + B)
            pos = np.stack([px[a:b], py[a:b]], -1).reshape(-1, 1, 2) - 2 + rng.random((b - a, spp, 2))
            oo, dd, mt = camo.sample_ray(torch.from_numpy(pos.reshape(-1, 2)), W, W)
            tr = c_oracle.trace(lib, grid, oo.numpy(), dd.numpy(), mt.numpy(), diff=diff)
            steps[a:b] = tr['steps'].reshape(-1, spp)
            if diff:
                need[a:b] = (tr['warp_weight'] > 0).reshape(-1, spp)
        return py, px, steps, need

    for name, now, then, diff in (('primal', ~empty & ~hit, ~empty2 & ~hit, False), ('sweep', ~empty_g, ~empty_g2, True)):
        t0 = time.time()
        py, px, steps, need = trace(now, diff)
        keep = then[py, px]
        lane, wave = steps.sum(), steps.max(1).sum()
        lane2, wave2 = steps[keep].sum(), steps[keep].max(1).sum()
        print(f'{name}: traced pixels {len(px)} -> {keep.sum()} ({keep.mean():.3f}); lane-steps {lane} -> {lane2} ({lane2 / lane:.3f}); '
              f'wave-steps (max over {spp} samples) {wave} -> {wave2} ({wave2 / wave:.3f})   [{time.time() - t0:.0f}s]', flush=True)
        drop = ~keep
        if diff:
            print(f'   dropped pixels holding a sample with warp weight > 0 (must be 0): {need[drop].any(1).sum()}')
        else:
            print(f'   dropped pixels: max steps {steps[drop].max() if drop.any() else 0}, mean {steps[drop].mean() if drop.any() else 0:.1f}; '
                  f'kept: mean {steps[keep].mean():.1f}')


if __name__ == '__main__':
    main()
