#!/usr/bin/env python3
"""Offline study of the primal march's lane utilisation, on the CPU with the fp32 C oracle.

Traces the 64-sample chunks of one 512^2 view of the bench scene (256^3 synthetic grid), records the per-ray step counts and
replays candidate wave shapes / schedules on them:
  pixel-wave      the shipped shape: a wave = the 64 samples of one pixel, marched until its slowest ray is done
  cap K           the wave stops after K lock-step iterations; survivors are compacted into full waves (ideal repacking)
  hand-off H      the wave stops when <= H rays are left
Prints utilisation = lane-steps / (64 x wave-steps) for each.  Test infrastructure (uses oracle/): never on the product path.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')):
    sys.path.insert(0, p)
import c_oracle
import sdf_oracle as O
from bench import synth_grid


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    view = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    spp = 64
    cache = f'/tmp/sim_waves_{res}_{W}_{view}.npz'
    if os.path.isfile(cache):
        z = np.load(cache)
        steps, hit = z['steps'], z['hit']
    else:
        lib = c_oracle.load(False)
        grid = synth_grid(res, 'cpu').numpy()
        cam = O.Camera(O.regular_camera_origins(12)[view]).rounded()
        Wb = W + 4
        rng = np.random.default_rng(0)
        steps = np.zeros((Wb * Wb, spp), np.int32)
        hit = np.zeros((Wb * Wb, spp), bool)
        rows = 16
        for y0 in range(0, Wb, rows):
            y1 = min(Wb, y0 + rows)
            py, px = np.meshgrid(np.arange(y0, y1), np.arange(Wb), indexing='ij')
            pos = np.stack([px, py], -1).reshape(-1, 1, 2) - 2 + rng.random(((y1 - y0) * Wb, spp, 2))
            o, d, maxt = cam.sample_ray(torch.from_numpy(pos.reshape(-1, 2)), W, W)
            tr = c_oracle.trace(lib, grid, o.numpy(), d.numpy(), maxt.numpy(), diff=False)
            steps[y0 * Wb:y1 * Wb] = tr['steps'].reshape(-1, spp)
            hit[y0 * Wb:y1 * Wb] = np.isfinite(tr['its_t']).reshape(-1, spp)
            print(f'rows {y0}-{y1}', file=sys.stderr, flush=True)
        np.savez_compressed(cache, steps=steps, hit=hit)
    Wb = W + 4
    # emulate the empty-space proof: a pixel is traced when a hit lies within `m` pixels (the min-grid margin), generated
    # (weight-only) within m + 4
    from scipy.ndimage import maximum_filter
    anyhit = hit.any(1).reshape(Wb, Wb)
    for m in (6, 8, 10):
        traced = maximum_filter(anyhit, size=2 * m + 1)
        print(f'margin {m}: traced fraction {traced.mean():.3f} (bench: 0.237), hit fraction {hit.mean():.3f} (bench 0.177)')
    traced = maximum_filter(anyhit, size=17).ravel()
    S = steps[traced].astype(np.int64)
    Hh = hit[traced]
    wmax = S.max(1)
    lane_steps, wave_steps = S.sum(), wmax.sum()
    print(f'traced pixels {traced.sum()}, mean steps {S.mean():.2f}, pixel-wave utilisation {lane_steps / (64 * wave_steps):.3f}')
    allhit, nohit = Hh.all(1), ~Hh.any(1)
    mixed = ~allhit & ~nohit
    for name, m in (('all-hit', allhit), ('no-hit', nohit), ('mixed', mixed)):
        if m.sum():
            print(f'  {name:8s}: {m.mean():.3f} of waves, {wmax[m].sum() / wave_steps:.3f} of wave-steps, utilisation '
                  f'{S[m].sum() / (64 * wmax[m].sum()):.3f}, mean max {wmax[m].mean():.1f}')
    # where are the idle lane-steps? by number of active lanes at each lock-step iteration
    srt = np.sort(S, 1)[:, ::-1]            # descending per wave: srt[:, k] = steps of the (k+1)-th longest ray
    # iterations during which exactly a lanes are active: srt[:, a-1] - srt[:, a]
    act = np.zeros(65)
    for a in range(1, 65):
        nxt = srt[:, a] if a < 64 else 0
        act[a] = (srt[:, a - 1] - nxt).sum()
    cum = np.cumsum(act[1:]) / wave_steps
    print('  share of wave-steps with <= a active lanes: ' + ', '.join(f'a={a}: {cum[a - 1]:.3f}' for a in (1, 2, 4, 8, 16, 24, 32, 48, 63)))
    print('cap K (ideal repacking of survivors into full waves):')
    for K in (8, 12, 16, 20, 24, 32, 48):
        p1 = np.minimum(wmax, K).sum()
        rest = np.maximum(S - K, 0)
        surv = (rest > 0).sum()
        # survivors re-packed in order into waves of 64 (neighbouring pixels), each marched to ITS end
        r = rest[rest > 0]
        pad = (-len(r)) % 64
        r2 = np.concatenate([r, np.zeros(pad, np.int64)]).reshape(-1, 64)
        p2 = r2.max(1).sum()
        # second-level cap: phase 2 also capped at K, then phase 3 ...
        tot, cur = p1, rest[rest > 0]
        while len(cur):
            pad = (-len(cur)) % 64
            c2 = np.concatenate([cur, np.zeros(pad, np.int64)]).reshape(-1, 64)
            tot += np.minimum(c2.max(1), K).sum()
            cur = cur - K
            cur = cur[cur > 0]
        print(f'  K={K:3d}: phase-1 wave-steps {p1 / wave_steps:.3f}, survivors {surv / S.size:.3f} of rays; one re-pack: total '
              f'{(p1 + p2) / wave_steps:.3f} (util {lane_steps / (64 * (p1 + p2)):.3f}); re-pack every K: total {tot / wave_steps:.3f} '
              f'(util {lane_steps / (64 * tot):.3f})')
    print('hand-off at <= H active rays (survivors ideally repacked, marched to their end):')
    for Hn in (4, 8, 16, 24, 32):
        stop = srt[:, Hn] if Hn < 64 else 0        # iterations until only Hn remain
        p1 = stop.sum()
        rest = np.maximum(S - stop[:, None], 0)
        r = rest[rest > 0]
        pad = (-len(r)) % 64
        r2 = np.sort(np.concatenate([r, np.zeros(pad, np.int64)]))[::-1].reshape(-1, 64)     # (sorted: best case)
        r3 = np.concatenate([r, np.zeros(pad, np.int64)]).reshape(-1, 64)
        print(f'  H={Hn:2d}: phase-1 {p1 / wave_steps:.3f}, survivors {len(r) / S.size:.3f}; total in-order {(p1 + r3.max(1).sum()) / wave_steps:.3f}, '
              f'sorted {(p1 + r2.max(1).sum()) / wave_steps:.3f}')
    # alternative wave shapes on the same rays: 16 samples x 2x2 pixels, 32 x 2
    P = steps.reshape(Wb, Wb, spp)
    T = traced.reshape(Wb, Wb)
    for (th, tw) in ((2, 2), (1, 2), (4, 4)):
        n = th * tw
        k = 64 // n
        A = P[:Wb // th * th, :Wb // tw * tw].reshape(Wb // th, th, Wb // tw, tw, spp).transpose(0, 2, 1, 3, 4)
        Tm = T[:Wb // th * th, :Wb // tw * tw].reshape(Wb // th, th, Wb // tw, tw).transpose(0, 2, 1, 3).any((2, 3))
        A = A[Tm]                                   # (tiles, th, tw, spp)
        ws = 0
        for c in range(spp // k):
            ws += A[:, :, :, c * k:(c + 1) * k].reshape(len(A), -1).max(1).sum()
        print(f'wave = {k} samples x {th}x{tw} pixels: wave-steps {ws / wave_steps:.3f} of pixel-waves (lane-steps {A.sum() / lane_steps:.3f})')


if __name__ == '__main__':
    main()
