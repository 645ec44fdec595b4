#!/usr/bin/env python3
"""Auxiliary timings quoted in DESIGN.md: redistancing at 16^3..512^3 (method of the reference's
figures/benchmark/benchmark.py:120-144), one optimisation iteration of `no-tex-12-hqq`-sized state,
and the bench workload at other sample counts / the shading integrator."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'))
sys.path.insert(0, ROOT)
import torch

import dsdf
from bench import synth_grid

dev = torch.device('cuda')
out = {}


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for res in (16, 32, 64, 128, 256, 512):
    g = synth_grid(res, dev) * 1.7 + 0.01
    out[f'redistance_{res}_ms'] = timed(lambda: dsdf.redistance(g), 3)

data = synth_grid(256, dev)
grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
gi = torch.randn(12, 512, 512, 3, device=dev) * 1e-6
gg = torch.zeros_like(data)
for name, sp, sg, integ in (('sil_256_64', 256, 64, 0), ('shade_256_64', 256, 64, 1), ('sil_64_64', 64, 64, 0), ('sil_4_1', 4, 1, 0)):
    p = timed(lambda: dsdf.render_forward(grid, sens, sp, seeds=list(range(12)), integrator=integ), 3)
    b = timed(lambda: dsdf.render_backward(grid, sens, sg, gi, grad_grid=gg, seeds=list(range(12)), integrator=integ), 3)
    out[name] = {'primal_ms': p, 'grad_ms': b, 'renders_per_s': 1e3 / (p + b)}
# BASELINE.json C5: sdf_direct_reparam with a 3-channel 256^3 albedo volume (12 views x 512^2, 256/64 spp)
alb = torch.rand(256, 256, 256, 3, device=dev) * 0.6 + 0.2
galb = torch.zeros_like(alb)
for hide in (True, False):
    sh = dsdf.Shading(alb, 1.0, hide_emitters=hide)
    p = timed(lambda: dsdf.render_forward(grid, sens, 256, seeds=list(range(12)), integrator='sdf_direct_reparam', shading=sh), 2)
    b = timed(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=gg, seeds=list(range(12)), integrator='sdf_direct_reparam',
                                           shading=sh, grad_albedo=galb), 2)
    out['direct_256_64' + ('_hide_emitters' if hide else '')] = {'primal_ms': p, 'grad_ms': b, 'renders_per_s': 1e3 / (p + b)}
del alb, galb
torch.cuda.empty_cache()
# BASELINE.json C4 grid size on one GPU (512^3, 12 views x 512^2, 256/64 spp)
big = synth_grid(512, dev)
gbig = dsdf.SdfGrid(big)
ggb = torch.zeros_like(big)
p = timed(lambda: dsdf.render_forward(gbig, sens, 256, seeds=list(range(12))), 2)
b = timed(lambda: dsdf.render_backward(gbig, sens, 64, gi, grad_grid=ggb, seeds=list(range(12))), 2)
out['sil_256_64_grid512'] = {'primal_ms': p, 'grad_ms': b, 'renders_per_s': 1e3 / (p + b)}
del big, gbig, ggb
torch.cuda.empty_cache()

# one optimisation iteration of a `no-tex-12-hqq`-shaped state at its final resolution (256^3, 512^2, 6 of 12 views)
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'))
import losses, regularizations, redistancing
from shapes import Grid3d
from integrators.reparam import Scene, create_integrator, render, traverse
from constants import SDF_DEFAULT_KEY
from variables import Adam
integ = create_integrator('sdf_silhouette_reparam', {'sdf': Grid3d(data.clone())})
import configs
integ.warp_field = configs.get_config('warp').get_warpfield(integ.sdf)
scene = Scene(sens, integ)
params = traverse(scene); params.keep([SDF_DEFAULT_KEY])
opt = Adam(lr=0.005, params=params)
params.update(opt)
refs = dsdf.render_forward(dsdf.SdfGrid(synth_grid(256, dev, seed=1)), sens, 64, seeds=list(range(12)))


def iteration(i=[0]):
    idx = [(j * 2 + i[0] % 2) % 12 for j in range(6)]
    img = render(scene, params, [sens[k] for k in idx], seed=i[0], spp=256, seed_grad=i[0] + 13, spp_grad=64)
    loss = sum(losses.multiscale_l1(img[j], refs[k]) / 6 for j, k in enumerate(idx))
    loss.backward()
    (1e-5 * regularizations.eval_discrete_laplacian_reg(opt[SDF_DEFAULT_KEY])).backward()
    g = opt[SDF_DEFAULT_KEY].grad
    opt[SDF_DEFAULT_KEY].grad = torch.nan_to_num(g).clamp_(-0.1, 0.1)
    opt.step()
    with torch.no_grad():
        opt[SDF_DEFAULT_KEY] = redistancing.redistance(opt[SDF_DEFAULT_KEY].detach().contiguous())
    params.update(opt)
    i[0] += 1


out['optimize_iteration_256cubed_6views_512sq_ms'] = timed(iteration, 4)
print(json.dumps(out, indent=1))
