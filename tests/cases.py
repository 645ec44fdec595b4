"""Seeded test cases shared by the CPU and GPU parity tests."""
import numpy as np
import torch

import sdf_oracle as O


def make_case(name):
    """Returns dict(grid fp64 (Z,Y,X), cam index/total, W, H, spp, offsets fp32 (n,2), grad_image fp32)."""
    cfg = {
        # name: (grid fn, n_cams, cam_idx, W, H, spp, seed)
        'sphere16': (lambda: O.sphere_grid(16), 1, 0, 16, 16, 4, 1),
        'blob32': (lambda: O.blob_grid(32, n=6, seed=1), 3, 1, 24, 24, 8, 2),
        'blob32_spp64': (lambda: O.blob_grid(32, n=6, seed=1), 3, 2, 12, 12, 64, 3),
        'blob48_rect': (lambda: O.blob_grid(48, n=10, seed=3), 12, 5, 32, 20, 4, 4),
        # three 64-sample chunks per pixel (not a power of two: work-list tiles, chunk -> unit arithmetic)
        'blob32_spp192': (lambda: O.blob_grid(32, n=6, seed=1), 3, 2, 6, 6, 192, 5),
        # 2 spp: 8 x 4 pixel-tile waves of the general pass, film window in LDS, ragged tiles at the right / bottom edge
        'blob32_spp2': (lambda: O.blob_grid(32, n=6, seed=1), 3, 2, 21, 13, 2, 6),
    }[name]
    gridfn, ncam, icam, W, H, spp, seed = cfg
    gen = torch.Generator().manual_seed(seed)
    offsets = torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float32)
    grad_image = torch.randn(H, W, 3, generator=gen, dtype=torch.float32)
    origin = O.regular_camera_origins(ncam)[icam]
    # the oracle's sensor is the one DEFINED by the fp32 record the C-ABI receives (bit-identical inputs on both sides)
    cam = O.Camera(origin).rounded()
    return dict(name=name, grid=gridfn(), ncam=ncam, icam=icam, origin=origin, cam=cam, W=W, H=H, spp=spp,
                offsets=offsets, grad_image=grad_image)


def oracle_forward(case, integrator, reparam=True):
    cam = case['cam']
    return O.render(O.Grid3d(case['grid']), cam, case['W'], case['H'], case['spp'], case['offsets'].double(),
                    integrator, reparam, return_aux=True)


def oracle_backward(case, integrator, reparam=True):
    cam = case['cam']
    return O.render_backward(O.Grid3d(case['grid']), cam, case['W'], case['H'], case['spp'], case['offsets'].double(),
                             case['grad_image'].double(), integrator, reparam)


def direct_inputs(case, ares=(6, 5, 4), seed=11):
    """Extra inputs of sdf_direct_reparam for a case: albedo volume (Z,Y,X,3) fp32 in [0.2, 0.8], per-lane
    emitter samples (n,2) fp32, environment radiance."""
    gen = torch.Generator().manual_seed(seed)
    n = case['offsets'].shape[0]
    albedo = torch.rand(*ares, 3, generator=gen, dtype=torch.float32) * 0.6 + 0.2
    emitter_u = torch.rand(n, 2, generator=gen, dtype=torch.float32)
    return dict(albedo=albedo, emitter_u=emitter_u, env=(1.0, 0.9, 0.8))


def oracle_direct(case, extra, reparam=True, hide_emitters=False, grads=False, p=None):
    """Oracle image of sdf_direct_reparam; with grads=True also (dL/d data, dL/d albedo[, dL/d p]) for
    L = sum(image * grad_image)."""
    cam = case['cam']
    data = case['grid'].clone().requires_grad_(grads)
    alb = extra['albedo'].double().clone().requires_grad_(grads)
    env = torch.tensor(extra['env'], dtype=torch.float64)
    img = O.render(O.Grid3d(data, p), cam, case['W'], case['H'], case['spp'], case['offsets'].double(), O.DIRECT, reparam,
                   albedo=alb, emitter_u=extra['emitter_u'].double(), env=env, hide_emitters=hide_emitters)
    if not grads:
        return img
    (img * case['grad_image'].double()).sum().backward()
    return img.detach(), data.grad, alb.grad
