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
    }[name]
    gridfn, ncam, icam, W, H, spp, seed = cfg
    gen = torch.Generator().manual_seed(seed)
    offsets = torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float32)
    grad_image = torch.randn(H, W, 3, generator=gen, dtype=torch.float32)
    origin = O.regular_camera_origins(ncam)[icam]
    return dict(name=name, grid=gridfn(), ncam=ncam, icam=icam, origin=origin, W=W, H=H, spp=spp,
                offsets=offsets, grad_image=grad_image)


def oracle_forward(case, integrator, reparam=True):
    cam = O.Camera(case['origin'])
    return O.render(O.Grid3d(case['grid']), cam, case['W'], case['H'], case['spp'], case['offsets'].double(),
                    integrator, reparam, return_aux=True)


def oracle_backward(case, integrator, reparam=True):
    cam = O.Camera(case['origin'])
    return O.render_backward(O.Grid3d(case['grid']), cam, case['W'], case['H'], case['spp'], case['offsets'].double(),
                             case['grad_image'].double(), integrator, reparam)
