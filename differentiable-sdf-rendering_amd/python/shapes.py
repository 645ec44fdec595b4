"""SDF shapes of the hot path, mirroring the reference's python/shapes.py protocol on top of
the HIP library: `Grid3d` (python/shapes.py:375-483) plus the grid constructors
`create_sphere_sdf` (:557-581) and the smooth `BoxSDF` (:539-554) used as a constraint.

All tensors are torch HIP tensors; every lookup / trace runs in libdsdf.so.
"""
import numpy as np
import torch

import dsdf
import redistancing


def atleast_4d(t):
    """python/util.py:219-223."""
    return t[..., None] if t.dim() == 3 else t


class Grid3d:
    """Grid-based SDF (python/shapes.py:375-483): a (Z,Y,X[,1]) fp32 tensor interpreted
    as a tricubic B-spline texture over the unit cube.  `transform` is not supported."""

    def __init__(self, data, transform=None):
        if transform is not None:
            raise NotImplementedError("Grid3d(transform=...) is outside the supported path (DESIGN.md section 9)")
        if isinstance(data, str):
            from util import read_vol
            data = redistancing.redistance(read_vol(data))
        self.data = atleast_4d(data)
        self.grid = dsdf.SdfGrid(self.data)
        self.p = torch.zeros(3)
        # SDFBase defaults, python/shapes.py:28-39 (held by the library's dsdf_params)
        self.refine_intersection = True

    # --- texture protocol -------------------------------------------------------------
    def eval(self, x, detached=False):
        return dsdf.eval_cubic(self.grid, x, 0)[0]

    def eval_grad(self, x, detached=False):
        return dsdf.eval_cubic(self.grid, x, 1)[1]

    def eval_and_grad(self, x, detached=False):
        v, g, _ = dsdf.eval_cubic(self.grid, x, 1)
        return v, g

    def eval_all(self, x, detach_w=False):
        """-> (v, v_detached, g, g_detached, H[n,3,3]) like python/shapes.py:438-450."""
        v, g, h = dsdf.eval_cubic(self.grid, x, 2)
        H = torch.stack([torch.stack([h[:, 0], h[:, 3], h[:, 4]], -1),
                         torch.stack([h[:, 3], h[:, 1], h[:, 5]], -1),
                         torch.stack([h[:, 4], h[:, 5], h[:, 2]], -1)], -2)
        return v, v, g, g, H

    def bbox(self, expand=True):
        d = 0.05 if expand else 0.0
        return torch.full((3,), -d), torch.full((3,), 1.0 + d)

    # --- tracing (python/shapes.py:115-339) ---------------------------------------------
    def ray_intersect(self, ray_o, ray_d, maxt, warp=None, active=True, extra_outputs=None):
        """-> (its_t, warp_t, warp_t_d, warp_weight, warp_weight_d); `warp=None` takes the
        non-differentiable path exactly like the reference (:116-118)."""
        out = dsdf.trace(self.grid, ray_o, ray_d, maxt, differentiable=warp is not None)
        if extra_outputs is not None:
            extra_outputs['i'] = out['steps']
        if warp is None:
            z = torch.zeros_like(out['its_t'])
            return out['its_t'], z, torch.zeros_like(ray_o), None, None
        return out['its_t'], out['warp_t'], out['warp_t_d'], out['warp_weight'], out['warp_weight_d']

    def ray_intersect_non_diff(self, ray_o, ray_d, maxt, active=True):
        return self.ray_intersect(ray_o, ray_d, maxt, warp=None)

    # --- parameter plumbing (python/shapes.py:469-483) ----------------------------------
    def traverse(self, callback):
        callback.put_parameter('sdf.data', self.data)
        callback.put_parameter('sdf.p', self.p)

    def set_data(self, data):
        self.data = atleast_4d(data)
        self.grid.update(self.data)

    def update(self):
        self.grid.update(self.data)

    def parameters_changed(self, keys=None):
        self.grid.update(self.data)

    @property
    def shape(self):
        return tuple(self.data.shape)


class BoxSDF:
    """Smooth box SDF (python/shapes.py:539-554; iquilezles.org/articles/distfunctions)."""

    def __init__(self, p, extents, smoothing=0.01):
        self.p, self.extents, self.smoothing = p, extents, smoothing

    def eval(self, x, detached=False):
        q = (x - self.p).abs() - self.extents
        return torch.linalg.norm(q.clamp(min=0.0), dim=-1) + q.max(-1).values.clamp(max=0.0) - self.smoothing


def create_sphere_sdf(res, center=(0.5, 0.5, 0.5), radius=0.3, noise_sigma=0.0, device='cuda'):
    """python/shapes.py:557-581: sphere sampled on linspace(0,1,res) corners, then redistanced."""
    lin = [torch.linspace(0, 1, int(r), device=device) for r in res[:3]]
    z, y, x = torch.meshgrid(*lin, indexing='ij')
    sd = torch.sqrt((x - center[0]) ** 2 + (y - center[1]) ** 2 + (z - center[2]) ** 2) - radius
    if noise_sigma > 0:
        sd = sd + torch.randn_like(sd) * noise_sigma / 4
    return redistancing.redistance(sd.contiguous())


def create_block_sdf(resolution, center=(0.5, 0.5, 0.5), device='cuda'):
    """python/shapes.py:584-590."""
    r2 = resolution // 2
    sd = torch.ones((resolution,) * 3, device=device)
    sd[r2 - r2 // 6:r2 + r2 // 6, r2 - r2 // 6:r2 + r2 // 6, r2 - r2 // 2:r2 + r2 // 2] = -1
    return redistancing.redistance(sd)
