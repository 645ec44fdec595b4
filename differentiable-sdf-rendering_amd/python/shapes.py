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


def _rigid_parts(transform):
    """4x4 `to_world` -> (to_world, A, b) with to_local(x) = A x + b, all float64 numpy.  Accepted: a translation combined with
    an AXIS-ALIGNED rotation (a signed permutation of the axes, det +1).  Then the reference's computation -- march through the
    world AABB of the transformed cube, distances to that AABB in the boundary fade of the trace weights (python/shapes.py:393-403,
    86-87) -- is the same computation in the cube's own frame, which is where the library works.  Any other rotation makes
    the reference's AABB larger than the cube (gradients differ by percents: measured 4.9 % at 25 degrees), scale / shear /
    mirror change the march itself: those raise HERE and Grid3d falls back to the world-space build of the library for them."""
    tw = np.asarray(transform.matrix if hasattr(transform, 'matrix') else transform, np.float64).reshape(4, 4)
    if not np.allclose(tw[3], [0, 0, 0, 1], atol=1e-12):
        raise NotImplementedError("Grid3d(transform=...): projective transforms are not supported")
    R = tw[:3, :3]
    if not np.allclose(R.T @ R, np.eye(3), atol=1e-5) or np.linalg.det(R) < 0:
        raise NotImplementedError("Grid3d(transform=...): scale / shear / mirror are not supported (DESIGN.md section 9)")
    if not np.allclose(np.abs(R), np.round(np.abs(R)), atol=1e-6):
        raise NotImplementedError("Grid3d(transform=...): only axis-aligned rotations (+ translation) reproduce the reference, "
                                  "whose traced box is the world AABB of the rotated cube (DESIGN.md section 9)")
    R = np.round(R)
    tw = tw.copy()
    tw[:3, :3] = R
    A = R.T
    return tw, A, -A @ tw[:3, 3]


class Grid3d:
    """Grid-based SDF (python/shapes.py:375-483): a (Z,Y,X[,1]) fp32 tensor interpreted as a tricubic B-spline texture
    over the unit cube.  `transform` (python/shapes.py:378-403, integrator property `sdf_to_world`): a 4x4 affine `to_world`.
      * translation + axis-aligned rotation (`_rigid_parts`): a change of frame.  The default library works in the cube's own
        frame: points, rays and sensors are taken there with `to_local` (`local_sensor`), scalars (t, weights, images) are
        frame-independent and vectors come back through to_local^T like the reference's gradients (:426-427, 446-448);
      * anything else (a general rotation, a scale): `_world` -- the grid is served by the world-space build of the library
        (dsdf.SdfGrid(to_world=...), lib/variants/libdsdf_xf.so), nothing is mapped on the host."""

    _world = False

    def __init__(self, data, transform=None):
        self.has_transform = transform is not None
        # `_world`: a transform that is NOT a change of frame (any other rotation, a scale).  Such a grid is served by the
        # world-space build of the library (lib/variants/libdsdf_xf.so; dsdf.SdfGrid(to_world=...)): sensors, rays and `sdf.p`
        # stay in world space and the library returns world-space derivatives, exactly the reference's formulation
        # (python/shapes.py:408-450) -- without the per-pixel proofs and the tuned schedule of the default build.
        self._world = False
        if self.has_transform:
            try:
                self.to_world, self._A, self._b = _rigid_parts(transform)
            except NotImplementedError:
                tw = np.asarray(transform.matrix if hasattr(transform, 'matrix') else transform, np.float64).reshape(4, 4)
                if not np.allclose(tw[3], [0, 0, 0, 1], atol=1e-12) or abs(np.linalg.det(tw[:3, :3])) < 1e-12:
                    raise
                self.to_world, self._world = tw, True
                inv = np.linalg.inv(tw)
                self._A, self._b = inv[:3, :3], inv[:3, 3]
        if isinstance(data, str):
            from util import read_vol
            data = redistancing.redistance(read_vol(data))
        self.data = atleast_4d(data)
        self.grid = dsdf.SdfGrid(self.data, to_world=self.to_world if self._world else None)
        self.p = torch.zeros(3)
        # SDFBase defaults, python/shapes.py:28-39 (held by the library's dsdf_params)
        self.refine_intersection = True

    # --- change of frame (python/shapes.py:408-414) -------------------------------------
    def _mat(self, like):
        return torch.as_tensor(self._A, dtype=like.dtype, device=like.device), torch.as_tensor(self._b, dtype=like.dtype, device=like.device)

    def to_local_points(self, x):
        """to_local @ x for (n,3) points.  (`sdf.p` is applied by the library: lookups at x_local - to_local3 @ p.)"""
        if not self.has_transform or self._world:
            return x
        A, b = self._mat(x)
        return x @ A.T + b

    def to_local_vectors(self, d):
        if not self.has_transform or self._world:
            return d
        return d @ self._mat(d)[0].T

    def to_world_covectors(self, g):
        """to_local3^T g: how gradients (and derivatives w.r.t. a world-space direction) come back."""
        if not self.has_transform or self._world:
            return g
        return g @ self._mat(g)[0]

    def local_translation(self, p=None):
        """`sdf.p` in the cube's frame: to_local @ (x - p) = to_local @ x - to_local3 @ p."""
        p = self.p if p is None else p
        if isinstance(p, torch.Tensor):
            # one read-back per (tensor object, in-place version): `_sync` runs before every eval / trace / render call, and a
            # device tensor would otherwise cost a blocking device-to-host sync each time
            c = getattr(self, '_p_read', None)
            if c is None or c[0] is not p or c[1] != p._version:
                c = self._p_read = (p, p._version, np.asarray(p.detach().cpu().tolist(), np.float64))
            vals = c[2]
        else:
            vals = np.asarray(p, np.float64)
        return (self._A @ vals).tolist() if (self.has_transform and not self._world) else vals.tolist()

    def local_sensor(self, sensor):
        """The sensor seen from the cube's frame: a rigid map takes the look-at frame to the look-at frame of the mapped
        origin / target / up, so the camera rays are the mapped camera rays."""
        if not self.has_transform or self._world:
            return sensor
        return dsdf.Sensor(self._A @ sensor.origin + self._b, self._A @ sensor.target + self._b, self._A @ sensor.up,
                           fov=sensor.fov, resx=sensor.resx, resy=sensor.resy)

    def _sync(self):
        """Frame-dependent fields of the library's parameter block: `sdf.p` and the fixed light of
        sdf_simple_shading_reparam (normalize(1,1,1) in WORLD space, sdf_simple_shading_reparam.py:20), both in the cube's frame."""
        self.grid.set_translation(self.local_translation())
        if self.has_transform and not self._world:
            l = self._A @ (np.ones(3) / np.sqrt(3.0))
            for k in range(3):
                self.grid.params.light_dir[k] = float(l[k])

    # --- texture protocol -------------------------------------------------------------
    def eval(self, x, detached=False):
        self._sync()
        return dsdf.eval_cubic(self.grid, self.to_local_points(x), 0)[0]

    def eval_grad(self, x, detached=False):
        self._sync()
        return self.to_world_covectors(dsdf.eval_cubic(self.grid, self.to_local_points(x), 1)[1])

    def eval_and_grad(self, x, detached=False):
        self._sync()
        v, g, _ = dsdf.eval_cubic(self.grid, self.to_local_points(x), 1)
        return v, self.to_world_covectors(g)

    def eval_all(self, x, detach_w=False):
        """-> (v, v_detached, g, g_detached, H[n,3,3]) like python/shapes.py:438-450."""
        self._sync()
        v, g, h = dsdf.eval_cubic(self.grid, self.to_local_points(x), 2)
        H = torch.stack([torch.stack([h[:, 0], h[:, 3], h[:, 4]], -1),
                         torch.stack([h[:, 3], h[:, 1], h[:, 5]], -1),
                         torch.stack([h[:, 4], h[:, 5], h[:, 2]], -1)], -2)
        if self.has_transform and not self._world:
            A = self._mat(g)[0]
            g = g @ A
            H = A.T @ H @ A
        return v, v, g, g, H

    def bbox(self, expand=True):
        """python/shapes.py:393-403, 416-418: AABB of the transformed cube corners (+- 0.05)."""
        d = 0.05 if expand else 0.0
        if self.has_transform:
            c = np.array([[x, y, z] for x in (0.0, 1.0) for y in (0.0, 1.0) for z in (0.0, 1.0)])
            w = c @ self.to_world[:3, :3].T + self.to_world[:3, 3]
            return torch.tensor(w.min(0) - d, dtype=torch.float32), torch.tensor(w.max(0) + d, dtype=torch.float32)
        return torch.full((3,), -d), torch.full((3,), 1.0 + d)

    # --- tracing (python/shapes.py:115-339) ---------------------------------------------
    def ray_intersect(self, ray_o, ray_d, maxt, warp=None, active=True, extra_outputs=None):
        """-> (its_t, warp_t, warp_t_d, warp_weight, warp_weight_d); `warp=None` takes the
        non-differentiable path exactly like the reference (:116-118)."""
        self._sync()
        out = dsdf.trace(self.grid, self.to_local_points(ray_o), self.to_local_vectors(ray_d), maxt,
                         differentiable=warp is not None)
        if extra_outputs is not None:
            extra_outputs['i'] = out['steps']
        if warp is None:
            z = torch.zeros_like(out['its_t'])
            return out['its_t'], z, torch.zeros_like(ray_o), None, None
        return (out['its_t'], out['warp_t'], self.to_world_covectors(out['warp_t_d']), out['warp_weight'],
                self.to_world_covectors(out['warp_weight_d']))

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
