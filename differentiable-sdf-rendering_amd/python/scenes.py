"""Target shapes standing in for the reference's scene assets.

The reference loads `scenes/<scene>/<scene>.xml` (meshes, BSDFs, emitters; a separate download
that is not part of the repository, README.md:59-68) and renders the reference images from the
mesh.  Here a scene is just a target SDF volume: `scenes/<scene>/<scene>.vol` if present, a
watertight mesh `scenes/<scene>/<scene>.obj` / `.ply` inside [-0.5, 0.5]^3 converted with
`mesh_to_sdf.create_sdf` (python/mesh_to_sdf.py:9-57), else a procedural shape (`sphere`, `torus`, `blobs`, or -- for any other name such as `dragon` -- a
union of primitives seeded by the scene name), rendered with the same integrator."""
import hashlib
import os

import numpy as np
import torch

from constants import SCENE_DIR
from util import read_vol


def _axes(res, device):
    lin = torch.linspace(0, 1, res, device=device)
    return torch.meshgrid(lin, lin, lin, indexing='ij')


def _blobs(res, seed, device, n=24):
    rng = np.random.default_rng(seed)
    z, y, x = _axes(res, device)
    sd = torch.full((res,) * 3, 1e9, device=device)
    for k in range(n):
        c = rng.uniform(0.3, 0.7, 3)
        if k % 2 == 0:
            sd = torch.minimum(sd, torch.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - rng.uniform(0.05, 0.12))
        else:
            R, r = rng.uniform(0.08, 0.16), rng.uniform(0.015, 0.035)
            q = [x - c[0], y - c[1], z - c[2]]
            ax = k % 3
            o = [a for a in range(3) if a != ax]
            ring = torch.sqrt(q[o[0]] ** 2 + q[o[1]] ** 2) - R
            sd = torch.minimum(sd, torch.sqrt(ring ** 2 + q[ax] ** 2) - r)
    return sd


def load_target_sdf(scene_name, res=128, device='cuda'):
    path = os.path.join(SCENE_DIR, scene_name, f'{scene_name}.vol')
    if os.path.isfile(path):
        return read_vol(path, device)
    for ext in ('.obj', '.ply'):
        mesh = os.path.join(SCENE_DIR, scene_name, scene_name + ext)
        if os.path.isfile(mesh):
            import mesh_to_sdf
            print(f"[scenes] target SDF of '{scene_name}' from {mesh} at {res}^3")
            return mesh_to_sdf.create_sdf(mesh, res, device=device)
    z, y, x = _axes(res, device)
    if scene_name == 'sphere':
        return torch.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) - 0.36
    if scene_name == 'torus':
        ring = torch.sqrt((x - 0.5) ** 2 + (z - 0.5) ** 2) - 0.25
        return torch.sqrt(ring ** 2 + (y - 0.5) ** 2) - 0.09
    seed = 0 if scene_name == 'blobs' else int(hashlib.sha1(scene_name.encode()).hexdigest()[:8], 16)
    print(f"[scenes] no assets for '{scene_name}' under {SCENE_DIR}; using a procedural stand-in (seed {seed})")
    return _blobs(res, seed, device)


def load_target_albedo(scene_name, res=32, device='cuda'):
    """Reflectance volume (Z,Y,X,3) of the target for `sdf_direct_reparam` references:
    `scenes/<scene>/<scene>-albedo.vol` if present, else a smooth procedural colour field in [0.15, 0.9]."""
    path = os.path.join(SCENE_DIR, scene_name, f'{scene_name}-albedo.vol')
    if os.path.isfile(path):
        return read_vol(path, device)
    z, y, x = _axes(res, device)
    ph = (int(hashlib.sha1(scene_name.encode()).hexdigest()[:4], 16) % 628) / 100.0
    r = 0.525 + 0.375 * torch.sin(6.0 * x + ph)
    g = 0.525 + 0.375 * torch.sin(5.0 * y + 2.0 * ph)
    b = 0.525 + 0.375 * torch.sin(7.0 * z + 3.0 * ph)
    return torch.stack([r, g, b], -1).contiguous()
