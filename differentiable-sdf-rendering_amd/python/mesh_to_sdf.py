"""`create_sdf(mesh_fn, resolution, refine_surface=True)` (python/mesh_to_sdf.py:9-57): watertight mesh -> SDF grid by
ray casting and redistancing.  The reference takes the ray casts from Mitsuba (`scene.ray_intersect` on an obj / ply
shape) and the redistancing from `fastsweep`; here both are libdsdf.so entry points (`dsdf_mesh_raycast`,
`dsdf_redistance`).  Returns a (res, res, res) float32 CUDA tensor indexed [z, y, x] over [0,1]^3 voxel centres shifted
to [-0.5, 0.5]^3, exactly the reference's placement (:20-22: the mesh is expected inside that cube).
"""
import math
import struct

import numpy as np
import torch

import dsdf
import redistancing

ANGULAR_RES = 16            # mesh_to_sdf.py:40: 16 x 16 stratified directions per near-surface voxel


def load_obj(fn):
    """Vertices / faces of a Wavefront obj (positions only; polygons are fan-triangulated, negative indices resolved)."""
    v, f = [], []
    with open(fn) as fh:
        for line in fh:
            s = line.split()
            if not s:
                continue
            if s[0] == 'v':
                v.append([float(s[1]), float(s[2]), float(s[3])])
            elif s[0] == 'f':
                idx = []
                for tok in s[1:]:
                    k = int(tok.split('/')[0])
                    idx.append(k - 1 if k > 0 else len(v) + k)
                for j in range(1, len(idx) - 1):
                    f.append([idx[0], idx[j], idx[j + 1]])
    return np.asarray(v, np.float32).reshape(-1, 3), np.asarray(f, np.int64).reshape(-1, 3)


_PLY_TYPES = {'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2', 'ushort': 'u2', 'uint16': 'u2',
              'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4', 'float': 'f4', 'float32': 'f4', 'double': 'f8', 'float64': 'f8'}


def load_ply(fn):
    """Vertices / faces of a ply file (ascii, binary_little_endian or binary_big_endian; x y z + one index list per face)."""
    with open(fn, 'rb') as fh:
        if fh.readline().strip() != b'ply':
            raise ValueError(f'{fn}: not a ply file')
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f'{fn}: truncated ply header')
            s = line.decode('ascii', 'replace').split()
            if not s or s[0] == 'comment':
                continue
            if s[0] == 'format':
                fmt = s[1]
            elif s[0] == 'element':
                elements.append((s[1], int(s[2]), []))
            elif s[0] == 'property':
                elements[-1][2].append(tuple(s[1:]))
            elif s[0] == 'end_header':
                break
        if fmt not in ('ascii', 'binary_little_endian', 'binary_big_endian'):
            raise ValueError(f'{fn}: unsupported ply format {fmt}')
        en = '>' if fmt == 'binary_big_endian' else '<'
        verts, faces = None, []
        tokens = None
        if fmt == 'ascii':
            tokens = iter(fh.read().split())
        for name, count, props in elements:
            has_list = any(p[0] == 'list' for p in props)
            if fmt == 'ascii':
                rows = []
                for _ in range(count):
                    row = []
                    for p in props:
                        if p[0] == 'list':
                            k = int(next(tokens))
                            row.append([int(float(next(tokens))) for _ in range(k)])
                        else:
                            row.append(float(next(tokens)))
                    rows.append(row)
            elif not has_list:
                dt = np.dtype([(p[1], en + _PLY_TYPES[p[0]]) for p in props])
                arr = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt, count=count)
                rows = arr
            else:
                rows = []
                for _ in range(count):
                    row = []
                    for p in props:
                        if p[0] == 'list':
                            ct, it = np.dtype(en + _PLY_TYPES[p[1]]), np.dtype(en + _PLY_TYPES[p[2]])
                            k = int(np.frombuffer(fh.read(ct.itemsize), ct)[0])
                            row.append(np.frombuffer(fh.read(it.itemsize * k), it).astype(np.int64).tolist())
                        else:
                            t = np.dtype(en + _PLY_TYPES[p[0]])
                            row.append(float(np.frombuffer(fh.read(t.itemsize), t)[0]))
                    rows.append(row)
            if name == 'vertex':
                names = [p[-1] for p in props]
                if isinstance(rows, np.ndarray):
                    verts = np.stack([rows['x'], rows['y'], rows['z']], -1).astype(np.float32)
                else:
                    ix = [names.index(c) for c in 'xyz']
                    verts = np.asarray([[r[i] for i in ix] for r in rows], np.float32).reshape(-1, 3)
            elif name == 'face':
                li = [i for i, p in enumerate(props) if p[0] == 'list'][0]
                for r in rows:
                    idx = r[li]
                    for j in range(1, len(idx) - 1):
                        faces.append([idx[0], idx[j], idx[j + 1]])
        if verts is None:
            raise ValueError(f'{fn}: no vertex element')
    return verts, np.asarray(faces, np.int64).reshape(-1, 3)


def load_mesh(mesh_fn):
    """(T, 3, 3) float32 triangle corners; the plugin is chosen the way the reference does (mesh_to_sdf.py:12)."""
    v, f = load_obj(mesh_fn) if mesh_fn.endswith('.obj') else load_ply(mesh_fn)
    if len(f) == 0:
        raise ValueError(f'{mesh_fn}: no faces')
    return v[f]


def voxel_centres(res, device):
    """(res^3, 3) points (x, y, z), x fastest -- mesh_to_sdf.py:20-22."""
    c = torch.linspace(-0.5 + 0.5 / res, 0.5 - 0.5 / res, res, dtype=torch.float32, device=device)
    z, y, x = torch.meshgrid(c, c, c, indexing='ij')
    return torch.stack([x.reshape(-1), y.reshape(-1), z.reshape(-1)], -1).contiguous()


def sphere_directions(device, angular_res=ANGULAR_RES):
    """The angular_res^2 stratified directions of mesh_to_sdf.py:40-45 (`mi.warp.square_to_uniform_sphere` of cell centres)."""
    r = (torch.arange(angular_res, dtype=torch.float32, device=device) + 0.5) / angular_res
    v, u = torch.meshgrid(r, r, indexing='ij')                   # dr.meshgrid default: the first argument varies fastest
    u, v = u.reshape(-1), v.reshape(-1)
    zc = 1.0 - 2.0 * v
    rad = torch.sqrt(torch.clamp(1.0 - zc * zc, min=0.0))
    phi = 2.0 * math.pi * u
    return torch.stack([rad * torch.cos(phi), rad * torch.sin(phi), zc], -1).contiguous()


def occupancy(triangles, res):
    """0.5 - inside, inside = the +y ray from the voxel centre leaves the solid through its first hit (mesh_to_sdf.py:23-27).
    The reference casts that one ray with Embree / OptiX, whose intersectors are edge-consistent; dsdf_mesh_raycast tests every
    triangle on its own (Moeller-Trumbore), so a ray through a shared edge or vertex -- voxel centres sit on the symmetry planes
    of many meshes -- can slip between two triangles.  The -y ray answers the same question (a watertight mesh is left
    through a back face in every direction); where the two disagree a third, generic direction decides."""
    dev = triangles.device
    o = voxel_centres(res, dev)

    def inside_along(direction):
        d = torch.tensor(direction, dtype=torch.float32, device=dev).expand_as(o).contiguous()
        t, back = dsdf.mesh_raycast(triangles, o, d)
        return torch.isfinite(t) & (back != 0)
    up, down = inside_along((0.0, 1.0, 0.0)), inside_along((0.0, -1.0, 0.0))
    inside = up
    split = up != down
    if bool(split.any()):
        n = (0.2718 ** 2 + 0.5772 ** 2 + 0.7071 ** 2) ** 0.5
        inside = torch.where(split, inside_along((0.2718 / n, 0.5772 / n, 0.7071 / n)), up)
    return (0.5 - inside.float()).reshape(res, res, res), o


def refine(triangles, grid, origins, res, chunk=1 << 16):
    """Near-surface voxels (|phi| < 1/res) take the minimum hit distance over the sphere directions, signed by the
    redistanced occupancy (mesh_to_sdf.py:32-55).  No hit in any direction keeps the reference's 100.0."""
    flat = grid.reshape(-1).clone()
    near = torch.nonzero(flat.abs() < 1.0 / res).reshape(-1)
    dirs = sphere_directions(flat.device)
    nd = dirs.shape[0]
    for s in range(0, near.numel(), chunk):
        idx = near[s:s + chunk]
        o = origins[idx].repeat_interleave(nd, 0)
        d = dirs.repeat(idx.numel(), 1)
        t, _ = dsdf.mesh_raycast(triangles, o, d)
        md = torch.clamp(t.reshape(-1, nd).min(1).values, max=100.0)
        flat[idx] = torch.where(flat[idx] < 0, -md, md)          # dr.sign(0) = +1
    return flat.reshape(res, res, res)


def create_sdf(mesh_fn, resolution, refine_surface=True, device='cuda'):
    """Convert a watertight mesh to an SDF using ray casting and redistancing.  `mesh_fn`: path of an obj / ply file, or a
    (T, 3, 3) tensor / array of triangle corners."""
    tri = load_mesh(mesh_fn) if isinstance(mesh_fn, str) else mesh_fn
    tri = torch.as_tensor(np.asarray(tri) if not torch.is_tensor(tri) else tri, dtype=torch.float32).to(device).reshape(-1, 3, 3).contiguous()
    res = int(resolution)
    values, origins = occupancy(tri, res)
    grid = redistancing.redistance(values)
    if refine_surface:
        grid = redistancing.redistance(refine(tri, grid, origins, res))
    return grid
