"""TEST INFRASTRUCTURE ONLY -- numpy fp64 restatement of the mesh -> SDF asset path (python/mesh_to_sdf.py:9-57) of the
reference; nothing under differentiable-sdf-rendering_amd/ imports it.

The ray casts themselves belong to Mitsuba 3 (`Scene.ray_intersect`; the reference pins mitsuba 3.x through its
requirements -- a pip dependency, absent from /root/reference), so they are restated from the published algorithm:
closest hit over all triangles, Moeller-Trumbore intersection, geometric normal n = (p1 - p0) x (p2 - p0),
`square_to_uniform_sphere(u, v) = (r cos 2 pi u, r sin 2 pi u, 1 - 2 v)`, r = sqrt(1 - z^2).  PARITY UNPINNED against
Mitsuba itself (not installable here); pinned instead on analytic shapes (tests/test_mesh_to_sdf*.py: sphere / box
distances) and on the reference's call sites (+y occupancy rays :23-27, 16 x 16 refinement directions :40-45, min over
hits x sign :47-55).
"""
import math

import numpy as np


def raycast(tri, o, d, t_min=0.0, block=2048):
    """tri (T,3,3), o/d (n,3) -> t (n,) (inf: miss), backface (n,) bool, margin (n,): how far the winning hit is from being
    decided differently (min over |u|, |v|, |1-u-v| of near-edge candidates and the gap to the runner-up) -- the tests skip rays
    whose closest hit is an fp32 coin toss."""
    tri = np.asarray(tri, np.float64); o = np.asarray(o, np.float64); d = np.asarray(d, np.float64)
    p0 = tri[:, 0]; e1 = tri[:, 1] - p0; e2 = tri[:, 2] - p0
    n = o.shape[0]
    t_best = np.full(n, np.inf); back = np.zeros(n, bool); margin = np.full(n, np.inf)
    for s in range(0, n, block):
        oo = o[s:s + block, None, :]; dd = d[s:s + block, None, :]
        pv = np.cross(dd, e2[None])
        det = (e1[None] * pv).sum(-1)
        safe = np.where(det != 0, det, 1.0)
        tv = oo - p0[None]
        u = (tv * pv).sum(-1) / safe
        qv = np.cross(tv, e1[None])
        v = (dd * qv).sum(-1) / safe
        t = (e2[None] * qv).sum(-1) / safe
        ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > t_min)
        tt = np.where(ok, t, np.inf)
        k = tt.argmin(1)
        r = np.arange(tt.shape[0])
        t_best[s:s + block] = tt[r, k]
        back[s:s + block] = det[r, k] < 0
        # barycentric distance to an edge for every candidate in front of the origin: a tiny one may flip under fp32
        edge = np.minimum(np.minimum(np.abs(u), np.abs(v)), np.abs(1 - u - v))
        infront = (det != 0) & (t > t_min) & np.isfinite(t)
        m = np.where(infront, edge, np.inf).min(1)
        tt2 = tt.copy(); tt2[r, k] = np.inf
        second, first = tt2.min(1), tt[r, k]
        gap = np.where(np.isfinite(second) & np.isfinite(first), np.abs(np.where(np.isfinite(second), second, 0.0) - np.where(np.isfinite(first), first, 0.0)), np.inf)
        margin[s:s + block] = np.minimum(m, gap)
    return t_best, back, margin


def voxel_centres(res):
    c = np.linspace(-0.5 + 0.5 / res, 0.5 - 0.5 / res, res)
    z, y, x = np.meshgrid(c, c, c, indexing='ij')
    return np.stack([x.ravel(), y.ravel(), z.ravel()], -1)


def sphere_directions(angular_res=16):
    r = (np.arange(angular_res) + 0.5) / angular_res
    v, u = np.meshgrid(r, r, indexing='ij')
    u, v = u.ravel(), v.ravel()
    z = 1.0 - 2.0 * v
    rad = np.sqrt(np.maximum(1.0 - z * z, 0.0))
    return np.stack([rad * np.cos(2 * math.pi * u), rad * np.sin(2 * math.pi * u), z], -1)


def occupancy(tri, res):
    o = voxel_centres(res)
    d = np.zeros_like(o); d[:, 1] = 1.0
    t, back, margin = raycast(tri, o, d)
    return (0.5 - (np.isfinite(t) & back).astype(np.float64)).reshape(res, res, res), margin.reshape(res, res, res)


def create_sdf(tri, res, redistance, refine_surface=True):
    """`redistance`: callable on a (res,res,res) array (the C oracle's fast sweeping, oracle/c_oracle.py:redistance)."""
    values, _ = occupancy(tri, res)
    grid = np.asarray(redistance(values.astype(np.float32)), np.float64)
    if refine_surface:
        flat = grid.reshape(-1).copy()
        near = np.nonzero(np.abs(flat) < 1.0 / res)[0]
        dirs = sphere_directions()
        o = voxel_centres(res)[near]
        md = np.full(near.size, 100.0)
        for j in range(dirs.shape[0]):
            t, _, _ = raycast(tri, o, np.broadcast_to(dirs[j], o.shape))
            md = np.minimum(md, t)
        flat[near] = md * np.sign(flat[near])
        grid = np.asarray(redistance(flat.reshape(res, res, res).astype(np.float32)), np.float64)
    return grid


# ---- procedural watertight test meshes (outward-facing counter-clockwise triangles) --------------------------------

def icosphere(radius=0.3, subdiv=2, centre=(0.0, 0.0, 0.0)):
    p = (1 + 5 ** 0.5) / 2
    v = [(-1, p, 0), (1, p, 0), (-1, -p, 0), (1, -p, 0), (0, -1, p), (0, 1, p), (0, -1, -p), (0, 1, -p), (p, 0, -1), (p, 0, 1), (-p, 0, -1), (-p, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(x, np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m)); cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    v = np.asarray(v) * radius + np.asarray(centre)
    return v.astype(np.float32), np.asarray(f, np.int64)


def box(half=(0.3, 0.2, 0.25), centre=(0.0, 0.0, 0.0)):
    h = np.asarray(half); c = np.asarray(centre)
    v = np.asarray([[sx, sy, sz] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)], np.float64) * h + c
    q = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]      # outward quads
    f = []
    for a, b, c_, d in q:
        f += [(a, b, c_), (a, c_, d)]
    return v.astype(np.float32), np.asarray(f, np.int64)


def write_obj(fn, v, f):
    with open(fn, 'w') as fh:
        fh.write('# procedural test mesh\n')
        for p in v:
            fh.write(f'v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n')
        for t in f:
            fh.write(f'f {t[0] + 1}//{t[0] + 1} {t[1] + 1}//{t[1] + 1} {t[2] + 1}//{t[2] + 1}\n')


def write_ply(fn, v, f, binary=True):
    with open(fn, 'wb') as fh:
        fh.write(('ply\nformat %s 1.0\ncomment procedural test mesh\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n'
                  'element face %d\nproperty list uchar int vertex_indices\nend_header\n' %
                  ('binary_little_endian' if binary else 'ascii', len(v), len(f))).encode())
        if binary:
            fh.write(np.asarray(v, '<f4').tobytes())
            for t in f:
                fh.write(bytes([3]) + np.asarray(t, '<i4').tobytes())
        else:
            for p in v:
                fh.write(f'{p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n'.encode())
            for t in f:
                fh.write(f'3 {t[0]} {t[1]} {t[2]}\n'.encode())
