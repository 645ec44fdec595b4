"""CPU side of the mesh -> SDF asset path (python/mesh_to_sdf.py:9-57): the obj / ply readers of the host module, and the
numpy oracle (oracle/mesh_oracle.py) pinned on analytic shapes -- Mitsuba's ray caster, which the reference calls, is a
pip dependency that is not installable here."""
import numpy as np
import pytest

import c_oracle
import mesh_oracle as M


def test_loaders_round_trip(tmp_path, built):
    import mesh_to_sdf
    v, f = M.icosphere(0.3, 1)
    M.write_obj(str(tmp_path / 'a.obj'), v, f)
    M.write_ply(str(tmp_path / 'b.ply'), v, f, binary=True)
    M.write_ply(str(tmp_path / 'c.ply'), v, f, binary=False)
    ref = v[f]
    for fn in ('a.obj', 'b.ply', 'c.ply'):
        tri = mesh_to_sdf.load_mesh(str(tmp_path / fn))
        assert tri.shape == ref.shape and tri.dtype == np.float32
        np.testing.assert_allclose(tri, ref, rtol=0, atol=1e-7)


def test_obj_polygons_and_negative_indices(tmp_path, built):
    import mesh_to_sdf
    (tmp_path / 'q.obj').write_text('v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4 -3 -2\n')
    tri = mesh_to_sdf.load_mesh(str(tmp_path / 'q.obj'))
    assert tri.shape == (3, 3, 3)
    np.testing.assert_array_equal(tri[1], np.asarray([[0, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32))
    np.testing.assert_array_equal(tri[2], tri[0])
    (tmp_path / 'e.obj').write_text('v 0 0 0\n')
    with pytest.raises(ValueError):
        mesh_to_sdf.load_mesh(str(tmp_path / 'e.obj'))


def test_direction_set_matches_host(built):
    import mesh_to_sdf
    d = mesh_to_sdf.sphere_directions('cpu').numpy()
    ref = M.sphere_directions()
    assert d.shape == (256, 3)
    np.testing.assert_allclose(d, ref, atol=2e-7)
    np.testing.assert_allclose(np.linalg.norm(ref, axis=1), 1.0, atol=1e-12)
    assert np.abs(ref.mean(0)).max() < 1e-12                          # stratified: the set is balanced
    np.testing.assert_allclose(mesh_to_sdf.voxel_centres(5, 'cpu').numpy(), M.voxel_centres(5), atol=1e-7)


def test_oracle_raycast_box_is_analytic():
    v, f = M.box(half=(0.3, 0.2, 0.25))
    rng = np.random.default_rng(3)
    o = rng.uniform(-0.19, 0.19, (500, 3))
    d = rng.normal(size=(500, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, back, _ = M.raycast(v[f], o, d)
    h = np.asarray([0.3, 0.2, 0.25])
    t_ref = np.min(np.where(d > 0, (h - o) / d, (-h - o) / d), axis=1)    # slab exit from inside
    np.testing.assert_allclose(t, t_ref, rtol=1e-6)                       # box corners are float32-rounded
    assert back.all()                                                     # leaving the solid: dot(n, d) > 0
    o2 = o + np.asarray([2.0, 0, 0]); d2 = np.tile([-1.0, 0, 0], (500, 1))
    t2, back2, _ = M.raycast(v[f], o2, d2)
    np.testing.assert_allclose(t2, o2[:, 0] - 0.3, rtol=1e-6)
    assert not back2.any()
    t3, _, _ = M.raycast(v[f], o2, -d2)
    assert np.isinf(t3).all()


def test_oracle_create_sdf_box_is_analytic(built):
    lib = c_oracle.load()
    res = 24
    h = np.asarray([0.3, 0.2, 0.25])
    v, f = M.box(half=h)
    grid = M.create_sdf(v[f], res, lambda p: c_oracle.redistance(lib, p))
    x = M.voxel_centres(res)
    q = np.abs(x) - h
    ref = (np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(1), 0)).reshape(res, res, res)
    assert ((grid < 0) == (ref < 0)).all()
    near = np.abs(ref) < 1.0 / res
    qs = np.sort(q, 1)
    face = (qs[:, 1] < -2.0 / res).reshape(res, res, res)                 # closest surface point well inside a face
    coarse = M.create_sdf(v[f], res, lambda p: c_oracle.redistance(lib, p), refine_surface=False)
    assert np.abs(grid - ref)[near & face].max() < 1e-6                   # refined values on both sides of a plane -> the crossing is exact
    assert np.abs(coarse - ref)[near & face].max() > 0.005                # occupancy alone: half a voxel off
    # edges / corners: the closing redistance re-initialises from axis-aligned crossings, first order (the algorithm's own error)
    assert np.abs(grid - ref).max() < 2.0 / res


def test_ply_big_endian_and_extra_properties(tmp_path, built):
    """binary_big_endian, per-vertex normals / colours before and after x y z, uint8 list counts with uint32 indices, quads."""
    import struct
    import mesh_to_sdf
    v = np.asarray([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    faces = [[0, 1, 2, 3], [0, 1, 4]]
    with open(tmp_path / 'be.ply', 'wb') as fh:
        fh.write(b'ply\nformat binary_big_endian 1.0\ncomment test\nelement vertex 5\nproperty uchar red\nproperty float x\n'
                 b'property float y\nproperty float z\nproperty double quality\nelement face 2\n'
                 b'property list uchar uint vertex_indices\nend_header\n')
        for p in v:
            fh.write(struct.pack('>Bfffd', 7, p[0], p[1], p[2], 0.25))
        for f in faces:
            fh.write(struct.pack('>B', len(f)) + struct.pack('>%dI' % len(f), *f))
    tri = mesh_to_sdf.load_mesh(str(tmp_path / 'be.ply'))
    assert tri.shape == (3, 3, 3)
    np.testing.assert_array_equal(tri[0], v[[0, 1, 2]])
    np.testing.assert_array_equal(tri[1], v[[0, 2, 3]])
    np.testing.assert_array_equal(tri[2], v[[0, 1, 4]])
    (tmp_path / 'bad.ply').write_bytes(b'plx\n')
    with pytest.raises(ValueError):
        mesh_to_sdf.load_mesh(str(tmp_path / 'bad.ply'))


def test_oracle_reproduces_mesh_golden(built):
    """tests/golden/mesh16.npz (make_golden.py) pins the mesh oracle against silent changes."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mesh16.npz'))
    lib = c_oracle.load()
    for tag in ('box', 'ico'):
        sdf = M.create_sdf(g[f'{tag}_tri'], 16, lambda p: c_oracle.redistance(lib, p))
        np.testing.assert_allclose(sdf, g[f'{tag}_sdf'], atol=1e-6)
    t, back, _ = M.raycast(g['box_tri'], g['ray_o'], g['ray_d'])
    np.testing.assert_allclose(t, g['ray_t'], rtol=1e-12)
    np.testing.assert_array_equal(back, g['ray_back'])
