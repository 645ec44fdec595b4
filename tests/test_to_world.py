"""`Grid3d(data, transform)` / integrator property `sdf_to_world` (/root/reference/python/shapes.py:378-450,
integrators/reparam.py:21-29).

The oracle restates the reference literally: lookups at to_local @ (x - p), gradients back through to_local^T, Hessians
through to_local^T H to_local, the traced box = AABB of the transformed corners -- and is pinned to the reference's OWN code run
with a general and an axis-aligned `to_world` (`tf_*` keys of tests/golden/refshim_sphere16.npz).  The product has two routes:
translation + axis-aligned rotation is a change of frame served by the default library (sensors / rays / sdf.p mapped by to_local);
any other affine transform goes to the world-space build of the same sources (lib/variants/libdsdf_xf.so, -DDSDF_XF=1).  Both are
checked against the reference-code fixtures on the host build of the kernel arithmetic and on the GPU.
"""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import make_case

FWD_TOL = 1e-4


def rot(axis, deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    R = np.eye(3)
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R


def about_centre(R, shift=(0.0, 0.0, 0.0)):
    """4x4 to_world: rotate the unit cube about its centre, then translate."""
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.array([0.5, 0.5, 0.5]) - R @ np.array([0.5, 0.5, 0.5]) + np.asarray(shift)
    return T


AXIS_ALIGNED = about_centre(rot(1, 90) @ rot(0, 180), (0.02, -0.015, 0.01))
GENERAL = about_centre(rot(1, 25) @ rot(2, -10), (0.01, 0.0, -0.02))


def local_camera(T, origin):
    A = T[:3, :3].T
    b = -A @ T[:3, 3]
    return O.Camera(A @ np.asarray(origin) + b, A @ np.array([0.5, 0.5, 0.5]) + b, A @ np.array([0.0, 1.0, 0.0]))


def test_oracle_transform_is_a_change_of_frame():
    """Axis-aligned rigid transform: rendering the transformed SDF with the world camera == rendering the plain SDF with the
    camera taken to the cube's frame -- image and dL/d(data) -- inside the fp64 oracle."""
    case = make_case('sphere16')
    data = O.blob_grid(16, n=4, seed=5)
    offs = case['offsets'].double()
    W, H, spp = case['W'], case['H'], case['spp']
    p = torch.tensor([0.01, -0.02, 0.015], dtype=torch.float64)
    A = torch.tensor(AXIS_ALIGNED[:3, :3].T)
    light = A @ (torch.ones(3, dtype=torch.float64) / 3.0 ** 0.5)       # the world-space light of simple shading, in the cube's frame
    gi = case['grad_image'].double()
    for integ in (O.SILHOUETTE, O.SIMPLE_SHADING):
        da = data.clone().requires_grad_(True)
        db = data.clone().requires_grad_(True)
        a = O.render(O.Grid3d(da, p, AXIS_ALIGNED), O.Camera(case['origin']), W, H, spp, offs, integ, True)
        b = O.render(O.Grid3d(db, A @ p), local_camera(AXIS_ALIGNED, case['origin']), W, H, spp, offs, integ, True, light_dir=light)
        assert float((a - b).abs().max()) < 1e-9
        (a * gi).sum().backward()
        (b * gi).sum().backward()
        assert float(db.grad.abs().max()) > 0
        assert float((da.grad - db.grad).abs().max()) <= 1e-8 * float(db.grad.abs().max())


def test_oracle_transform_gradients_follow_the_reference():
    """eval_grad / eval_all of the transformed grid are the derivatives of eval w.r.t. the WORLD point (autograd)."""
    data = O.blob_grid(16, n=4, seed=5)
    sdf = O.Grid3d(data, torch.tensor([0.01, 0.0, -0.01], dtype=torch.float64), GENERAL)
    x = (torch.rand(50, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(3)) * 0.6 + 0.2).requires_grad_(True)
    v, _, g, _, Hm = sdf.eval_all(x)
    ga = torch.autograd.grad(v.sum(), x, create_graph=True)[0]
    assert torch.allclose(g, ga, atol=1e-10)
    for k in range(3):
        hk = torch.autograd.grad(ga[:, k].sum(), x, retain_graph=True)[0]
        assert torch.allclose(Hm[:, k, :], hk, atol=1e-8)
    lo, hi = sdf.bbox()
    assert float(lo.min()) < -0.05 and float(hi.max()) > 1.05           # a rotated cube's AABB is larger than the cube


def _ref16():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'refshim_sphere16.npz'))


@pytest.mark.parametrize('key', ['tf_general', 'tf_axis'])
def test_oracle_transform_matches_reference_code(key):
    """The oracle's `Grid3d(data, p, to_world)` against the REFERENCE'S OWN python/shapes.py with a transform (run on the torch
    stand-in, tools/make_reference_fixtures.py --shim: the `tf_*` keys of tests/golden/refshim_sphere16.npz): eval_all, per-ray
    outputs of ray_intersect, image and gradients of both scene-free integrators -- for a rotation that is NOT axis-aligned (the
    traced box is the world AABB of the rotated cube, shapes.py:393-403) and for the axis-aligned one the product accepts.  This
    is what the product's transform test below is measured against."""
    from test_refshim_fixture import inputs
    ref = _ref16()
    x = inputs(ref)
    assert np.allclose(ref['tf_general_matrix'], GENERAL) and np.allclose(ref['tf_axis_matrix'], AXIS_ALIGNED)
    M, p0 = ref[f'{key}_matrix'], torch.from_numpy(ref['tf_p'])
    sdf = O.Grid3d(x['grid'], p0, M)
    v, _, g, _, Hm = sdf.eval_all(torch.from_numpy(ref['eval_pts']).double())
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a) - b) / np.linalg.norm(b))
    assert rel(v.numpy(), ref[f'{key}_eval_v']) < 1e-12 and rel(g.numpy(), ref[f'{key}_eval_g']) < 1e-12 and rel(Hm.numpy(), ref[f'{key}_eval_H']) < 1e-12
    o, d, maxt = (torch.from_numpy(ref[k]).double() for k in ('ray_o', 'ray_d', 'ray_maxt'))
    tr = O.ray_intersect(sdf, o, d, maxt)
    hit, fin = np.isfinite(ref[f'{key}_ri_its_t']), np.isfinite(ref[f'{key}_ri_warp_t'])
    assert hit.sum() > 50 and fin.sum() > 200
    assert np.array_equal(np.isfinite(tr['its_t'].numpy()), hit) and np.array_equal(np.isfinite(tr['warp_t'].numpy()), fin)
    assert rel(tr['its_t'].numpy()[hit], ref[f'{key}_ri_its_t'][hit]) < 1e-12
    for k, tol in (('warp_t', 1e-12), ('warp_weight', 1e-11), ('warp_t_d', 1e-9), ('warp_weight_d', 1e-11)):
        assert rel(tr[k].numpy()[fin], ref[f'{key}_ri_{k}'][fin]) < tol, k
    for tag, integ in (('sil', O.SILHOUETTE), ('shade', O.SIMPLE_SHADING)):
        data, p = x['grid'].clone().requires_grad_(True), p0.clone().requires_grad_(True)
        img = O.render(O.Grid3d(data, p, M), x['cam'], x['W'], x['H'], x['spp'], x['offs'], integ, True)
        (img * x['gi']).sum().backward()
        assert rel(img.detach().numpy(), ref[f'{key}_img_{tag}']) < 1e-12
        assert rel(data.grad.numpy(), ref[f'{key}_grad_{tag}']) < 1e-9 and rel(p.grad.numpy(), ref[f'{key}_gradp_{tag}']) < 1e-9
    # the transform matters: neither image is the un-transformed one
    assert rel(ref[f'{key}_img_sil'], ref['img_sil']) > 1e-3


def test_world_space_build_and_grid_state(built):
    """lib/variants/libdsdf_xf.so loads, reports the transform capability the default build lacks, and dsdf.SdfGrid turns a
    `to_world` into to_local + world AABB the way Grid3d.__init__ / update_bbox do (python/shapes.py:390-403).  No device needed."""
    import ctypes as C
    import dsdf
    from dsdf import _lib
    built.build_variant('xf', built.VARIANTS['xf'])
    xf, default = _lib.load_xf(), _lib.load()
    assert xf.dsdf_has_grid_transform() == 1 and default.dsdf_has_grid_transform() == 0
    assert xf.dsdf_version() == default.dsdf_version()
    z = (C.c_float * 12)(*([0.0] * 12))
    assert default.dsdf_set_grid_transform(z, z, z, None) != 0 and b'DSDF_XF' in default.dsdf_last_error()
    g = dsdf.SdfGrid.__new__(dsdf.SdfGrid)
    g.transform = None
    g.set_to_world(GENERAL)
    tl, lo, hi = (np.array(list(a), np.float64) for a in g.transform)
    inv = np.linalg.inv(GENERAL)
    assert np.allclose(tl.reshape(3, 4), inv[:3, :], atol=1e-6)
    osdf = O.Grid3d(O.sphere_grid(8), None, GENERAL)
    blo, bhi = osdf.bbox()                                          # (the oracle's box carries the 0.05 expansion)
    assert np.allclose(lo - 0.05, blo.numpy(), atol=1e-6) and np.allclose(hi + 0.05, bhi.numpy(), atol=1e-6)
    with pytest.raises(dsdf.DsdfError):
        g.set_to_world(np.diag([1.0, 1.0, 1.0, 2.0]))


def test_rigid_parts_and_local_sensor():
    import shapes
    import dsdf
    with pytest.raises(NotImplementedError):
        shapes._rigid_parts(np.diag([2.0, 2.0, 2.0, 1.0]))
    with pytest.raises(NotImplementedError):
        shapes._rigid_parts(np.diag([1.0, -1.0, 1.0, 1.0]))
    with pytest.raises(NotImplementedError):
        shapes._rigid_parts(GENERAL)                      # the reference's box is the world AABB: not the same computation
    tw, A, b = shapes._rigid_parts(AXIS_ALIGNED)
    assert np.allclose(A @ tw[:3, :3], np.eye(3)) and np.allclose(A @ tw[:3, 3] + b, 0)
    # the look-at frame of the mapped sensor is the mapped look-at frame
    s = dsdf.get_regular_cameras(5, resx=20, resy=12)[3]
    g = shapes.Grid3d.__new__(shapes.Grid3d)
    g.has_transform, g.to_world, g._A, g._b = True, tw, A, b
    ls = g.local_sensor(s)
    for u, v in zip(s.frame(), ls.frame()):
        assert np.allclose(A @ u, v, atol=1e-12)
    assert np.allclose(ls.origin, A @ s.origin + b) and (ls.resx, ls.resy, ls.fov) == (s.resx, s.resy, s.fov)
    assert np.allclose(g.local_translation([0.1, 0.2, 0.3]), A @ np.array([0.1, 0.2, 0.3]))


@pytest.mark.gpu
@pytest.mark.parametrize('T,exact', [(AXIS_ALIGNED, True), (about_centre(rot(2, -90), (0.0, 0.03, 0.0)), True)])
def test_transformed_grid_render_gpu(built, T, exact):
    """The HIP path with `Grid3d(data, transform=T)` and the WORLD sensor against the oracle with the same transform: image and
    dL/d(data), dL/d(sdf.p); protocol methods (eval_all, ray_intersect) against the oracle's."""
    import configs
    import dsdf
    import shapes
    import integrators  # noqa: F401  (plugin registration)
    from integrators.reparam import Scene, create_integrator, traverse
    from constants import SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P
    dsdf.load()
    case = make_case('blob32')
    W, H, spp = case['W'], case['H'], case['spp']
    data = case['grid'].float()
    sdf = shapes.Grid3d(data.cuda(), transform=T)
    p0 = torch.tensor([0.01, -0.02, 0.015])
    sdf.p = p0.clone()
    osdf = O.Grid3d(data.double(), p0.double(), T)
    sensor = dsdf.Sensor(case['origin'], resx=W, resy=H)
    cam = O.Camera(case['origin'])
    seed = 9
    offs = torch.tensor(O.independent_sampler_2d(seed, (W + 4) * (H + 4) * spp)).double()
    for name, integ in (('sdf_silhouette_reparam', O.SILHOUETTE), ('sdf_simple_shading_reparam', O.SIMPLE_SHADING)):
        it = create_integrator(name, {'sdf': sdf})
        scene = Scene([sensor], it)
        it.warp_field = configs.get_config('warp').get_warpfield(it.sdf)
        img = it.render(scene, 0, seed=seed, spp=spp).cpu().double()
        ref = O.render(osdf, cam, W, H, spp, offs, integ, True)
        e = float(torch.linalg.norm(img - ref) / torch.linalg.norm(ref))
        # (a general rotation: the reference also marches through the corners of the world AABB outside the cube, where it
        # reads the clamped texture -- empty space, no hits, vanishing weights)
        assert e < (FWD_TOL if exact else 5e-4), (name, e)
        # the module-level render() (mi.render) maps the WORLD sensor exactly once on both of its branches: primal-only
        # (params without grad / no_grad) and attached (the primal-only branch used to transform it twice)
        from integrators.reparam import render as mi_render
        with torch.no_grad():
            img_plain = mi_render(scene, None, sensor=0, seed=seed, spp=spp)
        assert float(torch.linalg.norm(img_plain.cpu().double() - img) / torch.linalg.norm(img)) < 1e-6, name
        att = traverse(scene)
        att[SDF_DEFAULT_KEY] = data.cuda().clone().requires_grad_(True)
        att.update()
        img_att = mi_render(scene, att, sensor=[sensor], seed=seed, spp=spp, seed_grad=seed, spp_grad=spp)[0]
        assert float(torch.linalg.norm(img_att.detach().cpu().double() - img) / torch.linalg.norm(img)) < 1e-6, name
        params = traverse(scene)
        leaf = data.cuda().clone().requires_grad_(True)
        pl = p0.clone().requires_grad_(True)
        params[SDF_DEFAULT_KEY], params[SDF_DEFAULT_KEY_P] = leaf, pl
        params.update()
        gi = case['grad_image']
        it.render_backward(scene, params, gi.cuda(), 0, seed=seed, spp=spp)
        d2 = data.double().clone().requires_grad_(True)
        p2 = p0.double().clone().requires_grad_(True)
        (O.render(O.Grid3d(d2, p2, T), cam, W, H, spp, offs, integ, True) * gi.double()).sum().backward()
        eg = float(torch.linalg.norm(leaf.grad.cpu().double().reshape(d2.shape) - d2.grad) / torch.linalg.norm(d2.grad))
        ep = float(torch.linalg.norm(pl.grad.double() - p2.grad) / torch.linalg.norm(p2.grad))
        assert eg < (5e-3 if exact else 2e-2), (name, eg)
        assert ep < (5e-3 if exact else 2e-2), (name, ep)
    # protocol methods, world-space arguments
    sdf.p = p0.clone()
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(200, 3, generator=gen) * 0.5 + 0.25
    v, _, g, _, Hm = sdf.eval_all(x.cuda())
    vo, _, go, _, Ho = osdf.eval_all(x.double())
    assert torch.allclose(v.cpu().double(), vo, atol=2e-6) and torch.allclose(g.cpu().double(), go, atol=2e-4)
    assert torch.allclose(Hm.cpu().double(), Ho, atol=3e-2, rtol=1e-3)
    o, d, maxt = cam.sample_ray(torch.rand(300, 2, generator=gen).double() * torch.tensor([W, H]), W, H)
    out = sdf.ray_intersect(o.float().cuda(), d.float().cuda(), maxt.float().cuda(), warp=True)
    tr = O.ray_intersect(osdf, o, d, maxt)
    hit = torch.isfinite(tr['its_t'])
    assert bool((torch.isfinite(out[0].cpu()) == hit).all()) or not exact
    both = hit & torch.isfinite(out[0].cpu())
    assert torch.allclose(out[0].cpu().double()[both], tr['its_t'][both], atol=1e-4)
    if exact:
        m = tr['warp_weight'] > 1e-3
        assert torch.allclose(out[1].cpu().double()[m], tr['warp_t'][m], rtol=2e-3, atol=1e-4)


def _fp32_floor_tf(x, ref, tag, integ):
    """The oracle run in fp32 with the transform against the fp64 fixture: (dL/d data, dL/d p) rel-L2."""
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    cam32 = O.Camera.from_params(x['cam'].params(), dtype=torch.float32)
    d32, p32 = x['grid'].float().clone().requires_grad_(True), torch.from_numpy(ref['tf_p']).float().requires_grad_(True)
    i32 = O.render(O.Grid3d(d32, p32, AXIS_ALIGNED), cam32, x['W'], x['H'], x['spp'], x['offs'].float(), integ, True)
    (i32 * x['gi'].float()).sum().backward()
    return rel(d32.grad.numpy(), ref[f'tf_axis_grad_{tag}']), rel(p32.grad.numpy(), ref[f'tf_axis_gradp_{tag}'])


@pytest.mark.parametrize('tag,integ', [('sil', O.SILHOUETTE), ('shade', O.SIMPLE_SHADING)])
def test_change_of_frame_kernel_math_matches_reference_code(harness, tag, integ):
    """The product's route for `Grid3d(data, transform)` on the host: the mirror's change of frame (shapes.Grid3d.local_sensor /
    local_translation / the light direction of simple shading taken to the cube's frame, dL/dp back through to_local^T) around
    the host build of the kernel arithmetic, against what the reference's own code produced in WORLD space with the axis-aligned
    transform (`tf_axis_*`).  The GPU test below runs the same route through the library."""
    import dsdf
    import shapes
    from test_refshim_fixture import inputs, check_fp32_gradient
    import precision as P
    ref = _ref16()
    x = inputs(ref)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    g = shapes.Grid3d.__new__(shapes.Grid3d)                              # (the constructor wants a device tensor: frame fields only)
    g.has_transform = True
    g.to_world, g._A, g._b = shapes._rigid_parts(AXIS_ALIGNED)
    ls = g.local_sensor(dsdf.Sensor(ref['origin'], resx=x['W'], resy=x['H']))
    s = ls.to_struct()
    cam = np.array(list(s.origin) + list(s.left) + list(s.up) + list(s.dir) + [s.tan_half_fov, 0, 0, 0], np.float32)
    pl = g.local_translation(ref['tf_p'])
    light = g._A @ (np.ones(3) / np.sqrt(3.0))
    old = [harness.params.sdf_p[k] for k in range(3)], [harness.params.light_dir[k] for k in range(3)]
    try:
        for k in range(3):
            harness.params.sdf_p[k] = float(pl[k]); harness.params.light_dir[k] = float(light[k])
        gg, img = harness.render_backward(ref['grid'], cam, x['W'], x['H'], x['spp'], ref['sampler_2d'], ref['grad_image'], integ)
        gp = g._A.T @ harness.last_grad_p.astype(np.float64)             # to_local3^T: back to the world frame
    finally:
        for k in range(3):
            harness.params.sdf_p[k] = old[0][k]; harness.params.light_dir[k] = old[1][k]
    assert rel(img, ref[f'tf_axis_img_{tag}']) < 1e-4
    fg, fp = _fp32_floor_tf(x, ref, tag, integ)
    e = check_fp32_gradient('refshim_host', 'sphere16', f'tf_axis_{tag}', gg, ref[f'tf_axis_grad_{tag}'], max(P.FLOOR_FACTOR * fg, P.NORTH_STAR))
    ep = rel(gp, ref[f'tf_axis_gradp_{tag}'])
    print(f"host tf_axis {tag}: dL/d data {e:.3e} (fp32 oracle {fg:.3e}), dL/d p {ep:.3e} (fp32 oracle {fp:.3e})")
    assert ep < max(P.FLOOR_FACTOR * fp, 2 * e, P.NORTH_STAR), (ep, fp, e)


def _fp32_floor(x, ref, key, tag, integ):
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    cam32 = O.Camera.from_params(x['cam'].params(), dtype=torch.float32)
    d32, p32 = x['grid'].float().clone().requires_grad_(True), torch.from_numpy(ref['tf_p']).float().requires_grad_(True)
    i32 = O.render(O.Grid3d(d32, p32, ref[f'{key}_matrix']), cam32, x['W'], x['H'], x['spp'], x['offs'].float(), integ, True)
    (i32 * x['gi'].float()).sum().backward()
    return rel(d32.grad.numpy(), ref[f'{key}_grad_{tag}']), rel(p32.grad.numpy(), ref[f'{key}_gradp_{tag}'])


@pytest.mark.parametrize('key', ['tf_general', 'tf_axis'])
def test_world_space_kernel_math_matches_reference_code(harness_xf, key):
    """The -DDSDF_XF=1 build of the kernel arithmetic (host): WORLD-space rays, sensor and sdf.p, every lookup through
    to_local @ (x - p), derivatives back through to_local3^T (csrc/dsdf_math.h: to_grid, xf_apply_t, box_lo) -- against the
    reference's own shapes.py / reparam.py run with the same `to_world`: a rotation that is not axis-aligned, and the axis-aligned
    one (which the default build serves by a change of frame: both routes must agree with the same fixture)."""
    from test_refshim_fixture import inputs, check_fp32_gradient
    import precision as P
    ref = _ref16()
    x = inputs(ref)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    h = harness_xf
    h.set_transform(ref[f'{key}_matrix'])
    old = [h.params.sdf_p[k] for k in range(3)]
    try:
        for k in range(3):
            h.params.sdf_p[k] = float(ref['tf_p'][k])
        v, g, H = h.eval_cubic(ref['grid'], ref['eval_pts'], 2)
        Hr = ref[f'{key}_eval_H']
        H6 = np.stack([Hr[:, 0, 0], Hr[:, 1, 1], Hr[:, 2, 2], Hr[:, 0, 1], Hr[:, 0, 2], Hr[:, 1, 2]], -1)
        assert rel(v, ref[f'{key}_eval_v']) < 1e-6 and rel(g, ref[f'{key}_eval_g']) < 1e-6 and rel(H, H6) < 1e-5
        out = h.trace(ref['grid'], ref['ray_o'], ref['ray_d'], ref['ray_maxt'])
        hit, fin = np.isfinite(ref[f'{key}_ri_its_t']), np.isfinite(ref[f'{key}_ri_warp_t'])
        assert (np.isfinite(out['its_t']) == hit).mean() > 0.995
        both = hit & np.isfinite(out['its_t'])
        assert rel(out['its_t'][both], ref[f'{key}_ri_its_t'][both]) < 1e-5
        m = fin & np.isfinite(out['warp_t']) & (ref[f'{key}_ri_warp_weight'] > 1e-3)
        assert m.sum() > 100
        for k, tol in (('warp_t', 1e-4), ('warp_weight', 1e-3), ('warp_t_d', 2e-2), ('warp_weight_d', 2e-2)):
            assert rel(out[k][m], ref[f'{key}_ri_{k}'][m]) < tol, k
        for tag, integ in (('sil', O.SILHOUETTE), ('shade', O.SIMPLE_SHADING)):
            gg, img = h.render_backward(ref['grid'], ref['cam16'], x['W'], x['H'], x['spp'], ref['sampler_2d'], ref['grad_image'], integ)
            assert rel(img, ref[f'{key}_img_{tag}']) < 1e-4
            fg, fp = _fp32_floor(x, ref, key, tag, integ)
            e = check_fp32_gradient('refshim_host_xf', 'sphere16', f'{key}_{tag}', gg, ref[f'{key}_grad_{tag}'], max(P.FLOOR_FACTOR * fg, P.NORTH_STAR))
            ep = rel(h.last_grad_p, ref[f'{key}_gradp_{tag}'])
            print(f"host xf {key} {tag}: dL/d data {e:.3e} (fp32 oracle {fg:.3e}), dL/d p {ep:.3e} (fp32 oracle {fp:.3e})")
            assert ep < max(P.FLOOR_FACTOR * fp, 2 * e, P.NORTH_STAR), (ep, fp, e)
    finally:
        for k in range(3):
            h.params.sdf_p[k] = old[k]
        h.set_transform(np.eye(4))


def _fp32_floor_direct(x, ref, key):
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    cam32 = O.Camera.from_params(x['cam'].params(), dtype=torch.float32)
    d32, p32 = x['grid'].float().clone().requires_grad_(True), torch.from_numpy(ref['tf_p']).float().requires_grad_(True)
    a32 = x['albedo'].float().clone().requires_grad_(True)
    i32 = O.render(O.Grid3d(d32, p32, ref[f'{key}_matrix']), cam32, x['W'], x['H'], x['spp'], x['offs'].float(), O.DIRECT, True,
                   albedo=a32, emitter_u=x['emitter_u'].float(), env=x['env'].float())
    (i32 * x['gi'].float()).sum().backward()
    return (rel(d32.grad.numpy(), ref[f'{key}_grad_direct']), rel(p32.grad.numpy(), ref[f'{key}_gradp_direct']),
            rel(a32.grad.numpy(), ref[f'{key}_galb_direct']))


def test_oracle_direct_with_transform_matches_reference_code():
    """sdf_direct_reparam (the default integrator of the method configs) with the general `to_world`: shadow rays, the attached
    shadow-ray warp and the WORLD-space reflectance volume through the transformed grid -- oracle against the reference's own code."""
    from test_refshim_fixture import inputs
    ref = _ref16()
    x = inputs(ref)
    key = 'tf_general'
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a) - b) / np.linalg.norm(b))
    data, p = x['grid'].clone().requires_grad_(True), torch.from_numpy(ref['tf_p']).clone().requires_grad_(True)
    alb = x['albedo'].clone().requires_grad_(True)
    img = O.render(O.Grid3d(data, p, ref[f'{key}_matrix']), x['cam'], x['W'], x['H'], x['spp'], x['offs'], O.DIRECT, True, albedo=alb,
                   emitter_u=x['emitter_u'], env=x['env'])
    (img * x['gi']).sum().backward()
    assert rel(img.detach().numpy(), ref[f'{key}_img_direct']) < 1e-12
    assert rel(data.grad.numpy(), ref[f'{key}_grad_direct']) < 1e-9 and rel(p.grad.numpy(), ref[f'{key}_gradp_direct']) < 1e-9
    assert rel(alb.grad.numpy(), ref[f'{key}_galb_direct']) < 1e-9


def test_world_space_kernel_math_direct_matches_reference_code(harness_xf):
    """The same run through the -DDSDF_XF=1 host build of the kernel arithmetic (lane_backward_direct and its shadow-ray trace)."""
    from test_refshim_fixture import inputs, check_fp32_gradient
    import precision as P
    ref = _ref16()
    x = inputs(ref)
    key = 'tf_general'
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    h = harness_xf
    h.set_transform(ref[f'{key}_matrix'])
    old = [h.params.sdf_p[k] for k in range(3)]
    try:
        for k in range(3):
            h.params.sdf_p[k] = float(ref['tf_p'][k])
        gg, galb, gp, img = h.render_direct_backward(ref['grid'], ref['cam16'], x['W'], x['H'], x['spp'], ref['sampler_2d'], x['emitter_u'].numpy(),
                                                     ref['albedo'], ref['grad_image'], tuple(ref['env']))
    finally:
        for k in range(3):
            h.params.sdf_p[k] = old[k]
        h.set_transform(np.eye(4))
    fg, fp, fa = _fp32_floor_direct(x, ref, key)
    assert rel(img, ref[f'{key}_img_direct']) < 1e-4
    e = check_fp32_gradient('refshim_host_xf', 'sphere16', f'{key}_direct', gg, ref[f'{key}_grad_direct'], max(P.FLOOR_FACTOR * fg, P.NORTH_STAR))
    ep, ea = rel(gp, ref[f'{key}_gradp_direct']), rel(galb, ref[f'{key}_galb_direct'])
    print(f"host xf {key} direct: dL/d data {e:.3e} (fp32 oracle {fg:.3e}), dL/d p {ep:.3e} ({fp:.3e}), dL/d albedo {ea:.3e} ({fa:.3e})")
    assert ep < max(P.FLOOR_FACTOR * fp, 2 * e, P.NORTH_STAR) and ea < max(P.FLOOR_FACTOR * fa, P.NORTH_STAR), (ep, fp, ea, fa)


def test_world_space_kernel_math_with_scale_and_shear(harness_xf):
    """An affine `to_world` with a non-uniform SCALE and a shear on top of the rotation (to_local3 is neither orthogonal nor
    symmetric: the transposes in g <- A^T g, H <- A^T H A and cg <- A cg matter): the world-space host build against the oracle's
    literal restatement of python/shapes.py:408-450 (pinned to the reference's code for rotations, matrix-generic)."""
    from test_refshim_fixture import inputs, check_fp32_gradient
    import precision as P
    ref = _ref16()
    x = inputs(ref)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    L = rot(1, 25) @ np.diag([0.9, 1.1, 0.8]) @ np.array([[1.0, 0.15, 0.0], [0.0, 1.0, 0.0], [0.05, 0.0, 1.0]])
    T = about_centre(L, (0.01, 0.0, -0.02))
    p0 = torch.from_numpy(ref['tf_p'])
    h = harness_xf
    h.set_transform(T)
    old = [h.params.sdf_p[k] for k in range(3)]
    try:
        for k in range(3):
            h.params.sdf_p[k] = float(p0[k])
        osdf = O.Grid3d(x['grid'], p0, T)
        pts = ref['eval_pts']
        v, g, H = h.eval_cubic(ref['grid'], pts, 2)
        ov, _, og, _, oH = osdf.eval_all(torch.from_numpy(pts).double())
        oH6 = torch.stack([oH[:, 0, 0], oH[:, 1, 1], oH[:, 2, 2], oH[:, 0, 1], oH[:, 0, 2], oH[:, 1, 2]], -1)
        assert rel(v, ov.numpy()) < 1e-6 and rel(g, og.numpy()) < 1e-6 and rel(H, oH6.numpy()) < 1e-5
        for tag, integ in (('sil', O.SILHOUETTE), ('shade', O.SIMPLE_SHADING)):
            def oracle(dt):
                d_, p_ = x['grid'].to(dt).clone().requires_grad_(True), p0.to(dt).clone().requires_grad_(True)
                cam = O.Camera.from_params(x['cam'].params(), dtype=dt)
                img = O.render(O.Grid3d(d_, p_, T), cam, x['W'], x['H'], x['spp'], x['offs'].to(dt), integ, True)
                (img * x['gi'].to(dt)).sum().backward()
                return img.detach(), d_.grad, p_.grad
            (img_ref, gd, gp), tols = P.torch_gate(oracle)
            gg, img = h.render_backward(ref['grid'], ref['cam16'], x['W'], x['H'], x['spp'], ref['sampler_2d'], ref['grad_image'], integ)
            assert rel(img, img_ref) < 1e-4
            e = check_fp32_gradient('host_xf_affine', 'sphere16', f'affine_{tag}', gg, gd, tols[1])
            assert rel(h.last_grad_p, gp) < max(tols[2], 2 * e), (rel(h.last_grad_p, gp), tols[2], e)
    finally:
        for k in range(3):
            h.params.sdf_p[k] = old[k]
        h.set_transform(np.eye(4))


@pytest.mark.gpu
def test_transformed_grid_matches_reference_code_gpu(built):
    """The HIP path with `Grid3d(data, transform=AXIS_ALIGNED)`, sdf.p and the WORLD sensor against what the reference's own
    shapes.py / reparam.py produced with that transform (`tf_axis_*` of the sphere16 fixture): image, dL/d(data), dL/d(sdf.p).
    Gates: max(2 x the oracle's own fp32-vs-fixture error, 1e-4), like every other fp32 comparison of the suite."""
    import configs
    import dsdf
    import shapes
    import integrators  # noqa: F401
    from integrators.reparam import Scene, create_integrator, traverse
    from constants import SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P
    from test_refshim_fixture import inputs
    dsdf.load()
    ref = _ref16()
    x = inputs(ref)
    W, H, spp, seed = x['W'], x['H'], x['spp'], x['seed']
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    sensor = dsdf.Sensor(ref['origin'], resx=W, resy=H)
    gi = torch.from_numpy(ref['grad_image']).cuda()
    p0 = torch.from_numpy(ref['tf_p']).float()
    import precision as P
    from test_refshim_fixture import check_fp32_gradient
    for name, tag, integ in (('sdf_silhouette_reparam', 'sil', O.SILHOUETTE), ('sdf_simple_shading_reparam', 'shade', O.SIMPLE_SHADING)):
        floor_g, floor_p = _fp32_floor_tf(x, ref, tag, integ)
        sdf = shapes.Grid3d(torch.from_numpy(ref['grid']).cuda(), transform=AXIS_ALIGNED)
        sdf.p = p0.clone()
        it = create_integrator(name, {'sdf': sdf})
        scene = Scene([sensor], it)
        it.warp_field = configs.get_config('warp').get_warpfield(it.sdf)
        img = it.render(scene, 0, seed=seed, spp=spp).cpu().numpy()
        assert rel(img, ref[f'tf_axis_img_{tag}']) < 1e-4, tag
        params = traverse(scene)
        leaf = torch.from_numpy(ref['grid']).cuda().clone().requires_grad_(True)
        pl = p0.clone().requires_grad_(True)
        params[SDF_DEFAULT_KEY], params[SDF_DEFAULT_KEY_P] = leaf, pl
        params.update()
        it.render_backward(scene, params, gi, 0, seed=seed, spp=spp)
        eg = check_fp32_gradient('refshim_gpu', 'sphere16', f'tf_axis_{tag}', leaf.grad.cpu().numpy().reshape(ref['grid'].shape),
                                 ref[f'tf_axis_grad_{tag}'], max(P.FLOOR_FACTOR * floor_g, P.NORTH_STAR))
        ep = rel(pl.grad.numpy(), ref[f'tf_axis_gradp_{tag}'])
        print(f"gpu tf_axis {tag}: dL/d data {eg:.3e} (fp32 oracle {floor_g:.3e}), dL/d p {ep:.3e} (fp32 oracle {floor_p:.3e})")
        assert ep < max(P.FLOOR_FACTOR * floor_p, 2 * eg, P.NORTH_STAR), (tag, ep, floor_p, eg)


@pytest.mark.gpu
def test_general_transform_matches_reference_code_gpu(built):
    """A `to_world` that is NOT a change of frame (25 / -10 degrees: the reference's traced box, the world AABB of the rotated cube,
    outgrows the cube) through `shapes.Grid3d(data, transform)` -> dsdf.SdfGrid(to_world) -> the world-space build of the library
    (lib/variants/libdsdf_xf.so, dsdf_set_grid_transform), against what the reference's own shapes.py / reparam.py produced
    (`tf_general_*`): eval_all, ray_intersect, image, dL/d(data), dL/d(sdf.p) of both scene-free integrators.
    The gradient statistic of this configuration is ONE sample's footprint (99.9998 % of the squared fp32 error of the oracle's own
    fp32 run sits in 64 voxels; the host build of the same arithmetic measures 1.5 x that run's error): gate 3 x the fp32 floor."""
    import configs
    import dsdf
    import shapes
    import integrators  # noqa: F401
    from integrators.reparam import Scene, create_integrator, traverse
    from constants import SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P
    from test_refshim_fixture import inputs
    dsdf.load()
    ref = _ref16()
    x = inputs(ref)
    key = 'tf_general'
    W, H, spp, seed = x['W'], x['H'], x['spp'], x['seed']
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    p0 = torch.from_numpy(ref['tf_p']).float()
    sdf = shapes.Grid3d(torch.from_numpy(ref['grid']).cuda(), transform=GENERAL)
    assert sdf._world and sdf.grid.transform is not None
    sdf.p = p0.clone()
    # protocol methods, world-space arguments and results
    v, _, g, _, Hm = sdf.eval_all(torch.from_numpy(ref['eval_pts']).float().cuda())
    assert rel(v.cpu().numpy(), ref[f'{key}_eval_v']) < 1e-6 and rel(g.cpu().numpy(), ref[f'{key}_eval_g']) < 1e-6
    assert rel(Hm.cpu().numpy(), ref[f'{key}_eval_H']) < 1e-5
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().cuda()
    out = sdf.ray_intersect(dev(ref['ray_o']), dev(ref['ray_d']), dev(ref['ray_maxt']), warp=True)
    its = out[0].cpu().numpy()
    hit = np.isfinite(ref[f'{key}_ri_its_t'])
    assert (np.isfinite(its) == hit).mean() > 0.995
    both = hit & np.isfinite(its)
    assert rel(its[both], ref[f'{key}_ri_its_t'][both]) < 1e-5
    m = np.isfinite(ref[f'{key}_ri_warp_t']) & np.isfinite(out[1].cpu().numpy()) & (ref[f'{key}_ri_warp_weight'] > 1e-3)
    assert m.sum() > 100 and rel(out[1].cpu().numpy()[m], ref[f'{key}_ri_warp_t'][m]) < 1e-4
    assert rel(out[2].cpu().numpy()[m], ref[f'{key}_ri_warp_t_d'][m]) < 2e-2
    # integrators: WORLD sensor, nothing mapped
    sensor = dsdf.Sensor(ref['origin'], resx=W, resy=H)
    gi = torch.from_numpy(ref['grad_image']).cuda()
    for name, tag, integ in (('sdf_silhouette_reparam', 'sil', O.SILHOUETTE), ('sdf_simple_shading_reparam', 'shade', O.SIMPLE_SHADING)):
        floor_g, floor_p = _fp32_floor(x, ref, key, tag, integ)
        it = create_integrator(name, {'sdf': sdf})
        scene = Scene([sensor], it)
        it.warp_field = configs.get_config('warp').get_warpfield(it.sdf)
        sdf.p = p0.clone()
        img = it.render(scene, 0, seed=seed, spp=spp).cpu().numpy()
        assert rel(img, ref[f'{key}_img_{tag}']) < 1e-4, tag
        params = traverse(scene)
        leaf = torch.from_numpy(ref['grid']).cuda().clone().requires_grad_(True)
        pl = p0.clone().requires_grad_(True)
        params[SDF_DEFAULT_KEY], params[SDF_DEFAULT_KEY_P] = leaf, pl
        params.update()
        it.render_backward(scene, params, gi, 0, seed=seed, spp=spp)
        eg = rel(leaf.grad.cpu().numpy().reshape(ref['grid'].shape), ref[f'{key}_grad_{tag}'])
        ep = rel(pl.grad.numpy(), ref[f'{key}_gradp_{tag}'])
        print(f"gpu xf {key} {tag}: dL/d data {eg:.3e} (fp32 oracle {floor_g:.3e}), dL/d p {ep:.3e} (fp32 oracle {floor_p:.3e})")
        assert eg < max(3 * floor_g, 1e-4) and ep < max(3 * floor_p, 2 * eg, 1e-4), (tag, eg, floor_g, ep, floor_p)
    # sdf_direct_reparam with a WORLD-space reflectance volume (the change-of-frame route cannot take one, this build can)
    if f'{key}_img_direct' in ref.files:
        fg, fp, fa = _fp32_floor_direct(x, ref, key)
        it = create_integrator('sdf_direct_reparam', {'sdf': sdf, 'reflectance': torch.from_numpy(ref['albedo']).cuda(),
                                                      'env_radiance': tuple(float(e) for e in ref['env'])})
        scene = Scene([sensor], it)
        it.warp_field = configs.get_config('warp').get_warpfield(it.sdf)
        sdf.p = p0.clone()
        img = it.render(scene, 0, seed=seed, spp=spp).cpu().numpy()
        assert rel(img, ref[f'{key}_img_direct']) < 1e-4
        params = traverse(scene)
        leaf = torch.from_numpy(ref['grid']).cuda().clone().requires_grad_(True)
        pl = p0.clone().requires_grad_(True)
        params[SDF_DEFAULT_KEY], params[SDF_DEFAULT_KEY_P] = leaf, pl
        akey = [k for k in params if k.endswith('reflectance.volume.data')][0]
        params[akey] = torch.from_numpy(ref['albedo']).cuda().clone().requires_grad_(True)
        params.update()
        it.render_backward(scene, params, gi, 0, seed=seed, spp=spp)
        eg = rel(leaf.grad.cpu().numpy().reshape(ref['grid'].shape), ref[f'{key}_grad_direct'])
        ep, ea = rel(pl.grad.numpy(), ref[f'{key}_gradp_direct']), rel(params[akey].grad.cpu().numpy(), ref[f'{key}_galb_direct'])
        print(f"gpu xf {key} direct: dL/d data {eg:.3e} (fp32 oracle {fg:.3e}), dL/d p {ep:.3e} ({fp:.3e}), dL/d albedo {ea:.3e} ({fa:.3e})")
        assert eg < max(3 * fg, 1e-4) and ep < max(3 * fp, 2 * eg, 1e-4) and ea < max(3 * fa, 1e-4), (eg, fg, ep, fp, ea, fa)
    # the default library is untouched by all of this: a plain grid still renders through it
    plain = shapes.Grid3d(torch.from_numpy(ref['grid']).cuda())
    it = create_integrator('sdf_silhouette_reparam', {'sdf': plain})
    it.warp_field = configs.get_config('warp').get_warpfield(plain)
    img = it.render(Scene([sensor], it), 0, seed=seed, spp=spp).cpu().numpy()
    assert rel(img, ref['img_sil']) < 1e-4
