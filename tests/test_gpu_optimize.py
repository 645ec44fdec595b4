"""GPU tests of the widened rows: HIP redistancing against the C fast-sweeping oracle, the
reference-mirroring classes (Grid3d, integrators, render op), tricubic upsampling, and a short
end-to-end `optimize.py` run."""
import json
import os

import numpy as np
import pytest
import torch

import sdf_oracle as O
from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dsdf(built):
    import dsdf as m
    m.load()
    return m


def distorted_sphere(R):
    lin = np.linspace(0, 1, R)
    z, y, x = np.meshgrid(lin, lin, lin, indexing='ij')
    sd = np.sqrt((x - .5) ** 2 + (y - .45) ** 2 + (z - .55) ** 2) - 0.3
    return sd, (sd * (1.5 + 0.5 * np.sin(7 * x))).astype(np.float32)


@pytest.mark.parametrize('shape', [(40, 40, 40), (24, 33, 47)])
def test_redistance_matches_c_oracle(dsdf, shape):
    import c_oracle
    lib = c_oracle.load()
    rng = np.random.default_rng(0)
    lin = [np.linspace(0, 1, s) for s in shape]
    z, y, x = np.meshgrid(*lin, indexing='ij')
    phi = ((np.sqrt((x - .5) ** 2 + (y - .5) ** 2 + (z - .5) ** 2) - 0.3) * (1.4 + 0.5 * np.sin(9 * y))).astype(np.float32)
    ref = c_oracle.redistance(lib, phi)
    out = dsdf.redistance(torch.from_numpy(phi).cuda()).cpu().numpy()
    assert ((out < 0) == (phi < 0)).all()
    assert np.abs(out - ref).max() < 1e-5          # same discrete fixed point (Godunov upwind, frozen band)
    out4, status = dsdf.redistance(torch.from_numpy(phi).cuda()[..., None], return_status=True)
    assert out4.shape == (*shape, 1)
    assert int(status.item()) == 0                  # converged within the launch budget (device-side flag, no sync in the library)


def test_redistance_large_and_idempotent(dsdf):
    sd, phi = distorted_sphere(128)
    u = dsdf.redistance(torch.from_numpy(phi).cuda())
    assert float((u.cpu() - torch.from_numpy(sd).float()).abs().max()) < 2.5 / 128     # first-order scheme, h = 1/res on a linspace(0,1,res) sampling
    u2 = dsdf.redistance(u)
    assert float((u2 - u).abs().max()) < 0.3 / 128


def test_grid3d_protocol_and_upsample(dsdf):
    import shapes
    import variables
    data = O.blob_grid(32, n=6, seed=1).float().cuda()
    g = shapes.Grid3d(data)
    assert g.shape == (32, 32, 32, 1)
    pts = torch.rand(1000, 3, device='cuda') * 0.8 + 0.1
    v, vd, gr, grd, H = g.eval_all(pts)
    vo, go, Ho = O.eval_cubic(data.cpu().double(), pts.cpu().double(), 2)
    assert rel_l2(v.cpu(), vo) < 1e-6 and rel_l2(gr.cpu(), go) < 1e-5 and rel_l2(H.cpu(), Ho) < 1e-5
    assert torch.equal(g.eval(pts), v) and torch.equal(g.eval_and_grad(pts)[1], gr)
    up = variables.upsample_sdf(data)                                # python/variables.py:18-23
    assert up.shape == (64, 64, 64, 1)
    ax = (torch.arange(64, dtype=torch.float64) + 0.5) / 64
    z, y, x = torch.meshgrid(ax, ax, ax, indexing='ij')
    ref = O.eval_cubic(data.cpu().double(), torch.stack([x, y, z], -1).reshape(-1, 3), 0)[0].reshape(64, 64, 64)
    assert rel_l2(up[..., 0].cpu(), ref) < 1e-5
    cam = O.Camera(O.regular_camera_origins(3)[1]).rounded()
    o, d, maxt = cam.sample_ray(torch.rand(500, 2, dtype=torch.float64) * 24, 24, 24)
    its, wt, wtd, ww, wwd = g.ray_intersect(o.float().cuda(), d.float().cuda(), maxt.float().cuda(), warp=object())
    its_nd = g.ray_intersect_non_diff(o.float().cuda(), d.float().cuda(), maxt.float().cuda())[0]
    fin = torch.isfinite(its)
    assert torch.equal(fin, torch.isfinite(its_nd)) and rel_l2(its[fin].cpu(), its_nd[fin].cpu()) < 1e-6
    sph = shapes.create_sphere_sdf([32, 32, 32])
    assert sph.shape == (32, 32, 32) and abs(float(sph[16, 16, 16]) + 0.3) < 0.08        # first-order Eikonal solve at 32^3


def test_integrator_plugins_and_render_op(dsdf):
    import configs
    import shapes
    from integrators.reparam import Scene, create_integrator, render, traverse
    from constants import SDF_DEFAULT_KEY
    data = O.blob_grid(32, n=6, seed=1).float().cuda()
    sens = dsdf.get_regular_cameras(3, resx=24, resy=24)
    for name, integ_id in (('sdf_silhouette_reparam', 0), ('sdf_simple_shading_reparam', 1)):
        integ = create_integrator(name, {'sdf': shapes.Grid3d(data.clone())})
        scene = Scene(sens, integ)
        integ.warp_field = configs.get_config('warp').get_warpfield(integ.sdf)
        params = traverse(scene)
        assert set(params) == {SDF_DEFAULT_KEY, 'SamplingIntegrator.sdf.p'}
        img = integ.render(scene, sensor=1, seed=5, spp=64)
        ref = dsdf.render_forward(dsdf.SdfGrid(data), sens[1], 64, seeds=[5], integrator=integ_id)[0]
        assert img.shape == (24, 24, 3) and rel_l2(img.cpu(), ref.cpu()) < 1e-6
        params.keep([SDF_DEFAULT_KEY])
        p = params[SDF_DEFAULT_KEY].clone().requires_grad_(True)
        params[SDF_DEFAULT_KEY] = p
        out = render(scene, params, sensor=[sens[0], sens[2]], seed=3, spp=64, seed_grad=9, spp_grad=64)
        assert out.shape == (2, 24, 24, 3)
        out.sum().backward()
        gref = dsdf.render_backward(dsdf.SdfGrid(data), [sens[0], sens[2]], 64, torch.ones(2, 24, 24, 3, device='cuda'),
                                    seeds=[9, 10], integrator=integ_id)
        assert rel_l2(p.grad[..., 0].cpu(), gref.cpu()) < 1e-5
        gi = torch.ones(24, 24, 3, device='cuda')
        p.grad = None
        integ.render_backward(scene, {SDF_DEFAULT_KEY: p}, gi, sensor=1, seed=2, spp=64)   # accumulates like dr.grad
        integ.render_backward(scene, {SDF_DEFAULT_KEY: p}, gi, sensor=1, seed=2, spp=64)
        g1 = dsdf.render_backward(dsdf.SdfGrid(data), sens[1], 64, gi[None], seeds=[2], integrator=integ_id)
        assert rel_l2(p.grad[..., 0].cpu(), 2 * g1.cpu()) < 1e-5
    with pytest.raises(ValueError):
        create_integrator('sdf_prb_reparam')
    with pytest.raises(Exception, match='develop=True'):
        integ.render(scene, develop=False)



def test_direct_integrator_plugin(dsdf):
    """The `sdf_direct_reparam` plugin publishes the reflectance volume next to sdf.data and its render op
    attaches both (python/opt_configs.py:286 optimises exactly these two keys)."""
    import configs
    import shapes
    from constants import SDF_DEFAULT_KEY
    from integrators.reparam import Scene, create_integrator, render, traverse
    from integrators.sdf_direct_reparam import REFLECTANCE_KEY
    data = O.blob_grid(32, n=6, seed=1).float().cuda()
    alb = (torch.rand(8, 8, 8, 3, device='cuda') * 0.6 + 0.2)
    sens = dsdf.get_regular_cameras(3, resx=24, resy=24)
    integ = create_integrator('sdf_direct_reparam', {'sdf': shapes.Grid3d(data.clone()), 'reflectance': alb, 'hide_emitters': True})
    scene = Scene(sens, integ)
    integ.warp_field = configs.get_config('warp').get_warpfield(integ.sdf)
    params = traverse(scene)
    assert set(params) == {SDF_DEFAULT_KEY, 'SamplingIntegrator.sdf.p', REFLECTANCE_KEY}
    sh = dsdf.Shading(alb, 1.0, hide_emitters=True)
    img = integ.render(scene, sensor=1, seed=5, spp=64)
    ref = dsdf.render_forward(dsdf.SdfGrid(data), sens[1], 64, seeds=[5], integrator='sdf_direct_reparam', shading=sh)[0]
    assert rel_l2(img.cpu(), ref.cpu()) < 1e-6 and float((img[..., 0] - img[..., 2]).abs().max()) > 0
    params.keep([SDF_DEFAULT_KEY, REFLECTANCE_KEY])
    p = params[SDF_DEFAULT_KEY].clone().requires_grad_(True)
    a = alb.clone().requires_grad_(True)
    params[SDF_DEFAULT_KEY], params[REFLECTANCE_KEY] = p, a
    params.update()
    out = render(scene, params, sensor=[sens[0], sens[2]], seed=3, spp=64, seed_grad=9, spp_grad=64)
    out.sum().backward()
    ga = torch.zeros_like(alb)
    gref = dsdf.render_backward(dsdf.SdfGrid(data), [sens[0], sens[2]], 64, torch.ones(2, 24, 24, 3, device='cuda'),
                                seeds=[9, 10], integrator='sdf_direct_reparam', shading=sh, grad_albedo=ga)
    assert rel_l2(p.grad[..., 0].cpu(), gref.cpu()) < 1e-5 and rel_l2(a.grad.cpu(), ga.cpu()) < 1e-5
    # the integrator properties of the reference (reparam.py:17, sdf_direct_reparam.py:12-14) reach the library
    mis = create_integrator('sdf_direct_reparam', {'use_mis': True, 'decouple_reparam': True, 'sdf': shapes.Grid3d(data.clone()),
                                                   'reflectance': alb, 'hide_emitters': True})
    sm = mis.shading()
    assert sm.use_mis and sm.decouple_reparam and not sm.detach_indirect_si
    scene_m = Scene(sens, mis)
    img_m = mis.render(scene_m, sensor=1, seed=5, spp=256)
    img_e = integ.render(scene, sensor=1, seed=5, spp=256)
    assert abs(float(img_m.mean()) - float(img_e.mean())) < 0.02 * float(img_e.mean())      # MIS estimates the same integral



def test_integrator_render_forward(dsdf):
    """`integrator.render_forward` (python/integrators/reparam.py:192-196) with the tangent on one axis of sdf.p, as
    the reference's gradient-image validation uses it (figures/result_utils.py:126-161): matches central differences
    of the un-reparameterised render in the image-space mean over the silhouette band (2048 spp)."""
    import configs
    import shapes
    from constants import SDF_DEFAULT_KEY_P
    from integrators.reparam import Scene, create_integrator, traverse
    data = O.sphere_grid(64, radius=0.3).float().cuda()
    sens = dsdf.get_regular_cameras(3, resx=32, resy=32)
    integ = create_integrator('sdf_silhouette_reparam', {'sdf': shapes.Grid3d(data.clone())})
    scene = Scene(sens, integ)
    integ.warp_field = configs.get_config('warp').get_warpfield(integ.sdf)
    params = traverse(scene)
    p = params[SDF_DEFAULT_KEY_P]
    p.grad = torch.tensor([1.0, 0.0, 0.0])                                   # dr.forward(p.x)
    g = integ.render_forward(scene, params, sensor=1, seed=7, spp=2048)
    assert g.shape == (32, 32, 3) and torch.isfinite(g).all()
    eps = 2e-3
    grid = dsdf.SdfGrid(data)
    hi = dsdf.render_forward(grid.set_translation([eps, 0, 0]), sens[1], 2048, seeds=[7], reparam=False)[0]
    lo = dsdf.render_forward(grid.set_translation([-eps, 0, 0]), sens[1], 2048, seeds=[7], reparam=False)[0]
    fd = (hi - lo) / (2 * eps)
    # per-pixel FD of a discontinuous integrand is noisy; compare column sums (the derivative of the covered area per column)
    a, b = g[..., 0].sum(0).cpu(), fd[..., 0].sum(0).cpu()
    assert rel_l2(a, b) < 0.1, (a, b)
    p.grad = None
    with pytest.raises(ValueError):
        integ.render_forward(scene, params, sensor=1, seed=7, spp=64)
