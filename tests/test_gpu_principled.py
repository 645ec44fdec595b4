"""GPU parity of the `principled` BSDF path of sdf_direct_reparam (principled-* configs,
/root/reference/python/opt_configs.py:288-299; base_color + roughness volumes) through the C-ABI against the oracle's
restatement of Mitsuba's plugin (oracle/sdf_oracle.py principled_eval; third-party, PARITY UNPINNED) and its autograd.
"""
import json

import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import make_case
import precision as P
from conftest import rel_l2
from test_principled_host import _oracle, _principled_inputs

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4


@pytest.fixture(scope='module')
def dsdf(built):
    import dsdf as m
    m.load()
    assert torch.cuda.is_available()
    return m


def setup(dsdf, case, ex):
    grid = dsdf.SdfGrid(case['grid'].float().cuda())
    sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
    sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], roughness=ex['roughness'].cuda())
    return grid, sen, sh


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
def test_principled_forward_gpu(dsdf, name):
    case = make_case(name)
    ex = _principled_inputs(case)
    ref = _oracle(case, ex, torch.float64, reparam=False, grads=False)
    grid, sen, sh = setup(dsdf, case, ex)
    for skip in (True, False):
        img = dsdf.render_forward(grid, sen, case['spp'], offsets=case['offsets'].cuda(), integrator='sdf_direct_reparam',
                                  shading=sh, emitter_samples=ex['emitter_u'].cuda(), empty_space_skip=skip)[0]
        assert rel_l2(img.cpu(), ref) < FWD_TOL


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
@pytest.mark.parametrize('reparam', [True, False])
def test_principled_backward_gpu(dsdf, name, reparam):
    case = make_case(name)
    ex = _principled_inputs(case)
    (img_ref, gd, ga, gr), tols = P.torch_gate(lambda dt: _oracle(case, ex, dt, reparam=reparam))
    grid, sen, sh = setup(dsdf, case, ex)
    galb = torch.zeros_like(sh.albedo)
    sh.grad_roughness = torch.zeros_like(sh.roughness)
    gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                   integrator='sdf_direct_reparam', reparam=reparam, return_image=True, shading=sh,
                                   emitter_samples=ex['emitter_u'].cuda(), grad_albedo=galb)
    assert rel_l2(img[0].cpu(), img_ref) < FWD_TOL
    assert torch.isfinite(gg).all() and torch.isfinite(galb).all() and torch.isfinite(sh.grad_roughness).all()
    ea, er = rel_l2(galb.cpu(), ga), rel_l2(sh.grad_roughness.cpu(), gr)
    P.record('grad_principled', case=name, reparam=reparam, err_base_color=ea, err_roughness=er, tol_base_color=tols[2], tol_roughness=tols[3])
    assert ea < tols[2], (ea, tols[2])
    assert er < tols[3], (er, tols[3])
    if reparam or float(np.abs(gd).max()) > 0:
        ed = rel_l2(gg.cpu(), gd)
        assert ed < tols[1], (ed, tols[1])


def test_principled_mis_is_routed_per_call(dsdf, monkeypatch):
    """`use_mis` with the principled BSDF lives in the extended build only: the DEFAULT library refuses it at the C-ABI, the Python
    layer routes exactly the calls that carry such a Shading to lib/variants/libdsdf_xf.so (identity transform) and leaves the grid
    what it was -- its other renders keep the default library (no permanent set_to_world(eye))."""
    import dsdf.renderer as R
    case = make_case('sphere16')
    ex = _principled_inputs(case)
    grid, sen, _ = setup(dsdf, case, ex)
    sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], use_mis=True, roughness=ex['roughness'].cuda())
    img = dsdf.render_forward(grid, sen, 4, seeds=[1], integrator='sdf_direct_reparam', shading=sh)
    assert torch.isfinite(img).all() and float(img.abs().sum()) > 0
    assert grid.transform is None and grid.lib().dsdf_has_grid_transform() == 0 and grid.lib(True).dsdf_has_grid_transform() == 1
    monkeypatch.setattr(R, '_needs_extended', lambda shading: False)          # the default library's own answer
    with pytest.raises(dsdf.DsdfError, match='extended build'):
        dsdf.render_forward(grid, sen, 4, seeds=[1], integrator='sdf_direct_reparam', shading=sh)


def test_principled_forward_mode_gpu(dsdf):
    """`render_forward` (integrators/reparam.py:192-196) with the principled BSDF through the C-ABI: the transpose identity against
    dsdf_render_backward on the same samples, tangents on sdf.data and on sdf.p."""
    R, W, H = 48, 32, 32
    data = O.blob_grid(R, n=10, seed=3).float().cuda()
    grid = dsdf.SdfGrid(data)
    sens = dsdf.get_regular_cameras(6, resx=W, resy=H)[:2]
    torch.manual_seed(2)
    sh = dsdf.Shading(torch.rand(6, 5, 4, 3, device='cuda') * 0.6 + 0.2, (1.0, 0.9, 0.8),
                      roughness=torch.rand(3, 4, 5, 1, device='cuda') * 0.7 + 0.1)
    gi = torch.randn(2, H, W, 3, device='cuda')
    gp = torch.zeros(3, device='cuda')
    gg = dsdf.render_backward(grid, sens, 64, gi, seeds=[4, 5], integrator='sdf_direct_reparam', shading=sh, grad_p=gp)
    tdata = torch.randn_like(data)
    tpv = [0.3, -0.2, 0.5]
    jd = dsdf.render_forward_grad(grid, sens, 64, tangent_data=tdata, seeds=[4, 5], integrator='sdf_direct_reparam', shading=sh)
    jp = dsdf.render_forward_grad(grid, sens, 64, tangent_p=tpv, seeds=[4, 5], integrator='sdf_direct_reparam', shading=sh)
    lhs_d, rhs_d = float((jd.double() * gi).sum()), float((tdata.double() * gg).sum())
    lhs_p, rhs_p = float((jp.double() * gi).sum()), float(sum(a * float(b) for a, b in zip(tpv, gp)))
    assert abs(lhs_d - rhs_d) <= 5e-3 * max(abs(rhs_d), 1e-6), (lhs_d, rhs_d)
    assert abs(lhs_p - rhs_p) <= 5e-3 * max(abs(rhs_p), 1e-6), (lhs_p, rhs_p)


def test_principled_plugin_and_render_op(dsdf):
    """The plugin with a `roughness` property publishes base_color / roughness (the keys python/opt_configs.py:291 optimises) and
    its render op attaches sdf.data and both volumes."""
    import configs
    import shapes
    from constants import SDF_DEFAULT_KEY
    from integrators.reparam import Scene, create_integrator, render, traverse
    from integrators.sdf_direct_reparam import BASE_COLOR_KEY, ROUGHNESS_KEY
    data = O.blob_grid(32, n=6, seed=1).float().cuda()
    base = torch.rand(8, 8, 8, 3, device='cuda') * 0.6 + 0.2
    rough = torch.rand(4, 4, 4, 1, device='cuda') * 0.7 + 0.1
    sens = dsdf.get_regular_cameras(3, resx=24, resy=24)
    integ = create_integrator('sdf_direct_reparam', {'sdf': shapes.Grid3d(data.clone()), 'base_color': base, 'roughness': rough,
                                                     'hide_emitters': True})
    scene = Scene(sens, integ)
    integ.warp_field = configs.get_config('warp').get_warpfield(integ.sdf)
    params = traverse(scene)
    assert set(params) == {SDF_DEFAULT_KEY, 'SamplingIntegrator.sdf.p', BASE_COLOR_KEY, ROUGHNESS_KEY}
    sh = dsdf.Shading(base, 1.0, hide_emitters=True, roughness=rough)
    img = integ.render(scene, sensor=1, seed=5, spp=64)
    ref = dsdf.render_forward(dsdf.SdfGrid(data), sens[1], 64, seeds=[5], integrator='sdf_direct_reparam', shading=sh)[0]
    assert rel_l2(img.cpu(), ref.cpu()) < 1e-6
    p = params[SDF_DEFAULT_KEY].clone().requires_grad_(True)
    a, r = base.clone().requires_grad_(True), rough.clone().requires_grad_(True)
    params[SDF_DEFAULT_KEY], params[BASE_COLOR_KEY], params[ROUGHNESS_KEY] = p, a, r
    params.update()
    out = render(scene, params, sensor=[sens[0], sens[2]], seed=3, spp=64, seed_grad=9, spp_grad=64)
    out.sum().backward()
    ga = torch.zeros_like(base)
    sh.grad_roughness = torch.zeros_like(rough)
    gref = dsdf.render_backward(dsdf.SdfGrid(data), [sens[0], sens[2]], 64, torch.ones(2, 24, 24, 3, device='cuda'),
                                seeds=[9, 10], integrator='sdf_direct_reparam', shading=sh, grad_albedo=ga)
    assert rel_l2(p.grad[..., 0].cpu(), gref.cpu()) < 1e-5 and rel_l2(a.grad.cpu(), ga.cpu()) < 1e-5
    assert rel_l2(r.grad.cpu(), sh.grad_roughness.cpu()) < 1e-5 and float(r.grad.abs().max()) > 0
    # use_mis with the principled BSDF is the extended build's business: the mirror routes such a grid there (identity transform)
    it_mis = create_integrator('sdf_direct_reparam', {'sdf': shapes.Grid3d(data.clone()), 'roughness': rough, 'base_color': base, 'use_mis': True})
    assert it_mis.sdf.grid.transform is None
    it_mis._configured()
    assert it_mis.sdf.grid.transform is None                           # (routed per call, the grid is not touched: dsdf.SdfGrid.lib)
    it_mis.warp_field = configs.get_config('warp').get_warpfield(it_mis.sdf)
    img_mis = it_mis.render(Scene(sens, it_mis), sensor=1, seed=5, spp=64)
    assert torch.isfinite(img_mis).all() and it_mis.sdf.grid.transform is None


@pytest.mark.parametrize('name', ['blob32', 'blob48_rect'])
def test_principled_mis_backward_gpu(dsdf, name):
    """`use_mis` with the principled BSDF (sdf_direct_reparam.py:77-105; Principled::sample / ::pdf at the plugin defaults, restated:
    PARITY UNPINNED) through the extended build of the library (lib/variants/libdsdf_xf.so; a grid with the identity `to_world`):
    image, dL/d(sdf.data), dL/d(base_color), dL/d(roughness) against the oracle's autograd on the same explicit samples."""
    from test_principled_mis_host import _oracle as oracle_mis, _samples
    case = make_case(name)
    ex = _principled_inputs(case)
    lobe, bu = _samples(case)
    (img_ref, gd, ga, gr), tols = P.torch_gate(lambda dt: oracle_mis(case, ex, lobe, bu, dt))
    grid = dsdf.SdfGrid(case['grid'].float().cuda(), to_world=np.eye(4))
    assert grid.lib().dsdf_has_grid_transform() == 1
    sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
    sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], use_mis=True, roughness=ex['roughness'].cuda())
    sh.lobe_samples = lobe.cuda()[None].contiguous()
    galb = torch.zeros_like(sh.albedo)
    sh.grad_roughness = torch.zeros_like(sh.roughness)
    gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                   integrator='sdf_direct_reparam', return_image=True, shading=sh,
                                   emitter_samples=ex['emitter_u'].cuda(), bsdf_samples=bu.cuda(), grad_albedo=galb)
    assert rel_l2(img[0].cpu(), img_ref) < FWD_TOL
    prim = dsdf.render_forward(grid, sen, case['spp'], offsets=case['offsets'].cuda(), integrator='sdf_direct_reparam', shading=sh,
                               emitter_samples=ex['emitter_u'].cuda(), bsdf_samples=bu.cuda())[0]
    assert rel_l2(prim.cpu(), img_ref) < FWD_TOL                    # (the primal pass: plain traces, same samples)
    e = (rel_l2(gg.cpu(), gd), rel_l2(galb.cpu(), ga), rel_l2(sh.grad_roughness.cpu(), gr))
    print(f"gpu principled+mis {name}: dL/d data {e[0]:.3e} (gate {tols[1]:.3e}), base_color {e[1]:.3e} ({tols[2]:.3e}), roughness {e[2]:.3e} ({tols[3]:.3e})")
    assert e[1] < tols[2] and e[2] < tols[3], (e, tols)
    from test_refshim_fixture import check_fp32_gradient          # (plain gate; one heavy-tailed sample footprint may be set aside)
    check_fp32_gradient('principled_mis_gpu', name, 'direct_mis', gg.cpu().numpy(), gd, tols[1])
    # a transform-free grid gets the same result: the call is routed to the extended build with the identity (dsdf.SdfGrid.lib)
    plain = dsdf.render_forward(dsdf.SdfGrid(case['grid'].float().cuda()), sen, case['spp'], offsets=case['offsets'].cuda(),
                                integrator='sdf_direct_reparam', shading=sh, emitter_samples=ex['emitter_u'].cuda(), bsdf_samples=bu.cuda())[0]
    assert rel_l2(plain.cpu(), prim.cpu()) < 1e-6
