"""GPU parity of the two WarpField2D settings the reference's method configs change -- `warpprimary` (max_reparam_depth = 0,
python/configs.py:63-75, python/warp.py:103) and `warpnotnormalized` (normalize_warp_field = False, configs.py:96-109,
warp.py:59-62) -- through the C-ABI (dsdf_params.normalize_warp_field / max_reparam_depth) against the oracle's autograd on the
same samples.  CPU twin of every test: tests/test_warp_settings_host.py; gates: max(2 x fp32 floor, 1e-4)."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import direct_inputs, make_case
import precision as P
from conftest import rel_l2
from test_warp_settings_host import _oracle_grad

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4


@pytest.fixture(scope='module')
def dsdf(built):
    import dsdf as m
    m.load()
    assert torch.cuda.is_available()
    return m


def _grid(dsdf, case, **fields):
    grid = dsdf.SdfGrid(case['grid'].float().cuda())
    for k, v in fields.items():
        setattr(grid.params, k, v)
    sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
    return grid, sen


def test_warp_eval_not_normalized_gpu(dsdf):
    case = make_case('blob32')
    o32, d32, m32, tr = P.silhouette_rays(case)
    tr32 = {k: v.float() for k, v in tr.items() if k != 'steps'}
    grid, _ = _grid(dsdf, case, normalize_warp_field=0)
    out = dsdf.warp_eval(grid, o32.cuda(), d32.cuda(), {k: v.cuda() for k, v in tr32.items()})
    P.check_warp_coefficients('gpu', case, o32, d32, tr32, {k: v.cpu().numpy() for k, v in out.items()}, normalize=False)


@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_render_backward_not_normalized_gpu(dsdf, name, integ):
    case = make_case(name)
    (g64, img64), (tol, _) = P.torch_gate(lambda dt: _oracle_grad(case, integ, dt, normalize_warp_field=False))
    tol = max(tol, P.grad_tol(case, integ, True))                  # (see the CPU twin for why)
    grid, sen = _grid(dsdf, case, normalize_warp_field=0)
    name_of = {O.SILHOUETTE: 'sdf_silhouette_reparam', O.SIMPLE_SHADING: 'sdf_simple_shading_reparam'}[integ]
    gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                   integrator=name_of, return_image=True)
    assert rel_l2(img[0].cpu(), img64) < FWD_TOL
    e = rel_l2(gg.cpu(), g64)
    P.record('grad_not_normalized', case=name, integ=int(integ), err=e, tol=tol)
    assert e < tol, (e, tol)
    base, _ = _grid(dsdf, case)
    g0 = dsdf.render_backward(base, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(), integrator=name_of)
    if name != 'sphere16':
        assert rel_l2(gg.cpu(), g0.cpu()) > 1e-3


@pytest.mark.parametrize('mode', ['emitter', 'mis'])
@pytest.mark.parametrize('setting', ['primary_only', 'not_normalized', 'both'])
def test_direct_backward_settings_gpu(dsdf, mode, setting):
    case = make_case('blob32')
    ex = direct_inputs(case)
    bu = None
    if mode == 'mis':
        gen = torch.Generator().manual_seed(3)
        bu = torch.rand(case['offsets'].shape[0], 2, generator=gen, dtype=torch.float32)
    okw, hkw = {}, {}
    if setting in ('primary_only', 'both'):
        okw['max_reparam_depth'] = 0; hkw['max_reparam_depth'] = 0
    if setting in ('not_normalized', 'both'):
        okw['normalize_warp_field'] = False; hkw['normalize_warp_field'] = 0
    (gd64, ga64, img64), (tol_d, tol_a, _) = P.torch_gate(lambda dt: _oracle_grad(case, O.DIRECT, dt, ex, bu, **okw))

    def run(**fields):
        grid, sen = _grid(dsdf, case, **fields)
        sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], use_mis=bu is not None)
        galb = torch.zeros_like(sh.albedo)
        gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                       integrator='sdf_direct_reparam', return_image=True, shading=sh,
                                       emitter_samples=ex['emitter_u'].cuda(), bsdf_samples=None if bu is None else bu.cuda(),
                                       grad_albedo=galb)
        return gg.cpu(), galb.cpu(), img[0].cpu()
    gg, galb, img = run(**hkw)
    base, _, _ = run()
    assert rel_l2(img, img64) < FWD_TOL
    ea, ed = rel_l2(galb, ga64), rel_l2(gg, gd64)
    P.record('grad_direct_settings', mode=mode, setting=setting, err_data=ed, err_albedo=ea, tol_data=tol_d, tol_albedo=tol_a)
    assert ea < tol_a, (ea, tol_a)
    assert ed < tol_d, (ed, tol_d)
    assert rel_l2(gg, base) > 1e-4


def test_depth_rule_leaves_primary_integrators_alone_gpu(dsdf):
    """max_reparam_depth = 0 keeps the depth-0 warp: the silhouette gradient agrees with the default's to float-atomic order."""
    case = make_case('blob32')
    a, sen = _grid(dsdf, case)
    b, _ = _grid(dsdf, case, max_reparam_depth=0)
    kw = dict(offsets=case['offsets'].cuda())
    ga = dsdf.render_backward(a, sen, case['spp'], case['grad_image'].cuda()[None], **kw)
    gb = dsdf.render_backward(b, sen, case['spp'], case['grad_image'].cuda()[None], **kw)
    assert rel_l2(gb.cpu(), ga.cpu()) < 1e-5
