"""The two WarpField2D settings that the reference's method configs change (python/configs.py:63-75 `warpprimary`:
max_reparam_depth = 0; :96-109 `warpnotnormalized`: normalize_warp_field = False), as fields of dsdf_params: the kernel
arithmetic compiled for the host (tests/harness, TEST-ONLY) against the oracle's autograd on the same samples.
Gates as everywhere: max(2 x the oracle's own fp32-vs-fp64 difference, 1e-4) (tests/precision.py)."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import direct_inputs, make_case
import precision as P
from conftest import rel_l2

FWD_TOL = 1e-4


def cam_params(case):
    return O.Camera(case['origin']).params()


def _oracle_grad(case, integ, dt, ex=None, bsdf_u=None, **kw):
    """(dL/d data[, dL/d albedo], image) of the torch oracle in precision `dt`."""
    data = case['grid'].float().to(dt).clone().requires_grad_(True)
    extra = {}
    alb = None
    if ex is not None:
        alb = ex['albedo'].to(dt).clone().requires_grad_(True)
        extra = dict(albedo=alb, emitter_u=ex['emitter_u'].to(dt), env=torch.tensor(ex['env'], dtype=dt))
        if bsdf_u is not None:
            extra.update(use_mis=True, bsdf_u=bsdf_u.to(dt))
    img = O.render(O.Grid3d(data), O.Camera.from_params(cam_params(case), dtype=dt), case['W'], case['H'], case['spp'],
                   case['offsets'].to(dt), integ, True, **extra, **kw)
    (img * case['grad_image'].to(dt)).sum().backward()
    out = (data.grad,) + ((alb.grad,) if alb is not None else ()) + (img.detach(),)
    return out


def test_defaults_are_the_reference_defaults(harness):
    assert harness.params.normalize_warp_field == 1 and harness.params.max_reparam_depth == -1     # warp.py:11, 20


def test_warp_eval_not_normalized_host(harness):
    """A9 per ray with V = -g v (warp.py:60-62): cdir, a, b, div against the autograd linearisation of the oracle's warp_eval."""
    case = make_case('blob32')
    o32, d32, m32, tr = P.silhouette_rays(case)
    tr32 = {k: v.float() for k, v in tr.items() if k != 'steps'}
    args = (case['grid'].float().numpy(), o32.numpy(), d32.numpy(), {k: v.numpy() for k, v in tr32.items()})
    base = harness.warp_eval(*args)
    with harness.settings(normalize_warp_field=0):
        out = harness.warp_eval(*args)
    P.check_warp_coefficients('host-harness', case, o32, d32, tr32, out, normalize=False)
    act = np.asarray(out['active']) != 0
    assert np.array_equal(act, np.asarray(base['active']) != 0)                  # the weight does not depend on the normalisation
    assert rel_l2(np.asarray(out['a'])[act], np.asarray(base['a'])[act]) > 1e-2     # ... the field does (|g| != 1 on this grid)
    again = harness.warp_eval(*args)                                             # the setting is per call, nothing sticks
    assert all(np.array_equal(np.asarray(again[k]), np.asarray(base[k])) for k in ('cdir', 'a', 'b', 'div'))


@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_render_backward_not_normalized_host(harness, name, integ):
    case = make_case(name)
    (g64, img64), (tol, _) = P.torch_gate(lambda dt: _oracle_grad(case, integ, dt, normalize_warp_field=False))
    a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(),
         case['grad_image'].numpy(), integ)
    base, _ = harness.render_backward(*a)
    with harness.settings(normalize_warp_field=0):
        gg, img = harness.render_backward(*a)
    assert rel_l2(img, img64) < FWD_TOL                                          # the primal image never sees the warp
    # the C restatement (normalised field only) measures a larger fp32 floor than torch on these samples -- same samples, same
    # arithmetic but for the scale of the field: its gate is the one the normalised configuration is held to
    tol = max(tol, P.grad_tol(case, integ, True))
    assert rel_l2(gg, g64) < tol, (rel_l2(gg, g64), tol)
    if name == 'blob32':
        assert rel_l2(gg, base) > 1e-3                                           # it IS another estimator on a non-distance field


@pytest.mark.parametrize('mode', ['emitter', 'mis'])
@pytest.mark.parametrize('setting', ['primary_only', 'not_normalized', 'both'])
def test_direct_backward_settings_host(harness, mode, setting):
    """sdf_direct_reparam: `warpprimary` drops the warp of the shadow ray (sdf_direct_reparam.py:52) and of the BSDF-sampled
    ray (:95) -- det_e = det_bsdf = 1 -- while the primary ray keeps its own; `warpnotnormalized` changes the field of all three."""
    case = make_case('blob32')                                                   # (the shadow-ray warp is active on this case)
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
    a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(),
         ex['emitter_u'].numpy(), ex['albedo'].numpy(), case['grad_image'].numpy(), ex['env'])
    kw = dict(bsdf_u=None if bu is None else bu.numpy())
    base, base_alb, _, _ = harness.render_direct_backward(*a, **kw)
    with harness.settings(**hkw):
        gg, galb, _, img = harness.render_direct_backward(*a, **kw)
    assert rel_l2(img, img64) < FWD_TOL
    assert rel_l2(galb, ga64) < tol_a, (rel_l2(galb, ga64), tol_a)
    assert rel_l2(gg, gd64) < tol_d, (rel_l2(gg, gd64), tol_d)
    assert rel_l2(gg, base) > 1e-4                                               # the setting changes dL/d(sdf) ...
    if setting == 'primary_only':
        # ... and, for the depth rule alone, it is neither the full method nor "no reparameterisation at all"
        none, _, _, _ = harness.render_direct_backward(*a, reparam=False, **kw)
        assert rel_l2(gg, none) > 1e-4


def test_depth_rule_leaves_primary_integrators_alone(harness):
    """max_reparam_depth = 0 still reparameterises depth 0: silhouette and simple shading are unchanged bit for bit."""
    case = make_case('blob32')
    for integ in (O.SILHOUETTE, O.SIMPLE_SHADING):
        a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(),
             case['grad_image'].numpy(), integ)
        base, _ = harness.render_backward(*a)
        with harness.settings(max_reparam_depth=0):
            gg, _ = harness.render_backward(*a)
        assert np.array_equal(gg, base)


def test_forward_mode_is_transpose_of_backward_under_settings_host(harness):
    """`render_forward` (integrators/reparam.py:192-196) honours the same two settings: <J dtheta, G> = <dtheta, J^T G>."""
    case = make_case('blob32')
    ex = direct_inputs(case)
    gen = torch.Generator().manual_seed(3)
    bu = torch.rand(case['offsets'].shape[0], 2, generator=gen, dtype=torch.float32).numpy()
    a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(),
         ex['emitter_u'].numpy(), ex['albedo'].numpy())
    gi = case['grad_image'].numpy()
    rng = np.random.default_rng(5)
    with harness.settings(normalize_warp_field=0, max_reparam_depth=0):
        gg, _, gp, _ = harness.render_direct_backward(*a, gi, ex['env'], bsdf_u=bu)
        tdata = rng.standard_normal(gg.shape).astype(np.float32)
        tp = np.array([0.3, -0.2, 0.5], np.float32)
        jd = harness.render_direct_forward_grad(*a, ex['env'], bsdf_u=bu, tangent=tdata)
        jp = harness.render_direct_forward_grad(*a, ex['env'], bsdf_u=bu, tangent_p=tp)
    lhs_d, rhs_d = float((jd.astype(np.float64) * gi).sum()), float((tdata.astype(np.float64) * gg).sum())
    lhs_p, rhs_p = float((jp.astype(np.float64) * gi).sum()), float((tp.astype(np.float64) * gp).sum())
    assert abs(lhs_d - rhs_d) <= 2e-3 * max(abs(rhs_d), 1e-6), (lhs_d, rhs_d)
    assert abs(lhs_p - rhs_p) <= 2e-3 * max(abs(rhs_p), 1e-6), (lhs_p, rhs_p)


def test_method_configs_reach_the_parameter_block(built):
    """`--configs warpprimary / warpnotnormalized` (python/configs.py) -> WarpField2D.apply -> dsdf_params."""
    import configs
    import dsdf
    p = configs.get_config('warp').get_warpfield(None).apply(dsdf.default_params())
    assert (p.normalize_warp_field, p.max_reparam_depth) == (1, -1)
    p = configs.get_config('warpprimary').get_warpfield(None).apply(dsdf.default_params())
    assert (p.normalize_warp_field, p.max_reparam_depth) == (1, 0)
    p = configs.get_config('warpnotnormalized').get_warpfield(None).apply(dsdf.default_params())
    assert (p.normalize_warp_field, p.max_reparam_depth) == (0, -1)
    assert abs(p.edge_eps - 0.01) < 1e-9 and p.weight_strategy == 6 and abs(p.clamping_thresh - 0.05) < 1e-9
