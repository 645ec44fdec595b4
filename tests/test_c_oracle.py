"""The plain-C oracle (oracle/dsdf_oracle.c, hand-written adjoint; fp32 build and fp64 build) against the torch
oracle (oracle/sdf_oracle.py, autograd, fp64): two independent restatements must agree -- the fp64 builds to
~1e-7 on bit-identical inputs (which is what licenses the C build as THE oracle at BASELINE.json config sizes,
where the torch one is too slow), the fp32 build to within the measured fp32 floor."""
import numpy as np
import pytest
import torch

import c_oracle
import sdf_oracle as O
from cases import direct_inputs, make_case, oracle_backward, oracle_forward
import precision as P
from conftest import rel_l2


@pytest.fixture(scope='module')
def clib(built):
    return c_oracle.load()


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_c_forward_matches_torch_oracle(clib, name, integ):
    case = make_case(name)
    ref, aux = oracle_forward(case, integ)
    img, st = c_oracle.render(clib, case['grid'].float().numpy(), case['cam'].params(), case['W'], case['H'],
                              case['spp'], case['offsets'].numpy(), integ)
    assert rel_l2(img, ref.numpy()) < 1e-4
    assert st['lanes'] == aux['lanes'] and st['hits'] == aux['hits'] and abs(st['steps'] - aux['steps']) <= 0.01 * aux['steps']


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
@pytest.mark.parametrize('reparam', [True, False])
def test_c_backward_matches_torch_oracle(clib, name, integ, reparam):
    case = make_case(name)
    gref = oracle_backward(case, integ, reparam).numpy()
    gg, img = c_oracle.render_backward(clib, case['grid'].float().numpy(), case['cam'].params(), case['W'],
                                       case['H'], case['spp'], case['offsets'].numpy(), case['grad_image'].numpy(), integ, reparam)
    if not reparam and integ == O.SILHOUETTE:
        assert np.abs(gg).max() == 0
        return
    assert rel_l2(gg, gref) < 2e-3        # fp32 build: this IS the floor measurement (tests/precision.py), sanity bound only


def test_c_oracle_against_golden(clib):
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'blob32.npz'))
    case = make_case('blob32')
    gg, img = c_oracle.render_backward(clib, case['grid'].float().numpy(), case['cam'].params(), case['W'],
                                       case['H'], case['spp'], case['offsets'].numpy(), case['grad_image'].numpy(), O.SILHOUETTE)
    assert rel_l2(img, gold['img_sil']) < 1e-4 and rel_l2(gg, gold['grad_sil']) < 3e-3


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('reparam', [True, False])
def test_c_direct_matches_torch_oracle(clib, name, reparam):
    """sdf_direct_reparam: image, dL/d(sdf.data) and dL/d(albedo) of the hand-written C adjoint against autograd."""
    from cases import direct_inputs, oracle_direct
    case = make_case(name)
    ex = direct_inputs(case)
    cam16 = case['cam'].params()
    for hide in (False, True):
        ref = oracle_direct(case, ex, reparam=False, hide_emitters=hide)
        img = c_oracle.render_direct(clib, case['grid'].float().numpy(), cam16, case['W'], case['H'], case['spp'],
                                     case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(), ex['env'], hide)
        assert rel_l2(img, ref.numpy()) < 1e-4
    img_ref, gd, ga = oracle_direct(case, ex, reparam=reparam, grads=True)
    gg, galb, img = c_oracle.render_direct_backward(clib, case['grid'].float().numpy(), cam16, case['W'], case['H'], case['spp'],
                                                    case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(),
                                                    case['grad_image'].numpy(), ex['env'], reparam=reparam)
    assert rel_l2(img, img_ref.numpy()) < 1e-4
    assert rel_l2(galb, ga.numpy()) < 3e-3
    assert rel_l2(gg, gd.numpy()) < 3e-3


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
@pytest.mark.parametrize('mode', ['mis', 'detach_indirect_si', 'decouple_reparam', 'mis+decouple'])
def test_c64_direct_variants_match_torch_oracle(name, mode):
    """use_mis (sdf_direct_reparam.py:77-105) and the two shadow-origin properties (:44-47): the fp64 C build's hand-written
    adjoint == torch autograd of the function-by-function restatement, on bit-identical inputs."""
    case = make_case(name)
    ex = direct_inputs(case)
    gen = torch.Generator().manual_seed(3)
    bu = torch.rand(case['offsets'].shape[0], 2, generator=gen, dtype=torch.float32) if 'mis' in mode else None
    variant = 1 if 'detach' in mode else (2 if 'decouple' in mode else 0)
    kw = dict(detach_indirect_si=variant == 1, decouple_reparam=variant == 2)
    if bu is not None:
        kw.update(use_mis=True, bsdf_u=bu.double())
    data = case['grid'].clone().requires_grad_(True)
    alb = ex['albedo'].double().clone().requires_grad_(True)
    img = O.render(O.Grid3d(data), case['cam'], case['W'], case['H'], case['spp'], case['offsets'].double(), O.DIRECT, True, albedo=alb,
                   emitter_u=ex['emitter_u'].double(), env=torch.tensor(ex['env'], dtype=torch.float64), **kw)
    gd, ga = torch.autograd.grad((img * case['grad_image'].double()).sum(), (data, alb))
    gg, galb, ci = c_oracle.render_direct_backward(P.clib(True), case['grid'].float().numpy(), case['cam'].params(), case['W'], case['H'],
                                                   case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(),
                                                   case['grad_image'].numpy(), ex['env'], bsdf_u=None if bu is None else bu.numpy(),
                                                   variant=variant)
    assert rel_l2(ci, img.detach().numpy()) < 2e-7
    assert rel_l2(galb, ga.numpy()) < 1e-6 and rel_l2(gg, gd.numpy()) < 1e-6, (rel_l2(galb, ga.numpy()), rel_l2(gg, gd.numpy()))


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
@pytest.mark.parametrize('reparam', [True, False])
def test_c64_matches_torch_oracle(name, integ, reparam):
    """fp64 build of the C restatement (hand-written adjoint) == torch autograd oracle on bit-identical inputs."""
    case = make_case(name)
    gref = oracle_backward(case, integ, reparam).numpy()
    ref, _ = oracle_forward(case, integ)
    g64, img64 = P.c_backward(case, integ, reparam, True)
    assert rel_l2(img64, ref.numpy()) < 2e-7                      # (the C build keeps fp32-VALUED constants)
    if np.abs(gref).max() == 0:
        assert np.abs(g64).max() == 0
    else:
        assert rel_l2(g64, gref) < 1e-6, rel_l2(g64, gref)


@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_c64_matches_torch_oracle_at_c1(integ):
    """The same at BASELINE.json config C1 (64^3 sphere, 128^2, one view; spp 4): the largest size the torch oracle
    handles in seconds.  tests/test_gpu_config_size.py then uses the C build at C1/C2/C3."""
    case = P.config_case('C1_spp4')
    gt = P.torch_backward(case, integ, True)
    g64, img64 = P.c_backward(case, integ, True, True)
    assert rel_l2(g64, gt) < 1e-6, rel_l2(g64, gt)
    img_t = O.render(O.Grid3d(case['grid'].float().double()), case['cam'], case['W'], case['H'], case['spp'],
                     case['offsets'].double(), integ)
    assert rel_l2(img64, img_t.numpy()) < 2e-7


def test_c64_per_ray_outputs_match_torch_oracle():
    """A2 per ray: its_t, warp_t, warp_t_d, warp_weight, warp_weight_d, step counts of the fp64 C build against
    SDFBase.ray_intersect of the torch oracle."""
    case = make_case('blob32')
    cam = case['cam']
    pos = torch.rand(4000, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(3)) * case['W']
    o, d, maxt = cam.sample_ray(pos, case['W'], case['H'])
    ref = O.ray_intersect(O.Grid3d(case['grid']), o, d, maxt)
    out = c_oracle.trace(P.clib(True), case['grid'].float().numpy(), o.numpy(), d.numpy(), maxt.numpy())
    assert np.array_equal(out['steps'], ref['steps'].numpy())
    # (the C build keeps fp32-VALUED constants -- 1e-6f vs 1e-6 -- which the 1/denom^3 weights amplify in the derivatives)
    for k, tol in (('its_t', 1e-7), ('warp_t', 1e-7), ('warp_weight', 1e-5), ('warp_t_d', 2e-5), ('warp_weight_d', 2e-5)):
        a, b = out[k], ref[k].numpy()
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin), k
        assert rel_l2(a[fin], b[fin]) < tol, (k, rel_l2(a[fin], b[fin]))
