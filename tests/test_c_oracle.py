"""The plain-C oracle (oracle/dsdf_oracle.c, hand-written adjoint, fp32) against the torch
oracle (oracle/sdf_oracle.py, autograd, fp64): two independent restatements must agree."""
import numpy as np
import pytest
import torch

import c_oracle
import sdf_oracle as O
from cases import make_case, oracle_backward, oracle_forward
from conftest import rel_l2


@pytest.fixture(scope='module')
def clib(built):
    return c_oracle.load()


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_c_forward_matches_torch_oracle(clib, name, integ):
    case = make_case(name)
    ref, aux = oracle_forward(case, integ)
    img, st = c_oracle.render(clib, case['grid'].float().numpy(), O.Camera(case['origin']).params(), case['W'], case['H'],
                              case['spp'], case['offsets'].numpy(), integ)
    assert rel_l2(img, ref.numpy()) < 1e-4
    assert st['lanes'] == aux['lanes'] and st['hits'] == aux['hits'] and abs(st['steps'] - aux['steps']) <= 0.01 * aux['steps']


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
@pytest.mark.parametrize('reparam', [True, False])
def test_c_backward_matches_torch_oracle(clib, name, integ, reparam):
    case = make_case(name)
    gref = oracle_backward(case, integ, reparam).numpy()
    gg, img = c_oracle.render_backward(clib, case['grid'].float().numpy(), O.Camera(case['origin']).params(), case['W'],
                                       case['H'], case['spp'], case['offsets'].numpy(), case['grad_image'].numpy(), integ, reparam)
    if not reparam and integ == O.SILHOUETTE:
        assert np.abs(gg).max() == 0
        return
    assert rel_l2(gg, gref) < 3e-3        # fp32 noise floor of the estimator, see test_kernel_math_host.py


def test_c_oracle_against_golden(clib):
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'blob32.npz'))
    case = make_case('blob32')
    gg, img = c_oracle.render_backward(clib, case['grid'].float().numpy(), O.Camera(case['origin']).params(), case['W'],
                                       case['H'], case['spp'], case['offsets'].numpy(), case['grad_image'].numpy(), O.SILHOUETTE)
    assert rel_l2(img, gold['img_sil']) < 1e-4 and rel_l2(gg, gold['grad_sil']) < 3e-3


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('reparam', [True, False])
def test_c_direct_matches_torch_oracle(clib, name, reparam):
    """sdf_direct_reparam: image, dL/d(sdf.data) and dL/d(albedo) of the hand-written C adjoint against autograd."""
    from cases import direct_inputs, oracle_direct
    case = make_case(name)
    ex = direct_inputs(case)
    cam16 = O.Camera(case['origin']).params()
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
