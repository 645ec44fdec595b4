"""sdf_direct_reparam (integrators/sdf_direct_reparam.py:16-75, emitter sampling only): the kernel arithmetic
compiled for the host (tests/harness, TEST-ONLY) against the oracle's restatement and its autograd.
BSDF / emitter are this repo's spec (diffuse over a trilinear albedo volume, constant environment emitter:
oracle/sdf_oracle.py header), since the reference's scene files are not part of its repository."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import direct_inputs, make_case, oracle_direct
import precision as P
from conftest import rel_l2

FWD_TOL = 1e-4


def cam_params(case):
    return O.Camera(case['origin']).params()


def test_emitter_sampler_host(harness):
    """floats 3,4 of the lane's PCG32 stream (film position = 0,1; wavelength sample = 2)."""
    ref = O.independent_sampler_emitter_2d(77, 5000)
    assert np.array_equal(harness.sampler_emitter(77, 5000), ref)
    assert np.array_equal(O.independent_sampler(77, 100, 5)[:, :2], O.independent_sampler_2d(77, 100))


def test_oracle_direct_known_answer():
    """A convex object under a constant environment is unoccluded: outgoing radiance = albedo * L
    (irradiance pi L, diffuse BRDF albedo / pi) -- Monte Carlo mean over the interior pixels."""
    torch.manual_seed(0)
    W = H = 24
    spp = 64
    n = (W + 4) * (H + 4) * spp
    cam = O.Camera(O.regular_camera_origins(4)[1])
    alb = torch.zeros(3, 3, 3, 3, dtype=torch.float64)
    alb[..., 0], alb[..., 1], alb[..., 2] = 0.8, 0.5, 0.2
    offs, eu = torch.rand(n, 2, dtype=torch.float64), torch.rand(n, 2, dtype=torch.float64)
    g = O.Grid3d(O.sphere_grid(32, radius=0.3))
    img = O.render(g, cam, W, H, spp, offs, O.DIRECT, reparam=False, albedo=alb, emitter_u=eu, env=2.0, hide_emitters=True)
    sil = O.render(g, cam, W, H, spp, offs, O.SILHOUETTE, reparam=False)
    inside = sil[..., 0] > 0.999
    assert inside.sum() > 20
    assert torch.allclose(img[inside].mean(0), 2.0 * torch.tensor([0.8, 0.5, 0.2], dtype=torch.float64), rtol=0.08)
    assert float(img[0, 0].abs().max()) == 0.0                      # hide_emitters: black background
    bg = O.render(g, cam, W, H, spp, offs, O.DIRECT, reparam=False, albedo=alb, emitter_u=eu, env=2.0, hide_emitters=False)
    assert torch.allclose(bg[0, 0], torch.full((3,), 2.0, dtype=torch.float64))


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('hide', [False, True])
def test_direct_forward_host(harness, name, hide):
    case = make_case(name)
    ex = direct_inputs(case)
    ref = oracle_direct(case, ex, reparam=False, hide_emitters=hide)
    for diff in (False, True):                                      # primal pass / gradient-pass forward sweep (F8)
        img = harness.render_direct_forward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                                            case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(), ex['env'],
                                            hide_emitters=hide, diff=diff)
        assert rel_l2(img, ref) < FWD_TOL


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('reparam', [True, False])
def test_direct_backward_host(harness, name, reparam):
    """dL/d(sdf.data) and dL/d(albedo) of the hand-derived adjoint against the oracle's autograd."""
    case = make_case(name)
    ex = direct_inputs(case)
    img_ref, gd, ga = oracle_direct(case, ex, reparam=reparam, grads=True)
    gg, galb, _, img = harness.render_direct_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'],
                                                      case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
                                                      ex['albedo'].numpy(), case['grad_image'].numpy(), ex['env'], reparam=reparam)
    assert rel_l2(img, img_ref) < FWD_TOL
    assert np.isfinite(gg).all() and np.isfinite(galb).all()
    r = P.reference_direct(case, ex, reparam)                                      # per-case gates: 2 x measured fp32 floor
    assert rel_l2(galb, ga) < r['tol_albedo'], (rel_l2(galb, ga), r['tol_albedo'])
    assert rel_l2(gg, gd) < r['tol_data'], (rel_l2(gg, gd), r['tol_data'])


def test_direct_translation_gradient_host(harness):
    case = make_case('blob32')
    ex = direct_inputs(case)

    def oracle(dt):
        p = torch.zeros(3, dtype=dt, requires_grad=True)
        img = O.render(O.Grid3d(case['grid'].to(dt), p), O.Camera.from_params(cam_params(case), dtype=dt), case['W'], case['H'],
                       case['spp'], case['offsets'].to(dt), O.DIRECT, albedo=ex['albedo'].to(dt), emitter_u=ex['emitter_u'].to(dt),
                       env=torch.tensor(ex['env'], dtype=dt))
        (img * case['grad_image'].to(dt)).sum().backward()
        return (p.grad,)
    (gp_ref,), (tol,) = P.torch_gate(oracle)
    _, _, gp, _ = harness.render_direct_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'],
                                                 case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
                                                 ex['albedo'].numpy(), case['grad_image'].numpy(), ex['env'])
    assert rel_l2(gp, gp_ref) < tol, (rel_l2(gp, gp_ref), tol)
