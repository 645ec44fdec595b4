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


# ---- use_mis (sdf_direct_reparam.py:77-105) and the detach_indirect_si / decouple_reparam properties (:13-14, 44-47)
def _bsdf_u(case, seed=3):
    gen = torch.Generator().manual_seed(seed)
    return torch.rand(case['offsets'].shape[0], 2, generator=gen, dtype=torch.float32)


def test_bsdf_sampler_host(harness):
    """floats 6,7 of the lane's PCG32 stream (film 0,1; wavelength 2; emitter 3,4; bsdf.sample's next_1d 5)."""
    assert np.array_equal(harness.sampler_bsdf(91, 4000), O.independent_sampler_bsdf_2d(91, 4000))


def test_cosine_hemisphere_and_frame():
    """The restated Mitsuba conventions of the BSDF-sampling branch: the concentric-disk cosine warp maps the unit square onto
    the upper unit hemisphere with density cos / pi; coordinate_system gives an orthonormal right-handed frame."""
    u = torch.rand(20000, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    w = O.square_to_cosine_hemisphere(u)
    assert torch.allclose(w.norm(dim=-1), torch.ones(len(w), dtype=torch.float64), atol=1e-12) and (w[:, 2] >= 0).all()
    assert abs(float(w[:, 2].mean()) - 2.0 / 3.0) < 5e-3                       # E[cos] under the density cos / pi
    n = torch.nn.functional.normalize(torch.randn(1000, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(2)), dim=-1)
    s, t = O.coordinate_system(n)
    for a, b in ((s, t), (s, n), (t, n)):
        assert float(O.dot(a, b).abs().max()) < 1e-12
    assert torch.allclose(torch.cross(s, t, dim=-1), n, atol=1e-12)


def test_oracle_mis_is_unbiased():
    """Emitter sampling and its MIS combination with BSDF sampling estimate the same integral: convex object under a constant
    environment -> both give albedo * L on the interior pixels."""
    torch.manual_seed(0)
    W = H = 24
    spp = 64
    n = (W + 4) * (H + 4) * spp
    cam = O.Camera(O.regular_camera_origins(4)[1])
    alb = torch.zeros(3, 3, 3, 3, dtype=torch.float64)
    alb[..., 0], alb[..., 1], alb[..., 2] = 0.8, 0.5, 0.2
    offs, eu, bu = (torch.rand(n, 2, dtype=torch.float64) for _ in range(3))
    g = O.Grid3d(O.sphere_grid(32, radius=0.3))
    img = O.render(g, cam, W, H, spp, offs, O.DIRECT, reparam=False, albedo=alb, emitter_u=eu, env=2.0, hide_emitters=True, use_mis=True, bsdf_u=bu)
    sil = O.render(g, cam, W, H, spp, offs, O.SILHOUETTE, reparam=False)
    inside = sil[..., 0] > 0.999
    assert torch.allclose(img[inside].mean(0), 2.0 * torch.tensor([0.8, 0.5, 0.2], dtype=torch.float64), rtol=0.05)


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('mode', ['mis', 'detach_indirect_si', 'decouple_reparam', 'mis+decouple'])
def test_direct_variants_host(harness, name, mode):
    """Kernel arithmetic (host build) of use_mis and of the two shadow-origin properties against the fp64 C oracle, whose
    hand-written adjoint test_c_oracle.py pins to torch autograd; gates = 2 x (fp32 C build vs fp64 C build)."""
    import c_oracle
    case = make_case(name)
    ex = direct_inputs(case)
    bu = _bsdf_u(case) if 'mis' in mode else None
    variant = 1 if 'detach' in mode else (2 if 'decouple' in mode else 0)
    a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
         ex['albedo'].numpy(), case['grad_image'].numpy(), ex['env'])
    kw = dict(bsdf_u=None if bu is None else bu.numpy(), variant=variant)
    gd64, ga64, img64 = c_oracle.render_direct_backward(P.clib(True), *a, **kw)
    gd32, ga32, _ = c_oracle.render_direct_backward(P.clib(False), *a, **kw)
    gg, galb, _, img = harness.render_direct_backward(*a[:8], a[8], a[9], bsdf_u=None if bu is None else bu.numpy(), variant=variant)
    assert rel_l2(img, img64) < FWD_TOL
    tol_d, tol_a = max(2 * rel_l2(gd32, gd64), 1e-4), max(2 * rel_l2(ga32, ga64), 1e-4)
    assert rel_l2(galb, ga64) < tol_a, (rel_l2(galb, ga64), tol_a)
    assert rel_l2(gg, gd64) < tol_d, (rel_l2(gg, gd64), tol_d)
    if mode != 'mis':                                                          # the variants change the gradient, never the image
        base = c_oracle.render_direct_backward(P.clib(True), *a, bsdf_u=kw['bsdf_u'])
        assert rel_l2(img64, base[2]) < 1e-12
        if name == 'blob32':                                                   # (the shadow-ray warp is active on this case only)
            assert rel_l2(gd64, base[0]) > 1e-4
    # primal pass (value-only traces) renders the same image
    fwd = harness.render_direct_forward(*a[:8], a[9], bsdf_u=None if bu is None else bu.numpy(), variant=variant)
    assert rel_l2(fwd, img64) < FWD_TOL


@pytest.mark.parametrize('mode', ['plain', 'mis', 'detach_indirect_si', 'mis+decouple'])
def test_direct_forward_mode_is_transpose_of_backward_host(harness, mode):
    """`render_forward` of sdf_direct_reparam (integrators/reparam.py:192-196): <J dtheta, G> = <dtheta, J^T G> for a tangent on
    sdf.data and on sdf.p -- forward mode and the hand-derived adjoint are transposes of each other on the same samples."""
    case = make_case('blob32')
    ex = direct_inputs(case)
    bu = _bsdf_u(case).numpy() if 'mis' in mode else None
    variant = 1 if 'detach' in mode else (2 if 'decouple' in mode else 0)
    a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
         ex['albedo'].numpy())
    gi = case['grad_image'].numpy()
    gg, _, gp, _ = harness.render_direct_backward(*a, gi, ex['env'], bsdf_u=bu, variant=variant)
    rng = np.random.default_rng(5)
    tdata = rng.standard_normal(gg.shape).astype(np.float32)
    tp = np.array([0.3, -0.2, 0.5], np.float32)
    jd = harness.render_direct_forward_grad(*a, ex['env'], bsdf_u=bu, variant=variant, tangent=tdata)
    jp = harness.render_direct_forward_grad(*a, ex['env'], bsdf_u=bu, variant=variant, tangent_p=tp)
    lhs_d, rhs_d = float((jd.astype(np.float64) * gi).sum()), float((tdata.astype(np.float64) * gg).sum())
    lhs_p, rhs_p = float((jp.astype(np.float64) * gi).sum()), float((tp.astype(np.float64) * gp).sum())
    assert abs(lhs_d - rhs_d) <= 2e-3 * max(abs(rhs_d), 1e-6), (lhs_d, rhs_d)
    assert abs(lhs_p - rhs_p) <= 2e-3 * max(abs(rhs_p), 1e-6), (lhs_p, rhs_p)
