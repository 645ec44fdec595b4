"""`use_mis` with the principled BSDF (/root/reference/python/integrators/sdf_direct_reparam.py:77-105 over the principled-* configs):
the extended build of the kernel arithmetic (-DDSDF_XF=1, host) against the oracle.

Principled::sample / ::pdf are third-party (Mitsuba 3 principled.cpp, microfacet.h; not in the reference repository) and restated
from the published plugin at its defaults, like ::eval -- PARITY UNPINNED.  What the tests hold:
  * the restatement is self-consistent: the pdf integrates to the accepted fraction of the samples and E[cos / pdf] = pi;
  * the kernel's sample / pdf equal the oracle's, per direction;
  * image, dL/d(sdf.data), dL/d(base_color), dL/d(roughness), dL/d(sdf.p) of the hand-derived adjoint -- including the derivative of
    the ATTACHED shading frame that bsdf.eval(ctx, si, bs.wo) with a fixed LOCAL wo brings in -- equal the oracle's autograd at the
    fp32 gates of every other gradient of the path."""
import math

import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import make_case
import precision as P
from conftest import rel_l2
from test_principled_host import _principled_inputs, cam_params

FWD_TOL = 1e-4


def test_oracle_sampling_is_consistent():
    torch.manual_seed(0)
    N = 400000
    for rough, wi in ((0.3, [0.3, 0.1, 0.9]), (0.6, [0.8, -0.2, 0.4]), (0.8, [-0.2, 0.5, 0.7])):
        wi = torch.tensor([wi], dtype=torch.float64)
        wi = (wi / wi.norm()).expand(N, 3)
        r = torch.full((N,), rough, dtype=torch.float64)
        u = torch.rand(N, 2, dtype=torch.float64)
        z, phi = u[:, 0], 2 * math.pi * u[:, 1]
        s = torch.sqrt(1 - z * z)
        wo = torch.stack([s * torch.cos(phi), s * torch.sin(phi), z], -1)
        integral = float(O.principled_pdf(r, wi, wo).mean() * 2 * math.pi)
        wos, p, act = O.principled_sample(r, wi, torch.rand(N, dtype=torch.float64), torch.rand(N, 2, dtype=torch.float64))
        accepted = float(act.double().mean())
        assert abs(integral - accepted) < 0.02, (rough, integral, accepted)          # (lost mass = reflections below the horizon)
        est = float(torch.where(act, wos[:, 2] / p.clamp(min=1e-300), torch.zeros_like(p)).mean())
        assert abs(est - math.pi * 1.0) < 0.05, (rough, est)                          # E[cos / pdf] over the accepted samples


def _oracle(case, ex, lobe, bsdf_u, dt, grads=True, p=None):
    data = case['grid'].to(dt).clone().requires_grad_(grads)
    alb = ex['albedo'].to(dt).clone().requires_grad_(grads)
    rough = ex['roughness'].to(dt).clone().requires_grad_(grads)
    img = O.render(O.Grid3d(data, p), O.Camera.from_params(cam_params(case), dtype=dt), case['W'], case['H'], case['spp'],
                   case['offsets'].to(dt), O.DIRECT, True, albedo=alb, emitter_u=ex['emitter_u'].to(dt),
                   env=torch.tensor(ex['env'], dtype=dt), roughness=rough, use_mis=True, bsdf_u=bsdf_u.to(dt), lobe_u=lobe.to(dt))
    if not grads:
        return img
    (img * case['grad_image'].to(dt)).sum().backward()
    return img.detach(), data.grad, alb.grad, rough.grad


def _samples(case):
    gen = torch.Generator().manual_seed(3)
    n = case['offsets'].shape[0]
    return torch.rand(n, generator=gen, dtype=torch.float32), torch.rand(n, 2, generator=gen, dtype=torch.float32)


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
def test_principled_mis_backward_host(harness_xf, name):
    case = make_case(name)
    ex = _principled_inputs(case)
    lobe, bu = _samples(case)
    (img_ref, gd, ga, gr), tols = P.torch_gate(lambda dt: _oracle(case, ex, lobe, bu, dt))
    plain = O.render(O.Grid3d(case['grid']), O.Camera.from_params(cam_params(case)), case['W'], case['H'], case['spp'],
                     case['offsets'].double(), O.DIRECT, False, albedo=ex['albedo'].double(), emitter_u=ex['emitter_u'].double(),
                     env=torch.tensor(ex['env'], dtype=torch.float64), roughness=ex['roughness'].double())
    assert rel_l2(img_ref, plain.numpy()) > 1e-3                     # MIS changes the estimate (same expectation, other samples)
    h = harness_xf
    h.set_transform(np.eye(4))
    h.set_lobe_samples(lobe.numpy())
    try:
        gg, galb, _, img = h.render_direct_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                                                    case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(),
                                                    case['grad_image'].numpy(), ex['env'], roughness=ex['roughness'].numpy(),
                                                    bsdf_u=bu.numpy())
        grough = h.last_grad_roughness
    finally:
        h.set_lobe_samples(None)
    assert rel_l2(img, img_ref) < FWD_TOL, rel_l2(img, img_ref)
    assert np.isfinite(gg).all() and np.isfinite(galb).all() and np.isfinite(grough).all()
    e = (rel_l2(gg, gd), rel_l2(galb, ga), rel_l2(grough, gr))
    print(f"principled+mis {name}: dL/d data {e[0]:.3e} (gate {tols[1]:.3e}), base_color {e[1]:.3e} ({tols[2]:.3e}), roughness {e[2]:.3e} ({tols[3]:.3e})")
    assert e[1] < tols[2] and e[2] < tols[3] and e[0] < tols[1], (e, tols)


def test_principled_mis_translation_gradient_host(harness_xf):
    case = make_case('blob32')
    ex = _principled_inputs(case)
    lobe, bu = _samples(case)

    def oracle(dt):
        p = torch.zeros(3, dtype=dt, requires_grad=True)
        img = _oracle(case, ex, lobe, bu, dt, grads=False, p=p)
        (img * case['grad_image'].to(dt)).sum().backward()
        return (p.grad,)
    (gp_ref,), (tol,) = P.torch_gate(oracle)
    h = harness_xf
    h.set_transform(np.eye(4))
    h.set_lobe_samples(lobe.numpy())
    try:
        _, _, gp, _ = h.render_direct_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                                               case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(),
                                               case['grad_image'].numpy(), ex['env'], roughness=ex['roughness'].numpy(), bsdf_u=bu.numpy())
    finally:
        h.set_lobe_samples(None)
    assert rel_l2(gp, gp_ref) < tol, (rel_l2(gp, gp_ref), tol)
