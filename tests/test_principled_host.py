"""The `principled` BSDF of sdf_direct_reparam's principled-* configs (/root/reference/python/opt_configs.py:288-299: base_color
and roughness volumes): kernel arithmetic compiled for the host (tests/harness, TEST-ONLY) against the oracle.

The plugin itself is third-party (Mitsuba 3 principled.cpp; not in the reference repository): oracle/sdf_oracle.py restates
its eval in the local shading frame, vector by vector, for the plugin defaults -- PARITY UNPINNED.  The kernel uses the
closed form in the four scalars (n.wi, n.wo, wi.wo, roughness) of csrc/dsdf_bsdf.h and hand-derived partials.
"""
import math

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


def _dirs(n, seed):
    gen = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 6, generator=gen, dtype=torch.float64)

    def hemi(a, b):
        z = 0.02 + 0.97 * a
        r = torch.sqrt(1 - z * z)
        return torch.stack([r * torch.cos(2 * math.pi * b), r * torch.sin(2 * math.pi * b), z], -1)
    return hemi(u[:, 0], u[:, 1]), hemi(u[:, 2], u[:, 3]), 0.1 + 0.7 * u[:, 4]


def test_principled_terms_host(harness):
    """Kd, Ks of dsdf_bsdf.h == the oracle's local-frame restatement; their partials == its autograd (projected onto the
    unit spheres of wi, wo: the closed form uses |wi| = |wo| = 1)."""
    wi, wo, r = _dirs(4000, 5)
    wi.requires_grad_(True); wo.requires_grad_(True); r.requires_grad_(True)
    zero, one = torch.zeros(wi.shape[0], 3, dtype=torch.float64), torch.ones(wi.shape[0], 3, dtype=torch.float64)
    ks = O.principled_eval(zero, r, wi, wo)[:, 0]
    kd = O.principled_eval(one, r, wi, wo)[:, 0] - ks
    xyur = torch.stack([wi[:, 2], wo[:, 2], (wi * wo).sum(-1), r], -1).detach()
    T = harness.principled_terms(xyur.float().numpy()).astype(np.float64)
    assert np.allclose(T[:, 0], kd.detach().numpy(), rtol=2e-5, atol=1e-7)
    assert np.allclose(T[:, 1], ks.detach().numpy(), rtol=3e-4, atol=1e-6)      # (GGX at roughness 0.1: alpha^2 = 1e-4 in fp32)
    ez = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    for k, f in ((2, kd), (6, ks)):
        gi, go, gr = torch.autograd.grad(f.sum(), (wi, wo, r), retain_graph=True)
        fx, fy, fu, fr = (torch.tensor(T[:, k + j]) for j in range(4))
        mi = fx[:, None] * ez + fu[:, None] * wo.detach()
        mo = fy[:, None] * ez + fu[:, None] * wi.detach()
        proj = lambda v, w: v - (v * w).sum(-1, keepdim=True) * w
        a, b = proj(gi, wi.detach()), proj(mi, wi.detach())
        assert float((a - b).norm() / a.norm()) < 2e-3, k
        a, b = proj(go, wo.detach()), proj(mo, wo.detach())
        assert float((a - b).norm() / a.norm()) < 2e-3, k
        assert float((gr - fr).norm() / gr.norm()) < 2e-3, k


def _principled_inputs(case):
    ex = direct_inputs(case)
    gen = torch.Generator().manual_seed(23)
    ex['roughness'] = torch.rand(5, 4, 6, 1, generator=gen, dtype=torch.float32) * 0.7 + 0.1        # variables.py:121 clamps to [0.1, 0.8]
    return ex


def _oracle(case, ex, dt, reparam=True, grads=True, p=None):
    data = case['grid'].to(dt).clone().requires_grad_(grads)
    alb = ex['albedo'].to(dt).clone().requires_grad_(grads)
    rough = ex['roughness'].to(dt).clone().requires_grad_(grads)
    img = O.render(O.Grid3d(data, p), O.Camera.from_params(cam_params(case), dtype=dt), case['W'], case['H'], case['spp'],
                   case['offsets'].to(dt), O.DIRECT, reparam, albedo=alb, emitter_u=ex['emitter_u'].to(dt),
                   env=torch.tensor(ex['env'], dtype=dt), roughness=rough)
    if not grads:
        return img
    (img * case['grad_image'].to(dt)).sum().backward()
    return img.detach(), data.grad, alb.grad, rough.grad


@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
def test_principled_forward_host(harness, name):
    case = make_case(name)
    ex = _principled_inputs(case)
    ref = _oracle(case, ex, torch.float64, reparam=False, grads=False)
    dif = O.render(O.Grid3d(case['grid']), O.Camera.from_params(cam_params(case)), case['W'], case['H'], case['spp'],
                   case['offsets'].double(), O.DIRECT, False, albedo=ex['albedo'].double(), emitter_u=ex['emitter_u'].double(),
                   env=torch.tensor(ex['env'], dtype=torch.float64))
    assert rel_l2(ref.numpy(), dif.numpy()) > 2e-3                              # not the diffuse image (the background is the same)
    for diff in (False, True):
        img = harness.render_direct_forward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                                            case['offsets'].numpy(), ex['emitter_u'].numpy(), ex['albedo'].numpy(), ex['env'],
                                            diff=diff, roughness=ex['roughness'].numpy())
        assert rel_l2(img, ref.numpy()) < FWD_TOL


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
def test_principled_backward_host(harness, name):
    """dL/d(sdf.data), dL/d(base_color), dL/d(roughness) of the hand-derived adjoint against the oracle's autograd, gated at
    2 x the oracle's own fp32-vs-fp64 gap like every other gradient of the path (tests/precision.py)."""
    case = make_case(name)
    ex = _principled_inputs(case)
    (img_ref, gd, ga, gr), tols = P.torch_gate(lambda dt: _oracle(case, ex, dt))
    gg, galb, _, img = harness.render_direct_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'],
                                                      case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
                                                      ex['albedo'].numpy(), case['grad_image'].numpy(), ex['env'],
                                                      roughness=ex['roughness'].numpy())
    grough = harness.last_grad_roughness
    assert rel_l2(img, img_ref) < FWD_TOL
    assert np.isfinite(gg).all() and np.isfinite(galb).all() and np.isfinite(grough).all()
    assert float(np.abs(grough).max()) > 0
    assert rel_l2(galb, ga) < tols[2], (rel_l2(galb, ga), tols[2])
    assert rel_l2(grough, gr) < tols[3], (rel_l2(grough, gr), tols[3])
    assert rel_l2(gg, gd) < tols[1], (rel_l2(gg, gd), tols[1])


def test_principled_translation_gradient_host(harness):
    case = make_case('blob32')
    ex = _principled_inputs(case)

    def oracle(dt):
        p = torch.zeros(3, dtype=dt, requires_grad=True)
        img = _oracle(case, ex, dt, grads=False, p=p)
        (img * case['grad_image'].to(dt)).sum().backward()
        return (p.grad,)
    (gp_ref,), (tol,) = P.torch_gate(oracle)
    _, _, gp, _ = harness.render_direct_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'],
                                                 case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
                                                 ex['albedo'].numpy(), case['grad_image'].numpy(), ex['env'],
                                                 roughness=ex['roughness'].numpy())
    assert rel_l2(gp, gp_ref) < tol, (rel_l2(gp, gp_ref), tol)


def test_principled_forward_mode_is_transpose_of_backward_host(harness):
    """`render_forward` with the principled BSDF: <J dtheta, G> = <dtheta, J^T G> for tangents on sdf.data and sdf.p."""
    case = make_case('blob32')
    ex = _principled_inputs(case)
    a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
         ex['albedo'].numpy())
    gi = case['grad_image'].numpy()
    rough = ex['roughness'].numpy()
    gg, _, gp, _ = harness.render_direct_backward(*a, gi, ex['env'], roughness=rough)
    rng = np.random.default_rng(5)
    tdata = rng.standard_normal(gg.shape).astype(np.float32)
    tp = np.array([0.3, -0.2, 0.5], np.float32)
    jd = harness.render_direct_forward_grad(*a, ex['env'], tangent=tdata, roughness=rough)
    jp = harness.render_direct_forward_grad(*a, ex['env'], tangent_p=tp, roughness=rough)
    lhs_d, rhs_d = float((jd.astype(np.float64) * gi).sum()), float((tdata.astype(np.float64) * gg).sum())
    lhs_p, rhs_p = float((jp.astype(np.float64) * gi).sum()), float((tp.astype(np.float64) * gp).sum())
    assert abs(lhs_d - rhs_d) <= 2e-3 * max(abs(rhs_d), 1e-6), (lhs_d, rhs_d)
    assert abs(lhs_p - rhs_p) <= 2e-3 * max(abs(rhs_p), 1e-6), (lhs_p, rhs_p)
    jdd = harness.render_direct_forward_grad(*a, ex['env'], tangent=tdata)          # (the diffuse tangent is another image)
    assert rel_l2(jd, jdd) > 1e-3
