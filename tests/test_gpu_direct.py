"""GPU parity of sdf_direct_reparam (integrators/sdf_direct_reparam.py:16-75, emitter sampling) through the
C-ABI against the oracle, plus properties at GPU-only sample counts.  BSDF / emitter: this repo's spec
(include/dsdf.h: dsdf_shading).  Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import direct_inputs, make_case, oracle_direct
import precision as P
from conftest import rel_l2

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4


@pytest.fixture(scope='module')
def dsdf(built):
    import dsdf as m
    m.load()
    assert torch.cuda.is_available()
    return m


def setup(dsdf, case, ex, hide=False):
    grid = dsdf.SdfGrid(case['grid'].float().cuda())
    sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
    sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], hide_emitters=hide)
    return grid, sen, sh


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
@pytest.mark.parametrize('hide', [False, True])
def test_direct_forward_gpu(dsdf, name, hide):
    case = make_case(name)
    ex = direct_inputs(case)
    ref = oracle_direct(case, ex, reparam=False, hide_emitters=hide)
    grid, sen, sh = setup(dsdf, case, ex, hide)
    for skip in (True, False):
        img = dsdf.render_forward(grid, sen, case['spp'], offsets=case['offsets'].cuda(), integrator='sdf_direct_reparam',
                                  shading=sh, emitter_samples=ex['emitter_u'].cuda(), empty_space_skip=skip)[0]
        assert rel_l2(img.cpu(), ref) < FWD_TOL


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
@pytest.mark.parametrize('reparam', [True, False])
def test_direct_backward_gpu(dsdf, name, reparam):
    case = make_case(name)
    ex = direct_inputs(case)
    img_ref, gd, ga = oracle_direct(case, ex, reparam=reparam, grads=True)
    grid, sen, sh = setup(dsdf, case, ex)
    galb = torch.zeros_like(sh.albedo)
    gp = torch.zeros(3, device='cuda')
    gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                   integrator='sdf_direct_reparam', reparam=reparam, return_image=True, shading=sh,
                                   emitter_samples=ex['emitter_u'].cuda(), grad_albedo=galb, grad_p=gp)
    assert rel_l2(img[0].cpu(), img_ref) < FWD_TOL
    assert torch.isfinite(gg).all() and torch.isfinite(galb).all()
    r = P.reference_direct(case, ex, reparam)
    assert rel_l2(gd, r['gd']) < 1e-6 and rel_l2(ga, r['ga']) < 1e-6          # torch autograd == hand-written C adjoint (fp64)
    ea, ed = rel_l2(galb.cpu(), r['ga']), rel_l2(gg.cpu(), r['gd'])
    P.record('grad_direct', case=name, reparam=reparam, err_data=ed, err_albedo=ea, floor_data=r['floor_data'], floor_albedo=r['floor_albedo'])
    assert ea < r['tol_albedo'], (ea, r['tol_albedo'])
    assert ed < r['tol_data'], (ed, r['tol_data'])


def test_direct_builtin_sampler_and_batch(dsdf):
    """In-kernel sampler (film position = floats 0,1; emitter sample = floats 3,4 of the lane's PCG32 stream)
    equals explicit samples; a batch of views equals the single views."""
    case = make_case('blob48_rect')
    ex = direct_inputs(case)
    grid, _, sh = setup(dsdf, case, ex)
    sens = dsdf.get_regular_cameras(12, resx=case['W'], resy=case['H'])[:3]
    n = (case['W'] + 4) * (case['H'] + 4) * 4
    offs = torch.cat([torch.tensor(O.independent_sampler_2d(5 + i, n)) for i in range(3)]).cuda()
    emit = torch.cat([torch.tensor(O.independent_sampler_emitter_2d(5 + i, n)) for i in range(3)]).cuda()
    a = dsdf.render_forward(grid, sens, 4, seeds=[5, 6, 7], integrator='sdf_direct_reparam', shading=sh)
    b = dsdf.render_forward(grid, sens, 4, offsets=offs, integrator='sdf_direct_reparam', shading=sh, emitter_samples=emit)
    assert rel_l2(a.cpu(), b.cpu()) < 1e-6
    for i, s in enumerate(sens):
        one = dsdf.render_forward(grid, s, 4, seeds=[5 + i], integrator='sdf_direct_reparam', shading=sh)
        assert rel_l2(a[i].cpu(), one[0].cpu()) < 1e-6


def test_direct_known_answer_and_autograd(dsdf):
    """Convex object under a constant environment: radiance = albedo * L (GPU-only sample count); the
    autograd op delivers gradients for sdf.data and the albedo volume."""
    R, W, H = 64, 48, 48
    data = O.sphere_grid(R, radius=0.3).float().cuda().requires_grad_(True)
    grid = dsdf.SdfGrid(data)
    alb = torch.zeros(4, 4, 4, 3, device='cuda')
    alb[..., 0], alb[..., 1], alb[..., 2] = 0.8, 0.5, 0.2
    alb.requires_grad_(True)
    sh = dsdf.Shading(alb, 2.0, hide_emitters=True)
    sens = dsdf.get_regular_cameras(4, resx=W, resy=H)[:2]
    img = dsdf.render(data, grid, sens, spp=256, seed=3, spp_grad=64, seed_grad=9, integrator='sdf_direct_reparam', shading=sh)
    sil = dsdf.render_forward(grid, sens, 256, seeds=[3, 4])
    inside = sil[..., 0] > 0.999
    mean = img.detach()[inside].mean(0).cpu()
    assert torch.allclose(mean, 2.0 * torch.tensor([0.8, 0.5, 0.2]), rtol=0.02), mean
    assert float(img.detach()[0, 0, 0].abs().max()) == 0.0
    (img - 0.3).abs().mean().backward()
    assert data.grad is not None and torch.isfinite(data.grad).all() and data.grad.abs().sum() > 0
    assert alb.grad is not None and torch.isfinite(alb.grad).all() and alb.grad.abs().sum() > 0


def test_direct_gradient_vs_finite_differences_gpu(dsdf):
    """Reparameterised gradient against central differences of the un-reparameterised render with common
    random numbers (figures/result_utils.py:126-161), 2048 spp: w.r.t. a grid perturbation, the
    translation sdf.p and a uniform albedo scale."""
    R, W, H, spp = 64, 48, 48, 2048
    base = O.sphere_grid(R, radius=0.3).float().cuda()
    lin = torch.linspace(0, 1, R, device='cuda')
    z, y, x = torch.meshgrid(lin, lin, lin, indexing='ij')
    r = torch.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2).clamp(min=1e-3)
    direction = (-(x - 0.5) / r + 0.5).contiguous()
    torch.manual_seed(1)
    alb = (torch.rand(5, 5, 5, 3, device='cuda') * 0.6 + 0.2)
    sens = dsdf.get_regular_cameras(3, resx=W, resy=H)
    yy, xx = torch.meshgrid(torch.arange(H, device='cuda'), torch.arange(W, device='cuda'), indexing='ij')
    G = torch.stack([xx / W, yy / H, (xx + yy) / (W + H)], -1).float()[None].repeat(3, 1, 1, 1).contiguous()
    seeds = [11, 12, 13]
    sh = dsdf.Shading(alb, (1.0, 0.9, 0.8), hide_emitters=False)
    galb, gp = torch.zeros_like(alb), torch.zeros(3, device='cuda')
    grad = dsdf.render_backward(dsdf.SdfGrid(base), sens, spp, G, seeds=seeds, integrator='sdf_direct_reparam', shading=sh,
                                grad_albedo=galb, grad_p=gp)

    def L(data=base, shift=(0.0, 0.0, 0.0), scale=1.0):
        g = dsdf.SdfGrid(data).set_translation(shift)
        s2 = dsdf.Shading(alb * scale, (1.0, 0.9, 0.8), hide_emitters=False)
        return float((dsdf.render_forward(g, sens, spp, seeds=seeds, integrator='sdf_direct_reparam', reparam=False, shading=s2) * G).sum())
    eps = 2e-3
    fd = (L(base + eps * direction) - L(base - eps * direction)) / (2 * eps)
    ad = float((grad * direction).sum())
    assert abs(ad - fd) < 0.08 * abs(fd) + 1.0, (ad, fd)
    fd_p = [(L(shift=[eps if a == k else 0.0 for k in range(3)]) - L(shift=[-eps if a == k else 0.0 for k in range(3)])) / (2 * eps)
            for a in range(3)]
    scale = max(abs(v) for v in fd_p)
    assert all(abs(float(gp[a]) - fd_p[a]) < 0.10 * scale + 1.0 for a in range(3)), (gp.tolist(), fd_p)
    fd_a = (L(scale=1.01) - L(scale=0.99)) / 0.02                      # the render is linear in the albedo
    ad_a = float((galb * alb).sum())
    assert abs(ad_a - fd_a) < 0.01 * abs(fd_a) + 1e-3, (ad_a, fd_a)


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
@pytest.mark.parametrize('mode', ['mis', 'detach_indirect_si', 'decouple_reparam', 'mis+decouple'])
def test_direct_variants_gpu(dsdf, name, mode):
    """use_mis (sdf_direct_reparam.py:77-105: emitter + BSDF sampling, power heuristic) and the detach_indirect_si /
    decouple_reparam properties (:13-14, 44-47) through the C-ABI against the fp64 C oracle (its adjoint is pinned to torch
    autograd by tests/test_c_oracle.py); gates = 2 x (fp32 C build vs fp64 C build)."""
    import c_oracle
    case = make_case(name)
    ex = direct_inputs(case)
    gen = torch.Generator().manual_seed(3)
    bu = torch.rand(case['offsets'].shape[0], 2, generator=gen, dtype=torch.float32) if 'mis' in mode else None
    variant = 1 if 'detach' in mode else (2 if 'decouple' in mode else 0)
    a = (case['grid'].float().numpy(), case['cam'].params(), case['W'], case['H'], case['spp'], case['offsets'].numpy(), ex['emitter_u'].numpy(),
         ex['albedo'].numpy(), case['grad_image'].numpy(), ex['env'])
    kw = dict(bsdf_u=None if bu is None else bu.numpy(), variant=variant)
    gd64, ga64, img64 = c_oracle.render_direct_backward(P.clib(True), *a, **kw)
    gd32, ga32, _ = c_oracle.render_direct_backward(P.clib(False), *a, **kw)
    grid = dsdf.SdfGrid(case['grid'].float().cuda())
    sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
    sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], use_mis=bu is not None, detach_indirect_si=variant == 1, decouple_reparam=variant == 2)
    galb = torch.zeros_like(sh.albedo)
    bs = None if bu is None else bu.cuda()
    gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=case['offsets'].cuda(),
                                   integrator='sdf_direct_reparam', return_image=True, shading=sh, emitter_samples=ex['emitter_u'].cuda(),
                                   grad_albedo=galb, bsdf_samples=bs)
    assert rel_l2(img[0].cpu(), img64) < FWD_TOL
    fwd = dsdf.render_forward(grid, sen, case['spp'], offsets=case['offsets'].cuda(), integrator='sdf_direct_reparam', shading=sh,
                              emitter_samples=ex['emitter_u'].cuda(), bsdf_samples=bs)
    assert rel_l2(fwd[0].cpu(), img64) < FWD_TOL                               # primal pass: value-only traces, same image
    tol_d, tol_a = max(2 * rel_l2(gd32, gd64), 1e-4), max(2 * rel_l2(ga32, ga64), 1e-4)
    ea, ed = rel_l2(galb.cpu(), ga64), rel_l2(gg.cpu(), gd64)
    P.record('grad_direct_variant', case=name, mode=mode, err_data=ed, err_albedo=ea, tol_data=tol_d, tol_albedo=tol_a)
    assert ea < tol_a, (ea, tol_a)
    assert ed < tol_d, (ed, tol_d)


def test_direct_mis_builtin_sampler_and_known_answer(dsdf):
    """The built-in sampler's BSDF sample (floats 6,7 of the lane's stream) equals explicit samples; MIS is unbiased: a convex
    object under a constant environment radiates albedo * L with and without it."""
    case = make_case('blob48_rect')
    ex = direct_inputs(case)
    grid = dsdf.SdfGrid(case['grid'].float().cuda())
    sh = dsdf.Shading(ex['albedo'].cuda(), ex['env'], use_mis=True)
    sens = dsdf.get_regular_cameras(12, resx=case['W'], resy=case['H'])[:2]
    n = (case['W'] + 4) * (case['H'] + 4) * 4
    offs = torch.cat([torch.tensor(O.independent_sampler_2d(5 + i, n)) for i in range(2)]).cuda()
    emit = torch.cat([torch.tensor(O.independent_sampler_emitter_2d(5 + i, n)) for i in range(2)]).cuda()
    bsdf = torch.cat([torch.tensor(O.independent_sampler_bsdf_2d(5 + i, n)) for i in range(2)]).cuda()
    a = dsdf.render_forward(grid, sens, 4, seeds=[5, 6], integrator='sdf_direct_reparam', shading=sh)
    b = dsdf.render_forward(grid, sens, 4, offsets=offs, integrator='sdf_direct_reparam', shading=sh, emitter_samples=emit, bsdf_samples=bsdf)
    assert rel_l2(a.cpu(), b.cpu()) < 1e-6
    R, W, H = 64, 48, 48
    g2 = dsdf.SdfGrid(O.sphere_grid(R, radius=0.3).float().cuda())
    alb = torch.zeros(4, 4, 4, 3, device='cuda')
    alb[..., 0], alb[..., 1], alb[..., 2] = 0.8, 0.5, 0.2
    s2 = dsdf.get_regular_cameras(4, resx=W, resy=H)[:2]
    sil = dsdf.render_forward(g2, s2, 256, seeds=[3, 4])
    inside = sil[..., 0] > 0.999
    for mis in (False, True):
        img = dsdf.render_forward(g2, s2, 256, seeds=[3, 4], integrator='sdf_direct_reparam',
                                  shading=dsdf.Shading(alb, 2.0, hide_emitters=True, use_mis=mis))
        mean = img[inside].mean(0).cpu()
        assert torch.allclose(mean, 2.0 * torch.tensor([0.8, 0.5, 0.2]), rtol=0.02), (mis, mean)


@pytest.mark.parametrize('mode', ['plain', 'mis', 'mis+decouple'])
def test_direct_forward_mode_gpu(dsdf, mode):
    """`render_forward` of sdf_direct_reparam through the C-ABI (dsdf_render_forward_grad): transpose identity
    <J dtheta, G> = <dtheta, J^T G> against dsdf_render_backward on the same samples, for tangents on sdf.data and sdf.p, and the
    translation tangent against central differences of the un-reparameterised render (figures/result_utils.py:126-161)."""
    R, W, H = 48, 32, 32
    data = O.blob_grid(R, n=10, seed=3).float().cuda()
    grid = dsdf.SdfGrid(data)
    sens = dsdf.get_regular_cameras(6, resx=W, resy=H)[:2]
    torch.manual_seed(2)
    alb = torch.rand(6, 5, 4, 3, device='cuda') * 0.6 + 0.2
    sh = dsdf.Shading(alb, (1.0, 0.9, 0.8), use_mis='mis' in mode, decouple_reparam='decouple' in mode)
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
    if mode == 'plain':
        # gradient image w.r.t. a translation along x at 2048 spp vs central differences with common random numbers
        spp = 2048
        yy, xx = torch.meshgrid(torch.arange(H, device='cuda'), torch.arange(W, device='cuda'), indexing='ij')
        Gw = torch.stack([xx / W, yy / H, (xx + yy) / (W + H)], -1).float()[None].repeat(2, 1, 1, 1)
        fwd = dsdf.render_forward_grad(grid, sens, spp, tangent_p=[1.0, 0.0, 0.0], seeds=[11, 12], integrator='sdf_direct_reparam', shading=sh)
        eps = 2e-3

        def L(shift):
            g = dsdf.SdfGrid(data).set_translation([shift, 0.0, 0.0])
            return float((dsdf.render_forward(g, sens, spp, seeds=[11, 12], integrator='sdf_direct_reparam', reparam=False, shading=sh) * Gw).sum())
        fd = (L(eps) - L(-eps)) / (2 * eps)
        ad = float((fwd * Gw).sum())
        assert abs(ad - fd) < 0.10 * abs(fd) + 1.0, (ad, fd)
