"""Pins the oracle against everything the reference repository itself holds for
this path: closed-form SDFs (python/shapes.py:494-514), B-spline
identities, primal invariance of the estimator (python/warp.py:81-83, 95,
114-115), masked-lane semantics (python/shapes.py:278-283, python/warp.py:91-93)
and finite differences in the style of figures/result_utils.py:126-161.  The
reference ships no golden vectors; Mitsuba/Dr.Jit cannot be imported here."""
import math

import numpy as np
import pytest
import torch

import sdf_oracle as O


def test_bspline_partition_of_unity():
    a = torch.linspace(0, 1, 101, dtype=torch.float64)
    w, dw, ddw = O.bspline_weights(a)
    assert torch.allclose(w.sum(-1), torch.ones_like(a), atol=1e-14)
    assert torch.allclose(dw.sum(-1), torch.zeros_like(a), atol=1e-14)
    assert torch.allclose(ddw.sum(-1), torch.zeros_like(a), atol=1e-14)
    # first moment: sum_k w_k (k-1) = a  (linear precision of the cubic B-spline)
    k = torch.tensor([-1.0, 0.0, 1.0, 2.0], dtype=torch.float64)
    assert torch.allclose((w * k).sum(-1), a, atol=1e-14)


def test_cubic_reproduces_linear_field():
    R = 16
    z, y, x = np.meshgrid(*[(np.arange(R) + 0.5) / R] * 3, indexing='ij')     # texel centres
    a, b, c, e = 0.3, -1.2, 0.7, 0.05
    data = torch.tensor(a * x + b * y + c * z + e)
    p = torch.rand(500, 3, dtype=torch.float64) * 0.6 + 0.2                   # away from clamped borders
    v, g, H = O.eval_cubic(data, p, 2)
    assert torch.allclose(v, a * p[:, 0] + b * p[:, 1] + c * p[:, 2] + e, atol=1e-12)
    assert torch.allclose(g, torch.tensor([a, b, c], dtype=torch.float64).expand_as(g), atol=1e-10)
    assert H.abs().max() < 1e-8


def test_cubic_matches_sphere_closed_form():
    R = 64
    z, y, x = np.meshgrid(*[(np.arange(R) + 0.5) / R] * 3, indexing='ij')
    data = torch.tensor(np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) - 0.3)
    p = torch.rand(500, 3, dtype=torch.float64) * 0.5 + 0.25
    p = p[(p - 0.5).norm(dim=-1) > 0.1]
    sph = O.SphereSDF(torch.tensor([0.5, 0.5, 0.5], dtype=torch.float64), 0.3)
    v, _, g, _, H = O.Grid3d(data).eval_all(p)
    vs, _, gs, _, Hs = sph.eval_all(p)
    assert (v - vs).abs().max() < 2e-3          # B-spline smoothing of a curved field
    assert (g - gs).abs().max() < 2e-2
    assert (H - Hs).abs().max() < 0.6


def test_clamped_border_is_constant():
    data = torch.rand(8, 8, 8, dtype=torch.float64)
    p = torch.tensor([[-0.4, 0.5, 0.5], [-0.3, 0.5, 0.5]], dtype=torch.float64)
    v, g, _ = O.eval_cubic(data, p, 1)
    assert abs(float(v[0] - v[1])) < 1e-14 and abs(float(g[0, 0])) < 1e-14


def test_analytic_sphere_hit_distance():
    sph = O.SphereSDF(torch.tensor([0.5, 0.5, 0.5], dtype=torch.float64), 0.3)
    cam = O.Camera(O.regular_camera_origins(1)[0])
    pos = torch.rand(400, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(11)) * 32
    o, d, maxt = cam.sample_ray(pos, 32, 32)
    tr = O.ray_intersect(sph, o, d, maxt)
    oc = o - 0.5
    b = O.dot(oc, d)
    disc = b * b - (O.dot(oc, oc) - 0.09)
    hit = disc > 1e-3          # exclude near-grazing rays (error ~ trace_eps / sin(angle))
    t_exact = -b - torch.sqrt(disc.clamp(min=0))
    assert bool((torch.isfinite(tr['its_t']) == (disc > 0))[disc.abs() > 1e-4].all())
    assert (tr["its_t"][hit] - t_exact[hit]).abs().max() < 5e-5   # trace_eps / sin(grazing angle)
    nd = O.ray_intersect_non_diff(sph, o, d, maxt)
    assert torch.equal(torch.isfinite(nd['its_t']), torch.isfinite(tr['its_t']))
    assert (nd['its_t'][hit] - tr['its_t'][hit]).abs().max() < 1e-12


def _small():
    torch.manual_seed(3)
    R, W, H, spp = 24, 16, 16, 4
    grid = O.blob_grid(R, n=5, seed=2)
    cam = O.Camera(O.regular_camera_origins(6)[2])
    offs = torch.rand((W + 4) * (H + 4) * spp, 2, dtype=torch.float64)
    return grid, cam, W, H, spp, offs


@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_primal_invariance(integ):
    """F8: the image is identical with WarpField2D and DummyWarpField."""
    grid, cam, W, H, spp, offs = _small()
    a = O.render(O.Grid3d(grid), cam, W, H, spp, offs, integ, reparam=True)
    b = O.render(O.Grid3d(grid), cam, W, H, spp, offs, integ, reparam=False)
    assert (a - b).abs().max() < 1e-12


def test_invalid_lanes_masked():
    """bbox miss / weight_sum < 1e-7 -> warp_t = inf, derivatives 0 (shapes.py:278-283)."""
    grid = O.sphere_grid(16)
    o = torch.tensor([[3.0, 3.0, 3.0], [0.5, 0.5, -2.0]], dtype=torch.float64)
    d = torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    tr = O.ray_intersect(O.Grid3d(grid), o, d, torch.full((2,), 1e4, dtype=torch.float64))
    assert math.isinf(float(tr['warp_t'][0])) and float(tr['warp_weight'][0]) == 0
    assert tr['warp_t_d'][0].abs().max() == 0 and tr['warp_weight_d'][0].abs().max() == 0
    assert math.isinf(float(tr['its_t'][0])) and math.isfinite(float(tr['its_t'][1]))
    assert abs(float(tr["its_t"][1]) - (2.0 + 0.2)) < 0.04   # half-voxel convention at R=16


def test_pcg32_known_answer():
    """pcg32 reference vector (pcg-c demo: initstate 42, initseq 54)."""
    inc = np.array([(54 << 1) | 1], np.uint64)
    state = np.zeros(1, np.uint64)
    state, _ = O._pcg32_step(state, inc)
    state = state + np.uint64(42)
    state, _ = O._pcg32_step(state, inc)
    outs = []
    for _ in range(6):
        state, u = O._pcg32_step(state, inc)
        outs.append(int(u[0]))
    assert outs == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_sampler_uniform():
    r = O.independent_sampler_2d(7, 20000)
    assert r.min() >= 0 and r.max() < 1
    assert abs(r.mean() - 0.5) < 0.01 and abs(np.corrcoef(r[:, 0], r[:, 1])[0, 1]) < 0.03


def test_develop_and_put_partition():
    """A constant value splatted everywhere develops to that constant."""
    W = H = 8
    spp = 16
    offs = torch.rand((W + 4) * (H + 4) * spp, 2, dtype=torch.float64)
    uv = O.lane_positions(W, H, spp, offs)
    vals = torch.cat([torch.full((uv.shape[0], 3), 0.7, dtype=torch.float64), torch.ones(uv.shape[0], 1, dtype=torch.float64)], 1)
    block = O.block_put(torch.zeros((W + 4) * (H + 4) * 4, dtype=torch.float64), uv, vals, W + 4, H + 4)
    img = O.develop(block, W, H)
    assert (img - 0.7).abs().max() < 1e-12


@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_gradient_vs_finite_differences(integ):
    """figures/result_utils.py:126-161: FD on the translation `sdf.p` (common random
    numbers) against the reparameterised derivative, reduced with a smooth test image."""
    torch.manual_seed(0)
    R, W, H, spp = 32, 16, 16, 512
    g = O.sphere_grid(R)
    cam = O.Camera(O.regular_camera_origins(1)[0])
    offs = torch.rand((W + 4) * (H + 4) * spp, 2, dtype=torch.float64)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing='ij')
    G = torch.stack([xx / W, yy / H, (xx + yy) / (W + H)], -1)
    p = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    (O.render(O.Grid3d(g, p), cam, W, H, spp, offs, integ) * G).sum().backward()
    eps = 4e-3
    for axis in (0, 2):
        e = torch.zeros(3, dtype=torch.float64); e[axis] = eps
        with torch.no_grad():
            lp = (O.render(O.Grid3d(g, e), cam, W, H, spp, offs, integ, reparam=False) * G).sum()
            lm = (O.render(O.Grid3d(g, -e), cam, W, H, spp, offs, integ, reparam=False) * G).sum()
        fd = float((lp - lm) / (2 * eps))
        ad = float(p.grad[axis])
        assert abs(ad - fd) < 0.08 * abs(fd) + 0.5, (axis, ad, fd)
