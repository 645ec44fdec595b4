"""External witnesses for the third-party layer ("the third-party layer is restated three
times by one author").  Everything here is checked against code that NONE of the repo's authors wrote and that ships with this
image: scipy's spline evaluators (ndimage.map_coordinates / interpolate.BSpline), scipy's Euclidean distance transform and
torch.optim.Adam.  What they pin:

  C.1  Dr.Jit `Texture3f` cubic lookups (`/root/reference/python/shapes.py:421-450`): the uniform cubic B-spline WITHOUT a prefilter
       on texel centres at (i + 0.5) / res with clamped indices -- value against `map_coordinates(order=3, prefilter=False,
       mode='nearest')`, the derivative weights against `BSpline.basis_element(...).derivative(k)`, gradient / Hessian of the
       lookup against the tensor product of those.  Legs: the fp64 oracle, the C oracle, the host build of the kernel arithmetic
       and (GPU) `dsdf_eval_cubic` through the C-ABI.
  C.7  `mi.ad.Adam` (`shape_opt.py:53`, `variables.py:183`): `variables.Adam` against `torch.optim.Adam` (the two differ only in
       where epsilon enters the bias-corrected step; bounded below).
  C.8  `fastsweep.redistance` (`redistancing.py:7`): the C restatement and (GPU) `dsdf_redistance` against scipy's exact Euclidean
       distance transform of the same sign field (first-order agreement, sign preservation).

What stays without an external witness: the sensor / film / sampler conventions (C.2 - C.4; only PCG32 has a published vector,
tests/test_oracle_known_answers.py) and the BSDF plugins."""
import numpy as np
import pytest
import torch
from scipy import ndimage
from scipy.interpolate import BSpline

import sdf_oracle as O
from conftest import rel_l2


# ------------------------------------------------------------------------------------------------ scipy's cubic B-spline
def scipy_cubic_value(data, pts):
    """`data` (Z, Y, X), unit-cube points (n, 3) = (x, y, z) -> the un-prefiltered cubic B-spline through scipy.ndimage."""
    Z, Y, X = data.shape
    res = np.array([X, Y, Z], np.float64)
    pf = np.asarray(pts, np.float64) * res - 0.5                      # texel space, (x, y, z)
    return ndimage.map_coordinates(np.asarray(data, np.float64), [pf[:, 2], pf[:, 1], pf[:, 0]], order=3, mode='nearest', prefilter=False)


_B3 = BSpline.basis_element(np.arange(5.0), extrapolate=False)       # the cardinal cubic B-spline on [0, 4], centred at 2
_B3D = [_B3, _B3.derivative(1), _B3.derivative(2)]


def scipy_weights(alpha, k):
    """k-th derivative (w.r.t. alpha) of the four tap weights at fractional offset alpha: tap j (index floor - 1 + j) sits at distance
    alpha + 1 - j from the sample, i.e. at argument 2 + (alpha + 1 - j) of the centred basis element."""
    a = np.asarray(alpha, np.float64)
    return np.stack([np.nan_to_num(_B3D[k](2.0 + a + 1.0 - j)) for j in range(4)], -1)


def scipy_cubic_all(data, pts):
    """value, gradient (x res) and Hessian (x res_i res_j) of the lookup: tensor products of scipy's basis functions over the
    clamped 4^3 neighbourhood."""
    data = np.asarray(data, np.float64)
    Z, Y, X = data.shape
    res = np.array([X, Y, Z], np.float64)
    pf = np.asarray(pts, np.float64) * res - 0.5
    fl = np.floor(pf)
    al = pf - fl
    i0 = fl.astype(np.int64) - 1
    offs = np.arange(4)
    ix = np.clip(i0[:, 0:1] + offs, 0, X - 1); iy = np.clip(i0[:, 1:2] + offs, 0, Y - 1); iz = np.clip(i0[:, 2:3] + offs, 0, Z - 1)
    taps = data[iz[:, :, None, None], iy[:, None, :, None], ix[:, None, None, :]]          # [n, kz, jy, ix]
    w = [[scipy_weights(al[:, a], k) for k in range(3)] for a in range(3)]                 # w[axis][derivative]
    c = lambda kz, ky, kx: np.einsum('nkji,nk,nj,ni->n', taps, w[2][kz], w[1][ky], w[0][kx])
    v = c(0, 0, 0)
    g = np.stack([c(0, 0, 1) * X, c(0, 1, 0) * Y, c(1, 0, 0) * Z], -1)
    H6 = np.stack([c(0, 0, 2) * X * X, c(0, 2, 0) * Y * Y, c(2, 0, 0) * Z * Z, c(0, 1, 1) * X * Y, c(1, 0, 1) * X * Z, c(1, 1, 0) * Y * Z], -1)
    return v, g, H6


def _h6(H):
    H = np.asarray(H)
    return np.stack([H[:, 0, 0], H[:, 1, 1], H[:, 2, 2], H[:, 0, 1], H[:, 0, 2], H[:, 1, 2]], -1)


def _grid_and_points(n=6000, seed=5, shape=(20, 24, 28)):
    """An anisotropic grid (so that an x / y / z mix-up cannot cancel) and points that cover the interior, the clamped rim and
    the outside of the unit cube."""
    rng = np.random.default_rng(seed)
    data = rng.standard_normal(shape)
    pts = rng.random((n, 3)) * 1.2 - 0.1
    return data, pts


def test_scipy_witness_is_self_consistent():
    """The two scipy routes (ndimage's evaluator, interpolate's basis functions) agree on the value: the tap / argument bookkeeping
    of `scipy_weights` is right before it is used to judge derivatives."""
    data, pts = _grid_and_points()
    v, _, _ = scipy_cubic_all(data, pts)
    assert np.abs(v - scipy_cubic_value(data, pts)).max() < 1e-12


def test_bspline_weights_match_scipy_basis():
    """SURVEY C.1's weight polynomials (value, first and second derivative) == scipy's cardinal cubic B-spline."""
    a = torch.linspace(0, 1, 257, dtype=torch.float64)[:-1]           # (alpha = 1 is alpha = 0 of the next cell)
    w, dw, ddw = O.bspline_weights(a)
    for ours, k in ((w, 0), (dw, 1), (ddw, 2)):
        assert np.abs(ours.numpy() - scipy_weights(a.numpy(), k)).max() < 1e-13


def test_oracle_cubic_lookup_matches_scipy():
    """fp64 oracle (`Texture3f.eval_cubic / _grad / _hessian`, shapes.py:421-450) against scipy: texel-centre convention, x-fastest
    layout, clamped borders, resolution factors of gradient and Hessian."""
    data, pts = _grid_and_points()
    v, g, H = O.eval_cubic(torch.from_numpy(data), torch.from_numpy(pts), 2)
    vs, gs, Hs = scipy_cubic_all(data, pts)
    assert np.abs(v.numpy() - scipy_cubic_value(data, pts)).max() < 1e-12
    assert rel_l2(v.numpy(), vs) < 1e-13 and rel_l2(g.numpy(), gs) < 1e-13 and rel_l2(_h6(H.numpy()), Hs) < 1e-13


def test_oracle_gradient_is_the_derivative_of_scipys_value():
    """... and the gradient is the derivative of ndimage's evaluator itself (central differences in fp64), not only of our own
    tensor-product bookkeeping."""
    data, pts = _grid_and_points(n=1500)
    pts = pts[np.all((pts > 0.12) & (pts < 0.88), -1)]               # (differences across a clamped border are one-sided)
    _, g, _ = O.eval_cubic(torch.from_numpy(data), torch.from_numpy(pts), 1)
    h = 1e-5
    fd = np.stack([(scipy_cubic_value(data, pts + h * e) - scipy_cubic_value(data, pts - h * e)) / (2 * h) for e in np.eye(3)], -1)
    assert rel_l2(g.numpy(), fd) < 1e-7


def test_c_oracle_cubic_lookup_matches_scipy(built):
    import c_oracle
    data, pts = _grid_and_points(n=3000)
    vs, gs, Hs = scipy_cubic_all(data.astype(np.float32), pts.astype(np.float32))
    if not hasattr(c_oracle, 'eval_cubic'):
        pytest.skip('the C oracle exposes no per-point lookup')
    v, g, H = c_oracle.eval_cubic(c_oracle.load(), data.astype(np.float32), pts.astype(np.float32))
    assert rel_l2(v, vs) < 1e-6 and rel_l2(g, gs) < 1e-5 and rel_l2(H, Hs) < 1e-5


def test_kernel_arithmetic_cubic_lookup_matches_scipy(harness):
    """The host build of the kernels' own `eval_cubic_rows` (csrc/dsdf_math.h) against scipy, fp32 in / fp64 witness."""
    data, pts = _grid_and_points()
    d32, p32 = data.astype(np.float32), pts.astype(np.float32)
    v, g, H = harness.eval_cubic(d32, p32, 2)
    vs, gs, Hs = scipy_cubic_all(d32, p32)
    assert rel_l2(v, vs) < 1e-6 and rel_l2(g, gs) < 1e-5 and rel_l2(H, Hs) < 1e-5
    # points whose texel coordinate rounds onto a cell boundary in fp32 may pick the neighbouring cell: the spline is C2, so the
    # value is continuous there and the per-point error stays at the rounding level
    assert np.abs(v - vs).max() < 2e-5


@pytest.mark.gpu
def test_gpu_cubic_lookup_matches_scipy(built):
    """`dsdf_eval_cubic` (C-ABI) against scipy -- no oracle in between."""
    import dsdf
    dsdf.load()
    data, pts = _grid_and_points(n=20000)
    d32, p32 = data.astype(np.float32), pts.astype(np.float32)
    g = dsdf.SdfGrid(torch.from_numpy(d32).cuda())
    v, gr, H = dsdf.eval_cubic(g, torch.from_numpy(p32).cuda(), 2)
    vs, gs, Hs = scipy_cubic_all(d32, p32)
    assert rel_l2(v.cpu().numpy(), vs) < 1e-6 and rel_l2(gr.cpu().numpy(), gs) < 1e-5 and rel_l2(H.cpu().numpy(), Hs) < 1e-5
    v0, _, _ = dsdf.eval_cubic(g, torch.from_numpy(p32).cuda(), 0)
    assert rel_l2(v0.cpu().numpy(), scipy_cubic_value(d32, p32)) < 1e-6


# ------------------------------------------------------------------------------------------------ torch.optim.Adam
def test_adam_matches_torch_optim():
    """`variables.Adam` (mi.ad.Adam) against torch.optim.Adam on the same gradient sequence.  The two place epsilon
    differently -- Mitsuba: lr sqrt(1 - b2^t) / (1 - b1^t) m / (sqrt(v) + eps); torch: lr / (1 - b1^t) m / (sqrt(v / (1 - b2^t)) + eps)
    -- a relative difference of at most eps / sqrt(v) * (1 / sqrt(1 - b2^t) - 1) per step, 3e-6 for |g| >= 0.1 at t = 1."""
    import variables as V
    torch.manual_seed(7)
    x0 = torch.randn(50, dtype=torch.float64)
    ours = V.Adam(lr=0.01, params={'x': x0})
    ours.set_learning_rate({'x': 0.02})
    xt = x0.clone().requires_grad_(True)
    theirs = torch.optim.Adam([xt], lr=0.02, betas=(0.9, 0.999), eps=1e-8)
    for t in range(40):
        g = torch.randn(50, dtype=torch.float64)
        g = torch.where(g.abs() < 0.1, torch.full_like(g, 0.1), g)
        ours['x'].grad = g.clone()
        ours.step()
        xt.grad = g.clone()
        theirs.step()
        step_size = 0.02
        assert (ours['x'].detach() - xt.detach()).abs().max() < 5e-6 * step_size * (t + 1)
    assert (ours['x'].detach() - x0).abs().min() > 1e-3               # (both moved)


# ------------------------------------------------------------------------------------------------ scipy's distance transform
def _sign_field(R):
    lin = np.linspace(0, 1, R)
    z, y, x = np.meshgrid(lin, lin, lin, indexing='ij')
    sd = np.minimum(np.sqrt((x - .4) ** 2 + (y - .45) ** 2 + (z - .55) ** 2) - 0.22, np.sqrt((x - .66) ** 2 + (y - .6) ** 2 + (z - .5) ** 2) - 0.16)
    return sd, (sd * (1.5 + 0.5 * np.sin(9 * y))).astype(np.float32)     # two merged spheres; a distorted field with the same zero set


def _edt_signed(inside, R):
    """Exact Euclidean distance to the nearest voxel of the other sign, in unit-cube units (the interface sits within half a voxel
    of it)."""
    dx = 1.0 / (R - 1)
    return (ndimage.distance_transform_edt(~inside) - ndimage.distance_transform_edt(inside)) * dx


def _check_redistance(u, phi, R):
    inside = phi < 0
    assert ((u < 0) == inside).all()                                   # the sign, i.e. the zero level set to within a voxel, is kept
    edt = _edt_signed(inside, R)
    dx = 1.0 / (R - 1)
    # edt measures to the nearest voxel CENTRE of the other sign: |u| lies between edt - dx and edt, up to the scheme's first order error
    assert np.abs(np.abs(u) - (np.abs(edt) - 0.5 * dx)).max() < 1.6 * dx
    far = np.abs(edt) > 4 * dx
    assert rel_l2(u[far], (edt - np.sign(edt) * 0.5 * dx)[far]) < 0.05


def test_c_redistance_matches_scipy_distance_transform(built):
    import c_oracle
    R = 48
    _, phi = _sign_field(R)
    _check_redistance(c_oracle.redistance(c_oracle.load(), phi), phi, R)


@pytest.mark.gpu
def test_gpu_redistance_matches_scipy_distance_transform(built):
    import dsdf
    dsdf.load()
    R = 48
    _, phi = _sign_field(R)
    u = dsdf.redistance(torch.from_numpy(phi).cuda()).cpu().numpy()
    _check_redistance(u, phi, R)
