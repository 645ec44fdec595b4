"""Kernel arithmetic (csrc/dsdf_math.h, dsdf_lane.h) compiled for the host
(tests/harness, TEST-ONLY) against the oracle on identical inputs.  This is what
lets the hand-derived adjoint be validated without a GPU; the GPU parity tests
(test_gpu_parity.py) repeat the same comparisons through the C-ABI.

Tolerances: forward images 1e-4 relative L2 (north_star).  Gradients: per case
max(2 x measured fp32 floor, 1e-4) against the fp64 oracle (tests/precision.py: the
warp-field weights are 1/denom^3 with denom down to 1e-6, so ANY fp32 evaluation of
the estimator -- the C restatement built in fp32, the torch oracle run in fp32 --
sits 3e-5 ... 2e-3 from its fp64 result on these cases; test_fp32_noise_floor records it)."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from cases import make_case, oracle_backward, oracle_forward
import precision as P
from conftest import rel_l2

FWD_TOL = 1e-4


def cam_params(case):
    return case['cam'].params()


def test_eval_cubic_host(harness):
    grid = O.blob_grid(24, n=5, seed=4)
    pts = (torch.rand(4000, 3, dtype=torch.float64) * 1.1 - 0.05).float()
    v, g, H = harness.eval_cubic(grid.float().numpy(), pts.numpy(), 2)
    vo, go, Ho = O.eval_cubic(grid, pts.double(), 2)
    assert rel_l2(v, vo) < 1e-6
    assert rel_l2(g, go) < 1e-5
    Ho6 = torch.stack([Ho[:, 0, 0], Ho[:, 1, 1], Ho[:, 2, 2], Ho[:, 0, 1], Ho[:, 0, 2], Ho[:, 1, 2]], -1)
    assert rel_l2(H, Ho6) < 1e-5


def test_trace_host(harness):
    case = make_case('blob32')
    cam = case['cam']
    pos = torch.rand(3000, 2, dtype=torch.float64) * torch.tensor([case['W'], case['H']], dtype=torch.float64)
    o, d, maxt = cam.sample_ray(pos, case['W'], case['H'])
    o32, d32, m32 = o.float(), d.float(), maxt.float()
    ref = O.ray_intersect(O.Grid3d(case['grid']), o32.double(), d32.double(), m32.double())
    out = harness.trace(case['grid'].float().numpy(), o32.numpy(), d32.numpy(), m32.numpy(), diff=True)
    fin = torch.isfinite(ref['its_t']).numpy()
    assert np.array_equal(np.isfinite(out['its_t']), fin)
    assert rel_l2(out['its_t'][fin], ref['its_t'].numpy()[fin]) < 1e-5
    wf = torch.isfinite(ref['warp_t']).numpy()
    assert (np.isfinite(out['warp_t']) == wf).mean() > 0.999
    both = wf & np.isfinite(out['warp_t'])
    assert rel_l2(out['warp_t'][both], ref['warp_t'].numpy()[both]) < 1e-4
    assert rel_l2(out['warp_weight'][both], ref['warp_weight'].numpy()[both]) < 1e-3
    same = out['steps'] == ref['steps'].numpy()
    assert same.mean() > 0.99
    plain = harness.trace(case['grid'].float().numpy(), o32.numpy(), d32.numpy(), m32.numpy(), diff=False)
    assert np.array_equal(np.isfinite(plain['its_t']), np.isfinite(out['its_t']))
    assert rel_l2(plain['its_t'][fin], out['its_t'][fin]) < 1e-6


def test_sampler_host_matches_oracle(harness):
    a = harness.sampler(12345, 5000)
    b = O.independent_sampler_2d(12345, 5000)
    assert np.array_equal(a, b)


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_render_forward_host(harness, name, integ):
    case = make_case(name)
    ref, aux = oracle_forward(case, integ)
    for diff in (False, True):
        img = harness.render_forward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                                     case['offsets'].numpy(), integ, diff=diff)
        assert rel_l2(img, ref.numpy()) < FWD_TOL


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
@pytest.mark.parametrize('reparam', [True, False])
def test_render_backward_host(harness, name, integ, reparam):
    case = make_case(name)
    gref = oracle_backward(case, integ, reparam).numpy()
    gg, img = harness.render_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                                      case['offsets'].numpy(), case['grad_image'].numpy(), integ, reparam=reparam)
    if not reparam and integ == O.SILHOUETTE:
        assert np.abs(gg).max() == 0 and np.abs(gref).max() == 0      # no shading, no warp -> no gradient
        return
    assert np.isfinite(gg).all()
    ok, msg = P.check_gradient('host-harness', case, integ, reparam, gg)
    assert ok, msg
    assert rel_l2(gref, P.reference_gradient(case, integ, reparam)['g64']) < 1e-6     # torch autograd == hand-written C adjoint (fp64)


def test_fp32_noise_floor():
    """Two independent fp32 evaluations of the reference algorithm (C restatement built in fp32, torch oracle run in
    fp32) against the fp64 oracle, on bit-identical inputs: the floor every gradient gate in this suite is derived from.
    Also: rounding the sensor record to fp32 alone (a 6e-8 relative input change) moves the fp64 gradient by ~1e-4."""
    for name in ('sphere16', 'blob32', 'blob32_spp64', 'blob48_rect'):
        case = make_case(name)
        for integ in (O.SILHOUETTE, O.SIMPLE_SHADING):
            r = P.reference_gradient(case, integ, True)
            P.record('floor', case=name, integ=integ, floor_c=r['floor_c'], floor_torch=r['floor_torch'])
            assert 1e-6 < r['floor_c'] < 5e-3 and 1e-6 < r['floor_torch'] < 5e-3, (name, integ, r['floor_c'], r['floor_torch'])
    case = make_case('blob32')
    exact_cam = O.Camera(case['origin'])                                   # un-rounded fp64 sensor
    g_exact = O.render_backward(O.Grid3d(case['grid']), exact_cam, case['W'], case['H'], case['spp'], case['offsets'].double(),
                                case['grad_image'].double(), O.SILHOUETTE).numpy()
    e = rel_l2(g_exact, P.reference_gradient(case, O.SILHOUETTE, True)['g64'])
    assert 1e-5 < e < 1e-3, e                                              # conditioning: ~1e-4 from a 6e-8 input change


@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_translation_gradient_host(harness, integ):
    """dL/d(sdf.p) (python/shapes.py:389, 412, 471: `sdf.p` is a differentiable parameter; the
    reference's forward-gradient validation differentiates with respect to it) against autograd."""
    case = make_case('blob32')

    def oracle(dt):
        p = torch.zeros(3, dtype=dt, requires_grad=True)
        img = O.render(O.Grid3d(case['grid'].to(dt), p), O.Camera.from_params(cam_params(case), dtype=dt), case['W'], case['H'],
                       case['spp'], case['offsets'].to(dt), integ)
        (img * case['grad_image'].to(dt)).sum().backward()
        return (p.grad,)
    (gp_ref,), (tol,) = P.torch_gate(oracle)
    harness.render_backward(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                            case['offsets'].numpy(), case['grad_image'].numpy(), integ)
    assert rel_l2(harness.last_grad_p, gp_ref) < tol, (rel_l2(harness.last_grad_p, gp_ref), tol)


@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_forward_mode_is_transpose_of_backward(harness, integ):
    """`render_forward` (integrators/reparam.py:192-196): <J dtheta, G> == <dtheta, J^T G> for a tangent on
    sdf.data and on sdf.p, J^T G being the (oracle-checked) backward."""
    case = make_case('blob32')
    g32 = case['grid'].float().numpy()
    args = (g32, cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy())
    gen = torch.Generator().manual_seed(5)
    tdata = torch.randn(g32.shape, generator=gen).numpy().astype(np.float32)
    tp = np.array([0.3, -0.7, 0.5], np.float32)
    G = case['grad_image'].numpy()
    gg, _ = harness.render_backward(*args, G, integ)
    gp = harness.last_grad_p.copy()
    jd = harness.render_forward_grad(*args, integ, tangent=tdata)
    jp = harness.render_forward_grad(*args, integ, tangent_p=tp)
    lhs_d, rhs_d = float((jd.astype(np.float64) * G).sum()), float((gg.astype(np.float64) * tdata).sum())
    lhs_p, rhs_p = float((jp.astype(np.float64) * G).sum()), float((gp.astype(np.float64) * tp).sum())
    assert abs(lhs_d - rhs_d) < 2e-3 * max(abs(rhs_d), 1.0), (lhs_d, rhs_d)
    assert abs(lhs_p - rhs_p) < 2e-3 * max(abs(rhs_p), 1.0), (lhs_p, rhs_p)
    assert np.abs(jd).max() > 0 and np.abs(jp).max() > 0


def test_forward_mode_matches_oracle_jvp(harness):
    """Gradient image w.r.t. a translation of the SDF (the reference's eval_forward_gradient,
    figures/result_utils.py:126-161) against forward-over-reverse autograd of the oracle."""
    case = make_case('sphere16')
    tp = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)

    def oracle(dt):
        c = O.Camera.from_params(cam_params(case), dtype=dt)

        def f(p):
            return O.render(O.Grid3d(case['grid'].to(dt), p), c, case['W'], case['H'], case['spp'], case['offsets'].to(dt), O.SIMPLE_SHADING)
        return (torch.autograd.functional.jvp(f, torch.zeros(3, dtype=dt), tp.to(dt))[1],)
    (ref,), (tol,) = P.torch_gate(oracle)
    out = harness.render_forward_grad(case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'],
                                      case['offsets'].numpy(), O.SIMPLE_SHADING, tangent_p=tp.numpy())
    assert rel_l2(out, ref) < tol, (rel_l2(out, ref), tol)


def test_warp_eval_host(harness):
    """A9 per ray: cdir, a, b, div of `warp_coefficients` (the statements k_warp_eval / k_backward execute) against the
    oracle's autograd linearisation of WarpField2D.eval (python/warp.py:47-96)."""
    case = make_case('blob32')
    o32, d32, m32, tr = P.silhouette_rays(case)
    tr32 = {k: v.float() for k, v in tr.items() if k != 'steps'}
    out = harness.warp_eval(case['grid'].float().numpy(), o32.numpy(), d32.numpy(), {k: v.numpy() for k, v in tr32.items()})
    P.check_warp_coefficients('host-harness', case, o32, d32, tr32, out)


def test_reuse_fetch_is_bit_identical(harness):
    """ReuseFetch (taps of the last visited cell kept in registers) against the per-step gather, on rays that start
    on the surface like shadow rays do: identical hit distances, warp quantities and step counts, bit for bit."""
    case = make_case('blob32')
    g = case['grid'].float().numpy()
    cam = case['cam']
    torch.manual_seed(4)
    pos = torch.rand(3000, 2, dtype=torch.float64) * torch.tensor([case['W'], case['H']], dtype=torch.float64)
    o, d, maxt = cam.sample_ray(pos, case['W'], case['H'])
    prim = harness.trace(g, o.float().numpy(), d.float().numpy(), maxt.float().numpy(), diff=0)
    hit = np.isfinite(prim['its_t'])
    # secondary rays leaving the hit points in random directions, origins 1e-4 off the surface
    oh = (o.float().numpy() + prim['its_t'][:, None] * d.float().numpy())[hit]
    dirs = torch.randn(oh.shape[0], 3).numpy().astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o2 = (oh + 1e-4 * dirs).astype(np.float32)
    m2 = np.full(o2.shape[0], 3.9, np.float32)
    for base_mode, reuse_mode in ((0, 2), (1, 3)):
        a = harness.trace(g, o2, dirs, m2, diff=base_mode)
        b = harness.trace(g, o2, dirs, m2, diff=reuse_mode)
        for k in a:
            assert np.array_equal(a[k], b[k], equal_nan=True), (base_mode, k)
    assert hit.sum() > 50 and prim["steps"].max() > 10


def test_resumable_diff_march_is_bit_identical(harness):
    """DiffMarch (begin / step / finish), the form a tail wave resumes, against the closed loop of trace_diff."""
    case = make_case('blob48_rect')
    cam = case['cam']
    torch.manual_seed(2)
    pos = torch.rand(6000, 2, dtype=torch.float64) * torch.tensor([case['W'], case['H']], dtype=torch.float64)
    o, d, maxt = cam.sample_ray(pos, case['W'], case['H'])
    g = case['grid'].float().numpy()
    a = harness.trace(g, o.float().numpy(), d.float().numpy(), maxt.float().numpy(), diff=1)
    b = harness.trace(g, o.float().numpy(), d.float().numpy(), maxt.float().numpy(), diff=4)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize('split', [1, 3, 8, 20])
def test_handed_off_march_resumes_bit_identically(harness, split):
    """Tail hand-off: a differentiable march stopped after `split` steps, exported in the tail-queue layout and
    resumed from a freshly begun march equals the uninterrupted one bit for bit (no piece of state is lost)."""
    case = make_case('blob48_rect')
    cam = case['cam']
    torch.manual_seed(6)
    pos = torch.rand(5000, 2, dtype=torch.float64) * torch.tensor([case['W'], case['H']], dtype=torch.float64)
    o, d, maxt = cam.sample_ray(pos, case['W'], case['H'])
    g = case['grid'].float().numpy()
    a = harness.trace(g, o.float().numpy(), d.float().numpy(), maxt.float().numpy(), diff=1)
    b = harness.trace_resumed(g, o.float().numpy(), d.float().numpy(), maxt.float().numpy(), split)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert (a['steps'] > split).sum() > 100


@pytest.mark.parametrize('name', ['sphere16', 'blob32', 'blob48_rect'])
@pytest.mark.parametrize('integ', [O.SILHOUETTE, O.SIMPLE_SHADING])
def test_split_adjoint_equals_fused(harness, name, integ):
    """lane_backward_coef (image-independent half, computed beside the primal pass) + lane_backward_apply (film adjoint
    gather + scatter requests) is the fused lane_backward with the film scalars factored out."""
    case = make_case(name)
    a = (case['grid'].float().numpy(), cam_params(case), case['W'], case['H'], case['spp'], case['offsets'].numpy(), case['grad_image'].numpy(), integ)
    fused, _ = harness.render_backward(*a)
    split, _ = harness.render_backward(*a, split=True)
    assert np.abs(fused).max() > 0
    assert rel_l2(split, fused) < 2e-6, rel_l2(split, fused)
