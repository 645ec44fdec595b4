"""The debug images of `use_aovs` + `WarpField2D.return_aovs` (/root/reference/python/integrators/reparam.py:130, 160-165, 263-267;
warp.py:17, 105-106; shapes.py:240-242): the film carries eleven channels behind RGB, of which the reference's code fills two -- the
loop state of the primary ray's differentiable trace, `i` and `weight_sum`.

The fixtures `aov_<tag>` of tests/golden/refshim_<case>.npz are the (H, W, 14) images the REFERENCE'S OWN files produced on the torch
stand-in (tools/make_reference_fixtures.py --shim, tags sil_aovs / direct_aovs / sil_aovs_noreparam).
  * the fp64 oracle reproduces them to rounding (gate 1e-9);
  * the kernel arithmetic (host build of k_render_aovs' statements) and the HIP path (dsdf_render_aovs through the integrator
    mirror) match them as far as fp32 can (check_fp32: half of the pixels to 1e-5, nine in ten to 5e-3, the image to 2e-2 -- the
    statistic of a handful of grazing samples; the oracle's own fp32 run is printed beside every figure)."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
from conftest import rel_l2
from test_refshim_fixture import inputs, load

I, WS = 3 + O.AOV_NAMES.index('i'), 3 + O.AOV_NAMES.index('weight_sum')
OTHER = [3 + k for k, n in enumerate(O.AOV_NAMES) if n not in ('i', 'weight_sum')]
CASES = ['sphere16', 'blob32']


def oracle_aovs(x, integ, reparam, dtype=torch.float64, **kw):
    cam = O.Camera.from_params(x['cam'].params(), dtype=dtype)
    if integ == O.DIRECT:
        kw.update(albedo=x['albedo'].to(dtype), emitter_u=x['emitter_u'].to(dtype), env=x['env'].to(dtype))
    return O.render(O.Grid3d(x['grid'].to(dtype)), cam, x['W'], x['H'], x['spp'], x['offs'].to(dtype), integ, reparam, aovs=True, **kw).numpy()


def fp32_floor(x, ref):
    """rel-L2 of the oracle's own fp32 run against the fp64 fixture, per filled channel."""
    a = oracle_aovs(x, O.SILHOUETTE, True, torch.float32)
    r = ref['aov_sil_aovs']
    return rel_l2(a[..., I], r[..., I]), rel_l2(a[..., WS], r[..., WS])


def check_fp32(two, ref, x, what):
    """fp32 (host build / HIP) against the fp64 fixture.  Both channels are heavy-tailed: `weight_sum` of a sample is a sum of
    1 / denom^3 weights (shapes.py:74-75) that reaches 1e10 for a grazing ray, where denom ~ 5e-4 turns the 1e-7 absolute rounding
    of the interpolated SDF value into 1e-3 ... 1e-2 relative; `i` moves by one when a march takes a step more or less.  The
    rel-L2 of an image is therefore carried by a handful of samples (measured: the ten worst of the 256 / 576 pixels hold 96-99.9 % of
    the squared error, and two fp32 codes -- this one and the oracle run in fp32 -- differ from each other as much as from fp64), so it
    gets the bound of such a sample, 2e-2, and the bulk of the pixels is gated separately and tightly."""
    r = ref['aov_sil_aovs']
    fi, fw = fp32_floor(x, ref)
    ei, ew = rel_l2(two[..., 0], r[..., I]), rel_l2(two[..., 1], r[..., WS])
    rel = lambda a, b: (np.abs(a - b) / np.maximum(np.abs(b), 1e-30))[b > 0]
    pi, pw = np.percentile(rel(two[..., 0], r[..., I]), [50, 90]), np.percentile(rel(two[..., 1], r[..., WS]), [50, 90])
    print(f"{what}: i {ei:.3e} (fp32 oracle {fi:.3e}; p50 {pi[0]:.1e}, p90 {pi[1]:.1e}), weight_sum {ew:.3e} (fp32 oracle {fw:.3e}; "
          f"p50 {pw[0]:.1e}, p90 {pw[1]:.1e})")
    assert ei < 2e-2 and ew < 2e-2, (ei, ew)
    assert pi[0] < 1e-5 and pi[1] < 5e-3, pi
    assert pw[0] < 1e-5 and pw[1] < 5e-3, pw


@pytest.mark.parametrize('name', CASES)
def test_oracle_aovs_match_reference_code(name):
    ref = load(name)
    x = inputs(ref)
    a = oracle_aovs(x, O.SILHOUETTE, True)
    r = ref['aov_sil_aovs']
    assert r.shape == (x['H'], x['W'], 3 + len(O.AOV_NAMES)) and r[..., I].max() > 1 and r[..., WS].max() > 1
    assert np.abs(r[..., OTHER]).max() == 0                      # nothing in the reference ever writes the other nine
    assert rel_l2(a[..., :3], r[..., :3]) < 1e-12 and rel_l2(a[..., I], r[..., I]) < 1e-12 and rel_l2(a[..., WS], r[..., WS]) < 1e-9
    assert np.abs(a[..., OTHER]).max() == 0
    # sdf_direct_reparam: the same two channels of the same primary ray (sdf_direct_reparam.py:58-60 copies a key that the shadow
    # ray's dictionary never holds); RGB is the primal image
    d, rd = oracle_aovs(x, O.DIRECT, True), ref['aov_direct_aovs']
    assert rel_l2(d, rd) < 1e-9 and np.array_equal(rd[..., 3:], r[..., 3:])
    # no reparameterisation (DummyWarpField, warp.py:179-196): the eleven channels exist and stay zero
    n, rn = oracle_aovs(x, O.SILHOUETTE, False), ref['aov_sil_aovs_noreparam']
    assert np.abs(rn[..., 3:]).max() == 0 and rel_l2(n, rn) < 1e-12


@pytest.mark.parametrize('name', CASES)
def test_kernel_math_aovs_match_reference_code(harness, name):
    ref = load(name)
    x = inputs(ref)
    two = harness.render_aovs(ref['grid'], ref['cam16'], x['W'], x['H'], x['spp'], ref['sampler_2d'])
    check_fp32(two, ref, x, f'host {name}')
    # the built-in sampler seeded like ReparamIntegrator.prepare draws the same samples
    assert np.array_equal(two, harness.render_aovs(ref['grid'], ref['cam16'], x['W'], x['H'], x['spp'], None, seed=x['seed']))


def test_mirror_surface():
    """use_aovs / return_aovs / aov_names on the integrator mirror (no device needed)."""
    import dsdf
    import integrators  # noqa: F401
    from integrators.reparam import create_integrator
    import warp
    assert dsdf.AOV_NAMES == O.AOV_NAMES
    for name in ('sdf_silhouette_reparam', 'sdf_simple_shading_reparam', 'sdf_direct_reparam'):
        assert create_integrator(name, {}).aov_names() == []
        it = create_integrator(name, {'use_aovs': True})
        assert it.use_aovs and it.aov_names() == O.AOV_NAMES
    wf = warp.WarpField2D(None)
    wf.return_aovs = True
    assert wf.apply(dsdf.default_params()).weight_strategy == wf.weight_strategy           # (used to raise)


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_gpu_aovs_match_reference_code(built, name):
    """integrator.render with `use_aovs` and `warp_field.return_aovs` through the mirror and dsdf_render_aovs: (H, W, 14)."""
    import configs
    import dsdf
    import shapes
    import integrators  # noqa: F401
    from integrators.reparam import Scene, create_integrator
    dsdf.load()
    ref = load(name)
    x = inputs(ref)
    sensor = dsdf.Sensor(ref['origin'], resx=x['W'], resy=x['H'])
    it = create_integrator('sdf_silhouette_reparam', {'sdf': shapes.Grid3d(torch.from_numpy(ref['grid']).cuda()), 'use_aovs': True})
    scene = Scene([sensor], it)
    it.warp_field = configs.get_config('warp').get_warpfield(it.sdf)
    img = it.render(scene, 0, seed=x['seed'], spp=x['spp']).cpu().numpy()
    assert img.shape == (x['H'], x['W'], 14) and np.abs(img[..., 3:]).max() == 0          # return_aovs is off: zeros, like the reference
    it.warp_field.return_aovs = True
    img = it.render(scene, 0, seed=x['seed'], spp=x['spp']).cpu().numpy()
    r = ref['aov_sil_aovs']
    assert img.shape == r.shape and rel_l2(img[..., :3], r[..., :3]) < 1e-4 and np.abs(img[..., OTHER]).max() == 0
    check_fp32(img[..., [I, WS]], ref, x, f'gpu {name}')
    # explicit offsets through the C-ABI give the same two channels as the built-in sampler; two views in one call
    g = it.sdf.grid
    offs = torch.from_numpy(ref['sampler_2d']).cuda()
    a = dsdf.render_aovs(g, [sensor, sensor], x['spp'], offsets=torch.stack([offs, offs]).contiguous()).cpu().numpy()
    assert rel_l2(a[0], a[1]) < 1e-6                              # (float atomics are unordered: equal up to summation order)
    check_fp32(a[0][..., [I - 3, WS - 3]], ref, x, f'gpu offsets {name}')
    # DummyWarpField (method config onlyshadinggrad): zeros
    it.warp_field = configs.get_config('onlyshadinggrad').get_warpfield(it.sdf)
    it.warp_field.return_aovs = True
    assert np.abs(it.render(scene, 0, seed=x['seed'], spp=x['spp']).cpu().numpy()[..., 3:]).max() == 0
