"""`antithetic_sampling` (/root/reference/python/integrators/reparam.py:19, 167-178): every lane is evaluated a second time at the
mirrored film position `pos - r + 1` with a clone of its sampler, both samples into one film block.

Fixtures: tags `sil_anti` / `direct_anti` of tests/golden/refshim_<case>.npz -- the reference's own render / render_backward with the
property set, on the torch stand-in (tools/make_reference_fixtures.py --shim).
  * the fp64 oracle (render(..., antithetic=True)) reproduces image, dL/d data, dL/d p, dL/d albedo to 1e-9;
  * the host build of the kernel arithmetic with the pair (1 - r as a second offsets set) meets the fp32 gates of the plain runs;
  * the product renders the pair through the film-level entry points (two passes over one film, the second with the offsets of
    dsdf_sampler_2d(mirror) NEXT TO the view's seeds): integrator mirror against the fixture on the GPU."""
import numpy as np
import pytest
import torch

import sdf_oracle as O
import precision as P
from conftest import rel_l2
from test_refshim_fixture import TAGS, _gates, check_fp32_gradient, inputs, load, oracle_run

TAGS.setdefault('sil_anti', (O.SILHOUETTE, dict(antithetic=True)))
TAGS.setdefault('direct_anti', (O.DIRECT, dict(antithetic=True)))


@pytest.mark.parametrize('name,tag', [('sphere16', 'sil_anti'), ('sphere16', 'direct_anti'), ('blob32', 'sil_anti'), ('blob32', 'direct_anti')])
def test_oracle_matches_reference_code_antithetic(name, tag):
    ref = load(name)
    out = oracle_run(inputs(ref), tag)
    assert rel_l2(out[0].numpy(), ref[f'img_{tag}']) < 1e-12
    assert np.abs(ref[f'img_{tag}'] - ref[f'img_{tag[:-5]}']).max() > 1e-3                 # (a different estimate, not the plain one)
    assert rel_l2(out[1].numpy(), ref[f'grad_{tag}']) < 1e-9 and rel_l2(out[2].numpy(), ref[f'gradp_{tag}']) < 1e-9
    if len(out) > 3:
        assert rel_l2(out[3].numpy(), ref[f'galb_{tag}']) < 1e-9


@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
def test_kernel_math_matches_reference_code_antithetic(harness, name):
    ref = load(name)
    x = inputs(ref)
    gates = _gates(x, 'sil_anti', ref)
    gates[1] = max(gates[1], P.grad_tol(dict(name='refshim_' + name, grid=x['grid'], cam=x['cam'], W=x['W'], H=x['H'], spp=x['spp'],
                                             offsets=x['offs'].float(), grad_image=x['gi'].float()), O.SILHOUETTE))
    r = ref['sampler_2d']
    gg, img = harness.render_backward(ref['grid'], ref['cam16'], x['W'], x['H'], x['spp'], r, ref['grad_image'], O.SILHOUETTE,
                                      offsets2=(np.float32(1.0) - r))
    assert rel_l2(img, ref['img_sil_anti']) < 1e-4
    check_fp32_gradient('refshim_host', name, 'sil_anti', gg, ref['grad_sil_anti'], gates[1])
    assert rel_l2(harness.last_grad_p, ref['gradp_sil_anti']) < max(gates[2], 2 * gates[1])


def test_mirror_surface():
    import integrators  # noqa: F401
    from integrators.reparam import create_integrator
    for n in ('sdf_silhouette_reparam', 'sdf_simple_shading_reparam', 'sdf_direct_reparam'):
        assert create_integrator(n, {}).antithetic_sampling is False
        assert create_integrator(n, {'antithetic_sampling': True}).antithetic_sampling is True


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
def test_gpu_antithetic_matches_reference_code(built, name):
    """integrator.render / render_backward / the render op with `antithetic_sampling` against the reference's own code."""
    import configs
    import dsdf
    import shapes
    import integrators  # noqa: F401
    from integrators.reparam import Scene, create_integrator, traverse, render as mi_render
    from constants import SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P
    dsdf.load()
    ref = load(name)
    x = inputs(ref)
    W, H, spp, seed = x['W'], x['H'], x['spp'], x['seed']
    sensor = dsdf.Sensor(ref['origin'], resx=W, resy=H)
    # dsdf_sampler_2d: the built-in sampler's film offsets, and their mirror images
    r = dsdf.sampler_offsets([sensor], spp, [seed]).cpu().numpy()[0]
    assert np.array_equal(r, ref['sampler_2d'])
    assert np.array_equal(dsdf.sampler_offsets([sensor], spp, [seed], mirror=True).cpu().numpy()[0], np.float32(1.0) - ref['sampler_2d'])
    gi = torch.from_numpy(ref['grad_image']).cuda()
    for integ_name, tag, integ in (('sdf_silhouette_reparam', 'sil_anti', O.SILHOUETTE), ('sdf_direct_reparam', 'direct_anti', O.DIRECT)):
        props = {'sdf': shapes.Grid3d(torch.from_numpy(ref['grid']).cuda()), 'antithetic_sampling': True}
        if integ == O.DIRECT:
            props.update(reflectance=torch.from_numpy(ref['albedo']).cuda(), env_radiance=tuple(float(e) for e in ref['env']))
        it = create_integrator(integ_name, props)
        scene = Scene([sensor], it)
        it.warp_field = configs.get_config('warp').get_warpfield(it.sdf)
        img = it.render(scene, 0, seed=seed, spp=spp).cpu().numpy()
        assert rel_l2(img, ref[f'img_{tag}']) < 1e-4, tag
        gates = _gates(x, tag, ref)
        if integ != O.DIRECT:
            gates[1] = max(gates[1], P.grad_tol(dict(name='refshim_' + name, grid=x['grid'], cam=x['cam'], W=W, H=H, spp=spp,
                                                     offsets=x['offs'].float(), grad_image=x['gi'].float()), integ))
        params = traverse(scene)
        leaf = torch.from_numpy(ref['grid']).cuda()[..., None].clone().requires_grad_(True)
        pl = torch.zeros(3).requires_grad_(True)
        params[SDF_DEFAULT_KEY], params[SDF_DEFAULT_KEY_P] = leaf, pl
        akey = [k for k in params if k.endswith('reflectance.volume.data')]
        if akey:
            params[akey[0]] = torch.from_numpy(ref['albedo']).cuda().clone().requires_grad_(True)
        params.update()
        it.render_backward(scene, params, gi, 0, seed=seed, spp=spp)
        e = check_fp32_gradient('refshim_gpu', name, tag, leaf.grad.reshape(ref['grid'].shape).cpu().numpy(), ref[f'grad_{tag}'], gates[1])
        assert rel_l2(pl.grad.cpu().numpy(), ref[f'gradp_{tag}']) < max(gates[2], 2 * gates[1], 2 * e)
        if akey:
            ea = rel_l2(params[akey[0]].grad.cpu().numpy(), ref[f'galb_{tag}'])
            assert ea < gates[3], (ea, gates[3])
        # the render op (mi.render with attached parameters): same image, same gradient up to the order of the float atomics
        att = traverse(scene)
        leaf2 = torch.from_numpy(ref['grid']).cuda()[..., None].clone().requires_grad_(True)
        att[SDF_DEFAULT_KEY] = leaf2
        att.update()
        img2 = mi_render(scene, att, sensor=[sensor], seed=seed, spp=spp, seed_grad=seed, spp_grad=spp)
        assert rel_l2(img2[0].detach().cpu().numpy(), img) < 1e-6
        (img2[0] * gi).sum().backward()
        assert rel_l2(leaf2.grad.cpu().numpy(), leaf.grad.cpu().numpy()) < 1e-4
