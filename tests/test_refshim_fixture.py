"""Fixtures made by the REFERENCE'S OWN python/ files (tests/golden/refshim_<case>.npz).

tools/make_reference_fixtures.py --shim imports python/shapes.py, warp.py, math_util.py, configs.py and integrators/*.py from the
reference checkout and runs them -- unchanged -- on tools/refshim/, a torch stand-in for the Dr.Jit / Mitsuba 3 subset they use.
The files pin the reference's first-party logic (sphere tracing and its silhouette weights, the warp field and its divergence,
surface interactions, eval_sample / render / render_backward, the three integrators' sample(), the method configs `warp`,
`warpprimary`, `warpnotnormalized`, `onlyshadinggrad`); the third-party layer underneath is restated by the stand-in, like the
oracle restates it (tools/refshim/_core.py).  The stand-in ran in fp64, so the comparison with the fp64 oracle is sharp.

  * test_oracle_matches_*            the oracle against the fixture: 1e-9 where the fp32-floor gates elsewhere are 1e-4
  * test_kernel_math_matches_*       the kernel arithmetic (host build, fp32) against it, gates = max(2 x fp32 floor, 1e-4)
  * test_gpu_matches_*               the HIP path through the C-ABI against it, same gates
  * test_fixture_is_what_the_reference_code_produces   (only where the reference checkout exists: not on the GPU box) re-runs the
                                     generator for the small case and compares with the committed file"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import sdf_oracle as O
import precision as P
from conftest import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('DSDF_REFERENCE', '/root/reference')

# tag -> (integrator, oracle keyword arguments); the generator's run list (tools/make_reference_fixtures.py)
TAGS = {
    'sil': (O.SILHOUETTE, {}), 'shade': (O.SIMPLE_SHADING, {}),
    'sil_notnorm': (O.SILHOUETTE, dict(normalize_warp_field=False)), 'shade_notnorm': (O.SIMPLE_SHADING, dict(normalize_warp_field=False)),
    'direct': (O.DIRECT, {}), 'direct_mis': (O.DIRECT, dict(use_mis=True)), 'direct_hide': (O.DIRECT, dict(hide_emitters=True)),
    'direct_detach': (O.DIRECT, dict(detach_indirect_si=True)), 'direct_decouple': (O.DIRECT, dict(decouple_reparam=True)),
    'direct_mis_decouple': (O.DIRECT, dict(use_mis=True, decouple_reparam=True)),
    'direct_primary': (O.DIRECT, dict(max_reparam_depth=0)), 'direct_mis_primary': (O.DIRECT, dict(use_mis=True, max_reparam_depth=0)),
    'direct_notnorm': (O.DIRECT, dict(normalize_warp_field=False)), 'direct_onlyshading': (O.DIRECT, dict(reparam=False)),
}
SMALL = ['sil', 'shade', 'sil_notnorm', 'direct', 'direct_mis', 'direct_primary', 'direct_notnorm', 'direct_decouple']   # on blob32 (time)


def _with_grid(z):
    """A fixture whose grid is stored by name + hash (c2_spp4: bench.py's seeded 128^3 scene) as a dict with the grid attached."""
    if 'grid' in z.files:
        return z
    import hashlib
    d = {k: z[k] for k in z.files}
    assert bytes(d['grid_recipe']) == b'bench.synth_grid(128)'
    # (the recipe's fp32 torch ops are not guaranteed bit-identical on another CPU, and this estimator amplifies a 1-ulp input change a
    # thousandfold: the array the generator saw is committed once, tests/golden/bench_grid128.npz, for both precisions)
    d['grid'] = np.load(os.path.join(GOLD, 'bench_grid128.npz'))['grid']
    assert hashlib.sha256(np.ascontiguousarray(d['grid']).tobytes()).hexdigest().encode() == bytes(d['grid_sha256']), 'bench_grid128.npz is not the grid this fixture was made from'
    class _D(dict):
        files = property(lambda self: list(self.keys()))
    return _D(d)


def load(name):
    return _with_grid(np.load(os.path.join(GOLD, f'refshim_{name}.npz')))


def inputs(ref):
    W, H, spp, seed = int(ref['W']), int(ref['H']), int(ref['spp']), int(ref['seed'])
    n = (W + 4) * (H + 4) * spp
    return dict(grid=torch.from_numpy(ref['grid']).double(), cam=O.Camera.from_params(ref['cam16']), W=W, H=H, spp=spp, seed=seed, n=n,
                offs=torch.from_numpy(ref['sampler_2d']).double(), gi=torch.from_numpy(ref['grad_image']).double(),
                albedo=torch.from_numpy(ref['albedo']).double(), env=torch.from_numpy(ref['env']).double(),
                emitter_u=torch.tensor(O.independent_sampler_emitter_2d(seed, n)).double(),
                bsdf_u=torch.tensor(O.independent_sampler_bsdf_2d(seed, n)).double())


def oracle_run(x, tag, dtype=torch.float64):
    """(image, dL/d data, dL/d p[, dL/d albedo]) of the oracle for a fixture tag."""
    integ, kw = TAGS[tag]
    kw = dict(kw)
    reparam = kw.pop('reparam', True)
    data = x['grid'].to(dtype).clone().requires_grad_(True)
    p = torch.zeros(3, dtype=dtype, requires_grad=True)
    alb = None
    if integ == O.DIRECT:
        alb = x['albedo'].to(dtype).clone().requires_grad_(True)
        kw.update(albedo=alb, emitter_u=x['emitter_u'].to(dtype), env=x['env'].to(dtype))
        if kw.get('use_mis'):
            kw['bsdf_u'] = x['bsdf_u'].to(dtype)
    cam = O.Camera.from_params(x['cam'].params(), dtype=dtype)
    img = O.render(O.Grid3d(data, p), cam, x['W'], x['H'], x['spp'], x['offs'].to(dtype), integ, reparam, **kw)
    (img * x['gi'].to(dtype)).sum().backward()
    z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    return (img.detach(), z(data), z(p)) + ((z(alb),) if alb is not None else ())


# ---------------------------------------------------------------------------------------------------------------- oracle (fp64)
@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
def test_oracle_matches_reference_code_per_ray(name):
    """Sampler stream, sensor rays, Grid3d.eval_all, SDFBase.ray_intersect (all five outputs), ray_intersect_non_diff,
    compute_surface_interaction, WarpField2D.eval (direction, divergence and its linearisation in v and g)."""
    ref = load(name)
    x = inputs(ref)
    assert np.array_equal(O.independent_sampler_2d(x['seed'], x['n']), ref['sampler_2d'])
    sdf = O.Grid3d(x['grid'])
    v, _, g, _, Hm = sdf.eval_all(torch.from_numpy(ref['eval_pts']).double())
    assert rel_l2(v.numpy(), ref['eval_v']) < 1e-12 and rel_l2(g.numpy(), ref['eval_g']) < 1e-12 and rel_l2(Hm.numpy(), ref['eval_H']) < 1e-12
    pos = torch.from_numpy(ref['ray_pos']).double() * torch.tensor([x['W'], x['H']], dtype=torch.float64)
    o, d, maxt = x['cam'].sample_ray(pos, x['W'], x['H'])                    # the oracle's closed-form sensor against the stand-in's
    assert rel_l2(o.numpy(), ref['ray_o']) < 1e-13 and rel_l2(d.numpy(), ref['ray_d']) < 1e-13 and rel_l2(maxt.numpy(), ref['ray_maxt']) < 1e-13
    o, d, maxt = (torch.from_numpy(ref[k]).double() for k in ('ray_o', 'ray_d', 'ray_maxt'))
    tr = O.ray_intersect(sdf, o, d, maxt)
    hit, fin = np.isfinite(ref['ri_its_t']), np.isfinite(ref['ri_warp_t'])
    assert hit.sum() > 10 and fin.sum() > 200
    assert np.array_equal(np.isfinite(tr['its_t'].numpy()), hit) and np.array_equal(np.isfinite(tr['warp_t'].numpy()), fin)
    assert rel_l2(tr['its_t'].numpy()[hit], ref['ri_its_t'][hit]) < 1e-12
    assert rel_l2(O.ray_intersect_non_diff(sdf, o, d, maxt)['its_t'].numpy()[hit], ref['ri_plain_its_t'][hit]) < 1e-12
    for k, tol in (('warp_t', 1e-12), ('warp_weight', 1e-11), ('warp_t_d', 1e-9), ('warp_weight_d', 1e-11)):
        assert rel_l2(tr[k].numpy()[fin], ref['ri_' + k][fin]) < tol, k
    # compute_surface_interaction (shapes.py:347-366)
    _, p, n = O.compute_surface_interaction(sdf, o[hit], d[hit], tr['its_t'][hit], None)
    assert rel_l2(p.numpy(), ref['si_p'][hit]) < 1e-12 and rel_l2(n.numpy(), ref['si_n'][hit]) < 1e-10
    # WarpField2D.eval: value of the divergence and the coefficients of its linearisation (what dsdf_warp_eval reports)
    # (the generator hands eval the NORMALISED direction, like shapes.py:124 does inside the trace; the stand-in's sensor inherits
    #  the 1e-7 non-orthonormality of the fp32 sensor record, a real Mitsuba ray is unit length already)
    tr32 = {k: tr[k] for k in ('warp_t', 'warp_t_d', 'warp_weight', 'warp_weight_d')}
    oc = P.oracle_warp_coefficients(dict(grid=x['grid']), o, d / torch.linalg.norm(d, dim=-1, keepdim=True), tr32)
    act = oc['active']
    assert act.sum() >= 20 and (act != (ref['we_div'] != 0)).mean() < 0.01       # (92 / 24 of the 512 fixture rays carry a warp)
    assert rel_l2(oc['div'][act], ref['we_div'][act]) < 1e-9
    assert rel_l2(oc['a'][act], ref['we_a'][act]) < 1e-9 and rel_l2(oc['b'][act], ref['we_b'][act]) < 1e-9
    assert rel_l2(oc['cdir'][act], ref['we_cdir'][act]) < 1e-9


@pytest.mark.parametrize('name,tag', [('sphere16', t) for t in TAGS] + [('blob32', t) for t in SMALL])
def test_oracle_matches_reference_code_render(name, tag):
    """integrator.render and render_backward of the reference (integrators/reparam.py:120-190 with the tag's integrator, properties
    and method config) against the oracle: image, dL/d(sdf.data), dL/d(sdf.p), dL/d(reflectance volume)."""
    ref = load(name)
    out = oracle_run(inputs(ref), tag)
    assert rel_l2(out[0].numpy(), ref[f'img_{tag}']) < 1e-12
    gref = ref[f'grad_{tag}']
    assert np.isfinite(gref).all() and np.abs(gref).max() > 0
    assert rel_l2(out[1].numpy(), gref) < 1e-9
    assert rel_l2(out[2].numpy(), ref[f'gradp_{tag}']) < 1e-9
    if len(out) > 3:
        assert rel_l2(out[3].numpy(), ref[f'galb_{tag}']) < 1e-9


def test_method_configs_are_different_estimators():
    """The fixture is not trivially insensitive: each setting moves dL/d(sdf.data) by more than any tolerance above."""
    ref = load('blob32')
    base = ref['grad_direct']
    for tag in ('direct_primary', 'direct_notnorm', 'direct_decouple', 'direct_mis'):
        assert rel_l2(ref[f'grad_{tag}'], base) > 1e-3, tag
    assert rel_l2(ref['grad_sil_notnorm'], ref['grad_sil']) > 1e-3
    ref16 = load('sphere16')
    assert rel_l2(ref16['img_direct_detach'], ref16['img_direct']) < 1e-12             # the variants change gradients, never the image
    assert rel_l2(ref16['grad_direct_onlyshading'], ref16['grad_direct']) > 1e-2


# ---------------------------------------------------------------------------------------------------------------- C oracle (fp64 build)
@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
def test_c64_oracle_matches_reference_code(name):
    """The plain-C restatement with the HAND-WRITTEN adjoint (oracle/dsdf_oracle.c: the checker at BASELINE.json config sizes, where
    the torch oracle is too slow) DIRECTLY against what the reference's own files produced -- not only through the torch oracle it
    is otherwise compared with (tests/test_c_oracle.py).
      * the fp64 build with fp64-valued literals (libdsdf_oracle64x.so: the same source, `f` suffixes removed) is the program the
        fixtures were made with: images to 1e-14, gradients to 1e-8 -- a code that shares no line with the reference's or the torch
        oracle's, and derives its backward by hand;
      * the regular fp64 build keeps the fp32 VALUES of its constants (0.05f: "the fp32 program in exact arithmetic", the yardstick
        of the kernels' rounding).  That 1.5e-8 relative change of a few constants moves ONE heavy-tailed sample of the blob32
        `direct` case by 5e-4 and the whole gradient by 1.4e-5 -- the conditioning of the estimator, measured
        here between two fp64 programs; every other output stays below 1e-6."""
    import c_oracle
    ref = load(name)
    x = inputs(ref)
    grid, cam, W, H, spp, offs, gi = ref['grid'], ref['cam16'], x['W'], x['H'], x['spp'], ref['sampler_2d'], ref['grad_image']
    eu, bu = x['emitter_u'].numpy(), x['bsdf_u'].numpy()
    direct = [(t, kw) for t, kw in (('direct', {}), ('direct_mis', dict(bsdf_u=bu)), ('direct_hide', dict(hide_emitters=True)),
                                    ('direct_detach', dict(variant=1)), ('direct_decouple', dict(variant=2)),
                                    ('direct_mis_decouple', dict(bsdf_u=bu, variant=2)), ('direct_onlyshading', dict(reparam=False)))
              if f'img_{t}' in ref.files]
    worst = {}
    for kind, gate_img, gate_g in (('exact', 1e-14, 1e-8), (True, 2e-7, 1e-4)):
        lib = c_oracle.load(kind)
        for tag, integ in (('sil', O.SILHOUETTE), ('shade', O.SIMPLE_SHADING)):
            gg, img = c_oracle.render_backward(lib, grid, cam, W, H, spp, offs, gi, integ, True)
            e = (rel_l2(img, ref[f'img_{tag}']), rel_l2(gg, ref[f'grad_{tag}']))
            assert e[0] < gate_img and e[1] < gate_g, (kind, tag, e)
            worst[kind] = max(worst.get(kind, 0.0), e[1])
        for tag, kw in direct:
            gg, galb, img = c_oracle.render_direct_backward(lib, grid, cam, W, H, spp, offs, eu, ref['albedo'], gi, tuple(ref['env']), **kw)
            e = (rel_l2(img, ref[f'img_{tag}']), rel_l2(gg, ref[f'grad_{tag}']), rel_l2(galb, ref[f'galb_{tag}']))
            assert e[0] < gate_img and e[1] < gate_g and e[2] < max(gate_g, 1e-6), (kind, tag, e)
            worst[kind] = max(worst.get(kind, 0.0), e[1])
    print(f"C oracle vs reference-code fixture {name}: worst gradient rel-L2 {worst['exact']:.2e} (fp64 literals), {worst[True]:.2e} (fp32-valued literals)")


# ---------------------------------------------------------------------------------------------------------------- kernel arithmetic (host build)
def _settings(tag):
    kw = TAGS[tag][1]
    s = {}
    if kw.get('normalize_warp_field') is False:
        s['normalize_warp_field'] = 0
    if kw.get('max_reparam_depth') == 0:
        s['max_reparam_depth'] = 0
    return s


def _gates(x, tag, ref):
    """max(2 x |oracle fp32 - fixture|, 1e-4) per output: the fp32 floor of this very configuration."""
    o32 = oracle_run(x, tag, torch.float32)
    keys = [f'img_{tag}', f'grad_{tag}', f'gradp_{tag}', f'galb_{tag}']
    return [max(P.FLOOR_FACTOR * rel_l2(a.numpy().astype(np.float64), ref[k]), P.NORTH_STAR) for a, k in zip(o32, keys)]


def check_fp32_gradient(kind, name, tag, gg, gref, gate):
    """Plain rel-L2 against the gate.  The sampler-seeded cases contain single samples whose 1 / denom^3 weight puts a third of
    |g|^2 into one 4^3 footprint: their fp32 error is one draw of a heavy-tailed variable, which
    the "2 x floor" rule (two other draws) does not bound.  If the plain statistic fails, ONE sample footprint (a 7^3 cube around
    the worst voxel: every 4^3 footprint containing it) may be set aside -- the rest must pass the gate and the footprint itself
    must still be reproduced to 0.5 %."""
    e = rel_l2(gg, gref)
    blocks = 0
    if e > gate:
        rest, centres, keep = P.greedy_blocks(gg, gref, gate, max_blocks=1)
        inside = rel_l2(np.asarray(gg)[~keep], np.asarray(gref)[~keep])
        blocks = len(centres)
        assert rest <= gate and inside < 5e-3, (kind, name, tag, e, rest, inside, gate, centres)
    P.record(kind, case=name, tag=tag, err=e, gate=gate, footprints_set_aside=blocks)
    return e


@pytest.mark.parametrize('name,tag', [('sphere16', 'sil'), ('sphere16', 'shade'), ('blob32', 'sil'), ('blob32', 'shade'), ('blob32', 'sil_notnorm'),
                                      ('blob32', 'direct'), ('blob32', 'direct_mis'), ('blob32', 'direct_primary'), ('blob32', 'direct_notnorm'),
                                      ('blob32', 'direct_decouple')])
def test_kernel_math_matches_reference_code(harness, name, tag):
    ref = load(name)
    x = inputs(ref)
    integ, kw = TAGS[tag]
    gates = _gates(x, tag, ref)
    grid, cam, gi = ref['grid'], ref['cam16'], ref['grad_image']
    with harness.settings(**_settings(tag)):
        if integ != O.DIRECT:
            gg, img = harness.render_backward(grid, cam, x['W'], x['H'], x['spp'], ref['sampler_2d'], gi, integ)
            gates[1] = max(gates[1], P.grad_tol(dict(name='refshim_' + name, grid=x['grid'], cam=x['cam'], W=x['W'], H=x['H'], spp=x['spp'],
                                                     offsets=x['offs'].float(), grad_image=x['gi'].float()), integ))
        else:
            variant = 1 if kw.get('detach_indirect_si') else (2 if kw.get('decouple_reparam') else 0)
            gg, galb, gp, img = harness.render_direct_backward(
                grid, cam, x['W'], x['H'], x['spp'], ref['sampler_2d'], x['emitter_u'].numpy(), ref['albedo'], gi, tuple(ref['env']),
                hide_emitters=bool(kw.get('hide_emitters')), bsdf_u=x['bsdf_u'].numpy() if kw.get('use_mis') else None, variant=variant)
            assert rel_l2(galb, ref[f'galb_{tag}']) < gates[3], (rel_l2(galb, ref[f'galb_{tag}']), gates[3])
            assert rel_l2(gp, ref[f'gradp_{tag}']) < max(gates[2], 2 * gates[1]), (rel_l2(gp, ref[f'gradp_{tag}']), gates)
    assert rel_l2(img, ref[f'img_{tag}']) < 1e-4
    check_fp32_gradient('refshim_host', name, tag, gg, ref[f'grad_{tag}'], gates[1])


def _check_per_ray(trace, warp_eval, ref):
    """The per-ray gates shared by the host build of the kernel arithmetic and the HIP path (fp32 against the fp64 fixture)."""
    o, d, maxt = (ref[k].astype(np.float32) for k in ('ray_o', 'ray_d', 'ray_maxt'))
    out = trace(o, d, maxt, True)
    hit, fin = np.isfinite(ref['ri_its_t']), np.isfinite(ref['ri_warp_t'])
    its = out['its_t']
    assert (np.isfinite(its) == hit).mean() > 0.995                             # (a grazing ray may flip between fp32 and fp64)
    both = hit & np.isfinite(its)
    assert rel_l2(its[both], ref['ri_its_t'][both]) < 1e-5
    m = fin & np.isfinite(out['warp_t']) & (ref['ri_warp_weight'] > 1e-3)
    assert m.sum() > 100
    for k, tol in (('warp_t', 1e-4), ('warp_weight', 1e-3), ('warp_t_d', 2e-2), ('warp_weight_d', 2e-2)):   # (1 / denom^3 weights)
        assert rel_l2(out[k][m], ref['ri_' + k][m]) < tol, k
    plain = trace(o, d, maxt, False)['its_t']
    assert rel_l2(plain[both], ref['ri_plain_its_t'][both]) < 1e-5
    # WarpField2D.eval on the REFERENCE's trace outputs: isolates the warp field from the trace
    we = warp_eval(o, d, {k: ref['ri_' + k].astype(np.float32) for k in ('warp_t', 'warp_t_d', 'warp_weight', 'warp_weight_d')})
    act = (we['active'] != 0) & (ref['we_div'] != 0)
    assert act.sum() >= 20 and ((we['active'] != 0) != (ref['we_div'] != 0)).mean() < 0.01   # (92 / 24 of the 512 fixture rays carry a warp)
    for k, rk in (('div', 'we_div'), ('a', 'we_a'), ('b', 'we_b'), ('cdir', 'we_cdir')):
        assert rel_l2(we[k][act], ref[rk][act]) < 2e-3, k                        # (fp32 evaluation of 1 / |g|^4 terms; per-ray gates: test_gpu_parity.py)
    return both


@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
def test_kernel_math_matches_reference_code_per_ray(harness, name):
    """Host build of trace_diff / trace_plain / warp_coefficients against what the reference's shapes.py / warp.py returned --
    the same gates as the HIP test below, so that a gate the GPU box would trip over is tripped over here first."""
    ref = load(name)
    _check_per_ray(lambda o, d, maxt, diff: harness.trace(ref['grid'], o, d, maxt, diff=diff),
                   lambda o, d, tr: harness.warp_eval(ref['grid'], o, d, tr), ref)


# ---------------------------------------------------------------------------------------------------------------- HIP path
@pytest.fixture(scope='module')
def dsdf(built):
    import dsdf as m
    m.load()
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['sphere16', 'blob32'])
def test_gpu_matches_reference_code_per_ray(dsdf, name):
    """dsdf_trace / dsdf_warp_eval / dsdf_surface_interaction against what the reference's shapes.py / warp.py returned."""
    ref = load(name)
    g = dsdf.SdfGrid(torch.from_numpy(ref['grid']).cuda())
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().cuda()
    host = lambda r: {k: v.cpu().numpy() for k, v in r.items()}
    both = _check_per_ray(lambda o, d, maxt, diff: host(dsdf.trace(g, dev(o), dev(d), dev(maxt), differentiable=diff)),
                          lambda o, d, tr: host(dsdf.warp_eval(g, dev(o), dev(d), {k: dev(v) for k, v in tr.items()})), ref)
    o, d = dev(ref['ray_o']), dev(ref['ray_d'])
    si = dsdf.surface_interaction(g, o[torch.from_numpy(both).cuda()], d[torch.from_numpy(both).cuda()],
                                  torch.from_numpy(ref['ri_its_t'][both]).float().cuda())
    assert rel_l2(si['p'].cpu().numpy(), ref['si_p'][both]) < 1e-5 and rel_l2(si['n'].cpu().numpy(), ref['si_n'][both]) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('name,tag', [('sphere16', 'sil'), ('sphere16', 'shade'), ('sphere16', 'direct'), ('blob32', 'sil'), ('blob32', 'shade'),
                                      ('blob32', 'sil_notnorm'), ('blob32', 'direct'), ('blob32', 'direct_mis'), ('blob32', 'direct_primary'),
                                      ('blob32', 'direct_notnorm'), ('blob32', 'direct_decouple')])
def test_gpu_matches_reference_code_render(dsdf, name, tag):
    """dsdf_render_backward with the BUILT-IN sampler seeded like ReparamIntegrator.prepare against integrator.render /
    render_backward of the reference's code."""
    ref = load(name)
    x = inputs(ref)
    integ, kw = TAGS[tag]
    gates = _gates(x, tag, ref)
    g = dsdf.SdfGrid(torch.from_numpy(ref['grid']).cuda())
    for k, v in _settings(tag).items():
        setattr(g.params, k, v)
    sen = dsdf.Sensor(ref['origin'], resx=x['W'], resy=x['H'])
    s = sen.to_struct()
    rec = np.array(list(s.origin) + list(s.left) + list(s.up) + list(s.dir) + [s.tan_half_fov, 0, 0, 0], np.float32)
    assert np.array_equal(rec, ref['cam16']), "the product's look_at yields the sensor record of the fixture"
    gi = torch.from_numpy(ref['grad_image']).cuda()[None]
    gp = torch.zeros(3, device='cuda')
    if integ != O.DIRECT:
        name_of = {O.SILHOUETTE: 'sdf_silhouette_reparam', O.SIMPLE_SHADING: 'sdf_simple_shading_reparam'}[integ]
        gg, img = dsdf.render_backward(g, sen, x['spp'], gi, seeds=[x['seed']], integrator=name_of, return_image=True, grad_p=gp)
        gates[1] = max(gates[1], P.grad_tol(dict(name='refshim_' + name, grid=x['grid'], cam=x['cam'], W=x['W'], H=x['H'], spp=x['spp'],
                                                 offsets=x['offs'].float(), grad_image=x['gi'].float()), integ))
    else:
        sh = dsdf.Shading(torch.from_numpy(ref['albedo']).cuda(), tuple(float(e) for e in ref['env']), use_mis=bool(kw.get('use_mis')),
                          hide_emitters=bool(kw.get('hide_emitters')), detach_indirect_si=bool(kw.get('detach_indirect_si')),
                          decouple_reparam=bool(kw.get('decouple_reparam')))
        galb = torch.zeros_like(sh.albedo)
        gg, img = dsdf.render_backward(g, sen, x['spp'], gi, seeds=[x['seed']], integrator='sdf_direct_reparam', return_image=True,
                                       shading=sh, grad_albedo=galb, grad_p=gp)
        ea = rel_l2(galb.cpu().numpy(), ref[f'galb_{tag}'])
        assert ea < gates[3], (ea, gates[3])
    assert rel_l2(img[0].cpu().numpy(), ref[f'img_{tag}']) < 1e-4
    e = check_fp32_gradient('refshim_gpu', name, tag, gg.cpu().numpy(), ref[f'grad_{tag}'], gates[1])
    assert rel_l2(gp.cpu().numpy(), ref[f'gradp_{tag}']) < max(gates[2], 2 * gates[1], 2 * e)


# ---------------------------------------------------------------------------------------------------------------- provenance
@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'python')), reason="the reference checkout is not on this machine (GPU box)")
def test_fixture_is_what_the_reference_code_produces(tmp_path):
    """Re-runs tools/make_reference_fixtures.py --shim for the small case: the reference's files, imported from the checkout,
    reproduce the committed fixture (so the file cannot have been edited, or made by anything but that code)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_reference_fixtures.py'), '--shim', '--reference', REFERENCE,
                        '--out', str(tmp_path), '--cases', 'sphere16', '--tags', 'sil', 'shade', 'direct_mis'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    new, old = np.load(tmp_path / 'refshim_sphere16.npz'), load('sphere16')
    for k in new.files:
        if new[k].dtype.kind == 'f':
            assert rel_l2(new[k][np.isfinite(new[k])], old[k][np.isfinite(old[k])]) < 1e-12, k
    assert b'refshim' in bytes(old['mitsuba_version'])


# ---------------------------------------------------------------------------------------------------------------- the reference at ITS OWN precision
# tests/golden/refshim32_<case>.npz: the same generator with REFSHIM_DTYPE=float32 -- the reference's statements, in the reference's
# operation order, in the arithmetic type of its CPU variant (llvm_ad_rgb, python/optimize.py:70-78).  north_star asks for gradients
# within 1e-4 of "the reference"; these files show what the reference's OWN fp32 evaluation keeps of its fp64 evaluation
# (dL/dsdf: 2e-4 ... 1.5e-3 on these three small cases, growing with the sample count like the oracles' fp32 floors) and give the
# kernels an fp32-vs-fp32 comparison next to the fp32-vs-fp64 one.  Three numbers per case:
#     floor = |ref32 - ref64|      e64 = |HIP - ref64|      e32 = |HIP - ref32|          (relative L2 of dL/dsdf)
# All three are draws of ONE heavy-tailed variable -- the fp32 rounding of an estimator whose 1 / denom^3 weights let single samples
# carry a third of |g|^2 -- so on cases of a few thousand lanes the ratio e64 / floor scatters between 0.5
# and 5 (host build on the nine runs below: 0.47 ... 4.8, geometric mean 1.0).  Gates: per run e64 <= max(6 x floor, 1e-4); over
# the nine runs the GEOMETRIC MEAN of e64 / floor <= 2 -- "as close to the reference's exact result as the reference's own fp32
# evaluation is, within a factor two".  (The per-case gates of the tests above, from the ORACLES' fp32 floors with their
# one-footprint rule, stay in force; these add the reference's own code at its own precision as the third witness.)
# Round 6: per run 3 x (was 6 x) with the one-footprint rule of check_fp32_gradient, geometric mean 1.5 (was 2) --
# possible since the camera ray of the gradient passes is built with the IEEE sequences (profiles/r06_ieee_sites.md: the hardware
# reciprocals THERE were the whole 5 x of `c1_spp4 shade`; GPU geometric mean 1.40 -> 0.95).
REF32_RUN_FACTOR, REF32_MEAN_FACTOR = 3.0, 1.5
FP32_RUNS = [('sphere16', 'sil'), ('sphere16', 'shade'), ('sphere16', 'direct'), ('blob32', 'sil'), ('blob32', 'shade'), ('blob32', 'direct'),
             ('blob32', 'direct_mis'), ('blob32_spp64', 'sil'), ('blob32_spp64', 'shade'),
             # BASELINE.json configs[0] sizes (64^3, 128 x 128; spp 4: 70 k lanes) -- the reference's own CPU-runnable case
             ('c1_spp4', 'sil'), ('c1_spp4', 'shade'),
             # round 6: the same case at spp 16 (279 k lanes)
             ('c1_spp16', 'sil'), ('c1_spp16', 'shade')]
# ... and BASELINE.json configs[1] sizes (the bench scene at 128^3, view 0 of the 12-ring, 256 x 256, spp 4: 270 k lanes) when the files
# are there (tools/make_reference_fixtures.py --cases c2_spp4: 45 min of the stand-in per precision)
if os.path.isfile(os.path.join(GOLD, 'refshim32_c2_spp4.npz')) and os.path.isfile(os.path.join(GOLD, 'refshim_c2_spp4.npz')):
    FP32_RUNS.append(('c2_spp4', 'sil'))
FP32_CASES = sorted({n for n, _ in FP32_RUNS}, key=[n for n, _ in FP32_RUNS].index)


def load32(name):
    return _with_grid(np.load(os.path.join(GOLD, f'refshim32_{name}.npz')))


def reference_floor(name, tag):
    r32, r64 = load32(name), load(name)
    return dict(grad=rel_l2(r32[f'grad_{tag}'], r64[f'grad_{tag}']), img=rel_l2(r32[f'img_{tag}'], r64[f'img_{tag}']),
                gradp=rel_l2(r32[f'gradp_{tag}'], r64[f'gradp_{tag}']))


@pytest.mark.parametrize('name', FP32_CASES)
def test_reference_fp32_run_sees_the_same_inputs(name):
    r32, r64 = load32(name), load(name)
    for k in ('grid', 'cam16', 'sampler_2d', 'grad_image', 'albedo', 'env'):
        assert np.array_equal(r32[k], r64[k]), k
    assert int(r32['spp']) == int(r64['spp']) and int(r32['seed']) == int(r64['seed'])


@pytest.mark.parametrize('name,tag', FP32_RUNS)
def test_reference_code_fp32_floor(name, tag):
    """What the reference's own code loses between fp64 and its own precision: images stay within north_star's 1e-4, dL/dsdf does
    NOT (that is the point), and the fp32 floors the other tests derive from the oracles' fp32 runs are floors of the same
    estimator -- the torch oracle in fp32 lands within a factor 10 of the reference's fp32 run, either way."""
    f = reference_floor(name, tag)
    assert f['img'] < 1e-4, f
    assert 1e-5 < f['grad'] < 1e-2, f
    ref = load(name)
    if name.startswith(('c1_', 'c2_')):
        # (config size: the C restatement's fp32 build instead of the torch oracle's -- the same floor the config-size gates use)
        import c_oracle
        integ = TAGS[tag][0]
        g32, _ = c_oracle.render_backward(c_oracle.load(False), ref['grid'], ref['cam16'], int(ref['W']), int(ref['H']), int(ref['spp']),
                                          ref['sampler_2d'], ref['grad_image'], integ, True)
        floor_oracle = rel_l2(g32, ref[f'grad_{tag}'])
    else:
        o32 = oracle_run(inputs(ref), tag, torch.float32)
        floor_oracle = rel_l2(o32[1].numpy().astype(np.float64), ref[f'grad_{tag}'])
    assert floor_oracle / 10 < f['grad'] < floor_oracle * 10, (f, floor_oracle)
    P.record('reference_fp32_floor', case=name, tag=tag, floor_reference=f['grad'], floor_torch_oracle=floor_oracle, image=f['img'])


def _three_columns(kind, name, tag, gg):
    r32, r64 = load32(name), load(name)
    f = rel_l2(r32[f'grad_{tag}'], r64[f'grad_{tag}'])
    e64, e32 = rel_l2(gg, r64[f'grad_{tag}']), rel_l2(gg, r32[f'grad_{tag}'])
    P.record(kind, case=name, tag=tag, ref32_vs_ref64=f, hip_vs_ref64=e64, hip_vs_ref32=e32)
    gate = max(REF32_RUN_FACTOR * f, P.NORTH_STAR)
    if e64 > gate:
        # ONE sample footprint may be set aside (check_fp32_gradient): the rest must pass, the footprint itself be reproduced to 0.5 %
        rest, centres, keep = P.greedy_blocks(gg, r64[f'grad_{tag}'], gate, max_blocks=1)
        inside = rel_l2(np.asarray(gg)[~keep], np.asarray(r64[f'grad_{tag}'])[~keep])
        P.record(kind + '_footprint', case=name, tag=tag, plain=e64, rest=rest, inside=inside, gate=gate, centre=list(centres[0]) if centres else None)
        assert rest <= gate and inside < 5e-3, (kind, name, tag, dict(ref32_vs_ref64=f, vs_ref64=e64, vs_ref32=e32, gate=gate, rest=rest, inside=inside))
    return e64 / f


def _mean_gate(kind, ratios):
    gm = float(np.exp(np.mean(np.log(ratios))))
    P.record(kind + '_mean', geometric_mean_ratio=gm, ratios=[float(r) for r in ratios])
    assert gm <= REF32_MEAN_FACTOR, (kind, gm, ratios)


def _host_backward(harness, ref, x, tag):
    integ, kw = TAGS[tag]
    grid, cam, gi = ref['grid'], ref['cam16'], ref['grad_image']
    with harness.settings(**_settings(tag)):
        if integ != O.DIRECT:
            return harness.render_backward(grid, cam, x['W'], x['H'], x['spp'], ref['sampler_2d'], gi, integ)[0]
        return harness.render_direct_backward(grid, cam, x['W'], x['H'], x['spp'], ref['sampler_2d'], x['emitter_u'].numpy(), ref['albedo'], gi,
                                              tuple(ref['env']), bsdf_u=x['bsdf_u'].numpy() if kw.get('use_mis') else None)[0]


def test_kernel_math_within_the_reference_fp32_floor(harness):
    """The kernel arithmetic (host build) is as close to the reference's fp64 result as the reference's own fp32 run is."""
    ratios = []
    for name, tag in FP32_RUNS:
        ref = load(name)
        ratios.append(_three_columns('reference_fp32_host', name, tag, _host_backward(harness, ref, inputs(ref), tag)))
    _mean_gate('reference_fp32_host', ratios)


def _gpu_backward(dsdf, name, tag):
    ref = load(name)
    x = inputs(ref)
    integ, kw = TAGS[tag]
    g = dsdf.SdfGrid(torch.from_numpy(ref['grid']).cuda())
    sen = dsdf.Sensor(ref['origin'], resx=x['W'], resy=x['H'])
    gi = torch.from_numpy(ref['grad_image']).cuda()[None]
    if integ != O.DIRECT:
        name_of = {O.SILHOUETTE: 'sdf_silhouette_reparam', O.SIMPLE_SHADING: 'sdf_simple_shading_reparam'}[integ]
        gg, img = dsdf.render_backward(g, sen, x['spp'], gi, seeds=[x['seed']], integrator=name_of, return_image=True)
    else:
        sh = dsdf.Shading(torch.from_numpy(ref['albedo']).cuda(), tuple(float(e) for e in ref['env']), use_mis=bool(kw.get('use_mis')))
        gg, img = dsdf.render_backward(g, sen, x['spp'], gi, seeds=[x['seed']], integrator='sdf_direct_reparam', return_image=True, shading=sh,
                                       grad_albedo=torch.zeros_like(sh.albedo))
    assert rel_l2(img[0].cpu().numpy(), ref[f'img_{tag}']) < 1e-4 and rel_l2(img[0].cpu().numpy(), load32(name)[f'img_{tag}']) < 1e-4
    return gg.cpu().numpy()


@pytest.mark.gpu
def test_gpu_within_the_reference_fp32_floor(dsdf):
    """dsdf_render_backward against the reference's code at BOTH precisions (built-in sampler, seeded like ReparamIntegrator.prepare)."""
    ratios = [_three_columns('reference_fp32_gpu', name, tag, _gpu_backward(dsdf, name, tag)) for name, tag in FP32_RUNS]
    _mean_gate('reference_fp32_gpu', ratios)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'python')), reason="the reference checkout is not on this machine (GPU box)")
def test_fp32_fixture_is_what_the_reference_code_produces(tmp_path):
    """Provenance of the fp32 files: REFSHIM_DTYPE=float32 + the same generator reproduce the committed file bit for bit in its fp32
    values (images and gradients to 1e-6: torch's fp32 reductions are not bit-reproducible across thread counts)."""
    env = dict(os.environ, REFSHIM_DTYPE='float32')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_reference_fixtures.py'), '--shim', '--reference', REFERENCE,
                        '--out', str(tmp_path), '--cases', 'sphere16', '--tags', 'sil', 'shade'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:]
    new, old = np.load(tmp_path / 'refshim32_sphere16.npz'), load32('sphere16')
    for k in new.files:
        if new[k].dtype.kind == 'f':
            assert rel_l2(new[k][np.isfinite(new[k])], old[k][np.isfinite(old[k])]) < 1e-5, k
