#!/usr/bin/env python3
"""REFERENCE-SIDE fixture generator: executes rgl-epfl/differentiable-sdf-rendering's OWN python/ files on the seeded inputs of
tests/cases.py and writes golden files in the layout tests/test_golden.py reads.  Two ways to run it:

(1) the real stack (Mitsuba 3 + Dr.Jit; not available in this repository's build container, never run by its authors):

        pip install mitsuba fastsweep                       # the reference's README.md:48
        python tools/make_reference_fixtures.py --reference /path/to/differentiable-sdf-rendering [--variant llvm_ad_rgb]
        -> tests/golden/ref_<case>.npz                      # pins oracle and HIP path to the reference AND its dependencies

(2) `--shim`: the same reference files, imported from --reference (default /root/reference), run on top of tools/refshim/ -- a
    torch-backed stand-in for the subset of Dr.Jit / Mitsuba 3 they use (tools/refshim/_core.py says exactly what is restated):

        python tools/make_reference_fixtures.py --shim
        -> tests/golden/refshim_<case>.npz                  # COMMITTED: pins the oracle to the reference's first-party code

    What (2) pins: python/shapes.py (SDFBase.ray_intersect, eval_trace_weight, ray_intersect_non_diff, the refinement loop,
    compute_surface_interaction, Grid3d.eval*), python/warp.py (WarpField2D.weight / eval / ray_intersect), python/math_util.py,
    python/integrators/reparam.py (prepare, render, eval_sample, render_backward, ray_intersect / ray_test),
    sdf_silhouette_reparam.py, sdf_simple_shading_reparam.py, sdf_direct_reparam.py and python/configs.py (the `warp` settings)
    run LITERALLY.  What it does not pin: the third-party layer underneath (arrays, AD, loops, cubic texture, sensor, film,
    sampler, BSDF, emitter), which the stand-in restates like the oracle does.  It runs in fp64 (REFSHIM_DTYPE=float32 for the
    reference's own precision), so the comparison with the fp64 oracle is sharp: ~1e-9 instead of the fp32 floor.

What is dumped, per case (sphere16, blob32 of tests/cases.py; the grid and the fp32 sensor record are stored, so the two sides
cannot drift):
  * `sampler_2d`          first `next_2d()` of the `independent` sampler seeded like ReparamIntegrator.prepare
                          (python/integrators/reparam.py:37-54)
  * `eval_*`              Grid3d.eval_all at random points (python/shapes.py:438-450): value, gradient, Hessian
  * `ri_*`                SDFBase.ray_intersect(ray, warp=WarpField2D) for camera rays (python/shapes.py:115-288): its_t, warp_t,
                          warp_t_d, warp_weight, warp_weight_d;  `ri_plain_its_t`: ray_intersect_non_diff (python/shapes.py:290-339)
  * `we_dir`, `we_div`    WarpField2D.eval at those rays (python/warp.py:47-96); with --shim also `we_a`, `we_b`, `we_cdir`:
                          d(div)/dv, d(div)/dg, d(dir)/dv by AD through the reference's eval
  * `si_p`, `si_n`        compute_surface_interaction for the hit rays (python/shapes.py:347-366)
  * per integrator tag (sil, shade[, direct, direct_mis]): `img_<tag>` = integrator.render (python/integrators/reparam.py:120-185),
    `grad_<tag>` = d(sum(img * grad_image))/d(sdf.data), `gradp_<tag>` = .../d(sdf.p) through render_backward (:187-190)
    [`galb_<tag>` = .../d(reflectance volume)]
  * `tf_general_*`, `tf_axis_*`: Grid3d(data, transform) (python/shapes.py:378-450) with a non-axis-aligned and an axis-aligned `to_world` and
    sdf.p = `tf_p`: eval_all at `eval_pts`, ray_intersect on the camera rays, image / gradients of the two scene-free integrators
  * `aov_<tag>` (sil_aovs, direct_aovs, sil_aovs_noreparam): the (H, W, 14) image of integrator.render with `use_aovs` and
    `warp_field.return_aovs` (python/integrators/reparam.py:160-165, 263-267)
"""
import argparse
import inspect
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def case_inputs(name):
    """The seeded inputs of tests/cases.py WITHOUT the product: grids from the oracle's closed forms, the rest from torch's
    seeded generator exactly as tests/cases.py draws it."""
    import torch
    cfg = {'sphere16': (16, 1, 0, 16, 16, 4, 1), 'blob32': (32, 3, 1, 24, 24, 8, 2),
           'blob32_spp64': (32, 3, 2, 12, 12, 64, 3),                # (the same tuples as tests/cases.py)
           # BASELINE.json configs[0] -- "sphere 64^3 SDF, 1 view, 128 x 128", the reference's own CPU-runnable case -- at spp 4
           # (tests/precision.py: C1_spp4; 70 k lanes)
           'c1_spp4': (64, 1, 0, 128, 128, 4, 21),
           # round 6: the same case at spp 16 (279 k lanes) and a BASELINE.json configs[1]-size case -- bench.py's
           # 128^3 blob scene, view 0 of the 12-ring, 256 x 256, spp 4 (270 k lanes); tests/precision.py: C1_spp16, C2 sizes
           'c1_spp16': (64, 1, 0, 128, 128, 16, 22), 'c2_spp4': (128, 12, 0, 256, 256, 4, 29)}[name]
    R, ncam, icam, W, H, spp, seed = cfg
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sdf_oracle as O                                          # grids + camera ring: plain torch, no product code
    if name == 'c2_spp4':
        sys.path.insert(0, ROOT)
        import bench                                                # (the bench scene's generator: plain torch, no product code)
        grid = bench.synth_grid(128, 'cpu').float().numpy()
    else:
        grid = (O.sphere_grid(16) if name == 'sphere16' else (O.sphere_grid(64) if name.startswith('c1_') else O.blob_grid(32, n=6, seed=1))).float().numpy()
    gen = torch.Generator().manual_seed(seed)
    torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float32)      # (cases.py draws the explicit offsets first)
    grad_image = torch.randn(H, W, 3, generator=gen, dtype=torch.float32).numpy()
    origin = np.asarray(O.regular_camera_origins(ncam)[icam], np.float64)
    cam16 = O.Camera(origin).params()                               # the fp32 sensor record the C-ABI receives (include/dsdf.h)
    gen2 = torch.Generator().manual_seed(11)                        # tests/cases.py: direct_inputs
    albedo = (torch.rand(6, 5, 4, 3, generator=gen2, dtype=torch.float32) * 0.6 + 0.2).numpy()
    return dict(name=name, grid=grid, W=W, H=H, spp=spp, origin=origin, cam16=cam16, grad_image=grad_image, render_seed=40 + seed,
                albedo=albedo, env=tuple(float(np.float32(e)) for e in (1.0, 0.9, 0.8)))   # (fp32 values: what the C-ABI receives)


def _plain(v):
    """JSON form of a configuration attribute: numbers / strings as they are, callables and classes by name, arrays as lists."""
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if callable(v):
        return getattr(v, '__name__', type(v).__name__)
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    a = np.asarray(v.cpu() if hasattr(v, 'cpu') else v)
    return a.tolist() if a.size <= 16 else {'shape': list(a.shape), 'sum': float(a.astype(np.float64).sum()), 'abs_sum': float(np.abs(a.astype(np.float64)).sum()),
                                            'min': float(a.min()), 'max': float(a.max())}


def sensor_record(s):
    """(origin, fov) of a sensor: a Mitsuba / stand-in perspective sensor or this repository's dsdf.Sensor.  (The film size is not
    part of the record: the reference builds every sensor at 128 x 128 and sets the render resolution per iteration from the
    config's resx / resy / init_res -- set_sensor_res, python/util.py -- the mirror's sensors are built at the config's size.)"""
    if hasattr(s, 'origin'):
        return [float(x) for x in s.origin] + [float(s.fov)]
    m = np.asarray(s.to_world.matrix if hasattr(s, 'to_world') else s.world_transform().matrix, np.float64).reshape(4, 4)
    return [float(x) for x in m[:3, 3]] + [float(getattr(s, 'fov', 39.0))]


def opt_config_record(c):
    """Every attribute of a resolved opt-config (python/opt_configs.py: SceneConfig / SdfConfig and the Variable objects of
    python/variables.py) in JSON form -- works on the reference's objects and on the mirror's."""
    rec = {k: _plain(v) for k, v in vars(c).items() if k not in ('sensors', 'variables', 'device')}
    rec['sensors'] = [sensor_record(s) for s in c.sensors]
    rec['variables'] = [dict({k: _plain(v) for k, v in vars(var).items() if k != 'device'}, cls=type(var).__name__) for var in c.variables]
    return rec


def method_config_record(cfg):
    rec = {k: _plain(v) for k, v in vars(cfg).items()}
    wf = cfg.get_warpfield(None) if type(cfg).__name__.lower() not in ('finitedifferences',) else None
    rec['warpfield'] = None if wf is None else dict({k: _plain(v) for k, v in vars(wf).items() if k in
                                                      ('max_reparam_depth', 'edge_eps', 'weight_strategy', 'clamping_thresh', 'normalize_warp_field')},
                                                     cls=type(wf).__name__)
    return rec


def config_table():
    """All named opt-configs that need no scene file (82 of the 85: `torus-shadow-1`, `mirror-opt-1`, `mirror-opt-hq` read their
    sensors from an XML scene, python/opt_configs.py:215-230) and all method configs whose warp field is on the path."""
    import configs
    import opt_configs
    out = {'opt': {}, 'skipped': [], 'method': {}}
    for name in opt_configs.SCENE_CONFIGS:
        try:
            c = opt_configs.get_opt_config(name)
        except (AttributeError, FileNotFoundError, OSError) as e:
            out['skipped'].append(name)
            continue
        out['opt'][name] = opt_config_record(c[0] if isinstance(c, tuple) else c)
    for name in ('warp', 'warpprimary', 'warpnotnormalized', 'onlyshadinggrad', 'finitedifferences'):
        out['method'][name] = method_config_record(configs.get_config(name))
    return out


def transforms():
    """The two `to_world` matrices of the transform fixtures (the same as tests/test_to_world.py: GENERAL, AXIS_ALIGNED): a rotation
    of the unit cube about its centre followed by a translation."""
    def rot(axis, deg):
        a = np.radians(deg)
        c, s = np.cos(a), np.sin(a)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        R = np.eye(3)
        R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
        return R

    def about_centre(R, shift):
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = np.array([0.5, 0.5, 0.5]) - R @ np.array([0.5, 0.5, 0.5]) + np.asarray(shift)
        return T
    return {'tf_general': about_centre(rot(1, 25) @ rot(2, -10), (0.01, 0.0, -0.02)),
            'tf_axis': about_centre(rot(1, 90) @ rot(0, 180), (0.02, -0.015, 0.01))}


def cols(a, k):
    """A k-vector array of the stack as (n, k) numpy."""
    arr = np.array(a)
    if arr.ndim == 2 and arr.shape[0] == k and arr.shape[1] != k:
        arr = arr.T
    return np.ascontiguousarray(arr.reshape(-1, k))


def mat33(m, mi):
    arr = np.array(m)
    if arr.ndim == 3 and arr.shape[-2:] == (3, 3):
        return arr
    return np.array([[np.array(m[i, j]) for j in range(3)] for i in range(3)]).transpose(2, 0, 1)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--reference', default='/root/reference', help="checkout of rgl-epfl/differentiable-sdf-rendering (its python/ is imported)")
    ap.add_argument('--shim', action='store_true', help="run the reference's files on tools/refshim instead of Mitsuba 3 / Dr.Jit")
    ap.add_argument('--variant', default='llvm_ad_rgb', help="Mitsuba variant; the reference's CPU path is llvm_ad_rgb")
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--cases', nargs='*', default=['sphere16', 'blob32'])
    ap.add_argument('--tags', nargs='*', default=None, help='subset of the integrator runs (default: all)')
    ap.add_argument('--config-table', action='store_true', help="dump python/opt_configs.py and python/configs.py (every named configuration, "
                    "resolved) to <out>/<prefix>_config_table.json instead of the render fixtures")
    ap.add_argument('--tf-cases', nargs='*', default=['sphere16'], help='cases that also get the Grid3d(transform) section (slow on the stand-in)')
    ap.add_argument('--light', action='store_true', help='images and gradients of the integrator runs only (no per-ray sections)')
    args = ap.parse_args()
    # REFSHIM_DTYPE=float32: the stand-in's arrays are fp32 -- the reference's own statements in the reference's own operation order at
    # the precision of its CPU variant (llvm_ad_rgb, python/optimize.py:70-78) -> tests/golden/refshim32_<case>.npz: the third witness
    # of the fp32 floor (next to the fp32 builds of the two oracles), and a direct fp32-vs-fp32 comparison for the HIP path
    fp32 = args.shim and os.environ.get('REFSHIM_DTYPE', 'float64') == 'float32'
    if fp32:
        args.tf_cases = []
        if args.tags is None:
            args.tags = ['sil', 'shade', 'direct', 'direct_mis']

    if args.shim:
        sys.path.insert(0, os.path.join(HERE, 'refshim'))           # `import drjit`, `import mitsuba`, `import fastsweep` -> the stand-in
    import drjit as dr
    import mitsuba as mi
    mi.set_variant(args.variant)
    sys.path.insert(0, os.path.join(args.reference, 'python'))
    import configs                                                  # registers the integrators (python/configs.py:4-7)
    from shapes import Grid3d                                       # python/shapes.py:375
    from constants import SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P        # python/constants.py:18-19
    prefix = ('refshim32' if fp32 else 'refshim') if args.shim else 'ref'
    if args.config_table:
        import json
        fn = os.path.join(args.out, f'{prefix}_config_table.json')
        t = config_table()
        with open(fn, 'w') as f:                          # one opt-config per line
            f.write('{"method": ' + json.dumps(t['method'], sort_keys=True) + ',\n"skipped": ' + json.dumps(t['skipped']) + ',\n"opt": {\n')
            names = sorted(t['opt'])
            for i, n in enumerate(names):
                f.write(json.dumps(n) + ': ' + json.dumps(t['opt'][n], sort_keys=True) + (',\n' if i + 1 < len(names) else '\n'))
            f.write('}}\n')
        print(fn)
        return

    for name in args.cases:
        c = case_inputs(name)
        W, H, spp, seed = c['W'], c['H'], c['spp'], c['render_seed']
        out = dict(grid=c['grid'], origin=c['origin'], cam16=c['cam16'], W=np.int64(W), H=np.int64(H), spp=np.int64(spp),
                   seed=np.int64(seed), grad_image=c['grad_image'], albedo=c['albedo'], env=np.asarray(c['env'], np.float32),
                   mitsuba_version=np.bytes_(mi.__version__), variant=np.bytes_(args.variant))

        # ---- sensor: python/util.py:115-138 (get_regular_cameras builds the same dict for its ring origins).  With --shim the
        # frame comes from the fp32 record (left, up, dir, origin: what look_at returns, rounded) so that both sides of the fp64
        # comparison see bit-identical inputs (the estimator amplifies a 6e-8 input rounding to 1e-4)
        o = c['origin']
        to_world = mi.ScalarTransform4f.look_at(mi.ScalarPoint3f(float(o[0]), float(o[1]), float(o[2])), [0.5, 0.5, 0.5], [0, 1, 0])
        fov = 39.0
        if args.shim:
            r = c['cam16'].astype(np.float64)
            frame = mi.ScalarTransform4f.from_frame(r[3:6], r[6:9], r[9:12], r[0:3])       # (stand-in only) look_at's result, rounded
            assert np.abs(np.asarray(frame.matrix) - np.asarray(to_world.matrix)).max() < 1e-6
            to_world = frame
            fov = float(np.degrees(2.0 * np.arctan(r[12])))
        sensor = mi.load_dict({
            'type': 'perspective', 'fov': fov, 'to_world': to_world, 'sampler': {'type': 'independent'},
            'film': {'type': 'hdrfilm', 'width': W, 'height': H, 'pixel_format': 'rgb', 'pixel_filter': {'type': 'gaussian'},
                     'sample_border': True}})

        # ---- the independent sampler, seeded like ReparamIntegrator.prepare (python/integrators/reparam.py:37-54)
        n_lanes = (W + 4) * (H + 4) * spp
        smp = sensor.sampler().clone()
        smp.set_sample_count(spp)
        smp.set_samples_per_wavefront(spp)
        smp.seed(seed, n_lanes)
        out['sampler_2d'] = cols(smp.next_2d(), 2).astype(np.float32)

        # ---- the SDF object on the raw tensor (a str argument would redistance it: python/shapes.py:384-386)
        def make_sdf():
            return Grid3d(mi.TensorXf(c['grid'][..., None]))

        sdf = make_sdf()
        wf = configs.get_config('warp').get_warpfield(sdf)          # python/configs.py:36-40: strategy 6, edge_eps 0.01, clamp 0.05
        pts = ray = None
        if not args.light:
            pts, ray = per_ray_sections(args, dr, mi, sdf, wf, sensor, out)

        # ---- A12-A17: image and gradients of the integrators
        runs = [('sdf_silhouette_reparam', 'sil', {}, 'warp'), ('sdf_simple_shading_reparam', 'shade', {}, 'warp')]
        if args.shim:
            D = 'sdf_direct_reparam'
            runs += [(D, 'direct', {}, 'warp'), (D, 'direct_mis', {'use_mis': True}, 'warp'),
                     # the integrator's own properties (sdf_direct_reparam.py:12-14) ...
                     (D, 'direct_hide', {'hide_emitters': True}, 'warp'), (D, 'direct_detach', {'detach_indirect_si': True}, 'warp'),
                     (D, 'direct_decouple', {'decouple_reparam': True}, 'warp'),
                     (D, 'direct_mis_decouple', {'use_mis': True, 'decouple_reparam': True}, 'warp'),
                     # ... and the method configs that change the warp field (python/configs.py:63-75, 96-109, 112-125)
                     (D, 'direct_primary', {}, 'warpprimary'), (D, 'direct_mis_primary', {'use_mis': True}, 'warpprimary'),
                     (D, 'direct_notnorm', {}, 'warpnotnormalized'), ('sdf_silhouette_reparam', 'sil_notnorm', {}, 'warpnotnormalized'),
                     ('sdf_simple_shading_reparam', 'shade_notnorm', {}, 'warpnotnormalized'), (D, 'direct_onlyshading', {}, 'onlyshadinggrad'),
                     # ... and the debug images: `use_aovs` on the integrator + `return_aovs` on the warp field (reparam.py:160-165,
                     # 263-267; warp.py:105-106; shapes.py:240-242) -> (H, W, 3 + 11) images, stored whole as `aov_<tag>`
                     ('sdf_silhouette_reparam', 'sil_aovs', {'use_aovs': True}, 'warp'), (D, 'direct_aovs', {'use_aovs': True}, 'warp'),
                     ('sdf_silhouette_reparam', 'sil_aovs_noreparam', {'use_aovs': True}, 'onlyshadinggrad'),
                     # ... and the antithetic pair of every sample (reparam.py:19, 167-178: a second eval_sample at the mirrored film
                     # position `pos - r + 1` with a clone of the sampler, into the same block)
                     ('sdf_silhouette_reparam', 'sil_anti', {'antithetic_sampling': True}, 'warp'),
                     (D, 'direct_anti', {'antithetic_sampling': True}, 'warp')]
        for integ_name, tag, props, method in runs:
            if args.tags is not None and tag not in args.tags:
                continue
            # one placeholder shape whose id contains '_sdf_' (python/integrators/reparam.py:66-76) carries the BSDF; with a
            # single shape the integrator never calls into Embree / OptiX (use_optix = len(shapes) > 1)
            scene_dict = {'type': 'scene', 'integrator': dict({'type': integ_name}, **props), 'sensor': sensor,
                          'placeholder_sdf_shape': {'type': 'sphere', 'center': [100, 100, 100], 'radius': 1e-3,
                                                    'bsdf': {'type': 'diffuse'}}}
            direct = tag.startswith('direct')
            if direct:                                              # this repository's spec of the scene side (oracle/sdf_oracle.py header)
                scene_dict['placeholder_sdf_shape']['bsdf'] = {'type': 'diffuse', 'reflectance': {'type': 'gridvolume',
                                                                                                  'data': mi.TensorXf(c['albedo'])}}
                scene_dict['emitter'] = {'type': 'constant', 'radiance': list(c['env'])}
            scene = mi.load_dict(scene_dict)
            integ = scene.integrator()
            integ.sdf = make_sdf()
            integ.warp_field = configs.get_config(method).get_warpfield(integ.sdf)
            if len(inspect.signature(integ.sample).parameters) == 5:
                # python/integrators/sdf_simple_shading_reparam.py:16 still has the 5-argument signature of an older base
                # class, ReparamIntegrator.eval_sample (reparam.py:94) passes 8: the body is run through this argument adapter
                body = integ.sample
                integ.sample = lambda mode, scene_, sampler, ray_, dL, state_in, reparam, active, **kw: body(scene_, sampler, ray_, None, active)
                out[f'adapter_{tag}'] = np.int64(1)
            if 'use_aovs' in props:
                integ.warp_field.return_aovs = True                                             # python/warp.py:17 (set by hand there too)
                with dr.suspend_grad():
                    img = mi.render(scene, sensor=sensor, seed=seed, spp=spp)
                out[f'aov_{tag}'] = np.array(img).astype(np.float64 if args.shim else np.float32)
                assert out[f'aov_{tag}'].shape == (H, W, 3 + len(integ.aov_names())), out[f'aov_{tag}'].shape
                continue
            with dr.suspend_grad():
                img = mi.render(scene, sensor=sensor, seed=seed, spp=spp)                       # python/shape_opt.py:61-63
            out[f'img_{tag}'] = np.array(img)[..., :3].astype(np.float64 if args.shim else np.float32)
            params = mi.traverse(scene)
            keys = [SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P] + [k for k in params if k.endswith('reflectance.volume.data')]
            params.keep(keys)
            for k in keys:
                dr.enable_grad(params[k])
            params.update()
            # the SAME samples for the primal image and the gradient pass: seed_grad = seed, spp_grad = spp (the in-repo tests
            # compare a gradient pass on given samples; python/shape_opt.py:78-80 uses independent ones)
            img = mi.render(scene, params=params, sensor=sensor, seed=seed, spp=spp, seed_grad=seed, spp_grad=spp)
            dr.backward(img * mi.TensorXf(c['grad_image']))
            ft = np.float64 if args.shim else np.float32
            out[f'grad_{tag}'] = np.array(dr.grad(params[SDF_DEFAULT_KEY])).reshape(c['grid'].shape).astype(ft)
            out[f'gradp_{tag}'] = np.array(dr.grad(params[SDF_DEFAULT_KEY_P])).reshape(3).astype(ft)
            if direct:
                out[f'galb_{tag}'] = np.array(dr.grad(params[keys[2]])).reshape(c['albedo'].shape).astype(ft)

        # ---- Grid3d(data, transform) / integrator property `sdf_to_world` (python/shapes.py:378-450, integrators/reparam.py:21-29):
        # lookups at to_local @ (x - p), gradients back through to_local^T, the traced box = world AABB of the transformed cube.
        # `tf_general`: a rotation that is NOT axis-aligned (the AABB outgrows the cube); `tf_axis`: translation + axis-aligned
        # rotation, the transforms this repository's product accepts (tests/test_to_world.py holds the same two matrices)
        if name in args.tf_cases and not args.light and (args.tags is None or any(t.startswith('tf_') for t in args.tags)):
            p0 = [0.01, -0.02, 0.015]
            out['tf_p'] = np.asarray(p0, np.float64)
            for key, T in transforms().items():
                if args.tags is not None and key not in args.tags:
                    continue
                def make_sdf_t():
                    s_ = Grid3d(mi.TensorXf(c['grid'][..., None]), transform=mi.ScalarTransform4f(T))
                    s_.p = mi.Vector3f(*p0)
                    return s_
                sdf_t = make_sdf_t()
                out[f'{key}_matrix'] = np.asarray(T, np.float64)
                v, _, g, _, Hm = sdf_t.eval_all(mi.Point3f(pts[:, 0], pts[:, 1], pts[:, 2]))                 # python/shapes.py:438-450
                out.update({f'{key}_eval_v': np.array(v), f'{key}_eval_g': cols(g, 3), f'{key}_eval_H': mat33(Hm, mi)})
                wf_t = configs.get_config('warp').get_warpfield(sdf_t)
                with dr.suspend_grad():
                    its_t, warp_t, warp_t_d, ww, ww_d = sdf_t.ray_intersect(ray, warp=wf_t)                  # python/shapes.py:115
                out.update({f'{key}_ri_its_t': np.array(its_t), f'{key}_ri_warp_t': np.array(warp_t), f'{key}_ri_warp_t_d': cols(warp_t_d, 3),
                            f'{key}_ri_warp_weight': np.array(ww), f'{key}_ri_warp_weight_d': cols(ww_d, 3)})
                tf_runs = [('sdf_silhouette_reparam', 'sil'), ('sdf_simple_shading_reparam', 'shade')]
                if args.shim and key == 'tf_general':
                    tf_runs.append(('sdf_direct_reparam', 'direct'))          # (the albedo volume stays in WORLD space: a Mitsuba gridvolume)
                for integ_name, tag in tf_runs:
                    sd = {'type': 'scene', 'integrator': {'type': integ_name}, 'sensor': sensor,
                          'placeholder_sdf_shape': {'type': 'sphere', 'center': [100, 100, 100], 'radius': 1e-3, 'bsdf': {'type': 'diffuse'}}}
                    if tag == 'direct':
                        sd['placeholder_sdf_shape']['bsdf'] = {'type': 'diffuse', 'reflectance': {'type': 'gridvolume', 'data': mi.TensorXf(c['albedo'])}}
                        sd['emitter'] = {'type': 'constant', 'radiance': list(c['env'])}
                    scene = mi.load_dict(sd)
                    integ = scene.integrator()
                    integ.sdf = make_sdf_t()
                    integ.warp_field = configs.get_config('warp').get_warpfield(integ.sdf)
                    if len(inspect.signature(integ.sample).parameters) == 5:                                 # (see above: argument adapter)
                        body = integ.sample
                        integ.sample = lambda mode, scene_, sampler, ray_, dL, state_in, reparam, active, **kw: body(scene_, sampler, ray_, None, active)
                    params = mi.traverse(scene)
                    tkeys = [SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P] + [k for k in params if k.endswith('reflectance.volume.data')]
                    params.keep(tkeys)
                    for k in tkeys:
                        dr.enable_grad(params[k])
                    params.update()
                    img = mi.render(scene, params=params, sensor=sensor, seed=seed, spp=spp, seed_grad=seed, spp_grad=spp)
                    dr.backward(img * mi.TensorXf(c['grad_image']))
                    ft = np.float64 if args.shim else np.float32
                    out[f'{key}_img_{tag}'] = np.array(img)[..., :3].astype(ft)
                    out[f'{key}_grad_{tag}'] = np.array(dr.grad(params[SDF_DEFAULT_KEY])).reshape(c['grid'].shape).astype(ft)
                    out[f'{key}_gradp_{tag}'] = np.array(dr.grad(params[SDF_DEFAULT_KEY_P])).reshape(3).astype(ft)
                    if len(tkeys) > 2:
                        out[f'{key}_galb_{tag}'] = np.array(dr.grad(params[tkeys[2]])).reshape(c['albedo'].shape).astype(ft)

        fn = os.path.join(args.out, f'{prefix}_{name}.npz')
        if name == 'c2_spp4':
            # (the 128^3 grid is 7.4 MB of the file and comes from a seeded recipe: stored by name + hash, rebuilt by the tests)
            import hashlib
            out['grid_sha256'] = np.bytes_(hashlib.sha256(np.ascontiguousarray(out.pop('grid')).tobytes()).hexdigest())
            out['grid_recipe'] = np.bytes_('bench.synth_grid(128)')
        np.savez_compressed(fn, **out)
        print(fn, {k: (getattr(v, 'shape', None) or v) for k, v in out.items()})


def per_ray_sections(args, dr, mi, sdf, wf, sensor, out):
    """A1 / A2 / A4 / A6 / A9 of SURVEY 8(a) on random points and camera rays: what the reference's shapes.py / warp.py return per ray."""
    # ---- A1: eval_all at random points (python/shapes.py:438-450)
    rng = np.random.default_rng(7)
    pts = rng.uniform(0.1, 0.9, (256, 3)).astype(np.float32)
    v, _, g, _, Hm = sdf.eval_all(mi.Point3f(pts[:, 0], pts[:, 1], pts[:, 2]))
    out.update(eval_pts=pts, eval_v=np.array(v), eval_g=cols(g, 3), eval_H=mat33(Hm, mi))

    # ---- A2 / A4 / A9: per-ray outputs for camera rays at random film positions
    pos = rng.uniform(0.0, 1.0, (512, 2)).astype(np.float32)
    ray, _ = sensor.sample_ray_differential(0.0, 0.5, mi.Point2f(pos[:, 0], pos[:, 1]), mi.Point2f(0.5))
    with dr.suspend_grad():
        its_t, warp_t, warp_t_d, ww, ww_d = sdf.ray_intersect(ray, warp=wf)                  # python/shapes.py:115
        plain = sdf.ray_intersect_non_diff(ray, True)                                         # python/shapes.py:290
    rayn = mi.Ray3f(ray)
    rayn.d = dr.normalize(rayn.d)
    warp_dir, div = wf.eval(rayn(warp_t), rayn.d, t=warp_t, dt_dx=warp_t_d, active=True,      # python/warp.py:47
                            warp_weight=ww, warp_weight_d=ww_d)
    out.update(ray_pos=pos, ray_o=cols(ray.o, 3), ray_d=cols(ray.d, 3), ray_maxt=np.array(ray.maxt),
               ri_its_t=np.array(its_t), ri_warp_t=np.array(warp_t), ri_warp_t_d=cols(warp_t_d, 3), ri_warp_weight=np.array(ww),
               ri_warp_weight_d=cols(ww_d, 3), ri_plain_its_t=np.array(plain if not isinstance(plain, tuple) else plain[0]),
               we_dir=cols(dr.detach(warp_dir), 3), we_div=np.array(dr.detach(div)))
    with dr.suspend_grad():
        si = sdf.compute_surface_interaction(ray, its_t)                                      # python/shapes.py:347
    out.update(si_p=cols(si.p, 3), si_n=cols(si.n, 3))
    if args.shim:
        out.update(warp_coefficients_by_ad(dr, mi, sdf, wf, rayn, warp_t, warp_t_d, ww, ww_d))
    return pts, ray


def warp_coefficients_by_ad(dr, mi, sdf, wf, rayn, warp_t, warp_t_d, ww, ww_d):
    """d(div)/dv, d(div)/dg and d(dir)/dv of WarpField2D.eval (python/warp.py:47-96) by AD through the reference's own eval: the SDF
    handed to it returns value and gradient as AD leaves (Hessian and the detached copies from the real grid), which is exactly
    the linearisation the HIP path's k_warp_eval reports (--shim only: uses torch autograd directly)."""
    import torch
    x = rayn(warp_t)
    with dr.suspend_grad():
        v0, _, g0, _, H0 = sdf.eval_all(x)
    fin = np.isfinite(np.array(warp_t))
    v = mi.Float(np.where(fin, np.array(v0), 0.0)); g = mi.Vector3f(np.where(fin[:, None], cols(g0, 3), 1.0))
    dr.enable_grad(v, g)

    class Leaf:
        bbox = sdf.bbox

        def eval_all(self, x_):
            return v, dr.detach(v), g, dr.detach(g), H0
    saved = wf.sdf
    wf.sdf = Leaf()
    try:
        wdir, div = wf.eval(x, rayn.d, t=warp_t, dt_dx=warp_t_d, active=True, warp_weight=ww, warp_weight_d=ww_d)
    finally:
        wf.sdf = saved
    a, b = torch.autograd.grad(div.v.sum(), (v.v, g.v), retain_graph=True, allow_unused=True)
    cd = []
    for k in range(3):
        (ck,) = torch.autograd.grad(wdir.v[:, k].sum(), (v.v,), retain_graph=True, allow_unused=True)
        cd.append(torch.zeros_like(v.v) if ck is None else ck)
    z = lambda q, like: torch.zeros_like(like) if q is None else q
    return dict(we_a=z(a, v.v).numpy(), we_b=z(b, g.v).numpy(), we_cdir=torch.stack(cd, -1).numpy())


if __name__ == '__main__':
    main()
