#!/usr/bin/env python3
"""REFERENCE-SIDE fixture generator: runs the real stack (Mitsuba 3 + Dr.Jit + rgl-epfl/differentiable-sdf-rendering's python/)
on the seeded inputs of tests/cases.py and writes tests/golden/ref_<case>.npz in the layout tests/test_golden.py reads.

It CANNOT run in this repository's build container (no mitsuba / drjit, no network) and has never been run by the authors of
this repository: it is written against the reference's sources (file:line cited at every call) so that a maintainer with the
reference's environment can pin the in-repo oracle -- and through it the HIP path -- to the reference itself:

    pip install mitsuba fastsweep                       # the reference's README.md:48
    python tools/make_reference_fixtures.py --reference /path/to/differentiable-sdf-rendering [--variant llvm_ad_rgb]
    python -m pytest tests/test_golden.py -k reference  # oracle (CPU) and HIP (GPU) against the new files

Until such a file exists `tests/test_golden.py::test_*_reference_fixture` SKIP loudly and DESIGN.md keeps saying "parity unpinned".

What is dumped, per case (sphere16, blob32 of tests/cases.py; the grid itself is stored, so the two sides cannot drift):
  * `sampler_2d`          first `next_2d()` of Mitsuba's `independent` sampler seeded like ReparamIntegrator.prepare
                          (python/integrators/reparam.py:37-54)                       -> pins oracle.independent_sampler_2d
  * `ri_*`                SDFBase.ray_intersect(ray, warp=WarpField2D) for the camera rays of `ray_pos`
                          (python/shapes.py:115-288): its_t, warp_t, warp_t_d, warp_weight, warp_weight_d
  * `ri_plain_its_t`      SDFBase.ray_intersect_non_diff (python/shapes.py:290-339)
  * `we_dir`, `we_div`    WarpField2D.eval at those rays (python/warp.py:47-96): warped direction (primal) and divergence
  * `eval_*`              Grid3d.eval_all at random points (python/shapes.py:438-450): value, gradient, Hessian
  * per integrator tag (sil, shade): `img_<tag>` = integrator.render (python/integrators/reparam.py:120-185),
    `grad_<tag>` = d(sum(img * grad_image))/d(sdf.data) and `gradp_<tag>` = .../d(sdf.p) through render_backward (:187-190)
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def case_inputs(name):
    """The seeded inputs of tests/cases.py WITHOUT importing the in-repo oracle for anything but the grids' closed forms
    (numpy only: sphere_grid / blob_grid are restated here so that this script depends on nothing of this repository)."""
    import torch                                                    # (only for the seeded generator that tests/cases.py uses)
    cfg = {'sphere16': (16, 1, 0, 16, 16, 4, 1), 'blob32': (32, 3, 1, 24, 24, 8, 2)}[name]
    R, ncam, icam, W, H, spp, seed = cfg
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sdf_oracle as O                                          # grids + camera ring: plain torch, no product code
    grid = (O.sphere_grid(16) if name == 'sphere16' else O.blob_grid(32, n=6, seed=1)).float().numpy()
    gen = torch.Generator().manual_seed(seed)
    torch.rand((W + 4) * (H + 4) * spp, 2, generator=gen, dtype=torch.float32)      # (cases.py draws the explicit offsets first)
    grad_image = torch.randn(H, W, 3, generator=gen, dtype=torch.float32).numpy()
    origin = np.asarray(O.regular_camera_origins(ncam)[icam], np.float64)
    return dict(name=name, grid=grid, W=W, H=H, spp=spp, origin=origin, grad_image=grad_image, render_seed=40 + seed)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--reference', required=True, help="checkout of rgl-epfl/differentiable-sdf-rendering (its python/ is imported)")
    ap.add_argument('--variant', default='llvm_ad_rgb', help="Mitsuba variant; the reference's CPU path is llvm_ad_rgb")
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    ap.add_argument('--cases', nargs='*', default=['sphere16', 'blob32'])
    args = ap.parse_args()

    import drjit as dr
    import mitsuba as mi
    mi.set_variant(args.variant)
    sys.path.insert(0, os.path.join(args.reference, 'python'))
    import configs                                                  # registers the integrators (python/configs.py:4-7)
    from shapes import Grid3d                                       # python/shapes.py:375
    from constants import SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P        # python/constants.py:18-19

    for name in args.cases:
        c = case_inputs(name)
        W, H, spp, seed = c['W'], c['H'], c['spp'], c['render_seed']
        out = dict(grid=c['grid'], origin=c['origin'], W=np.int64(W), H=np.int64(H), spp=np.int64(spp), seed=np.int64(seed),
                   grad_image=c['grad_image'], mitsuba_version=np.bytes_(mi.__version__), variant=np.bytes_(args.variant))

        # ---- sensor: exactly python/util.py:115-138 (get_regular_cameras builds the same dict for its ring origins)
        o = c['origin']
        sensor = mi.load_dict({
            'type': 'perspective', 'fov': 39.0,
            'to_world': mi.ScalarTransform4f.look_at(mi.ScalarPoint3f(float(o[0]), float(o[1]), float(o[2])), [0.5, 0.5, 0.5], [0, 1, 0]),
            'sampler': {'type': 'independent'},
            'film': {'type': 'hdrfilm', 'width': W, 'height': H, 'pixel_format': 'rgb', 'pixel_filter': {'type': 'gaussian'},
                     'sample_border': True}})

        # ---- the independent sampler, seeded like ReparamIntegrator.prepare (python/integrators/reparam.py:37-54)
        n_lanes = (W + 4) * (H + 4) * spp
        smp = sensor.sampler().clone()
        smp.set_sample_count(spp)
        smp.set_samples_per_wavefront(spp)
        smp.seed(seed, n_lanes)
        out['sampler_2d'] = np.array(smp.next_2d()).T.astype(np.float32).reshape(n_lanes, 2)

        # ---- the SDF object on the raw tensor (a str argument would redistance it: python/shapes.py:384-386)
        def make_sdf():
            return Grid3d(mi.TensorXf(c['grid'][..., None]))

        sdf = make_sdf()
        wf = configs.get_config('warp').get_warpfield(sdf)          # python/configs.py:36-40: strategy 6, edge_eps 0.01, clamp 0.05

        # ---- A1: eval_all at random points (python/shapes.py:438-450)
        rng = np.random.default_rng(7)
        pts = rng.uniform(0.1, 0.9, (256, 3)).astype(np.float32)
        v, _, g, _, Hm = sdf.eval_all(mi.Point3f(pts[:, 0], pts[:, 1], pts[:, 2]))
        out.update(eval_pts=pts, eval_v=np.array(v), eval_g=np.array(g).T.reshape(-1, 3),
                   eval_H=np.array([[np.array(Hm[i, j]) for j in range(3)] for i in range(3)]).transpose(2, 0, 1))

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
        as3 = lambda a: np.array(a).T.reshape(-1, 3)
        out.update(ray_pos=pos, ray_o=as3(ray.o), ray_d=as3(ray.d), ray_maxt=np.array(ray.maxt),
                   ri_its_t=np.array(its_t), ri_warp_t=np.array(warp_t), ri_warp_t_d=as3(warp_t_d), ri_warp_weight=np.array(ww),
                   ri_warp_weight_d=as3(ww_d), ri_plain_its_t=np.array(plain if not isinstance(plain, tuple) else plain[0]),
                   we_dir=as3(dr.detach(warp_dir)), we_div=np.array(dr.detach(div)))

        # ---- A12-A17: image and gradients of the two primary-ray integrators
        for integ_name, tag in (('sdf_silhouette_reparam', 'sil'), ('sdf_simple_shading_reparam', 'shade')):
            # one placeholder shape whose id contains '_sdf_' (python/integrators/reparam.py:66-76); with a single shape the
            # integrator never calls into Embree / OptiX (use_optix = len(shapes) > 1)
            scene = mi.load_dict({'type': 'scene', 'integrator': {'type': integ_name}, 'sensor': sensor,
                                  'placeholder_sdf_shape': {'type': 'sphere', 'center': [100, 100, 100], 'radius': 1e-3,
                                                            'bsdf': {'type': 'diffuse'}}})
            integ = scene.integrator()
            integ.sdf = make_sdf()
            integ.warp_field = configs.get_config('warp').get_warpfield(integ.sdf)
            with dr.suspend_grad():
                img = mi.render(scene, sensor=sensor, seed=seed, spp=spp)                       # python/shape_opt.py:61-63
            out[f'img_{tag}'] = np.array(img)[..., :3].astype(np.float32)
            params = mi.traverse(scene)
            params.keep([SDF_DEFAULT_KEY, SDF_DEFAULT_KEY_P])
            dr.enable_grad(params[SDF_DEFAULT_KEY]); dr.enable_grad(params[SDF_DEFAULT_KEY_P])
            params.update()
            # the SAME samples for the primal image and the gradient pass: seed_grad = seed, spp_grad = spp (the in-repo tests
            # compare a gradient pass on given samples; python/shape_opt.py:78-80 uses independent ones)
            img = mi.render(scene, params=params, sensor=sensor, seed=seed, spp=spp, seed_grad=seed, spp_grad=spp)
            dr.backward(img * mi.TensorXf(c['grad_image']))
            out[f'grad_{tag}'] = np.array(dr.grad(params[SDF_DEFAULT_KEY])).reshape(c['grid'].shape).astype(np.float32)
            out[f'gradp_{tag}'] = np.array(dr.grad(params[SDF_DEFAULT_KEY_P])).reshape(3).astype(np.float32)

        fn = os.path.join(args.out, f'ref_{name}.npz')
        np.savez_compressed(fn, **out)
        print(fn, {k: (getattr(v, 'shape', None) or v) for k, v in out.items()})


if __name__ == '__main__':
    main()
