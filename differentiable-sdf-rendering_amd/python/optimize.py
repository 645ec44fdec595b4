#!/usr/bin/env python3
"""Reconstructs an object as an SDF (python/optimize.py): `python optimize.py <scene...> --optconfigs <cfg...>`.

Same flags as the reference (argparse prefix matching keeps `--optconfig` working); `--llvm` is
accepted and ignored (there is one backend: HIP on the MI355X).  Reference images come from the
scene's target SDF (scenes.py) rendered with the method's integrator at `--refspp`."""
import argparse
import os
import sys
from os.path import join

import torch

from constants import OUTPUT_DIR, RENDER_DIR


def render_reference_images(scene_config, config, ref_spp=1024, force=False, verbose=False, mts_args=None):
    """python/optimize.py:11-29."""
    from integrators.reparam import Scene, create_integrator, render
    from scenes import load_target_albedo, load_target_sdf
    from shapes import Grid3d
    from util import set_sensor_res, write_image
    folder = join(RENDER_DIR, scene_config.scene, scene_config.name, config.integrator, 'ref')
    os.makedirs(folder, exist_ok=True)
    scene = None
    for idx, sensor in enumerate(scene_config.sensors):
        set_sensor_res(sensor, (scene_config.resx, scene_config.resy))
        fn = join(folder, f'ref-{idx:02d}.npy')
        if os.path.isfile(fn) and not force:
            if verbose:
                print(f'File exists, not rendering of {fn}')
            continue
        if scene is None:
            target = load_target_sdf(scene_config.scene, res=max(128, 2 * 64))
            props = {'sdf': Grid3d(target)}
            if config.integrator == 'sdf_direct_reparam':
                if any(k.endswith('roughness.volume.data') for k in scene_config.param_keys):    # principled-* configs
                    props['base_color'] = load_target_albedo(scene_config.scene)
                    props['roughness'] = 0.4
                else:
                    props['reflectance'] = load_target_albedo(scene_config.scene)
            scene = Scene(scene_config.sensors, create_integrator(config.integrator, props))
        with torch.no_grad():
            # 64-sample waves: round the reference spp up to a multiple of 64
            img = render(scene, sensor=sensor, seed=idx + 41, spp=((ref_spp + 63) // 64) * 64)
        write_image(fn, img)


def copy_reference_images_to_output_dir(scene_config, config, output_dir):
    from shutil import copyfile
    folder = join(RENDER_DIR, scene_config.scene, scene_config.name, config.integrator, 'ref')
    paths = []
    for idx in range(len(scene_config.sensors)):
        dst = join(output_dir, f'ref-{idx:02d}.npy')
        copyfile(join(folder, f'ref-{idx:02d}.npy'), dst)
        paths.append(dst)
    return paths


def optimize(scene_name, config, opt_name, output_dir, ref_spp=1024, force=False, verbose=False, opt_config_args=None):
    from opt_configs import get_opt_config
    from shape_opt import optimize_shape
    cur = join(output_dir, scene_name, opt_name, config.name)
    os.makedirs(cur, exist_ok=True)
    opt_config, mts_args = get_opt_config(opt_name, opt_config_args)
    opt_config.scene = scene_name
    render_reference_images(opt_config, config, ref_spp=ref_spp, force=force, verbose=verbose, mts_args=mts_args)
    refs = copy_reference_images_to_output_dir(opt_config, config, cur)
    return optimize_shape(opt_config, mts_args, refs, cur, config)


def main(args):
    parser = argparse.ArgumentParser(description='Reconstructs an object as an SDF')
    parser.add_argument('scenes', default=None, nargs='*', help='Reference scenes (target shapes) to optimize')
    parser.add_argument('--optconfigs', '--opt', nargs='+', help='Optimization configurations to run')
    parser.add_argument('--outputdir', default=OUTPUT_DIR, help='Output directory. Default: ../outputs')
    parser.add_argument('--configs', default=['warp'], type=str, nargs='*', help='Method(s) to use. Default: warp')
    parser.add_argument('--force', action='store_true', help='Force rendering of reference images')
    parser.add_argument('--llvm', action='store_true', help='Accepted for compatibility; ignored (single HIP backend)')
    parser.add_argument('--refspp', type=int, default=2048, help='Samples per pixel for reference images. Default: 2048')
    parser.add_argument('--verbose', action='store_true', help='Print additional log information')
    parser.add_argument('--print_params', '-pp', action='store_true', help='Print the parameters of the scene and exit.')
    args, uargs = parser.parse_known_args(args)

    from configs import apply_cmdline_args, get_config
    from opt_configs import get_opt_config, is_valid_opt_config
    if args.optconfigs is None:
        raise ValueError('Must at least specify one opt. config!')
    if any(not is_valid_opt_config(o) for o in args.optconfigs):
        raise ValueError(f'Unknown opt config detected: {args.optconfigs}')
    for scene_name in args.scenes:
        for config_name in args.configs:
            for opt_config in args.optconfigs:
                config = get_config(config_name)
                remaining = apply_cmdline_args(config, uargs, return_dict=True)
                if args.print_params:
                    oc, mts_args = get_opt_config(opt_config, remaining)
                    print(f'Scene arguments: {mts_args}')
                    print('Parameters: ', oc.param_keys)
                    continue
                optimize(scene_name, config, opt_config, args.outputdir, args.refspp, args.force, args.verbose, remaining)


if __name__ == '__main__':
    main(sys.argv[1:])
