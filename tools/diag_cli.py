#!/usr/bin/env python3
"""Why does `optimize.py sphere --optconfig principled-6` stall at 70 % of its initial loss?

Runs the textured CLI configurations on the GPU box with (a) the procedural target of tests (a smooth colour FIELD, which the
1^3 base-colour volume of a 40-iteration run -- sdf_res 32 // 2^5 texture upsamplings still to come -- cannot represent) and
(b) a target the optimised volumes CAN represent (constant colour, roughness 0.4), and prints the loss curve, the per-key
learning rates, the recovered colour / roughness and the loss of the TARGET parameters themselves rendered at the optimiser's
spp against the stored references (= the Monte-Carlo floor of the L1 loss).  Output: one `DIAG {json}` line per run.
    python tools/diag_cli.py [--iters 40 80]"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python'))
sys.path.insert(0, ROOT)


def make_scene(scene_dir, name, colour):
    """scenes/<name>/<name>.vol (sphere of radius 0.36) + a constant albedo volume."""
    import util
    d = os.path.join(scene_dir, name)
    os.makedirs(d, exist_ok=True)
    lin = torch.linspace(0, 1, 128)
    z, y, x = torch.meshgrid(lin, lin, lin, indexing='ij')
    util.write_vol(os.path.join(d, f'{name}.vol'), torch.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) - 0.36)
    util.write_vol(os.path.join(d, f'{name}-albedo.vol'), torch.tensor(colour).view(1, 1, 1, 3).expand(2, 2, 2, 3).contiguous())


def run(scene, optconfig, n_iter, tmp, extra=()):
    import optimize
    import scenes
    import util
    out = os.path.join(tmp, f'out_{scene}_{optconfig}_{n_iter}')
    optimize.RENDER_DIR = os.path.join(tmp, f'renders_{scene}_{optconfig}_{n_iter}')
    args = [scene, '--optconfig', optconfig, '--configs', 'warp', '--outputdir', out, '--refspp', '128', f'--n_iter={n_iter}',
            '--spp=64', '--sdf_res=32', '--resx=48', '--resy=48'] + list(extra)
    optimize.main(args)
    o = os.path.join(out, scene, optconfig, 'warp')
    lv = json.load(open(os.path.join(o, 'metadata.json')))['loss_values']
    rec = dict(scene=scene, optconfig=optconfig, n_iter=n_iter, first3=float(np.mean(lv[:3])), last5=float(np.mean(lv[-5:])),
               ratio=float(np.mean(lv[-5:]) / np.mean(lv[:3])), curve=[round(v, 5) for v in lv])
    for key in ('base_color', 'reflectance', 'roughness'):
        fn = os.path.join(o, 'params', f'main-bsdf-{key}-volume-data-final.vol')
        if os.path.isfile(fn):
            v = util.read_vol(fn)
            rec[key] = dict(shape=list(v.shape), mean=[round(float(m), 4) for m in v.reshape(-1, v.shape[-1] if v.dim() == 4 else 1).mean(0)],
                            std=round(float(v.std()), 5) if v.numel() > 1 else 0.0)
    sdf = util.read_vol(os.path.join(o, 'params', 'sdf-data-final.vol'))
    rec['sdf_shape'] = list(sdf.shape)
    print('DIAG ' + json.dumps(rec), flush=True)
    return rec


if __name__ == '__main__':
    iters = [int(a) for a in sys.argv[sys.argv.index('--iters') + 1:]] if '--iters' in sys.argv else [40, 80]
    import scenes
    tmp = tempfile.mkdtemp(prefix='diag_cli_')
    scenes.SCENE_DIR = os.path.join(tmp, 'scenes')
    make_scene(scenes.SCENE_DIR, 'ball', [0.7, 0.35, 0.2])
    for n in iters:
        for cfg in ('principled-6', 'diffuse-6'):
            run('sphere', cfg, n, tmp)        # procedural colour field: not representable by the 1^3 volume of a short run
            run('ball', cfg, n, tmp)          # constant colour: representable
