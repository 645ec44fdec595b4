#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the fp64 oracle (oracle/sdf_oracle.py).

The reference itself ships no golden vectors and cannot be run here (Mitsuba /
Dr.Jit absent), so these fixtures pin the ORACLE (and through it the HIP path)
against silent regressions; they are not outputs of the reference.
Run:  python tests/golden/make_golden.py            (all fixtures)
      python tests/golden/make_golden.py principled (only that section)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)

import sdf_oracle as O
from cases import direct_inputs, make_case, oracle_backward, oracle_forward, oracle_direct

ONLY = sys.argv[1] if len(sys.argv) > 1 else None
for name in (() if ONLY else ('sphere16', 'blob32')):
    case = make_case(name)
    out = {}
    for integ, tag in ((O.SILHOUETTE, 'sil'), (O.SIMPLE_SHADING, 'shade')):
        img, aux = oracle_forward(case, integ)
        out[f'img_{tag}'] = img.numpy().astype(np.float32)
        out[f'grad_{tag}'] = oracle_backward(case, integ).numpy().astype(np.float32)
        out[f'steps_{tag}'] = np.int64(aux['steps'])
        out[f'hits_{tag}'] = np.int64(aux['hits'])
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(name, {k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items()})

# sdf_direct_reparam (BSDF / emitter: this repo's spec, oracle/sdf_oracle.py header)
for name in (() if ONLY else ('sphere16', 'blob32')):
    case = make_case(name)
    ex = direct_inputs(case)
    img, gd, ga = oracle_direct(case, ex, reparam=True, grads=True)
    out = {'img': img.numpy().astype(np.float32), 'grad_data': gd.numpy().astype(np.float32),
           'grad_albedo': ga.numpy().astype(np.float32),
           'img_hidden': oracle_direct(case, ex, reparam=False, hide_emitters=True).numpy().astype(np.float32)}
    np.savez_compressed(os.path.join(HERE, f'{name}_direct.npz'), **out)
    print(name + '_direct', {k: v.shape for k, v in out.items()})

# sdf_direct_reparam with use_mis (BSDF sampling + power heuristic) and with decouple_reparam / detach_indirect_si:
# torch-autograd oracle, blob32 (the case on which the shadow-ray warp is active)
import torch
if not ONLY:
    case = make_case('blob32')
    ex = direct_inputs(case)
    bu = torch.rand(case['offsets'].shape[0], 2, generator=torch.Generator().manual_seed(3), dtype=torch.float32)
    out = {'bsdf_u': bu.numpy()}
    for tag, kw in (('mis', dict(use_mis=True, bsdf_u=bu.double())), ('detach', dict(detach_indirect_si=True)),
                    ('decouple', dict(decouple_reparam=True)), ('mis_decouple', dict(use_mis=True, bsdf_u=bu.double(), decouple_reparam=True))):
        data = case['grid'].clone().requires_grad_(True)
        alb = ex['albedo'].double().clone().requires_grad_(True)
        img = O.render(O.Grid3d(data), case['cam'], case['W'], case['H'], case['spp'], case['offsets'].double(), O.DIRECT, True, albedo=alb,
                       emitter_u=ex['emitter_u'].double(), env=torch.tensor(ex['env'], dtype=torch.float64), **kw)
        gd, ga = torch.autograd.grad((img * case['grad_image'].double()).sum(), (data, alb))
        out[f'img_{tag}'] = img.detach().numpy().astype(np.float32)
        out[f'grad_data_{tag}'] = gd.numpy().astype(np.float32)
        out[f'grad_albedo_{tag}'] = ga.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'blob32_direct_variants.npz'), **out)
    print('blob32_direct_variants', {k: v.shape for k, v in out.items()})

# mesh -> SDF (oracle/mesh_oracle.py + the C redistancing oracle): a box and an icosphere at 16^3, per-ray casts of the box
if not ONLY:
    import c_oracle
    import mesh_oracle as M

    lib = c_oracle.load()
    out = {}
    for tag, (v, f) in (('box', M.box(half=(0.3, 0.2, 0.25))), ('ico', M.icosphere(0.3, 1, centre=(0.02, -0.01, 0.03)))):
        out[f'{tag}_tri'] = v[f]
        out[f'{tag}_sdf'] = M.create_sdf(v[f], 16, lambda p: c_oracle.redistance(lib, p)).astype(np.float32)
        out[f'{tag}_sdf_coarse'] = M.create_sdf(v[f], 16, lambda p: c_oracle.redistance(lib, p), refine_surface=False).astype(np.float32)
    rng = np.random.default_rng(5)
    o = rng.uniform(-0.5, 0.5, (512, 3)).astype(np.float32)
    d = rng.normal(size=(512, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    t, back, margin = M.raycast(out['box_tri'], o, d)
    out.update(ray_o=o, ray_d=d, ray_t=t, ray_back=back, ray_margin=margin)
    np.savez_compressed(os.path.join(HERE, 'mesh16.npz'), **out)
    print('mesh16', {k: v.shape for k, v in out.items()})


# sdf_direct_reparam with the `principled` BSDF (base_color + roughness volumes; the plugin restated from memory, unpinned):
# torch-autograd oracle, blob32
if len(sys.argv) < 2 or sys.argv[1] == 'principled':
    from test_principled_host import _oracle, _principled_inputs
    case = make_case('blob32')
    ex = _principled_inputs(case)
    img, gd, ga, gr = _oracle(case, ex, torch.float64)
    out = {'roughness': ex['roughness'].numpy(), 'img': img.numpy().astype(np.float32), 'grad_data': gd.numpy().astype(np.float32),
           'grad_base_color': ga.numpy().astype(np.float32), 'grad_roughness': gr.numpy().astype(np.float32),
           'img_noreparam': _oracle(case, ex, torch.float64, reparam=False, grads=False).numpy().astype(np.float32)}
    np.savez_compressed(os.path.join(HERE, 'blob32_principled.npz'), **out)
    print('blob32_principled', {k: v.shape for k, v in out.items()})
