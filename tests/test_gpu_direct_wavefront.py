"""The wavefront form of sdf_direct_reparam (primary march into per-sample records, compacted shadow queue, streaming
shadow rays, shading pass) computes the SAME samples as the fused worker it replaced: same rays, same steps, same values.  The switch
(DSDF_DIRECT_WAVEFRONT) is read once per process, so each form runs in a process of its own on the same seeded inputs; images and
gradients must agree to the order of the film / gradient atomics, the ray and step counters exactly."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(sys.argv[1], 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, os.path.join(sys.argv[1], 'oracle'))
import dsdf, sdf_oracle as O
dev = torch.device('cuda')
data = O.blob_grid(64, n=8, seed=5).float().to(dev)
grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(3, resx=96, resy=80)[:2]
g = torch.Generator().manual_seed(7)
albedo = (torch.rand(24, 20, 16, 3, generator=g) * 0.6 + 0.2).to(dev)
gi = torch.randn(2, 80, 96, 3, generator=g).to(dev) * 1e-3
for hide in (False, True):
    sh = dsdf.Shading(albedo, (1.0, 0.9, 0.8), hide_emitters=hide)
    st = dsdf.new_stats(dev)
    img = dsdf.render_forward(grid, sens, 256, seeds=[3, 4], integrator='sdf_direct_reparam', shading=sh, stats=st)
    sd = dsdf.stats_dict(st)
    gg = torch.zeros_like(data); ga = torch.zeros_like(albedo)
    st2 = dsdf.new_stats(dev)
    dsdf.render_backward(grid, sens, 64, gi, grad_grid=gg, seeds=[13, 14], integrator='sdf_direct_reparam', shading=sh, grad_albedo=ga, stats=st2)
    sd2 = dsdf.stats_dict(st2)
    torch.cuda.synchronize()
    np.savez(sys.argv[2] + ('_hide' if hide else '') + '.npz', img=img.cpu().numpy(), gg=gg.cpu().numpy(), ga=ga.cpu().numpy(),
             counts=np.array([sd['lanes'], sd['tail_steps'], sd['tail_rays'], sd2['lanes'], sd2['tail_steps'], sd2['tail_rays']], dtype=np.int64))
'''


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def test_wavefront_equals_fused_worker(built, tmp_path):
    script = tmp_path / 'child.py'
    script.write_text(CHILD)
    for tag, mode in (('fused', '0'), ('wave', '2')):
        env = dict(os.environ, DSDF_DIRECT_WAVEFRONT=mode)
        r = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / tag)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
    for suffix in ('', '_hide'):
        a, b = np.load(tmp_path / f'fused{suffix}.npz'), np.load(tmp_path / f'wave{suffix}.npz')
        # the same samples and the same shadow rays, step for step: generated lanes, shadow-ray steps, shadow rays -- of the primal and
        # of the sweep (slots 8..10 of the statistics count the shadow rays of this integrator in both forms; the primary rays' own
        # counters differ by what the wavefront's tail kernel marches, which it does not report)
        assert (a['counts'] == b['counts']).all(), (a['counts'], b['counts'])
        assert a['counts'][2] > 10000 and a['counts'][5] > 2000
        assert rel(b['img'], a['img']) < 2e-6
        assert rel(b['gg'], a['gg']) < 2e-5 and rel(b['ga'], a['ga']) < 2e-5
