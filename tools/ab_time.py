"""A/B timing of one build of libdsdf.so (DSDF_LIB_PATH) on the bench workload (256^3, 12 views x 512^2): the two
256/64-spp launches, the 4/1-spp pair, and the same launches over an EMPTY grid (every pixel proven empty: what the
launch + skipped waves alone cost).  Prints one JSON line.  `--direct` adds the sdf_direct_reparam pair."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
empty = dsdf.SdfGrid(torch.full((256, 256, 256), 0.5, device=dev))
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
gi = torch.randn(12, 512, 512, 3, device=dev) * 1e-6
g = torch.zeros_like(data)


def t(fn, n=3):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)


S = list(range(12))
out = {'lib': os.path.basename(os.environ.get('DSDF_LIB_PATH', 'libdsdf.so'))}
out['primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S))
out['grad64'] = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S))
out['primal4'] = t(lambda: dsdf.render_forward(grid, sens, 4, seeds=S), 10)
out['grad1'] = t(lambda: dsdf.render_backward(grid, sens, 1, gi, grad_grid=g, seeds=S), 10)
out['shade_primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, integrator=1))
out['shade_grad64'] = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S, integrator=1))
out['empty_primal256'] = t(lambda: dsdf.render_forward(empty, sens, 256, seeds=S))
out['empty_grad64'] = t(lambda: dsdf.render_backward(empty, sens, 64, gi, grad_grid=g, seeds=S))
out['noskip_primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, empty_space_skip=False), 2)
if '--direct' in sys.argv:
    alb = torch.rand(256, 256, 256, 3, device=dev) * 0.6 + 0.2
    galb = torch.zeros_like(alb)
    sh = dsdf.Shading(alb, 1.0, hide_emitters=True)
    out['direct_primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, integrator='sdf_direct_reparam', shading=sh), 2)
    out['direct_grad64'] = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S, integrator='sdf_direct_reparam',
                                                          shading=sh, grad_albedo=galb), 2)
print('AB ' + json.dumps(out))
