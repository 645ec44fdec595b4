"""A/B of one build (DSDF_LIB_PATH): launch times + checksums of every output on the bench scene (image sums of both
integrators at 256 / 64 / 4 spp, |dL/dsdf| at 64 / 1 spp) -- two builds that compute the same cells give the same sums up to
the order of the float atomics."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
S = list(range(12))
gi = torch.sin(torch.arange(12 * 512 * 512 * 3, device=dev, dtype=torch.float32)).reshape(12, 512, 512, 3) * 1e-6
def t(fn, n=4):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)
out = {'lib': os.path.basename(os.environ.get('DSDF_LIB_PATH', 'libdsdf.so'))}
out['primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S))
g = torch.zeros_like(data)
out['grad64'] = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S))
out['primal4'] = t(lambda: dsdf.render_forward(grid, sens, 4, seeds=S), 10)
cs = {}
for spp in (256, 64, 4):
    for integ in (0, 1):
        a = dsdf.render_forward(grid, sens, spp, seeds=S, integrator=integ).double()
        cs[f'img{spp}_{integ}'] = [float(a.sum()), float((a * a).sum())]
for spp in (64, 1):
    for integ in (0, 1):
        g = torch.zeros_like(data)
        dsdf.render_backward(grid, sens, spp, gi, grad_grid=g, seeds=S, integrator=integ)
        cs[f'grad{spp}_{integ}'] = [float(g.double().abs().sum()), float((g.double() ** 2).sum())]
out['checksums'] = cs
print('AB ' + json.dumps(out))
