"""A/B timing of the 256-spp primal launch of one build (DSDF_LIB_PATH) + a checksum of its image."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
S = list(range(12))
def t(fn, n=4):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)
out = {'lib': os.path.basename(os.environ.get('DSDF_LIB_PATH', 'libdsdf.so'))}
out['primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S))
out['shade256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, integrator=1))
out['primal64'] = t(lambda: dsdf.render_forward(grid, sens, 64, seeds=S))
a = dsdf.render_forward(grid, sens, 256, seeds=S); out['sum'] = float(a.double().sum()); out['sumsq'] = float((a.double() ** 2).sum())
b = dsdf.render_forward(grid, sens, 64, seeds=S, integrator=1); out['sum_shade'] = float(b.double().sum())
print('AB ' + json.dumps(out))
