"""A/B of one build / one setting (DSDF_LIB_PATH, DSDF_GROUPS, DSDF_TAIL_STREAMS ...) on the bench workload (256^3, 12 views x
512^2, silhouette, 256 / 64 spp): launch times of the primal render, the gradient pass and the two-stream step of bench.py,
plus checksums of the image and of dL/dsdf (two settings that compute the same samples agree to the order of the float
atomics).  Prints one line `AB {json}`.  `--shade` adds the simple-shading integrator, `--low` the 4/1-spp step."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
NV = int(os.environ.get('AB_VIEWS', 12))                 # (AB_VIEWS=n: the first n views of the ring -- a rank's shard of a multi-GPU run)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)[:NV]
S = list(range(NV))
gi = torch.sin(torch.arange(NV * 512 * 512 * 3, device=dev, dtype=torch.float32)).reshape(NV, 512, 512, 3) * 1e-6
g = torch.zeros_like(data)


def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)


out = {'tag': os.environ.get('AB_TAG', ''), 'lib': os.path.basename(os.environ.get('DSDF_LIB_PATH', 'libdsdf.so')),
       'env': {k: v for k, v in os.environ.items() if k.startswith('DSDF_') and k != 'DSDF_LIB_PATH'}}
out['primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S))
out['grad64'] = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S))
out['step'] = t(lambda: dsdf.render_step(grid, sens, 256, 64, lambda im: gi, g, S, [s + 100 for s in S]), 8)
if '--shade' in sys.argv:
    out['shade_primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, integrator=1), 3)
    out['shade_grad64'] = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S, integrator=1), 3)
if '--low' in sys.argv:
    out['step_4_1'] = t(lambda: dsdf.render_step(grid, sens, 4, 1, lambda im: gi, g, S, [s + 100 for s in S]), 20)
st = dsdf.new_stats(dev)
a = dsdf.render_forward(grid, sens, 256, seeds=S, stats=st).double()
cs = {'img256': [float(a.sum()), float((a * a).sum())]}
sd = dsdf.stats_dict(st)
out['primal_stats'] = {k: sd[k] for k in ('lanes', 'steps', 'hits', 'wave_steps', 'tail_rays', 'tail_steps', 'tail_wave_steps', 'tail_waves') if k in sd}
g.zero_()
st = dsdf.new_stats(dev)
dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S, stats=st)
sd = dsdf.stats_dict(st)
out['grad_stats'] = {k: sd[k] for k in ('lanes', 'steps', 'wave_steps', 'tail_rays', 'tail_steps', 'tail_wave_steps', 'tail_waves') if k in sd}
cs['grad64'] = [float(g.double().abs().sum()), float((g.double() ** 2).sum())]
if '--shade' in sys.argv:
    st = dsdf.new_stats(dev)
    a = dsdf.render_forward(grid, sens, 256, seeds=S, integrator=1, stats=st).double()
    cs['img256_shade'] = [float(a.sum()), float((a * a).sum())]
    sd = dsdf.stats_dict(st)
    out['shade_stats'] = {k: sd[k] for k in ('lanes', 'steps', 'hits', 'refine_steps', 'wave_steps', 'tail_rays') if k in sd}
    g.zero_()
    dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=S, integrator=1)
    cs['grad64_shade'] = [float(g.double().abs().sum()), float((g.double() ** 2).sum())]
g.zero_()
dsdf.render_step(grid, sens, 256, 64, lambda im: gi, g, S, [s + 100 for s in S])
torch.cuda.synchronize()
cs['step_grad'] = [float(g.double().abs().sum()), float((g.double() ** 2).sum())]
out['checksums'] = cs
print('AB ' + json.dumps(out))
