"""A/B timing of sdf_direct_reparam at BASELINE configs[4] sizes (256^3 SDF + 256^3 x 3 albedo, 12 views x 512^2, 256 / 64 spp) for one
build (DSDF_LIB_PATH): primal call, gradient call, two-stream step, checksums of image / dL/dsdf / dL/d(albedo).  Prints `AB {json}`."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
S = list(range(12)); SG = [s + 100 for s in S]
torch.manual_seed(0)
albedo = torch.rand(256, 256, 256, 3, device=dev) * 0.6 + 0.2
sh = dsdf.Shading(albedo, 1.0, hide_emitters=False)
gi = torch.sin(torch.arange(12 * 512 * 512 * 3, device=dev, dtype=torch.float32)).reshape(12, 512, 512, 3) * 1e-6
g = torch.zeros_like(data); ga = torch.zeros_like(albedo)
kw = dict(integrator='sdf_direct_reparam', shading=sh)


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)


out = {'tag': os.environ.get('AB_TAG', ''), 'lib': os.path.basename(os.environ.get('DSDF_LIB_PATH', 'libdsdf.so'))}
out['primal256'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, **kw))
out['grad64'] = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=SG, grad_albedo=ga, **kw))
out['step'] = t(lambda: dsdf.render_step(grid, sens, 256, 64, lambda im: gi, g, S, SG, grad_albedo=ga, **kw))
g.zero_(); ga.zero_()
img = dsdf.render_forward(grid, sens, 256, seeds=S, **kw)
dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=SG, grad_albedo=ga, **kw)
torch.cuda.synchronize()
out['checksums'] = {'img': float(img.double().sum()), 'grad': [float(g.double().abs().sum()), float((g.double() ** 2).sum())],
                    'grad_albedo': [float(ga.double().abs().sum()), float((ga.double() ** 2).sum())]}
print('AB ' + json.dumps(out))
