"""The 4 / 1-spp step of bench.py's `low_spp` block alone (256^3, 12 views x 512^2, silhouette): times + a few steps for a kernel trace."""
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
g = torch.zeros_like(data)
sp, sg = int(os.environ.get('AB_SPP', 4)), int(os.environ.get('AB_SPP_GRAD', 1))
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)
out = {'tag': os.environ.get('AB_TAG', ''), 'spp': [sp, sg]}
out['primal'] = t(lambda: dsdf.render_forward(grid, sens, sp, seeds=S))
out['grad'] = t(lambda: dsdf.render_backward(grid, sens, sg, gi, grad_grid=g, seeds=S))
out['step'] = t(lambda: dsdf.render_step(grid, sens, sp, sg, lambda im: gi, g, S, [s + 100 for s in S]))
a = dsdf.render_forward(grid, sens, sp, seeds=S).double(); g.zero_()
dsdf.render_backward(grid, sens, sg, gi, grad_grid=g, seeds=S)
out['checksums'] = {'img': [float(a.sum()), float((a * a).sum())], 'grad': [float(g.double().abs().sum()), float((g.double() ** 2).sum())]}
if os.environ.get('AB_DUMP'):
    torch.save({'img': a.float().cpu(), 'grad': g.cpu()}, os.environ['AB_DUMP'])
if os.environ.get('AB_CMP'):          # the same outputs of another build: images to the last bit up to the order of the film's float adds
    o = torch.load(os.environ['AB_CMP'])
    out['vs'] = {'img_max_abs': float((a.float().cpu() - o['img']).abs().max()), 'img_differing': int((a.float().cpu() != o['img']).sum()),
                 'grad_rel_l2': float((g.cpu().double() - o['grad'].double()).norm() / o['grad'].double().norm())}
print('AB ' + json.dumps(out))
