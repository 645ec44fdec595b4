"""Times the two launches of the C5-shaped workload (sdf_direct_reparam, 12 views, 256^3 + 256^3 x 3 albedo, 512^2):
A/B runs of kernel variants via DSDF_LIB_PATH."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
gi = torch.randn(12, 512, 512, 3, device=dev) * 1e-6
alb = torch.rand(256, 256, 256, 3, device=dev) * 0.6 + 0.2
galb = torch.zeros_like(alb); g = torch.zeros_like(data)
sh = dsdf.Shading(alb, 1.0, hide_emitters=True)
def t(fn, n=2):
    fn(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
p = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=list(range(12)), integrator='sdf_direct_reparam', shading=sh))
b = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=list(range(12)), integrator='sdf_direct_reparam', shading=sh, grad_albedo=galb))
print(os.path.basename(os.environ.get('DSDF_LIB_PATH', 'default')), 'direct primal256x12 %.2f ms   gradpass64x12 %.2f ms' % (p, b))
