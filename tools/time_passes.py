"""Times the two launches of the bench workload (12 views, 256^3, 512^2): used for A/B runs of kernel variants
(set DSDF_LIB_PATH to another build of libdsdf.so)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
gi = torch.randn(12,512,512,3,device=dev)*1e-6
def t(fn, n=3):
    fn(); fn(); torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
g = torch.zeros_like(data)
p = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=list(range(12))))
b = t(lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=list(range(12))))
print(os.path.basename(os.environ.get('DSDF_LIB_PATH','default')), 'primal256x12 %.2f ms   gradpass64x12 %.2f ms' % (p, b))
