"""Times the gradient pass of the bench workload only (A/B of backward-kernel variants via DSDF_LIB_PATH)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
gi = torch.randn(12, 512, 512, 3, device=dev) * 1e-6
g = torch.zeros_like(data)
fn = lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=list(range(12)))
fn(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); [fn() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get('DSDF_LIB_PATH', 'default')), 'gradpass64x12 %.2f ms' % (e0.elapsed_time(e1) / 3))
