import os, sys, time, torch
sys.path.insert(0, 'differentiable-sdf-rendering_amd/python')
sys.path.insert(0, '.')
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev)
grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
def run(spp, n=6):
    for i in range(2): dsdf.render_forward(grid, sens[i], spp, seeds=[i])
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): dsdf.render_forward(grid, sens[i], spp, seeds=[10+i])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
gi = torch.randn(1,512,512,3,device=dev)*1e-6
def runb(spp, n=6):
    g = torch.zeros_like(data)
    for i in range(2): dsdf.render_backward(grid, sens[i], spp, gi, grad_grid=g, seeds=[i])
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): dsdf.render_backward(grid, sens[i], spp, gi, grad_grid=g, seeds=[10+i])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
print(os.environ.get('DSDF_LIB_PATH'), 'primal256 %.3f ms  primal64 %.3f ms  grad64 %.3f ms' % (run(256), run(64), runb(64)))
