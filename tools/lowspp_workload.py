"""4/1-spp step of the bench scene, a few times (for rocprofv3 --kernel-trace / --pmc)."""
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
S = list(range(12))
for _ in range(4):
    dsdf.render_forward(grid, sens, 4, seeds=S)
    dsdf.render_backward(grid, sens, 1, gi, grad_grid=g, seeds=S)
torch.cuda.synchronize()
