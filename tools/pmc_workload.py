"""The workload of the rocprofv3 PMC passes: exactly ONE primal launch (256 spp) and ONE gradient-pass launch (64 spp)
of the bench scene (256^3, `--views` x 512^2), nothing else on the library's kernels -> per-dispatch counters are per
launch.  (The target render that bench.py needs is replaced by a fixed image gradient.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 12
spp_p, spp_g = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 64)
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)[:nv]
gi = torch.sign(torch.randn(nv, 512, 512, 3, device=dev)) / (512 * 512 * 3)
g = torch.zeros_like(data)
dsdf.render_forward(grid, sens, spp_p, seeds=list(range(nv)))
dsdf.render_backward(grid, sens, spp_g, gi, grad_grid=g, seeds=list(range(100, 100 + nv)))
torch.cuda.synchronize()
print('pmc workload done')
