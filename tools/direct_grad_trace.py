"""One gradient call of sdf_direct_reparam at C5 sizes (for a kernel trace): bash tools/gpu_run.sh trace <tag> <spec> with AB_SCRIPT=tools/direct_grad_trace.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
SG = [s + 100 for s in range(12)]
torch.manual_seed(0)
albedo = torch.rand(256, 256, 256, 3, device=dev) * 0.6 + 0.2
sh = dsdf.Shading(albedo, 1.0, hide_emitters=False)
gi = torch.sin(torch.arange(12 * 512 * 512 * 3, device=dev, dtype=torch.float32)).reshape(12, 512, 512, 3) * 1e-6
g = torch.zeros_like(data); ga = torch.zeros_like(albedo)
st = dsdf.new_stats(dev)
dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=SG, grad_albedo=ga, integrator='sdf_direct_reparam', shading=sh, stats=st)
torch.cuda.synchronize()
import json
print('GSTATS', json.dumps({k: v for k, v in dsdf.stats_dict(st).items() if not isinstance(v, dict)}))
