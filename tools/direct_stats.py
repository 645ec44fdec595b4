import json, os, sys, torch
ROOT = '/root/repo'
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
S = list(range(12))
torch.manual_seed(0)
albedo = torch.rand(256, 256, 256, 3, device=dev) * 0.6 + 0.2
sh = dsdf.Shading(albedo, 1.0, hide_emitters=False)
st = dsdf.new_stats(dev)
dsdf.render_forward(grid, sens, 256, seeds=S, integrator='sdf_direct_reparam', shading=sh, stats=st)
torch.cuda.synchronize()
print('DSTATS', json.dumps({k: v for k, v in dsdf.stats_dict(st).items() if not isinstance(v, dict)}))
