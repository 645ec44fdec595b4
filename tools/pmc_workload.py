"""The workload of the rocprofv3 PMC passes: exactly ONE primal call (256 spp) and ONE gradient-pass call (64 spp)
of the bench scene (256^3, `--views` x 512^2), nothing else on the library's non-STATS kernels -> per-kernel counter sums are
per library call.  (The target render that bench.py needs is replaced by a fixed image gradient.)  Afterwards the same two
calls run once more with the library's own statistics (the <..., STATS = true> instantiations: other kernel names, so they
do not mix into the counters) and the wave-iteration counts are written to $PMC_STATS_OUT (json): together with
SQ_INSTS_VALU of the profiled call they give the VALU instructions per lock-step wave iteration that bench.py's roofline
uses (profiles/summarize_pmc.py -> profiles/valu_model.json)."""
import json, os, sys, torch
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
out = os.environ.get('PMC_STATS_OUT')
if out:
    sp, sg = dsdf.new_stats(dev), dsdf.new_stats(dev)
    dsdf.render_forward(grid, sens, spp_p, seeds=list(range(nv)), stats=sp)
    dsdf.render_backward(grid, sens, spp_g, gi, grad_grid=g, seeds=list(range(100, 100 + nv)), stats=sg)
    torch.cuda.synchronize()
    json.dump({'views': nv, 'spp': [spp_p, spp_g], 'primal': dsdf.stats_dict(sp), 'grad': dsdf.stats_dict(sg)}, open(out, 'w'))
print('pmc workload done')
