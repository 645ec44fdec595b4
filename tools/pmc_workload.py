"""The workload of the rocprofv3 PMC passes: a FIXED sequence of library calls on the bench scene (256^3, 12 views x 512^2), one
of each kind, so that the n-th dispatch of a kernel in the counter CSV belongs to a known call (profiles/summarize_pmc.py reads the
manifest this script writes next to the statistics):

  0  primal 256 spp                      dsdf_render_forward                 (all proofs: the shipped path)
  1  primal 256 spp, hit proof off       dsdf_render_forward, DSDF_NO_HIT_PROOF   -> with call 0: VALU = a * wave iterations + b * chunks
  2  gradient pass 64 spp                dsdf_render_backward                (sweep, tail, fused k_backward)
  3  the two-stream step 256 / 64        dsdf.render_step                    (sweep + k_backward_coef | primal ... k_backward_apply)
  4  (--low)    the 4 / 1-spp step       dsdf.render_step                    (k_render_pass: the general pass)
  5  (--direct) sdf_direct_reparam 256 / 64 at BASELINE configs[4] sizes (256^3 x 3 albedo): primal + gradient pass

Afterwards the calls 0, 1, 2 (and the primal / gradient calls of 4, 5) run once more with the library's own statistics (the
<..., STATS = true> instantiations: other kernel names, so they do not mix into the counters) and the wave-iteration / chunk counts
go to $PMC_STATS_OUT (json).  (The target render that bench.py needs is replaced by a fixed image gradient.)"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
nv = 12
low, direct = '--low' in sys.argv, '--direct' in sys.argv
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)[:nv]
gi = torch.sign(torch.randn(nv, 512, 512, 3, device=dev)) / (512 * 512 * 3)
g = torch.zeros_like(data)
S, SG = list(range(nv)), list(range(100, 100 + nv))
calls = []


def call(tag, fn):
    calls.append(tag)
    fn()
    torch.cuda.synchronize()


call('primal', lambda: dsdf.render_forward(grid, sens, 256, seeds=S))
call('primal_no_hit_proof', lambda: dsdf.render_forward(grid, sens, 256, seeds=S, empty_space_skip='empty-only'))
call('grad', lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=SG))
call('step', lambda: dsdf.render_step(grid, sens, 256, 64, lambda im: gi, g, S, SG))
if low:
    call('step_4_1', lambda: dsdf.render_step(grid, sens, 4, 1, lambda im: gi, g, S, SG))
if direct:
    albedo = torch.rand(256, 256, 256, 3, device=dev) * 0.6 + 0.2
    sh = dsdf.Shading(albedo, 1.0, hide_emitters=False)
    galb = torch.zeros_like(albedo)
    call('direct_primal', lambda: dsdf.render_forward(grid, sens, 256, seeds=S, integrator='sdf_direct_reparam', shading=sh))
    call('direct_grad', lambda: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=SG, integrator='sdf_direct_reparam', shading=sh,
                                                     grad_albedo=galb))
out = os.environ.get('PMC_STATS_OUT')
if out:
    def stats_of(fn):
        st = dsdf.new_stats(dev)
        fn(st)
        torch.cuda.synchronize()
        return dsdf.stats_dict(st)
    res = {'views': nv, 'spp': [256, 64], 'calls': calls,
           'primal': stats_of(lambda st: dsdf.render_forward(grid, sens, 256, seeds=S, stats=st)),
           'primal_no_hit_proof': stats_of(lambda st: dsdf.render_forward(grid, sens, 256, seeds=S, stats=st, empty_space_skip='empty-only')),
           'grad': stats_of(lambda st: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=SG, stats=st))}
    if low:
        res['low_primal'] = stats_of(lambda st: dsdf.render_forward(grid, sens, 4, seeds=S, stats=st))
        res['low_grad'] = stats_of(lambda st: dsdf.render_backward(grid, sens, 1, gi, grad_grid=g, seeds=SG, stats=st))
    if direct:
        res['direct_primal'] = stats_of(lambda st: dsdf.render_forward(grid, sens, 256, seeds=S, integrator='sdf_direct_reparam', shading=sh, stats=st))
        res['direct_grad'] = stats_of(lambda st: dsdf.render_backward(grid, sens, 64, gi, grad_grid=g, seeds=SG, integrator='sdf_direct_reparam',
                                                                     shading=sh, grad_albedo=galb, stats=st))
    json.dump(res, open(out, 'w'))
print('pmc workload done')
