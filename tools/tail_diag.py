"""Where are the tail waves of the two-stream step in time?  (dsdf.tail_stats_arm: start / end of the earliest and the latest wave of
k_tail_trace_plain and k_tail_trace_diff on the 100 MHz wall clock.)  Bench workload: 256^3, 12 views x 512^2, 256 / 64 spp."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
S = list(range(12))
gi = torch.sin(torch.arange(12 * 512 * 512 * 3, device=dev, dtype=torch.float32)).reshape(12, 512, 512, 3) * 1e-6
g = torch.zeros_like(data)
step = lambda: dsdf.render_step(grid, sens, 256, 64, lambda im: gi, g, S, [s + 100 for s in S])
for _ in range(3): step()
torch.cuda.synchronize()
for rep in range(2):
    st = dsdf.new_stats(dev)
    dsdf.tail_stats_arm(st)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); step(); e1.record(); torch.cuda.synchronize()
    dsdf.tail_stats_arm(None)
    tw = dsdf.stats_dict(st)['tail_waves']
    p, s = tw.get('primal_tail'), tw.get('sweep_tail')
    out = {'step_ms': round(e0.elapsed_time(e1), 3)}
    if p and s:
        t0 = s['first_start_tick']
        rel = lambda d, k: round((d['first_start_tick'] - t0) / 100.0 + d[k], 1)
        out['sweep_tail_us'] = {'first_start': 0.0, 'latest_start': rel(s, 'latest_start_us'), 'earliest_end': rel(s, 'earliest_end_us'), 'latest_end': rel(s, 'latest_end_us')}
        out['primal_tail_us'] = {'first_start': round((p['first_start_tick'] - t0) / 100.0, 1), 'latest_start': rel(p, 'latest_start_us'),
                                 'earliest_end': rel(p, 'earliest_end_us'), 'latest_end': rel(p, 'latest_end_us')}
    out['both'] = {k: tw[k] for k in ('sum_us', 'sum_clocks', 'sum_refill_clocks', 'refills')}
    print('DIAG ' + json.dumps(out))
