#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c20; mkdir -p $O
for v in default reuse; do
  if [ $v = default ]; then L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so; else L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$v.so; fi
  DSDF_LIB_PATH=$L timeout 200 python - > $O/ab_$v.log 2>&1 <<'PY'
import json, os, sys, torch
sys.path.insert(0, 'differentiable-sdf-rendering_amd/python'); sys.path.insert(0, '.')
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
S = list(range(12))
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)
out = {'lib': os.path.basename(os.environ['DSDF_LIB_PATH'])}
for spp in (1, 2, 4, 8, 16, 32):
    out[f'primal{spp}'] = t(lambda: dsdf.render_forward(grid, sens, spp, seeds=S))
out['shade4'] = t(lambda: dsdf.render_forward(grid, sens, 4, seeds=S, integrator=1))
a = dsdf.render_forward(grid, sens, 4, seeds=S); out['sum4'] = float(a.double().sum())
print('AB ' + json.dumps(out))
PY
  grep "^AB" $O/ab_$v.log || tail -5 $O/ab_$v.log
done
