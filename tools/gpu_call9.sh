#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c9; mkdir -p $O
V=differentiable-sdf-rendering_amd/lib/variants
cat > /tmp/ab_small.py <<'PY'
import json, os, sys, torch
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')); sys.path.insert(0, ROOT)
import dsdf
from bench import synth_grid
dev = torch.device('cuda')
data = synth_grid(256, dev); grid = dsdf.SdfGrid(data)
sens = dsdf.get_regular_cameras(12, resx=512, resy=512)
def t(fn, n=3):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)
S = list(range(12))
out = {'lib': os.path.basename(os.environ.get('DSDF_LIB_PATH', 'libdsdf.so'))}
out['primal256_stream'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S))
out['primal256_chunks'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, stream=False))
out['shade256_stream'] = t(lambda: dsdf.render_forward(grid, sens, 256, seeds=S, integrator=1))
a = dsdf.render_forward(grid, sens, 256, seeds=S); b = dsdf.render_forward(grid, sens, 256, seeds=S, stream=False)
out['rel'] = float((a - b).norm() / b.norm())
print('AB ' + json.dumps(out))
PY
for T in default sr8 sr32 smw5 smw5r32; do
  if [ $T = default ]; then L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so; else L=$PWD/$V/libdsdf_$T.so; fi
  DSDF_LIB_PATH=$L timeout 200 python /tmp/ab_small.py > $O/ab_$T.log 2>&1; grep "^AB" $O/ab_$T.log | tee -a $O/ab.log; tail -1 $O/ab_$T.log | cut -c1-200
done
