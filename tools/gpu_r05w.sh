#!/bin/bash
# Round 5: what are the tail waves' 2.4 / 3.7 ms made of (tail_waves in the stats: iterations, residence, shader clocks, clocks spent
# refilling), and the views in LDS (default build) against the kernel arguments (nolds).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05w old:old base:default scan0:scan0 str2:str2 old2:old base2:default
python - <<'PY'
import json
for l in open('gpurun_out/r05w/ab.jsonl'):
    d = json.loads(l[3:])
    print(d['tag'], json.dumps({k: d[k] for k in ('primal256', 'grad64', 'step')}), 'primal tail', json.dumps(d['primal_stats'].get('tail_waves')), 'grad tail', json.dumps(d['grad_stats'].get('tail_waves')))
PY
