#!/bin/bash
# Round 5: where the tail waves of the two-stream step are in time (tools/tail_diag.py); the cooperative march in the primal tail again,
# now that a tail wave's time is its rays' steps (ct1 = -DDSDF_COOP_TAIL=1)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r05y
timeout 300 python tools/tail_diag.py 2>&1 | grep DIAG | tee gpurun_out/r05y/diag.txt
bash tools/gpu_ab.sh r05y base:default ct1:ct1 base2:default ct1b:ct1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('AB'):
        d = json.loads(l[3:] if l.rstrip().endswith('}') else '{}') if l.rstrip().endswith('}') else None
        print(l[:150])
"
python - <<'PY'
import json
for l in open('gpurun_out/r05y/ab.jsonl'):
    d = json.loads(l[3:])
    print(d['tag'], json.dumps({k: d[k] for k in ('primal256', 'grad64', 'step')}), 'primal tail', json.dumps({k: d['primal_stats']['tail_waves'][k] for k in ('max_wave_steps', 'max_us', 'sum_us')}))
PY
