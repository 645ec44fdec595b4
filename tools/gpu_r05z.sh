#!/bin/bash
# Round 5: k_backward_coef of the sweep's own samples BESIDE the tail kernels (DSDF_COEF_EARLY=2), again, now that the tail kernels
# no longer fight over four cache lines of queue counters
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05z base:default ce2:default:DSDF_COEF_EARLY=2 base2:default ce2b:default:DSDF_COEF_EARLY=2 ce1:default:DSDF_COEF_EARLY=1 | cut -c1-120
python - <<'PY'
import json
for l in open('gpurun_out/r05z/ab.jsonl'):
    d = json.loads(l[3:])
    print(d['tag'], json.dumps({k: d[k] for k in ('primal256', 'grad64', 'step')}), d['checksums']['step_grad'], d['checksums']['grad64'])
PY
