#!/bin/bash
# Round 5: with the tail kernels' claims cheap -- hand-off thresholds of the render kernels once more (variant builds)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05z2 base:default ph12:ph12 ph16:ph16 ph20:ph20 c3:c3 c4:c4 ph16g6:ph16g6 ph16tb8:ph16:DSDF_TAIL_BLOCKS=8 ph16b:ph16 base2:default > /dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r05z2/ab.jsonl'):
    d = json.loads(l[3:])
    print(d['tag'], {k: d[k] for k in ('primal256', 'grad64', 'step')}, d['primal_stats']['tail_rays'], d['grad_stats']['tail_rays'])
PY
