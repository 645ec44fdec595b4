#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_principled.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"; tail -5 $O/tests.log
export AB_VIEWS=2
bash tools/gpu_ab.sh r05e v2:default v2tl64:default:DSDF_TAIL_LONG=64 v2tl128:default:DSDF_TAIL_LONG=128 v2tl256:default:DSDF_TAIL_LONG=256 v2tb2:default:DSDF_TAIL_BLOCKS=2 v2tb1:default:DSDF_TAIL_BLOCKS=1 v2b:default
bash tools/gpu_trace.sh r05e v2t:default v2ttl:default:DSDF_TAIL_LONG=128 > /dev/null
for t in v2t v2ttl; do python tools/step_timeline.py $O/trace_${t}_kernels.csv > $O/timeline_$t.md 2>/dev/null; done
export AB_VIEWS=3
bash tools/gpu_ab.sh r05e v3:default v3tl128:default:DSDF_TAIL_LONG=128
find $O -name "*.csv" -size +2M -delete
