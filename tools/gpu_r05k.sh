#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_trace.sh r05k dry:dry new:default > /dev/null 2>&1
for t in dry new; do echo "== $t"; python tools/step_timeline.py gpurun_out/r05k/trace_${t}_kernels.csv | tail -8; done
bash tools/gpu_abd.sh r05k d0:default dp4:dp4 dry:dry
find gpurun_out/r05k -name "*.csv" -size +2M -delete
