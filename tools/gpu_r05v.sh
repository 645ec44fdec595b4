#!/bin/bash
# Round 5: on top of DSDF_TAIL_BATCH=1 (the default build): refill threshold of the tail waves, hand-off thresholds of the render kernels.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05v base:default r16:r16 r32:r32 r40:r40 ph12:ph12 th12:th12 base2:default
