#!/bin/bash
# Round 5: tail kernels -- samples of finished rays completed in batches at refill time (DSDF_TAIL_BATCH), gathers of rays that
# change cell overlapped with the others' step (DSDF_TAIL_DEFER bit 0: primal tail, bit 1: the sweep's tail); default build = base.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05u base:default b1:b1 b1d1:b1d1 b1d3:b1d3 d1:d1 base2:default b1b:b1 b1d1b:b1d1 b1d3b:b1d3 d1b:d1
