#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05c new fused:default:DSDF_BWD_SPLIT=0 cearly:default:DSDF_COEF_EARLY=1 new2 fused2:default:DSDF_BWD_SPLIT=0 \
   sw2:default:DSDF_SWEEP_WORKERS=2 sw3pw6:default:DSDF_SWEEP_WORKERS=3,DSDF_PRIMAL_WORKERS=6 new3
