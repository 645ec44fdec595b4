#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05h new sw5:sw5 sw6:sw6 new2 sw5b:sw5
bash tools/gpu_abd.sh r05h d0:default dp4:dp4 ds4:ds4 bh:bh bh4:bh4 d1:default
