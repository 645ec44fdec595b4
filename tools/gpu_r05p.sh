#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05p d1 th16:th16 th12g1:th12g1 th4:th4 d2 pth16:pth16 pth4:pth4 ptg8:ptg8 ptg2:ptg2 d3
