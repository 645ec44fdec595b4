#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x > $O/parity.log 2>&1; echo "parity rc $?"; tail -4 $O/parity.log
bash tools/gpu_ab.sh r05l new base:base new2
AB_ARGS="--shade --low" bash tools/gpu_ab.sh r05l newsl
