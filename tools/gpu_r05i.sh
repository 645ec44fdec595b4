#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_abd.sh r05i d0:default base:base
AB_SCRIPT=tools/ab_direct.py bash tools/gpu_trace.sh r05i dtr:default > /dev/null 2>&1
AB_ARGS="" ; for t in dtr; do head -14 gpurun_out/r05i/trace_${t}_stats.csv | cut -c1-150; done
