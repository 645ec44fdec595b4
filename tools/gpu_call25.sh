#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c25; mkdir -p $O
for v in default fastcell; do
  if [ $v = default ]; then L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so; else L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$v.so; fi
  DSDF_LIB_PATH=$L timeout 100 python tools/ab_check.py > $O/ab_$v.log 2>&1; grep "^AB" $O/ab_$v.log || tail -5 $O/ab_$v.log
done
