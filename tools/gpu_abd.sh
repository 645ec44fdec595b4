#!/bin/bash
# A/B of sdf_direct_reparam builds: bash tools/gpu_abd.sh <outdir-tag> "<tag>[:<variant lib tag>][:ENV=val,...]" ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$1; shift; mkdir -p $O
for spec in "$@"; do
  IFS=':' read -r tag lib envs <<< "$spec"
  L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so
  [ -n "$lib" ] && [ "$lib" != default ] && L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$lib.so
  E="AB_TAG=$tag DSDF_LIB_PATH=$L"
  [ -n "$envs" ] && E="$E ${envs//,/ }"
  env $E timeout 300 python tools/ab_direct.py > $O/abd_$tag.log 2>&1
  grep "^AB" $O/abd_$tag.log | tee -a $O/abd.jsonl | cut -c1-420 || tail -3 $O/abd_$tag.log
done
