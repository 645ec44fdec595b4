#!/bin/bash
# Parametrised A/B on the GPU box (through gpurun, from the repository root):
#   bash tools/gpu_ab.sh <outdir-tag> "<tag>[:<variant lib tag>][:ENV=val,ENV=val]" ...   [AB_ARGS="--shade --low"]
# every spec runs tools/ab_step.py in its own process; results (one `AB {json}` line each) -> gpurun_out/<outdir-tag>/ab.jsonl
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$1; shift; mkdir -p $O
for spec in "$@"; do
  IFS=':' read -r tag lib envs <<< "$spec"
  L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so
  [ -n "$lib" ] && [ "$lib" != default ] && L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$lib.so
  E="AB_TAG=$tag DSDF_LIB_PATH=$L"
  [ -n "$envs" ] && E="$E ${envs//,/ }"
  env $E timeout 150 python tools/ab_step.py $AB_ARGS > $O/ab_$tag.log 2>&1
  grep "^AB" $O/ab_$tag.log | tee -a $O/ab.jsonl | cut -c1-420 || tail -3 $O/ab_$tag.log
done
