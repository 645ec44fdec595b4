#!/bin/bash
# Kernel trace (rocprofv3 --kernel-trace --stats) of tools/ab_step.py under a setting:
#   bash tools/gpu_trace.sh <outdir-tag> "<tag>[:<variant lib tag>][:ENV=val,ENV=val]" ...
# -> gpurun_out/<outdir-tag>/trace_<tag>_stats.csv (+ the dispatch list), summarised by profiles/summarize.py
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$1; shift; mkdir -p $O
for spec in "$@"; do
  IFS=':' read -r tag lib envs <<< "$spec"
  L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so
  [ -n "$lib" ] && [ "$lib" != default ] && L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$lib.so
  E="AB_TAG=$tag DSDF_LIB_PATH=$L"
  [ -n "$envs" ] && E="$E ${envs//,/ }"
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -o t -- python ${AB_SCRIPT:-tools/ab_step.py} $AB_ARGS > $O/trace_$tag.log 2>&1
  f=$(find $O/trace_$tag -name "t_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/trace_${tag}_stats.csv && head -12 $f | cut -c1-160
  k=$(find $O/trace_$tag -name "t_kernel_trace.csv" | head -1)
  [ -n "$k" ] && cp $k $O/trace_${tag}_kernels.csv
  rm -rf $O/trace_$tag
done
