#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05m; mkdir -p $O
for v in default slow noreuse slownoreuse; do
  L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so
  [ $v != default ] && L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$v.so
  DSDF_LIB_PATH=$L timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "test_large_spp_not_multiple_of_64" > $O/t_$v.log 2>&1
  echo "$v: $(grep -o 'assert [0-9.e-]* < 0.0001\|[0-9]* passed\|[0-9]* failed' $O/t_$v.log | tr '\n' ' ')"
done
bash tools/gpu_ab.sh r05m n1 s1:slow n2 s2:slow n3 s3:slow
