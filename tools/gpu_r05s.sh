#!/bin/bash
# Round 5: cooperative march of the last rays (csrc/dsdf_coop.h; -DDSDF_COOP=1: value-only traces, =3: + differentiable traces)
# against the default build, low-spp steps.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05s; mkdir -p $O
V=differentiable-sdf-rendering_amd/lib/variants
for cfg in "4 1" "16 4"; do
  set -- $cfg
  for rep in 1 2; do
    AB_TAG=base_$1_$2 AB_SPP=$1 AB_SPP_GRAD=$2 AB_DUMP=/tmp/base_$1.pt timeout 300 python tools/ab_low.py 2>&1 | grep "^AB" | tee -a $O/ab.txt
    for v in coop1 coop3; do
      AB_TAG=${v}_$1_$2 AB_SPP=$1 AB_SPP_GRAD=$2 AB_CMP=/tmp/base_$1.pt DSDF_LIB_PATH=$V/libdsdf_$v.so timeout 300 python tools/ab_low.py 2>&1 | grep "^AB" | tee -a $O/ab.txt
    done
  done
done
