#!/bin/bash
# Round 5: the cooperative march (csrc/dsdf_coop.h) in the tail kernels of the 256 / 64-spp step (DSDF_COOP_TAIL bit 0: primal tail,
# bit 1: the sweep's tail) and in the general pass (DSDF_COOP=3), against the default build; then the parity suite on the full variant.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05t; mkdir -p $O
AB_ARGS="--low" bash tools/gpu_ab.sh r05t base:default t1:coopt1 t2:coopt2 t3:coopt3 base2:default t3b:coopt3 t1b:coopt1 t2b:coopt2
V=$PWD/differentiable-sdf-rendering_amd/lib/variants
DSDF_LIB_PATH=$V/libdsdf_coopt3.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_refshim_fixture.py -q -m gpu -p no:cacheprovider > $O/parity_coopt3.log 2>&1; echo "parity rc $?"; tail -8 $O/parity_coopt3.log
