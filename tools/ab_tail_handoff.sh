#!/bin/bash
# Next-round A/B of the tail hand-off (branch tail-handoff): build the variants, time the passes, run the primal parity tests
# on the hand-off build.  Run through gpurun from the repository root.
set -e
cd "$(dirname "$0")/.."
V=differentiable-sdf-rendering_amd/lib/variants; mkdir -p $V
F="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -fPIC -shared -Iinclude"
S=differentiable-sdf-rendering_amd/csrc/dsdf_kernels.hip
for T in 4 8 16; do [ -f $V/libdsdf_t$T.so ] || hipcc $F -DDSDF_TAIL_HANDOFF=$T -o $V/libdsdf_t$T.so $S; done
timeout 120 python tools/time_passes.py < /dev/null 2>&1 | tail -1
for T in 4 8 16; do DSDF_LIB_PATH=$V/libdsdf_t$T.so timeout 120 python tools/time_passes.py < /dev/null 2>&1 | tail -1; done
DSDF_LIB_PATH=$PWD/$V/libdsdf_t8.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu < /dev/null 2>&1 | tail -3
