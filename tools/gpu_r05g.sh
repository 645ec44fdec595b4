#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -q -m gpu -p no:cacheprovider -k "tile_split or rccl or autograd" > $O/tests.log 2>&1; echo "pytest rc $?"; tail -4 $O/tests.log
for r in 6/8 7/8 3/4; do
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration --emulate-rank $r 2>/dev/null | grep -o '"scaling_prediction".*'
done
