#!/bin/bash
# Round 5, the last 4 GPU-minutes: the new external-witness GPU legs, then as much of tests/test_gpu_parity.py as fits (one process:
# the xdist attempt of gpu_r05_last.sh spent its 6 minutes on 8 workers bringing themselves up).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05last2; mkdir -p $O
find . -name "*.so" | xargs touch
date
timeout 120 python -m pytest tests/test_external_witnesses.py -q -m gpu -x -p no:cacheprovider > $O/witness.txt 2>&1; echo "witness rc $?" | tee -a $O/witness.txt; tail -4 $O/witness.txt
date
timeout 170 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider --durations=10 > $O/parity.txt 2>&1; echo "parity rc $?" | tee -a $O/parity.txt; tail -6 $O/parity.txt
date
