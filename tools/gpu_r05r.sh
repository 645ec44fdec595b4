#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05r; mkdir -p $O
timeout 600 python -m pytest tests/test_refshim_fixture.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "reference_fp32 or large_spp" > $O/tests.log 2>&1; echo "pytest rc $?"; tail -3 $O/tests.log
timeout 1200 python tools/precision_table.py --out $O/precision.json > $O/precision_table.md 2> $O/precision_table.err; tail -16 $O/precision_table.md
