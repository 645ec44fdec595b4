#!/bin/bash
# tile-window film splat of the general (spp < 64) pass: parity + timing
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_golden.py -q -x -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"; tail -3 $O/tests.log
timeout 300 python tools/ab_time.py > $O/ab_new.log 2>&1; grep "^AB" $O/ab_new.log || tail -5 $O/ab_new.log
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python tools/lowspp_workload.py > $O/log.txt 2>&1; find $O -name "*.db" -delete; grep "^\"void k_\|^\"k_" $O/trace/t_kernel_stats.csv | cut -c1-60,200-400 | head -8
