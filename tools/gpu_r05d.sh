#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05d; mkdir -p $O; rm -f gpurun_out/precision.jsonl
date
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_refshim_fixture.py tests/test_gpu_principled.py tests/test_to_world.py tests/test_aovs.py tests/test_antithetic.py tests/test_gpu_optimize.py -q -m gpu -p no:cacheprovider \
  -k "autograd or reference_fp32 or principled or to_world or transformed or aov or antithetic or optimize" > $O/tests.log 2>&1; echo "pytest rc $?"; tail -15 $O/tests.log
cp gpurun_out/precision.jsonl $O/precision_tests.jsonl 2>/dev/null
date
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration > $O/bench_scaling.json 2> $O/bench_scaling.err; tail -c 3000 $O/bench_scaling.json; tail -3 $O/bench_scaling.err
date
