#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04k; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
date; echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration --overlap 0 > $O/bench_seq.json 2>> $O/bench.err; cut -c1-400 $O/bench_seq.json
date; echo "== tests"
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -40 $O/gpu_tests.log
date
