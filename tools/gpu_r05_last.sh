#!/bin/bash
# Round 5, last GPU call of the round (11 GPU-minutes left): the whole GPU suite at HEAD on 8 xdist workers (one GPU, separate
# processes), the default bench command, smoke(), and -- if the budget lasts -- the kernel trace of the bench command.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05last; mkdir -p $O; rm -f gpurun_out/precision.jsonl
date
timeout 360 python -m pytest tests -q -m gpu -n 8 -p no:cacheprovider --durations=15 > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" | tee -a $O/gpu_tests.txt; tail -30 $O/gpu_tests.txt
cp gpurun_out/precision.jsonl $O/precision_tests.jsonl 2>/dev/null
date
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 3000 $O/bench.json; tail -2 $O/bench.err
date
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.log
date
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration --no-scaling-prediction > $O/trace.log 2>&1
f=$(find $O/trace -name "t_kernel_stats.csv" | head -1); k=$(find $O/trace -name "t_kernel_trace.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && head -12 $f | cut -c1-160
[ -n "$k" ] && python tools/step_timeline.py $k > $O/step_timeline.md 2> $O/step_timeline.err
find $O -name "*.db" -delete; rm -rf $O/trace
date
