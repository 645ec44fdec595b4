#!/bin/bash
# The ONE GPU-box script: `gpurun -- bash tools/gpu_run.sh <job> [tag]`.  Every job writes its raw logs to gpurun_out/<tag>/, which
# gpurun merges back; copy what is to be judged into profiles/ in the same commit.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1
JOB=${1:-bench}; TAG=${2:-$JOB}
O=gpurun_out/$TAG; mkdir -p $O
find . -name "*.so" | xargs touch
DRIVER="python3 bench.py --gpus 1 --steps 20 --warmup 5"
run() {  # run <name> <timeout> <cmd...>: stdout -> $O/<name>.out, stderr -> $O/<name>.err, wall clock + rc -> $O/<name>.rc
  local name=$1 to=$2; shift 2
  local t0=$(date +%s.%N)
  timeout $to "$@" > $O/$name.out 2> $O/$name.err; local rc=$?
  echo "rc $rc wall $(python3 -c "import time; print(round(time.time() - $t0, 1))") s: $*" | tee $O/$name.rc
  tail -3 $O/$name.err
  return $rc
}
case $JOB in
  stall)   # the exact driver command, three times; on a stall, bisect with the --no-* flags
    for i in 1 2 3; do
      run driver$i 300 $DRIVER || { BAD=1; break; }
    done
    if [ -n "$BAD" ]; then
      run no_scaling 240 $DRIVER --no-scaling-prediction
      run no_low 240 $DRIVER --no-scaling-prediction --no-low-spp
      run only_head 240 $DRIVER --no-scaling-prediction --no-low-spp --no-direct --no-opt-iteration
    fi ;;
  soak)    # N short full-size runs with every GPU block on (no CPU baseline): looks for the rare stall of BENCH_r05
    N=${3:-12}
    for i in $(seq 1 $N); do
      BENCH_HEADLINE_S=120 BENCH_BLOCK_S=90 run soak$i 400 $DRIVER --no-cpu-baseline || echo "soak $i FAILED"
      grep -c . $O/soak$i.out; grep -o '"aborted": "[^"]*"' $O/soak$i.out
    done ;;
  bench)   # the driver's command once + its rocprofv3 kernel trace
    run driver 300 $DRIVER
    tail -1 $O/driver.out > $O/bench.json ;;
  tests)   # the GPU suite as the driver runs it
    run gputests 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=15 ;;
  smoke)
    run smoke 300 python __graft_entry__.py smoke ;;
  *) shift; run custom 1200 bash -c "$JOB" ;;
esac
date
