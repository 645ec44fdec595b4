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
  ab)      # bash tools/gpu_run.sh ab <tag> "<name>[:<variant lib tag>][:ENV=val,ENV=val]" ...   (AB_SCRIPT / AB_ARGS / AB_REPEAT from the env)
    shift 2
    for rep in $(seq 1 ${AB_REPEAT:-1}); do
    for spec in "$@"; do
      IFS=':' read -r name lib envs <<< "$spec"
      L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so
      [ -n "$lib" ] && [ "$lib" != default ] && L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$lib.so
      E="AB_TAG=$name DSDF_LIB_PATH=$L"
      [ -n "$envs" ] && E="$E ${envs//,/ }"
      env $E timeout 200 python ${AB_SCRIPT:-tools/ab_step.py} $AB_ARGS > $O/ab_${name}_$rep.log 2>&1
      grep "^AB" $O/ab_${name}_$rep.log | tee -a $O/ab.jsonl | cut -c1-300 || tail -3 $O/ab_${name}_$rep.log
    done; done ;;
  trace)   # kernel trace of tools/ab_step.py (or AB_SCRIPT) under the same spec syntax as `ab`
    shift 2
    for spec in "$@"; do
      IFS=':' read -r name lib envs <<< "$spec"
      L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so
      [ -n "$lib" ] && [ "$lib" != default ] && L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$lib.so
      E="AB_TAG=$name DSDF_LIB_PATH=$L"
      [ -n "$envs" ] && E="$E ${envs//,/ }"
      env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o t -- python ${AB_SCRIPT:-tools/ab_step.py} $AB_ARGS > $O/trace_$name.log 2>&1
      f=$(find $O/trace_$name -name "t_kernel_stats.csv" | head -1)
      [ -n "$f" ] && cp $f $O/trace_${name}_stats.csv && head -12 $f | cut -c1-160
      k=$(find $O/trace_$name -name "t_kernel_trace.csv" | head -1)
      [ -n "$k" ] && cp $k $O/trace_${name}_kernels.csv
      rm -rf $O/trace_$name
    done ;;
  final)   # full validation of a round: bash tools/gpu_run.sh final <tag> [skip-tests]
    rm -f gpurun_out/precision.jsonl
    python -c "import __graft_entry__ as g; g.build()" || exit 1
    if [ "$3" != "skip-tests" ]; then
      run gpu_tests 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15; tail -25 $O/gpu_tests.out
      run smoke 600 python __graft_entry__.py smoke; tail -5 $O/smoke.out
      cp gpurun_out/precision.jsonl $O/precision_tests.jsonl 2>/dev/null
    fi
    P="timeout 900 rocprofv3 --kernel-trace --output-format csv"
    export PMC_STATS_OUT=$PWD/$O/pmc_stats.json
    $P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_a -o a -- python tools/pmc_workload.py --low --direct > $O/pmc_a.log 2>&1; tail -1 $O/pmc_a.log
    unset PMC_STATS_OUT
    $P --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_b -o b -- python tools/pmc_workload.py --low --direct > $O/pmc_b.log 2>&1; tail -1 $O/pmc_b.log
    $P --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/pmc_c -o c -- python tools/pmc_workload.py --low --direct > $O/pmc_c.log 2>&1; tail -1 $O/pmc_c.log
    $P --pmc FETCH_SIZE -d $O/pmc_f -o f -- python tools/pmc_workload.py --low --direct > $O/pmc_f.log 2>&1; tail -1 $O/pmc_f.log
    $P --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_w -o w -- python tools/pmc_workload.py --low --direct > $O/pmc_w.log 2>&1; tail -1 $O/pmc_w.log
    find $O -name "*.db" -delete
    python profiles/summarize_pmc.py $TAG $O $O/pmc_stats.json > $O/summarize_pmc.log 2>&1; tail -4 $O/summarize_pmc.log | cut -c1-600
    run bench 600 $DRIVER; tail -1 $O/bench.out > $O/bench.json
    run bench_seq 300 $DRIVER --overlap 0 --no-cpu-baseline --no-direct --no-opt-iteration --no-scaling-prediction
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration --no-scaling-prediction > $O/trace.log 2>&1
    f=$(find $O/trace -name "t_kernel_stats.csv" | head -1); k=$(find $O/trace -name "t_kernel_trace.csv" | head -1)
    [ -n "$f" ] && python profiles/summarize.py $TAG $f $k > $O/summarize.log 2>&1
    [ -n "$k" ] && python tools/step_timeline.py $k > profiles/${TAG}_step_timeline.md 2> $O/step_timeline.err
    find $O -name "*.db" -delete
    mkdir -p $O/profiles && cp profiles/${TAG}_* profiles/valu_model.json $O/profiles/ 2>/dev/null
    timeout 1200 python tools/precision_table.py --out $O/precision.json > $O/precision_table.md 2> $O/precision_table.err; tail -20 $O/precision_table.md
    du -sh $O ;;
  *) shift; run custom 1200 bash -c "$JOB" ;;
esac
date
