#!/bin/bash
# Final validation of a round: full GPU parity suite, smoke(), PMC passes (-> profiles/<tag>_sq.json, profiles/valu_model.json),
# bench (reads that model), kernel trace of the same bench command, precision table (fast vs IEEE build).
# Usage (through gpurun, from the repository root): bash tools/gpu_final.sh <tag> [skip-tests]      -> gpurun_out/<tag>/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-final}
O=gpurun_out/$TAG; mkdir -p $O; rm -f gpurun_out/precision.jsonl
python -c "import __graft_entry__ as g; g.build()" || exit 1          # (library, IEEE variant, harness, oracles: all at this tree's state)
if [ "$2" != "skip-tests" ]; then
echo "== tests"; date
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/gpu_tests.log; tail -25 $O/gpu_tests.log
echo "== smoke"; date
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -5 $O/smoke.log
cp gpurun_out/precision.jsonl $O/precision_tests.jsonl 2>/dev/null
fi
echo "== PMC"; date
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
echo "== bench (reads profiles/valu_model.json written above)"; date
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json; tail -2 $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --overlap 0 --no-cpu-baseline --no-direct --no-opt-iteration > $O/bench_seq.json 2> $O/bench_seq.err
echo "== kernel trace (the default bench command, without the side blocks)"; date
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration --no-scaling-prediction > $O/trace.log 2>&1
f=$(find $O/trace -name "t_kernel_stats.csv" | head -1); k=$(find $O/trace -name "t_kernel_trace.csv" | head -1)
[ -n "$f" ] && python profiles/summarize.py $TAG $f $k > $O/summarize.log 2>&1
[ -n "$k" ] && python tools/step_timeline.py $k > profiles/${TAG}_step_timeline.md 2> $O/step_timeline.err
find $O -name "*.db" -delete
mkdir -p $O/profiles && cp profiles/${TAG}_* profiles/valu_model.json $O/profiles/ 2>/dev/null
echo "== precision table"; date
timeout 1200 python tools/precision_table.py --out $O/precision.json > $O/precision_table.md 2> $O/precision_table.err; tail -20 $O/precision_table.md
du -sh $O; date
