#!/bin/bash
# dry run of the measurement plumbing: one PMC pass of the new workload, the summariser, a short bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04r; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
export PMC_STATS_OUT=$PWD/$O/pmc_stats.json
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_a -o a -- python tools/pmc_workload.py --low --direct > $O/pmc_a.log 2>&1; tail -2 $O/pmc_a.log
unset PMC_STATS_OUT
find $O -name "*.db" -delete
cp profiles/valu_model.json $O/valu_model_before.json
python profiles/summarize_pmc.py r04r $O $O/pmc_stats.json > $O/summarize_pmc.log 2>&1; tail -3 $O/summarize_pmc.log | cut -c1-1500
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; tail -3 $O/bench.err
cp profiles/valu_model.json $O/valu_model_after.json; cp profiles/r04r_sq.json $O/
