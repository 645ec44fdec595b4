#!/bin/bash
# Round-2 GPU call 1: parity suite, precision table (fast vs IEEE build), A/B of the staged kernel variants, kernel
# trace and PMC passes of the shipped kernels.  Everything lands under gpurun_out/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c1; mkdir -p $O; rm -f gpurun_out/precision.jsonl
V=differentiable-sdf-rendering_amd/lib/variants
echo "== tests"; date
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/gpu_tests.log; tail -5 $O/gpu_tests.log
echo "== precision table"; date
timeout 900 python tools/precision_table.py --out $O/r02_precision.json > $O/precision_table.md 2> $O/precision_table.err; tail -20 $O/precision_table.md
echo "== A/B"; date
timeout 300 python tools/ab_time.py --direct 2>&1 | grep "^AB" | tee $O/ab.log
for T in t4 t8 t16 reuse; do DSDF_LIB_PATH=$PWD/$V/libdsdf_$T.so timeout 300 python tools/ab_time.py 2>&1 | grep "^AB" | tee -a $O/ab.log; done
DSDF_LIB_PATH=$PWD/$V/libdsdf_shadow.so timeout 300 python tools/ab_time.py --direct 2>&1 | grep "^AB" | tee -a $O/ab.log
# parity of the hand-off build (primal + gradient sweep)
DSDF_LIB_PATH=$PWD/$V/libdsdf_t8.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -m gpu -p no:cacheprovider > $O/gpu_tests_t8.log 2>&1; tail -3 $O/gpu_tests_t8.log
echo "== kernel trace"; date
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1; tail -2 $O/trace.log | cut -c1-600
echo "== PMC"; date
P="timeout 600 rocprofv3 --kernel-trace --output-format csv"
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_a -o a -- python tools/pmc_workload.py > $O/pmc_a.log 2>&1; tail -1 $O/pmc_a.log
$P --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_b -o b -- python tools/pmc_workload.py > $O/pmc_b.log 2>&1; tail -1 $O/pmc_b.log
$P --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $O/pmc_c -o c -- python tools/pmc_workload.py > $O/pmc_c.log 2>&1; tail -1 $O/pmc_c.log
$P --pmc FETCH_SIZE -d $O/pmc_f -o f -- python tools/pmc_workload.py > $O/pmc_f.log 2>&1; tail -1 $O/pmc_f.log
$P --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_w -o w -- python tools/pmc_workload.py > $O/pmc_w.log 2>&1; tail -1 $O/pmc_w.log
find $O -name "*.db" -delete; du -sh $O; date
