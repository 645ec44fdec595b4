#!/bin/bash
# counters of the 4/1-spp step (general pass with the LDS film window)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c19; mkdir -p $O
P="timeout 200 rocprofv3 --kernel-trace --output-format csv"
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_a -o a -- python tools/lowspp_workload.py > $O/pmc_a.log 2>&1
$P --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/pmc_c -o c -- python tools/lowspp_workload.py > $O/pmc_c.log 2>&1
$P --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_b -o b -- python tools/lowspp_workload.py > $O/pmc_b.log 2>&1
$P --pmc FETCH_SIZE -d $O/pmc_f -o f -- python tools/lowspp_workload.py > $O/pmc_f.log 2>&1
$P --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_w -o w -- python tools/lowspp_workload.py > $O/pmc_w.log 2>&1
find $O -name "*.db" -delete; du -sh $O
