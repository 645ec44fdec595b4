#!/bin/bash
# pixel items (one film flush per pixel) on the XCD-share list: parity, A/B, counters
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_size.py tests/test_gpu_direct.py tests/test_gpu_mesh_to_sdf.py -q -x -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?"; tail -4 $O/tests.log
for v in pix0 pix1 pix1s256; do
  DSDF_LIB_PATH=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$v.so timeout 300 python tools/ab_time.py --direct > $O/ab_$v.log 2>&1; grep "^AB" $O/ab_$v.log || tail -5 $O/ab_$v.log
done
P="timeout 300 rocprofv3 --kernel-trace --output-format csv"
$P --pmc FETCH_SIZE -d $O/pmc_f -o f -- python tools/pmc_workload.py > $O/pmc_f.log 2>&1
$P --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_w -o w -- python tools/pmc_workload.py > $O/pmc_w.log 2>&1
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_a -o a -- python tools/pmc_workload.py > $O/pmc_a.log 2>&1
$P --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/pmc_c -o c -- python tools/pmc_workload.py > $O/pmc_c.log 2>&1
find $O -name "*.db" -delete; du -sh $O
