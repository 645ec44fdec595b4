#!/bin/bash
# Round-2 GPU call 8: streaming primal workers -- parity first, then A/B against the chunk workers and occupancy variants.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c8; mkdir -p $O
V=differentiable-sdf-rendering_amd/lib/variants
echo "== parity of the streaming pass"; date
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_size.py tests/test_golden.py -q -m gpu -x -p no:cacheprovider -k "streaming or render_forward_gpu or config_size or golden or odd_film or skip_is_exact or tile_split or large_spp" > $O/gpu_quick.log 2>&1; echo "quick rc $?"; tail -6 $O/gpu_quick.log
echo "== A/B"; date
timeout 300 python tools/ab_time.py > $O/ab_default.log 2>&1; grep "^AB" $O/ab_default.log | tee $O/ab.log; tail -2 $O/ab_default.log | cut -c1-300
for T in smw6 smw5 treuse; do
  DSDF_LIB_PATH=$PWD/$V/libdsdf_$T.so timeout 300 python tools/ab_time.py > $O/ab_$T.log 2>&1; grep "^AB" $O/ab_$T.log | tee -a $O/ab.log
done
echo "== PMC"; date
P="timeout 600 rocprofv3 --kernel-trace --output-format csv"
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_a -o a -- python tools/pmc_workload.py > $O/pmc_a.log 2>&1; tail -1 $O/pmc_a.log
$P --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/pmc_c -o c -- python tools/pmc_workload.py > $O/pmc_c.log 2>&1; tail -1 $O/pmc_c.log
find $O -name "*.db" -delete; du -sh $O; date
