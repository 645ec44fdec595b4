#!/bin/bash
# Round-2 GPU call 10: two-stream step (bench with / without overlap), occupancy variants of k_backward / tail, quick parity.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c10; mkdir -p $O
V=differentiable-sdf-rendering_amd/lib/variants
echo "== parity"; date
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "two_stream or render_backward_gpu or tile_split or render_forward_gpu" > $O/gpu_quick.log 2>&1; echo "quick rc $?"; tail -4 $O/gpu_quick.log
echo "== bench overlap on/off"; date
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_overlap.json 2> $O/bench_overlap.err; python -c "
import json; d=json.load(open('$O/bench_overlap.json')); print('overlap', d['value'], d['ms_per_step'], d['low_spp']['value'], d['roofline']['frac'], d['config'].get('primal_ms_per_launch'), d['config'].get('grad_ms_per_launch'))"; tail -2 $O/bench_overlap.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --overlap 0 > $O/bench_seq.json 2> $O/bench_seq.err; python -c "
import json; d=json.load(open('$O/bench_seq.json')); print('sequential', d['value'], d['ms_per_step'], d['low_spp']['value'], d['roofline']['frac'])"
echo "== A/B"; date
for T in default bmw3 tmw3; do
  if [ $T = default ]; then L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so; else L=$PWD/$V/libdsdf_$T.so; fi
  DSDF_LIB_PATH=$L timeout 300 python tools/ab_time.py > $O/ab_$T.log 2>&1; grep "^AB" $O/ab_$T.log | tee -a $O/ab.log
done
date
