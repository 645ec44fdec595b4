#!/bin/bash
# kernel traces of the bench step, sequential (clean per-kernel durations) and two-stream
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" || exit 1
for mode in 0 1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$mode -o t -- python bench.py --steps 5 --warmup 2 --overlap $mode --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration > $O/trace$mode.log 2>&1
  f=$(find $O/trace$mode -name "t_kernel_stats.csv" | head -1); k=$(find $O/trace$mode -name "t_kernel_trace.csv" | head -1)
  cp $f $O/stats$mode.csv; cp $k $O/kernels$mode.csv; rm -rf $O/trace$mode
  echo "== overlap $mode"; head -16 $O/stats$mode.csv | cut -c1-150
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "hit_proof or two_stream or autograd or shared_pixel" -p no:cacheprovider 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_optimize.py tests/test_to_world.py tests/test_gpu_principled.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
