#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "two_stream or autograd or shared_pixel or tile_split or render_backward_gpu or large_spp or gradient_descends" > $O/parity.log 2>&1; echo "parity rc $?"; tail -3 $O/parity.log
bash tools/gpu_ab.sh r05n h1 s1:default:DSDF_COEF_IN_TAIL=0 h2 s2:default:DSDF_COEF_IN_TAIL=0 h3 s3:default:DSDF_COEF_IN_TAIL=0 w1:default:DSDF_COEF_IN_TAIL_WAVES=1
AB_ARGS="--shade" bash tools/gpu_ab.sh r05n tsh ssh:default:DSDF_COEF_IN_TAIL=0
bash tools/gpu_trace.sh r05n tr:default > /dev/null 2>&1
python tools/step_timeline.py gpurun_out/r05n/trace_tr_kernels.csv | tail -9
find gpurun_out/r05n -name "*.csv" -size +2M -delete
