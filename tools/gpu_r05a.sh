#!/bin/bash
# Round 5, first A/B batch on the GPU box (through gpurun, from the repository root):  bash tools/gpu_r05a.sh
# variants are built beforehand with tools/build_ab_variants.py (they travel with the snapshot)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05a; mkdir -p $O
date
# forward parity of the streaming primal kernel first (cheap, and a wrong kernel should not be timed)
DSDF_STREAM=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x \
  -k "test_render_forward_gpu or test_hit_proof_is_exact or test_empty_space_skip_is_exact or test_tile_split or test_multi_view or test_two_stream or test_builtin_sampler" \
  > $O/stream_parity.log 2>&1; echo "stream parity rc $?"; tail -5 $O/stream_parity.log
date
bash tools/gpu_ab.sh r05a \
  base:base \
  fast:default \
  tl64:default:DSDF_TAIL_LONG=64 \
  tl128:default:DSDF_TAIL_LONG=128 \
  tl256:default:DSDF_TAIL_LONG=256 \
  treuse:treuse:DSDF_TAIL_LONG=128 \
  treuse0:treuse \
  sweepc:sweepc \
  st10:default:DSDF_STREAM=1 \
  st8:default:DSDF_STREAM=1,DSDF_STREAM_SEG_LOG2=8 \
  spf:spf:DSDF_STREAM=1 \
  s7:s7:DSDF_STREAM=1 \
  st10tl:default:DSDF_STREAM=1,DSDF_TAIL_LONG=128
date
