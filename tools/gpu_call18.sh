#!/bin/bash
# dry run of bench.py's multi-process paths on ONE GPU (gloo collectives, ranks share the device): whole-view shard at
# world 2, pixel-row windows at world 8.  Checks that the code runs end to end and that the gradients agree across N.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo
O=gpurun_out/c18; mkdir -p $O
for n in 1 2 8; do
  if [ $n = 1 ]; then
    timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --no-low-spp --no-cpu-baseline --overlap 0 > $O/b$n.json 2> $O/b$n.err
  else
    timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 2 --warmup 1 --no-low-spp --no-cpu-baseline > $O/b$n.json 2> $O/b$n.err
  fi
  echo "N=$n rc $?"; tail -c 600 $O/b$n.json; tail -3 $O/b$n.err
done
