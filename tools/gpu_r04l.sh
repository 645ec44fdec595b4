#!/bin/bash
# bisect of the arithmetic changes against the gradient gates
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T="tests/test_gpu_parity.py::test_render_backward_gpu tests/test_gpu_direct.py::test_direct_backward_gpu tests/test_gpu_config_size.py::test_config_size_parity"
for v in default oldw oldf oldwf; do
  L=$PWD/differentiable-sdf-rendering_amd/lib/libdsdf.so; [ $v != default ] && L=$PWD/differentiable-sdf-rendering_amd/lib/variants/libdsdf_$v.so
  echo "== $v"
  DSDF_LIB_PATH=$L timeout 900 python -m pytest $T -q -m gpu -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-150
  grep '"grad"' gpurun_out/precision.jsonl 2>/dev/null | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin]
bad=[(r['case'],r['integ'],r['reparam'],round(r['err']/max(r['floor'],1e-12),2),round(r.get('err_trim',0)/max(r.get('floor_trim',1e-12),1e-12),2)) for r in rows]
print(sorted(bad,key=lambda b:-b[3])[:8])"
  rm -f gpurun_out/precision.jsonl
done
