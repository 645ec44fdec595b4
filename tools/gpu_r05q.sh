#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_ab.sh r05q d1 pc128:pc128 pc256:pc256 pc512:pc512 d2 sc64:sc64 sc128:sc128 sc256:sc256 c256_128:c256_128 d3
export AB_VIEWS=2
bash tools/gpu_ab.sh r05q v2d1 v2pc128:pc128 v2pc256:pc256 v2sc64:sc64 v2sc128:sc128 v2c:c256_128 v2d2
