#!/bin/bash
# Round 5: kernel trace of tools/ab_step.py on the default build (timeline of its last two-stream step: tools/step_timeline.py <csv> 1)
cd "$(dirname "$0")/.."
bash tools/gpu_trace.sh r05x base:default
python tools/step_timeline.py gpurun_out/r05x/trace_base_kernels.csv 1 > gpurun_out/r05x/step_timeline.md; cat gpurun_out/r05x/step_timeline.md | tail -30
