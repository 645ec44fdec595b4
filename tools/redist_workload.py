"""One dsdf.redistance call on the bench grid (rocprofv3 --kernel-trace shows the duration of every round)."""
import sys, torch
sys.path.insert(0, "differentiable-sdf-rendering_amd/python"); sys.path.insert(0, ".")
import dsdf
from bench import synth_grid
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
phi = synth_grid(R, "cuda")
out = dsdf.redistance(phi); torch.cuda.synchronize()
out, cnt = dsdf.redistance(phi, return_counters=True); torch.cuda.synchronize()
print("counters", cnt.tolist())
