"""How much of |HIP - reference fp64| at the C1-size fixture sits in single sample footprints?  rel-L2 of dL/dsdf after setting aside the
k worst 7^3 cubes (tests/precision.py: greedy_blocks), for the HIP path, the host build of the kernel arithmetic and the reference's own
fp32 run.  GPU box; test infrastructure."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')):
    sys.path.insert(0, p)
import dsdf
import precision as P
import test_refshim_fixture as T
import __graft_entry__ as g
from conftest import HostHarness
dsdf.load()
hh = HostHarness(g.build_harness())
print('| case | run | who | plain | minus 1 footprint | minus 3 | minus 10 |')
print('|---|---|---|---|---|---|---|')
for name, tag in (('c1_spp4', 'sil'), ('c1_spp4', 'shade'), ('blob32', 'sil'), ('sphere16', 'shade')):
    ref, r32 = T.load(name), T.load32(name)
    a64 = ref[f'grad_{tag}']
    for who, gg in (('HIP', T._gpu_backward(dsdf, name, tag)), ('host build', T._host_backward(hh, ref, T.inputs(ref), tag)), ('reference fp32', r32[f'grad_{tag}'])):
        row = [P.greedy_blocks(gg, a64, 0.0, k)[0] if k else T.rel_l2(gg, a64) for k in (0, 1, 3, 10)]
        print(f'| {name} | {tag} | {who} | ' + ' | '.join(f'{v:.2e}' for v in row) + ' |')
