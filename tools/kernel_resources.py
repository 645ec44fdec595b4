#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel of libdsdf.so (hipcc -Rpass-analysis=kernel-resource-usage), optionally with
extra -D flags:  python tools/kernel_resources.py [-DDSDF_X=1 ...] [--filter render_items]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

defs = [a for a in sys.argv[1:] if a.startswith('-D')]
flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
src = os.path.join(g.PKG, 'csrc', 'dsdf_kernels.hip')
r = subprocess.run(['/opt/rocm/bin/hipcc'] + g.HIPCC_FLAGS + defs + ['-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/_res.so', src],
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd='/tmp')
rows, cur = [], None
for line in r.stdout.splitlines():
    m = re.search(r'remark:\s+(.*?)\s+\[-Rpass', line)
    if not m:
        if 'error' in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        name = subprocess.run(['c++filt', t.split(':', 1)[1].strip()], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', name).replace('void ', '')}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for c in rows:
    if flt in c['name']:
        print(f"{c['name'][:58]:58s} {c.get('VGPRs', '?'):>5s} {c.get('AGPRs', '?'):>5s} {c.get('SGPRs', '?'):>5s} "
              f"{c.get('ScratchSize [bytes/lane]', '?'):>8s} {c.get('Occupancy [waves/SIMD]', '?'):>4s} {c.get('LDS Size [bytes/block]', '?'):>7s}")
