#!/usr/bin/env python3
"""Static instruction mix per basic block of one kernel of the device assembly (hipcc --cuda-device-only -S):
    python tools/isa_blocks.py [-DX=1 ...] --kernel 'k_render_items<false, false, false>' [--min 8] [--dump out.s]
Used to attribute the VALU-per-wave-iteration figure of profiles/valu_model.json to loop bodies (march, refine, fill, per-item
code) without a GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g


def device_asm(defs, out='/tmp/_dsdf_dev.s'):
    flags = [f for f in g.HIPCC_FLAGS if f not in ('-fPIC', '-shared')]
    src = os.path.join(g.PKG, 'csrc', 'dsdf_kernels.hip')
    r = subprocess.run(['/opt/rocm/bin/hipcc'] + flags + list(defs) + ['--cuda-device-only', '-S', '-o', out, src],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd='/tmp')
    if r.returncode:
        sys.exit(r.stdout)
    return open(out).read().splitlines()


def kernel_lines(lines, want):
    names = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            names[i] = m.group(1)
    dem = subprocess.run(['c++filt'] + list(names.values()), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for (i, n), d in zip(names.items(), dem):
        if want in d:
            j = i
            while 's_endpgm' not in lines[j]:
                j += 1
            return d, lines[i:j + 1]
    sys.exit(f'kernel {want} not found')


def blocks(body):
    out, cur = [], None
    for n, l in enumerate(body):
        m = re.match(r'^(\.LBB\w+):|^; %bb\.(\d+):', l)
        if m:
            cur = dict(name=m.group(1) or 'bb.' + m.group(2), line=n, valu=0, salu=0, lds=0, vmem=0, depth='')
            d = re.search(r'Depth=(\d)', l)
            if d:
                cur['depth'] = d.group(1)
            out.append(cur)
            continue
        if cur is None:
            cur = dict(name='entry', line=n, valu=0, salu=0, lds=0, vmem=0, depth='')
            out.append(cur)
        t = l.strip().split(' ')[0] if l.strip() else ''
        if t.startswith('v_'):
            cur['valu'] += 1
        elif t.startswith('s_'):
            cur['salu'] += 1
        elif t.startswith('ds_'):
            cur['lds'] += 1
        elif t.split('_')[0] in ('global', 'buffer', 'scratch', 'flat'):
            cur['vmem'] += 1
    return out


if __name__ == '__main__':
    a = sys.argv[1:]
    defs = [x for x in a if x.startswith('-D')]
    want = a[a.index('--kernel') + 1]
    mn = int(a[a.index('--min') + 1]) if '--min' in a else 8
    name, body = kernel_lines(device_asm(defs), want)
    if '--dump' in a:
        open(a[a.index('--dump') + 1], 'w').write('\n'.join(body) + '\n')
    bl = blocks(body)
    print(name[:100])
    print('total valu', sum(b['valu'] for b in bl), 'salu', sum(b['salu'] for b in bl), 'lds', sum(b['lds'] for b in bl),
          'vmem', sum(b['vmem'] for b in bl))
    for b in bl:
        if b['valu'] + b['salu'] >= mn:
            print(f"{b['name']:12s} line {b['line']:5d} depth {b['depth']:1s} valu {b['valu']:4d} salu {b['salu']:4d} lds {b['lds']:3d} vmem {b['vmem']:3d}")
