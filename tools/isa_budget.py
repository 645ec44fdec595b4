#!/usr/bin/env python3
"""VALU budget of a render kernel BY SOURCE FUNCTION.  The device code is built with debug info, disassembled
(llvm-objdump) and every vector-ALU instruction is attributed through its INLINE STACK (llvm-symbolizer --inlines: the chain of
functions inlined at that address, innermost first) to the first frame that names a function of csrc/ -- fmaf / __shfl / atomics from the
HIP headers go to their caller -- and that function to a class (spline FMAs, B-spline weights, address + clamp, cell-cache protocol,
march logic, trace weight, sampler, camera ray + box, film weights, film window reduce, hand-off, queue ...).  Loop depth comes from
the backward branches of the kernel: depth >= 2 = the march / refine / cache-fill loops (executed per wave iteration), depth 1 = the
item loop (once per 64-sample chunk; alternative paths each counted once), depth 0 = once per worker.

    python tools/isa_budget.py > profiles/r06_isa_budget.md

STATIC counts (every instruction of the binary once; debug info does not change the -O3 code).  The render kernels contain several
copies of the march loop (with / without hand-off, the refinement loop of A5, the per-lane fallback of the cell cache), so the
depth->=2 totals are a multiple of one iteration; the SHARES are what the table is for.  The dynamic totals measured with
SQ_INSTS_VALU (profiles/valu_model.json) are printed beside them."""
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

CSRC = os.path.join(g.PKG, 'csrc')
LLVM = '/opt/rocm/lib/llvm/bin'

CLASSES = [   # regex on the demangled function name of an inline frame -> class
    (r'eval_cubic_rows|eval_cubic<|eval_value|LdsRows::get|RegRows::get|GlobalRows::get', 'spline FMAs (64 + 16 + 4 per value; 328 with gradient + Hessian)'),
    (r'bspline_(w|dw|ddw)', 'B-spline weights'),
    (r'cubic_cell|cubic_setup|to_grid|global_rows|iclamp', 'address + clamp'),
    (r'WaveCellCache|ReuseFetch|DirectFetch', 'cell-cache protocol / fetch policy (grouping loop, fills, slot reads)'),
    (r'eval_trace_weight|bbox_distance_inside', 'trace weight (eval_trace_weight)'),
    (r'plain_march_|trace_plain|trace_diff|diff_march_|refine_hit|MarchToEnd|clear_trace', 'march logic (step, warp accumulators, refine control)'),
    (r'sampler_|sample_tea_32|pcg32_|sample_offsets', 'sampler (sample_tea_32 + PCG32)'),
    (r'camera_ray|bbox_ray_intersect|box_lo|box_hi|box_face_distance|closest_axis|lane_setup|lane_pixel|reproject', 'camera ray + box + re-projection'),
    (r'gauss_(f|exp|df)', 'film weights (Gaussian)'),
    (r'film_accum_|film_flush_wave|splat_', 'film window: products, LDS transpose, reduce, flush'),
    (r'HandOffCtl|tail_reserve|tail_subq|xcc_id', 'tail hand-off'),
    (r'queue_unit|store_record|warp_weight_positive|view_queue', 'backward queue (weight test, compaction, records)'),
    (r'shade_value|direct_value', 'shading value'),
    (r'add_stats|flush_stats', 'statistics'),
    (r'k_render_items|k_render_pass', 'kernel body (tickets, item decode, proofs, branches)'),
]
SKIP = re.compile(r'^(__|fmaf|fminf|fmaxf|fabsf|floorf|sqrtf|expf|min\b|max\b|atomic|hip|operator|(float |int |unsigned int |void )?(dsdf::)?(mk|dot|fma3|operator|splat2|mk2|drsign|rcpf|rsqf|rcpf_s|rsqf_s|wave_\w+|lane_id|mask_prefix|opaque|iclamp)\b)')


def build():
    out = '/tmp/_dsdf_dbg.o'
    flags = [f for f in g.HIPCC_FLAGS if f not in ('-fPIC', '-shared')]
    r = subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['-g', '--cuda-device-only', '--no-gpu-bundle-output', '-c', '-o', out,
                                                         os.path.join(CSRC, 'dsdf_kernels.hip')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd='/tmp')
    if r.returncode:
        sys.exit(r.stdout)
    return out


def disassemble(obj):
    """{demangled kernel name: [(address, mnemonic, branch target offset within the kernel or None)]}"""
    txt = subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', '-C', obj], stdout=subprocess.PIPE, text=True).stdout
    kernels, cur = {}, None
    for l in txt.splitlines():
        m = re.match(r'^([0-9a-f]+) <(.*)>:$', l)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            continue
        m = re.match(r'^\s+(\S+).*//\s*([0-9A-Fa-f]+):(.*)$', l)      # "\tv_fma_f32 ...   // 000000001234: [<sym+0xoff>]"
        if m and cur is not None:
            t = re.search(r'\+0x([0-9a-f]+)>\s*$', m.group(3))
            cur.append((int(m.group(2), 16), m.group(1), int(t.group(1), 16) if t else None))
    return kernels


def inline_stacks(obj, addrs):
    p = subprocess.run([f'{LLVM}/llvm-symbolizer', f'--obj={obj}', '--inlines', '--functions=linkage', '-C', '--output-style=LLVM'] +
                       [hex(a) for a in addrs], stdout=subprocess.PIPE, text=True)
    stacks, cur = [], []
    for l in p.stdout.splitlines():
        if not l.strip():
            if cur:
                stacks.append(cur); cur = []
            continue
        if not re.match(r'^\S.*:\d+:\d+$', l) and not l.startswith('/') and not l.startswith('??:'):
            cur.append(l.strip())
    if cur:
        stacks.append(cur)
    return stacks


def classify(stack):
    for fn in stack:
        if SKIP.match(fn):
            continue
        for pat, cls in CLASSES:
            if re.search(pat, fn):
                return cls
        return f'other ({fn.split("(")[0][:40]})'
    return 'unattributed'


def main():
    want = sys.argv[1:] or ['k_render_items<false, false, false>', 'k_render_items<true, false, false>']
    obj = build()
    kernels = disassemble(obj)
    model = None
    try:
        model = json.load(open(os.path.join(ROOT, 'profiles', 'valu_model.json')))
    except (OSError, ValueError):
        pass
    print("# Instruction budget of the render kernels by source function (`tools/isa_budget.py`)\n")
    print(' '.join(__doc__.split('\n\n')[0].split()) + '\n')
    print(' '.join(__doc__.split('\n\n')[2].split()) + '\n')
    names = {2: 'depth >= 2: the march / refine / cache-fill loops (per wave iteration)',
             1: 'depth 1: the ITEM loop (per 64-sample chunk; alternative paths counted once each)', 0: 'depth 0 (once per worker)'}
    for w in want:
        name = next((k for k in kernels if k.startswith('void ' + w) or k.startswith(w)), None)
        if name is None:
            sys.exit(f'kernel {w} not found')
        ins = kernels[name]
        # loops: backward branches (target address <= branch address)
        base = ins[0][0]
        heads = {}
        for a, mn, t in ins:                                  # one loop per HEADER: several back edges (`continue`) share it
            if mn.startswith(('s_cbranch', 's_branch')) and t is not None and base + t <= a:
                heads[base + t] = max(heads.get(base + t, 0), a)
        loops = sorted(heads.items())
        valu = [(a, mn) for a, mn, _ in ins if mn.startswith('v_')]
        stacks = inline_stacks(obj, [a for a, _ in valu])
        if len(stacks) != len(valu):
            sys.exit(f'symbolizer returned {len(stacks)} stacks for {len(valu)} instructions')
        counts = defaultdict(lambda: defaultdict(int))
        other = defaultdict(lambda: defaultdict(int))
        # the block layout of the item loop is not reducible to nested intervals (its back edges and `continue`s interleave); the loops
        # that span most of the kernel are the item loop, every SHORTER loop (march, refine, fills, grouping) counts as depth 2
        span = ins[-1][0] - base
        inner = [(lo, hi) for lo, hi in loops if hi - lo < 0.5 * span]
        outer = [(lo, hi) for lo, hi in loops if hi - lo >= 0.5 * span]
        depth_of = lambda a: 2 if any(lo <= a <= hi for lo, hi in inner) else (1 if any(lo <= a <= hi for lo, hi in outer) else 0)
        for (a, mn), st in zip(valu, stacks):
            counts[depth_of(a)][classify(st)] += 1
        for a, mn, _ in ins:
            d = depth_of(a)
            if mn.startswith('s_'):
                other[d]['SALU'] += 1
            elif mn.startswith('ds_'):
                other[d]['LDS'] += 1
            elif mn.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                other[d]['VMEM'] += 1
        print(f"## `{w}`\n")
        key = 'primal' if w.startswith('k_render_items<false, false') else ('sweep' if w.startswith('k_render_items<true, false') else None)
        if model and key in model:
            m = model[key]
            print("Dynamic (SQ_INSTS_VALU, profiles/valu_model.json, tag %s): " % model.get('tag') + ', '.join(
                f"{n} = {m[n]:.1f}" for n in ('valu_per_wave_iteration', 'valu_per_chunk', 'valu_per_wave_step') if n in m) + '\n')
        whole = defaultdict(int)
        for d in counts:
            for cls, n in counts[d].items():
                whole[cls] += n
        tot = sum(whole.values())
        print(f"**whole kernel** -- {tot} VALU instructions\n\n| class | VALU | share | in the short loops | item loop |\n|---|---|---|---|---|")
        for cls, n in sorted(whole.items(), key=lambda kv: -kv[1]):
            print(f"| {cls} | {n} | {100.0 * n / max(tot, 1):.0f} % | {counts[2].get(cls, 0)} | {counts[1].get(cls, 0)} |")
        print()
        for d in ():
            tot = sum(counts[d].values())
            print(f"**{names[d]}** -- {tot} VALU, {other[d]['SALU']} SALU, {other[d]['LDS']} LDS, {other[d]['VMEM']} VMEM instructions\n")
            print("| class | VALU | share |\n|---|---|---|")
            for cls, n in sorted(counts[d].items(), key=lambda kv: -kv[1]):
                print(f"| {cls} | {n} | {100.0 * n / max(tot, 1):.0f} % |")
            print()


if __name__ == '__main__':
    main()
