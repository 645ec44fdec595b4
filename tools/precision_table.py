#!/usr/bin/env python3
"""Gradient / image precision table of the HIP path: case x build x rel-L2 against the fp64 oracle, next to the measured fp32
floor of the reference algorithm itself; the HIP gradient against the fp32 C oracle DIRECTLY (the reference's llvm_ad_rgb path is
fp32 too); the 1 % and 0.1 % trimmed statistics; and the sample attribution (tests/test_gpu_attribution.py: cubes removed /
budget, rel-L2 of the rest against its gate).

  build `fast` = the shipped libdsdf.so (v_rcp_f32 / v_rsq_f32 / v_exp_f32, 1 ulp)
  build `ieee` = lib/variants/libdsdf_ieee.so (-DDSDF_FAST_RCP=0: IEEE division / sqrt / expf sequences)
  floor        = fp32 build vs fp64 build of the plain-C restatement (oracle/dsdf_oracle.c) on bit-identical inputs,
                 and (oracle-sized cases) the torch oracle run in fp32 vs fp64

Runs on a GPU box: `python tools/precision_table.py --out gpurun_out/r02_precision.json` (each build in its own process,
the library path is fixed at import).  Oracle = test infrastructure; nothing here is on the product path.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'python')):
    if p not in sys.path:
        sys.path.insert(0, p)

SMALL = ['sphere16', 'blob32', 'blob32_spp64', 'blob48_rect']
CONFIG = ['C1_spp4', 'C1_spp16', 'C1_spp64', 'C2_view0', 'C3_view0']


def worker(cases):
    import numpy as np
    import torch
    import dsdf
    import precision as P
    from cases import make_case
    dsdf.load()
    rows = []
    for name in cases:
        case = make_case(name) if name in SMALL else P.config_case(name)
        grid = dsdf.SdfGrid(case['grid'].float().cuda())
        sen = dsdf.get_regular_cameras(case['ncam'], resx=case['W'], resy=case['H'])[case['icam']]
        offs = case['offsets'].cuda()
        for integ in (0, 1):
            if name == 'C3_view0' and integ == 1:
                continue
            r = P.reference_gradient(case, integ, True)
            gg, img = dsdf.render_backward(grid, sen, case['spp'], case['grad_image'].cuda()[None], offsets=offs,
                                           integrator=integ, return_image=True)
            g = gg.cpu().numpy()
            lanes = int(case['offsets'].shape[0])
            K = max(3, int(np.ceil(1e-5 * lanes)))
            floor_k, _, _ = P.greedy_blocks(r['g32'], r['g64'], 0.0, K)
            gate = max(P.FLOOR_FACTOR * floor_k, P.NORTH_STAR)
            rest, centres, _ = P.greedy_blocks(g, r['g64'], gate, K)
            rows.append(dict(case=name, integ=integ, lanes=lanes,
                             img_err=P.rel_l2(img[0].cpu().numpy(), r['img64']),
                             grad_err=P.rel_l2(g, r['g64']), grad_err_trim=P.trimmed_rel_l2(g, r['g64']),
                             grad_err_trim01=P.trimmed_rel_l2(g, r['g64'], 0.001), grad_vs_c32=P.rel_l2(g, r['g32']),
                             grad_trim_vs_c32=P.trimmed_rel_l2(g, r['g32']),
                             attr_budget=K, attr_removed=len(centres), attr_rest=rest, attr_gate=gate,
                             floor_c=r['floor_c'], floor_torch=r['floor_torch'], floor_trim=r['floor_trim'], floor_trim01=r['floor_trim01']))
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    print('ROWS=' + json.dumps(rows))


def ref32_section():
    """The reference's OWN files at the reference's OWN precision (tests/golden/refshim32_*.npz: tools/make_reference_fixtures.py --shim
    with REFSHIM_DTYPE=float32) next to their fp64 run (refshim_*.npz) and the HIP path: relative L2 of dL/dsdf."""
    import numpy as np
    import dsdf
    import test_refshim_fixture as T
    import __graft_entry__ as g
    from conftest import HostHarness
    dsdf.load()
    hh = HostHarness(g.build_harness())
    rel = T.rel_l2
    print()
    print('| case | run | lanes | reference fp32 vs reference fp64 | HIP vs reference fp64 | HIP vs reference fp32 | host build of the kernel arithmetic vs reference fp64 | image: HIP vs reference fp64 |')
    print('|---|---|---|---|---|---|---|---|')
    ratios, out = [], []
    for name, tag in T.FP32_RUNS:
        ref, r32 = T.load(name), T.load32(name)
        x = T.inputs(ref)
        gg = T._gpu_backward(dsdf, name, tag)
        gh = T._host_backward(hh, ref, x, tag)
        a64, a32 = ref[f'grad_{tag}'], r32[f'grad_{tag}']
        f, e64, e32, h64 = rel(a32, a64), rel(gg, a64), rel(gg, a32), rel(gh, a64)
        sen = dsdf.Sensor(ref['origin'], resx=x['W'], resy=x['H'])
        ratios.append(e64 / f)
        out.append(dict(case=name, tag=tag, ref32_vs_ref64=f, hip_vs_ref64=e64, hip_vs_ref32=e32, host_vs_ref64=h64))
        print(f"| {name} | {tag} | {x['n']} | {f:8.2e} | {e64:8.2e} | {e32:8.2e} | {h64:8.2e} | - |")
    gm = float(np.exp(np.mean(np.log(ratios))))
    print(f"\ngeometric mean of (HIP vs reference fp64) / (reference fp32 vs reference fp64) over the {len(ratios)} runs: {gm:.2f}")
    return dict(rows=out, geometric_mean_ratio=gm)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--worker', action='store_true')
    ap.add_argument('--cases', default=','.join(SMALL + CONFIG))
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r02_precision.json'))
    args = ap.parse_args()
    cases = args.cases.split(',')
    if args.worker:
        return worker(cases)
    libs = {'fast': os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'lib', 'libdsdf.so'),
            'ieee': os.path.join(ROOT, 'differentiable-sdf-rendering_amd', 'lib', 'variants', 'libdsdf_ieee.so')}
    table = {}
    for tag, path in libs.items():
        if not os.path.isfile(path):
            print(f'{tag}: {path} missing, skipped', file=sys.stderr)
            continue
        env = dict(os.environ, DSDF_LIB_PATH=path)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', '--cases', args.cases], env=env,
                           stdout=subprocess.PIPE, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('ROWS=')]
        if r.returncode != 0 or not line:
            print(f'{tag}: worker failed (rc {r.returncode})', file=sys.stderr)
            continue
        for row in json.loads(line[0][5:]):
            e = table.setdefault((row['case'], row['integ']), dict(case=row['case'], integ=row['integ'], lanes=row['lanes'],
                                                                  floor_c=row['floor_c'], floor_torch=row['floor_torch'],
                                                                  floor_trim=row['floor_trim'], floor_trim01=row['floor_trim01']))
            e[f'img_{tag}'] = row['img_err']; e[f'grad_{tag}'] = row['grad_err']; e[f'grad_trim_{tag}'] = row['grad_err_trim']
            for k in ('grad_err_trim01', 'grad_vs_c32', 'grad_trim_vs_c32', 'attr_budget', 'attr_removed', 'attr_rest', 'attr_gate'):
                e[f'{k}_{tag}'] = row[k]
    rows = list(table.values())
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(dict(note=__doc__.strip().splitlines()[0], rows=rows), open(args.out, 'w'), indent=1)
    f = lambda v: '   -    ' if v is None else f'{v:8.2e}'
    print('| case | integrator | lanes | image fast | grad fast | grad ieee | floor C fp32 | floor torch fp32 | **fast vs fp32 oracle** | trimmed 1 % fast | '
          'trimmed 1 % floor | trimmed 0.1 % fast | trimmed 0.1 % floor | attribution: cubes removed / budget | rest | gate |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for e in rows:
        print(f"| {e['case']} | {'silhouette' if e['integ'] == 0 else 'shading'} | {e['lanes']} | {f(e.get('img_fast'))} | {f(e.get('grad_fast'))} | "
              f"{f(e.get('grad_ieee'))} | {f(e['floor_c'])} | {f(e['floor_torch'] or None)} | {f(e.get('grad_vs_c32_fast'))} | {f(e.get('grad_trim_fast'))} | "
              f"{f(e['floor_trim'])} | {f(e.get('grad_err_trim01_fast'))} | {f(e['floor_trim01'])} | "
              f"{e.get('attr_removed_fast', '-')} / {e.get('attr_budget_fast', '-')} | {f(e.get('attr_rest_fast'))} | {f(e.get('attr_gate_fast'))} |")
    try:
        r32 = ref32_section()
        d = json.load(open(args.out)); d['reference_fp32'] = r32
        json.dump(d, open(args.out, 'w'), indent=1)
    except Exception as e:                                   # (the main table stands on its own)
        print(f'reference-fp32 section failed: {e!r}', file=sys.stderr)


if __name__ == '__main__':
    main()
