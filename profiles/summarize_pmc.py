#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/pmc_workload.py (one primal + one gradient-pass launch of the bench
scene; separate passes, <= 8 SQ counters each, never combined with tracing domains other than --kernel-trace) into
profiles/<tag>_sq.json: per library kernel the per-launch counter values, launch duration in each pass, and the derived
figures the VALU-issue roofline uses.

  python profiles/summarize_pmc.py <tag> <dir with pmc_*/ subdirs> [<pmc_stats.json written by tools/pmc_workload.py>]

With the statistics file it also writes profiles/valu_model.json: VALU instructions per lock-step wave iteration of the
primal render kernel and of the gradient sweep (SQ_INSTS_VALU of the one profiled call / the wave iterations the kernel counted
on the same inputs) and the measured HBM bytes per call -- the only constants bench.py's roofline block uses.

Derived (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, wave64 VALU instruction = 2 clk; SQ_* are summed over the 32 SEs;
SQ_CYCLES / 32 = shader clock ticks; SQ_WAVE_CYCLES in quad-cycles):
  clock_GHz        = SQ_CYCLES / 32 / duration
  valu_util        = SQ_INSTS_VALU * 2 / (1024 * SQ_CYCLES / 32)             (fraction of the VALU issue slots used)
  lane_util        = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU)
  waves_per_simd   = 4 * SQ_WAVE_CYCLES / (1024 * SQ_CYCLES / 32)
  lds_busy         = SQ_LDS_IDX_ACTIVE / (256 * SQ_CYCLES / 32)
  hbm_bytes        = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (FETCH_SIZE counts half of wide reads on gfx950)
"""
import collections
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(n):
    return n.split('(')[0].replace('void ', '').strip()


def main():
    tag, root = sys.argv[1:3]
    res = collections.defaultdict(dict)
    for path in sorted(glob.glob(os.path.join(root, 'pmc_*', '*_counter_collection.csv'))):
        p = os.path.basename(os.path.dirname(path))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            if not k.startswith('k_'):
                continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
            res[k].setdefault('vgpr_alloc', int(r['VGPR_Count'])); res[k].setdefault('lds_bytes', int(r['LDS_Block_Size']))
            res[k].setdefault('grid', int(r['Grid_Size'])); res[k].setdefault('workgroup', int(r['Workgroup_Size']))
        for k in agg:
            n = len(disp[k])
            for c, v in agg[k].items():
                res[k][c] = v / n
                res[k].setdefault('per_call', {})[c] = v          # summed over the dispatches of the (one) library call
            res[k].setdefault('launch_ms', {})[p] = round(sum(disp[k].values()) / n, 4)
            res[k]['dispatches_per_pass'] = n
    for k, v in res.items():
        ms = list(v.get('launch_ms', {}).values())
        if 'SQ_CYCLES' in v and ms:
            ticks = v['SQ_CYCLES'] / 32.0
            t = v['launch_ms'].get('pmc_c', ms[0]) * 1e-3
            d = v.setdefault('derived', {})
            d['clock_GHz'] = ticks / t / 1e9
            if 'SQ_INSTS_VALU' in v:
                d['valu_util'] = v['SQ_INSTS_VALU'] * 2.0 / (1024 * ticks)
                d['valu_insts_per_wave'] = v['SQ_INSTS_VALU'] / max(v.get('SQ_WAVES', 1), 1)
            if 'SQ_THREAD_CYCLES_VALU' in v and 'SQ_INSTS_VALU' in v:
                d['lane_util'] = v['SQ_THREAD_CYCLES_VALU'] / (64.0 * v['SQ_INSTS_VALU'])
            if 'SQ_WAVE_CYCLES' in v:
                d['waves_per_simd'] = 4.0 * v['SQ_WAVE_CYCLES'] / (1024 * ticks)
            if 'SQ_LDS_IDX_ACTIVE' in v:
                d['lds_busy'] = v['SQ_LDS_IDX_ACTIVE'] / (256 * ticks)
            if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
                d['hbm_bytes'] = (2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024
                d['hbm_GBps'] = d['hbm_bytes'] / t / 1e9
    json.dump(res, open(os.path.join(HERE, f'{tag}_sq.json'), 'w'), indent=1, sort_keys=True)
    if len(sys.argv) > 3:
        st = json.load(open(sys.argv[3]))
        model = {'tag': tag, 'source': f'profiles/{tag}_sq.json', 'workload': 'tools/pmc_workload.py: bench scene, %d views x 512^2, spp %d/%d' % (st['views'], st['spp'][0], st['spp'][1])}
        for name, kern, key in (('primal', 'k_render_items<false, false, false>', 'primal'), ('sweep', 'k_render_items<true, false, false>', 'grad')):
            v = res.get(kern)
            if not v or 'SQ_INSTS_VALU' not in v.get('per_call', {}):
                continue
            valu, ws = v['per_call']['SQ_INSTS_VALU'], st[key]['wave_steps']
            m = {'kernel': kern, 'valu_insts_per_launch': valu, 'wave_steps_per_launch': ws, 'valu_per_wave_step': valu / ws,
                 'launch_ms': sum(v.get('launch_ms', {}).values()) / max(len(v.get('launch_ms', {})), 1) * v.get('dispatches_per_pass', 1)}
            if 'FETCH_SIZE' in v['per_call'] and 'WRITE_SIZE' in v['per_call']:
                m['hbm_bytes_per_launch'] = (2 * v['per_call']['FETCH_SIZE'] + v['per_call']['WRITE_SIZE']) * 1024
            model[name] = m
        json.dump(model, open(os.path.join(HERE, 'valu_model.json'), 'w'), indent=1, sort_keys=True)
        print('valu_model', json.dumps(model))
    for k, v in res.items():
        if 'derived' in v:
            print(k, json.dumps({a: round(b, 4) for a, b in v['derived'].items()}), v['launch_ms'])


if __name__ == '__main__':
    main()
