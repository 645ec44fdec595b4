#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/pmc_workload.py (one primal + one gradient-pass launch of the bench
scene; separate passes, <= 8 SQ counters each, never combined with tracing domains other than --kernel-trace) into
profiles/<tag>_sq.json: per library kernel the per-launch counter values, launch duration in each pass, and the derived
figures the VALU-issue roofline uses.  Per kernel: `dispatches` = its launches in call order (tools/pmc_workload.py's manifest), the
top-level counters / `derived` = its FIRST launch.

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
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel -> dispatch -> counter
        dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            if not k.startswith('k_'):
                continue
            d = int(r['Dispatch_Id'])
            per[k][d][r['Counter_Name']] += float(r['Counter_Value'])
            dur[k][d] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
            res[k].setdefault('vgpr_alloc', int(r['VGPR_Count'])); res[k].setdefault('lds_bytes', int(r['LDS_Block_Size']))
            res[k].setdefault('grid', int(r['Grid_Size'])); res[k].setdefault('workgroup', int(r['Workgroup_Size']))
        for k in per:
            order = sorted(per[k])                                   # launch order = the call order of tools/pmc_workload.py
            n = len(order)
            lst = res[k].setdefault('dispatches', [dict() for _ in range(n)])
            while len(lst) < n:
                lst.append(dict())
            for i, d in enumerate(order):
                lst[i].update(per[k][d])
                lst[i].setdefault('ms', {})[p] = round(dur[k][d], 4)
            # top level: the FIRST dispatch (call 0 / 2 / 3 ... of the workload: the shipped path of every kernel)
            res[k].update(per[k][order[0]])
            res[k].setdefault('launch_ms', {})[p] = round(dur[k][order[0]], 4)
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
            if 'TCC_HIT_sum' in v and 'TCC_MISS_sum' in v:
                d['l2_miss_rate'] = v['TCC_MISS_sum'] / max(v['TCC_HIT_sum'] + v['TCC_MISS_sum'], 1.0)
    json.dump(res, open(os.path.join(HERE, f'{tag}_sq.json'), 'w'), indent=1, sort_keys=True)
    if len(sys.argv) > 3:
        st = json.load(open(sys.argv[3]))
        model = {'tag': tag, 'source': f'profiles/{tag}_sq.json',
                 'workload': 'tools/pmc_workload.py: bench scene, %d views x 512^2, spp %d/%d; calls %s' % (st['views'], st['spp'][0], st['spp'][1], st.get('calls'))}

        def valu(kern, i=0):
            v = res.get(kern)
            if not v or len(v.get('dispatches', [])) <= i or 'SQ_INSTS_VALU' not in v['dispatches'][i]:
                return None
            return v['dispatches'][i]['SQ_INSTS_VALU']

        def ratio(name, kern, key, i=0):
            V = valu(kern, i)
            if V is None or key not in st:
                return
            v = res[kern]
            m = {'kernel': kern, 'valu_insts_per_launch': V, 'wave_steps_per_launch': st[key]['wave_steps'],
                 'valu_per_wave_step': V / max(st[key]['wave_steps'], 1), 'launch_ms': sum(v['dispatches'][i]['ms'].values()) / len(v['dispatches'][i]['ms'])}
            dd = v['dispatches'][i]
            if 'FETCH_SIZE' in dd and 'WRITE_SIZE' in dd:
                m['hbm_bytes_per_launch'] = (2 * dd['FETCH_SIZE'] + dd['WRITE_SIZE']) * 1024
            model[name] = m

        # primal render kernel: the hit proof removes the march of whole chunks but not their set-up / film code, so one ratio
        # no longer describes it -- VALU = a x wave iterations + b x chunks (64 generated samples), a and b from the two primal calls
        # of the workload (hit proof on: dispatch 0, off: dispatch 1), whose wave iterations and chunk counts the kernel counted itself
        kp = 'k_render_items<false, false, false>'
        ratio('primal', kp, 'primal')
        V0, V1 = valu(kp, 0), valu(kp, 1)
        if V0 and V1 and 'primal_no_hit_proof' in st:
            W0, C0 = st['primal']['wave_steps'], st['primal']['lanes'] / 64.0
            W1, C1 = st['primal_no_hit_proof']['wave_steps'], st['primal_no_hit_proof']['lanes'] / 64.0
            det = W0 * C1 - W1 * C0
            if abs(det) > 0:
                a = (V0 * C1 - V1 * C0) / det
                b = (V0 - a * W0) / C0
                model['primal'].update({'valu_per_wave_iteration': a, 'valu_per_chunk': b,
                                        'fit': {'hit_proof_on': {'valu': V0, 'wave_steps': W0, 'chunks': C0},
                                                'hit_proof_off': {'valu': V1, 'wave_steps': W1, 'chunks': C1}}})
        ratio('sweep', 'k_render_items<true, false, false>', 'grad')
        ratio('low_primal', 'k_render_pass<false, false>', 'low_primal')
        ratio('low_sweep', 'k_render_pass<true, false>', 'low_grad')
        ratio('direct_primal', 'k_render_items_store<false, false>', 'direct_primal')   # (round 6: the primary march of the wavefront primal)
        ratio('direct_sweep', 'k_render_items_store<true, false>', 'direct_grad')
        json.dump(model, open(os.path.join(HERE, 'valu_model.json'), 'w'), indent=1, sort_keys=True)
        print('valu_model', json.dumps(model))
    for k, v in res.items():
        if 'derived' in v:
            print(k, json.dumps({a: round(b, 4) for a, b in v['derived'].items()}), v['launch_ms'])


if __name__ == '__main__':
    main()
