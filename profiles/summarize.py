#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/...) into the small summaries committed here.

  python profiles/summarize.py <tag> <kernel_stats.csv> <kernel_trace.csv> [<fetch_counter.csv> <write_counter.csv>]

Writes profiles/<tag>_kernel_stats.csv (top kernels of `rocprofv3 --kernel-trace --stats`),
profiles/<tag>_dispatches.csv (per-dispatch durations of the library's kernels) and, when the
two PMC passes are given, profiles/<tag>_traffic.json with HBM bytes per launch of every
library kernel: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE reads half the bytes of
wide coalesced streams on gfx950 (MI355X_MICROARCH.md, HBM section; calibrated here on
k_pad_grid: 2 x 38170 KB ~ 64 MiB grid + apron reads, WRITE_SIZE 70253 KB = 262^3 x 4 B exactly).
"""
import collections
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


def main():
    tag, stats, trace = sys.argv[1:4]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(HERE, f'{tag}_kernel_stats.csv'), 'w') as f:
        f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n')
        for r in rows[:12]:
            f.write(f"{short(r['Name'])[:70]},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},"
                    f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    d = collections.defaultdict(list)
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r) for r in csv.DictReader(open(trace))]
    renders = [(a, b) for a, b, n, _ in ev if n.startswith('k_render_items')]
    for a, b, n, r in ev:
        if n.startswith('k_'):
            key = f"{n} grid={r['Grid_Size_X']}x{r.get('Grid_Size_Y', '1')}"
            # a render / tail kernel that shares the chip with ANOTHER render kernel for more than a tenth of its run time (the
            # two-stream step: primal render beside gradient sweep) is stretched by it: such dispatches get their own rows
            shared = 0
            if n.startswith(('k_render_items', 'k_tail_trace')):
                shared = sum(max(0, min(b, d1) - max(a, c1)) for c1, d1 in renders if (c1, d1) != (a, b))
            d[key].append(((b - a) / 1e6, shared > 0.1 * (b - a)))
    with open(os.path.join(HERE, f'{tag}_dispatches.csv'), 'w') as f:
        f.write('kernel,dispatches,median_ms,min_ms,max_ms\n')
        for k, vv in d.items():
            for label, v in (('', [x for x, sh in vv if not sh]), (' [beside another render kernel]', [x for x, sh in vv if sh])):
                if not v:
                    continue
                s = sorted(v)
                kk = k + label
                f.write(f"{kk},{len(v)},{s[len(s) // 2]:.3f},{s[0]:.3f},{s[-1]:.3f}\n")
                # the persistent-worker kernels have ONE grid for every workload (bench launches of 12 views x 256 spp and the
                # single-view 64-spp target renders of the set-up share a row above): split the dispatches where consecutive
                # sorted durations jump by more than 3x, so that each workload has its own line
                cuts = [i for i in range(1, len(s)) if s[i] > 3.0 * s[i - 1]]
                if cuts:
                    for a, b in zip([0] + cuts, cuts + [len(s)]):
                        c = s[a:b]
                        f.write(f"{kk} [durations {c[0]:.2f}-{c[-1]:.2f} ms],{len(c)},{c[len(c) // 2]:.3f},{c[0]:.3f},{c[-1]:.3f}\n")
    if len(sys.argv) >= 6:
        out = {}
        for col, path in (('FETCH_SIZE', sys.argv[4]), ('WRITE_SIZE', sys.argv[5])):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                if r['Counter_Name'] == col and short(r['Kernel_Name']).startswith('k_'):
                    agg[f"{short(r['Kernel_Name'])} grid={r['Grid_Size']}"].append(float(r['Counter_Value']))
            for k, v in agg.items():
                s = sorted(v)
                out.setdefault(k, {})[col + '_KB'] = s[0]     # min over dispatches: excludes the bench's stats-enabled launch
        for k, v in out.items():
            if 'FETCH_SIZE_KB' in v and 'WRITE_SIZE_KB' in v:
                v['hbm_bytes_per_launch'] = (2 * v['FETCH_SIZE_KB'] + v['WRITE_SIZE_KB']) * 1024
        json.dump(out, open(os.path.join(HERE, f'{tag}_traffic.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
