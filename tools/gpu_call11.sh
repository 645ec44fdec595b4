#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/c11; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_overlap.json 2> $O/bench_overlap.err; python -c "
import json; d=json.load(open('$O/bench_overlap.json')); print('overlap(prio)', d['value'], d['ms_per_step'], d['low_spp']['value'], d['roofline']['frac'])"; tail -2 $O/bench_overlap.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-low-spp > $O/trace.log 2>&1; python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/c11/trace/t_kernel_trace.csv')) if r['Kernel_Name'].startswith(('void k_','k_'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# print the last overlapped step's timeline
t0=None
sel=[r for r in rows if 'k_render_items' in r['Kernel_Name'] or 'k_tail' in r['Kernel_Name'] or 'k_backward' in r['Kernel_Name']]
for r in sel[-14:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print('%-45s start %8.3f ms  dur %7.3f ms  stream %s' % (r['Kernel_Name'].split('(')[0][-45:], (s-t0)/1e6, (e-s)/1e6, r.get('Stream_Id', r.get('Queue_Id'))))
PY
