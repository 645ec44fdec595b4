#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05f; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-low-spp --no-direct --no-opt-iteration --emulate-rank 6/8 > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
k=$(find $O/tr -name "t_kernel_trace.csv" | head -1)
python - "$k" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:46], r['Queue_Id']) for r in rows)
# last step: find the last 2 k_backward_apply clusters
marks = [i for i, e in enumerate(ev) if e[2].startswith('k_backward_apply')]
# steps are separated by gaps: take events after the apply that precedes the last group of applies
last = marks[-1]
# walk back to find previous step's final apply: an apply followed by a k_pixel_skip later
i = last
while i > 0 and not (ev[i][2].startswith('k_backward_apply') and any(e[2].startswith('k_pixel_skip') for e in ev[i+1:last])):
    i -= 1
t0 = ev[i][1]
print('last emulated step: %.2f ms' % ((ev[last][1] - t0) / 1e6))
for e in ev[i+1:last+1]:
    if e[1] - e[0] > 20000 or e[2].startswith('k_'):
        print('%8.3f %8.3f %7.3f q%s %s' % ((e[0]-t0)/1e6, (e[1]-t0)/1e6, (e[1]-e[0])/1e6, e[3], e[2]))
PY
find $O -name "*.csv" -size +1M -delete
