#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05o; mkdir -p $O
python tools/ab_low.py | grep "^AB"
AB_SPP=16 AB_SPP_GRAD=4 python tools/ab_low.py | grep "^AB"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/ab_low.py > $O/tr.log 2>&1
k=$(find $O/tr -name "t_kernel_trace.csv" | head -1)
python - "$k" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r['Queue_Id'], r['VGPR_Count'], r['Grid_Size_X']) for r in rows)
marks = [i for i, e in enumerate(ev) if e[2].startswith('k_backward_apply')]
a, b = marks[-6], marks[-5]
t0 = ev[a][1]
print('one 4/1 step: %.3f ms' % ((ev[b][1] - t0) / 1e6))
for e in ev[a+1:b+1]:
    print('%8.3f %8.3f %7.3f q%s vgpr %s grid %s %s' % ((e[0]-t0)/1e6, (e[1]-t0)/1e6, (e[1]-e[0])/1e6, e[3], e[4], e[5], e[2]))
PY
find $O -name "*.csv" -size +1M -delete
