#!/usr/bin/env python3
"""Timeline of ONE two-stream step of bench.py from a `rocprofv3 --kernel-trace` CSV (tools/gpu_final.sh: <out>/trace/t_kernel_trace.csv):
start / end / duration of every dispatch between two consecutive k_backward_apply launches, per HIP queue, as a markdown table.
  python tools/step_timeline.py gpurun_out/r03/trace/t_kernel_trace.csv [step index from the end, default 2] > profiles/<tag>_step_timeline.md"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'], r['VGPR_Count'], r['Grid_Size_X'])
            for r in rows)
marks = [i for i, e in enumerate(ev) if e[2].startswith('k_backward_apply')]
a, b = marks[-back - 1], marks[-back]
t0 = ev[a][1]


def short(n):
    n = n.replace('void ', '')
    for cut in ('(', ):
        if cut in n and not n.startswith('k_render_items'):
            n = n[:n.index(cut)]
    if n.startswith('k_render_items'):
        n = n[:n.index('>') + 1]
    return n.replace('at::native::', '')[:60]


print(f"One step of `bench.py` (two-stream `dsdf.render_step`, 12 views, 256 / 64 spp) from the committed run's kernel trace: "
      f"{(ev[b][1] - t0) / 1e6:.2f} ms between the ends of two consecutive `k_backward_apply` launches.\n")
print("| start ms | end ms | ms | queue | kernel |")
print("|---|---|---|---|---|")
for e in ev[a + 1:b + 1]:
    print(f"| {(e[0] - t0) / 1e6:.3f} | {(e[1] - t0) / 1e6:.3f} | {(e[1] - e[0]) / 1e6:.3f} | {e[3]} | `{short(e[2])}` |")
