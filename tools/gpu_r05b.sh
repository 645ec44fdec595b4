#!/bin/bash
# Round 5, batch b: kernel traces of the step (new default = tail reuse) with and without the streaming primal, tail-wave counts
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05b; mkdir -p $O
date
bash tools/gpu_trace.sh r05b new st10:default:DSDF_STREAM=1
for t in new st10; do python tools/step_timeline.py $O/trace_${t}_kernels.csv > $O/timeline_$t.md 2>/dev/null; done
date
bash tools/gpu_ab.sh r05b new tb2:default:DSDF_TAIL_BLOCKS=2 tb8:default:DSDF_TAIL_BLOCKS=8 tb16:default:DSDF_TAIL_BLOCKS=16 \
   pw7:default:DSDF_PRIMAL_WORKERS=7 pw6:default:DSDF_PRIMAL_WORKERS=6 new2
date
# PMC of the primal call: chunk kernel vs streaming kernel
P="timeout 300 rocprofv3 --kernel-trace --output-format csv"
for t in chunk stream; do
  E=""; [ $t = stream ] && E="DSDF_STREAM=1"
  env $E $P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_a_$t -o a -- python tools/ab_primal.py > $O/pmc_a_$t.log 2>&1
  env $E $P --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_b_$t -o b -- python tools/ab_primal.py > $O/pmc_b_$t.log 2>&1
  env $E $P --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_w_$t -o w -- python tools/ab_primal.py > $O/pmc_w_$t.log 2>&1
  env $E $P --pmc FETCH_SIZE -d $O/pmc_f_$t -o f -- python tools/ab_primal.py > $O/pmc_f_$t.log 2>&1
done
find $O -name "*.db" -delete
for f in $(find $O -name "*counter_collection.csv"); do
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:40]
    if not (k.startswith('void k_render') or k.startswith('k_tail')): continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
print(sys.argv[1].split('/')[-3] if len(sys.argv[1].split('/'))>3 else sys.argv[1])
for k, d in agg.items():
    print('  ', k, {c: f"{v / max(cnt[(k, c)],1):.4g}" for c, v in d.items()}, 'dispatches', max(cnt[(k, c)] for c in d))
PY
done
find $O -name "*.csv" -size +2M -delete
date
