#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PETAL_GRAM_FORM=5
for w in ${WEIGHTS:-1.36}; do
export PETAL_GRAM_OFFDIAG_COST=$w
bash dev/kt.sh g5 "k_gram|k_presplit|k_atb_f64" dev/gram4_bench.py short 2>&1 | grep "form="
python3 - $w <<'PY'
import csv,glob,collections,sys
for tag in ('g5',):
    f=glob.glob(f'gpurun_out/kt_{tag}/*/*_kernel_trace.csv')[0]
    by=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'k_gram' in n or 'k_presplit' in n:
            by[(n.split('(')[0][-26:], r.get('Grid_Size_X'), r.get('Grid_Size_Y'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,v in by.items(): print('w',sys.argv[1],k,len(v),'median %.1f us'%sorted(v)[len(v)//2], 'min %.1f'%min(v))
PY
rm -rf gpurun_out/kt_g5
done
