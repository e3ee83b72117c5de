#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PETAL_GRAM_FORM=5
for lib in petal-decomposition_amd/libpetal_hip.so dev/libpetal_nopark.so; do
export PETAL_HIP_LIBRARY=$PWD/$lib
rm -rf gpurun_out/kt_g5
bash dev/kt.sh g5 "k_gram" dev/gram4_bench.py short > /dev/null 2>&1
python3 - $lib <<'PY'
import csv,glob,collections,sys
f=glob.glob(f'gpurun_out/kt_g5/*/*_kernel_trace.csv')[0]
by=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'k_gram5' in n:
        by[(r.get('Grid_Size_X'), r.get('Grid_Size_Y'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in by.items(): print(sys.argv[1],k,len(v),'median %.1f us'%sorted(v)[len(v)//2])
PY
done
rm -rf gpurun_out/kt_g5
