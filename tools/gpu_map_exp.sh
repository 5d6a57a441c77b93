#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for gm in 1 2 4 8; do
  rm -rf /tmp/mx
  FTMI_MAP_GM=$gm timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/mx -o p -- python $R/tools/bench_gemm.py 7 > /tmp/mx_out.txt 2>&1
  grep "TF/s" /tmp/mx_out.txt | awk -v g=$gm '{print "gm="g, $1,$2,$3, $6,$7, $8,$9}'
  python - <<PY
import csv, glob, collections
f=glob.glob('/tmp/mx/*counter_collection.csv')
rows=[r for r in csv.DictReader(open(f[0])) if 'gemm_nt_kernel' in r['Kernel_Name']]
# 5 shapes x 24 launches in order
per=len(rows)//5
for i in range(5):
    v=[float(r['Counter_Value']) for r in rows[i*per:(i+1)*per]]
    print('   gm=$gm shape %d: FETCH avg %.1f MB (x2 corrected %.1f MB)'%(i, sum(v)/len(v)/1024, 2*sum(v)/len(v)/1024))
PY
done
