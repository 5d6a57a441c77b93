#!/bin/bash
# HBM traffic (PMC, separate passes) of the dominant kernel inside the real step: FETCH_SIZE / WRITE_SIZE per launch of gemm_nt_kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$C
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/tr_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f=glob.glob('/tmp/tr_$C/*counter_collection.csv')
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    n=r['Kernel_Name']
    k='gemm_nt_kernel' if 'gemm_nt_kernel' in n else 'gemm_nt_skinny' if 'skinny' in n else 'gemm_tn' if 'gemm_tn' in n else 'attn_fwd' if 'attn_fwd' in n else 'attn_bwd_dkdv' if 'dkdv' in n else 'attn_bwd_dq' if 'bwd_dq' in n else None
    if k is None: continue
    agg[k]+=float(r['Counter_Value']); cnt[k]+=1
for k in agg: print('$C %-16s launches %5d  avg per launch %.1f KB (raw counter units: KB)'%(k,cnt[k],agg[k]/cnt[k]))
PY
done
