#!/bin/bash
# PMC counters for a micro-benchmark command ("$@"); two SQ passes.  Output summarised per kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in 1 2 3; do
  case $pass in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES";;
    2) C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES";;
    3) C="GRBM_GUI_ACTIVE FETCH_SIZE";;
  esac
  rm -rf /tmp/pmc$pass
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc$pass -o p -- "$@" > /tmp/pmc_out$pass.log 2>&1; tail -2 /tmp/pmc_out$pass.log | cut -c1-200
  python - <<PY
import csv, collections, glob
f=glob.glob('/tmp/pmc$pass/*counter_collection.csv')
if not f: print('no counter file', glob.glob('/tmp/pmc$pass/*')); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'].replace('ftmi::','')[:48]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']=="$C".split()[0]: cnt[k]+=1
for k,v in agg.items():
    if 'at::' in k or 'rocclr' in k: continue
    print('%-48s n=%d '%(k,cnt[k])+' '.join('%s=%.3g'%(c.replace('SQ_',''),x/max(cnt[k],1)) for c,x in v.items()))
PY
done
