#!/bin/bash
# PMC passes over tools/pmc_gemm.py; args: M N K variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in 1 2 3 4; do
  case $pass in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS";;
    2) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD";;
    3) C="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum";;
    4) C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE";;
  esac
  rm -rf /tmp/pg$pass
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pg$pass -o p -- python $R/tools/pmc_gemm.py "$@" > /tmp/pg_out$pass.log 2>&1 || tail -3 /tmp/pg_out$pass.log | cut -c1-300
  python - <<PY
import csv, collections, glob
f=glob.glob('/tmp/pg$pass/*counter_collection.csv')
if not f: print('no counter file'); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
first="$C".split()[0]
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'gemm_nt' not in k: continue
    k=k[k.index('<'):k.index('>')+1]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']==first: cnt[k]+=1
for k,v in agg.items():
    print('%-44s n=%d '%(k,cnt[k])+' '.join('%s=%.4g'%(c.replace('SQ_','').replace('_sum',''),x/max(cnt[k],1)) for c,x in v.items()))
PY
done
