#!/bin/bash
# Round-6 profiles of the default bench command on one MI355X (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats                       -> gpurun_out/<tag>_kernel_stats.csv (+ printed summary)
#   2. rocprofv3 --pmc passes on the same step (counters only) -> gpurun_out/<tag>_pmc_mfma.json, <tag>_pmc_traffic.json
# Counter passes never combine --pmc with sys/runtime tracing (pool rule); FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ltx -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_prof_bench.json 2> $R/gpurun_out/${tag}_prof_bench.err
echo "rocprof stats rc=$?"
cp /tmp/prof/ltx_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
python $R/tools/step_trace.py /tmp/prof/ltx_kernel_trace.csv 4 $R/gpurun_out/${tag}_step_kernels.csv   # steady state: the last 4 of the 7 steps, cut at the optimiser launch
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/prof/ltx_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms (7 steps + setup): %.1f'%(tot/1e6))
for r in rows[:30]:
    n=r['Name'].replace('ftmi::','').replace('void ','')
    print('%-86s calls %5s tot %8.2f ms avg %8.1f us %5.1f%%'%(n[:86], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
i=0
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm$i -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > /tmp/pm$i.log 2>&1
  echo "pmc pass $i ($C) rc=$?"
done
python - <<PY
import csv, glob, collections, json
def cls(n):
    for key, k in (('gemm_nt16_kernel','gemm_nt16_kernel'),('gemm_nt_kernel','gemm_nt_kernel'),('skinny','gemm_nt_skinny (skinny4 + skinny2)'),('gemm_tn','gemm_tn_kernel'),('attn_fwd','attn_fwd_kernel'),
                   ('dkdv_pl','attn_bwd_dkdv_pl_kernel'),('dkdv','attn_bwd_dkdv_sq_kernel (cross-attention)'),('dq_pl','attn_bwd_dq_pl_kernel'),
                   ('bwd_dq','attn_bwd_dq_res / dq2 kernels (cross-attention)'),('norm_modulate','rowwise: norm_modulate'),('qknorm','rowwise: qknorm_rope'),('adamw','optimiser: adamw / sumsq'),('sumsq','optimiser: adamw / sumsq'),
                   ('lora_split','lora_split')):
        if key in n: return k
    return 'other (torch / misc)'
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for i in (1,2,3,4):
    f=glob.glob('/tmp/pm%d/*counter_collection.csv'%i)
    if not f: print('pass',i,'no counter file'); continue
    for r in csv.DictReader(open(f[0])):
        k=cls(r['Kernel_Name']); c=r['Counter_Name']+('' if i!=2 or r['Counter_Name']!='GRBM_GUI_ACTIVE' else '_pass2')
        agg[k][c]+=float(r['Counter_Value']); cnt[k][c]+=1
NSIMD=1024.0
# every NT GEMM launch of the step (the launch-site class bench.py's roofline object prices): 16x16x32 kernels + 32x32x16 kernels
for k in ('gemm_nt16_kernel','gemm_nt_kernel'):
    for c,x in agg.get(k,{}).items():
        agg['gemm_nt (all NT GEMM kernels)'][c]+=x; cnt['gemm_nt (all NT GEMM kernels)'][c]+=cnt[k][c]
out={'source':'tools/gpu_profile_r06.sh: rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof (3 steps in the trace), 1x MI355X',
     'units':'SQ_* summed over the chip per launch set; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per v_mfma_f32_16x16x32_bf16 -- the same FLOPs per busy cycle); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8; mfma_util = MFMA_BUSY / (1024 SIMDs x kernel cycles)'}
tot_mfma=tot_cyc=0.0
for k,v in agg.items():
    e={c:x for c,x in v.items()}
    gui=v.get('GRBM_GUI_ACTIVE',0.0)/8.0
    if gui>0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        e['launches_in_trace']=cnt[k]['GRBM_GUI_ACTIVE']
        e['kernel_cycles']=gui
        e['mfma_util']=v['SQ_VALU_MFMA_BUSY_CYCLES']/(NSIMD*gui)
        e['valu_issue_share_of_simd_time']=4.0*v.get('SQ_ACTIVE_INST_VALU',0.0)/(NSIMD*gui)
        e['wave_wait_share']=v.get('SQ_WAIT_ANY',0.0)/max(v.get('SQ_WAVE_CYCLES',1.0),1.0)
        e['wave_issue_stall_share']=v.get('SQ_WAIT_INST_ANY',0.0)/max(v.get('SQ_WAVE_CYCLES',1.0),1.0)
        if k!='gemm_nt (all NT GEMM kernels)': tot_mfma+=v['SQ_VALU_MFMA_BUSY_CYCLES']; tot_cyc+=gui
    out[k]=e
out['step']={'mfma_busy_cycles_all_kernels':tot_mfma,'kernel_cycles_all_kernels':tot_cyc,'mfma_util_over_kernel_time':tot_mfma/(NSIMD*tot_cyc) if tot_cyc else None,
             'note':'matrix-pipe busy cycles of every kernel of the step / (1024 SIMDs x summed kernel cycles); includes the 25 % all-ones row-sum MFMAs of the attention forward'}
json.dump(out,open('$R/gpurun_out/${tag}_pmc_mfma.json','w'),indent=1)
for k,v in out.items():
    if isinstance(v,dict) and 'mfma_util' in v: print('%-34s mfma_util %.3f  valu_issue %.3f  wait %.2f  issue_stall %.2f  launches %d'%(k,v['mfma_util'],v['valu_issue_share_of_simd_time'],v['wave_wait_share'],v['wave_issue_stall_share'],v['launches_in_trace']))
print('step', out['step'])
tr={'source':'tools/gpu_profile_r06.sh (FETCH_SIZE pass / WRITE_SIZE pass, counters only)','units':'bytes per launch; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (128-B fabric requests tallied at 64 B on gfx950 for wide coalesced reads); WRITE_SIZE (KB) as reported'}
for k,v in agg.items():
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        n=cnt[k]['FETCH_SIZE']
        tr[k]={'launches_in_trace':n,'fetch_kb_raw':v['FETCH_SIZE']/n,'write_kb_raw':v['WRITE_SIZE']/max(cnt[k]['WRITE_SIZE'],1),
               'hbm_bytes_per_launch':int((2*v['FETCH_SIZE']/n+v['WRITE_SIZE']/max(cnt[k]['WRITE_SIZE'],1))*1024)}
tot=sum(v['hbm_bytes_per_launch']*v['launches_in_trace'] for k,v in tr.items() if isinstance(v,dict) and k!='gemm_nt (all NT GEMM kernels)')
tr['step']={'hbm_bytes_per_step':tot/3.0,'steps_in_trace':3,'note':'sum over kernel classes of bytes per launch x launches, / 3 steps (set-up launches of the process included: < 1 %)'}
json.dump(tr,open('$R/gpurun_out/${tag}_pmc_traffic.json','w'),indent=1)
print({k:v.get('hbm_bytes_per_launch') for k,v in tr.items() if isinstance(v,dict)})
PY

# the other three workloads: whole-process kernel statistics of the shipped orchestration (C block stacks for CogVideoX / HunyuanVideo)
for wl in cogvideox wan hunyuan; do
  rm -rf /tmp/prof_$wl && mkdir -p /tmp/prof_$wl
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o w -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_${wl}_prof_bench.json 2> $R/gpurun_out/${tag}_${wl}_prof_bench.err
  echo "rocprof $wl rc=$?"
  cp /tmp/prof_$wl/w_kernel_stats.csv $R/gpurun_out/${tag}_${wl}_kernel_stats.csv
  python $R/tools/step_trace.py /tmp/prof_$wl/w_kernel_trace.csv 2 $R/gpurun_out/${tag}_${wl}_step_kernels.csv "mse_loss_kernel(" | head -n 14
done
