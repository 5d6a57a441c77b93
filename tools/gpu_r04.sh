#!/bin/bash
# Round-4 GPU visits, one parametrised script: tools/gpu_r04.sh <step> [args]   (run through gpurun from the repository root)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export FTMI_REPORT_DIR=gpurun_out
step=$1; shift
case $step in
  a)  # first visit: store-path probe, the hand-placed GEMM pipelines against the shipped kernels, their parity tests
    timeout 120 tools/bin/probe_store > gpurun_out/r04a_probe_store.log 2>&1; echo "probe rc=$?"
    timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt or linear_lora" > gpurun_out/r04a_kernel_tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r04a_kernel_tests.log
    SHAPES=5376x2048x2048,5376x6144x2048,5376x8192x2048,5376x2048x8192,4096x4096x4096,8192x8192x8192 timeout 900 python tools/bench_gemm_ab.py 61,47,70,71,72 > gpurun_out/r04a_bench_gemm.log 2>&1
    LORA=1 timeout 600 python tools/bench_gemm_ab.py 61,47,70,72 > gpurun_out/r04a_bench_gemm_lora.log 2>&1
    cat gpurun_out/r04a_probe_store.log gpurun_out/r04a_bench_gemm.log gpurun_out/r04a_bench_gemm_lora.log
    bash tools/gpu_pmc_gemm.sh 5376 8192 2048 47,70 > gpurun_out/r04a_pmc_gemm.log 2>&1; cat gpurun_out/r04a_pmc_gemm.log | cut -c1-400
    ;;
  b)  # ablations of the hand-placed pipeline (stand-alone lab binary) + which vendor kernels run on the same shapes
    timeout 300 tools/bin/gemm_lab "5376x8192x2048,8192x8192x8192" 47,70,170,270,370,470,72,172,272,372 > gpurun_out/r04b_lab.log 2>&1; cat gpurun_out/r04b_lab.log
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/vend && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vend -o v -- python $GRAFT_REPO_ROOT/tools/probe_lib_gemm.py > /tmp/vend.log 2>&1
    python - <<PY
import csv
for r in csv.DictReader(open('/tmp/vend/v_kernel_stats.csv')):
    print(r['Name'][:400], r['Calls'], float(r['AverageNs'])/1e3)
PY
    ;;
  c)  # counters of the lab variants: cycles vs time (is a saving real cycles or a higher clock?)   args: shapes variants
    cd /tmp && export TMPDIR=/tmp
    for pass in 1 2; do
      case $pass in
        1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS";;
        2) C="SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE";;
      esac
      rm -rf /tmp/lc$pass
      LAB_FAST=1 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/lc$pass -o p -- $GRAFT_REPO_ROOT/tools/bin/gemm_lab "$1" "$2" > /tmp/lc_out$pass.log 2>&1 || tail -3 /tmp/lc_out$pass.log
      python - <<PY
import csv, collections, glob
f=glob.glob('/tmp/lc$pass/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
first="$C".split()[0]
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'gemm_nt' not in k: continue
    k=k[k.index('<'):k.index('>')+1]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']==first: cnt[k]+=1
for k,v in agg.items():
    print('%-46s n=%d '%(k,cnt[k])+' '.join('%s=%.4g'%(c.replace('SQ_','').replace('_sum',''),x/max(cnt[k],1)) for c,x in v.items()))
PY
    done
    python - <<PY
import csv, collections, glob
f=glob.glob('/tmp/lc1/*kernel_trace.csv')
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name']
    if 'gemm_nt' not in k: continue
    d[k[k.index('<'):k.index('>')+1]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print('%-46s mean duration %.1f us over %d'%(k, sum(v)/len(v), len(v)))
PY
    ;;
  d)  # parity of the 16 x 16 x 32 kernels (all epilogues, K-extension) + A/B with the fused LoRA extension and the fused epilogues
    timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt or linear_lora" > gpurun_out/r04d_kernel_tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r04d_kernel_tests.log
    LORA=1 timeout 600 python tools/bench_gemm_ab.py 61,42,47,80,86 > gpurun_out/r04d_bench_lora.log 2>&1
    EPI=gelu SHAPES=5376x8192x2048 timeout 600 python tools/bench_gemm_ab.py 61,80 > gpurun_out/r04d_bench_gelu.log 2>&1
    EPI=resid SHAPES=5376x2048x8192,5376x2048x2048 timeout 600 python tools/bench_gemm_ab.py 61,80,86 > gpurun_out/r04d_bench_resid.log 2>&1
    cat gpurun_out/r04d_bench_lora.log gpurun_out/r04d_bench_gelu.log gpurun_out/r04d_bench_resid.log
    ;;
  f)  # per-kernel steady-state step tables for two settings of one environment switch:  f VAR v1 v2
    cd /tmp && export TMPDIR=/tmp
    for v in $2 $3; do
      rm -rf /tmp/st_$v
      env $1=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/st_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > /tmp/st_$v.json 2> /tmp/st_$v.err
      python $GRAFT_REPO_ROOT/tools/step_trace.py /tmp/st_$v/t_kernel_trace.csv 4 $GRAFT_REPO_ROOT/gpurun_out/r04f_step_kernels_$1_$v.csv > /tmp/st_$v.txt 2>&1
      echo "== $1=$v"; head -30 /tmp/st_$v.txt
    done
    ;;
  t)  # the GPU suites other than the kernel / DiT parity files (those run in steps d and on their own), then smoke()
    timeout 1500 python -m pytest tests/test_gpu_wire.py tests/test_gpu_cogvideox.py tests/test_gpu_wan.py tests/test_gpu_hunyuan.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py tests/test_gpu_gemm_sk.py -q -m gpu > gpurun_out/r04t_suites.log 2>&1; echo "suites rc=$?"; tail -5 gpurun_out/r04t_suites.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04t_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r04t_smoke.log
    ;;
  T)  # the whole GPU suite as the driver runs it
    timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04T_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r04T_gpu_suite.log
    ;;
  e)  # evidence: default bench line (+ CPU baseline), --no-prof line, FTMI_NT16 A/B, rocprof statistics + counter passes, the other workloads
    timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r04_bench_default.json
    timeout 300 python bench.py --no-prof --no-cpu-baseline > gpurun_out/r04_bench_noprof.json 2> gpurun_out/r04_bench_noprof.err; echo "noprof rc=$?"; cut -c1-300 gpurun_out/r04_bench_noprof.json
    for rnd in 1 2; do for v in 0 3 7; do
      FTMI_NT16=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-prof --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FTMI_NT16=$v ms_per_step', d['ms_per_step'], d['step_ms_min_median_max'])"
    done; done > gpurun_out/r04_nt16_ab.txt 2>&1; cat gpurun_out/r04_nt16_ab.txt
    bash tools/gpu_profile_r04.sh r04 > gpurun_out/r04_profile.log 2>&1; tail -60 gpurun_out/r04_profile.log | cut -c1-260
    ;;
  g)  # GEMM evidence of the round in one visit: MFMA power probe, the lab table (round-3 256x256 8-wave kernel 47, hand-placed 32x32x16 pipeline 70,
      # 16x16x32 kernels 80 / 86, their ablations 180 = no DMA / 280 = no rendezvous / 380 = no fragment reads), cold activations, warm weights
    timeout 200 tools/bin/probe_mfma_power > gpurun_out/r04_gemm_power.txt 2>&1; cat gpurun_out/r04_gemm_power.txt
    SH="5376x2048x2048,5376x6144x2048,5376x8192x2048,5376x2048x8192,5376x2048x6144,8192x8192x8192"
    { echo "# weights cold (rotating copies), activations fixed (Infinity-Cache resident)"; timeout 300 tools/bin/gemm_lab "$SH" 47,70,80,86
      echo "# ablations of the 16x16x32 pipeline (results wrong on purpose): 180 no DMA in the loop, 280 no rendezvous, 380 no fragment reads"; timeout 200 tools/bin/gemm_lab "5376x8192x2048,8192x8192x8192" 80,180,280,380
      echo "# LAB_COLDX=1: activations rotate as well (what a step's launch sees: its input was written by the previous kernel, read once)"; LAB_COLDX=1 timeout 300 tools/bin/gemm_lab "$SH" 47,80,86
      echo "# LAB_WARMW=1: one weight copy (Infinity-Cache resident)"; LAB_WARMW=1 timeout 300 tools/bin/gemm_lab "5376x8192x2048,5376x2048x8192" 47,80,86
    } > gpurun_out/r04_gemm_lab.txt 2>&1; cat gpurun_out/r04_gemm_lab.txt
    ;;
  *) echo "unknown step $step"; exit 1;;
esac
