#!/bin/bash
# Round-3 GPU visits as they were issued through gpurun, folded into one parametrised script:  tools/gpu_r03.sh <step>
# (steps a..x in the order of the round; p = the evidence run: whole -m gpu suite, smoke, default bench line, tools/gpu_profile_r03.sh)
step=$1; shift
case $step in
  a)  # round 3, first GPU visit: stream-K GEMM parity, A/B micro-benchmark, step A/B
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    export FTMI_REPORT_DIR=gpurun_out
    timeout 900 python -m pytest tests/test_gpu_gemm_sk.py -x -q -s > gpurun_out/r03a_sk_tests.log 2>&1; echo "sk tests rc=$?" | tee -a gpurun_out/r03a_sk_tests.log
    timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or lora" > gpurun_out/r03a_kernel_tests.log 2>&1; echo "kernel tests rc=$?" | tee -a gpurun_out/r03a_kernel_tests.log
    timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03a_bench_gemm.log 2>&1
    LORA=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03a_bench_gemm_lora.log 2>&1
    FTMI_SK=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a_bench_sk0.json 2> gpurun_out/r03a_bench_sk0.err
    FTMI_SK=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a_bench_sk1.json 2> gpurun_out/r03a_bench_sk1.err
    tail -3 gpurun_out/r03a_sk_tests.log; cat gpurun_out/r03a_bench_gemm.log gpurun_out/r03a_bench_gemm_lora.log; cat gpurun_out/r03a_bench_sk0.json gpurun_out/r03a_bench_sk1.json | cut -c1-600
    ;;
  b)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    timeout 300 python tools/sk_trace.py > gpurun_out/r03b_trace.log 2>&1
    LORA=1 timeout 300 python tools/sk_trace.py > gpurun_out/r03b_trace_lora.log 2>&1
    FTMI_SK_TAIL=1 timeout 300 python tools/sk_trace.py > gpurun_out/r03b_trace_tail1.log 2>&1
    FTMI_SK_TAIL=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03b_bench_gemm_tail1.log 2>&1
    cat gpurun_out/r03b_trace.log gpurun_out/r03b_trace_lora.log gpurun_out/r03b_trace_tail1.log gpurun_out/r03b_bench_gemm_tail1.log
    ;;
  c)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    timeout 900 python -m pytest tests/test_gpu_gemm_sk.py -q -s > gpurun_out/r03c_sk_tests.log 2>&1; echo "sk tests rc=$?" | tee -a gpurun_out/r03c_sk_tests.log
    timeout 300 python tools/sk_trace.py > gpurun_out/r03c_trace.log 2>&1
    timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03c_bench_gemm.log 2>&1
    LORA=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03c_bench_gemm_lora.log 2>&1
    FTMI_SK_EPI_COST=6 FTMI_SK_ACOST=3 FTMI_SK_PCOST=2 timeout 600 python tools/bench_gemm_sk.py 60 > gpurun_out/r03c_bench_gemm_costs2.log 2>&1
    grep -v "^\[sk\]\|^$" gpurun_out/r03c_sk_tests.log | tail -15; grep "^\[sk\]" gpurun_out/r03c_sk_tests.log; cat gpurun_out/r03c_trace.log gpurun_out/r03c_bench_gemm.log gpurun_out/r03c_bench_gemm_lora.log gpurun_out/r03c_bench_gemm_costs2.log
    ;;
  d)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    timeout 900 python -m pytest tests/test_gpu_gemm_sk.py -q -s > gpurun_out/r03d_sk_tests.log 2>&1; echo "sk tests rc=$?" | tee -a gpurun_out/r03d_sk_tests.log
    timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03d_bench_gemm.log 2>&1
    FTMI_SK_TAIL=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03d_bench_gemm_tail1.log 2>&1
    LORA=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03d_bench_gemm_lora.log 2>&1
    timeout 300 python tools/sk_trace.py > gpurun_out/r03d_trace.log 2>&1
    grep -v "^\[sk\]\|^$" gpurun_out/r03d_sk_tests.log | tail -8; grep "^\[sk\]" gpurun_out/r03d_sk_tests.log; cat gpurun_out/r03d_bench_gemm.log gpurun_out/r03d_bench_gemm_tail1.log gpurun_out/r03d_bench_gemm_lora.log; grep "==\|wg   [01] \|wg 100" gpurun_out/r03d_trace.log
    ;;
  e)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "attention" > gpurun_out/r03e_attn_tests.log 2>&1; echo "attn tests rc=$?" | tee -a gpurun_out/r03e_attn_tests.log
    timeout 600 python tools/bench_attn.py fwd "FTMI_ATTN_FWD8=0" "FTMI_ATTN_FWD8=1" "FTMI_ATTN_FWD8=2" > gpurun_out/r03e_bench_attn_fwd.log 2>&1
    BHS=1,30,17776 timeout 600 python tools/bench_attn.py fwd "FTMI_ATTN_FWD8=0" "FTMI_ATTN_FWD8=1" "FTMI_ATTN_FWD8=2" > gpurun_out/r03e_bench_attn_fwd_cog.log 2>&1
    tail -5 gpurun_out/r03e_attn_tests.log; cat gpurun_out/r03e_bench_attn_fwd.log gpurun_out/r03e_bench_attn_fwd_cog.log
    ;;
  f)  # steady-state per-step kernel table of the default bench workload (rocprofv3 --kernel-trace, no counters)
    tag=${1:-r03f}
    R=${GRAFT_REPO_ROOT:-/root/repo}
    mkdir -p $R/gpurun_out
    cd /tmp && export TMPDIR=/tmp
    rm -rf /tmp/prof && mkdir -p /tmp/prof
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ltx -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_prof_bench.json 2> $R/gpurun_out/${tag}_prof_bench.err
    echo "rocprof rc=$?"
    ls /tmp/prof | head
    cp /tmp/prof/ltx_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
    python $R/tools/step_trace.py /tmp/prof/ltx_kernel_trace.csv 6 $R/gpurun_out/${tag}_step_kernels.csv
    cut -c1-400 $R/gpurun_out/${tag}_prof_bench.json
    ;;
  g)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    export FTMI_REPORT_DIR=gpurun_out
    timeout 1500 python -m pytest tests/test_gpu_dit.py -q -s -x > gpurun_out/r03g_dit.log 2>&1; echo "dit rc=$?"
    timeout 900 python -m pytest tests/test_gpu_cogvideox.py -q -s -k "block_forward or model_step" > gpurun_out/r03g_cog.log 2>&1; echo "cog rc=$?"
    timeout 900 python -m pytest tests/test_gpu_wan.py -q -s -k "block_full or 1_3b or full_size or full_depth or model" > gpurun_out/r03g_wan.log 2>&1; echo "wan rc=$?"
    timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -s -k "single_stream or model_and_step" > gpurun_out/r03g_hy.log 2>&1; echo "hy rc=$?"
    for f in dit cog wan hy; do echo "== $f"; grep -h "^\[dit\]\|^\[cog-\|^\[wan-\|^\[hunyuan-\|passed\|failed\|Error\|error" gpurun_out/r03g_$f.log | grep -v "dit-trace\|dit-grad" | tail -40; done
    ;;
  h)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    export FTMI_REPORT_DIR=gpurun_out
    timeout 900 python -m pytest tests/test_gpu_wire.py -q -s > gpurun_out/r03h_wire.log 2>&1; echo "wire rc=$?"
    timeout 900 python -m pytest tests/test_gpu_cogvideox.py -q -s -k "rank32 or two_ranks" > gpurun_out/r03h_cog.log 2>&1; echo "cog rc=$?"
    timeout 1500 python -m pytest tests/test_gpu_hunyuan.py -q -s > gpurun_out/r03h_hy.log 2>&1; echo "hy rc=$?"
    timeout 900 python -m pytest tests/test_gpu_wan.py -q -s -k "block_full or 1_3b or two_ranks or model" > gpurun_out/r03h_wan.log 2>&1; echo "wan rc=$?"
    for w in cogvideox wan; do timeout 900 python bench.py --workload $w --steps 10 --warmup 2 > gpurun_out/r03h_bench_$w.json 2> gpurun_out/r03h_bench_$w.err; echo "bench $w rc=$?"; done
    for f in wire cog hy wan; do echo "== $f"; grep -h "^\.\?\[\|passed\|failed\|^E " gpurun_out/r03h_$f.log | tail -25; done
    cut -c1-1500 gpurun_out/r03h_bench_cogvideox.json; cut -c1-1500 gpurun_out/r03h_bench_wan.json; tail -3 gpurun_out/r03h_bench_*.err
    ;;
  i)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    export FTMI_REPORT_DIR=gpurun_out
    timeout 900 python -m pytest tests/test_gpu_wire.py -q -s -k feeder > gpurun_out/r03i_wire.log 2>&1; echo "wire rc=$?"
    timeout 1500 python -m pytest tests/test_gpu_hunyuan.py -q -s -k "fp8 or single_stream or model_and_step" > gpurun_out/r03i_hy.log 2>&1; echo "hy rc=$?"
    timeout 900 python -m pytest tests/test_gpu_wan.py -q -s -k "two_ranks" > gpurun_out/r03i_wan.log 2>&1; echo "wan rc=$?"
    timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "mse_loss" > gpurun_out/r03i_k.log 2>&1; echo "mse rc=$?"
    timeout 1200 python bench.py --workload hunyuan --steps 5 --warmup 1 > gpurun_out/r03i_bench_hunyuan.json 2> gpurun_out/r03i_bench_hunyuan.err; echo "bench hunyuan rc=$?"
    for f in wire hy wan k; do echo "== $f"; grep -h "^\.\?\[\|passed\|failed\|^E " gpurun_out/r03i_$f.log | grep -v "W924\|Gloo" | tail -16; done
    cut -c1-2500 gpurun_out/r03i_bench_hunyuan.json; tail -n 3 gpurun_out/r03i_bench_hunyuan.err
    ;;
  j)
    cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
    mkdir -p gpurun_out
    export FTMI_REPORT_DIR=gpurun_out
    timeout 900 python -m pytest tests/test_gpu_wire.py -q -s -k feeder > gpurun_out/r03j_wire.log 2>&1; echo "wire rc=$?"
    timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -s -k "fp8" > gpurun_out/r03j_hy.log 2>&1; echo "hy rc=$?"
    timeout 1500 python bench.py --workload hunyuan --steps 5 --warmup 1 > gpurun_out/r03j_bench_hunyuan.json 2> gpurun_out/r03j_bench_hunyuan.err; echo "bench hunyuan rc=$?"
    for f in wire hy; do echo "== $f"; grep -h "^\.\?\[\|passed\|failed\|^E " gpurun_out/r03j_$f.log | tail -10; done
    cut -c1-3000 gpurun_out/r03j_bench_hunyuan.json; tail -n 3 gpurun_out/r03j_bench_hunyuan.err
    ;;
  k)  # whole GPU suite + smoke + the default bench line + the steady-state kernel table of the current tree
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/r03k_suite.log 2>&1
    echo "suite rc=$?"
    tail -n 30 $O/r03k_suite.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03k_smoke.log 2>&1
    echo "smoke rc=$?"; tail -n 3 $O/r03k_smoke.log
    timeout 600 python bench.py > $O/r03k_bench_default.json 2> $O/r03k_bench_default.err
    echo "bench rc=$?"; cut -c1-300 $O/r03k_bench_default.json
    bash tools/gpu_r03.sh f r03k
    ;;
  l)  # the test files the -x run of batch k did not reach + the new down-projection kernel / checkpointing tests, then in-step A/B of the new kernel
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 600 python -m pytest tests/test_gpu_gemm_sk.py -q -x > $O/r03l_sk.log 2>&1; echo "sk rc=$?"; tail -n 3 $O/r03l_sk.log
    timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s > $O/r03l_kernels.log 2>&1; echo "kernels rc=$?"; grep -n "lora\|passed\|failed" $O/r03l_kernels.log | tail -n 20
    timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -x -s > $O/r03l_hy.log 2>&1; echo "hy rc=$?"; grep -n "hunyuan-\|passed\|failed\|Error" $O/r03l_hy.log | tail -n 20
    timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_wire.py tests/test_gpu_dp.py -q -x > $O/r03l_wan.log 2>&1; echo "wan/wire/dp rc=$?"; tail -n 3 $O/r03l_wan.log
    timeout 600 python -m pytest tests/test_gpu_dit.py -q -x -s -k "not full_depth and not 28-1" > $O/r03l_dit.log 2>&1; echo "dit rc=$?"; grep -n "LoRA-grad\|passed\|failed" $O/r03l_dit.log | tail -n 12
    {
    bash tools/ab_env.sh FTMI_SKINNY3 "0 1" 2
    FTMI_SKINNY3_SPLIT=1 bash tools/ab_env.sh FTMI_SKINNY3 "1" 1
    FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3 "1" 1
    FTMI_SKINNY3_NST=4 bash tools/ab_env.sh FTMI_SKINNY3 "1" 1
    } > $O/r03l_ab.log 2>&1
    cat $O/r03l_ab.log
    ;;
  m)  # second look at the 64 x 128-tile down-projection kernel (explicit read scheduling; K split across the waves of a stage), checkpointing test
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lora" > $O/r03m_lora_kw1.log 2>&1; echo "lora kw1 rc=$?"; tail -n 2 $O/r03m_lora_kw1.log
    FTMI_SKINNY3_KW=0 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lora" > $O/r03m_lora_kw0.log 2>&1; echo "lora kw0 rc=$?"; tail -n 2 $O/r03m_lora_kw0.log
    timeout 600 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "checkpointing" > $O/r03m_hy.log 2>&1; echo "hy rc=$?"; grep -n "hunyuan-\|passed\|failed\|Error" $O/r03m_hy.log | tail -n 8
    timeout 600 python -m pytest tests/test_gpu_dit.py -q -x -k "2-2-3-4-6 or other_ranks" > $O/r03m_dit.log 2>&1; echo "dit rc=$?"; tail -n 2 $O/r03m_dit.log
    {
    bash tools/ab_env.sh FTMI_SKINNY3 "0 1" 1
    FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3_KW "1 0" 1
    FTMI_SKINNY3_NST=5 bash tools/ab_env.sh FTMI_SKINNY3_KW "1 0" 1
    FTMI_SKINNY3_SPLIT=1 FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3_KW "1" 1
    FTMI_SKINNY3_SPLIT=2 FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3_KW "1" 1
    } > $O/r03m_ab.log 2>&1
    cat $O/r03m_ab.log
    ;;
  n)  # the single-stream block as one C call per direction: against the Python composition, then everything that runs through it
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "c_call or checkpointing or fp8 or 2-2 or (single_stream and not 32640)" > $O/r03n_hy.log 2>&1; echo "hy rc=$?"
    grep -n "hunyuan-\|passed\|failed\|Error\|assert" $O/r03n_hy.log | tail -n 25
    ;;
  o)  # dual-stream block as one C call per sample and direction; then the whole Hunyuan file (model parity at 2+2 and 20+40, fp8, checkpointing) and the bench line
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "c_call" > $O/r03o_hy_c.log 2>&1; echo "c_call rc=$?"; tail -n 4 $O/r03o_hy_c.log
    timeout 1500 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "not c_call" > $O/r03o_hy.log 2>&1; echo "hy rc=$?"
    grep -n "hunyuan-\|passed\|failed\|Error\|assert" $O/r03o_hy.log | tail -n 25
    timeout 900 python bench.py --workload hunyuan --steps 5 --warmup 1 > $O/r03o_bench_hunyuan.json 2> $O/r03o_bench_hunyuan.err; echo "bench rc=$?"; cut -c1-400 $O/r03o_bench_hunyuan.json
    ;;
  p)  # round-3 evidence run: the whole -m gpu suite (with the parity lines), smoke, the default bench line, rocprofv3 stats + counter passes, the other workloads' stats
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 2400 python -m pytest tests -m gpu -q -s --durations=10 > $O/r03p_suite.log 2>&1
    echo "suite rc=$?"
    grep -n "passed\|failed" $O/r03p_suite.log | tail -n 3
    grep -n "^FAILED\|^ERROR" $O/r03p_suite.log | head -n 10
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03p_smoke.log 2>&1
    echo "smoke rc=$?"; tail -n 2 $O/r03p_smoke.log
    timeout 600 python bench.py > $O/r03p_bench_default.json 2> $O/r03p_bench_default.err
    echo "bench rc=$?"; cut -c1-260 $O/r03p_bench_default.json
    bash tools/gpu_profile_r03.sh r03p > $O/r03p_profile.log 2>&1
    echo "profile rc=$?"; tail -n 45 $O/r03p_profile.log
    timeout 600 python bench.py --workload hunyuan --gradient-checkpointing --steps 3 --warmup 1 --no-cpu-baseline > $O/r03p_bench_hunyuan_ckpt.json 2> $O/r03p_bench_hunyuan_ckpt.err
    echo "hunyuan ckpt rc=$?"; cut -c1-200 $O/r03p_bench_hunyuan_ckpt.json
    ;;
  q)  # the Wan block as one C call per direction: against the Python composition, then the whole Wan file through it, then the bench line
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 600 python -m pytest tests/test_gpu_wan.py -q -x -s -k "c_call" > $O/r03q_wan_c.log 2>&1; echo "c_call rc=$?"; tail -n 12 $O/r03q_wan_c.log
    timeout 1500 python -m pytest tests/test_gpu_wan.py -q -x -s -k "not c_call" > $O/r03q_wan.log 2>&1; echo "wan rc=$?"
    grep -n "^\.*\[wan\|passed\|failed\|Error" $O/r03q_wan.log | tail -n 14 | cut -c1-330
    timeout 900 python bench.py --workload wan > $O/r03q_bench_wan.json 2> $O/r03q_bench_wan.err; echo "bench rc=$?"; cut -c1-300 $O/r03q_bench_wan.json
    ;;
  r)  # same-box A/B: Wan and HunyuanVideo steps with the blocks issued from Python (FTMI_NATIVE_BLOCKS=0) and as C calls
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    cd $R
    for wl in wan hunyuan; do
      st=6; [ $wl = hunyuan ] && st=3
      for nb in 0 1 0 1; do
        echo -n "$wl FTMI_NATIVE_BLOCKS=$nb  "
        FTMI_NATIVE_BLOCKS=$nb timeout 600 python bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
    import json,sys
    d=json.loads(sys.stdin.read()); print('ms/step %.1f'%d['ms_per_step'], 'min/med/max', d.get('step_ms_min_median_max'), 'peak GiB %.1f'%d['peak_memory_gib'])"
        [ $wl = hunyuan ] && [ $nb = 1 ] && break
      done
    done > $O/r03r_ab.log 2>&1
    cat $O/r03r_ab.log
    ;;
  s)  # CogVideoX 1.5 architecture (patch_size_t, ofs) against the oracle, next to the 2b / 5b cases of the same test
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    cd $R
    timeout 900 python -m pytest tests/test_gpu_cogvideox.py -q -x -s -k "model_step_parity and (1.5 or True-2-True or False-2-True)" > $O/r03s_cog.log 2>&1; echo "cog rc=$?"
    grep -n "cog-model\|cog-step\|passed\|failed\|Error" $O/r03s_cog.log | tail -n 14 | cut -c1-300
    ;;
  t)  # end-of-round check of the final tree: the whole -m gpu suite, smoke, the default bench line
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    mkdir -p $O
    cd $R
    timeout 2400 python -m pytest tests -m gpu -q -x > $O/r03t_suite.log 2>&1
    echo "suite rc=$?"
    grep -n "passed\|failed" $O/r03t_suite.log | tail -n 3
    grep -n "^FAILED\|^ERROR" $O/r03t_suite.log | head -n 10
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03t_smoke.log 2>&1
    echo "smoke rc=$?"; tail -n 2 $O/r03t_smoke.log
    timeout 600 python bench.py > $O/r03t_bench_default.json 2> $O/r03t_bench_default.err
    echo "bench rc=$?"; cut -c1-260 $O/r03t_bench_default.json
    ;;
  u)  # where does the stream-K status word get raised?  suite-like order (other GPU tests first in the same process), twice; then the file alone
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    cd $R
    for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_gemm_sk.py -q -s > $O/r03u_$i.log 2>&1; echo "run $i rc=$?"; grep -n "\[sk\] hand-off\|passed\|failed" $O/r03u_$i.log | tail -n 3; done
    timeout 600 python -m pytest tests/test_gpu_gemm_sk.py -q -s > $O/r03u_3.log 2>&1; echo "alone rc=$?"; grep -n "\[sk\] hand-off\|passed\|failed" $O/r03u_3.log | tail -n 3
    ;;
  v)  # the test files the -x run of batch t did not reach (it stopped at the stream-K liveness test, reworked since)
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    cd $R
    timeout 2400 python -m pytest tests/test_gpu_gemm_sk.py tests/test_gpu_hunyuan.py tests/test_gpu_kernels.py tests/test_gpu_wan.py tests/test_gpu_wire.py tests/test_gpu_fullsize.py -m gpu -q -s > $O/r03v_rest.log 2>&1
    echo "rest rc=$?"; grep -n "\[sk\] hand-off\|passed\|failed\|xfail" $O/r03v_rest.log | tail -n 5; grep -n "^FAILED\|^ERROR" $O/r03v_rest.log | head
    ;;
  x)  # stream-K share numbering: G - 1 - blockIdx (every hand-off wait on an earlier-dispatched workgroup; default) vs the XCD-contiguous one (FTMI_SK_ORDER=0)
    R=${GRAFT_REPO_ROOT:-/root/repo}
    O=$R/gpurun_out
    cd $R
    timeout 300 python -m pytest tests/test_gpu_gemm_sk.py -q -s > $O/r03x_sk1.log 2>&1; echo "order 1 tests rc=$?"; grep -n "hand-off\|passed\|failed\|xfail" $O/r03x_sk1.log | tail -n 3
    FTMI_SK_ORDER=0 timeout 300 python -m pytest tests/test_gpu_gemm_sk.py -q -s > $O/r03x_sk0.log 2>&1; echo "order 0 tests rc=$?"; grep -n "hand-off\|passed\|failed\|xfail" $O/r03x_sk0.log | tail -n 3
    echo "== order 1"; SHAPES=5376x2048x2048,5376x8192x2048,5376x2048x8192 timeout 300 python tools/bench_gemm_sk.py 61,60 2>/dev/null | tee $O/r03x_bench1.log
    echo "== order 0"; FTMI_SK_ORDER=0 SHAPES=5376x2048x2048,5376x8192x2048,5376x2048x8192 timeout 300 python tools/bench_gemm_sk.py 60 2>/dev/null | tee $O/r03x_bench0.log
    ;;
  *) echo "unknown step $step"; exit 1;;
esac
