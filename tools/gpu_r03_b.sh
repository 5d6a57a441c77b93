#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/sk_trace.py > gpurun_out/r03b_trace.log 2>&1
LORA=1 timeout 300 python tools/sk_trace.py > gpurun_out/r03b_trace_lora.log 2>&1
FTMI_SK_TAIL=1 timeout 300 python tools/sk_trace.py > gpurun_out/r03b_trace_tail1.log 2>&1
FTMI_SK_TAIL=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03b_bench_gemm_tail1.log 2>&1
cat gpurun_out/r03b_trace.log gpurun_out/r03b_trace_lora.log gpurun_out/r03b_trace_tail1.log gpurun_out/r03b_bench_gemm_tail1.log
