#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm_sk.py -q -s > gpurun_out/r03c_sk_tests.log 2>&1; echo "sk tests rc=$?" | tee -a gpurun_out/r03c_sk_tests.log
timeout 300 python tools/sk_trace.py > gpurun_out/r03c_trace.log 2>&1
timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03c_bench_gemm.log 2>&1
LORA=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03c_bench_gemm_lora.log 2>&1
FTMI_SK_EPI_COST=6 FTMI_SK_ACOST=3 FTMI_SK_PCOST=2 timeout 600 python tools/bench_gemm_sk.py 60 > gpurun_out/r03c_bench_gemm_costs2.log 2>&1
grep -v "^\[sk\]\|^$" gpurun_out/r03c_sk_tests.log | tail -15; grep "^\[sk\]" gpurun_out/r03c_sk_tests.log; cat gpurun_out/r03c_trace.log gpurun_out/r03c_bench_gemm.log gpurun_out/r03c_bench_gemm_lora.log gpurun_out/r03c_bench_gemm_costs2.log
