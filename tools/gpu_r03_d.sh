#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm_sk.py -q -s > gpurun_out/r03d_sk_tests.log 2>&1; echo "sk tests rc=$?" | tee -a gpurun_out/r03d_sk_tests.log
timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03d_bench_gemm.log 2>&1
FTMI_SK_TAIL=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03d_bench_gemm_tail1.log 2>&1
LORA=1 timeout 600 python tools/bench_gemm_sk.py 61,60 > gpurun_out/r03d_bench_gemm_lora.log 2>&1
timeout 300 python tools/sk_trace.py > gpurun_out/r03d_trace.log 2>&1
grep -v "^\[sk\]\|^$" gpurun_out/r03d_sk_tests.log | tail -8; grep "^\[sk\]" gpurun_out/r03d_sk_tests.log; cat gpurun_out/r03d_bench_gemm.log gpurun_out/r03d_bench_gemm_tail1.log gpurun_out/r03d_bench_gemm_lora.log; grep "==\|wg   [01] \|wg 100" gpurun_out/r03d_trace.log
