#!/bin/bash
# round 3, first GPU visit: stream-K GEMM parity, A/B micro-benchmark, step A/B
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
