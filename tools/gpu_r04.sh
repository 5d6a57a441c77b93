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
  *) echo "unknown step $step"; exit 1;;
esac
