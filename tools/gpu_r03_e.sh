#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "attention" > gpurun_out/r03e_attn_tests.log 2>&1; echo "attn tests rc=$?" | tee -a gpurun_out/r03e_attn_tests.log
timeout 600 python tools/bench_attn.py fwd "FTMI_ATTN_FWD8=0" "FTMI_ATTN_FWD8=1" "FTMI_ATTN_FWD8=2" > gpurun_out/r03e_bench_attn_fwd.log 2>&1
BHS=1,30,17776 timeout 600 python tools/bench_attn.py fwd "FTMI_ATTN_FWD8=0" "FTMI_ATTN_FWD8=1" "FTMI_ATTN_FWD8=2" > gpurun_out/r03e_bench_attn_fwd_cog.log 2>&1
tail -5 gpurun_out/r03e_attn_tests.log; cat gpurun_out/r03e_bench_attn_fwd.log gpurun_out/r03e_bench_attn_fwd_cog.log
