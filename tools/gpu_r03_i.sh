#!/bin/bash
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
