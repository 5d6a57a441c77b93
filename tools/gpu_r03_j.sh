#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export FTMI_REPORT_DIR=gpurun_out
timeout 900 python -m pytest tests/test_gpu_wire.py -q -s -k feeder > gpurun_out/r03j_wire.log 2>&1; echo "wire rc=$?"
timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -s -k "fp8" > gpurun_out/r03j_hy.log 2>&1; echo "hy rc=$?"
timeout 1500 python bench.py --workload hunyuan --steps 5 --warmup 1 > gpurun_out/r03j_bench_hunyuan.json 2> gpurun_out/r03j_bench_hunyuan.err; echo "bench hunyuan rc=$?"
for f in wire hy; do echo "== $f"; grep -h "^\.\?\[\|passed\|failed\|^E " gpurun_out/r03j_$f.log | tail -10; done
cut -c1-3000 gpurun_out/r03j_bench_hunyuan.json; tail -n 3 gpurun_out/r03j_bench_hunyuan.err
