#!/bin/bash
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
