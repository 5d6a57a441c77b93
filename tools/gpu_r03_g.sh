#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export FTMI_REPORT_DIR=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dit.py -q -s -x > gpurun_out/r03g_dit.log 2>&1; echo "dit rc=$?"
timeout 900 python -m pytest tests/test_gpu_cogvideox.py -q -s -k "block_forward or model_step" > gpurun_out/r03g_cog.log 2>&1; echo "cog rc=$?"
timeout 900 python -m pytest tests/test_gpu_wan.py -q -s -k "block_full or 1_3b or full_size or full_depth or model" > gpurun_out/r03g_wan.log 2>&1; echo "wan rc=$?"
timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -s -k "single_stream or model_and_step" > gpurun_out/r03g_hy.log 2>&1; echo "hy rc=$?"
for f in dit cog wan hy; do echo "== $f"; grep -h "^\[dit\]\|^\[cog-\|^\[wan-\|^\[hunyuan-\|passed\|failed\|Error\|error" gpurun_out/r03g_$f.log | grep -v "dit-trace\|dit-grad" | tail -40; done
