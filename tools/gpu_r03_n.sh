#!/bin/bash
# the single-stream block as one C call per direction: against the Python composition, then everything that runs through it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "c_call or checkpointing or fp8 or 2-2 or (single_stream and not 32640)" > $O/r03n_hy.log 2>&1; echo "hy rc=$?"
grep -n "hunyuan-\|passed\|failed\|Error\|assert" $O/r03n_hy.log | tail -n 25
