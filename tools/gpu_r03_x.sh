#!/bin/bash
# stream-K share numbering: G - 1 - blockIdx (every hand-off wait on an earlier-dispatched workgroup; default) vs the XCD-contiguous one (FTMI_SK_ORDER=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 300 python -m pytest tests/test_gpu_gemm_sk.py -q -s > $O/r03x_sk1.log 2>&1; echo "order 1 tests rc=$?"; grep -n "hand-off\|passed\|failed\|xfail" $O/r03x_sk1.log | tail -n 3
FTMI_SK_ORDER=0 timeout 300 python -m pytest tests/test_gpu_gemm_sk.py -q -s > $O/r03x_sk0.log 2>&1; echo "order 0 tests rc=$?"; grep -n "hand-off\|passed\|failed\|xfail" $O/r03x_sk0.log | tail -n 3
echo "== order 1"; SHAPES=5376x2048x2048,5376x8192x2048,5376x2048x8192 timeout 300 python tools/bench_gemm_sk.py 61,60 2>/dev/null | tee $O/r03x_bench1.log
echo "== order 0"; FTMI_SK_ORDER=0 SHAPES=5376x2048x2048,5376x8192x2048,5376x2048x8192 timeout 300 python tools/bench_gemm_sk.py 60 2>/dev/null | tee $O/r03x_bench0.log
