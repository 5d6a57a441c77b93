#!/bin/bash
# the test files the -x run of batch t did not reach (it stopped at the stream-K liveness test, reworked since)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 2400 python -m pytest tests/test_gpu_gemm_sk.py tests/test_gpu_hunyuan.py tests/test_gpu_kernels.py tests/test_gpu_wan.py tests/test_gpu_wire.py tests/test_gpu_fullsize.py -m gpu -q -s > $O/r03v_rest.log 2>&1
echo "rest rc=$?"; grep -n "\[sk\] hand-off\|passed\|failed\|xfail" $O/r03v_rest.log | tail -n 5; grep -n "^FAILED\|^ERROR" $O/r03v_rest.log | head
