#!/bin/bash
# where does the stream-K status word get raised?  suite-like order (other GPU tests first in the same process), twice; then the file alone
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_gemm_sk.py -q -s > $O/r03u_$i.log 2>&1; echo "run $i rc=$?"; grep -n "\[sk\] hand-off\|passed\|failed" $O/r03u_$i.log | tail -n 3; done
timeout 600 python -m pytest tests/test_gpu_gemm_sk.py -q -s > $O/r03u_3.log 2>&1; echo "alone rc=$?"; grep -n "\[sk\] hand-off\|passed\|failed" $O/r03u_3.log | tail -n 3
