#!/bin/bash
# the test files the -x run of batch k did not reach + the new down-projection kernel / checkpointing tests, then in-step A/B of the new kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_gemm_sk.py -q -x > $O/r03l_sk.log 2>&1; echo "sk rc=$?"; tail -n 3 $O/r03l_sk.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s > $O/r03l_kernels.log 2>&1; echo "kernels rc=$?"; grep -n "lora\|passed\|failed" $O/r03l_kernels.log | tail -n 20
timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -x -s > $O/r03l_hy.log 2>&1; echo "hy rc=$?"; grep -n "hunyuan-\|passed\|failed\|Error" $O/r03l_hy.log | tail -n 20
timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_wire.py tests/test_gpu_dp.py -q -x > $O/r03l_wan.log 2>&1; echo "wan/wire/dp rc=$?"; tail -n 3 $O/r03l_wan.log
timeout 600 python -m pytest tests/test_gpu_dit.py -q -x -s -k "not full_depth and not 28-1" > $O/r03l_dit.log 2>&1; echo "dit rc=$?"; grep -n "LoRA-grad\|passed\|failed" $O/r03l_dit.log | tail -n 12
{
bash tools/ab_env.sh FTMI_SKINNY3 "0 1" 2
FTMI_SKINNY3_SPLIT=1 bash tools/ab_env.sh FTMI_SKINNY3 "1" 1
FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3 "1" 1
FTMI_SKINNY3_NST=4 bash tools/ab_env.sh FTMI_SKINNY3 "1" 1
} > $O/r03l_ab.log 2>&1
cat $O/r03l_ab.log
