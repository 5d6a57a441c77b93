#!/bin/bash
# second look at the 64 x 128-tile down-projection kernel (explicit read scheduling; K split across the waves of a stage), checkpointing test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lora" > $O/r03m_lora_kw1.log 2>&1; echo "lora kw1 rc=$?"; tail -n 2 $O/r03m_lora_kw1.log
FTMI_SKINNY3_KW=0 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lora" > $O/r03m_lora_kw0.log 2>&1; echo "lora kw0 rc=$?"; tail -n 2 $O/r03m_lora_kw0.log
timeout 600 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "checkpointing" > $O/r03m_hy.log 2>&1; echo "hy rc=$?"; grep -n "hunyuan-\|passed\|failed\|Error" $O/r03m_hy.log | tail -n 8
timeout 600 python -m pytest tests/test_gpu_dit.py -q -x -k "2-2-3-4-6 or other_ranks" > $O/r03m_dit.log 2>&1; echo "dit rc=$?"; tail -n 2 $O/r03m_dit.log
{
bash tools/ab_env.sh FTMI_SKINNY3 "0 1" 1
FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3_KW "1 0" 1
FTMI_SKINNY3_NST=5 bash tools/ab_env.sh FTMI_SKINNY3_KW "1 0" 1
FTMI_SKINNY3_SPLIT=1 FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3_KW "1" 1
FTMI_SKINNY3_SPLIT=2 FTMI_SKINNY3_NST=3 bash tools/ab_env.sh FTMI_SKINNY3_KW "1" 1
} > $O/r03m_ab.log 2>&1
cat $O/r03m_ab.log
