#!/bin/bash
# dual-stream block as one C call per sample and direction; then the whole Hunyuan file (model parity at 2+2 and 20+40, fp8, checkpointing) and the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "c_call" > $O/r03o_hy_c.log 2>&1; echo "c_call rc=$?"; tail -n 4 $O/r03o_hy_c.log
timeout 1500 python -m pytest tests/test_gpu_hunyuan.py -q -x -s -k "not c_call" > $O/r03o_hy.log 2>&1; echo "hy rc=$?"
grep -n "hunyuan-\|passed\|failed\|Error\|assert" $O/r03o_hy.log | tail -n 25
timeout 900 python bench.py --workload hunyuan --steps 5 --warmup 1 > $O/r03o_bench_hunyuan.json 2> $O/r03o_bench_hunyuan.err; echo "bench rc=$?"; cut -c1-400 $O/r03o_bench_hunyuan.json
