#!/bin/bash
# the Wan block as one C call per direction: against the Python composition, then the whole Wan file through it, then the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_wan.py -q -x -s -k "c_call" > $O/r03q_wan_c.log 2>&1; echo "c_call rc=$?"; tail -n 12 $O/r03q_wan_c.log
timeout 1500 python -m pytest tests/test_gpu_wan.py -q -x -s -k "not c_call" > $O/r03q_wan.log 2>&1; echo "wan rc=$?"
grep -n "^\.*\[wan\|passed\|failed\|Error" $O/r03q_wan.log | tail -n 14 | cut -c1-330
timeout 900 python bench.py --workload wan > $O/r03q_bench_wan.json 2> $O/r03q_bench_wan.err; echo "bench rc=$?"; cut -c1-300 $O/r03q_bench_wan.json
