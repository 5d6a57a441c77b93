#!/bin/bash
# round-3 evidence run: the whole -m gpu suite (with the parity lines), smoke, the default bench line, rocprofv3 stats + counter passes, the other workloads' stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -s --durations=10 > $O/r03p_suite.log 2>&1
echo "suite rc=$?"
grep -n "passed\|failed" $O/r03p_suite.log | tail -n 3
grep -n "^FAILED\|^ERROR" $O/r03p_suite.log | head -n 10
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03p_smoke.log 2>&1
echo "smoke rc=$?"; tail -n 2 $O/r03p_smoke.log
timeout 600 python bench.py > $O/r03p_bench_default.json 2> $O/r03p_bench_default.err
echo "bench rc=$?"; cut -c1-260 $O/r03p_bench_default.json
bash tools/gpu_profile_r03.sh r03p > $O/r03p_profile.log 2>&1
echo "profile rc=$?"; tail -n 45 $O/r03p_profile.log
timeout 600 python bench.py --workload hunyuan --gradient-checkpointing --steps 3 --warmup 1 --no-cpu-baseline > $O/r03p_bench_hunyuan_ckpt.json 2> $O/r03p_bench_hunyuan_ckpt.err
echo "hunyuan ckpt rc=$?"; cut -c1-200 $O/r03p_bench_hunyuan_ckpt.json
