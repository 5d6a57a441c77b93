#!/bin/bash
# end-of-round check of the final tree: the whole -m gpu suite, smoke, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/r03t_suite.log 2>&1
echo "suite rc=$?"
grep -n "passed\|failed" $O/r03t_suite.log | tail -n 3
grep -n "^FAILED\|^ERROR" $O/r03t_suite.log | head -n 10
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03t_smoke.log 2>&1
echo "smoke rc=$?"; tail -n 2 $O/r03t_smoke.log
timeout 600 python bench.py > $O/r03t_bench_default.json 2> $O/r03t_bench_default.err
echo "bench rc=$?"; cut -c1-260 $O/r03t_bench_default.json
