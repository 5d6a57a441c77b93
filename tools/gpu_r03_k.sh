#!/bin/bash
# whole GPU suite + smoke + the default bench line + the steady-state kernel table of the current tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/r03k_suite.log 2>&1
echo "suite rc=$?"
tail -n 30 $O/r03k_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03k_smoke.log 2>&1
echo "smoke rc=$?"; tail -n 3 $O/r03k_smoke.log
timeout 600 python bench.py > $O/r03k_bench_default.json 2> $O/r03k_bench_default.err
echo "bench rc=$?"; cut -c1-300 $O/r03k_bench_default.json
bash tools/gpu_r03_f.sh r03k
