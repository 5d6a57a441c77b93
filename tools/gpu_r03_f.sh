#!/bin/bash
# steady-state per-step kernel table of the default bench workload (rocprofv3 --kernel-trace, no counters)
tag=${1:-r03f}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ltx -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_prof_bench.json 2> $R/gpurun_out/${tag}_prof_bench.err
echo "rocprof rc=$?"
ls /tmp/prof | head
cp /tmp/prof/ltx_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
python $R/tools/step_trace.py /tmp/prof/ltx_kernel_trace.csv 6 $R/gpurun_out/${tag}_step_kernels.csv
cut -c1-400 $R/gpurun_out/${tag}_prof_bench.json
