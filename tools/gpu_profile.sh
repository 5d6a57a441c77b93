#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summaries copied under gpurun_out/prof_* for profiles/.
mkdir -p gpurun_out
python -m finetrainers_amd.csrc.build > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ltx -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
echo "rocprof rc=$?"
find /tmp/prof -type f | head -20
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/prof_kernel_stats.csv; done
head -40 $R/gpurun_out/prof_kernel_stats.csv
tail -c 600 $R/gpurun_out/prof_bench.json
