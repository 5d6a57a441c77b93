#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; compact summary printed, CSV saved to gpurun_out/.
mkdir -p gpurun_out
python -m finetrainers_amd.csrc.build > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ltx -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
echo "rocprof rc=$?"
cp /tmp/prof/ltx_kernel_stats.csv $R/gpurun_out/prof_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/prof/ltx_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms (7 steps + setup): %.1f'%(tot/1e6))
for r in rows[:28]:
    n=r['Name'].replace('ftmi::','').replace('void ','')
    print('%-78s calls %5s tot %8.2f ms avg %8.1f us %5.1f%%'%(n[:78], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
python -c "
import json; d=json.loads(open('$R/gpurun_out/prof_bench.json').read()); print('under rocprof: ms/step', d['ms_per_step'])"
