#!/bin/bash
# One gpurun call: smoke + full-size bench (both GEMM variants) + rocprofv3 kernel stats.
mkdir -p gpurun_out
python -m finetrainers_amd.csrc.build > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
for v in 0 1; do
  timeout 900 python bench.py --steps 5 --warmup 2 --gemm-variant $v --no-cpu-baseline > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err
  echo "bench v$v rc=$?"; tail -c 1500 gpurun_out/bench_v$v.json; tail -3 gpurun_out/bench_v$v.err
done
