#!/bin/bash
# Run on the GPU box (gpurun): the -m gpu test files one process each (a fault in one does not take the others down), then a short bench.
# usage: tools/gpu_suite.sh <tag> [files...]
tag=${1:-r02}; shift
files=${@:-"tests/test_gpu_kernels.py tests/test_gpu_dit.py tests/test_gpu_dp.py tests/test_gpu_fullsize.py"}
mkdir -p gpurun_out
export FTMI_REPORT_DIR=gpurun_out
for f in $files; do
  b=$(basename $f .py)
  timeout 1500 python -m pytest $f -m gpu -q -s -p no:cacheprovider > gpurun_out/${tag}_${b}.log 2>&1
  echo "== $f rc=$? : $(tail -1 gpurun_out/${tag}_${b}.log)"
  grep -E "FAILED|ERROR|^\[dit\]|^\[smoke\]|^\[step\]|^\[accumulate\]|^\[dp-|^\[ranges\]" gpurun_out/${tag}_${b}.log | head -40
done
